# final check of the round-4 end build (host-side changes after the last counter passes; the HIP library is unchanged, so
# profiles/r04_pmc.json stays valid): GPU tests, smoke, the driver's command, host traces, the N > 1 rehearsals
O=gpurun_out/r04final
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
for i in 1 2; do
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench20_$i.json 2> $O/bench20_$i.err || exit 1
  python3 - <<PY
import json
d = json.loads(open("$O/bench20_$i.json").read().strip().splitlines()[-1])
h = d.get("host", {})
print("run $i value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "matcher_only", round(d.get("value_matcher_only") or 0), "pnp_ceiling", round(h.get("pnp_ceiling_fps") or 0), "c2_hard", round(d.get("value_c2_hard") or 0), "c3_b32", round(d.get("value_c3_b32") or 0), "c4", round(d.get("value_c4") or 0), "traffic", d["roofline"].get("traffic"), "cpu", d["cpu_baseline"]["value"])
PY
done
for i in 1 2 3; do OPHIP_BENCH_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only --no-side-legs > $O/trace20_$i.json 2> $O/trace20_$i.err; grep -h "host us/step" $O/trace20_$i.err | tail -1 | cut -c1-200; done > $O/host_trace.txt; cat $O/host_trace.txt
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-side-legs > $O/bench100.json 2> $O/bench100.err && python3 -c "
import json; d=json.loads(open('$O/bench100.json').read().strip().splitlines()[-1]); print('100 steps value', round(d['value'],1), 'matcher only', round(d.get('value_matcher_only') or 0))"
python3 bench.py --gpus 2 --share-device --dist-backend gloo --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_2ranks_gloo_shared.json 2> $O/bench_2ranks.err || echo "2-rank rehearsal failed"
tail -c 300 $O/bench_2ranks_gloo_shared.json; echo
OPHIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || echo "RCCL rehearsal failed"
tail -c 300 $O/bench_rccl_1rank.json; echo
