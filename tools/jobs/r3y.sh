O=gpurun_out/r3y
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
OPHIP_BENCH_TRACE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_$rep.json 2> $O/b20_$rep.err || exit 1
grep "host us" $O/b20_$rep.err | tail -1
python - <<PY
import json
d=json.loads(open("$O/b20_$rep.json").read().strip().splitlines()[-1]); print("20 steps:", {k: (round(d[k],1) if isinstance(d.get(k), float) else d.get(k)) for k in ("value","value_lazy_conf","lazy_conf_frames_rerun_eagerly","value_matcher_only","value_matcher_only_object_cached","value_pnp_adaptive")})
PY
done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/b100.json 2> $O/b100.err
python - <<PY
import json
d=json.loads(open("$O/b100.json").read().strip().splitlines()[-1]); print("100 steps:", round(d["value"],1), d["host"])
PY
