# host PnP change check on the box: host tests, GPU pose-parity tests, three 20-step driver-style lines, the hard workload's line
O=gpurun_out/r04pnp
mkdir -p $O
python -m pytest tests/test_pnp_host.py -q > $O/pnp_host.txt 2>&1; tail -2 $O/pnp_host.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -k "hard or pose or pnp or full_forward or pipeline" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench20_$i.json 2> $O/bench20_$i.err || exit 1
  python - <<PY
import json
d = json.loads(open("$O/bench20_$i.json").read().strip().splitlines()[-1])
h = d.get("host", {})
print("run $i value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "matcher_only", d.get("value_matcher_only"), "pnp_ceiling", h.get("pnp_ceiling_fps"), "c2_hard", d.get("value_c2_hard"))
PY
done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-side-legs > $O/bench100.json 2> $O/bench100.err && python -c "
import json; d=json.loads(open('$O/bench100.json').read().strip().splitlines()[-1]); print('100 steps value', round(d['value'],1))"
