O=gpurun_out/r3t
mkdir -p $O
timeout -k 10 500 python bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","value_matcher_only","value_matcher_only_object_cached","value_lazy_conf","value_pnp_adaptive","roofline","host"): print(k, d.get(k))
PY
