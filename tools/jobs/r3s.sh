O=gpurun_out/r3s
mkdir -p $O
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench.json 2> $O/bench.err; grep "host us" $O/bench.err
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline --depth 4 > $O/bench4.json 2> $O/bench4.err; grep "host us" $O/bench4.err
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline --no-pnp > $O/benchnp.json 2> $O/benchnp.err; grep "host us" $O/benchnp.err
python - <<PY
import json
for n in ("bench","bench4","benchnp"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, round(d["value"],1))
PY
