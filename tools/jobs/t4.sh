O=gpurun_out/t4
mkdir -p $O
for rep in 1 2 3; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 100 --warmup 5 --main-region-only --no-cpu-baseline > $O/b_$rep.json 2> $O/b_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1]); print("value", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"))
PY
grep "cpu over" $O/b_$rep.err
done
