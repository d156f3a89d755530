O=gpurun_out/t9
mkdir -p $O
for thr in 14 10; do
for rep in 1 2; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 300 --warmup 5 --main-region-only --no-cpu-baseline --pnp-threads $thr > $O/b_${thr}_$rep.json 2> $O/b_${thr}_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b_${thr}_$rep.json").read().strip().splitlines()[-1]); print("threads", $thr, "value", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"))
PY
grep "cpu over" $O/b_${thr}_$rep.err
done
done
