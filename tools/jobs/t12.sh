O=gpurun_out/t12
mkdir -p $O
for thr in 12 10; do
for rep in 1 2 3 4 5 6 7 8 9 10; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --main-region-only --no-cpu-baseline --pnp-threads $thr > $O/b20_${thr}_$rep.json 2> $O/b20_${thr}_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b20_${thr}_$rep.json").read().strip().splitlines()[-1]); print("threads", $thr, "value @20", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"), "ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
done
done
