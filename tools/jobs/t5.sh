O=gpurun_out/t8
mkdir -p $O
for rep in 1 2 3 4 5 6 7 8 9 10; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --main-region-only --no-cpu-baseline > $O/b20_$rep.json 2> $O/b20_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b20_$rep.json").read().strip().splitlines()[-1]); print("value @20", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"), "pinned", d["host"]["pinned"], d["host"].get("pnp_workers_pinned"), d["host"]["cpus_of_rank0"], "ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
done
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/b_$rep.json 2> $O/b_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1]); print("value @300", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"))
PY
done
