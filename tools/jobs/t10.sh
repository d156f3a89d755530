O=gpurun_out/t11
mkdir -p $O
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --main-region-only --no-cpu-baseline > $O/b20_$rep.json 2> $O/b20_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b20_$rep.json").read().strip().splitlines()[-1]); print("value @20", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"), "threads", d["host"]["pnp_threads_per_rank"], "confined", d["host"].get("confined_to_cpus"), "ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
done
grep "cpu over" $O/b20_12.err
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/b_$rep.json 2> $O/b_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1]); print("value @300", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"))
PY
done
