O=gpurun_out/u1
mkdir -p $O
for b in 1 2 4; do
timeout -k 10 300 python bench.py --steps 150 --warmup 10 --main-region-only --no-cpu-baseline --batch $b > $O/b_$b.json 2> $O/b_$b.err
python - <<PY
import json
d=json.loads(open("$O/b_$b.json").read().strip().splitlines()[-1]); print("batch", $b, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "attn us", round(d["roofline"]["avg_launch_ms"]*1e3,1), d["host"].get("cgroup_cpu_throttled_in_timed_region"), "ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
done
for b in 1 2 4; do
timeout -k 10 300 python bench.py --steps 150 --warmup 10 --main-region-only --no-cpu-baseline --batch $b --no-pnp > $O/bn_$b.json 2> $O/bn_$b.err
python - <<PY
import json
d=json.loads(open("$O/bn_$b.json").read().strip().splitlines()[-1]); print("no-pnp batch", $b, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3))
PY
done
