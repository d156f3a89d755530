export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for rep in 1 2 3; do
for m in 0 -1; do
OPHIP_EXP_MAIN_PRIO=$m timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_prio${m}_$rep.json 2> $O/bench_prio${m}_$rep.err || exit 1
python - <<PY
import json
d=json.loads(open("$O/bench_prio${m}_$rep.json").read().strip().splitlines()[-1])
print("main prio", $m, "rep", $rep, "value", round(d["value"],1), "pnp_ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
done
done
export OPHIP_EXP_MAIN_PRIO=-1
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench1.json 2> $O/prof1.err || exit 1
python3 tools/timeline.py $O/trace1 1 > $O/timeline1.txt 2>&1
find $O/trace1 -name "*.csv" -size +3M -delete
tail -38 $O/timeline1.txt
