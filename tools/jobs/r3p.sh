export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
for m in 1 0; do
OPHIP_FRAME_DEFER_FINE=$m timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_defer${m}_$rep.json 2> $O/bench_defer${m}_$rep.err || exit 1
python - <<PY
import json
d=json.loads(open("$O/bench_defer${m}_$rep.json").read().strip().splitlines()[-1])
print("defer", $m, "rep", $rep, "value", round(d["value"],1), "pnp_ceiling", round(d["host"]["pnp_ceiling_fps"]), "host_bound", d["host"]["host_bound"])
PY
done
done
export OPHIP_FRAME_DEFER_FINE=1
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench1.json 2> $O/prof1.err || exit 1
python3 tools/timeline.py $O/trace1 1 > $O/timeline1.txt 2>&1
find $O/trace1 -name "*.csv" -size +3M -delete
tail -42 $O/timeline1.txt
