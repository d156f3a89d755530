mkdir -p gpurun_out/r3k
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3k/tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3k/tests.log; tail -6 gpurun_out/r3k/tests.log
if [ $rc -eq 0 ]; then
for rep in 1 2 3; do
for m in 1 0; do
OPHIP_FRAME_DEFER_FINE=$m timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > gpurun_out/r3k/bench_defer${m}_$rep.json 2> gpurun_out/r3k/bench_defer${m}_$rep.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3k/bench_defer${m}_$rep.json").read().strip().splitlines()[-1])
print("defer", $m, "rep", $rep, "value", d["value"], "matcher_only", d.get("value_matcher_only"), "host", d.get("host"))
PY
done
done
fi
