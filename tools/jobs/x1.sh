export TMPDIR=/tmp
O=gpurun_out/x2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/soak_pipeline.py 1500 5 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -1 $O/soak.txt
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); print("$name", "value", round(d["value"],1), d["host"].get("cgroup_cpu_throttled_in_timed_region"))
PY
}
for rep in 1 2 3 4; do
run pp0_$rep OPHIP_FRAME_PINGPONG=0 || exit 1
run pp1_$rep OPHIP_X=0 || exit 1
done
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench1.json 2> $O/prof1.err || exit 1
python3 tools/timeline.py $O/trace1 2 > $O/timeline1.txt 2>&1
find $O/trace1 -name "*.csv" -size +3M -delete
tail -64 $O/timeline1.txt
