export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
print("$name", "value", round(d["value"],1))
PY
}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_masked.py -m gpu -q -x -k "coarse_match or c2 or c1 or lazy or b2 or c4" > $O/tests_u2.log 2>&1; tail -3 $O/tests_u2.log
for rep in 1 2 3 4; do
run u3_$rep OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_u3.so || exit 1
run u2_$rep OPHIP_X=0 || exit 1
done
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench1.json 2> $O/prof1.err || exit 1
python3 tools/timeline.py $O/trace1 1 > $O/timeline1.txt 2>&1
find $O/trace1 -name "*.csv" -size +3M -delete
tail -12 $O/timeline1.txt
