O=gpurun_out/r3u
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kpt or coarse_match" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
print("$name", "value", round(d["value"],1), "pnp_ceiling", round(d["host"]["pnp_ceiling_fps"]))
PY
}
for rep in 1 2 3 4; do
run base_$rep OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_base.so || exit 1
run kpt65_$rep OPHIP_X=0 || exit 1
run s64_$rep OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_s64.so || exit 1
done
