export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
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
run base_$rep OPHIP_X=0 || exit 1
run prepnt_$rep OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_nt.so || exit 1
done
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench1.json 2> $O/prof1.err || exit 1
python3 tools/timeline.py $O/trace1 1 > $O/timeline1.txt 2>&1
find $O/trace1 -name "*.csv" -size +3M -delete
tail -36 $O/timeline1.txt
