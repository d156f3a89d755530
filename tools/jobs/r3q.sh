export TMPDIR=/tmp
O=gpurun_out/r3q
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
for rep in 1 2 3; do
run B_$rep OPHIP_X=0 || exit 1
run Bnt_$rep OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_nt.so || exit 1
run A_$rep OPHIP_EXP_FINE_SIDE=1 || exit 1
run Ant_$rep OPHIP_EXP_FINE_SIDE=1 OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_nt.so || exit 1
run inorder_$rep OPHIP_FRAME_DEFER_FINE=0 || exit 1
done
