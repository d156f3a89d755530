# host PnP builds on the box's CPU: previous library, this tree with the 256-bit scorer, this tree (AVX-512 when the CPU has it); then driver-style lines
O=gpurun_out/r04pnp2
mkdir -p $O
grep -o "avx512[a-z_0-9]*" /proc/cpuinfo | sort -u | tr '\n' ' ' > $O/cpu_flags.txt; nproc >> $O/cpu_flags.txt
for r in 1 2; do
OPPNP_LIB=onepose_st_amd/lib/variants/libonepose_pnp_old.so python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
OPPNP_NO_AVX512=1 python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
done
for i in 1 2; do
  for v in "OPPNP_NO_AVX512=1" "OPPNP_NO_AVX512="; do
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench20_${v}_$i.json 2> $O/bench20_$i.err || exit 1
  python - <<PY
import json
d = json.loads(open("$O/bench20_${v}_$i.json").read().strip().splitlines()[-1])
h = d.get("host", {})
print("$v run $i value", round(d["value"], 1), "pnp_ceiling", round(h.get("pnp_ceiling_fps") or 0))
PY
  done
done
