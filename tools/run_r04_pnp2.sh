# host PnP builds on the box's CPU: previous library (lib/variants/libonepose_pnp_old.so, built by hand from revision 7c61c77), this tree with the
# 256-bit scorer, this tree (AVX-512 when the CPU has it); then the GPU tests, driver-style lines and a 100-step line
O=gpurun_out/r04pnp2
mkdir -p $O
rm -f $O/time_pnp.txt
for r in 1 2; do
[ -f onepose_st_amd/lib/variants/libonepose_pnp_old.so ] && OPPNP_LIB=onepose_st_amd/lib/variants/libonepose_pnp_old.so python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
OPPNP_NO_AVX512=1 python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
python tools/time_pnp.py 12 | tee -a $O/time_pnp.txt
done
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench20_$i.json 2> $O/bench20_$i.err || exit 1
  python - <<PY
import json
d = json.loads(open("$O/bench20_$i.json").read().strip().splitlines()[-1])
h = d.get("host", {})
print("run $i value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "matcher_only", round(d.get("value_matcher_only") or 0), "pnp_ceiling", round(h.get("pnp_ceiling_fps") or 0), "c2_hard", round(d.get("value_c2_hard") or 0), "c3_b32", round(d.get("value_c3_b32") or 0), "c4", round(d.get("value_c4") or 0))
PY
done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-side-legs > $O/bench100.json 2> $O/bench100.err && python -c "
import json; d=json.loads(open('$O/bench100.json').read().strip().splitlines()[-1]); print('100 steps value', round(d['value'],1), 'matcher only', round(d.get('value_matcher_only') or 0))"
