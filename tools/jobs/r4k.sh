O=gpurun_out/r4k
mkdir -p $O
for rep in 1 2; do
OPHIP_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b100_$rep.json 2> $O/b100_$rep.err || exit 1
python - <<PY
import json
d=json.loads(open("$O/b100_$rep.json").read().strip().splitlines()[-1]); print("100 steps:", {k: (round(d[k],1) if isinstance(d.get(k), float) else d.get(k)) for k in ("value","value_lazy_conf","value_pnp_adaptive","value_matcher_only","value_matcher_only_object_cached")})
PY
grep "host us" $O/b100_$rep.err
done
