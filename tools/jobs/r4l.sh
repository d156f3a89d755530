O=gpurun_out/r4l
mkdir -p $O
timeout -k 10 300 python tools/soak_pipeline.py 600 1 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak.txt
timeout -k 10 400 python tools/fuzz_parity.py 7 16 > $O/fuzz7.txt 2>&1; echo "fuzz rc=$?"; grep -c " ok" $O/fuzz7.txt; grep -E "MISMATCH|mismatches|Error|error" $O/fuzz7.txt | head
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("100 steps:", {k: (round(d[k],1) if isinstance(d.get(k), float) else d.get(k)) for k in ("value","value_lazy_conf","value_pnp_adaptive","value_matcher_only","value_matcher_only_object_cached")}, d["cpu_baseline"]["value"])
PY
