O=gpurun_out/r4m
mkdir -p $O
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("100 steps:", {k: (round(d[k],1) if isinstance(d.get(k), float) else d.get(k)) for k in ("value","value_lazy_conf","value_pnp_adaptive","value_matcher_only","value_matcher_only_object_cached")}, d["cpu_baseline"]["value"], d["roofline"]["traffic"], d["roofline"]["counters_from"], d["roofline"]["mfma_busy_frac_at_peak_clock"])
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; python - <<PY
import json
d=json.loads(open("$O/bench20.json").read().strip().splitlines()[-1]); print("20 steps:", round(d["value"],1), d["roofline"]["traffic"], d["roofline"]["frac"])
PY
