O=gpurun_out/r3x
mkdir -p $O
for st in 20 100; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --steps $st --warmup 5 --main-region-only --no-cpu-baseline --conf-matrix lazy > $O/lazy_$st.json 2> $O/lazy_$st.err; grep "host us" $O/lazy_$st.err
python - <<PY
import json
d=json.loads(open("$O/lazy_$st.json").read().strip().splitlines()[-1]); print("lazy steps", $st, round(d["value"],1))
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20.json 2> $O/b20.err
python - <<PY
import json
d=json.loads(open("$O/b20.json").read().strip().splitlines()[-1]); print({k: d.get(k) for k in ("value","value_lazy_conf","lazy_conf_frames_rerun_eagerly","value_matcher_only")})
PY
