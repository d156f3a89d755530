O=gpurun_out/t2
mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do
OPHIP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --main-region-only --no-cpu-baseline > $O/b20_$rep.json 2> $O/b20_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b20_$rep.json").read().strip().splitlines()[-1]); print("value @20", round(d["value"],1), "setup steps", d["setup_steps_untimed"])
PY
grep "host us" $O/b20_$rep.err | tail -1
done
