O=gpurun_out/t1
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -6 $O/tests.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/b_$rep.json 2> $O/b_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1]); print("value", round(d["value"],1))
PY
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --main-region-only --no-cpu-baseline > $O/b20_$rep.json 2> $O/b20_$rep.err
python - <<PY
import json
d=json.loads(open("$O/b20_$rep.json").read().strip().splitlines()[-1]); print("value @20", round(d["value"],1))
PY
done
