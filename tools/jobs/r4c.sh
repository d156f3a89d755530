O=gpurun_out/r4c
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fine or c1 or c2 or b2 or pipelined or deferred" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/stamps_fine.py > $O/stamps.txt 2>&1; grep -E "total cycles|q gemm|kv gemm|gather|mlp0|merge|LN|mlp2|end|H store" $O/stamps.txt
timeout -k 10 120 python tools/time_fine.py > $O/time_fine.txt 2>&1; tail -4 $O/time_fine.txt
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 300 --warmup 10 --main-region-only --no-cpu-baseline > $O/bench_$rep.json 2> $O/bench_$rep.err || exit 1
python - <<PY
import json
d=json.loads(open("$O/bench_$rep.json").read().strip().splitlines()[-1]); print("value", round(d["value"],1))
PY
done
