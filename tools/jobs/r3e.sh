mkdir -p gpurun_out/r3e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -k "lazy or coarse or frame or c3 or pipelin" > gpurun_out/r3e/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/tests.log; tail -25 gpurun_out/r3e/tests.log
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3e/bench200.json 2> gpurun_out/r3e/bench200.err; tail -c 300 gpurun_out/r3e/bench200.err; python -c "
import json; d=json.loads(open('gpurun_out/r3e/bench200.json').read().strip().splitlines()[-1]); print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('value')}, d['host'])"
