mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "fine or custom_ops or frame_op or c1_ or planted or kv" > gpurun_out/r3c/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/tests.log; tail -15 gpurun_out/r3c/tests.log
timeout -k 10 120 python tools/time_fine.py > gpurun_out/r3c/time_fine_pair.txt 2>&1
OPHIP_FINE_PAIR=0 timeout -k 10 120 python tools/time_fine.py > gpurun_out/r3c/time_fine_single.txt 2>&1
cat gpurun_out/r3c/time_fine_pair.txt gpurun_out/r3c/time_fine_single.txt
timeout -k 10 120 python tools/stamps_x3.py > gpurun_out/r3c/stamps_enc_x3w8.txt 2>&1; cat gpurun_out/r3c/stamps_enc_x3w8.txt | head -12
timeout -k 10 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3c/bench200.json 2> gpurun_out/r3c/bench200.err; tail -c 900 gpurun_out/r3c/bench200.json
