mkdir -p gpurun_out/r3d
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "fine" > gpurun_out/r3d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/tests.log; tail -5 gpurun_out/r3d/tests.log
timeout -k 10 120 python tools/stamps_fine.py > gpurun_out/r3d/stamps_fine_pair.txt 2>&1; cat gpurun_out/r3d/stamps_fine_pair.txt
OPHIP_FINE_PAIR=0 timeout -k 10 120 python tools/stamps_fine.py > gpurun_out/r3d/stamps_fine_single.txt 2>&1; cat gpurun_out/r3d/stamps_fine_single.txt
