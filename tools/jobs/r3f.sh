mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests/test_gpu_loftr.py -q -x > gpurun_out/r3f/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/tests.log; tail -40 gpurun_out/r3f/tests.log
