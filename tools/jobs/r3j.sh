mkdir -p gpurun_out/r3j
timeout -k 10 600 python -m pytest tests/test_gpu_masked.py -m gpu -q -x -s > gpurun_out/r3j/masked.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3j/masked.log; tail -25 gpurun_out/r3j/masked.log
if [ $rc -eq 0 ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3j/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3j/tests.log; tail -6 gpurun_out/r3j/tests.log
fi
