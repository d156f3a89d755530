O=gpurun_out/r3o
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deferred or pipelined or dropped or frame_call or lazy" > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -5 $O/tests.log
