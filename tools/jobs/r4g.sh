O=gpurun_out/r4g
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fine or c1 or c2 or b2" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/time_fine.py > $O/time_fine.txt 2>&1; tail -2 $O/time_fine.txt
timeout -k 10 120 python tools/stamps_fine.py > $O/stamps.txt 2>&1; grep -E "total cycles|q gemm|kv gemm|gather|end" $O/stamps.txt
