mkdir -p gpurun_out/r3h
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3h/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3h/tests.log; tail -8 gpurun_out/r3h/tests.log
timeout -k 10 120 python tools/time_fine.py > gpurun_out/r3h/time_fine_pair.txt 2>&1; cat gpurun_out/r3h/time_fine_pair.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3h/smoke.txt 2>&1; tail -3 gpurun_out/r3h/smoke.txt
