mkdir -p gpurun_out/r3b
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b/bench20.json 2> gpurun_out/r3b/bench20.err
timeout -k 10 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3b/bench200.json 2> gpurun_out/r3b/bench200.err
tail -c 1500 gpurun_out/r3b/bench200.json
