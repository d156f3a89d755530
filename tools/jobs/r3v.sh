O=gpurun_out/r3v
mkdir -p $O
OPHIP_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo rc=$?
tail -c 900 $O/bench_rccl1.json; tail -5 $O/bench_rccl1.err
