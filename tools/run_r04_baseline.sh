# Round-4 baseline of the round-3 build on today's box: 20-step bench lines (driver's flags), host trace, kernel trace of a 20-step run
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
cat $O/bench20.json
OPHIP_BENCH_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench20_main.json 2> $O/bench20_main.err
tail -3 $O/bench20_main.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace20 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/b.json 2> $O/b.err
echo trace done
python3 tools/time_fine.py > $O/time_fine.txt 2>&1
cat $O/time_fine.txt
python3 tools/time_coarse.py > $O/time_coarse.txt 2>&1
cat $O/time_coarse.txt
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "eight_input_shapes or failed_enqueue or frame_op or pipelined" > $O/pytest_new.txt 2>&1 || true
tail -5 $O/pytest_new.txt
