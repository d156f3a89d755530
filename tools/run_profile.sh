set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/v7
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/v7/pytest.log 2>&1
tail -3 gpurun_out/v7/pytest.log
python bench.py --steps 100 --warmup 10 > gpurun_out/v7/bench.json 2> gpurun_out/v7/bench.err
cat gpurun_out/v7/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v7/stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/v7/bench_prof.json 2> gpurun_out/v7/prof.err
cat gpurun_out/v7/bench_prof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/v7/pmc_fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> gpurun_out/v7/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/v7/pmc_write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> gpurun_out/v7/pmc_w.err
echo done
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --with-backbone > gpurun_out/v7/bench_with_backbone.json 2> gpurun_out/v7/bench_bb.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v7/stats_bb -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline --with-backbone > /dev/null 2> gpurun_out/v7/prof_bb.err
echo done2
