set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/v5
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/v5/pytest.log 2>&1
tail -3 gpurun_out/v5/pytest.log
python bench.py --steps 100 --warmup 10 > gpurun_out/v5/bench.json 2> gpurun_out/v5/bench.err
cat gpurun_out/v5/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v5/stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/v5/bench_prof.json 2> gpurun_out/v5/prof.err
cat gpurun_out/v5/bench_prof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/v5/pmc_fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> gpurun_out/v5/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/v5/pmc_write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> gpurun_out/v5/pmc_w.err
echo done
