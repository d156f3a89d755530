# Round-2 profile run (on the GPU box, from the repo root):  bash tools/run_profile_r02.sh
# kernel-trace stats of the default bench, then the counter passes (each in its own run, no --stats / trace domains beside
# --pmc): FETCH_SIZE, WRITE_SIZE, and the matrix-pipe pass (SQ_VALU_MFMA_BUSY_CYCLES with SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
# GRBM_GUI_ACTIVE).  Summaries are copied to profiles/ by hand afterwards (tools/pmc_summary.py writes the JSON).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02prof
mkdir -p $O
python3 bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err
cat $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_under_rocprof.json 2> $O/prof.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> $O/pmc_w.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp > /dev/null 2> $O/pmc_m.err
echo mfma done
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp (c2, bf16x3, B=1; tools/run_profile_r02.sh)" $O/pmc_mfma > $O/pmc_summary.txt
cat $O/pmc_summary.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -25 $O/kernel_stats.csv
