# Round-3 profile run (on the GPU box, from the repo root):  bash tools/run_profile_r03.sh
# bench line, kernel-trace stats of the contract's region alone, then the counter passes (each in its own run, never with --stats or
# trace domains beside --pmc): FETCH_SIZE, WRITE_SIZE, the matrix-pipe pass, and an L2 pass (TCC hit / miss / request counters) for the
# fine stage's weight stream.  Summaries are copied to profiles/ by hand afterwards (tools/pmc_summary.py writes the JSON and records
# the library's build stamp, which bench.py checks before quoting the counters).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03prof
mkdir -p $O
python3 bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err
cat $O/bench.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_under_rocprof.json 2> $O/prof.err
echo stats done
python3 tools/timeline.py $O/stats 2 > $O/timeline.txt 2>&1 || true
tail -12 $O/timeline.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_w.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_m.err
echo mfma done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_l2.err || echo "L2 pass failed (counter names?)"
echo l2 done
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only (c2, bf16x3, B=1; tools/run_profile_r03.sh)" $O/pmc_mfma > $O/pmc_summary.txt
cat $O/pmc_summary.txt
python3 tools/l2_summary.py $O/pmc_l2 > $O/l2_summary.txt 2>&1 || true
cat $O/l2_summary.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -30 $O/kernel_stats.csv
python3 tools/stamps_x3.py > $O/stamps_enc_x3w8.txt 2>&1
python3 tools/stamps_fine.py > $O/stamps_fine_pair.txt 2>&1
python3 tools/time_coarse.py > $O/time_coarse.txt 2>&1
# the N > 1 code path on this one-GPU box: two gloo ranks sharing the device (pinned CPU slices, per-rank PnP pools, broadcast)
python3 bench.py --gpus 2 --share-device --dist-backend gloo --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_2ranks_gloo_shared.json 2> $O/bench_2ranks.err || echo "2-rank rehearsal failed"
tail -c 1200 $O/bench_2ranks_gloo_shared.json
# the RCCL code path itself (process group on the device, broadcast of weights + object block, barriers, max / min over ranks) with ONE rank
OPHIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || echo "RCCL rehearsal failed"
tail -c 400 $O/bench_rccl_1rank.json
find $O -name "*.csv" -size +3M -delete
