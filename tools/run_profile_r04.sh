# Round-4 profile run (on the GPU box, from the repo root):  bash tools/run_profile_r04.sh
# Same recipe as round 3 (bench lines, kernel-trace stats of the contract's region alone, counter passes each in its own run, stamps) plus:
# the driver's own command with the side legs (configs 3, 4, c2_hard), a kernel-trace summary of BASELINE config 4, the fine stage's stamps
# with ONE workgroup per CU beside the shipped two, a marker (roctx) trace sample, and the input kernels / selection stand-alone.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
tail -c 1500 $O/bench20.json
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-legs > $O/bench.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_under_rocprof.json 2> $O/prof.err
echo stats done
python3 tools/timeline.py $O/stats 2 > $O/timeline.txt 2>&1 || true
tail -14 $O/timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 bench.py --workload c4 --steps 16 --warmup 4 --no-cpu-baseline --main-region-only --no-side-legs --roofline-kernel conf > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
echo c4 stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_w.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_m.err
echo mfma done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_l2.err || echo "L2 pass failed"
echo l2 done
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only (c2, bf16x3, B=1; tools/run_profile_r04.sh)" $O/pmc_mfma > $O/pmc_summary.txt
cat $O/pmc_summary.txt
python3 tools/l2_summary.py $O/pmc_l2 > $O/l2_summary.txt 2>&1 || true
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv
head -24 $O/kernel_stats.csv
head -16 $O/c4_kernel_stats.csv
python3 tools/stamps_x3.py > $O/stamps_enc_x3w8.txt 2>&1
python3 tools/stamps_fine.py > $O/stamps_fine_pair.txt 2>&1
OPHIP_FINE_LDS_PAD=40000 python3 tools/stamps_fine.py > $O/stamps_fine_pair_one_wg_per_cu.txt 2>&1
python3 tools/time_coarse.py > $O/time_coarse.txt 2>&1
python3 tools/time_fine.py > $O/time_fine.txt 2>&1
python3 tools/time_inputs.py > $O/time_inputs.txt 2>&1
python3 tools/time_kernel.py sim_stats stat_combine conf select select_place > $O/time_kernel.txt 2>&1
cat $O/time_inputs.txt $O/time_kernel.txt $O/time_fine.txt
# roctx ranges of the default path beside the kernel trace (marker trace only: no counters in this run)
OPHIP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/marker -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pnp --main-region-only --no-side-legs > /dev/null 2> $O/marker.err || echo "marker trace failed"
find $O/marker -name "*marker*csv" | head -1 | xargs -I{} sh -c 'head -40 {} > gpurun_out/r04prof/marker_sample.csv' || true
# the N > 1 code path on this one-GPU box: two gloo ranks sharing the device, and the RCCL path itself with ONE rank
python3 bench.py --gpus 2 --share-device --dist-backend gloo --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_2ranks_gloo_shared.json 2> $O/bench_2ranks.err || echo "2-rank rehearsal failed"
tail -c 900 $O/bench_2ranks_gloo_shared.json
OPHIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || echo "RCCL rehearsal failed"
tail -c 600 $O/bench_rccl_1rank.json
find $O -name "*.csv" -size +3M -delete
