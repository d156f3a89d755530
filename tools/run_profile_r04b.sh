# Round-4 counter passes + marker trace on the SHIPPED build (run after the last library change; the bench line quotes counters only
# from a pmc summary whose library_build_stamp is the running library's):  bash tools/run_profile_r04b.sh
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_l2 $O/marker
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_w.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_m.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only > /dev/null 2> $O/pmc_l2.err || echo "L2 pass failed"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only (c2, bf16x3, B=1; tools/run_profile_r04b.sh)" $O/pmc_mfma > $O/pmc_summary.txt
cat $O/pmc_summary.txt
python3 tools/l2_summary.py $O/pmc_l2 > $O/l2_summary.txt 2>&1 || true
OPHIP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/marker -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pnp --main-region-only --no-side-legs > /dev/null 2> $O/marker.err || echo "marker trace failed"
ls $O/marker/*/
for f in $O/marker/*/*marker*.csv; do head -60 $f > $O/marker_sample.csv; wc -l $f; done
python3 bench.py --steps 20 --warmup 5 > $O/bench20_final.json 2> $O/bench20_final.err || true
tail -c 600 $O/bench20_final.json
find $O -name "*.csv" -size +3M -delete
# where a 20-step region's time goes on the host side (per-step enqueue / wait / finish / submit and the tail after the last frame)
for i in 1 2 3; do OPHIP_BENCH_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only --no-side-legs > $O/trace20_$i.json 2> $O/trace20_$i.err; tail -2 $O/trace20_$i.err; python3 -c "
import json,sys
d=json.loads(open('$O/trace20_$i.json').read().strip().splitlines()[-1]); print('value', round(d['value'],1), 'ceiling', round(d['host']['pnp_ceiling_fps']))"; done
for i in 1 2; do OPHIP_BENCH_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only --no-side-legs --no-pnp > $O/trace20np_$i.json 2> $O/trace20np_$i.err; tail -1 $O/trace20np_$i.err; done
