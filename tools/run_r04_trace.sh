# kernel trace + timeline of the contract's region alone (rocprofv3 --kernel-trace --stats), c2
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_under_rocprof.json 2> $O/prof.err
python3 tools/timeline.py $O/stats 3 > $O/timeline.txt 2>&1 || true
cat $O/timeline.txt
for f in $O/stats/*/*kernel_stats.csv; do cp $f $O/kernel_stats.csv; done
head -20 $O/kernel_stats.csv | cut -c1-160
find $O -name "*.csv" -size +3M -delete
