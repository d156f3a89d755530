export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
for m in 1 0; do
export OPHIP_FRAME_DEFER_FINE=$m
rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench$m.json 2> $O/prof$m.err || exit 1
python3 tools/timeline.py $O/trace$m 2 > $O/timeline$m.txt 2>&1
find $O/trace$m -name "*.csv" -size +3M -delete
done
cat $O/timeline1.txt | tail -75
tail -14 $O/timeline0.txt
