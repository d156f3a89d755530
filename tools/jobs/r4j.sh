export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
export OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_probe.so
for on in 0 1; do
if [ $on -eq 1 ]; then export OPHIP_EXP_PROBE_ON=1; fi
rocprofv3 --kernel-trace --output-format csv -d $O/trace$on -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --main-region-only > $O/bench$on.json 2> $O/prof$on.err || exit 1
python3 tools/timeline.py $O/trace$on 1 > $O/timeline$on.txt 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$O/trace$on/**/*kernel_trace.csv", recursive=True))[0]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "probe_stream" in r["Kernel_Name"]]
print("probe on=$on launches", len(d), "mean us", (sum(d)/len(d)/1e3 if d else None))
PY
find $O/trace$on -name "*.csv" -size +3M -delete
tail -11 $O/timeline$on.txt
done
