# distribution of the driver-style 20-step line: N runs back to back on one box, host trace kept for each
O=gpurun_out/r04many
mkdir -p $O
N=${1:-30}
for r in $(seq 1 $N); do
  env $2 OPHIP_BENCH_TRACE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/r${r}.json 2> $O/r${r}.err || { echo "run $r failed"; tail -3 $O/r${r}.err; exit 1; }
  python3 - $O/r${r}.json $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"run {sys.argv[2]}: value {d['value']:.1f}")
PY
  grep -h "host us/step" $O/r${r}.err | tail -1 | cut -c1-160
done
