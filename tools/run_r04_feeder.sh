# sporadic slow runs with the reference-policy RANSAC pool running: does keeping the pool's workers off the feeder thread's CPUs (or lowering
# their priority further) remove them?  Variants interleaved, 8 rounds of 20-step regions.
O=gpurun_out/r04feeder
mkdir -p $O
for r in 1 2 3 4 5 6; do
  for v in "X=0" "OPHIP_SPLIT_FEEDER=4" "OPHIP_SPLIT_FEEDER=8" "OPPNP_WORKER_IDLE=1"; do
    env $v OPHIP_BENCH_TRACE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/${v}_r${r}.json 2> $O/${v}_r${r}.err || { echo "$v failed"; tail -3 $O/${v}_r${r}.err; exit 1; }
    python3 - $O/${v}_r${r}.json "$v" $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"round {sys.argv[3]} [{sys.argv[2]}]: value {d['value']:.1f}")
PY
    grep -h "host us/step" $O/${v}_r${r}.err | tail -1 | cut -c1-200
  done
done
