# outliers of the 20-step line (the last frames' poses waiting for piled-up background workers): variants interleaved, N rounds
O=gpurun_out/r04many_ab
mkdir -p $O
N=${1:-20}
for r in $(seq 1 $N); do
  for v in "X=0" "OPPNP_WORKER_NICE=0" "OPPNP_WORKER_NICE=0 OPHIP_SPLIT_FEEDER=4"; do
    env $v OPHIP_BENCH_TRACE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/${v// /_}_r${r}.json 2> $O/${v// /_}_r${r}.err || { echo "$v run $r failed"; tail -3 $O/${v// /_}_r${r}.err; exit 1; }
    python3 - $O/${v// /_}_r${r}.json "$v" $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"round {sys.argv[3]} [{sys.argv[2]}]: value {d['value']:.1f}")
PY
  done
done
