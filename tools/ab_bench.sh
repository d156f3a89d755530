# A/B of environment-switched variants on ONE box, interleaved rounds:  bash tools/ab_bench.sh <out dir> <rounds> <steps> "<NAME=VAL ...>" "<...>" ...
# each variant string is a space-separated list of environment assignments ("-" = none); prints value / matcher-less ms per variant and round
O=$1; R=$2; S=$3; shift 3
mkdir -p $O
for r in $(seq 1 $R); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e python3 bench.py --steps $S --warmup 5 --no-cpu-baseline --main-region-only > $O/v${i}_r${r}.json 2> $O/v${i}_r${r}.err || echo "variant $i failed"
    python3 - $O/v${i}_r${r}.json "$v" $r <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"round {sys.argv[3]} [{sys.argv[2]}] value {d['value']:.1f} ms/step {d['ms_per_step']:.4f} attn {d['roofline']['avg_launch_ms']*1e3:.1f} us")
except Exception as e:
    print("parse failed", sys.argv[1], e)
PY
  done
done
