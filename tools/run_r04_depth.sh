# frames in flight (bench.py --depth) 3 / 6 / 10, interleaved on one box, 20-step driver-style regions and 100-step ones
O=gpurun_out/r04depth
mkdir -p $O
for S in 20 100; do
for r in 1 2 3; do
  for d in 3 6 10; do
    timeout -k 10 200 python3 bench.py --steps $S --warmup 5 --depth $d --no-cpu-baseline --main-region-only > $O/s${S}_d${d}_r${r}.json 2> $O/s${S}_d${d}_r${r}.err || { echo "depth $d failed"; tail -3 $O/s${S}_d${d}_r${r}.err; exit 1; }
    python3 - $O/s${S}_d${d}_r${r}.json $S $d $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h=d.get("host",{})
print(f"steps {sys.argv[2]} depth {sys.argv[3]} round {sys.argv[4]}: value {d['value']:.1f}  attn {d['roofline']['avg_launch_ms']*1e3:.1f} us  pnp ceiling {h.get('pnp_ceiling_fps') or 0:.0f}")
PY
  done
done
done
