# frames in flight (bench.py --depth), interleaved on one box: with the host PnP and matcher only, 100-step regions
O=gpurun_out/r04depth2
mkdir -p $O
for r in 1 2; do
  for d in 3 4 6; do
    for mode in pnp nopnp; do
    extra=""; [ $mode = nopnp ] && extra="--no-pnp"
    OPHIP_BENCH_TRACE=1 timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --depth $d $extra --no-cpu-baseline --main-region-only > $O/${mode}_d${d}_r${r}.json 2> $O/${mode}_d${d}_r${r}.err || { echo "depth $d failed"; tail -3 $O/${mode}_d${d}_r${r}.err; exit 1; }
    python3 - $O/${mode}_d${d}_r${r}.json $mode $d $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} depth {sys.argv[3]} round {sys.argv[4]}: value {d['value']:.1f}  attn {d['roofline']['avg_launch_ms']*1e3:.1f} us")
PY
    grep -h "host side\|enqueue" $O/${mode}_d${d}_r${r}.err | tail -2 | cut -c1-250
    done
  done
done
