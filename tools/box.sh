#!/bin/bash
# ONE parameterised script for everything that runs on the GPU box (replaces the per-round run_profile_r0N.sh / run_r04_*.sh files):
#     /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/box.sh <out name> <step> [<step> ...]'
# Output goes to gpurun_out/<out name>/ (merged back by gpurun); summaries worth keeping are copied into profiles/ by hand.
# Steps (run in the order given; a failing step stops the call -- no GPU step is started after a failed or timed-out one):
#   check            pytest -m gpu + __graft_entry__.smoke()
#   bench            the driver's command (python3 bench.py --steps 20 --warmup 5), twice, + a 100-step run without side legs
#   bench1           the driver's command once
#   times            stand-alone kernel times (tools/time_kernel.py, time_fine.py, time_coarse.py, time_inputs.py)
#   stamps           in-kernel cycle stamps of attn_apply, the fine stage (two and one workgroup per CU) and the similarity tiles
#   stats            rocprofv3 --kernel-trace --stats of the contract's region alone (c2) and of config 4, + the timeline
#   pmc              counter passes, each in its own run (FETCH_SIZE | WRITE_SIZE | matrix-pipe busy | L2 hit / miss) + the summaries
#   issue            instruction counts per kernel (SQ_INSTS_VALU / _MFMA / _LDS / _SALU, own pass): what loads the SIMD's issue port
#   counters         the counter names rocprofv3 offers on the box (gpurun_out/<out>/counters.txt)
#   many:<n>         the contract's region n times (python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only): the 20-step line's spread
#   ranks            the N > 1 code path on this one-GPU box: two gloo ranks sharing the device; RCCL with one rank
#   ab:<rounds>:<steps>:<variant>[:<variant>...]   interleaved A/B of bench.py (main region only); a variant is "-" (shipped build) or a
#                    comma-separated list of NAME=VALUE environment assignments (OPHIP_LIB=onepose_st_amd/lib/variants/lib....so picks a variant build)
#   py:<script>[:<arg>...]   python3 <script> <args> > <out>/<script name>.txt   (tools/micro drivers, one-off timings)
#   epy:<NAME=VALUE[,NAME=VALUE...]>:<script>[:<arg>...]   the same with environment assignments (kernel variants behind an environment switch)
#   pt:<NAME=VALUE[,...] or ->:<pytest -k expression>   a selection of the GPU tests under environment assignments (diagnostic: does not stop the call)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
summ() { python3 - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, h = d["roofline"], d.get("host", {})
f = lambda v, n=1: (round(v, n) if isinstance(v, (int, float)) else v)
print(sys.argv[2] if len(sys.argv) > 2 else "", "value", f(d["value"]), "ms/step", f(d["ms_per_step"], 4), "| matcher", f(d.get("value_matcher_only")), "cached", f(d.get("value_matcher_only_object_cached")),
      "no_hot", f(d.get("value_no_hot_steps")), "dependent", f(d.get("value_dependent_sequence")), "lat", (d.get("latency_ms") or {}).get("median"),
      "| attn us", f(r.get("avg_launch_ms", 0) * 1e3), "frac", f(r.get("frac"), 4), "alone us", f(((r.get("alone") or {}).get("avg_launch_ms") or 0) * 1e3),
      "| c3", f(d.get("value_c3_b32")), "c4", f(d.get("value_c4")), "c4 stage_frac", f((((d.get("side_legs") or {}).get("c4") or {}).get("conf_kernel") or {}).get("stage_frac"), 4),
      "hard", f(d.get("value_c2_hard")), "fine_bf16", f(d.get("value_fine_bf16")), "| pnp ceil", f(h.get("pnp_ceiling_fps")), "cpu", f((d.get("cpu_baseline") or {}).get("value"), 2))
PY
}
for step in "$@"; do
  echo "== $step"
  case $step in
    check)
      timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; tail -6 $O/tests.log
      [ $rc -eq 0 ] || exit 1
      python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }
      tail -2 $O/smoke.txt ;;
    bench1)
      timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err || { tail -5 $O/bench20.err; exit 1; }
      summ $O/bench20.json "20 steps" ;;
    bench)
      for i in 1 2; do
        timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench20_$i.json 2> $O/bench20_$i.err || { tail -5 $O/bench20_$i.err; exit 1; }
        summ $O/bench20_$i.json "20 steps run $i"
      done
      timeout -k 10 400 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-legs > $O/bench100.json 2> $O/bench100.err || { tail -5 $O/bench100.err; exit 1; }
      summ $O/bench100.json "100 steps" ;;
    times)
      timeout -k 10 300 python3 tools/time_kernel.py pe_add_transpose transpose_cl kpt_stats kpt_encode sim_stats stat_combine conf select select_place kv_sum > $O/time_kernels.txt 2>&1 || { tail -5 $O/time_kernels.txt; exit 1; }
      timeout -k 10 200 python3 tools/time_fine.py >> $O/time_kernels.txt 2>&1 || { tail -5 $O/time_kernels.txt; exit 1; }
      timeout -k 10 200 python3 tools/time_coarse.py >> $O/time_kernels.txt 2>&1 || { tail -5 $O/time_kernels.txt; exit 1; }
      cat $O/time_kernels.txt ;;
    stamps)
      timeout -k 10 200 python3 tools/stamps_x3.py > $O/stamps_enc_x3w8.txt 2>&1 || { tail -5 $O/stamps_enc_x3w8.txt; exit 1; }
      timeout -k 10 200 python3 tools/stamps_fine.py > $O/stamps_fine_pair.txt 2>&1 || { tail -5 $O/stamps_fine_pair.txt; exit 1; }
      OPHIP_FINE_LDS_PAD=40000 timeout -k 10 200 python3 tools/stamps_fine.py > $O/stamps_fine_pair_one_wg_per_cu.txt 2>&1 || exit 1
      timeout -k 10 200 python3 tools/stamps_sim.py > $O/stamps_sim.txt 2>&1 || { tail -5 $O/stamps_sim.txt; exit 1; }
      tail -n 30 $O/stamps_enc_x3w8.txt $O/stamps_fine_pair.txt $O/stamps_fine_pair_one_wg_per_cu.txt $O/stamps_sim.txt ;;
    stats)
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_under_rocprof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
      python3 tools/timeline.py $O/stats 2 > $O/timeline.txt 2>&1 || true
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 bench.py --workload c4 --steps 16 --warmup 4 --no-cpu-baseline --main-region-only --no-side-legs --roofline-kernel conf > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err || { tail -5 $O/prof_c4.err; exit 1; }
      find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
      find $O/stats_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv
      head -16 $O/kernel_stats.csv; head -8 $O/c4_kernel_stats.csv; tail -40 $O/timeline.txt
      find $O -name "*.csv" -size +3M -delete ;;
    pmc)
      B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only"
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2> $O/pmc_f.err || { tail -5 $O/pmc_f.err; exit 1; }
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2> $O/pmc_w.err || { tail -5 $O/pmc_w.err; exit 1; }
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2> $O/pmc_m.err || { tail -5 $O/pmc_m.err; exit 1; }
      rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- $B > /dev/null 2> $O/pmc_l2.err || echo "L2 pass failed"
      python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- $B (c2, bf16x3, B=1; tools/box.sh pmc)" $O/pmc_mfma > $O/pmc_summary.txt || exit 1
      python3 tools/l2_summary.py $O/pmc_l2 > $O/l2_summary.txt 2>&1 || true
      cat $O/pmc_summary.txt; cat $O/l2_summary.txt
      find $O -name "*.csv" -size +3M -delete ;;
    issue)
      B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pnp --main-region-only"
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_issue -- $B > /dev/null 2> $O/pmc_i.err || { tail -5 $O/pmc_i.err; exit 1; }
      python3 tools/issue_summary.py $O/pmc_issue enc_x3w8 fine_pair sim_frag conf_kernel > $O/issue_summary.txt 2>&1 || true
      cat $O/issue_summary.txt
      find $O -name "*.csv" -size +3M -delete ;;
    many:*)
      n=${step#many:}
      for i in $(seq 1 $n); do
        timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only > $O/many_$i.json 2> $O/many_$i.err || { tail -3 $O/many_$i.err; exit 1; }
      done
      python3 - $O $n > $O/bench20_distribution.txt <<'PY'
import json, sys, statistics
vals = [json.loads(open(f"{sys.argv[1]}/many_{i}.json").read().strip().splitlines()[-1])["value"] for i in range(1, int(sys.argv[2]) + 1)]
print(f"{len(vals)} runs of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --main-region-only` on one box")
print("values:", " ".join(f"{v:.0f}" for v in vals))
print(f"median {statistics.median(vals):.1f}  min {min(vals):.1f}  max {max(vals):.1f}  mean {statistics.mean(vals):.1f}  stdev {statistics.pstdev(vals):.1f}")
PY
      cat $O/bench20_distribution.txt ;;
    ranks)
      python3 bench.py --gpus 2 --share-device --dist-backend gloo --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_2ranks_gloo_shared.json 2> $O/bench_2ranks.err || { tail -5 $O/bench_2ranks.err; exit 1; }
      tail -c 600 $O/bench_2ranks_gloo_shared.json; echo
      OPHIP_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --main-region-only > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || { tail -5 $O/bench_rccl_1rank.err; exit 1; }
      tail -c 400 $O/bench_rccl_1rank.json; echo ;;
    ab:*)
      IFS=: read -r _ R S rest <<< "$step"
      IFS=: read -r -a VARS <<< "$rest"
      for r in $(seq 1 $R); do
        i=0
        for v in "${VARS[@]}"; do
          i=$((i+1))
          if [ "$v" = "-" ]; then e=""; else e="${v//,/ }"; fi
          env $e timeout -k 10 300 python3 bench.py --steps $S --warmup 5 --no-cpu-baseline --main-region-only > $O/ab_v${i}_r${r}.json 2> $O/ab_v${i}_r${r}.err || { echo "variant $i failed"; tail -3 $O/ab_v${i}_r${r}.err; exit 1; }
          summ $O/ab_v${i}_r${r}.json "round $r [$v]"
        done
      done | tee $O/ab.txt ;;
    py:*)
      IFS=: read -r -a A <<< "${step#py:}"
      n=$(basename ${A[0]} .py)
      timeout -k 10 400 python3 "${A[@]}" > $O/$n.txt 2>&1 || { tail -8 $O/$n.txt; exit 1; }
      tail -40 $O/$n.txt ;;
    epy:*)
      IFS=: read -r -a A <<< "${step#epy:}"
      e="${A[0]//,/ }"; n=$(basename ${A[1]} .py)_$(echo "${A[0]}_${A[*]:2}" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-120)
      [ "${A[0]}" = "-" ] && e=""
      env $e timeout -k 10 400 python3 "${A[@]:1}" > $O/$n.txt 2>&1 || { tail -8 $O/$n.txt; exit 1; }
      echo "[${A[0]}]"; tail -20 $O/$n.txt ;;
    counters)
      rocprofv3 --list-avail > $O/counters_all.txt 2>&1 || rocprofv3 -L > $O/counters_all.txt 2>&1 || true
      grep -o -E "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" $O/counters_all.txt | sort -u > $O/counters.txt; wc -l $O/counters.txt; grep -E "SQ_INSTS|SQ_ACTIVE_INST|SQ_INST_CYCLES|SQ_VALU|SQ_WAIT_INST|SQ_BUSY_CY" $O/counters.txt | tr '\n' ' ' ;;
    pt:*)
      IFS=: read -r -a A <<< "${step#pt:}"
      if [ "${A[0]}" = "-" ]; then e=""; else e="${A[0]//,/ }"; fi
      n=pt_$(echo "${A[0]}_${A[1]}" | tr -c 'A-Za-z0-9\n' '_')
      env $e timeout -k 10 600 python3 -m pytest tests -m gpu -q -k "${A[1]}" > $O/$n.log 2>&1; rc=$?
      echo "[${A[0]}] -k '${A[1]}' rc=$rc"; tail -6 $O/$n.log ;;          # (a failing selection does not stop the call: these are diagnostic)
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
