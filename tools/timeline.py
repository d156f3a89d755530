"""Kernel timeline of a rocprofv3 --kernel-trace run of bench.py:  python tools/timeline.py <dir or kernel_trace.csv> [n_frames]
Prints, for a few steady-state frames in the middle of the trace, every dispatch in start order (offset from the frame's first
attn_apply, duration, queue) and how long the fine stage ran beside the similarity / confidence kernels; then the period between
consecutive frames' first encoder kernels over the whole trace."""
import csv
import glob
import os
import sys


def short(name):
    for key, tag in (("fine_pair", "fine"), ("fine_refine", "fine"), ("sim_frag_kernel<3, 0>", "sim_frag"), ("sim_frag_kernelILi3ELi0", "sim_frag"),
                     ("sim_frag", "sim_frag*"), ("conf_kernel", "conf"), ("select_place", "select_place"), ("select_decide", "select"), ("select_kernel", "select"), ("stat_combine", "stat_combine"), ("kv_sum", "kv_sum"),
                     ("enc_x3w8_kernel<true", "kv_reduce"), ("enc_x3w8_kernelILb1", "kv_reduce"), ("enc_x3w8_kernel<false", "attn_apply"),
                     ("enc_x3w8_kernelILb0", "attn_apply"), ("pe_add", "pe_add"), ("transpose", "transpose"), ("kpt", "kpt")):
        if key in name:
            return tag
    return name[:40]


def main():
    src = sys.argv[1]
    nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    firsts = []          # index of the first attn_apply after each kv_reduce
    for i, r in enumerate(rows):
        if r[2] == "kv_reduce":
            firsts.append(i)
    if len(firsts) < 8:
        print("too few frames in the trace")
        return
    mid = len(firsts) // 2
    for f in range(mid, mid + nshow):
        t0 = rows[firsts[f]][0]
        t1 = rows[firsts[f + 1]][0]
        print(f"--- frame {f}: period {(t1 - t0) / 1e3:.1f} us")
        for r in rows:
            if t0 - 50_000 <= r[0] < t1:
                print(f"  {(r[0] - t0) / 1e3:9.1f} us  +{(r[1] - r[0]) / 1e3:7.1f} us  q{r[3]:>3}  {r[2]}")
    periods = [(rows[firsts[i + 1]][0] - rows[firsts[i]][0]) / 1e3 for i in range(len(firsts) // 4, 3 * len(firsts) // 4)]
    periods.sort()
    print(f"period over the middle half of the trace: median {periods[len(periods) // 2]:.1f} us, min {periods[0]:.1f}, max {periods[-1]:.1f}")
    # overlap of fine with the coarse kernels
    fines = [r for r in rows if r[2] == "fine"]
    others = [r for r in rows if r[2] in ("sim_frag", "conf", "select", "stat_combine")]
    ov = 0
    for fs, fe, _, _ in fines:
        for os_, oe, _, _ in others:
            ov += max(0, min(fe, oe) - max(fs, os_))
    tot = sum(r[1] - r[0] for r in fines)
    print(f"fine stage: mean {tot / max(1, len(fines)) / 1e3:.1f} us; ran beside similarity / confidence / selection for {ov / max(1, len(fines)) / 1e3:.1f} us per frame")
    for tag in ("attn_apply", "sim_frag", "conf", "fine", "kv_sum", "kv_reduce", "select", "stat_combine"):
        d = [r[1] - r[0] for r in rows if r[2] == tag]
        if d:
            print(f"  {tag:12s} n={len(d):5d} mean {sum(d) / len(d) / 1e3:7.1f} us")


if __name__ == "__main__":
    main()
