#!/usr/bin/env python3
"""Per-kernel L2 (TCC) request / hit / miss medians from a `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum` pass.
usage: python tools/l2_summary.py <dir of the pass>.  A TCC request is one 128-byte line on gfx950: requests x 128 B is the L2 -> CU
traffic of a launch (what the fine stage's per-match weight stream costs)."""
import csv, glob, os, re, statistics, sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()


per = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            per.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, c in sorted(per.items(), key=lambda kv: -statistics.median(kv[1].get("TCC_REQ_sum", [0]))):
    if k.startswith("__amd"):
        continue
    req, hit, miss = (statistics.median(c.get(n, [0])) for n in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"))
    print(f"{k:58s} req {req:12.0f} (x128 B = {req * 128 / 1e6:9.1f} MB)  hit {hit:12.0f}  miss {miss:12.0f}  hit rate {hit / max(hit + miss, 1):.3f}  x{len(c.get('TCC_REQ_sum', []))}")
