#!/usr/bin/env python3
"""Summarise separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes into profiles/*_pmc_traffic.json.

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [command text]

Each pass is `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -- python bench.py ...` (counters in
their own runs, never with --stats/--sys-trace).  HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md:
both counters are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import statistics
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def collect(d, counter):
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                per.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    command = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in fetch:
        if k.startswith("__amd_rocclr") or k not in write:
            continue
        fk, wk = statistics.median(fetch[k]), statistics.median(write[k])
        kernels[k] = {
            "FETCH_SIZE_KiB_median": fk,
            "WRITE_SIZE_KiB_median": wk,
            "launches": len(fetch[k]),
            "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
        }
    json.dump({
        "command": command,
        "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md HBM: on gfx950 FETCH_SIZE "
                      "reports 1/2 of a wide 16 B/lane coalesced read; WRITE_SIZE is exact)",
        "kernels": kernels,
    }, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  x{v['launches']}")


if __name__ == "__main__":
    main()
