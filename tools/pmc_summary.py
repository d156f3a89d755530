#!/usr/bin/env python3
"""Summarise separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes into profiles/*_pmc_traffic.json.

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [command text] [dir of the matrix-pipe pass]

The optional matrix-pipe pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE`) adds, per kernel,
the MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
sums the 8 XCDs; MI355X_MICROARCH.md "DVFS give-back").  No clock is derived from it: GRBM_GUI_ACTIVE / 8 / duration reads high on
dispatches shorter than ~0.3 ms (the guide says so; round 2's `clock_ghz` field printed 6 GHz for small kernels) -- the ONE clock
this project quotes for a kernel is the in-kernel s_memtime / s_memrealtime ratio of tools/stamps_x3.py (profiles/r03_stamps_*).
The output also records the build stamp of the library the passes ran on (ophip_build_stamp) and its ABI version.

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


def durations(d):
    """median kernel duration (ns) per kernel from the pass's kernel trace"""
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                per.setdefault(short(row["Kernel_Name"]), []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return {k: statistics.median(v) for k, v in per.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    command = sys.argv[4] if len(sys.argv) > 4 else ""
    mfma_dir = sys.argv[5] if len(sys.argv) > 5 else None
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    mf = {c: collect(mfma_dir, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")} if mfma_dir else {}
    dur = durations(mfma_dir) if mfma_dir else {}
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
        if mf and k in mf["SQ_VALU_MFMA_BUSY_CYCLES"] and k in mf["GRBM_GUI_ACTIVE"]:
            busy = statistics.median(mf["SQ_VALU_MFMA_BUSY_CYCLES"][k])
            gui = statistics.median(mf["GRBM_GUI_ACTIVE"][k])
            cyc = gui / 8.0
            kernels[k].update({
                "SQ_VALU_MFMA_BUSY_CYCLES_median": busy,
                "GRBM_GUI_ACTIVE_median": gui,
                "SQ_BUSY_CYCLES_median": statistics.median(mf["SQ_BUSY_CYCLES"].get(k, [0])),
                "SQ_WAVE_CYCLES_median": statistics.median(mf["SQ_WAVE_CYCLES"].get(k, [0])),
                "mfma_busy_frac": busy / (1024.0 * cyc) if cyc else None,
                "duration_us_under_pmc": dur.get(k, 0) / 1e3,
            })
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from onepose_st_amd import hip
        stamp, abi = hip.build_stamp(), int(hip.load().ophip_abi_version())
    except Exception as e:          # the summary is still written; bench.py then refuses to quote it
        stamp, abi = f"unknown ({e})", None
    json.dump({
        "command": command,
        "library_build_stamp": stamp,
        "ophip_abi_version": abi,
        "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md HBM: on gfx950 FETCH_SIZE "
                      "reports 1/2 of a wide 16 B/lane coalesced read; WRITE_SIZE is exact)",
        "kernels": kernels,
    }, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
        extra = f"  mfma busy {v['mfma_busy_frac']:.3f}, {v['duration_us_under_pmc']:.1f} us" if v.get("mfma_busy_frac") is not None else ""
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  x{v['launches']}{extra}")


if __name__ == "__main__":
    main()
