#!/usr/bin/env python3
"""Per-kernel average duration of the FULL-SIZE launches in a rocprofv3 --kernel-trace CSV (the calibration requests of haf_create
launch every kernel once or twice at a fraction of the size: launches below half of a kernel's longest are left out, as
tools/pmc_summary.py does for the counters).   python tools/kernel_avg.py <kernel_trace.csv> [-o out.json] [--note text]"""
import argparse
import csv
import json
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    return name.split("::")[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("-o", "--out")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    dur = defaultdict(list)
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    out = {}
    for k, v in dur.items():
        full = [x for x in v if x >= 0.5 * max(v)]
        out[k] = {"avg_ms": sum(full) / len(full), "launches": len(full), "small_launches": len(v) - len(full), "total_ms": sum(v)}
    doc = {"note": a.note or "average duration of the full-size launches per kernel, from rocprofv3 --kernel-trace (tools/kernel_avg.py)", "kernels": out}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])[:24]:
        print("%-44s full-size avg %9.3f ms x %3d   (+ %d small)  total %8.2f ms" % (k[:44], v["avg_ms"], v["launches"], v["small_launches"], v["total_ms"]))


if __name__ == "__main__":
    main()
