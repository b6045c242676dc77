#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

    python tools/pmc_summary.py --fetch A_counter_collection.csv --write B_counter_collection.csv \
        [--fetch-f32 C.csv --write-f32 D.csv] [--fetch-f16x3 E.csv --write-f16x3 F.csv] --grid 512 --rolls 36 --nsv 4096 -o profiles/pmc_traffic.json

Per kernel the value is the AVERAGE over that kernel's FULL-SIZE launches in the pass (KiB, as the counter reports it): since round 3
haf_create scores a small synthetic request once (calibrate(), engine.cpp), so every kernel has one launch of a fraction of the
size in each process -- launches below half of the kernel's largest value are left out of the average (and counted in `small`).
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts wide coalesced reads at half their size
(MI355X_MICROARCH.md, HBM / rocprofv3 section); writes are reported as they are.
"""
import argparse
import csv
import json
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("::")[-1]


def averages(path, counter):
    vals = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            vals[k].append(float(row["Counter_Value"]))
    out = {}
    for k, v in vals.items():
        full = [x for x in v if x >= 0.5 * max(v)] or v
        out[k] = (sum(full) / len(full), len(full))
    return out


def fold(fetch_csv, write_csv, run):
    fetch = averages(fetch_csv, "FETCH_SIZE")
    write = averages(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes": (2.0 * f + w) * 1024.0,
                  "launches": nf, "run": run}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--fetch-f32")
    ap.add_argument("--write-f32")
    ap.add_argument("--fetch-f16x3")
    ap.add_argument("--write-f16x3")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--rolls", type=int, default=36)
    ap.add_argument("--nsv", type=int, default=4096)
    ap.add_argument("-o", "--out", default="profiles/pmc_traffic.json")
    a = ap.parse_args()
    kernels = {}
    if a.fetch_f32 and a.write_f32:
        kernels.update(fold(a.fetch_f32, a.write_f32, "f32 mode"))
    if a.fetch_f16x3 and a.write_f16x3:
        kernels.update(fold(a.fetch_f16x3, a.write_f16x3, "f16x3 mode"))
    kernels.update(fold(a.fetch, a.write, "default (screened) mode"))
    doc = {
        "workload": {"grid": a.grid, "rolls": a.rolls, "n_sv": a.nsv},
        "note": "FETCH_SIZE/WRITE_SIZE in KiB per launch (average over the pass) from separate rocprofv3 --pmc passes; "
                "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half the bytes of wide "
                "coalesced streams, MI355X_MICROARCH.md HBM section); made by tools/pmc_summary.py",
        "kernels": kernels,
    }
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:28s} {v['hbm_bytes'] / 1e9:9.3f} GB/launch  ({v['launches']} launches, {v['run']})")


if __name__ == "__main__":
    main()
