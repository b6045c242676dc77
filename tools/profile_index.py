#!/usr/bin/env python3
"""profiles/index.json: ONE provenance-checked index of the committed rocprofv3 evidence, per profiled run and kernel instance.

bench.py hands out a profile-derived figure (roofline.kernel_ms_rocprof / frac_rocprof / mfma_busy / traffic) only when the run it
reports is the run that was profiled: same grid, rolls, SV count, model, contraction mode and kernel instance (template arguments
included -- the screening form is one of them).  This script folds the files tools/collect_profiles.sh copied into profiles/ into
that index:

    runs["<model>/<precision>"] = {workload: {grid, rolls, n_sv}, files: [...], kernels: {instance: {avg_ms, launches, mfma_busy, hbm_bytes}}}

  avg_ms     full-size launches under rocprofv3 --kernel-trace (tools/kernel_avg.py)
  mfma_busy  SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) of the kernel's largest launch in the SQ pass
  hbm_bytes  (2 FETCH_SIZE + WRITE_SIZE) KiB x 1024, full-size launches (gfx950 correction: MI355X_MICROARCH.md, HBM section)

    python tools/profile_index.py --round r05 --commit $(git rev-parse --short HEAD) [--nsv-trained 8964] [--nsv-hard 4128]
"""
import argparse
import collections
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("::")[-1]


def counter_rows(path):
    """{kernel instance: [ {counter: value} per dispatch ]}"""
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    out = collections.defaultdict(list)
    for (k, _), c in rows.items():
        out[k].append(dict(c))
    return out


def mfma_busy(path):
    out = {}
    for k, ls in counter_rows(path).items():
        big = max(ls, key=lambda c: c.get("GRBM_GUI_ACTIVE", 0.0))
        g = big.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if g >= 1e4 and "SQ_VALU_MFMA_BUSY_CYCLES" in big:
            out[k] = big["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / g
    return out


def traffic(fetch_csv, write_csv):
    def avg(path, counter):
        res = {}
        for k, ls in counter_rows(path).items():
            v = [c[counter] for c in ls if counter in c]
            if v:
                full = [x for x in v if x >= 0.5 * max(v)] or v
                res[k] = sum(full) / len(full)
        return res
    f, w = avg(fetch_csv, "FETCH_SIZE"), avg(write_csv, "WRITE_SIZE")
    return {k: (2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024.0 for k in set(f) | set(w)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--rolls", type=int, default=36)
    ap.add_argument("--nsv", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=42, help="the model seed of the profiled default-mode passes (the bench's median seed)")
    ap.add_argument("--nsv-trained", type=int, default=8964)
    ap.add_argument("--nsv-hard", type=int, default=4128)
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "profiles", "index.json"))
    a = ap.parse_args()
    P = os.path.join(ROOT, "profiles")
    R = a.round
    B = "%s_bench_c5_nsv%d" % (R, a.nsv)
    # run key -> (n_sv, kernel-average file, SQ pass, FETCH pass, WRITE pass); a missing file leaves its figures out
    plan = {
        "seed%d/f16s" % a.seed: (a.nsv, "%s_kernel_avg.json" % R, B + "_f16s_pmc_SQ.csv", B + "_f16s_pmc_FETCH_SIZE.csv", B + "_f16s_pmc_WRITE_SIZE.csv"),
        "seed%d/f16x3" % a.seed: (a.nsv, "%s_f16x3_kernel_avg.json" % R, None, B + "_f16x3_pmc_FETCH_SIZE.csv", B + "_f16x3_pmc_WRITE_SIZE.csv"),
        "seed%d/f32" % a.seed: (a.nsv, "%s_f32_kernel_avg.json" % R, None, B + "_f32_pmc_FETCH_SIZE.csv", B + "_f32_pmc_WRITE_SIZE.csv"),
        "seed11/f16s": (a.nsv, "%s_seed11_kernel_avg.json" % R, "%s_seed11_pmc_SQ.csv" % R, None, None),
        "trained/f16s": (a.nsv_trained, "%s_seedtrained_kernel_avg.json" % R, "%s_seedtrained_pmc_SQ.csv" % R,
                         "%s_seedtrained_pmc_FETCH_SIZE.csv" % R, "%s_seedtrained_pmc_WRITE_SIZE.csv" % R),
        "hard/f16s": (a.nsv_hard, "%s_seedhard_kernel_avg.json" % R, None, None, None),
    }
    runs = {}
    for key, (nsv, avg_f, sq_f, fe_f, wr_f) in plan.items():
        path = os.path.join(P, avg_f)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            ks = json.load(f)["kernels"]
        files = [avg_f]
        busy = tr = {}
        if sq_f and os.path.exists(os.path.join(P, sq_f)):
            busy = mfma_busy(os.path.join(P, sq_f))
            files.append(sq_f)
        if fe_f and wr_f and os.path.exists(os.path.join(P, fe_f)) and os.path.exists(os.path.join(P, wr_f)):
            tr = traffic(os.path.join(P, fe_f), os.path.join(P, wr_f))
            files += [fe_f, wr_f]
        kernels = {}
        for k, v in ks.items():
            if not k.startswith("k_"):
                continue
            kernels[k] = {"avg_ms": v["avg_ms"], "launches": v["launches"], "mfma_busy": busy.get(k), "hbm_bytes": tr.get(k)}
        runs[key] = {"workload": {"grid": a.grid, "rolls": a.rolls, "n_sv": nsv}, "files": files, "kernels": kernels}
    doc = {"round": R, "commit": a.commit,
           "note": "per profiled run and kernel instance: avg_ms = full-size launches under rocprofv3 --kernel-trace; mfma_busy = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024 "
                   "(null: that pass was not taken for the run).  Made by tools/profile_index.py from the files listed per run.",
           "runs": runs}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for key, run in runs.items():
        top = sorted(run["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"])[:3]
        print("%-14s %s" % (key, "; ".join("%s %.3f ms busy %s hbm %s" % (k, v["avg_ms"], "-" if v["mfma_busy"] is None else "%.2f" % v["mfma_busy"],
                                                                           "-" if v["hbm_bytes"] is None else "%.2f GB" % (v["hbm_bytes"] / 1e9)) for k, v in top)))


if __name__ == "__main__":
    main()
