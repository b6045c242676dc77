#!/usr/bin/env python3
"""Per-kernel reading of an SQ counter pass (rocprofv3 --pmc ..., tools/profile_round.sh): for the LARGEST launch of every kernel
(the calibration requests of haf_create launch most kernels at a fraction of the size as well) the MFMA-busy share
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), wave cycles, issue and wait cycles; with --lds the LDS pass
(instructions, bank conflicts, index-active cycles).   python tools/pmc_sq_summary.py <counter_collection.csv> [--lds]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void\s+", "", name)
    return name.split("::")[-1]


def load(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path, newline="")):
        rows[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    out = collections.defaultdict(list)
    for (k, _), c in rows.items():
        out[k].append(dict(c))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--lds", action="store_true")
    a = ap.parse_args()
    d = load(a.csv)
    if a.lds:
        print("%-30s %12s %12s %12s %12s" % ("kernel (largest launch)", "SQ_INSTS_LDS", "BANK_CONFL", "IDX_ACTIVE", "WAIT_INST_LDS"))
        for k, ls in sorted(d.items(), key=lambda kv: -max(c.get("SQ_INSTS_LDS", 0) for c in kv[1])):
            big = max(ls, key=lambda c: c.get("SQ_INSTS_LDS", 0))
            if big.get("SQ_INSTS_LDS", 0) < 1e5:
                continue
            print("%-30s %12.3g %12.3g %12.3g %12.3g" % (k[:30], big.get("SQ_INSTS_LDS", 0), big.get("SQ_LDS_BANK_CONFLICT", 0),
                                                       big.get("SQ_LDS_IDX_ACTIVE", 0), big.get("SQ_WAIT_INST_LDS", 0)))
        return
    print("%-30s %10s %9s %11s %11s %11s %11s" % ("kernel (largest launch)", "cycles", "MFMA busy", "WAVE_CYCLES", "ACTIVE_ANY", "WAIT_ANY", "INSTS_VALU"))
    for k, ls in sorted(d.items(), key=lambda kv: -max(c.get("GRBM_GUI_ACTIVE", 0) for c in kv[1])):
        big = max(ls, key=lambda c: c.get("GRBM_GUI_ACTIVE", 0))
        g = big.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if g < 1e5:
            continue
        print("%-30s %10.3g %9.2f %11.3g %11.3g %11.3g %11.3g" % (k[:30], g, big.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / g,
                                                              big.get("SQ_WAVE_CYCLES", 0), big.get("SQ_ACTIVE_INST_ANY", 0),
                                                              big.get("SQ_WAIT_INST_ANY", 0), big.get("SQ_INSTS_VALU", 0)))


if __name__ == "__main__":
    main()
