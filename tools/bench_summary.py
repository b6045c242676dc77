#!/usr/bin/env python3
"""Prints the per-seed table of a bench.py line: python tools/bench_summary.py gpurun_out/r03/bench_b.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step %.2f  value %.3e  k_svm_screen %.2f ms frac %.3f  bare %.0f TF" % (d["ms_per_step"], d["value"], r["kernel_ms"], r["frac"], r.get("box_bare_mfma_tflops", 0)))
print("stages", {k: round(v, 3) for k, v in d["stage_ms_per_step"].items()})
print("seed   ms    refined%  three-pass  integer  fp64   screen  refine recheck")
for s in d["seeds"]["per_seed"]:
    print("%5d %6.2f %7.2f %10d %8d %6d %7.2f %6.2f %6.2f" % (s["seed"], s["ms_per_step"], 100 * s["refined_share"], s.get("three_pass_tier", 0), s.get("exact_integer_tier", 0),
                                                             s.get("fp64_mfma_tier", 0), s["kernel_ms"], s["refine_ms"], s["recheck_ms"]))
print("worst/median", round(d["seeds"]["worst_over_median_ms"], 3))
g = d["grasp_latency"]
print("C2 %.3f ms  C3 %.3f ms" % (g["ms_median"], g["c3"]["ms_median"]) + ("  C4 %.3f ms (%.3f per cloud)" % (g["c4"]["ms_median"], g["c4"]["ms_per_cloud"]) if "c4" in g else ""))
h = d["hard_model"]
print("hard_model %.3e (%.2f ms, refined %.2f %%)" % (h["value"], h["ms_per_step"], 100 * h["refined_share"]))
for k in ("f16x3_mode", "f32_mode"):
    if k in d:
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if not isinstance(b, dict)})
