#!/usr/bin/env python3
"""How does v_mfma_f32_16x16x32_f16 round?  (GPU box, testing build.)

Per output element the instruction forms c + sum_{k<32} a_k b_k: the 32 products are exact in fp32 (11 x 11 significant bits), the
sum is not.  The guard bands of the fp16 contraction tiers need a bound on its error in units of u = 2^-24 times (|c| + sum|a_k b_k|);
the architecture manuals give none.  This probe measures it on families of inputs chosen to separate the plausible designs:

  random      products of random signs and magnitudes, c comparable
  big+small   one product of 1, 31 products just under half an ulp of it: a sequential fp32 accumulation (round to nearest after
              every addition) loses every one of them (error ~ 31 * 0.49 ulp), a wide adder keeps their sum
  cancel      two large products that cancel exactly, 30 small ones: alignment to the largest exponent with few guard bits loses them
  ladder      products 2^0, 2^-1, ..., 2^-31
  c-heavy     |c| = 2^12 against products of order 1 (the accumulator dominates: truncation of the aligned products shows)
  chain10     ten instructions in a row, the result fed back as c (what the contraction kernels do)

Output: per family the largest |d - exact| / (u (|c| + sum|a_k b_k|)) -- for the chain, u times the sum over the ten steps of
(|c_s| + sum|products|) -- and whether every result is the round-to-nearest of the exact sum.

  python tools/mfma_rounding_probe.py [--trials 64] [--out profiles/r03_mfma_rounding.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from haf_grasping_amd import capi  # noqa: E402

U = 2.0 ** -24


def run(tl, a, b, c, chain=1):
    T = a.shape[0]
    a16, b16 = np.ascontiguousarray(a.astype(np.float16)), np.ascontiguousarray(b.astype(np.float16))
    c32 = np.ascontiguousarray(c.astype(np.float32))
    d = np.zeros((T, 16, 16), np.float32)
    rc = tl.haf_test_f16_mfma(a16.ctypes.data_as(C.c_void_p), b16.ctypes.data_as(C.c_void_p), c32.ctypes.data_as(C.c_void_p),
                              d.ctypes.data_as(C.c_void_p), T, chain)
    assert rc == 0, rc
    return a16.astype(np.float64), b16.astype(np.float64), c32.astype(np.float64), d.astype(np.float64)


def exact(a, b, c):
    """c + a @ b per trial in (effectively) exact arithmetic: the products are exact in fp64, the 33-term sums are done in
    long double from the smallest magnitude up (error far below 2^-53 of the scale)."""
    prod = a[:, :, :, None] * b[:, None, :, :]                                  # [T][16][32][16]
    prod = np.transpose(prod, (0, 1, 3, 2))                                      # [T][16][16][32]
    terms = np.concatenate([prod, c[:, :, :, None]], axis=3).astype(np.longdouble)
    order = np.argsort(np.abs(terms), axis=3)
    s = np.take_along_axis(terms, order, axis=3).sum(axis=3)
    scale = np.abs(terms).sum(axis=3)
    return s.astype(np.float64), scale.astype(np.float64), s


def report(name, tl, a, b, c, out):
    A, B, Cc, D = run(tl, a, b, c)
    ex, scale, exl = exact(A, B, Cc)
    ratio = np.abs(D - ex) / (U * np.maximum(scale, 1e-300))
    rn = (exl.astype(np.float32).astype(np.float64) == D)
    out[name] = dict(max_err_over_u_scale=float(ratio.max()), mean=float(ratio.mean()), round_to_nearest_of_exact_sum=float(rn.mean()),
                     max_err_in_ulps_of_result=float((np.abs(D - ex) / np.maximum(np.spacing(np.abs(D).astype(np.float32)).astype(np.float64), 1e-300)).max()))
    print("%-10s max |err| = %.3f u*scale (mean %.3f)   = RN(exact) for %.1f %% of elements; worst %.2f ulp of the result" % (
        name, ratio.max(), ratio.mean(), 100 * rn.mean(), out[name]["max_err_in_ulps_of_result"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=64)
    ap.add_argument("--out", default=None)
    a_ = ap.parse_args()
    tl = capi.testlib()
    tl.haf_test_f16_mfma.restype = C.c_int
    tl.haf_test_f16_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.RandomState(5)
    T = a_.trials
    out = {}
    # random
    a = rng.standard_normal((T, 16, 32)) * np.exp2(rng.randint(-3, 4, (T, 16, 32)))
    b = rng.standard_normal((T, 32, 16)) * np.exp2(rng.randint(-3, 4, (T, 32, 16)))
    c = rng.standard_normal((T, 16, 16)) * 8
    report("random", tl, a, b, c, out)
    # big + small: a row = [1, e, e, ...], b = 1: products 1 and 31 x e with e just under half an ulp of 1 (ulp(1) = 2^-23)
    for e, nm in ((2.0 ** -25 * 1.9375, "big+small"), (2.0 ** -26 * 1.5, "big+tiny")):
        a = np.full((T, 16, 32), 1.0)
        b = np.full((T, 32, 16), e)
        b[:, 0, :] = 1.0
        for t in range(T):                                       # the big product at varying positions
            b[t] = np.roll(b[t], t % 32, axis=0)
        c = np.zeros((T, 16, 16))
        report(nm, tl, a, b, c, out)
    # one big product, 31 random small ones of one sign, not on any grid: each truncation loses up to one unit of the last kept place
    for lo_e, nm in ((-22, "big+rand22"), (-20, "big+rand20"), (-24, "big+rand24")):
        a = np.full((T, 16, 32), 1.0)
        b = rng.uniform(1.0, 2.0, (T, 32, 16)) * 2.0 ** lo_e
        b[:, 0, :] = rng.uniform(1.0, 2.0, (T, 16))
        for t in range(T):
            b[t] = np.roll(b[t], (5 * t) % 32, axis=0)
        report(nm, tl, a, b, np.zeros((T, 16, 16)), out)
    # the same against a dominating accumulator
    a = np.full((T, 16, 32), 1.0)
    b = rng.uniform(1.0, 2.0, (T, 32, 16)) * 2.0 ** -22
    report("c+rand22", tl, a, b, rng.uniform(1.0, 2.0, (T, 16, 16)), out)
    report("c-rand22", tl, a, -b, rng.uniform(1.0, 2.0, (T, 16, 16)), out)
    # cancellation
    a = np.full((T, 16, 32), 1.0)
    b = rng.uniform(0.5, 1.0, (T, 32, 16)) * 2.0 ** -13
    b[:, 0, :] = 1024.0
    b[:, 1, :] = -1024.0
    for t in range(T):
        b[t] = np.roll(b[t], (3 * t) % 32, axis=0)
    report("cancel", tl, a, b, np.zeros((T, 16, 16)), out)
    # ladder
    a = np.full((T, 16, 32), 1.0) * rng.choice([-1.0, 1.0], (T, 16, 32))
    b = np.tile(np.exp2(-np.arange(32, dtype=np.float64) * 0.75)[None, :, None], (T, 1, 16)) * rng.uniform(1.0, 1.999, (T, 32, 16))
    report("ladder", tl, a, b, np.zeros((T, 16, 16)), out)
    # accumulator dominates
    a = rng.standard_normal((T, 16, 32))
    b = rng.standard_normal((T, 32, 16))
    c = rng.choice([-1.0, 1.0], (T, 16, 16)) * 4096.0 * rng.uniform(1.0, 2.0, (T, 16, 16))
    report("c-heavy", tl, a, b, c, out)
    # chain of ten
    a = rng.standard_normal((T, 16, 32))
    b = rng.standard_normal((T, 32, 16))
    c = rng.standard_normal((T, 16, 16))
    A, B, Cc, D = run(tl, a, b, c, chain=10)
    cur = Cc.astype(np.longdouble)
    tot_scale = np.zeros_like(Cc)
    prod = np.transpose(A[:, :, :, None] * B[:, None, :, :], (0, 1, 3, 2))
    for _ in range(10):
        tot_scale += np.abs(cur).astype(np.float64) + np.abs(prod).sum(axis=3)
        cur = cur + prod.astype(np.longdouble).sum(axis=3)
    ratio = np.abs(D - cur.astype(np.float64)) / (U * tot_scale)
    out["chain10"] = dict(max_err_over_u_sum_of_step_scales=float(ratio.max()), mean=float(ratio.mean()))
    print("%-10s max |err| = %.3f u * sum over the steps of (|c_s| + sum|products|) (mean %.3f)" % ("chain10", ratio.max(), ratio.mean()))
    if a_.out:
        with open(a_.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
