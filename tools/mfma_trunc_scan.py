#!/usr/bin/env python3
"""Where does v_mfma_f32_16x16x32_f16 lose bits?  One large term (a product or the accumulator) of exponent 0 and 31 / 32 small
same-sign products whose magnitude is scanned over 2^-18 .. 2^-32 with mantissas near 2 (just under the next power of two): a
design that aligns every term to the largest exponent and truncates below some bit loses almost one unit per small term at the
scan point where the terms sit just under that unit.  Prints error / (2^-24 scale) per scan point.  GPU box, testing build."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from haf_grasping_amd import capi  # noqa: E402

U = 2.0 ** -24


def run(tl, a, b, c):
    T = a.shape[0]
    a16, b16 = np.ascontiguousarray(a.astype(np.float16)), np.ascontiguousarray(b.astype(np.float16))
    c32 = np.ascontiguousarray(c.astype(np.float32))
    d = np.zeros((T, 16, 16), np.float32)
    assert tl.haf_test_f16_mfma(a16.ctypes.data, b16.ctypes.data, c32.ctypes.data, d.ctypes.data, T, 1) == 0
    A, B, Cc = a16.astype(np.float64), b16.astype(np.float64), c32.astype(np.float64)
    prod = np.transpose(A[:, :, :, None] * B[:, None, :, :], (0, 1, 3, 2))
    terms = np.concatenate([prod, Cc[:, :, :, None]], axis=3).astype(np.longdouble)
    order = np.argsort(np.abs(terms), axis=3)
    s = np.take_along_axis(terms, order, axis=3).sum(axis=3)
    scale = np.abs(terms).sum(axis=3).astype(np.float64)
    return np.abs(d.astype(np.longdouble) - s).astype(np.float64) / (U * np.maximum(scale, 1e-300))


def main():
    tl = capi.testlib()
    rng = np.random.RandomState(1)
    # small products: a = 2^ea * ma, b = 2^eb * mb with ma, mb in [1, 2) fp16 mantissas: the product's mantissa scans [1, 4)
    for big_in in ("product", "accumulator"):
        for sign in (+1.0, -1.0):
            print("== big term = %s, small terms %s" % (big_in, "same sign as it" if sign > 0 else "opposite sign"))
            for e in range(-16, -34, -1):
                T = 64
                a = np.ones((T, 16, 32)); b = np.zeros((T, 32, 16)); c = np.zeros((T, 16, 16))
                ea = e // 2
                eb = e - ea
                ma = 1.0 + rng.randint(512, 1024, (T, 16, 32)) / 1024.0           # mantissas in [1.5, 2)
                mb = 1.0 + rng.randint(512, 1024, (T, 32, 16)) / 1024.0
                a = ma * 2.0 ** ea
                b = sign * mb * 2.0 ** eb
                bigv = 1.0 + rng.randint(0, 1024, T) / 1024.0
                if big_in == "product":
                    a[:, :, 0] = 1.0
                    b[:, 0, :] = bigv[:, None]
                else:
                    c[:] = bigv[:, None, None]
                r = run(tl, a, b, c)
                print("   small products ~ 2^%d..2^%d: worst %.2f  mean %.2f" % (e, e + 2, r.max(), r.mean()))
    # two scales of small terms under one big one
    print("== one big product, 15 products at 2^-e1, 16 at 2^-e2")
    best = (0, None)
    for e1 in range(-18, -28, -1):
        for e2 in range(e1 - 1, -30, -1):
            T = 16
            a = np.ones((T, 16, 32)); b = np.zeros((T, 32, 16)); c = np.zeros((T, 16, 16))
            m = 1.0 + rng.randint(896, 1024, (T, 32, 16)) / 1024.0
            b[:, 1:16, :] = m[:, 1:16, :] * 2.0 ** e1
            b[:, 16:, :] = m[:, 16:, :] * 2.0 ** e2
            b[:, 0, :] = 1.0 + rng.randint(0, 1024, (T, 16)) / 1024.0
            r = run(tl, a, b, c)
            if r.max() > best[0]:
                best = (r.max(), (e1, e2))
    print("   worst %.2f at exponents %s" % best)


if __name__ == "__main__":
    main()
