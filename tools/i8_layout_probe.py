#!/usr/bin/env python3
"""Operand layout of v_mfma_i32_16x16x64_i8, checked with exact integer data on the GPU box (testing build: haf_test_i8_mfma):
python tools/i8_layout_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from haf_grasping_amd import capi
tl = capi.testlib()
rng = np.random.RandomState(3)
ok = True
for t in range(4):
    a = rng.randint(-64, 64, (16, 64)).astype(np.int8)
    b = rng.randint(-64, 64, (64, 16)).astype(np.int8)
    c = np.zeros((16, 16), np.int32)
    rc = tl.haf_test_i8_mfma(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    want = a.astype(np.int32) @ b.astype(np.int32)
    ok = ok and rc == 0 and (c == want).all()
    print("trial", t, "rc", rc, "match", bool((c == want).all()), "transposed match", bool((c == want.T).all()))
print("I8_LAYOUT_OK" if ok else "I8_LAYOUT_MISMATCH")
