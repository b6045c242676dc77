#!/usr/bin/env python3
"""Bare MFMA loops on this GPU (testing build): v_mfma_f32_16x16x32_f16 and v_mfma_i32_16x16x64_i8, operands in registers."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from haf_grasping_amd import capi
tl = capi.testlib()
tf = C.c_double()
for rep in range(3):
    tl.haf_test_mfma_rate(0, 36000, C.byref(tf)); f16 = tf.value
    tl.haf_test_mfma_rate(0, -36000, C.byref(tf)); i8 = tf.value
    print("bare loops: f16 16x16x32 %.0f TFLOP/s   i8 16x16x64 %.0f TOP/s   ratio %.2f" % (f16, i8, i8 / f16))
