#!/usr/bin/env python3
"""Timing models of the screening kernel's inner loop on this GPU (testing build): the bare MFMA loop and the tilings of
DESIGN.md 5 (4 / 8 row blocks per wave, hand-placed AccVGPR form).  python tools/model_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from haf_grasping_amd import capi
tl = capi.testlib()
tf = C.c_double()
for rep in range(2):
    tl.haf_test_mfma_rate(0, 36000, C.byref(tf)); print("bare loop            %.0f TFLOP/s" % tf.value)
    for mb in (4, 5, 8, 9):
        rc = tl.haf_test_mfma_model(0, mb, 128, C.byref(tf)); print("model MB=%d rc=%d      %.0f TFLOP/s executed" % (mb, rc, tf.value))
