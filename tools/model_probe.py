import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from haf_grasping_amd import capi
tl = capi.testlib()
tf = C.c_double()
for rep in range(2):
    tl.haf_test_mfma_rate(0, 36000, C.byref(tf)); print("bare loop            %.0f TFLOP/s" % tf.value)
    for mb in (4, 5, 8, 9):
        rc = tl.haf_test_mfma_model(0, mb, 128, C.byref(tf)); print("model MB=%d rc=%d      %.0f TFLOP/s executed" % (mb, rc, tf.value))
