"""ctypes binding of include/hafgrasp.h (libhafgrasp.so).  No torch types cross this boundary: plain pointers and
sizes only.  The library is gfx950-only and has no CPU fallback; loading works anywhere, haf_create needs the GPU."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAF_LIB", os.path.join(HERE, "libhafgrasp.so"))   # HAF_LIB: A/B another build of the same ABI
# the TESTING build (-DHAF_TESTING): same kernels, plus the haf_test_* hooks and the environment switches that scale the
# guard bands.  Only tests/ load it (testlib(), Engine(..., testing=True)); the product library has neither.
TESTLIB_PATH = os.environ.get("HAF_TESTLIB", os.path.join(HERE, "libhafgrasp_testing.so"))   # HAF_TESTLIB: likewise (tools/ablate_h.sh)

HAF_OK, HAF_E_ARG, HAF_E_IO, HAF_E_DEVICE, HAF_E_CAPACITY, HAF_E_BUDGET, HAF_E_INTERNAL = 0, -1, -2, -3, -4, -5, -6
FLAG_KEEP_DEBUG, FLAG_PROFILE, FLAG_FP32_MFMA, FLAG_SPLIT_F16, FLAG_PROBABILITY, FLAG_FULL_RANK = 1, 2, 4, 8, 16, 32
DBG_HEIGHTS, DBG_INTEGRAL, DBG_MASK, DBG_LABELS, DBG_DECISION, DBG_TRANSFORM, DBG_SCREEN_MARGIN, DBG_PROBABILITY, DBG_GRASPSGRID = range(9)
SHARD_ROLLS, SHARD_CLOUDS = 0, 1
STAGES = ["upload", "bin", "integral", "mask", "features", "svm", "refine", "recheck", "vote", "download"]


class Config(C.Structure):
    _fields_ = [("feature_file", C.c_char_p), ("range_file", C.c_char_p), ("model_file", C.c_char_p),
                ("nr_features_without_shaf", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32),
                ("n_rolls", C.c_int32), ("roll_step_deg", C.c_int32), ("z_shift", C.c_float),
                ("graspval_top", C.c_int32), ("device", C.c_int32), ("max_clouds", C.c_int32),
                ("max_points", C.c_int64), ("flags", C.c_uint32), ("graspval_th", C.c_int32),
                ("max_rolls_per_call", C.c_int32)]


class GraspInput(C.Structure):
    _fields_ = [("grasp_area_center", C.c_double * 3), ("grasp_area_length_x", C.c_float),
                ("grasp_area_length_y", C.c_float), ("approach_vector", C.c_double * 3),
                ("max_calculation_time", C.c_double), ("show_only_best_grasp", C.c_int32),
                ("threshold_grasp_evaluation", C.c_int32), ("gripper_opening_width", C.c_int32)]


class GraspOutput(C.Structure):
    _fields_ = [("eval", C.c_int32), ("grasp_point1", C.c_double * 3), ("grasp_point2", C.c_double * 3),
                ("averaged_grasp_point", C.c_double * 3), ("approach_vector", C.c_double * 3), ("roll", C.c_float),
                ("best_row", C.c_int32), ("best_col", C.c_int32), ("best_roll", C.c_int32), ("best_vote", C.c_int32),
                ("rolls_done", C.c_int32), ("n_evals", C.c_int64), ("n_rechecked", C.c_int64)]


class RollRecord(C.Structure):
    _fields_ = [("vote", C.c_int32), ("row", C.c_int16), ("col", C.c_int16), ("h_locmax", C.c_float),
                ("n_evals", C.c_int32)]


class Cloud(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("n_points", C.c_size_t), ("stride_floats", C.c_size_t), ("on_device", C.c_int32)]


ATTR_RECORD_DTYPE = np.dtype([("feature", np.float32), ("pad", np.float32), ("q4", np.float64), ("scaled", np.float64)])
assert ATTR_RECORD_DTYPE.itemsize == 24

ROLL_RECORD_DTYPE = np.dtype([("vote", np.int32), ("row", np.int16), ("col", np.int16), ("h_locmax", np.float32),
                              ("n_evals", np.int32)])
assert ROLL_RECORD_DTYPE.itemsize == C.sizeof(RollRecord) == 16

_lib = None
_testlib = None


class HafError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hafgrasp error %d: %s" % (code, msg))
        self.code = code


def _bind(path, testing):
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950); "
                           "there is no CPU fallback" % path)
    L = C.CDLL(path)
    E = C.c_void_p
    L.haf_abi_version.restype = C.c_int
    L.haf_config_default.argtypes = [C.POINTER(Config)]
    L.haf_grasp_input_default.argtypes = [C.POINTER(GraspInput)]
    L.haf_create.argtypes = [C.POINTER(Config), C.POINTER(E)]
    L.haf_destroy.argtypes = [E]
    L.haf_last_error.restype = C.c_char_p
    L.haf_last_error.argtypes = [E]
    L.haf_score.argtypes = [E, C.POINTER(Cloud), C.POINTER(GraspInput), C.POINTER(GraspOutput)]
    L.haf_score_batch.argtypes = [E, C.c_int32, C.POINTER(Cloud), C.POINTER(GraspInput), C.POINTER(GraspOutput)]
    L.haf_score_rolls.argtypes = [E, C.c_int32, C.POINTER(Cloud), C.POINTER(GraspInput), C.c_int32, C.c_int32,
                                  C.c_void_p]
    L.haf_finalize.argtypes = [E, C.POINTER(GraspInput), C.c_void_p, C.POINTER(GraspOutput)]
    L.haf_roll_pose.argtypes = [E, C.POINTER(GraspInput), C.c_void_p, C.c_int32, C.POINTER(GraspOutput),
                                C.POINTER(C.c_int32)]
    L.haf_get_roll_grid.argtypes = [E, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.haf_debug_fetch.argtypes = [E, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]
    L.haf_debug_fetch_attr.argtypes = [E, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int32)]
    L.haf_set_stream.argtypes = [E, C.c_void_p]
    L.haf_get_stream.restype = C.c_void_p
    L.haf_get_stream.argtypes = [E]
    L.haf_get_stage_ms.argtypes = [E, C.POINTER(C.c_float)]
    L.haf_model_info.argtypes = [E, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.haf_last_counts.argtypes = [E, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.haf_last_tiers.argtypes = [E] + [C.POINTER(C.c_int64)] * 4
    L.haf_last_prestage.argtypes = [E, C.POINTER(C.c_int64)]
    L.haf_multi_plan.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.haf_last_strict_host.argtypes = [E, C.POINTER(C.c_int64)]
    L.haf_last_exact_tiers.argtypes = [E, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.haf_screen_form.argtypes = [E, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.haf_screen_low_rank.argtypes = [E, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.haf_register_host_cloud.argtypes = [E, C.c_void_p, C.c_size_t]
    L.haf_unregister_host_cloud.argtypes = [E, C.c_void_p]
    L.haf_pcd_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.c_char_p,
                               C.c_size_t]
    L.haf_free.argtypes = [C.c_void_p]
    # several GPUs in one process (csrc/multi.cpp)
    L.haf_create_multi.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(E)]
    L.haf_destroy_multi.argtypes = [E]
    L.haf_multi_last_error.restype = C.c_char_p
    L.haf_multi_last_error.argtypes = [E]
    L.haf_score_sharded.argtypes = [E, C.POINTER(Cloud), C.POINTER(GraspInput), C.POINTER(GraspOutput)]
    L.haf_score_batch_sharded.argtypes = [E, C.c_int32, C.POINTER(Cloud), C.POINTER(GraspInput), C.POINTER(GraspOutput),
                                          C.POINTER(C.c_int32)]
    L.haf_multi_info.argtypes = [E, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.haf_multi_engine.restype = C.c_void_p
    L.haf_multi_engine.argtypes = [E, C.c_int32]
    L.haf_multi_last_records.argtypes = [E, C.c_int32, C.c_void_p]
    L.haf_multi_last_timing.argtypes = [E, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
    if testing:
        L.haf_test_decq_host.restype = C.c_double
        L.haf_test_decq_host.argtypes = [C.c_double, C.c_int]
        L.haf_test_scale_host.restype = C.c_double
        L.haf_test_scale_host.argtypes = [C.c_double] * 5
        L.haf_test_sigma_upper.restype = C.c_double
        L.haf_test_sigma_upper.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.haf_test_decq4_scr.restype = C.c_double
        L.haf_test_decq4_scr.argtypes = [C.c_float]
        L.haf_test_split3.restype = C.c_double
        L.haf_test_split3.argtypes = [C.c_double, C.c_void_p]
        L.haf_test_decq_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.haf_test_mfma_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.haf_test_mfma_rate.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.haf_test_mfma_model.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.haf_test_scale_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                            C.c_int]
        L.haf_test_feature_table.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
        L.haf_test_range_table.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.haf_test_model.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.haf_test_model_kernel.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.haf_test_roll_geo.argtypes = [C.POINTER(Config), C.POINTER(GraspInput), C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.haf_test_finalize.argtypes = [C.POINTER(Config), C.POINTER(GraspInput), C.c_void_p, C.POINTER(GraspOutput)]
        L.haf_test_roll_pose.argtypes = [C.POINTER(Config), C.POINTER(GraspInput), C.c_void_p, C.c_int,
                                         C.POINTER(GraspOutput), C.POINTER(C.c_int32)]
        L.haf_test_mfma_kappa.argtypes = [E, C.c_void_p, C.c_void_p]
        L.haf_test_f16_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.haf_test_screen_state.argtypes = [E, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.haf_test_set_screen_inactive.argtypes = [E]
        L.haf_test_check_canaries.argtypes = [C.c_char_p, C.c_int]
        L.haf_test_canary_buffers.argtypes = []
        L.haf_test_poke_flag0_list.argtypes = [E, C.c_int, C.c_int, C.c_int]
        L.haf_test_overflow_stats.argtypes = [E, C.c_void_p]
        L.haf_test_fetch_list.argtypes = [E, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return L


def lib():
    """Loads libhafgrasp.so (the product).  Raises if it has not been built: nothing here falls back to anything else."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH, testing=False)
    return _lib


def multi_plan(devices, shard_mode, n_rolls):
    """haf_multi_plan: the partition haf_create_multi would build for `devices` (no device is touched).  Raises HafError like
    MultiEngine would.  -> dict(rank_of, slot_of, roll_first, roll_count, n_ranks)"""
    L = lib()
    n = len(devices)
    dev = (C.c_int32 * max(1, n))(*devices)
    rk, sl, rf, rc_ = [(C.c_int32 * max(1, n))() for _ in range(4)]
    nr = C.c_int32()
    rc = L.haf_multi_plan(dev, n, shard_mode, n_rolls, rk, sl, rf, rc_, C.byref(nr))
    if rc != 0:
        raise HafError(rc, L.haf_multi_last_error(None).decode())
    return dict(rank_of=list(rk)[:n], slot_of=list(sl)[:n], roll_first=list(rf)[:n], roll_count=list(rc_)[:n], n_ranks=nr.value)


def testlib():
    """Loads libhafgrasp_testing.so: the same kernels with the haf_test_* hooks and the guard-band environment switches."""
    global _testlib
    if _testlib is None:
        _testlib = _bind(TESTLIB_PATH, testing=True)
    return _testlib


def check_canaries():
    """Testing build: the guard zones in front of and behind EVERY device buffer of every engine of this process (csrc/engine_state.h:
    DevBuf).  -> (number of damaged buffers, report naming them by the source line that allocated them, buffers registered)"""
    L = testlib()
    msg = C.create_string_buffer(4096)
    bad = L.haf_test_check_canaries(msg, 4096)
    return bad, msg.value.decode(errors="replace"), L.haf_test_canary_buffers()


def default_config(**kw):
    cfg = Config()
    lib().haf_config_default(C.byref(cfg))
    for k, v in kw.items():
        if k in ("feature_file", "range_file", "model_file") and isinstance(v, str):
            v = v.encode()
        setattr(cfg, k, v)
    return cfg


def default_input(**kw):
    gi = GraspInput()
    lib().haf_grasp_input_default(C.byref(gi))
    for k, v in kw.items():
        if k in ("grasp_area_center", "approach_vector"):
            v = (C.c_double * 3)(*v)
        setattr(gi, k, v)
    return gi


def output_to_dict(o):
    return dict(eval=o.eval, grasp_point1=tuple(o.grasp_point1), grasp_point2=tuple(o.grasp_point2),
                averaged_grasp_point=tuple(o.averaged_grasp_point), approach_vector=tuple(o.approach_vector),
                roll=o.roll, best_row=o.best_row, best_col=o.best_col, best_roll=o.best_roll, best_vote=o.best_vote,
                rolls_done=o.rolls_done, n_evals=o.n_evals, n_rechecked=o.n_rechecked)


def load_pcd(path):
    """PCD file -> float32 [N, 3] through the library's own reader (haf_pcd_load)."""
    p = C.POINTER(C.c_float)()
    n = C.c_size_t()
    err = C.create_string_buffer(256)
    rc = lib().haf_pcd_load(path.encode(), C.byref(p), C.byref(n), err, 256)
    if rc != HAF_OK:
        raise HafError(rc, err.value.decode())
    try:
        return np.ctypeslib.as_array(p, shape=(n.value, 3)).copy()
    finally:
        lib().haf_free(p)


class Engine:
    """Owns one haf_engine handle (one GPU)."""

    def __init__(self, feature_file, range_file, model_file, testing=False, **cfg):
        self._L = testlib() if testing else lib()
        self.cfg = default_config(feature_file=feature_file, range_file=range_file, model_file=model_file, **cfg)
        self._h = C.c_void_p()
        rc = self._L.haf_create(C.byref(self.cfg), C.byref(self._h))
        if rc != HAF_OK:
            raise HafError(rc, (self._L.haf_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.haf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != HAF_OK:
            raise HafError(rc, (self._L.haf_last_error(self._h) or b"").decode())

    def register_host(self, arr):
        """haf_register_host_cloud: page-locks the numpy array's buffer; clouds passed as views into it then go with on_device = 2
        (DMA straight from the caller's memory).  The array must outlive the registration."""
        assert isinstance(arr, np.ndarray) and arr.flags["C_CONTIGUOUS"]
        self._check(self._L.haf_register_host_cloud(self._h, C.c_void_p(arr.ctypes.data), arr.nbytes))
        if not hasattr(self, "_regs"):
            self._regs = []
        self._regs.append((arr.ctypes.data, arr.nbytes, arr))

    def unregister_host(self, arr):
        self._check(self._L.haf_unregister_host_cloud(self._h, C.c_void_p(arr.ctypes.data)))
        self._regs = [r for r in getattr(self, "_regs", []) if r[0] != arr.ctypes.data]

    def _cloud(self, xyz):
        """numpy float32 [N, >=3] (host; inside a register_host buffer: page-locked, on_device = 2) or (device_ptr, n_points,
        stride_floats) tuple (HBM resident)."""
        if isinstance(xyz, tuple):
            ptr, n, stride = xyz
            return Cloud(C.c_void_p(ptr), n, stride, 1), None
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] >= 3
        where = 0
        if a.shape[1] == 3:
            for base, nbytes, _ in getattr(self, "_regs", []):
                if base <= a.ctypes.data and a.ctypes.data + a.nbytes <= base + nbytes:
                    where = 2
        return Cloud(a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], where), a

    def model_info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.haf_model_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_sv=a.value, dim=b.value, n_features=c.value)

    def last_counts(self):
        a, r, b, c = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._L.haf_last_tiers(self._h, C.byref(a), C.byref(r), C.byref(b), C.byref(c)))
        return dict(n_evals=a.value, n_refined=r.value, n_rechecked=b.value, n_strict=c.value)

    def fetch_list(self, which, cap=1 << 22):
        """Testing build: a device list of the last request as it lies in memory (0 evaluation cells, 1 exact tiers' input, 2 exact-integer
        tier's hand-over, 3 strict tier's, 4 screening pass's)."""
        buf = np.empty(cap, dtype=np.int32)
        n = C.c_int()
        self._check(self._L.haf_test_fetch_list(self._h, which, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return buf[:min(n.value, cap)].copy()

    def overflow_stats(self):
        """Testing build: how often this engine's requests met a list smaller than what it had to hold."""
        out = (C.c_longlong * 2)()
        self._check(self._L.haf_test_overflow_stats(self._h, out))
        return dict(screening_list_overflows=int(out[0]), extra_windows=int(out[1]))

    def last_exact_tiers(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self._L.haf_last_exact_tiers(self._h, C.byref(a), C.byref(b)))
        return dict(n_integer=a.value, n_fp64=b.value)

    def last_strict_host(self):
        a = C.c_int64()
        self._check(self._L.haf_last_strict_host(self._h, C.byref(a)))
        return a.value

    def last_prestage(self):
        a = C.c_int64()
        self._check(self._L.haf_last_prestage(self._h, C.byref(a)))
        return dict(n_inexact_grids=a.value)

    def score(self, xyz, grasp_input):
        return self.score_batch([xyz], [grasp_input])[0]

    def score_batch(self, clouds, inputs):
        n = len(clouds)
        keep = []
        arr = (Cloud * n)()
        for i, c in enumerate(clouds):
            arr[i], k = self._cloud(c)
            keep.append(k)
        gi = (GraspInput * n)(*inputs)
        out = (GraspOutput * n)()
        self._check(self._L.haf_score_batch(self._h, n, arr, gi, out))
        return [output_to_dict(o) for o in out]

    def score_rolls(self, clouds, inputs, roll_first, roll_count):
        n = len(clouds)
        keep = []
        arr = (Cloud * n)()
        for i, c in enumerate(clouds):
            arr[i], k = self._cloud(c)
            keep.append(k)
        gi = (GraspInput * n)(*inputs)
        rec = np.zeros((n, roll_count), dtype=ROLL_RECORD_DTYPE)
        self._check(self._L.haf_score_rolls(self._h, n, arr, gi, roll_first, roll_count, rec.ctypes.data))
        return rec

    def finalize(self, grasp_input, records):
        rec = np.ascontiguousarray(records, dtype=ROLL_RECORD_DTYPE)
        assert rec.shape == (self.cfg.n_rolls,)
        out = GraspOutput()
        self._check(self._L.haf_finalize(self._h, C.byref(grasp_input), rec.ctypes.data, C.byref(out)))
        return output_to_dict(out)

    def roll_pose(self, grasp_input, records, roll):
        """One roll's own hypothesis (server.cpp:962-969): (output dict, published flag)."""
        rec = np.ascontiguousarray(records, dtype=ROLL_RECORD_DTYPE)
        assert rec.shape == (self.cfg.n_rolls,)
        out, pub = GraspOutput(), C.c_int32()
        self._check(self._L.haf_roll_pose(self._h, C.byref(grasp_input), rec.ctypes.data, roll, C.byref(out), C.byref(pub)))
        return output_to_dict(out), bool(pub.value)

    def debug_attr(self, cloud, roll):
        """Attribute records of the masked cells of (cloud, roll): cells [n, 2], records [n, 324], computed [n]."""
        n = C.c_int32()
        self._check(self._L.haf_debug_fetch_attr(self._h, cloud, roll, 0, None, None, None, C.byref(n)))
        cells = np.zeros((n.value, 2), np.int32)
        attr = np.zeros((n.value, 324), ATTR_RECORD_DTYPE)
        comp = np.zeros(n.value, np.uint8)
        if n.value:
            self._check(self._L.haf_debug_fetch_attr(self._h, cloud, roll, n.value, cells.ctypes.data, attr.ctypes.data,
                                                     comp.ctypes.data, C.byref(n)))
        return cells, attr, comp.astype(bool)

    def roll_grid(self, cloud, roll):
        H, W = self.cfg.grid_h, self.cfg.grid_w
        ev = np.zeros((H, W), np.float32)
        mask = np.zeros((H, W), np.uint8)
        self._check(self._L.haf_get_roll_grid(self._h, cloud, roll, ev.ctypes.data, mask.ctypes.data))
        return ev, mask

    def debug(self, what, cloud, roll):
        H, W = self.cfg.grid_h, self.cfg.grid_w
        shape, dt = {DBG_HEIGHTS: ((H, W), np.float32), DBG_INTEGRAL: ((H + 1, W + 1), np.float32),
                     DBG_MASK: ((H, W), np.uint8), DBG_LABELS: ((H, W), np.int8), DBG_DECISION: ((H, W), np.float64),
                     DBG_TRANSFORM: ((4, 4), np.float32), DBG_SCREEN_MARGIN: ((H, W), np.float32),
                     DBG_PROBABILITY: ((H, W, 2), np.float64), DBG_GRASPSGRID: ((H, W), np.float32)}[what]
        a = np.zeros(shape, dt)
        self._check(self._L.haf_debug_fetch(self._h, what, cloud, roll, a.ctypes.data, a.nbytes))
        return a

    SCREEN_FORMS = ("plain", "sumsq", "centred-remainder/exp", "centred-remainder/poly")

    def screen_form(self):
        """haf_screen_form: the form of the screening kernel that serves the model, or "off" (three-pass kernel for everything)."""
        f, a = C.c_int32(), C.c_int32()
        self._check(self._L.haf_screen_form(self._h, C.byref(f), C.byref(a)))
        return self.SCREEN_FORMS[f.value] if a.value else "off"

    def screen_state(self):
        """TESTING build: which form of the screening pass serves the model (0 plain, 1 sumsq, 2 centred-remainder with exp, 3 with the
        polynomial), whether the pass is on, and the undecided share of every form on the calibration scene (-1: not tried)."""
        v, a, sh = C.c_int(), C.c_int(), (C.c_double * 4)()
        self._check(self._L.haf_test_screen_state(self._h, C.byref(v), C.byref(a), sh))
        return dict(variant=v.value & 15, tier0b=bool(v.value & 16), tier1_skipped=bool(v.value & 32), low_rank=bool(v.value & 64), active=bool(a.value),
                    shares=list(sh))

    def screen_low_rank(self):
        """haf_screen_low_rank: (tables available, rank of the HAF attributes' span, the last request's screening pass ran in the low-rank form)."""
        a, r, u = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.haf_screen_low_rank(self._h, C.byref(a), C.byref(r), C.byref(u)))
        return dict(available=bool(a.value), rank=r.value, last_used=bool(u.value))

    def set_screen_inactive(self):
        """TESTING build: switches the screening pass off as the adaptive rule does after a request every form failed on."""
        self._check(self._L.haf_test_set_screen_inactive(self._h))

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.haf_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def stage_ms(self):
        ms = (C.c_float * len(STAGES))()
        self._check(self._L.haf_get_stage_ms(self._h, ms))
        return dict(zip(STAGES, list(ms)))


class MultiEngine:
    """Owns one haf_multi handle: several GPUs of one node in this process, RCCL collectives behind the C-ABI."""

    def __init__(self, feature_file, range_file, model_file, devices, shard_mode=SHARD_ROLLS, **cfg):
        self._L = lib()
        self.cfg = default_config(feature_file=feature_file, range_file=range_file, model_file=model_file, **cfg)
        self.devices = list(devices)
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        rc = self._L.haf_create_multi(C.byref(self.cfg), dev, len(self.devices), shard_mode, C.byref(self._h))
        if rc != HAF_OK:
            raise HafError(rc, (self._L.haf_multi_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.haf_destroy_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != HAF_OK:
            raise HafError(rc, (self._L.haf_multi_last_error(self._h) or b"").decode())

    def info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.haf_multi_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_shards=a.value, n_ranks=b.value, rccl_version=c.value)

    def score_sharded(self, xyz, grasp_input):
        cl, keep = Engine._cloud(self, xyz)
        out = GraspOutput()
        self._check(self._L.haf_score_sharded(self._h, C.byref(cl), C.byref(grasp_input), C.byref(out)))
        return output_to_dict(out)

    def score_batch_sharded(self, clouds, inputs):
        n = len(clouds)
        keep = []
        arr = (Cloud * n)()
        for i, c in enumerate(clouds):
            arr[i], k = Engine._cloud(self, c)
            keep.append(k)
        gi = (GraspInput * n)(*inputs)
        out = (GraspOutput * n)()
        best = C.c_int32(-1)
        self._check(self._L.haf_score_batch_sharded(self._h, n, arr, gi, out, C.byref(best)))
        return [output_to_dict(o) for o in out], best.value

    def last_timing(self):
        """haf_multi_last_timing: host wall-clock of the last sharded call's parts."""
        n = self.info()["n_shards"]
        tot, bc, co, sh = C.c_float(), C.c_float(), C.c_float(), (C.c_float * n)()
        self._check(self._L.haf_multi_last_timing(self._h, C.byref(tot), C.byref(bc), C.byref(co), sh))
        return dict(total_ms=tot.value, bcast_us=bc.value, collective_us=co.value, shard_ms=list(sh))

    def last_records(self, rank):
        rec = np.zeros(self.cfg.n_rolls, dtype=ROLL_RECORD_DTYPE)
        self._check(self._L.haf_multi_last_records(self._h, rank, rec.ctypes.data))
        return rec

    def shard_stage_ms(self, shard):
        e = self._L.haf_multi_engine(self._h, shard)
        ms = (C.c_float * len(STAGES))()
        if self._L.haf_get_stage_ms(C.c_void_p(e), ms) != HAF_OK:
            return None
        return dict(zip(STAGES, list(ms)))
