"""ctypes binding of oracle/libhaforacle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/haf_oracle.h).  The product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """(Re)build the oracle library; also the reference libsvm tools when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class Cfg(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("n_rolls", C.c_int), ("roll_step_deg", C.c_int),
                ("z_shift", C.c_float), ("graspval_top", C.c_int), ("nshaf", C.c_int), ("skip_text", C.c_int),
                ("probability", C.c_int)]


class Input(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("length_x", C.c_float), ("length_y", C.c_float),
                ("approach", C.c_double * 3), ("show_only_best", C.c_int), ("gripper_width", C.c_int)]


class Output(C.Structure):
    _fields_ = [("eval", C.c_int), ("gp1", C.c_double * 3), ("gp2", C.c_double * 3), ("avg", C.c_double * 3),
                ("av", C.c_double * 3), ("roll", C.c_float), ("row", C.c_int), ("col", C.c_int),
                ("roll_idx", C.c_int), ("top", C.c_int), ("n_evals", C.c_long), ("rolls_done", C.c_int)]


class Debug(C.Structure):
    _fields_ = [("heights", C.c_void_p), ("integral", C.c_void_p), ("mask", C.c_void_p), ("labels", C.c_void_p),
                ("dec", C.c_void_p), ("graspseval", C.c_void_p), ("roll_best", C.c_void_p), ("M", C.c_void_p),
                ("sabs", C.c_void_p), ("prob", C.c_void_p), ("graspsgrid", C.c_void_p)]


class Features(C.Structure):
    _fields_ = [("n", C.c_int), ("reg", C.POINTER(C.c_int)), ("w", C.POINTER(C.c_float))]


class Range(C.Structure):
    _fields_ = [("lower", C.c_double), ("upper", C.c_double), ("max_index", C.c_int),
                ("fmin", C.POINTER(C.c_double)), ("fmax", C.POINTER(C.c_double)),
                ("present", C.POINTER(C.c_ubyte))]


class Model(C.Structure):
    _fields_ = [("svm_type", C.c_int), ("kernel_type", C.c_int), ("gamma", C.c_double), ("rho", C.c_double),
                ("nr_class", C.c_int), ("l", C.c_int), ("nSV", C.c_int * 2), ("label", C.c_int * 2),
                ("D", C.c_int), ("coef", C.POINTER(C.c_double)), ("sv", C.POINTER(C.c_double)),
                ("has_prob", C.c_int), ("probA", C.c_double), ("probB", C.c_double), ("degree", C.c_int), ("coef0", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libhaforacle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.hafo_features_load.restype = C.POINTER(Features)
        L.hafo_features_load.argtypes = [C.c_char_p]
        L.hafo_range_load.restype = C.POINTER(Range)
        L.hafo_range_load.argtypes = [C.c_char_p]
        L.hafo_model_load.restype = C.POINTER(Model)
        L.hafo_model_load.argtypes = [C.c_char_p]
        L.hafo_features_free.argtypes = [C.POINTER(Features)]
        L.hafo_range_free.argtypes = [C.POINTER(Range)]
        L.hafo_model_free.argtypes = [C.POINTER(Model)]
        L.hafo_q4.restype = C.c_double
        L.hafo_q4.argtypes = [C.c_float]
        L.hafo_q6.restype = C.c_double
        L.hafo_q6.argtypes = [C.c_double]
        L.hafo_decision.restype = C.c_double
        L.hafo_decision.argtypes = [C.POINTER(Model), C.c_void_p]
        L.hafo_decision_rows.argtypes = [C.POINTER(Model), C.c_void_p, C.c_long, C.c_void_p]
        L.hafo_label_gridval.restype = C.c_int
        L.hafo_label_gridval.argtypes = [C.c_int]
        L.hafo_feature_values.argtypes = [C.POINTER(Features), C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.hafo_feature_line.restype = C.c_int
        L.hafo_feature_line.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.hafo_scale_row.argtypes = [C.POINTER(Range), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.hafo_vote.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hafo_vote_f.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hafo_probability.restype = C.c_int
        L.hafo_probability.argtypes = [C.POINTER(Model), C.c_double, C.POINTER(C.c_double)]
        L.hafo_probability_gridval.restype = C.c_float
        L.hafo_probability_gridval.argtypes = [C.c_char_p]
        L.hafo_transform.argtypes = [C.POINTER(Cfg), C.POINTER(Input), C.c_int, C.c_int, C.c_void_p]
        L.hafo_height_grid.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.hafo_integral.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p]
        L.hafo_mask.argtypes = [C.POINTER(Cfg), C.POINTER(Input), C.c_int, C.c_void_p, C.c_void_p]
        L.hafo_run.restype = C.c_int
        L.hafo_run.argtypes = [C.POINTER(Cfg), C.POINTER(Features), C.POINTER(Range), C.POINTER(Model), C.c_void_p,
                               C.c_size_t, C.c_size_t, C.POINTER(Input), C.POINTER(Output), C.POINTER(Debug)]
        L.hafo_set_variant.argtypes = [C.c_int]
        L.hafo_set_roll_first.argtypes = [C.c_int]
        L.hafo_get_variant.restype = C.c_int
        L.hafo_dump_feature_file.restype = C.c_long
        L.hafo_dump_feature_file.argtypes = [C.POINTER(Cfg), C.POINTER(Features), C.c_void_p, C.c_size_t, C.c_size_t,
                                             C.POINTER(Input), C.c_int, C.c_char_p]
        _LIB = L
    return _LIB


# alternative evaluation orders of the unpinned third-party arithmetic (haf_oracle.h); 0 = the definition of record
V_EIGEN_TREE, V_CHAIN_RTL, V_PCL_SSE, V_FMA, V_INTEGRAL_COLFIRST = 1, 2, 4, 8, 16


def set_variant(flags):
    lib().hafo_set_variant(int(flags))


def ref_dir():
    return os.path.join(_HERE, "_ref")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_cfg(H=56, W=56, n_rolls=12, roll_step_deg=15, z_shift=0.15, graspval_top=119, nshaf=302, skip_text=0, probability=0):
    return Cfg(H, W, n_rolls, roll_step_deg, z_shift, graspval_top, nshaf, skip_text, probability)


def make_input(center=(0, 0, 0), length_x=32, length_y=32, approach=(0, 0, 1), show_only_best=0, gripper_width=1):
    return Input((C.c_double * 3)(*center), float(length_x), float(length_y), (C.c_double * 3)(*approach),
                 int(show_only_best), int(gripper_width))


class Oracle:
    """Loaded (features, range, model) triple plus the request runner."""

    def __init__(self, feature_file, range_file, model_file):
        L = lib()
        self.ft = L.hafo_features_load(feature_file.encode())
        self.rg = L.hafo_range_load(range_file.encode())
        self.m = L.hafo_model_load(model_file.encode()) if model_file else None
        if not self.ft or not self.rg or (model_file and not self.m):
            raise RuntimeError("oracle: cannot load %s / %s / %s" % (feature_file, range_file, model_file))

    @property
    def n_features(self):
        return self.ft.contents.n

    def feature_table(self):
        n = self.ft.contents.n
        reg = np.ctypeslib.as_array(self.ft.contents.reg, shape=(n, 16)).copy()
        w = np.ctypeslib.as_array(self.ft.contents.w, shape=(n, 4)).copy()
        return reg, w

    def range_table(self):
        r = self.rg.contents
        n = r.max_index + 1
        return (r.lower, r.upper, np.ctypeslib.as_array(r.fmin, shape=(n,)).copy(),
                np.ctypeslib.as_array(r.fmax, shape=(n,)).copy(),
                np.ctypeslib.as_array(r.present, shape=(n,)).copy())

    def model_arrays(self):
        m = self.m.contents
        coef = np.ctypeslib.as_array(m.coef, shape=(m.l,)).copy()
        sv = np.ctypeslib.as_array(m.sv, shape=(m.l, m.D)).copy()
        return dict(gamma=m.gamma, rho=m.rho, l=m.l, D=m.D, nSV=(m.nSV[0], m.nSV[1]),
                    label=(m.label[0], m.label[1]), coef=coef, sv=sv)

    def feature_values(self, window, nshaf=302):
        window = np.ascontiguousarray(window, dtype=np.float32)
        assert window.shape == (15, 15)
        out = np.empty(self.n_features, dtype=np.float32)
        lib().hafo_feature_values(self.ft, nshaf, _p(window), 15, _p(out))
        return out

    def feature_line(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float32)
        buf = C.create_string_buffer(len(vals) * 32 + 16)
        n = lib().hafo_feature_line(_p(vals), len(vals), buf, len(buf))
        return buf.raw[:n].decode()

    def scale_row(self, q4, D, skip=None, skip_text=0):
        q4 = np.ascontiguousarray(q4, dtype=np.float64)
        n = len(q4)
        if skip is None:
            skip = np.zeros(n + 1, dtype=np.uint8)
        skip = np.ascontiguousarray(skip, dtype=np.uint8)
        xs = np.zeros(D, dtype=np.float64)
        lib().hafo_scale_row(self.rg, _p(skip), _p(q4), n, skip_text, _p(xs), D)
        return xs

    def decision(self, xs):
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        assert xs.shape[-1] == self.m.contents.D
        if xs.ndim == 1:
            return lib().hafo_decision(self.m, _p(xs))
        dec = np.empty(xs.shape[0], dtype=np.float64)
        lib().hafo_decision_rows(self.m, _p(xs), xs.shape[0], _p(dec))
        return dec

    def probability(self, dec):
        """svm_predict_probability on a decision value: (label, p0, p1); None without probA/probB."""
        pr = (C.c_double * 2)()
        lab = lib().hafo_probability(self.m, float(dec), pr)
        return (lab, pr[0], pr[1]) if self.m.contents.has_prob else None

    def run(self, xyz, cfg, inp, debug=True, roll_first=0):
        """hafo_run; roll_first > 0 (test hook): only rolls [roll_first, cfg.n_rolls) are scored, arrays stay indexed by absolute roll"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        out = Output()
        R, H, W = cfg.n_rolls, cfg.H, cfg.W
        dbg = None
        arrays = {}
        if debug:
            arrays = dict(heights=np.zeros((R, H, W), np.float32), integral=np.zeros((R, H + 1, W + 1), np.float32),
                          mask=np.zeros((R, H, W), np.uint8), labels=np.full((R, H, W), -1, np.int8),
                          dec=np.full((R, H, W), np.nan, np.float64), graspseval=np.zeros((R, H, W), np.float32),
                          roll_best=np.full((R, 3), -1, np.int32), M=np.zeros((R, 16), np.float32),
                          sabs=np.zeros((R, H, W), np.float64))
            if cfg.probability:
                arrays.update(prob=np.full((R, H, W, 2), np.nan, np.float64), graspsgrid=np.full((R, H, W), -1, np.float32))
            dbg = Debug(*[_p(arrays[k]) if k in arrays else None
                          for k in ("heights", "integral", "mask", "labels", "dec", "graspseval", "roll_best", "M", "sabs",
                                    "prob", "graspsgrid")])
        lib().hafo_set_roll_first(int(roll_first))
        try:
            rc = lib().hafo_run(C.byref(cfg), self.ft, self.rg, self.m, _p(xyz), xyz.shape[0], xyz.shape[1],
                                C.byref(inp), C.byref(out), C.byref(dbg) if dbg else None)
        finally:
            lib().hafo_set_roll_first(0)
        if rc != 0:
            raise RuntimeError("hafo_run failed: %d" % rc)
        res = dict(eval=out.eval, gp1=tuple(out.gp1), gp2=tuple(out.gp2), avg=tuple(out.avg), av=tuple(out.av),
                   roll=out.roll, row=out.row, col=out.col, roll_idx=out.roll_idx, top=out.top,
                   n_evals=out.n_evals, rolls_done=out.rolls_done)
        res.update(arrays)
        return res

    def dump_feature_file(self, xyz, cfg, inp, roll, path):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        return lib().hafo_dump_feature_file(C.byref(cfg), self.ft, _p(xyz), xyz.shape[0], xyz.shape[1], C.byref(inp),
                                            roll, path.encode())


def q4(v):
    return lib().hafo_q4(float(np.float32(v)))


def q6(v):
    return lib().hafo_q6(float(v))
