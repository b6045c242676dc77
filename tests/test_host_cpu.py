"""CPU-only tests of the product's host side: the C-ABI library loads and exports what include/hafgrasp.h declares,
the decimal round-trip arithmetic (host build of csrc/decq.h) equals glibc printf+strtod, and the parsers, per-roll
geometry, cross-roll rule and pose agree with the oracle.  No kernel is launched here."""
import ctypes as C
import json
import os
import re
import struct

import numpy as np
import pytest

import pcdio
from haf_grasping_amd import capi
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    with open(os.path.join(ROOT, "include", "hafgrasp.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b(haf_[a-z_0-9]+)\s*\(", text))
    assert {"haf_create", "haf_score", "haf_score_batch", "haf_score_rolls", "haf_finalize", "haf_destroy",
            "haf_last_error", "haf_get_roll_grid", "haf_pcd_load"} <= names
    assert {"haf_create_multi", "haf_score_sharded", "haf_score_batch_sharded", "haf_roll_pose", "haf_debug_fetch_attr"} <= names
    for n in sorted(names):
        assert hasattr(L, n), n
    assert L.haf_abi_version() == 2
    # the product library carries no test hooks; the testing build has the ABI and the hooks
    assert not hasattr(L, "haf_test_finalize") and not hasattr(L, "haf_test_decq_device")
    T = capi.testlib()
    for n in sorted(names) + ["haf_test_finalize", "haf_test_roll_pose", "haf_test_decq_device"]:
        assert hasattr(T, n), n


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(capi.RollRecord) == 16
    assert C.sizeof(capi.Cloud) == 32
    assert C.sizeof(capi.Config) == 88 and C.sizeof(capi.GraspInput) == 80 and C.sizeof(capi.GraspOutput) == 144
    cfg = capi.default_config()
    assert (cfg.grid_h, cfg.grid_w, cfg.n_rolls, cfg.roll_step_deg, cfg.graspval_top) == (56, 56, 12, 15, 119)
    assert cfg.nr_features_without_shaf == 302 and abs(cfg.z_shift - 0.15) < 1e-7
    gi = capi.default_input()
    assert (gi.grasp_area_length_x, gi.grasp_area_length_y, gi.gripper_opening_width) == (32.0, 44.0, 1)
    assert tuple(gi.approach_vector) == (0.0, 0.0, 1.0) and gi.max_calculation_time == 50.0


def test_create_fails_loudly_without_gpu(data_dir, golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.HafError) as ei:
        capi.Engine(os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures"),
                    os.path.join(golden_dir, "surrogate.model"))
    assert ei.value.code == capi.HAF_E_DEVICE and "no CPU fallback" in str(ei.value)


def _glibc_q(x, digits):
    return float(("%." + str(digits) + "g") % x)      # CPython: PyOS_double_to_string -> correctly rounded, == glibc


def _bits(x):
    return struct.pack("<d", x)


@pytest.mark.parametrize("digits", [4, 6])
def test_decq_host_matches_printf_strtod(digits):
    L = capi.testlib()
    rng = np.random.RandomState(digits)
    vals = []
    # feature-like magnitudes, float32 inputs for %.4g, doubles for %g
    for scale in [1e-12, 1e-9, 1e-6, 1e-3, 1.0, 30.0, 1e3, 1e5, 1e7, 1e12, 1e-17, 1e20]:
        v = rng.standard_normal(4000) * scale
        vals.append(v.astype(np.float32).astype(np.float64) if digits == 4 else v)
    # exact ties and near-ties: k + 0.5 at the rounding position, +- 1 ulp
    base = rng.randint(10 ** (digits - 1), 10 ** digits, size=3000).astype(np.float64)
    for e in range(-8, 12):
        t = (base + 0.5) * (10.0 ** e)
        vals += [t, np.nextafter(t, np.inf), np.nextafter(t, -np.inf)]
        if digits == 4:
            vals.append(t.astype(np.float32).astype(np.float64))
    # powers of ten and their neighbours (decimal-exponent boundary)
    p = np.array([10.0 ** e for e in range(-22, 23)])
    vals += [p, np.nextafter(p, np.inf), np.nextafter(p, -np.inf), p * 9.9995, p * 9.99995, p * 9.999995]
    vals.append(np.array([0.0, -0.0, 1.0, -1.0, 0.5, 123.25, 123.75, 1234.5, 9999.5, 99999.5, 999999.5, 0.03, 2.0 ** -53 - 1.0]))
    allv = np.concatenate(vals)
    allv = np.concatenate([allv, -allv])
    bad = 0
    for x in allv:
        x = float(x)
        got = L.haf_test_decq_host(x, digits)
        want = _glibc_q(x, digits)
        if _bits(got) != _bits(want):
            bad += 1
            assert bad < 5, (x, got, want)
    assert bad == 0


def test_decq4_float_entry_matches_printf_strtod():
    """The fp32 "%.4g" entry the kernels use (digits code 40): exact-product shortcut for k <= 12, general path beyond."""
    L = capi.testlib()
    rng = np.random.RandomState(40)
    parts = [(rng.standard_normal(30000) * s).astype(np.float32) for s in (1e-12, 1e-9, 1e-7, 1e-4, 1e-2, 1.0, 30.0, 1e3, 1e5, 1e9)]
    base = rng.randint(1000, 10000, size=4000).astype(np.float64)
    for e in range(-10, 8):                       # exact ties representable in fp32 and their fp32 neighbours
        t = ((base + 0.5) * 10.0 ** e).astype(np.float32)
        parts += [t, np.nextafter(t, np.float32(np.inf)), np.nextafter(t, np.float32(-np.inf))]
    p10 = np.array([10.0 ** e for e in range(-20, 21)], np.float32)
    parts += [p10, np.nextafter(p10, np.float32(np.inf)), np.nextafter(p10, np.float32(-np.inf)),
              np.array([0.0, -0.0, 123.25, 123.75, 1234.5, 9999.5, 99995.0, 1e-45, 3.4e38], np.float32)]
    x = np.concatenate(parts)
    x = np.concatenate([x, -x])
    for v in x:
        v = float(v)
        assert _bits(L.haf_test_decq_host(v, 40)) == _bits(_glibc_q(v, 4)), v


def test_decq_wide_window_against_glibc():
    """Outside 1e-19..1e26 the double-double path is used; it is expected (not proven) to agree with glibc."""
    L = capi.testlib()
    rng = np.random.RandomState(7)
    for digits in (4, 6):
        for e in list(range(-300, -20, 7)) + list(range(27, 300, 7)):
            for x in rng.uniform(1, 10, size=40) * 10.0 ** e:
                assert _bits(L.haf_test_decq_host(float(x), digits)) == _bits(_glibc_q(float(x), digits)), (x, digits)
    # float32 subnormals and extremes, as %.4g sees them
    for x in np.array([1e-45, 3e-42, 1.17549435e-38, 3.4028235e38], np.float32).astype(np.float64):
        assert _bits(L.haf_test_decq_host(float(x), 4)) == _bits(_glibc_q(float(x), 4))
    assert np.isnan(L.haf_test_decq_host(float("nan"), 4)) and L.haf_test_decq_host(float("inf"), 6) == float("inf")


def test_scale_host_matches_oracle(data_dir):
    L = capi.testlib()
    orc = O.Oracle(os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures"), None)
    lo, up, fmin, fmax, present = orc.range_table()
    rng = np.random.RandomState(3)
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    for _ in range(60):
        feats = (rng.standard_normal(324) * rng.choice([0.01, 1, 5, 30])).astype(np.float32)
        feats[rng.randint(0, 323, 5)] = 0.0
        q4 = np.array([O.q4(v) for v in feats])
        q4[7] = fmin[8]
        q4[9] = fmax[10]
        want = orc.scale_row(q4, 323, skip)
        got = np.array([L.haf_test_scale_host(q4[k], fmin[k + 1], fmax[k + 1], lo, up) for k in range(323)])
        assert (got == want).all()


def test_parsers_match_oracle(data_dir, golden_dir):
    L = capi.testlib()
    orc = O.Oracle(os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures"),
                   os.path.join(golden_dir, "surrogate.model"))
    n = C.c_int()
    reg = np.zeros((400, 16), np.int32)
    w = np.zeros((400, 4), np.float32)
    assert L.haf_test_feature_table(os.path.join(data_dir, "Features.txt").encode(), C.byref(n), reg.ctypes.data,
                                    w.ctypes.data, 400) == 0
    oreg, ow = orc.feature_table()
    assert n.value == 324 and (reg[:324] == oreg).all() and (w[:324] == ow).all()

    lo, up, mi = C.c_double(), C.c_double(), C.c_int()
    fmin, fmax, pres = np.zeros(400), np.zeros(400), np.zeros(400, np.uint8)
    assert L.haf_test_range_table(os.path.join(data_dir, "range21062012_allfeatures").encode(), C.byref(lo), C.byref(up),
                                  C.byref(mi), fmin.ctypes.data, fmax.ctypes.data, pres.ctypes.data, 400) == 0
    olo, oup, ofmin, ofmax, opres = orc.range_table()
    assert (lo.value, up.value, mi.value) == (olo, oup, 323)
    assert (fmin[:324] == ofmin).all() and (fmax[:324] == ofmax).all() and (pres[:324] == opres).all()

    om = orc.model_arrays()
    g, r, nsv, dim = C.c_double(), C.c_double(), C.c_int(), C.c_int()
    cls, lab = (C.c_int * 2)(), (C.c_int * 2)()
    coef = np.zeros(om["l"])
    sv = np.zeros((om["l"], om["D"]))
    assert L.haf_test_model(os.path.join(golden_dir, "surrogate.model").encode(), C.byref(g), C.byref(r), C.byref(nsv),
                            C.byref(dim), cls, lab, coef.ctypes.data, sv.ctypes.data, sv.size) == 0
    assert (g.value, r.value, nsv.value, dim.value) == (om["gamma"], om["rho"], om["l"], om["D"])
    assert tuple(cls) == om["nSV"] and tuple(lab) == om["label"]
    assert (coef == om["coef"]).all() and (sv == om["sv"]).all()


def test_model_parser_reads_every_vector_kernel(golden_dir, tmp_path):
    """Round 5: svm_load_model's header (svm.cpp:2714-2860) for libsvm's four vector kernels and both classifier types -- the model
    texts the reference svm-train wrote (kernel_models.npz) -- and the two things that are refused: a precomputed kernel (no attribute
    vectors to score) and a regression / one-class type (not a classifier svm-predict's label path serves)."""
    import models
    L = capi.testlib()
    want = {"linear": (0, 0, 0.0, 0.0), "poly": (1, 3, 1.0, 0.0031), "sigmoid": (3, 0, -0.5, 0.0031), "nu_rbf": (2, 0, 0.0, 0.0031)}
    for kname, (kt, deg, c0, gam) in want.items():
        path = models.unpack_kernel_model(golden_dir, kname, str(tmp_path / (kname + ".model")))
        k, d, c, g = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        assert L.haf_test_model_kernel(path.encode(), C.byref(k), C.byref(d), C.byref(c), C.byref(g)) == 0, kname
        assert (k.value, d.value, c.value, g.value) == (kt, deg, c0, gam), (kname, k.value, d.value, c.value, g.value)
        om = O.lib().hafo_model_load(path.encode())
        assert om and (om.contents.kernel_type, om.contents.degree, om.contents.coef0) == (kt, deg, c0)
    text = open(str(tmp_path / "linear.model")).read()
    for bad in (text.replace("kernel_type linear", "kernel_type precomputed"), text.replace("svm_type c_svc", "svm_type epsilon_svr")):
        p = tmp_path / "bad.model"
        p.write_text(bad)
        k, d, c, g = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        assert L.haf_test_model_kernel(str(p).encode(), C.byref(k), C.byref(d), C.byref(c), C.byref(g)) == capi.HAF_E_IO


def test_feature_file_trailing_line_quirks(tmp_path):
    """fv.cpp:60-82: a final EMPTY line adds a phantom all-zero feature; a last line WITHOUT newline is dropped."""
    L = capi.testlib()
    row = "\t".join(["1", "2", "3", "4"] * 4 + ["1", "-1", "2", "5"])
    reg = np.zeros((8, 16), np.int32)
    w = np.zeros((8, 4), np.float32)
    n = C.c_int()
    for text, expect in [(row + "\n" + row + "\n", 2), (row + "\n" + row + "\n\n", 3), (row + "\n" + row, 1)]:
        p = tmp_path / "f.txt"
        p.write_text(text)
        assert L.haf_test_feature_table(str(p).encode(), C.byref(n), reg.ctypes.data, w.ctypes.data, 8) == 0
        assert n.value == expect
        oft = O.lib().hafo_features_load(str(p).encode())
        assert oft.contents.n == expect
    assert w[0, 3] == 0.0 and w[0, 2] == 2.0      # 4th weight dropped


def test_pcd_reader_matches_test_reader(data_dir):
    for name in sorted(os.listdir(data_dir)):
        if name.endswith(".pcd"):
            a = capi.load_pcd(os.path.join(data_dir, name))
            b = pcdio.load_pcd(os.path.join(data_dir, name))
            assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), name


def test_pcd_reader_organised_clouds_nan_and_extra_fields(tmp_path):
    """organised clouds (HEIGHT > 1) with NaN holes, as a depth camera delivers them, extra fields around
    x/y/z (rgb, a 3-float normal), in the ascii and binary forms; the points come back in file order with the NaNs in place
    (the binning kernel drops them, like the comparisons of generate_grid do)."""
    rng = np.random.RandomState(3)
    W, H = 7, 5
    xyz = rng.uniform(-0.3, 0.3, size=(W * H, 3)).astype(np.float32)
    xyz[[3, 4, 17]] = np.nan
    xyz[20, 2] = np.nan
    rgb = rng.randint(0, 1 << 24, size=W * H).astype(np.uint32)
    nrm = rng.standard_normal((W * H, 3)).astype(np.float32)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS rgb x y normal z\nSIZE 4 4 4 4 4\nTYPE U F F F F\n"
            "COUNT 1 1 1 3 1\nWIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\n" % (W, H, W * H))
    rows = []
    for i in range(W * H):
        f = lambda v: "nan" if np.isnan(v) else repr(float(v))
        rows.append("%d %s %s %s %s %s %s" % (rgb[i], f(xyz[i, 0]), f(xyz[i, 1]), f(nrm[i, 0]), f(nrm[i, 1]), f(nrm[i, 2]), f(xyz[i, 2])))
    pa = tmp_path / "org_ascii.pcd"
    pa.write_text(head + "DATA ascii\n" + "\n".join(rows) + "\n")
    rec = np.zeros(W * H, dtype=[("rgb", "<u4"), ("x", "<f4"), ("y", "<f4"), ("n", "<f4", 3), ("z", "<f4")])
    rec["rgb"], rec["x"], rec["y"], rec["n"], rec["z"] = rgb, xyz[:, 0], xyz[:, 1], nrm, xyz[:, 2]
    pb = tmp_path / "org_binary.pcd"
    pb.write_bytes((head + "DATA binary\n").encode() + rec.tobytes())
    for path in (pa, pb):
        got = capi.load_pcd(str(path))
        assert got.shape == (W * H, 3)
        assert (np.isnan(got) == np.isnan(xyz)).all()
        assert (got[~np.isnan(xyz)].view(np.uint32) == xyz[~np.isnan(xyz)].view(np.uint32)).all(), path
        ref = pcdio.load_pcd(str(path))
        assert (np.isnan(ref) == np.isnan(xyz)).all()


INPUTS = [dict(), dict(grasp_area_center=(0.13, 0.25, 0.02), grasp_area_length_x=56, grasp_area_length_y=56),
          dict(approach_vector=(0.2, -0.1, 1.0)), dict(approach_vector=(0.0, 0.0, -2.0), gripper_opening_width=2),
          dict(approach_vector=(1.0, 1.0, 0.3), grasp_area_center=(-0.05, 0.01, 0.1), grasp_area_length_x=47.9)]


def _oracle_input(kw):
    return O.make_input(center=kw.get("grasp_area_center", (0, 0, 0)), length_x=kw.get("grasp_area_length_x", 32),
                        length_y=kw.get("grasp_area_length_y", 44), approach=kw.get("approach_vector", (0, 0, 1)),
                        show_only_best=kw.get("show_only_best_grasp", 0), gripper_width=kw.get("gripper_opening_width", 1))


@pytest.mark.parametrize("kw", INPUTS)
@pytest.mark.parametrize("step,rolls", [(15, 12), (9, 20), (5, 36)])
def test_roll_geometry_matches_oracle(kw, step, rolls):
    L = capi.testlib()
    cfg = capi.default_config(n_rolls=rolls, roll_step_deg=step)
    gi = capi.default_input(**kw)
    ocfg = O.make_cfg(n_rolls=rolls, roll_step_deg=step)
    oin = _oracle_input(kw)
    for roll in range(rolls):
        out = np.zeros(22, np.float32)
        m16 = np.zeros(16, np.float32)
        m16p = np.zeros(16, np.float32)
        L.haf_test_roll_geo(C.byref(cfg), C.byref(gi), roll, out.ctypes.data, m16.ctypes.data, m16p.ctypes.data)
        om = np.zeros(16, np.float32)
        O.lib().hafo_transform(C.byref(ocfg), C.byref(oin), roll, 0, om.ctypes.data)
        assert (m16.view(np.uint32) == om.view(np.uint32)).all()
        assert (out[:12].view(np.uint32) == om[:12].view(np.uint32)).all()
        O.lib().hafo_transform(C.byref(ocfg), C.byref(oin), roll, 1, om.ctypes.data)
        assert (m16p.view(np.uint32) == om.view(np.uint32)).all()


def test_finalize_matches_oracle_pose(data_dir, golden_dir):
    """Cross-roll rule + pose (host code) from the ORACLE's per-roll winners and height grids."""
    L = capi.testlib()
    orc = O.Oracle(os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures"),
                   os.path.join(golden_dir, "surrogate.model"))
    cases = [("pcd2", dict(grasp_area_length_x=32, grasp_area_length_y=32)),
             ("pcd2", dict(grasp_area_length_x=32, grasp_area_length_y=32, show_only_best_grasp=1)),
             ("pcd2", dict(approach_vector=(0.2, -0.1, 1.0))), ("pcd6", dict()), ("pcd7", dict()),
             ("plastic_mug2", dict(grasp_area_center=(0.02, 0.0, 0.0)))]
    for name, kw in cases:
        xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
        ocfg = O.make_cfg()
        r = orc.run(xyz, ocfg, _oracle_input(kw))
        rec = np.zeros(12, capi.ROLL_RECORD_DTYPE)
        for roll in range(12):
            row, col, val = r["roll_best"][roll]
            if val == -1 and row == -1:         # roll not executed by the oracle (early exit): any content will do
                continue
            h = r["heights"][roll]
            win = h[max(0, row - 4):row + 5, max(0, col - 4):col + 4]
            rec[roll] = (val, row, col, max(np.float32(-10.0), win.max()), int(r["mask"][roll].sum()))
        if kw.get("show_only_best_grasp"):
            # rolls the oracle skipped must not matter: poison them
            for roll in range(r["rolls_done"], 12):
                rec[roll] = (123, 5, 5, 9.0, 0)
        out = capi.GraspOutput()
        assert L.haf_test_finalize(C.byref(capi.default_config()), C.byref(capi.default_input(**kw)), rec.ctypes.data,
                                   C.byref(out)) == 0
        assert (out.eval, out.best_row, out.best_col, out.best_roll, out.best_vote) == \
               (r["eval"], r["row"], r["col"], r["roll_idx"], r["top"]), name
        assert out.rolls_done == r["rolls_done"]
        np.testing.assert_allclose(tuple(out.grasp_point1), r["gp1"], atol=1e-6)
        np.testing.assert_allclose(tuple(out.grasp_point2), r["gp2"], atol=1e-6)
        np.testing.assert_allclose(tuple(out.averaged_grasp_point), r["avg"], atol=1e-6)
        np.testing.assert_allclose(tuple(out.approach_vector), r["av"], atol=1e-7)
        assert out.roll == r["roll"]


def test_create_reports_file_errors_before_touching_a_device(data_dir, golden_dir, tmp_path):
    """Unreadable / unsupported construction files fail with HAF_E_IO and a message, on any machine."""
    f = os.path.join(data_dir, "Features.txt")
    r = os.path.join(data_dir, "range21062012_allfeatures")
    m = os.path.join(golden_dir, "surrogate.model")
    bad_kernel = tmp_path / "lin.model"
    bad_kernel.write_text(open(m).read().replace("kernel_type rbf", "kernel_type precomputed", 1))     # (round 5: the four vector kernels are served)
    three = tmp_path / "three.model"
    three.write_text(open(m).read().replace("nr_class 2", "nr_class 3", 1))
    trunc = tmp_path / "trunc.model"
    trunc.write_text("".join(open(m).readlines()[:20]))
    norange = tmp_path / "norange"
    norange.write_text("y\n-1 1\n0 1\n")
    cases = [((str(tmp_path / "nope.txt"), r, m), "cannot open feature file"),
             ((f, str(tmp_path / "nope"), m), "cannot open range file"),
             ((f, str(norange), m), "no x section"),
             ((f, r, str(tmp_path / "nope.model")), "cannot open model file"),
             ((f, r, str(bad_kernel)), "no attribute vectors"),
             ((f, r, str(three)), "nr_class must be 2"),
             ((f, r, str(trunc)), "fewer SV lines")]
    for args, msg in cases:
        with pytest.raises(capi.HafError) as ei:
            capi.Engine(*args)
        assert ei.value.code == capi.HAF_E_IO and msg in str(ei.value), (args, str(ei.value))
    with pytest.raises(capi.HafError) as ei:
        capi.Engine(f, r, m, grid_h=56, grid_w=64)
    assert ei.value.code == capi.HAF_E_ARG and "square" in str(ei.value)


def test_screening_band_host_pieces():
    """Host arithmetic behind the screening pass's guard band: the spectral-norm bound is an UPPER bound of
    numpy's largest singular value and at most d^(1/128) above it; the three-term fp16 split reports exactly what its three
    products add up to and misses at most 2^-25 + 2^-33 |a|."""
    L = capi.testlib()
    rng = np.random.RandomState(5)
    for n, d, kind in ((300, 40, "uniform"), (64, 324, "uniform"), (500, 324, "lowrank"), (10, 7, "zero")):
        if kind == "uniform":
            M = rng.uniform(-1, 1, size=(n, d))
        elif kind == "lowrank":
            M = np.outer(rng.uniform(-1, 1, n), rng.uniform(-1, 1, d)) + 1e-3 * rng.standard_normal((n, d))
        else:
            M = np.zeros((n, d))
        M = np.ascontiguousarray(M, dtype=np.float64)
        got = L.haf_test_sigma_upper(M.ctypes.data, n, d)
        want = np.linalg.svd(M, compute_uv=False)[0] if kind != "zero" else 0.0
        assert want <= got <= want * d ** (1.0 / 128.0) * (1 + 1e-6) + 1e-300, (kind, got, want)
    parts = np.zeros(3, np.float32)
    for a in list(-rng.uniform(0, 40, 200)) + [0.0, -1e-9, -0.5, -1.0, -65000.0, 3.25]:
        rep = L.haf_test_split3(float(a), parts.ctypes.data)
        assert rep == float(parts[0]) + (float(parts[1]) + float(parts[2])) / 4096.0
        assert abs(rep - a) <= 2.0 ** -25 + 2.0 ** -33 * abs(a), (a, rep)
        for v in parts:                                     # every part is an fp16 value, none of them subnormal
            assert float(np.float16(v)) == float(v) and (v == 0 or abs(v) >= 2.0 ** -14)


def test_fast_decimal_path_of_the_screening_features():
    """decq4_float_scr (table-driven, branch-free, screening pass only): for v == 0 and 1e-9 <= |v| < 1e4 it picks the same four
    digits as the exact "%.4g" round trip -- the result differs by the rounding of one multiplication at most; everything else
    comes back NaN (which makes the feature kernel distrust the whole evaluation), fp32 subnormals come back 0."""
    L = capi.testlib()
    rng = np.random.RandomState(11)
    edge = []
    for e in range(-10, 6):                                   # both sides of every power of ten and of every rounding carry
        for m in (1.0, 9.9995, 9.99949, 9.99951, 1.0005, 1.00049, 5.0):
            x = np.float32(m * 10.0 ** e)
            edge += [x, np.nextafter(x, np.float32(0)), np.nextafter(x, np.float32(np.inf))]
    vals = np.concatenate([rng.standard_normal(20000) * s for s in (1e-8, 1e-5, 1e-2, 1.0, 30.0, 3e3)] +
                          [np.array([0.0, -0.0, 999.95, 9999.5, 1e-9, 0.12345, 0.12355, 2.5e-7, 9999.4999]), np.array(edge)]
                          ).astype(np.float32)
    vals = np.concatenate([vals, -vals])
    n_in = 0
    for v in vals:
        v = float(v)
        got = L.haf_test_decq4_scr(v)
        want = float("%.4g" % v)
        inside = (v == 0.0) or (1e-9 <= abs(v) < 1e4)
        if inside:
            n_in += 1
            assert abs(got - want) <= 2.3e-16 * abs(want), (v, got, want)
            assert np.signbit(got) == np.signbit(want), (v, got, want)
        else:
            assert np.isnan(got) or (abs(v) < 1.2e-38 and got == 0.0), (v, got, want)
    assert n_in > 0.9 * len(vals)
    for v in (1e-12, 1e5, float("inf"), float("-inf"), float("nan"), 3e38, 1e4, 10000.001):
        assert np.isnan(L.haf_test_decq4_scr(v)), v
    assert L.haf_test_decq4_scr(1e-40) == 0.0


def test_roll_pose_rule_and_pose(data_dir, golden_dir):
    """haf_roll_pose = what show_predicted_gps hands to transform_gp_in_wcs_and_publish for ONE roll (server.cpp:962-969):
    published iff !show_only_best and vote > 70, eval = max(vote - 20, 10); the pose of the overall winner's roll equals
    haf_finalize's pose (same record, same transform)."""
    L = capi.testlib()
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        gold = json.load(f)
    g = gold["pcd2/C2"]
    rec = np.zeros(12, capi.ROLL_RECORD_DTYPE)
    for r, (row, col, vote) in enumerate(g["roll_best"]):
        rec[r] = (vote, row, col, 0.25 + 0.01 * r, g["masked"][r])
    cfg = capi.default_config()
    for show_best in (0, 1):
        gi = capi.default_input(grasp_area_length_x=32, grasp_area_length_y=32, show_only_best_grasp=show_best)
        fin = capi.GraspOutput()
        assert L.haf_test_finalize(C.byref(cfg), C.byref(gi), rec.ctypes.data, C.byref(fin)) == 0
        for r in range(12):
            out, pub = capi.GraspOutput(), C.c_int32(-1)
            assert L.haf_test_roll_pose(C.byref(cfg), C.byref(gi), rec.ctypes.data, r, C.byref(out), C.byref(pub)) == 0
            vote = int(rec["vote"][r])
            assert pub.value == (1 if (not show_best and vote > 70) else 0)
            assert out.eval == max(vote - 20, 10) and (out.best_row, out.best_col, out.best_roll) == (int(rec["row"][r]), int(rec["col"][r]), r)
            assert abs(out.roll - r * 15 * 3.141592653 / 180) < 1e-6
            if r == fin.best_roll:
                assert tuple(out.grasp_point1) == tuple(fin.grasp_point1) and tuple(out.grasp_point2) == tuple(fin.grasp_point2)
                assert tuple(out.approach_vector) == tuple(fin.approach_vector)
    out, pub = capi.GraspOutput(), C.c_int32()
    assert L.haf_test_roll_pose(C.byref(cfg), C.byref(gi), rec.ctypes.data, 12, C.byref(out), C.byref(pub)) == capi.HAF_E_ARG


def test_hostile_files_return_errors_not_exceptions(data_dir, golden_dir, tmp_path):
    """Numbers read out of the input files are bounded before they size an allocation, and no C++ exception crosses the
    C-ABI (a corrupt model must come back as HAF_E_IO / HAF_E_INTERNAL, not as std::terminate of the action server)."""
    L = capi.testlib()
    feat = os.path.join(data_dir, "Features.txt")
    rng = os.path.join(data_dir, "range21062012_allfeatures")
    model = os.path.join(golden_dir, "surrogate.model")
    text = open(model).read()
    head, body = text.split("SV\n", 1)

    def create(f=feat, r=rng, m=model):
        with pytest.raises(capi.HafError) as ei:
            capi.Engine(str(f), str(r), str(m))
        return ei.value

    # model: absurd total_sv, absurd attribute index, more SVs promised than lines present
    p = tmp_path / "m1.model"
    p.write_text(head.replace("total_sv 172", "total_sv 2000000000") + "SV\n" + body)
    assert create(m=p).code == capi.HAF_E_IO
    p = tmp_path / "m2.model"
    p.write_text(head + "SV\n" + body.replace(" 1:", " 2000000000:", 1))
    assert create(m=p).code == capi.HAF_E_IO
    p = tmp_path / "m3.model"
    p.write_text(head.replace("total_sv 172", "total_sv 4000000").replace("nr_sv 84 88", "nr_sv 2000000 2000000") + "SV\n" + body)
    e = create(m=p)
    assert e.code == capi.HAF_E_IO and "fewer SV lines" in str(e)
    # range file: index far beyond anything the engine takes
    p = tmp_path / "r1"
    p.write_text(open(rng).read() + "2000000000 0 1\n")
    e = create(r=p)
    assert e.code == capi.HAF_E_IO and "exceeds" in str(e)
    # PCD: POINTS that the file cannot back, negative SIZE, compressed sizes that wrap
    err = C.create_string_buffer(256)
    ptr, n = C.POINTER(C.c_float)(), C.c_size_t()
    hdr = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE %s\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
    cases = [("4 4 4", 4000000000, "ascii", b"0 0 0\n"), ("4 4 4", 3000000000, "binary", b"\0" * 24),
             ("-4 4 4", 2, "binary", b"\0" * 24), ("4 4 4", 2, "binary_compressed", struct.pack("<II", 0xFFFFFFF0, 24) + b"\0" * 8),
             ("4 4 4", 2, "binary_compressed", struct.pack("<II", 4, 0xFFFFFFFF) + b"\0\0\0\0")]
    for k, (sz, pts, mode, payload) in enumerate(cases):
        p = tmp_path / ("bad%d.pcd" % k)
        p.write_bytes((hdr % (sz, pts, pts, mode)).encode() + payload)
        rc = L.haf_pcd_load(str(p).encode(), C.byref(ptr), C.byref(n), err, 256)
        assert rc in (capi.HAF_E_IO, capi.HAF_E_INTERNAL) and err.value, (k, rc)


def test_host_paths_under_address_and_ub_sanitizers(golden_dir, tmp_path):
    """CPU sanitizer job (GPU sanitizers are not available on the pool): the engine's host translation units + parsers.cpp built with
    -fsanitize=address,undefined by the ROCm clang (host only) and driven over the parsers with truncated / bit-flipped
    files, the roll geometry, the cross-roll rule and the poses (tests/sanitize/host_paths.cpp).  Any report fails."""
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang")
    csrc = os.path.join(ROOT, "haf_grasping_amd", "csrc")
    from haf_grasping_amd import build as B
    host = B.ENGINE_SOURCES + ["engine_testing.cpp", "parsers.cpp"]           # compiled here, instrumented
    objs = [os.path.join(csrc, os.path.splitext(src)[0] + ".o") for src in B.SOURCES if src not in host] + \
           [os.path.join(csrc, "testkernels_testing.o")]
    if not all(os.path.exists(o) for o in objs):
        from haf_grasping_amd import build as b
        b.build(force=True)
    exe = str(tmp_path / "host_paths")
    flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-DHAF_TESTING", "-I/opt/rocm/include"]
    cmd = [clang] + flags + [os.path.join(csrc, f) for f in host] + [os.path.join(ROOT, "tests", "sanitize", "host_paths.cpp"), "-x", "none"] + objs + \
          ["-fsanitize=address,undefined", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    scratch = tmp_path / "fuzz"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, golden_dir, str(scratch)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "sanitizer job ok" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, \
        (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    shutil.rmtree(str(scratch), ignore_errors=True)


def test_failed_exp_hazard_check_leaves_no_library(tmp_path, monkeypatch):
    """the v_exp_f32 distance check runs BEFORE the libraries get the names up_to_date() looks for, so a build that
    fails it (or cannot run it) is not mistaken for a finished one by the next build() call.  Compiler calls are faked."""
    from haf_grasping_amd import build as b
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    monkeypatch.setattr(b, "HERE", str(tmp_path))
    monkeypatch.setattr(b, "CSRC", str(csrc))
    monkeypatch.setattr(b, "LIB", str(tmp_path / "libhafgrasp.so"))
    monkeypatch.setattr(b, "LIB_TESTING", str(tmp_path / "libhafgrasp_testing.so"))
    for stale in (b.LIB, b.LIB_TESTING):
        with open(stale, "w") as f:
            f.write("stale")

    def fake_call(cmd, **kw):
        with open(cmd[cmd.index("-o") + 1], "w") as f:
            f.write("built")
    monkeypatch.setattr(b.subprocess, "check_call", fake_call)
    monkeypatch.setattr(b, "check_no_spills", lambda *a, **k: {})       # (the faked libraries have no code objects)
    monkeypatch.setattr(b, "check_descriptor_loads", lambda *a, **k: {})

    def failing(*a, **k):
        raise RuntimeError("build check failed: hazard")
    monkeypatch.setattr(b, "check_exp_hazard", failing)
    with pytest.raises(RuntimeError):
        b.build(force=True)
    assert not os.path.exists(b.LIB) and not os.path.exists(b.LIB_TESTING) and not b.up_to_date()
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".unchecked")]
    with pytest.raises(RuntimeError):
        b.build()                                   # NOT "up to date": it tries again (and fails again)
    monkeypatch.setattr(b, "check_exp_hazard", lambda *a, **k: {})
    b.build()
    assert open(b.LIB).read() == "built" and open(b.LIB_TESTING).read() == "built" and b.up_to_date()


def test_contraction_kernels_do_not_spill():
    """Round 4: build() also reads the code objects' metadata and refuses a library whose contraction kernels spill registers (they
    live at two waves of ~250 VGPRs per SIMD; a packed-fp32 form of the polynomial epilogue once compiled to 54 spills).  On the
    library as built: every instance of k_svm_screen / k_svm_rbf_h / k_svm_rbf / k_recheck_i8 is there, none spills, all fit 256."""
    from haf_grasping_amd import build as b
    if not b.up_to_date():
        b.build()
    rep = b.check_no_spills()
    names = " ".join(rep)
    for k in ("k_svm_screenILi0", "k_svm_screenILi1", "k_svm_screenILi2", "k_svm_screenILi3", "k_svm_rbf_hILb1ELb0", "k_svm_rbf_hILb1ELb1",
              "k_svm_rbf_hILb0ELb0", "k_recheck_i8"):
        assert k in names, (k, sorted(rep))
    assert all(sp == 0 and 0 < vg <= 256 for vg, sp in rep.values()), rep


def test_group_parallel_feature_kernels_read_descriptors_by_scalar_loads():
    """Round 5: every instance of k_features<MODE, WAVES> keeps its wave index in an SGPR, so the descriptor words of the wave's attribute
    group are scalar loads; as a VGPR expression they were ~17 vector loads and waits per slot (138 in the slot loop of k_features<2, 8>:
    40 of a C3 request's 250 us).  build() counts the vector loads of every instance in the disassembly and refuses more than the
    staging, list and operand traffic account for; on the library as built: all ten instances, 18-49 loads each."""
    from haf_grasping_amd import build as b
    if not b.up_to_date():
        b.build()
    rep = b.check_descriptor_loads()
    assert len(rep) == 10 and all(0 < n <= b.MAX_VECTOR_LOADS for n in rep.values()), rep


def test_ros_adapter_translation_unit_parses():
    """the catkin-side adapter (ros_shim/calc_grasppoints_action_server_hip.cpp) cannot be built here (no ROS / PCL),
    but it can be TYPE-CHECKED: g++ -fsyntax-only against the minimal declarations of tests/mock_ros/ (own code, see its README).
    Every use of shim_core.h -- goal fields, grid callback, preemption callback, result fields -- must compile."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "tests", "mock_ros"),
           "-I", os.path.join(root, "ros_shim"), "-I", os.path.join(root, "include"),
           os.path.join(root, "ros_shim", "calc_grasppoints_action_server_hip.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    src = open(os.path.join(root, "ros_shim", "calc_grasppoints_action_server_hip.cpp")).read()
    for needle in ("on_grid", "preempted", "setPreempted", "visualization_marker_array", "run_goal"):
        assert needle in src


def test_haf_attributes_span_158_dimensions(data_dir):
    """What the low-rank form of the screening pass (kernels.h: kLrK) rests on: the 302 HAF rows of the reference's Features.txt are
    linear functionals of the 225 corners of the 15 x 15 integral-image window (fv.cpp:155-164: sums of w * (A - B - C + D)) and span
    158 dimensions; with the 21 SHAF rows (not linear: fv.cpp:187-191) that is 179 <= 192 operand slots."""
    rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(data_dir, "Features.txt"))]
    rows = [q for q in rows if len(q) >= 20]
    assert len(rows) == 323
    A = np.zeros((302, 225))
    for a, q in enumerate(rows[:302]):
        reg = [int(x) for x in q[:16]]
        w = [float(np.float32(float(x))) for x in q[16:19]] + [0.0]          # the 4th weight is never assigned (CHaarFeature.cpp:56-60)
        for k in range(4):
            x1, x2, y1, y2 = reg[4 * k:4 * k + 4]
            if w[k] == 0 or x2 < x1 or y2 < y1 or (x2 == 0 and y2 == 0):
                continue
            for (rr, cc, sg) in ((x2 + 1, y2 + 1, 1), (x1, y2 + 1, -1), (x2 + 1, y1, -1), (x1, y1, 1)):
                A[a, rr * 15 + cc] += sg * w[k]
    sv = np.linalg.svd(A, compute_uv=False)
    assert int((sv > 1e-9 * sv[0]).sum()) == 158
    assert sv[157] / sv[0] > 1e-4 and sv[158] / sv[0] < 1e-12
