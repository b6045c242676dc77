"""Pins the CPU oracle (oracle/haf_oracle.c) against
  (a) golden vectors produced by the REAL reference libsvm-3.12 tools (tests/golden/g23_*, g5_*), and
  (b) the real tools themselves (oracle/_ref/svm-scale, svm-predict) run live when they are present.
CPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

import pcdio
from oracle import oracle as O


@pytest.fixture(scope="module")
def orc(data_dir, golden_dir):
    return O.Oracle(os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures"),
                    os.path.join(golden_dir, "surrogate.model"))


def test_feature_table_quirks(orc):
    # fv.cpp:58-82: 323 rows + one phantom all-zero row from the trailing empty line
    assert orc.n_features == 324
    reg, w = orc.feature_table()
    assert (reg[323] == 0).all() and (w[323] == 0).all()
    # CHaarFeature.cpp:56-60: 4th weight never assigned, although rows 299-302 carry one in the file
    assert (w[:, 3] == 0).all()
    assert tuple(w[298, :3]) == (1.0, -1.0, -10.0)
    assert tuple(reg[299, 12:16]) == (5, 8, 6, 7)
    assert reg.min() >= 0 and reg.max() <= 13


def test_range_table(orc):
    lo, up, fmin, fmax, present = orc.range_table()
    assert (lo, up) == (-1.0, 1.0)
    assert present[1:324].all() and not present[0] and len(present) == 324
    assert fmin[1] == -2.38319 and fmax[1] == 2.38931
    assert (fmin[303:324] == 0).all()          # SHAF rows
    assert (fmin[1:] != fmax[1:]).all()


def test_model_parse(orc, golden_dir):
    m = orc.model_arrays()
    with open(os.path.join(golden_dir, "surrogate.model")) as f:
        head = dict(line.split(" ", 1) for line in [next(f) for _ in range(8)])
    assert m["l"] == int(head["total_sv"]) == sum(m["nSV"])
    assert m["gamma"] == float(head["gamma"]) and m["rho"] == float(head["rho"])
    assert m["label"] == tuple(int(t) for t in head["label"].split())
    assert m["D"] <= 323
    # class-0 coefficients positive, class-1 negative (y_i * alpha_i)
    assert (m["coef"][:m["nSV"][0]] > 0).all() and (m["coef"][m["nSV"][0]:] < 0).all()


@pytest.mark.parametrize("name", ["g23_pcd2_r0", "g23_pcd2_r5", "g23_pcd3_r2", "g23_plastic_mug2_r7"])
def test_scale_and_predict_match_reference_tools(orc, golden_dir, name):
    """G2/G3: oracle's %.4g quantisation, svm-scale restatement and RBF decision vs the real tools' outputs."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    feats, q4, scaled, labels, dec = g["features"], g["q4"], g["scaled"], g["labels"], g["dec"]
    m = orc.model_arrays()
    D = scaled.shape[1]
    assert D >= m["D"]
    skip = np.zeros(325, np.uint8)
    skip[324] = 1       # phantom attribute: constant -1 in every row -> dropped by svm-scale.c:336-337
    for r in range(len(feats)):
        mine_q4 = np.array([O.q4(v) for v in feats[r]])
        assert (mine_q4 == q4[r]).all()                          # bit-exact: what sscanf("%lf") read
        xs = orc.scale_row(mine_q4, D, skip)
        assert (xs == scaled[r]).all()                           # bit-exact: what strtod read in svm-predict
        d = orc.decision(xs[:m["D"]])
        assert d == dec[r] or abs(d - dec[r]) <= 1e-15 * max(1.0, abs(dec[r]))
        lab = m["label"][0] if d > 0 else m["label"][1]
        assert lab == labels[r]


def test_heart_scale_known_answer(golden_dir):
    """G5: libsvm KAT: 190 SVs, 259/270 correct, decision values from the reference library."""
    g = np.load(os.path.join(golden_dir, "g5_heart.npz"))
    L = O.lib()
    m = L.hafo_model_load(os.path.join(golden_dir, "heart_scale.model").encode())
    assert m and m.contents.l == 190 and (m.contents.nSV[0], m.contents.nSV[1]) == (90, 100)
    assert abs(m.contents.rho - (-0.00327282)) < 1e-12
    X = np.zeros((270, m.contents.D))
    X[:, :13] = g["X"][:, :m.contents.D]
    dec = np.empty(270)
    L.hafo_decision_rows(m, X.ctypes.data, 270, dec.ctypes.data)
    np.testing.assert_allclose(dec, g["dec"], rtol=0, atol=1e-14)
    lab = np.where(dec > 0, m.contents.label[0], m.contents.label[1])
    assert (lab == g["labels"]).all()
    assert int((lab == g["y"]).sum()) == 259


def test_q4_q6_text_roundtrip_examples():
    assert O.q4(0.123456) == 0.1235
    assert O.q4(123.25) == 123.2          # exact tie -> round-half-even on the exact binary value
    assert O.q4(123.75) == 123.8
    assert O.q4(1234.5) == 1234.0
    assert O.q4(-9999.5) == -10000.0
    assert O.q4(1.0000001e-7) == 1e-7
    assert O.q6(0.12345649999) == 0.123456
    assert O.q6(-1.0 + 2 ** -53) == -1.0
    assert O.q6(1234567.0) == 1.23457e6


def test_vote_rules():
    cfg = O.make_cfg()
    H = W = 56
    g = np.full((H, W), -1, np.int8)
    ev = np.zeros((H, W), np.float32)
    best = np.zeros(3, np.int32)
    O.lib().hafo_vote(O.C.byref(cfg), g.ctypes.data, ev.ctypes.data, best.ctypes.data)
    assert tuple(best) == (0, 27, 0)                      # empty roll: first longest run of zeros, centred
    g[20:31, 20:36] = 1
    O.lib().hafo_vote(O.C.byref(cfg), g.ctypes.data, ev.ctypes.data, best.ctypes.data)
    assert best[2] == 123 and ev.max() == 123
    # interior of the block where all 29 taps are +1: rows 22..28, cols 24..31 -> run of 8, centre = 31 - 8//2
    assert tuple(best[:2]) == (22, 27)
    g[25, 27] = -1                                        # a negative label scores 0 itself and lowers neighbours
    O.lib().hafo_vote(O.C.byref(cfg), g.ctypes.data, ev.ctypes.data, best.ctypes.data)
    assert ev[25, 27] == 0 and ev[25, 28] == 123 - 2 * 4


def test_end_to_end_goldens(orc, data_dir, golden_dir):
    """G6 regression: the committed end-to-end goldens are what the oracle produces today."""
    import make_fixtures as mf
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        gold = json.load(f)
    for key in ["pcd2/C2", "pcd2/C2best", "pcd7/C4", "pcd6/C4", "pcd2/tilt"]:
        name, cname = key.split("/")
        spec = mf.CONFIGS[cname]
        xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
        r = orc.run(xyz, O.make_cfg(**spec["cfg"]), O.make_input(**spec["inp"]))
        g = gold[key]
        for k in ("eval", "row", "col", "roll_idx", "top", "n_evals", "rolls_done"):
            assert r[k] == g[k], (key, k)
        assert r["roll_best"].tolist() == g["roll_best"]
        np.testing.assert_allclose(r["gp1"], g["gp1"], atol=1e-7)


def test_masked_cell_upper_bounds(orc, data_dir):
    """pnt_in_box geometry bounds 361,321,325,313,325,321,... for a 32x32 area."""
    xyz = np.zeros((56 * 56, 3), np.float32)
    ii, jj = np.meshgrid(np.arange(56), np.arange(56), indexing="ij")
    xyz[:, 0] = (ii.ravel() + 0.5) * 0.01 - 0.28
    xyz[:, 1] = (jj.ravel() + 0.5) * 0.01 - 0.28
    xyz[:, 2] = 0.05
    r = orc.run(xyz, O.make_cfg(), O.make_input())
    assert r["mask"].reshape(12, -1).sum(1).tolist() == [361, 321, 325, 313, 325, 321] * 2
    # every cell non-empty at roll 0 (heights 0.05 + 0.15)
    assert np.allclose(r["heights"][0], 0.2)


@pytest.mark.skipif(not os.path.exists(os.path.join(O.ref_dir(), "svm-scale")), reason="oracle/_ref not built")
def test_live_against_reference_binaries(orc, data_dir, golden_dir, tmp_path):
    """The oracle's feature text fed through the REAL svm-scale + svm-predict gives the oracle's labels (live)."""
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd12.pcd"))
    cfg, inp = O.make_cfg(), O.make_input(length_y=44)
    res = orc.run(xyz, cfg, inp)
    for roll in (1, 8):
        f = str(tmp_path / ("f%d.txt" % roll))
        n = orc.dump_feature_file(xyz, cfg, inp, roll, f)
        assert n == res["mask"][roll].sum()
        with open(f + ".scale", "w") as out:
            subprocess.run([os.path.join(O.ref_dir(), "svm-scale"), "-r",
                            os.path.join(data_dir, "range21062012_allfeatures"), f], stdout=out, check=True)
        subprocess.run([os.path.join(O.ref_dir(), "svm-predict"), f + ".scale",
                        os.path.join(golden_dir, "surrogate.model"), f + ".out"], stdout=subprocess.DEVNULL, check=True)
        labels = np.loadtxt(f + ".out").astype(int).reshape(-1)
        mine = res["labels"][roll][res["mask"][roll] == 1]
        assert (labels == mine).all()
