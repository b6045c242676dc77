"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle on the same inputs.

Bars (north_star / SURVEY.md §8d): height grid, integral image, mask, label grid, per-roll winners, overall
(row, col, roll) and eval BIT-EXACT / identical; grasp points within 1e-4 m.  Decision values: the fast path is an
fp32 contraction, so its error scales with the cancellation in the sum, S = sum_n |coef_n| K_n: the bar is
|dec - dec_oracle| <= 2^-20 * S + 1e-6 (that is <= 1e-4 whenever S <= 100, e.g. the seeded random models; the
surrogate model trained with C = 512 has S up to 7e3), and <= 1e-12 * S where the guard band re-evaluated in fp64.
Labels are exact in every case because |dec| <= 2^-15 * S is always re-evaluated in libsvm's own order."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import models
import pcdio
from haf_grasping_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEC_REL = 2.0 ** -20      # of S = sum |coef| K
DEC_ABS = 1e-6
STATS = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_stats():
    """What the tests measured on the way (refined shares, worst errors) goes to gpurun_out/test_stats.json."""
    yield
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "test_stats.json"), "w") as f:
            json.dump(STATS, f, indent=1, sort_keys=True, default=str)
    except OSError:
        pass


def _files(data_dir):
    return os.path.join(data_dir, "Features.txt"), os.path.join(data_dir, "range21062012_allfeatures")


@pytest.fixture(scope="module")
def surrogate(golden_dir):
    return os.path.join(golden_dir, "surrogate.model")


@pytest.fixture(scope="module")
def orc(data_dir, surrogate):
    f, r = _files(data_dir)
    return O.Oracle(f, r, surrogate)


MODES = [pytest.param(capi.FLAG_FP32_MFMA, id="f32mfma"), pytest.param(capi.FLAG_SPLIT_F16, id="splitf16"),
         pytest.param(0, id="screen")]
# default mode: evaluations the single-pass fp16 screening kernel decides keep its decision value, whose error is bounded
# by the screening band ln2*(|du||v| + |u||dv|)*S < 2^-8 * S for attributes inside the svm-scale range
DEC_REL_SCREEN = 2.0 ** -8


# environment switches of the TESTING build (libhafgrasp_testing.so, -DHAF_TESTING); the product library ignores them
TEST_KNOBS = ("HAF_GUARD_REL", "HAF_GUARD0_REL", "HAF_GUARD2_REL", "HAF_LARGE_EVALS", "HAF_NO_FAST_GROUPS", "HAF_FLAG_WINDOW",
              "HAF_SCREEN_NO_CENTRE", "HAF_NO_DIRECT", "HAF_NO_FUSED_PRE", "HAF_HOST_EXP_ALL",
              "HAF_NO_CALIBRATE", "HAF_NO_I8", "HAF_GUARD_I8_REL", "HAF_SCREEN_VARIANT", "HAF_NO_CR", "HAF_CR_NO_CENTRE", "HAF_KAPPA",
              "HAF_KAPPA_T1_MEASURED", "HAF_NO_CR_T1", "HAF_T0B", "HAF_T1_SKIP", "HAF_PROB_HOST_ALL", "HAF_REPROBE_EVERY", "HAF_SCREEN_PARTS",
              "HAF_NO_LR", "HAF_LR_UNFUSED", "HAF_NO_LR_PLAIN", "HAF_CANARY_CHECK", "HAF_FLAG0_CAP", "HAF_T0B_NO_GATHER", "HAF_NO_SHORT_GATE")


def make_engine(data_dir, model, mode=0, **cfg):
    """The product library, unless a test has set one of the guard-band / kernel-choice switches: those exist in the
    testing build only (same kernels; the engine's host units compiled with -DHAF_TESTING)."""
    f, r = _files(data_dir)
    cfg.setdefault("flags", capi.FLAG_KEEP_DEBUG | capi.FLAG_PROFILE)
    cfg["flags"] |= mode
    cfg.setdefault("testing", any(k in os.environ for k in TEST_KNOBS))
    return capi.Engine(f, r, model, **cfg)


def oracle_input(kw):
    return O.make_input(center=kw.get("grasp_area_center", (0, 0, 0)), length_x=kw.get("grasp_area_length_x", 32),
                        length_y=kw.get("grasp_area_length_y", 44), approach=kw.get("approach_vector", (0, 0, 1)),
                        show_only_best=kw.get("show_only_best_grasp", 0), gripper_width=kw.get("gripper_opening_width", 1))


def compare_full(eng, orc, xyz, cfg_kw, in_kw, check_dec=True):
    ocfg = O.make_cfg(**{k: v for k, v in cfg_kw.items() if k in ("n_rolls", "roll_step_deg")},
                      H=cfg_kw.get("grid_h", 56), W=cfg_kw.get("grid_w", 56))
    want = orc.run(xyz, ocfg, oracle_input(in_kw))
    got = eng.score(xyz, capi.default_input(**in_kw))
    R = ocfg.n_rolls
    # With show_only_best the reference stops early; the engine still computes every roll, compare the executed ones.
    for roll in range(want["rolls_done"]):
        h = eng.debug(capi.DBG_HEIGHTS, 0, roll)
        assert (h.view(np.uint32) == want["heights"][roll].view(np.uint32)).all(), ("heights", roll)
        ii = eng.debug(capi.DBG_INTEGRAL, 0, roll)
        assert (ii.view(np.uint32) == want["integral"][roll].view(np.uint32)).all(), ("integral", roll)
        m = eng.debug(capi.DBG_MASK, 0, roll)
        assert (m == want["mask"][roll]).all(), ("mask", roll)
        lab = eng.debug(capi.DBG_LABELS, 0, roll)
        assert (lab == want["labels"][roll]).all(), ("labels", roll, int((lab != want["labels"][roll]).sum()))
        if check_dec:
            d = eng.debug(capi.DBG_DECISION, 0, roll)
            msk = want["mask"][roll] == 1
            assert np.isnan(d[~msk]).all()
            if msk.any():
                err = np.abs(d[msk] - want["dec"][roll][msk])
                screened = not (eng.cfg.flags & (capi.FLAG_FP32_MFMA | capi.FLAG_SPLIT_F16))
                bound = (DEC_REL_SCREEN if screened else DEC_REL) * want["sabs"][roll][msk] + DEC_ABS
                assert (err <= bound).all(), ("decision", roll, float((err / bound).max()))
                STATS["max_rel_err"] = max(STATS.get("max_rel_err", 0.0), float((err / want["sabs"][roll][msk]).max()))
        ev, _ = eng.roll_grid(0, roll)
        assert (ev == want["graspseval"][roll]).all(), ("vote grid", roll)
    for k_e, k_o in [("eval", "eval"), ("best_row", "row"), ("best_col", "col"), ("best_roll", "roll_idx"),
                     ("best_vote", "top"), ("rolls_done", "rolls_done"), ("n_evals", "n_evals")]:
        assert got[k_e] == want[k_o], (k_e, got[k_e], want[k_o])
    np.testing.assert_allclose(got["grasp_point1"], want["gp1"], atol=1e-4)
    np.testing.assert_allclose(got["grasp_point2"], want["gp2"], atol=1e-4)
    np.testing.assert_allclose(got["averaged_grasp_point"], want["avg"], atol=1e-4)
    np.testing.assert_allclose(got["approach_vector"], want["av"], atol=1e-6)
    assert got["roll"] == want["roll"]
    assert R >= want["rolls_done"]
    return got, want


def test_device_decimal_roundtrip_matches_host():
    """csrc/decq.h compiled for gfx950 gives the same bits as its host build (which test_host_cpu pins to glibc)."""
    L = capi.testlib()
    rng = np.random.RandomState(11)
    for digits in (4, 6):
        parts = [rng.standard_normal(20000) * s for s in (1e-9, 1e-4, 1.0, 50.0, 1e4, 1e9, 1e-18, 1e24, 1e-30, 1e35)]
        base = rng.randint(10 ** (digits - 1), 10 ** digits, size=4000).astype(np.float64)
        for e in range(-6, 8):
            t = (base + 0.5) * 10.0 ** e
            parts += [t, np.nextafter(t, np.inf), np.nextafter(t, -np.inf)]
        x = np.concatenate(parts + [np.array([0.0, -0.0, np.inf, -np.inf, 1e-310, 123.25, 9999.5])])
        if digits == 4:
            x = x.astype(np.float32).astype(np.float64)
        out = np.empty_like(x)
        assert L.haf_test_decq_device(x.ctypes.data, out.ctypes.data, len(x), digits) == 0
        host = np.array([L.haf_test_decq_host(float(v), digits) for v in x])
        assert (out.view(np.uint64) == host.view(np.uint64)).all()
        if digits == 4:     # the fp32 entry the feature kernel calls
            out40 = np.empty_like(x)
            assert L.haf_test_decq_device(x.ctypes.data, out40.ctypes.data, len(x), 40) == 0
            assert (out40.view(np.uint64) == host.view(np.uint64)).all()


def test_device_scale_matches_host(data_dir):
    L = capi.testlib()
    f, r = _files(data_dir)
    o = O.Oracle(f, r, None)
    lo, up, fmin, fmax, _ = o.range_table()
    rng = np.random.RandomState(5)
    n = 323 * 300
    idx = np.tile(np.arange(1, 324), 300)
    v = (rng.standard_normal(n) * rng.choice([0.01, 1.0, 8.0], size=n)).astype(np.float32)
    q4 = np.array([L.haf_test_decq_host(float(x), 4) for x in v])
    q4[::97] = fmin[idx[::97]]
    q4[5::101] = fmax[idx[5::101]]
    out = np.empty(n)
    a, b = np.ascontiguousarray(fmin[idx]), np.ascontiguousarray(fmax[idx])
    assert L.haf_test_scale_device(q4.ctypes.data, a.ctypes.data, b.ctypes.data, lo, up, out.ctypes.data, n) == 0
    host = np.array([L.haf_test_scale_host(q4[i], a[i], b[i], lo, up) for i in range(n)])
    assert (out.view(np.uint64) == host.view(np.uint64)).all()


@pytest.mark.parametrize("mode", MODES)
def test_c1_c2_pcd2_stage_by_stage(data_dir, surrogate, orc, mode):
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    eng = make_engine(data_dir, surrogate, mode, n_rolls=1)
    compare_full(eng, orc, xyz, dict(n_rolls=1), dict(grasp_area_length_x=32, grasp_area_length_y=32))   # C1
    eng.close()
    eng = make_engine(data_dir, surrogate, mode)
    got, _ = compare_full(eng, orc, xyz, dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32))   # C2
    assert got["eval"] == 103
    compare_full(eng, orc, xyz, dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32, show_only_best_grasp=1))
    compare_full(eng, orc, xyz, dict(n_rolls=12), dict(approach_vector=(0.2, -0.1, 1.0)))
    compare_full(eng, orc, xyz, dict(n_rolls=12), dict(grasp_area_center=(0.01, -0.02, 0.005), gripper_opening_width=2))
    eng.close()


@pytest.mark.parametrize("knobs", [(), ("HAF_NO_DIRECT",), ("HAF_NO_FUSED_PRE",), ("HAF_NO_DIRECT", "HAF_NO_FUSED_PRE")])
def test_small_request_paths_agree_with_the_oracle(data_dir, surrogate, orc, monkeypatch, knobs):
    """Round 3: a request on a grid that fits LDS runs its pre-stages in ONE launch (k_small_pre), and a request whose whole SVM
    work is tiny goes straight to the fp64 MFMA tier (a somewhat larger one sends what its three-pass kernel flags through the same
    kernel in list mode).  Both shortcuts, either one, and neither (the separate kernels, the fast
    tiers) must give the oracle's grids, labels and grasp: C1, C2, a tilted approach vector, the client's default area, a large
    cloud on the small grid (k_bin_lds feeds the fused kernel), an empty and a one-point cloud."""
    for k in knobs:
        monkeypatch.setenv(k, "1")
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    eng = make_engine(data_dir, surrogate, testing=True)
    got, _ = compare_full(eng, orc, xyz, dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32))   # C2
    assert got["eval"] == 103
    cnt = eng.last_counts()
    if "HAF_NO_DIRECT" in knobs:
        assert cnt["n_rechecked"] < cnt["n_evals"]
    else:
        assert cnt["n_rechecked"] == cnt["n_evals"] == got["n_evals"] and cnt["n_refined"] == 0     # every evaluation through the fp64 tier
    compare_full(eng, orc, xyz, dict(n_rolls=12), dict(approach_vector=(0.2, -0.1, 1.0), show_only_best_grasp=1))
    compare_full(eng, orc, xyz, dict(n_rolls=12), dict())                                       # the client's default 32 x 44 area
    big = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    compare_full(eng, orc, big, dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0)))
    # the exact stage of such a request (too large for the direct path, small enough that launches count): what the three-pass
    # kernel flags goes through ONE launch of the direct kernel in list mode -- or, with that switched off, tier 2a and the fp64 tier
    cnt, ex = eng.last_counts(), eng.last_exact_tiers()
    assert 0 < cnt["n_rechecked"] < cnt["n_evals"]
    if "HAF_NO_DIRECT" in knobs:
        assert ex["n_integer"] == cnt["n_rechecked"] and ex["n_fp64"] < ex["n_integer"]
    else:
        assert ex["n_integer"] == 0 and ex["n_fp64"] == cnt["n_rechecked"]
    compare_full(eng, orc, np.zeros((0, 3), np.float32), dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32))
    compare_full(eng, orc, np.array([[0.0, 0.0, 0.05]], np.float32), dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32))
    eng.close()


def test_trained_model_against_committed_goldens(data_dir, golden_dir, trained_model):
    """Round 4: the 8964-SV trained model (tests/golden/trained.model.npz) on six cloud x configuration cases of
    tests/golden/g6_trained.json (oracle results): argmax-identical cell / roll, same eval, per-roll winners, in the default mode."""
    _against_goldens(data_dir, golden_dir, trained_model, 0, "g6_trained.json")


@pytest.mark.parametrize("mode", MODES)
def test_all_clouds_against_committed_goldens(data_dir, golden_dir, surrogate, mode):
    """Every data/*.pcd x configuration of tests/golden/g6_end_to_end.json: argmax-identical cell/roll, same eval."""
    _against_goldens(data_dir, golden_dir, surrogate, mode, "g6_end_to_end.json")


def _against_goldens(data_dir, golden_dir, surrogate, mode, gold_name):
    import make_fixtures as mf
    with open(os.path.join(golden_dir, gold_name)) as f:
        gold = json.load(f)
    engines = {}
    for key, g in sorted(gold.items()):
        name, cname = key.split("/")
        spec = mf.CONFIGS[cname]
        ck = (spec["cfg"].get("n_rolls", 12), spec["cfg"].get("roll_step_deg", 15))
        if ck not in engines:
            engines[ck] = make_engine(data_dir, surrogate, mode, n_rolls=ck[0], roll_step_deg=ck[1], max_points=1 << 18)
        eng = engines[ck]
        xyz = capi.load_pcd(os.path.join(data_dir, name + ".pcd"))
        i = spec["inp"]
        kw = dict(grasp_area_length_x=i.get("length_x", 32), grasp_area_length_y=i.get("length_y", 44))
        if "center" in i:
            kw["grasp_area_center"] = i["center"]
        if "approach" in i:
            kw["approach_vector"] = i["approach"]
        if "show_only_best" in i:
            kw["show_only_best_grasp"] = i["show_only_best"]
        got = eng.score(xyz, capi.default_input(**kw))
        assert (got["eval"], got["best_row"], got["best_col"], got["best_roll"], got["best_vote"]) == \
               (g["eval"], g["row"], g["col"], g["roll_idx"], g["top"]), key
        assert got["n_evals"] == g["n_evals"] and got["rolls_done"] == g["rolls_done"], key
        np.testing.assert_allclose(got["grasp_point1"], g["gp1"], atol=1e-4, err_msg=key)
        np.testing.assert_allclose(got["grasp_point2"], g["gp2"], atol=1e-4, err_msg=key)
        np.testing.assert_allclose(got["approach_vector"], g["av"], atol=1e-6, err_msg=key)
        rec = eng.score_rolls([xyz], [capi.default_input(**kw)], 0, ck[0])[0]
        for roll in range(g["rolls_done"]):
            assert [int(rec["row"][roll]), int(rec["col"][roll]), int(rec["vote"][roll])] == g["roll_best"][roll], (key, roll)
            assert int(rec["n_evals"][roll]) == g["masked"][roll], (key, roll)
    for e in engines.values():
        e.close()


@pytest.mark.parametrize("mode", MODES)
def test_table_cloud_live_against_oracle(data_dir, surrogate, orc, mode):
    """C3: 102 876-point binary_compressed cloud, 56x56 area, 20 rolls of 9 degrees, full stage comparison."""
    xyz = capi.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    eng = make_engine(data_dir, surrogate, mode, n_rolls=20, roll_step_deg=9, max_points=1 << 18)
    compare_full(eng, orc, xyz, dict(n_rolls=20, roll_step_deg=9),
                 dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0)))
    eng.close()


@pytest.mark.parametrize("mode", [0, capi.FLAG_SPLIT_F16])
def test_identical_calls_give_identical_decision_values(data_dir, surrogate, mode):
    """The same request six times through one engine (C3: a model the three-pass kernel serves, 6 short SV tiles -- the regime where
    the LDS-DMA ring of that kernel really waits for its data): every decision value, label and tier counter identical from call
    to call.  A timing experiment of round 3 (a build whose ONLY difference was the code of other kernels) showed ~4 000 decision
    values of this request moving by up to 1e-5 S between calls -- inside the band, so no label changed and no parity test saw
    it; the ring kernels now wait for all of their DMA pieces, and this test makes the next build like that one fail."""
    xyz = capi.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    eng = make_engine(data_dir, surrogate, mode, n_rolls=20, roll_step_deg=9, max_points=1 << 18)
    inp = capi.default_input(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0))
    ref = None
    for call in range(6):
        eng.score(xyz, inp)
        dec = np.stack([eng.debug(capi.DBG_DECISION, 0, r) for r in range(20)])
        lab = np.stack([eng.debug(capi.DBG_LABELS, 0, r) for r in range(20)])
        cur = (np.nan_to_num(dec, nan=-1e300), lab, eng.last_counts(), eng.last_exact_tiers())
        if ref is None:
            ref = cur
            continue
        assert int((cur[0] != ref[0]).sum()) == 0, (call, int((cur[0] != ref[0]).sum()), float(np.abs(cur[0] - ref[0]).max()))
        assert (cur[1] == ref[1]).all() and cur[2] == ref[2] and cur[3] == ref[3], (call, cur[2], ref[2])
    eng.close()


def test_identical_calls_give_identical_device_lists(data_dir, surrogate, tmp_path, monkeypatch):
    """Round 5.  The evaluation list of a small request and the hand-over lists of the exact tiers were filled by
    arrival (atomicAdd): labels never depended on the order, but the contents of list windows and every debug list changed from run to
    run.  Now (cloud, roll) r's evaluations follow (cloud, roll) r - 1's (k_small_pre: counts published per workgroup, summed by the ones
    behind) and an undecided entry keeps its place in the next tier's list (one ballot word per 64 entries + a one-workgroup compaction):
    the same request five times -- C3 through the fused pre-stage, and a 96 x 96 request against a 517-SV model through the screening
    pass, the exact-integer tier and the fp64 tier with 64-entry windows -- must leave the SAME lists in device memory, entry for entry,
    and the evaluation list must be the row-major masked cells of roll 0, then roll 1, ..."""
    xyz = capi.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    cases = [(surrogate, dict(n_rolls=20, roll_step_deg=9, max_points=1 << 18), dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0)), xyz, {})]
    path = models.write_random_model(str(tmp_path / "r517.model"), 517, seed=77, gamma=1.0 / 323, balanced=True, rho=0.01)
    G = 96
    cases.append((path, dict(n_rolls=5, roll_step_deg=36, grid_h=G, grid_w=G, max_points=2 * G * G),
                  dict(grasp_area_length_x=G, grasp_area_length_y=G), models.synthetic_cloud(grid=G, k=2, seed=4),
                  dict(HAF_FLAG_WINDOW="64", HAF_GUARD0_REL="1000", HAF_GUARD_REL="300", HAF_GUARD_I8_REL="2000")))
    seen = {}
    for ci, (model, cfg, inp, cloud, env) in enumerate(cases):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            eng = make_engine(data_dir, model, testing=True, **cfg)
            ref = None
            for call in range(5):
                eng.score(cloud, capi.default_input(**inp))
                cur = [eng.fetch_list(w) for w in range(5)]
                if ref is None:
                    ref = cur
                    # the evaluation list: per roll the masked cells in row-major order (calc_featurevectors, server.cpp:637-643), roll after roll
                    H, W = cfg.get("grid_h", 56), cfg.get("grid_w", 56)
                    want = np.concatenate([np.flatnonzero(eng.debug(capi.DBG_MASK, 0, r).reshape(-1)) + r * H * W for r in range(cfg["n_rolls"])])
                    if ci == 0:
                        assert (np.sort(cur[0]) == np.sort(want)).all() and (cur[0] == want).all(), "evaluation list is not roll after roll, row-major"
                    continue
                for w in range(5):
                    assert cur[w].shape == ref[w].shape and (cur[w] == ref[w]).all(), (ci, call, w, cur[w].shape, ref[w].shape)
            seen[ci] = [int(x.size) for x in ref]
            eng.close()
    assert seen[0][0] > 30000 and seen[0][1] > 0 and seen[1][1] > 64 and seen[1][2] > 0, seen        # the lists in question were not empty
    STATS["ordered_lists"] = seen


def test_batch_of_eight_equals_single_and_shards_compose(data_dir, surrogate):
    """C4: pcd1..8 in one batch call == eight single calls; roll shards + haf_finalize == the unsharded call."""
    names = ["pcd%d" % i for i in range(1, 9)]
    clouds = [capi.load_pcd(os.path.join(data_dir, n + ".pcd")) for n in names]
    inputs = [capi.default_input(grasp_area_length_x=32, grasp_area_length_y=44) for _ in names]
    inputs[6] = capi.default_input(grasp_area_center=(0.30, 0.46, 0.0))       # pcd7: centred on the object
    eng = make_engine(data_dir, surrogate, n_rolls=20, roll_step_deg=9, max_clouds=8)
    batch = eng.score_batch(clouds, inputs)
    singles = [eng.score(c, i) for c, i in zip(clouds, inputs)]
    for b, s in zip(batch, singles):
        for k in ("eval", "best_row", "best_col", "best_roll", "best_vote", "n_evals", "grasp_point1", "roll"):
            assert b[k] == s[k], k
    assert batch[7]["eval"] == -20 and (batch[7]["best_row"], batch[7]["best_col"]) == (0, 27)   # pcd8 outside the grid
    # roll shards of different sizes, gathered like ranks would
    full = eng.score_rolls(clouds[:3], inputs[:3], 0, 20)
    parts = [eng.score_rolls(clouds[:3], inputs[:3], a, n) for a, n in ((0, 5), (5, 5), (10, 7), (17, 3))]
    gathered = np.concatenate(parts, axis=1)
    assert (gathered == full).all()
    for c in range(3):
        assert eng.finalize(inputs[c], gathered[c])["eval"] == batch[c]["eval"]
    eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_random_models_label_order_and_guard_band(data_dir, tmp_path, mode):
    """Seeded random libsvm models ('label 1 -1', balanced coefficients -> decision values crowd around zero):
    stresses the guard band; labels must still be identical to the fp64 oracle."""
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    f, r = _files(data_dir)
    for nsv, seed in ((96, 1), (300, 2)):
        path = str(tmp_path / ("rand%d.model" % nsv))
        models.write_random_model(path, nsv, seed=seed, balanced=True)
        o = O.Oracle(f, r, path)
        eng = make_engine(data_dir, path, mode)
        got, want = compare_full(eng, o, xyz, dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=44))
        assert (want["labels"] > 0).sum() > 50 and ((want["labels"] == -1) & (want["mask"] == 1)).sum() > 50
        eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_generalised_grid_and_rolls(data_dir, surrogate, orc, mode):
    """H, W, roll step and roll count are parameters here.  96x96 grid, 8 rolls of 22 degrees."""
    xyz = models.synthetic_cloud(grid=96, k=2, seed=3)
    eng = make_engine(data_dir, surrogate, mode, grid_h=96, grid_w=96, n_rolls=8, roll_step_deg=22)
    compare_full(eng, orc, xyz, dict(n_rolls=8, roll_step_deg=22, grid_h=96, grid_w=96),
                 dict(grasp_area_length_x=96, grasp_area_length_y=80))
    eng.close()
    # odd grid size, roll step that does not divide 180
    xyz = models.synthetic_cloud(grid=61, k=3, seed=5)
    eng = make_engine(data_dir, surrogate, mode, grid_h=61, grid_w=61, n_rolls=3, roll_step_deg=37)
    compare_full(eng, orc, xyz, dict(n_rolls=3, roll_step_deg=37, grid_h=61, grid_w=61),
                 dict(grasp_area_length_x=61, grasp_area_length_y=45, grasp_area_center=(0.004, -0.003, 0.0)))
    eng.close()


def test_screening_feature_paths_agree_with_the_oracle(data_dir, surrogate, orc, monkeypatch):
    """The thread-per-evaluation screening feature kernel (large requests; forced here through HAF_LARGE_EVALS) has three routes
    to an attribute: waves of 64 neighbouring cells read an LDS band of the
    integral image, two-region HAF groups through screen_quad and every other group through screen_pair3; all other waves
    address per lane.  A 96-wide area gives rows of 82 cells (one whole 64-chunk + a left-over each, so both wave kinds
    occur); HAF_NO_FAST_GROUPS sends every group through screen_pair3.  Stage by stage against the oracle both times."""
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")        # the thread-per-evaluation kernel whatever the request size
    xyz = models.synthetic_cloud(grid=96, k=2, seed=11)
    for env in (None, "1"):
        if env: monkeypatch.setenv("HAF_NO_FAST_GROUPS", env)
        eng = make_engine(data_dir, surrogate, 0, grid_h=96, grid_w=96, n_rolls=4, roll_step_deg=45)
        compare_full(eng, orc, xyz, dict(n_rolls=4, roll_step_deg=45, grid_h=96, grid_w=96),
                     dict(grasp_area_length_x=96, grasp_area_length_y=96))
        c = eng.last_counts()
        assert c["n_evals"] > 4 * 40 * 40
        eng.close()
    monkeypatch.delenv("HAF_NO_FAST_GROUPS")
    # rows with holes: 13x13-cell patches without points leave 5x5 cells outside the mask in the middle of long rows, so some
    # 64-chunks of region A are not 64 neighbouring cells and take the per-lane route next to band waves of the same workgroup
    grid = 160
    xyz = models.synthetic_cloud(grid=grid, k=2, seed=12)
    cx = np.floor((xyz[:, 0] + grid * 0.005) / 0.01).astype(int)
    cy = np.floor((xyz[:, 1] + grid * 0.005) / 0.01).astype(int)
    keep = np.ones(len(xyz), bool)
    rng = np.random.RandomState(5)
    for _ in range(7):
        a, b = rng.randint(20, grid - 33, size=2)
        keep &= ~((cx >= a) & (cx < a + 13) & (cy >= b) & (cy < b + 13))
    xyz = np.ascontiguousarray(xyz[keep])
    eng = make_engine(data_dir, surrogate, 0, grid_h=grid, grid_w=grid, n_rolls=2, roll_step_deg=90)
    compare_full(eng, orc, xyz, dict(n_rolls=2, roll_step_deg=90, grid_h=grid, grid_w=grid),
                 dict(grasp_area_length_x=grid, grasp_area_length_y=grid))
    m0 = eng.debug(capi.DBG_MASK, 0, 0)
    inner = m0[7:grid - 7, 7:grid - 7]
    assert inner.sum() < inner.size and (inner.sum(axis=1) >= 64).all()      # holes, and every row still has a whole chunk
    eng.close()


def test_edge_inputs(data_dir, surrogate, orc):
    eng = make_engine(data_dir, surrogate)
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=32)
    # empty cloud: nothing masked, reference answer (row 0, col 27, roll 0, eval -20)
    got = eng.score(np.zeros((0, 3), np.float32), capi.default_input(**inp))
    assert (got["eval"], got["best_row"], got["best_col"], got["best_roll"], got["n_evals"]) == (-20, 0, 27, 0, 0)
    # NaN / inf coordinates never reach a cell; PCL-style stride 4
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    bad = np.concatenate([xyz, np.array([[np.nan, 0, 1], [0, np.nan, 1], [0.01, 0.01, np.nan], [np.inf, 0, 1]], np.float32)])
    padded = np.zeros((bad.shape[0], 4), np.float32)
    padded[:, :3] = bad
    compare_full(eng, orc, padded, dict(n_rolls=12), inp)
    # points exactly on cell edges and on the +-0.28 border
    edge = np.array([[-0.28, 0, 0.1], [0.28, 0, 0.1], [0.27999997, 0.27999997, 0.2], [-0.27999997, -0.27999997, 0.2],
                     [0.0, 0.0, 0.05], [0.01, 0.01, 0.05], [-0.01, -0.01, -0.3], [0.05, 0.05, -1.2]], np.float32)
    compare_full(eng, orc, np.concatenate([xyz, edge]), dict(n_rolls=12), inp)
    # search area smaller than the border: no cell can be masked
    got = eng.score(xyz, capi.default_input(grasp_area_length_x=10, grasp_area_length_y=10))
    assert got["n_evals"] == 0 and got["eval"] == -20
    # capacity and argument errors are loud
    with pytest.raises(capi.HafError) as ei:
        eng.score_batch([xyz, xyz], [capi.default_input(), capi.default_input()])
    assert ei.value.code == capi.HAF_E_CAPACITY
    # a negative budget stops the reference's loop before roll 0 and the goal still succeeds with the untouched overall best
    # (server.cpp:322-326, 367-374, 390): eval -1000 - 20, roll -1, no roll executed
    got = eng.score(xyz, capi.default_input(max_calculation_time=-1.0))
    assert (got["eval"], got["best_roll"], got["best_row"], got["best_col"], got["rolls_done"]) == (-1020, -1, -1, -1, 0)
    assert eng.score(xyz, capi.default_input(max_calculation_time=-0.5))["rolls_done"] > 0      # (int)(-0.5) == 0: not negative
    eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_full_size_c5_properties(data_dir, tmp_path, mode):
    """BASELINE config C5 at full size (512x512, 36 rolls of 5 degrees, 524 288 points): size-independent properties.
    The oracle cannot run 7.9M evaluations, so: (1) the eval count equals the pure-geometry count (all cells are
    non-empty), (2) spot-check 300 random masked cells per sampled roll against the oracle's own feature/scale/decision
    chain fed with the engine's integral image, (3) rolls r and r+18 (90 degrees apart, square area) mask the same number of cells,
    (4) the same cloud scored twice gives identical records (determinism), (5) sharded == unsharded."""
    nsv = 256
    path = str(tmp_path / "rand256.model")
    models.write_random_model(path, nsv, seed=4, balanced=True)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, path)
    xyz = models.synthetic_cloud(grid=512, k=2, seed=0)
    assert xyz.shape == (524288, 3)
    eng = make_engine(data_dir, path, mode, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20)
    inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
    rec = eng.score_rolls([xyz], [inp], 0, 36)[0]
    # 7 883 478 is the pure-geometry bound (every 9x9 neighbourhood non-empty); 248 004 at 0 degrees
    assert int(rec["n_evals"][0]) == 248004 and 7800000 < int(rec["n_evals"].sum()) <= 7883478
    assert (rec["n_evals"][1:18] == rec["n_evals"][19:]).all()       # square area: rolls r and r+18 are 90 degrees apart
    # stages a1-a3 for EVERY roll against the oracle's stage functions (cheap on the CPU even at this size)
    ocfg = O.make_cfg(H=512, W=512, n_rolls=36, roll_step_deg=5)
    oin = O.make_input(length_x=512, length_y=512)
    M = np.zeros(16, np.float32)
    oh = np.zeros((512, 512), np.float32)
    oii = np.zeros((513, 513), np.float32)
    om = np.zeros((512, 512), np.uint8)
    for roll in range(36):
        O.lib().hafo_transform(C.byref(ocfg), C.byref(oin), roll, 0, M.ctypes.data)
        O.lib().hafo_height_grid(C.byref(ocfg), xyz.ctypes.data, xyz.shape[0], 3, M.ctypes.data, oh.ctypes.data)
        O.lib().hafo_integral(C.byref(ocfg), oh.ctypes.data, oii.ctypes.data)
        O.lib().hafo_mask(C.byref(ocfg), C.byref(oin), roll, oii.ctypes.data, om.ctypes.data)
        assert (eng.debug(capi.DBG_HEIGHTS, 0, roll).view(np.uint32) == oh.view(np.uint32)).all(), roll
        assert (eng.debug(capi.DBG_INTEGRAL, 0, roll).view(np.uint32) == oii.view(np.uint32)).all(), roll
        assert (eng.debug(capi.DBG_MASK, 0, roll) == om).all(), roll
        assert int(om.sum()) == int(rec["n_evals"][roll])
    rng = np.random.RandomState(9)
    m = o.model_arrays()
    lo, up, fmin, fmax, _ = o.range_table()
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    for roll in (0, 7, 23):
        ii = eng.debug(capi.DBG_INTEGRAL, 0, roll)
        lab = eng.debug(capi.DBG_LABELS, 0, roll)
        dec = eng.debug(capi.DBG_DECISION, 0, roll)
        msk = eng.debug(capi.DBG_MASK, 0, roll)
        cells = np.argwhere(msk == 1)
        assert len(cells) == rec["n_evals"][roll]
        for i, j in cells[rng.choice(len(cells), 300, replace=False)]:
            feats = o.feature_values(ii[i - 7:i + 8, j - 7:j + 8])
            xs = o.scale_row(np.array([O.q4(v) for v in feats]), m["D"], skip)
            d = o.decision(xs)
            want = m["label"][0] if d > 0 else m["label"][1]
            assert lab[i, j] == want, (roll, i, j, d, dec[i, j])
            assert abs(dec[i, j] - d) <= (0.02 if mode == 0 else 1e-4)     # screening pass: 2^-8 * S, S <= ~5 for this model
    rec2 = eng.score_rolls([xyz], [inp], 0, 36)[0]
    assert (rec == rec2).all()
    parts = np.concatenate([eng.score_rolls([xyz], [inp], a, 9)[0] for a in (0, 9, 18, 27)])
    assert (parts == rec).all()
    out = eng.finalize(inp, rec)
    assert out["best_vote"] == rec["vote"].max() and out["best_roll"] == int(np.argmax(rec["vote"]))
    eng.close()


def test_cli_and_server_mirror(data_dir, golden_dir, surrogate):
    """The C++ command-line front end (client parameter surface, client.cpp:79-118) and the Python mirror of the action
    interface give the golden answer for the README demo cloud."""
    import subprocess
    from haf_grasping_amd import CalcGraspPointsServer, GraspInputMsg
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        g = json.load(f)["pcd2/C2"]
    f_, r_ = _files(data_dir)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "haf_grasp_cli")
    out = subprocess.run([cli, "--features", f_, "--range", r_, "--model", surrogate, "--search-size", "18", "18",
                          os.path.join(data_dir, "pcd2.pcd")], check=True, capture_output=True, text=True)
    tok = out.stdout.split()
    assert int(tok[0]) == g["eval"] and int(tok[13]) == g["roll_idx"] * 15          # 18 + 14 = 32 cm search area
    np.testing.assert_allclose([float(t) for t in tok[1:4]], g["gp1"], atol=1e-4)
    srv = CalcGraspPointsServer(f_, r_, surrogate)
    res = srv.execute(GraspInputMsg(input_pc=capi.load_pcd(os.path.join(data_dir, "pcd2.pcd")),
                                    grasp_area_length_x=32, grasp_area_length_y=32))
    assert res.eval == g["eval"] and res.frame_id == "/base_link"
    np.testing.assert_allclose(res.graspPoint2, g["gp2"], atol=1e-4)
    assert res.hypothesis_string().split()[0] == str(g["eval"])
    srv.close()
    # ---- ros_shim/shim_core.h through the CLI: the per-roll hypotheses of show_predicted_gps (server.cpp:962-969) and the
    # exact text of /haf_grasping/grasp_hypothesis_with_eval (1384), against the oracle-derived goldens ----
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        gold = json.load(f)
    for name, key, extra in (("plastic_mug2", "plastic_mug2/default", []), ("pcd2", "pcd2/default", []),
                             ("plastic_mug2", "plastic_mug2/C2best", ["--search-size", "18", "18", "--show-only-best"])):
        w = gold[key]
        args = [cli, "--features", f_, "--range", r_, "--model", surrogate, "--hypotheses"] + (extra or ["--search-size", "18", "30"])
        out = subprocess.run(args + [os.path.join(data_dir, name + ".pcd")], check=True, capture_output=True, text=True)
        lines = out.stdout.strip().splitlines()
        hyp = [l.split()[1:] for l in lines if l.startswith("hypothesis ")]
        final = lines[-1].split()
        show_best = "--show-only-best" in args
        want = [] if show_best else [(max(v[2] - 20, 10), r * 15) for r, v in enumerate(w["roll_best"][:w["rolls_done"]]) if v[2] > 70]
        assert [(int(h[0]), int(h[13])) for h in hyp] == want, (key, hyp, want)
        assert int(final[0]) == w["eval"] and int(final[13]) == w["roll_idx"] * 15
        np.testing.assert_allclose([float(t) for t in final[1:7]], list(w["gp1"]) + list(w["gp2"]), atol=1e-4)
        np.testing.assert_allclose([float(t) for t in final[7:10]], w["av"], atol=1e-6)
        # the text itself: floats streamed with 6 significant digits, like the reference's stringstream
        eng = make_engine(data_dir, surrogate)
        o = eng.score(capi.load_pcd(os.path.join(data_dir, name + ".pcd")),
                      capi.default_input(grasp_area_length_x=32, grasp_area_length_y=32 if show_best else 44, show_only_best_grasp=int(show_best)))
        eng.close()
        f32 = lambda v: "%g" % float(np.float32(v))
        text = " ".join(["%d" % o["eval"]] + [f32(v) for v in o["grasp_point1"] + o["grasp_point2"] + o["approach_vector"]] +
                        ["%g" % v for v in o["averaged_grasp_point"]] + ["%d" % (o["best_roll"] * 15)])
        assert lines[-1] == text, (lines[-1], text)
    # --gpus 1: the same request through haf_create_multi / haf_score_sharded (one RCCL rank)
    out1 = subprocess.run([cli, "--features", f_, "--range", r_, "--model", surrogate, "--search-size", "18", "18", "--gpus", "1",
                           os.path.join(data_dir, "pcd2.pcd")], check=True, capture_output=True, text=True)
    # (RCCL may print its version banner to stdout when NCCL_DEBUG is set on the box: the result is the last line)
    assert out1.stdout.strip().splitlines()[-1].split() == tok and "1 shards on 1 RCCL ranks" in out1.stderr
    # rolls of pcd2/C2 sharded 3 ways (4 + 4 + 4) on this box's one GPU, and -- where the box has more -- over two GPUs
    out3 = subprocess.run([cli, "--features", f_, "--range", r_, "--model", surrogate, "--search-size", "18", "18", "--gpus", "1",
                           "--shards-per-gpu", "3", os.path.join(data_dir, "pcd2.pcd")], check=True, capture_output=True, text=True)
    assert out3.stdout.strip().splitlines()[-1].split() == tok and "3 shards on 1 RCCL ranks" in out3.stderr
    if _n_gpus() >= 2:
        out2 = subprocess.run([cli, "--features", f_, "--range", r_, "--model", surrogate, "--search-size", "18", "18", "--gpus", "2",
                               os.path.join(data_dir, "pcd2.pcd")], check=True, capture_output=True, text=True)
        assert out2.stdout.strip().splitlines()[-1].split() == tok and "2 shards on 2 RCCL ranks" in out2.stderr


def test_shim_grid_callback_delivers_the_per_roll_grasp_grid(data_dir, surrogate, orc, tmp_path):
    """Round 3: ros_shim/shim_core.h hands the adapter, roll by roll, what publish_grasp_grid (server.cpp:901-902,
    979-1016) turns into markers: every cell of point_inside_box_grid with its base-frame position (987-989, 996) and
    graspseval[row][col].  Through the CLI (--grid-out): the cells are exactly the oracle's mask, the values the oracle's vote grid,
    the positions the reference's expression; with show_only_best the rolls behind the early exit (362-365) are not delivered."""
    import subprocess
    f_, r_ = _files(data_dir)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "haf_grasp_cli")
    xyz = pcdio.load_pcd(os.path.join(data_dir, "plastic_mug2.pcd"))
    for extra, okw in (([], dict()), (["--show-only-best"], dict(show_only_best=1)),
                       (["--center", "0.01", "-0.02", "0.005", "--gripper-width", "2"], dict(center=(0.01, -0.02, 0.005), gripper_width=2))):
        gout = str(tmp_path / "grid.txt")
        subprocess.run([cli, "--features", f_, "--range", r_, "--model", surrogate, "--search-size", "18", "30", "--grid-out", gout] + extra +
                       [os.path.join(data_dir, "plastic_mug2.pcd")], check=True, capture_output=True, text=True)
        want = orc.run(xyz, O.make_cfg(), O.make_input(length_x=32, length_y=44, **okw))
        rows = np.loadtxt(gout, ndmin=2)
        assert sorted(set(rows[:, 0].astype(int))) == list(range(want["rolls_done"]))
        cx, cy, cz = okw.get("center", (0.0, 0.0, 0.0))
        gw = okw.get("gripper_width", 1)
        for roll in range(want["rolls_done"]):
            rr = rows[rows[:, 0] == roll]
            cells = np.argwhere(want["mask"][roll] == 1)
            assert (rr[:, 1:3].astype(int) == cells).all()                                        # row-major, exactly the mask
            assert (rr[:, 6] == want["graspseval"][roll][cells[:, 0], cells[:, 1]]).all()
            x0 = np.float32(cx - 0.01 * 56 / 2 / gw)
            y0 = np.float32(cy - 0.01 * 56 / 2)
            np.testing.assert_allclose(rr[:, 3], np.float32(x0) + 0.01 / gw * cells[:, 0], atol=2e-7)
            np.testing.assert_allclose(rr[:, 4], np.float32(y0) + 0.01 * cells[:, 1], atol=2e-7)
            np.testing.assert_allclose(rr[:, 5], np.float32(cz + 0.15), atol=1e-7)


def test_recheck_tiers_forced(data_dir, surrogate, orc, monkeypatch):
    """Guard tiers: fast contraction -> fp64 MFMA (GEMM form) -> libsvm's strict fp64 order.  Forcing wide bands sends
    every evaluation through tier 2, then through tier 3; labels stay identical and the decision values of tier 3 are
    the oracle's to 1e-12 of the cancellation scale (same order, same arithmetic; only exp may differ in the last bit)."""
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    want = orc.run(xyz, O.make_cfg(), oracle_input(inp))
    monkeypatch.setenv("HAF_GUARD0_REL", "1e30")           # default mode: nothing is decided by the screening pass either
    monkeypatch.setenv("HAF_NO_I8", "1")                   # the fp64 MFMA tier itself (the exact-integer tier in front of it: test_exact_integer_tier)
    for g1, g2, tol in (("1e30", None, 2.0 ** -40), ("1e30", "1e30", 1e-13)):
        monkeypatch.setenv("HAF_GUARD_REL", g1)
        if g2:
            monkeypatch.setenv("HAF_GUARD2_REL", g2)
        eng = make_engine(data_dir, surrogate)
        got = eng.score(xyz, capi.default_input(**inp))
        cnt = eng.last_counts()
        assert cnt["n_refined"] == cnt["n_rechecked"] == cnt["n_evals"] == want["n_evals"]
        assert cnt["n_strict"] == (cnt["n_evals"] if g2 else 0) or (not g2 and cnt["n_strict"] < 5)
        for roll in range(12):
            assert (eng.debug(capi.DBG_LABELS, 0, roll) == want["labels"][roll]).all()
            m = want["mask"][roll] == 1
            if m.any():
                d = eng.debug(capi.DBG_DECISION, 0, roll)
                rel = np.abs(d[m] - want["dec"][roll][m]) / want["sabs"][roll][m]
                assert rel.max() <= tol, (g2, roll, rel.max())
        assert (got["eval"], got["best_row"], got["best_col"], got["best_roll"]) == \
               (want["eval"], want["row"], want["col"], want["roll_idx"])
        eng.close()
        monkeypatch.delenv("HAF_GUARD_REL")
        if g2:
            monkeypatch.delenv("HAF_GUARD2_REL")


def test_small_requests_go_from_the_screening_pass_straight_to_the_exact_kernel(data_dir, surrogate, orc):
    """Round 4: a request whose evaluations x support vectors stay under 2^25 (BASELINE C3: 31 093 x 192) hands
    what the screening pass cannot decide straight to the one-launch exact kernel (exact attributes + fp64 MFMA decision): no
    three-pass tier in between (n_rechecked == n_refined), every stage and label the oracle's; C3 0.33 -> 0.25 ms."""
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    kw = dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0))
    eng = make_engine(data_dir, surrogate, 0, n_rolls=20, roll_step_deg=9, max_points=1 << 18)
    assert eng.screen_form() == "centred-remainder/exp"
    compare_full(eng, orc, xyz, dict(n_rolls=20, roll_step_deg=9), kw, check_dec=False)
    cnt = eng.last_counts()
    assert 0 < cnt["n_refined"] == cnt["n_rechecked"] < 0.3 * cnt["n_evals"], cnt
    eng.close()


def test_registered_host_cloud(data_dir, surrogate, orc):
    """haf_register_host_cloud / haf_cloud.on_device = 2: a cloud inside a page-locked caller buffer is read by the DMA engine where
    it lies (large clouds) or staged like any host cloud (small ones): same result either way; a cloud outside every registered
    buffer, or one with a stride, is refused."""
    f, r = _files(data_dir)
    kw = dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0))
    for name, cfg in (("table1_mult_obj_rcs_1428580506606673", dict(n_rolls=20, roll_step_deg=9)), ("pcd2", dict(n_rolls=12))):
        xyz = np.ascontiguousarray(pcdio.load_pcd(os.path.join(data_dir, name + ".pcd")), dtype=np.float32)
        eng = make_engine(data_dir, surrogate, 0, max_points=1 << 18, **cfg)
        inp = capi.default_input(**kw)
        a = eng.score(xyz, inp)
        eng.register_host(xyz)
        cl, _ = eng._cloud(xyz)
        assert cl.on_device == 2
        b = eng.score(xyz, inp)
        assert a == b, (name, a, b)
        other = xyz.copy()
        bad = capi.Cloud(other.ctypes.data_as(C.c_void_p), other.shape[0], 3, 2)
        out = capi.GraspOutput()
        assert eng._L.haf_score(eng._h, C.byref(bad), C.byref(inp), C.byref(out)) == capi.HAF_E_ARG
        eng.unregister_host(xyz)
        cl, _ = eng._cloud(xyz)
        assert cl.on_device == 0
        assert eng._L.haf_unregister_host_cloud(eng._h, C.c_void_p(xyz.ctypes.data)) == capi.HAF_E_ARG
        assert eng.score(xyz, inp) == a
        eng.close()


def test_model_is_classified_at_creation(data_dir, surrogate, orc, monkeypatch, tmp_path):
    """Round 3: haf_create scores a synthetic table scene and settles the form of the screening pass for the MODEL
    before the first goal, so identical calls report identical counters from the first one on.  Round 4: the ill-conditioned
    surrogate -- which the plain and the SUMSQ form leave 90 % and 70 % undecided -- is served by the centred-remainder form
    (SCREEN_CR_EXP: under a third undecided on this cloud, 2 % on the calibration scene); a well-conditioned random model keeps a form
    that leaves a few per cent."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")                  # tiers as such, on requests of reference size
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    eng = make_engine(data_dir, surrogate)
    st = eng.screen_state()
    assert st["active"] and st["variant"] == 2 and st["shares"][0] > 0.6 and 0 <= st["shares"][2] < 0.1, st
    counts = []
    for _ in range(3):
        compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
        counts.append(eng.last_counts())
    assert 0 < counts[0]["n_refined"] < 0.4 * counts[0]["n_evals"] and counts[0] == counts[1] == counts[2], counts
    st2 = eng.screen_state()
    assert (st2["variant"], st2["active"]) == (st["variant"], st["active"])
    eng.close()
    f, r = _files(data_dir)
    path = str(tmp_path / "rand300.model")
    models.write_random_model(path, 300, seed=2, balanced=True)
    eng = make_engine(data_dir, path)
    o = O.Oracle(f, r, path)
    counts = []
    for _ in range(2):
        compare_full(eng, o, xyz, dict(n_rolls=12), inp)
        counts.append(eng.last_counts())
    assert 0 < counts[0]["n_refined"] < 0.25 * counts[0]["n_evals"] and counts[0] == counts[1], counts
    eng.close()


@pytest.fixture(scope="module")
def trained_model(tmp_path_factory):
    """The repo's large genuine libsvm-3.12 model (tests/golden/trained.model.npz: 8964 SVs, C 2048, gamma 2^-13, trained by the
    reference svm-train on 24000 harvested rows, tools/make_trained_model.py), unpacked to its text file."""
    path = str(tmp_path_factory.mktemp("trained") / "trained.model")
    models.unpack_trained_model(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained.model.npz"), path)
    return path


def test_every_form_of_the_screening_pass_gives_the_oracles_labels(data_dir, surrogate, orc, trained_model, monkeypatch, tmp_path):
    """Round 4: the screening pass has four forms (kernels.h SCREEN_*: plain, SUMSQ, and the centred-remainder form with the exp or
    the polynomial epilogue).  Each of them pinned (HAF_SCREEN_VARIANT), on three models -- the 172-SV surrogate, a seeded random
    model and the 8964-SV trained model, whose decisions are 1e-7 of sum|coef|K -- every stage and every label must be the
    oracle's; what a form cannot decide goes on (n_refined), and a form that overflows the refinement list falls back to the
    three-pass kernel for that call.  The form calibrate() picks by itself must be the one with the fewest undecided evaluations
    among those it tried, and for the trained model that is the polynomial form with under a fifth undecided on a real cloud (the
    other three: everything)."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    f, r = _files(data_dir)
    rnd = str(tmp_path / "rand600.model")
    models.write_random_model(rnd, 600, seed=7, balanced=True)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    refined = {}
    for name, model, o in (("surrogate", surrogate, orc), ("random", rnd, O.Oracle(f, r, rnd)), ("trained", trained_model, O.Oracle(f, r, trained_model))):
        for v in (0, 1, 2, 3):
            monkeypatch.setenv("HAF_SCREEN_VARIANT", str(v))
            eng = make_engine(data_dir, model)
            assert eng.screen_state()["variant"] == v
            got, want = compare_full(eng, o, xyz, dict(n_rolls=12), inp, check_dec=False)
            refined[(name, v)] = eng.last_counts()["n_refined"] / float(want["n_evals"])
            eng.close()
        monkeypatch.delenv("HAF_SCREEN_VARIANT")
        eng = make_engine(data_dir, model)
        st = eng.screen_state()
        tried = [s_ for s_ in st["shares"] if s_ >= 0]
        assert st["active"] and st["shares"][st["variant"]] <= min(tried) + 0.01, st
        compare_full(eng, o, xyz, dict(n_rolls=12), inp, check_dec=False)
        refined[(name, "auto")] = (st["variant"], eng.last_counts()["n_refined"] / float(want["n_evals"]))
        eng.close()
    STATS["screening_forms_refined_share"] = {"%s/%s" % k: v for k, v in refined.items()}
    assert refined[("surrogate", 2)] < 0.4 < refined[("surrogate", 0)]
    assert refined[("trained", "auto")][0] == 3 and refined[("trained", 3)] < 0.2 and min(refined[("trained", v)] for v in (0, 1, 2)) > 0.95
    assert refined[("random", 2)] <= refined[("random", 0)] + 0.01


def test_sv_range_split_of_the_screening_pass(data_dir, tmp_path, monkeypatch):
    """Round 4: a request that does not fill the chip (up to 65 536 evaluations: every reference-sized request against a model of
    thousands of SVs) is split over SV ranges -- workgroup (x, k) of k_svm_screen<., PART> sweeps the k-th slice of either coefficient
    group, k_screen_combine adds the partial class sums in fp64 and runs the same decision tail; the number of slices follows the
    LIVE evaluation count on the device.  Pinned here (HAF_SCREEN_PARTS: 1 = never, 2, 16; unset = the engine's rule) for all four
    forms of the pass and for its list mode (tier 0b): every stage and label the oracle's, and the bands the same up to the one extra
    rounding of the combined sums (the undecided counts agree within 2 %)."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    f, r = _files(data_dir)
    rnd = str(tmp_path / "rand1200.model")                    # (38 SV tiles: the engine's own rule splits from 32 tiles on)
    models.write_random_model(rnd, 1200, seed=3, balanced=True)
    clu = str(tmp_path / "clustered1100.model")
    models.write_clustered_model(clu, 1100, seed=5)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table2_mult_obj_rcs_1428580941635676.pcd"))
    cfg, inp = dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56)
    counts = {}
    for model, forms in ((rnd, (0, 1, 2)), (clu, (2, 3))):
        o = O.Oracle(f, r, model)
        for v in forms:
            monkeypatch.setenv("HAF_SCREEN_VARIANT", str(v))
            monkeypatch.setenv("HAF_T0B", "1" if v == 0 else "0")          # (plain form: the list mode behind it as well)
            for parts in ("1", "2", "16", None):
                if parts is None:
                    monkeypatch.delenv("HAF_SCREEN_PARTS", raising=False)
                else:
                    monkeypatch.setenv("HAF_SCREEN_PARTS", parts)
                eng = make_engine(data_dir, model, testing=True)
                assert eng.screen_state()["variant"] == v
                compare_full(eng, o, xyz, cfg, inp, check_dec=False)
                c = eng.last_counts()
                counts[(os.path.basename(model), v, parts or "auto")] = c["n_refined"]
                assert c["n_refined"] < c["n_evals"] and (c["n_refined"] > 0 or v == 3), (model, v, parts, c)
                eng.close()
            base = counts[(os.path.basename(model), v, "1")]
            for parts in ("2", "16", "auto"):
                assert abs(counts[(os.path.basename(model), v, parts)] - base) <= 0.02 * base + 2, (model, v, parts, counts)
    STATS["sv_range_split_refined"] = {"%s/form%d/parts-%s" % k: n for k, n in counts.items()}
    # the three-pass tier's list mode cuts short lists into up to sixteen SV ranges as well (models of >= 1024 SVs: h_list_parts)
    big = rnd
    o = O.Oracle(f, r, big)
    monkeypatch.setenv("HAF_SCREEN_VARIANT", "0")
    monkeypatch.setenv("HAF_T0B", "0")
    monkeypatch.setenv("HAF_T1_SKIP", "0")
    monkeypatch.delenv("HAF_SCREEN_PARTS", raising=False)
    eng = make_engine(data_dir, big, testing=True)
    for name, cf, ip in (("pcd3", dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=44)),
                         ("table2_mult_obj_rcs_1428580941635676", cfg, inp)):
        compare_full(eng, o, pcdio.load_pcd(os.path.join(data_dir, name + ".pcd")), cf, ip, check_dec=False)
        c = eng.last_counts()
        assert 0 < c["n_rechecked"] < c["n_refined"] < c["n_evals"], c
    eng.close()


def test_tier_0b_behind_a_first_pass_that_overflows_its_list(data_dir, golden_dir, monkeypatch):
    """Round 4 (found by fuzz campaign 48 as a GPU memory fault): a pinned plain first pass on a model it cannot serve -- the committed
    surrogate: nine tenths of the evaluations inside the band -- leaves more undecided than its list holds.  The host notices after the
    request and redoes the decision stage; until then the second pass and the compaction behind it must stop at the list's END, not at
    the counter: beyond it lie stale flag words and foreign memory, and evaluation ids read from there were written through by the
    tiers behind.  Ten-step form on the reference's grid and low-rank form on a 128 x 128 grid: every stage and label the oracle's."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_SCREEN_VARIANT", "0")
    monkeypatch.setenv("HAF_T0B", "1")
    f, r = _files(data_dir)
    model = os.path.join(golden_dir, "surrogate.model")
    o = O.Oracle(f, r, model)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    cfg, inp = dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0))
    eng = make_engine(data_dir, model, testing=True)
    compare_full(eng, o, xyz, cfg, inp, check_dec=False)
    eng.close()
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")
    G = 128
    cloud = models.synthetic_cloud(grid=G, k=2, seed=9)
    cfg = dict(n_rolls=3, roll_step_deg=60, grid_h=G, grid_w=G, max_points=2 * G * G)
    inp = dict(grasp_area_length_x=G + 14, grasp_area_length_y=G)
    eng = make_engine(data_dir, model, testing=True, **cfg)
    compare_full(eng, o, cloud, cfg, inp, check_dec=False)
    assert eng.screen_low_rank()["rank"] == 158
    eng.close()


@pytest.mark.parametrize("kname", ["linear", "poly", "sigmoid", "nu_rbf"])
def test_other_libsvm_kernels_and_nu_svc(data_dir, golden_dir, tmp_path, kname):
    """Round 5 (VERDICT r4 missing 3: the drop-in was narrower than the binary it replaces).  svm-predict serves LINEAR / POLY / SIGMOID
    kernels (Kernel::k_function, svm.cpp:318-371) and nu-SVC; the reference's own model is an easy.py product (RBF C-SVC), so the fast
    tiers stay RBF-only and a model with another kernel goes through the tier that IS libsvm's arithmetic for every evaluation
    (k_recheck: index order, model order, unfused fp64; the C library's tanh decides what lies within a last-bit error of zero).
    Models: written by the REFERENCE svm-train on the surrogate's training rows (kernel_models.npz); the oracle's decisions with them
    are pinned to the reference library bit for bit (tests/test_oracle.py).  Here: C2 on pcd2 and C3 on a table cloud, every stage and
    label the oracle's; the nu-SVC model is RBF and takes the fast tiers."""
    path = models.unpack_kernel_model(golden_dir, kname, str(tmp_path / (kname + ".model")))
    f, r = _files(data_dir)
    o = O.Oracle(f, r, path)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    eng = make_engine(data_dir, path, n_rolls=12)
    got, want = compare_full(eng, o, xyz, dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=32), check_dec=False)
    n2 = eng.last_counts()
    assert (n2["n_strict"] == want["n_evals"]) == (kname != "nu_rbf"), (kname, n2)      # every evaluation through the libsvm-order tier, or none
    eng.close()
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    cfg, inp = dict(n_rolls=20, roll_step_deg=9, max_points=1 << 18), dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0))
    eng = make_engine(data_dir, path, **cfg)
    compare_full(eng, o, xyz, cfg, inp, check_dec=False)
    eng.close()


def test_guard_zones_around_every_device_buffer(data_dir, golden_dir, tmp_path, monkeypatch):
    """Round 5 (the list-overflow bug of round 4 lived through a round with every test green).  In the testing build every
    device buffer -- lists, operand images, flag words, tables -- lies between two guard zones of a fixed pattern (csrc/engine_state.h:
    DevBuf) and, with HAF_CANARY_CHECK set, the engine checks all of them after EVERY request and fails the request when one is damaged.
    Here: (1) the hook sees the engine's buffers; (2) requests in the regimes where lists overflow -- a pinned first pass on a model it
    cannot serve (round 4's case), screening lists of 256 entries, exact-tier windows of 64, bands scaled up so that the lists behind them
    fill -- leave every zone intact and match the oracle stage by stage (the tier lists hold an entry per evaluation and cannot
    overflow: their capacity is not a knob); (3) a write one element past a list's end, which is
    what a producer that ignores its capacity does, IS reported, by the allocating source line."""
    f, r = _files(data_dir)
    model = os.path.join(golden_dir, "surrogate.model")
    o = O.Oracle(f, r, model)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    cfg, inp = dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0))
    monkeypatch.setenv("HAF_CANARY_CHECK", "1")
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    regimes = [dict(HAF_SCREEN_VARIANT="0", HAF_T0B="1"),                                    # round 4's case: the counter exceeds the list
               dict(HAF_SCREEN_VARIANT="0", HAF_T0B="1", HAF_FLAG0_CAP="256"),
               dict(HAF_SCREEN_VARIANT="2", HAF_T0B="0", HAF_FLAG0_CAP="256", HAF_FLAG_WINDOW="64"),
               dict(HAF_SCREEN_VARIANT="3", HAF_T1_SKIP="1", HAF_FLAG0_CAP="512"),
               dict(HAF_SCREEN_VARIANT="1", HAF_T1_SKIP="0", HAF_FLAG_WINDOW="64", HAF_GUARD_REL="4000"),
               dict(HAF_FLAG_WINDOW="64", HAF_GUARD0_REL="50"),
               dict(HAF_FLAG0_CAP="256", HAF_FLAG_WINDOW="64", HAF_GUARD0_REL="50", HAF_GUARD_REL="4000")]
    outcomes = []
    crumbs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "canary_regimes.log")          # (which regime a GPU fault, should one ever come, belongs to)
    os.makedirs(os.path.dirname(crumbs), exist_ok=True)
    for reg in regimes:
        with open(crumbs, "a") as fh:
            fh.write("regime %r\n" % (reg,))
        with monkeypatch.context() as mp:
            for k, v in reg.items():
                mp.setenv(k, v)
            eng = make_engine(data_dir, model, testing=True)
            compare_full(eng, o, xyz, cfg, inp, check_dec=False)
            outcomes.append(eng.overflow_stats())
            bad, rep, n_buf = capi.check_canaries()
            assert bad == 0, (reg, rep)
            assert n_buf >= 40                                                                 # every DevBuf of the engine is registered
            eng.close()
    assert sum(x["screening_list_overflows"] for x in outcomes) >= 3 and sum(x["extra_windows"] for x in outcomes) >= 3, outcomes
    # (3) the detector detects: one int behind the end of the screening list
    eng = make_engine(data_dir, model, testing=True)
    assert capi.check_canaries()[0] == 0
    assert eng._L.haf_test_poke_flag0_list(eng._h, 0, 1, 12345) == 0
    bad, rep, _ = capi.check_canaries()
    assert bad == 1 and "engine_tables.cpp" in rep and "back guard damaged at end+0" in rep and "0x00003039" in rep, rep
    with pytest.raises(capi.HafError, match="guard zones damaged"):
        eng.score(xyz, capi.default_input(**inp))                                            # ... and a request on such an engine fails loudly
    eng.close()
    assert capi.check_canaries()[0] == 0                                                       # (released buffers leave the registry)
    STATS["canary_regimes"] = dict(regimes=len(regimes), outcomes=outcomes, buffers=n_buf)


def test_low_rank_form_of_the_screening_pass(data_dir, tmp_path, monkeypatch):
    """Round 4: the 299 HAF slots are linear functionals of the 15 x 15 window (fv.cpp:141-199) spanning 158 dimensions, so whole requests
    on large grids are swept on a projected operand (k_project: y = fp16(B'p); k_svm_screen_lr: 6 k-steps instead of 10) and the band
    carries what the projection drops -- the "%.4g" rounding and the fp32 roundings of the reference's feature arithmetic, bounded per
    evaluation -- and the rounding of y.  Against the oracle, stage by stage, on a 128 x 128 grid with rolled (partly empty) grids, both
    centred-remainder forms, the fast and the general feature paths, a cloud with negative heights (the integral image is not
    monotone: no wave passes the exactness test of the region sums as a whole, every region is tested on its own four corners) -- and
    against the ten-step form (HAF_FLAG_FULL_RANK): the same labels, and about as many evaluations left undecided."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")                 # every request takes the thread-per-evaluation feature kernel
    G = 128
    f, r = _files(data_dir)
    rnd = str(tmp_path / "rand300.model")
    models.write_random_model(rnd, 300, seed=3, balanced=True)
    clu = str(tmp_path / "clustered260.model")
    models.write_clustered_model(clu, 260, seed=5)
    xyz = models.synthetic_cloud(grid=G, k=2, seed=4)
    neg = xyz.copy()
    cells = np.arange(0, G * G, 53)                           # both points of every 53rd cell (the cloud holds the cells' points pairwise)
    neg[2 * cells, 2] = neg[2 * cells + 1, 2] = -0.4           # heights in (-0.99, 0) survive generate_grid (server.cpp:522-528)
    cfg = dict(n_rolls=4, roll_step_deg=25, grid_h=G, grid_w=G, max_points=2 * G * G)
    inp = dict(grasp_area_length_x=G, grasp_area_length_y=G)
    stats = {}
    for model, v in ((rnd, 2), (clu, 3), (rnd, 0)):               # (0: the plain epilogue on the projected operands, tier 0b behind it)
        o = O.Oracle(f, r, model)
        monkeypatch.setenv("HAF_SCREEN_VARIANT", str(v))
        monkeypatch.setenv("HAF_T0B", "1" if v == 0 else "0")
        left = {}
        for name, cloud, flags, nofast in (("low-rank", xyz, 0, False), ("full-rank", xyz, capi.FLAG_FULL_RANK, False), ("low-rank/general groups", xyz, 0, True),
                                           ("low-rank/negative heights", neg, 0, False)):
            if nofast:
                monkeypatch.setenv("HAF_NO_FAST_GROUPS", "1")
            else:
                monkeypatch.delenv("HAF_NO_FAST_GROUPS", raising=False)
            eng = make_engine(data_dir, model, mode=flags, testing=True, **cfg)
            lr = eng.screen_low_rank()
            assert lr["available"] == (flags == 0) and lr["rank"] == 158, lr
            compare_full(eng, o, cloud, cfg, inp, check_dec=False)
            lr = eng.screen_low_rank()
            assert lr["last_used"] == (flags == 0), (name, lr)
            c = eng.last_counts()
            left[name] = c["n_refined"]
            assert c["n_refined"] <= c["n_evals"], (name, c)
            eng.close()
        # the projected operand leaves about as much undecided as the ten-step form (the borders of the grid and of the rolled cloud
        # get a wider band)
        assert left["low-rank"] <= 1.5 * left["full-rank"] + 0.02 * c["n_evals"], left
        assert left["low-rank/general groups"] <= 1.5 * left["full-rank"] + 0.02 * c["n_evals"], left
        assert left["low-rank/negative heights"] <= 0.25 * c["n_evals"], left
        stats["form%d" % v] = left
        # the projection as the sweep's prologue (the shipped form) against projection and sweep as two launches (k_project): the same
        # operand bits, the same |y^ - y32|^2, so every decision value and every band (|dec^| / band of the decided cells) is identical
        monkeypatch.delenv("HAF_NO_FAST_GROUPS", raising=False)
        monkeypatch.setenv("HAF_T0B_NO_GATHER", "1")     # (the gather form of tier 0b exists on the fused kernel only: compare like with like)
        grids = {}
        for unfused in (False, True):
            if unfused:
                monkeypatch.setenv("HAF_LR_UNFUSED", "1")
            else:
                monkeypatch.delenv("HAF_LR_UNFUSED", raising=False)
            eng = make_engine(data_dir, model, testing=True, **cfg)
            eng.score(xyz, capi.default_input(**inp))
            assert eng.screen_low_rank()["last_used"]
            grids[unfused] = [(eng.debug(capi.DBG_DECISION, 0, roll), eng.debug(capi.DBG_SCREEN_MARGIN, 0, roll)) for roll in range(cfg["n_rolls"])]
            eng.close()
        monkeypatch.delenv("HAF_LR_UNFUSED", raising=False)
        monkeypatch.delenv("HAF_T0B_NO_GATHER", raising=False)
        for (d0, m0), (d1, m1) in zip(grids[False], grids[True]):
            assert np.array_equal(d0, d1, equal_nan=True) and np.array_equal(m0, m1, equal_nan=True), "fused and two-launch forms differ"
            assert (np.nan_to_num(m0) > 0).any()
    STATS["low_rank_undecided"] = stats


def test_short_lists_of_a_big_model_go_straight_to_the_fp64_tier(data_dir, tmp_path, monkeypatch):
    """Round 5, the short-list gate (engine_request.cpp, k_short_list_gate): a SMALL request against a BIG model -- the reference's C3
    request (table1, 56 x 56 cm, 20 rolls) against the bench's 4 096-SV model -- leaves the screening passes a few dozen evaluations;
    tier 1 and the exact-integer tier behind it would be ~120 us of launches and minimum chains for them.  At most 256 entries in front
    of tier 1 of a model of >= 2 048 SVs go straight to the fp64 MFMA tier's list.  Every stage and label the oracle's with the gate
    (the product's rule) and without it (HAF_NO_SHORT_GATE); with it nothing enters tier 1's successor lists except through the gate:
    the exact-integer tier sees no entry, the fp64 tier sees exactly what the screening passes left; guard zones intact."""
    monkeypatch.setenv("HAF_CANARY_CHECK", "1")
    path = models.write_random_model(str(tmp_path / "rand4096.model"), 4096, seed=42, balanced=True)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, path)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    cfg = dict(n_rolls=20, roll_step_deg=9, max_points=1 << 18)
    inp = dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0))
    seen = {}
    for gate in (True, False):
        if gate:
            monkeypatch.delenv("HAF_NO_SHORT_GATE", raising=False)
        else:
            monkeypatch.setenv("HAF_NO_SHORT_GATE", "1")
        eng = make_engine(data_dir, path, testing=True, **cfg)
        compare_full(eng, o, xyz, cfg, inp, check_dec=False)
        cnt, ex = eng.last_counts(), eng.last_exact_tiers()
        seen[gate] = (cnt, ex)
        assert 0 < cnt["n_refined"] <= 256, cnt                      # (what the screening passes left: the gate's range)
        if gate:
            assert ex["n_integer"] == 0 and ex["n_fp64"] == cnt["n_refined"] == cnt["n_rechecked"], (cnt, ex)
        else:
            assert cnt["n_rechecked"] <= cnt["n_refined"] and ex["n_integer"] == cnt["n_rechecked"], (cnt, ex)
        bad, rep, _ = capi.check_canaries()
        assert bad == 0, rep
        eng.close()
    assert seen[True][0]["n_refined"] == seen[False][0]["n_refined"]
    STATS["short_list_gate"] = {"with": seen[True], "without": seen[False]}


def test_low_rank_pass_never_trusts_a_negative_computed_region_sum(data_dir, tmp_path, monkeypatch):
    """ADVICE r4 (features.hip, path A of the low-rank feature kernel): a wave of 64 neighbouring cells whose windows pass three wave-wide
    checks takes every region sum ((a - b) - c) + d as EXACT.  The checks take R >= 0 from the monotone TRUE integral image; the STORED
    corners are fp32 roundings of fp64 sums, and over an area without heights that has heights beside it (true R = 0, a - b = c - d
    only before rounding) the computed R comes out at -1 ... -2 ulp for a good part of the regions -- and then (a - b) - c may have
    rounded without the band being charged.  Since round 5 the smallest computed region sum rides along and a negative one voids
    the evaluation's pass.  The adversarial scene: a 128 x 128 grid with eight-row stripes without points to the right of column 30
    (corners up to ~3 000: an ulp of 2.4e-4; every cell keeps heights within its 9 x 9 box, so every cell of the area is evaluated), a
    grasp area 100 cells wide (a wave that starts in column 0 of the integral image never passes the wave-wide checks on a grid this
    small), roll 0 only, centred-remainder/exp form, no tier 0b behind it.  From the oracle's integral image the test rebuilds the
    device's evaluation order (k_compact: the whole 64-cell chunks of every row first, taken from the row's END), recomputes -- fp32, the reference's order --
    the kernel's wave-wide rule and every region sum of every HAF attribute svm-scale keeps, and asserts: path-A waves and evaluations
    with a negative computed sum exist in number, NONE of those was decided by the pass (screening margin not above 1), evaluations
    of the same waves without one were -- and every label is the oracle's (compare_full)."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")
    monkeypatch.setenv("HAF_SCREEN_VARIANT", "2")
    monkeypatch.setenv("HAF_T0B", "0")
    G = 128
    f, r = _files(data_dir)
    model = models.write_random_model(str(tmp_path / "rand300.model"), 300, seed=3, balanced=True)
    o = O.Oracle(f, r, model)
    xyz = models.synthetic_cloud(grid=G, k=2, seed=4).reshape(G, G, 2, 3)
    keep = np.ones((G, G), bool)
    for a0 in range(30, 120, 12):
        keep[a0:a0 + 8, 30:] = False                                  # (first grid index = the integral image's row)
    xyz = np.ascontiguousarray(xyz[keep].reshape(-1, 3))
    cfg = dict(n_rolls=1, roll_step_deg=25, grid_h=G, grid_w=G, max_points=2 * G * G)
    inp = dict(grasp_area_length_x=G, grasp_area_length_y=100)
    eng = make_engine(data_dir, model, testing=True, **cfg)
    _, want = compare_full(eng, o, xyz, cfg, inp, check_dec=False)
    assert eng.screen_low_rank()["last_used"]
    margin = eng.debug(capi.DBG_SCREEN_MARGIN, 0, 0)
    cnt = eng.last_counts()
    eng.close()
    II, msk = want["integral"][0], want["mask"][0] == 1
    assert (want["heights"][0] >= 0).all()
    # the waves of the thread-per-evaluation feature kernel that are 64 neighbours of one row (prestages.hip, k_compact: of every
    # row the masked cells behind its first count % 64, in chunks of 64; the remainders follow behind all of them)
    ci, cj = [], []
    for i in range(G):
        cols = np.nonzero(msk[i])[0]
        for c0 in range(len(cols) % 64, len(cols), 64):
            ch = cols[c0:c0 + 64]
            if (np.diff(ch) == 1).all():
                ci.append(np.full(64, i))
                cj.append(ch)
    ci, cj = np.concatenate(ci), np.concatenate(cj)
    # the kernel's wave-wide rule, recomputed (features.hip: bottom row <= 2 x top row per column of the window, window total -- rounded
    # up -- below the window's first corner; a window that starts in column 0 of the integral image takes its second corner)
    tl, bl = II[ci - 7, cj - 7], II[ci + 7, cj - 7]
    tr, br = II[ci - 7, cj + 7], II[ci + 7, cj + 7]
    colok = np.ones(len(ci), bool)
    for c in range(15):
        colok &= II[ci + 7, cj - 7 + c] <= np.float32(2.0) * II[ci - 7, cj - 7 + c]
    tA, tB = (br - tr).astype(np.float32), (bl - tl).astype(np.float32)
    T = ((tA - tB).astype(np.float32) + np.float32(2.0e-7) * (np.abs(tA) + np.abs(tB))) * np.float32(1.0001)
    lane = np.tile(np.arange(64), len(ci) // 64)
    dmin = np.where((cj - 7 == 0) & (lane == 0), II[ci - 7, 1], tl)
    clear = colok & (T * np.float32(1.01) < dmin)                      # (a margin: only waves that pass the rule beyond doubt are looked at)
    path_a = clear.reshape(-1, 64).all(axis=1)
    assert path_a.sum() >= 40, int(path_a.sum())
    sel = np.repeat(path_a, 64)
    si, sj = ci[sel], cj[sel]
    reg, wgt = o.feature_table()
    _, _, fmin, fmax, present = o.range_table()
    rmin = np.zeros(len(si), np.float32)
    for fi in range(min(reg.shape[0], 302)):                           # (the HAF attributes: a SHAF slot is passed through, not bounded)
        if fi + 1 < len(present) and present[fi + 1] and fmin[fi + 1] == fmax[fi + 1]:
            continue                                                  # (an attribute svm-scale drops: the screening pass has no slot for it)
        for k in range(4):
            x1, x2, y1, y2 = (int(v) for v in reg[fi, 4 * k:4 * k + 4])
            if wgt[fi, k] == 0.0 or x2 < x1 or y2 < y1 or (x2 == 0 and y2 == 0):
                continue
            a, b = II[si - 7 + x2 + 1, sj - 7 + y2 + 1], II[si - 7 + x1, sj - 7 + y2 + 1]
            c, d0 = II[si - 7 + x2 + 1, sj - 7 + y1], II[si - 7 + x1, sj - 7 + y1]
            R = (((a - b).astype(np.float32) - c).astype(np.float32) + d0).astype(np.float32)    # fv.cpp:161-162, fp32, left to right
            rmin = np.minimum(rmin, R)
    neg = rmin < 0
    decided = np.nan_to_num(margin[si, sj]) > 1.0
    assert neg.sum() >= 1000, int(neg.sum())
    assert not decided[neg].any(), "the low-rank pass decided %d evaluations whose computed region sums include a negative one" % int(decided[neg].sum())
    assert decided[~neg].sum() > 0.5 * (~neg).sum(), (int(decided[~neg].sum()), int((~neg).sum()))
    STATS["low_rank_negative_region_sums"] = dict(path_a_waves=int(path_a.sum()), evaluations=int(len(si)), with_a_negative_sum=int(neg.sum()),
                                                  decided_with=int(decided[neg].sum()), decided_without=int(decided[~neg].sum()),
                                                  without=int((~neg).sum()), smallest=float(rmin.min()), left_by_the_pass=int(cnt["n_refined"]))


def test_tier_0b_gathers_from_the_low_rank_first_pass(data_dir, tmp_path, monkeypatch):
    """Round 5: behind a low-rank first pass with the plain epilogue, tier 0b is the low-rank sweep with the centred-remainder epilogue in
    its GATHER form (k_svm_screen_lr<CR_EXP, FUSED, GATHER>): operand images and raw sums of the first pass by evaluation id (L = ln2 p.g
    rides in raw[6]), no second feature kernel.  A rolled 128 x 128 grid with negative heights and a balanced 517-SV random model
    (a plain first pass leaves a few per cent), the form pinned: every stage and label the oracle's in the gather form AND in the
    list-image form it replaces (HAF_T0B_NO_GATHER), tier 0b really ran, and the projected form's band leaves at most a fifth more
    undecided than the full-rank one's (within 10 % per evaluation)."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")
    monkeypatch.setenv("HAF_SCREEN_VARIANT", "0")
    monkeypatch.setenv("HAF_T0B", "1")
    monkeypatch.setenv("HAF_CANARY_CHECK", "1")
    path = models.write_random_model(str(tmp_path / "r517.model"), 517, seed=5, gamma=1.0 / 323, balanced=True, rho=0.01)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, path)
    G = 128
    cloud = models.synthetic_cloud(grid=G, k=2, seed=11)
    cloud[::7, 2] -= 0.3                                                     # some heights below zero: the per-region exactness path
    cfg = dict(n_rolls=4, roll_step_deg=35, grid_h=G, grid_w=G, max_points=2 * G * G)
    inp = dict(grasp_area_length_x=G + 14, grasp_area_length_y=G)
    left = {}
    for mode in ("gather", "images"):
        with monkeypatch.context() as mp:
            if mode == "images":
                mp.setenv("HAF_T0B_NO_GATHER", "1")
            eng = make_engine(data_dir, path, testing=True, **cfg)
            compare_full(eng, o, cloud, cfg, inp, check_dec=False)
            assert eng.screen_low_rank()["last_used"] and eng.screen_form() == "plain"
            c = eng.last_counts()
            left[mode] = (c["n_refined"], c["n_rechecked"])
            eng.close()
    assert 0 < left["gather"][0] <= 1.2 * left["images"][0] + 16, left
    STATS["tier0b_gather_vs_images_left"] = left


@pytest.mark.parametrize("t0b,skip", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_second_screening_pass_and_tier_1_hand_over(data_dir, tmp_path, monkeypatch, t0b, skip):
    """Round 4: behind a PLAIN first pass the centred-remainder form may run once more on the first pass's LIST (tier 0b: k_svm_screen in
    list mode, operand images / bands / common factors indexed by list slot), and what the screening passes leave may skip tier 1 and
    go straight to the exact tiers (t1_skip).  All four combinations pinned (HAF_T0B, HAF_T1_SKIP): every stage and label the
    oracle's; with tier 0b fewer evaluations leave the screening passes; with the skip every one of them reaches the exact tiers."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_SCREEN_VARIANT", "0")
    monkeypatch.setenv("HAF_T0B", str(t0b))
    monkeypatch.setenv("HAF_T1_SKIP", str(skip))
    f, r = _files(data_dir)
    path = str(tmp_path / "rand900.model")
    models.write_random_model(path, 900, seed=11, balanced=True)
    o = O.Oracle(f, r, path)
    eng = make_engine(data_dir, path)
    st = eng.screen_state()
    assert st["variant"] == 0 and st["tier0b"] == bool(t0b) and st["tier1_skipped"] == bool(skip), st
    for name, cfg, inp in (("pcd3", dict(n_rolls=12), dict(grasp_area_length_x=32, grasp_area_length_y=44)),
                           ("table2_mult_obj_rcs_1428580941635676", dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56))):
        xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
        compare_full(eng, o, xyz, cfg, inp, check_dec=False)
        cnt = eng.last_counts()
        STATS.setdefault("tier0b_counts", {})["%s/t0b%d/skip%d" % (name, t0b, skip)] = cnt
        assert 0 < cnt["n_refined"] < 0.3 * cnt["n_evals"]
        if skip:
            assert cnt["n_rechecked"] == cnt["n_refined"]
        else:
            assert cnt["n_rechecked"] <= cnt["n_refined"]
    eng.close()
    ref = STATS["tier0b_counts"].get("pcd3/t0b0/skip0")
    if t0b and ref:
        assert STATS["tier0b_counts"]["pcd3/t0b1/skip%d" % skip]["n_refined"] < ref["n_refined"]


def test_screening_pass_switched_off_is_tried_again(data_dir, tmp_path, monkeypatch):
    """the adaptive rule's switch-off of the screening pass used to last for the engine's lifetime.  Now every
    reprobe_every-th full-size request (64; HAF_REPROBE_EVERY=3 here) runs the pass again and switches it back on when it leaves at
    most 60 % undecided.  The switch-off is injected (testing hook) on a model the pass serves well: two requests run in the three-pass
    mode, the third re-tries and re-enables; every request's stages and labels are the oracle's."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    monkeypatch.setenv("HAF_REPROBE_EVERY", "3")
    f, r = _files(data_dir)
    path = str(tmp_path / "rand900.model")
    models.write_random_model(path, 900, seed=11, balanced=True)
    o = O.Oracle(f, r, path)
    eng = make_engine(data_dir, path)
    assert eng.screen_state()["active"]
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table2_mult_obj_rcs_1428580941635676.pcd"))
    cfg, inp = dict(n_rolls=12), dict(grasp_area_length_x=56, grasp_area_length_y=56)
    compare_full(eng, o, xyz, cfg, inp, check_dec=False)
    screened = eng.last_counts()
    eng.set_screen_inactive()
    seen = []
    for _ in range(4):
        compare_full(eng, o, xyz, cfg, inp, check_dec=False)
        seen.append((eng.screen_state()["active"], eng.last_counts()["n_refined"]))
    # requests 1, 2: three passes for everything (nothing is "refined" behind a screening pass); request 3 re-tries and re-enables;
    # request 4 is served like the one before the switch-off
    assert [a for a, _ in seen] == [False, False, True, True], seen
    assert seen[3][1] == screened["n_refined"] and seen[2][1] == screened["n_refined"], (seen, screened)
    assert eng.screen_form() != "off"
    eng.close()


def test_a_worse_matrix_core_widens_the_bands_and_a_far_worse_one_is_refused(data_dir, surrogate, orc, monkeypatch, tmp_path):
    """VERDICT r3 item 2b.  The one measured constant in the bands is kappa, the rounding of one fp16 MFMA instruction in units of
    2^-24 (|c| + sum|a b|), taken as max(12, 1.5 x what haf_create's probe sees).  Injected (HAF_KAPPA, testing build): 24, 60 -- the
    screening band must widen monotonically (never fewer evaluations handed on, more at 60 than at the measured value), every label
    must stay the oracle's; from 64 on haf_create
    must refuse the device with the documented text.  Both for the plain form and for the centred-remainder form, on a random model
    and on the surrogate.  (Tier 1 does not narrow on the measurement at all since round 4: its band has the worst case of 32
    truncating additions, 64 u, as a floor -- ADVICE r3 -- so the injected values below 64 leave its hand-over count unchanged.)"""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")
    f, r = _files(data_dir)
    rnd = str(tmp_path / "rand600.model")
    models.write_random_model(rnd, 600, seed=3, balanced=True)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    seen = {}
    for name, model, o, variant in (("random/plain", rnd, O.Oracle(f, r, rnd), 0), ("random/cr", rnd, O.Oracle(f, r, rnd), 2), ("surrogate/cr", surrogate, orc, 2)):
        monkeypatch.setenv("HAF_SCREEN_VARIANT", str(variant))
        refined, rechecked = [], []
        for kappa in (None, 24, 60):
            if kappa is None:
                monkeypatch.delenv("HAF_KAPPA", raising=False)
            else:
                monkeypatch.setenv("HAF_KAPPA", str(kappa))
            eng = make_engine(data_dir, model)
            meas, used = (C.c_double * 2)(), (C.c_double * 2)()
            assert eng._L.haf_test_mfma_kappa(eng._h, meas, used) == 0
            assert used[0] == (kappa if kappa is not None else max(12.0, 1.5 * meas[0])) and 0 < meas[0] < 12
            compare_full(eng, o, xyz, dict(n_rolls=12), inp, check_dec=False)
            cnt = eng.last_counts()
            refined.append(cnt["n_refined"])
            rechecked.append(cnt["n_rechecked"])
            eng.close()
        assert refined[0] <= refined[1] <= refined[2] < cnt["n_evals"] and refined[2] > refined[0], (name, refined)
        seen[name] = dict(refined=refined, rechecked=rechecked)
    monkeypatch.setenv("HAF_KAPPA", "70")
    with pytest.raises(capi.HafError) as ei:
        make_engine(data_dir, rnd)
    assert ei.value.code == capi.HAF_E_DEVICE and "rounds far worse than the guard bands allow" in str(ei.value)
    STATS["kappa_injection"] = seen


def test_exact_integer_tier(data_dir, surrogate, orc, monkeypatch, tmp_path):
    """Round 3, tier 2a (csrc/exact8.hip): attributes and support vectors as fixed-point numbers in four int8 digit planes, the dot
    products EXACT in int32 on the matrix cores.  (1) Every evaluation forced through it (HAF_GUARD_REL wide open): labels, votes
    and grasp are the oracle's, its decision values are within its own band -- ~2e-7 S, nine times inside the three-pass
    kernel's -- of the oracle's, and only what lies inside that band goes on to the fp64 tier; (2) with its band forced wide
    open everything goes on, same labels; (3) a model with a support-vector component beyond the fixed-point range (|s| >= 15.87)
    is served without the tier."""
    monkeypatch.setenv("HAF_GUARD_REL", "1e30")
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    f, r = _files(data_dir)
    path = str(tmp_path / "rand300.model")
    models.write_random_model(path, 300, seed=4, balanced=True)
    for model, o in ((surrogate, orc), (path, O.Oracle(f, r, path))):
        eng = make_engine(data_dir, model, capi.FLAG_SPLIT_F16)
        got, want = compare_full(eng, o, xyz, dict(n_rolls=12), inp, check_dec=False)
        cnt, ex = eng.last_counts(), eng.last_exact_tiers()
        assert cnt["n_rechecked"] == cnt["n_evals"] == ex["n_integer"] == want["n_evals"] and ex["n_fp64"] < 0.05 * cnt["n_evals"] and cnt["n_strict"] == 0
        worst = 0.0
        for roll in range(12):
            m = want["mask"][roll] == 1
            if m.any():
                rel = np.abs(eng.debug(capi.DBG_DECISION, 0, roll)[m] - want["dec"][roll][m]) / want["sabs"][roll][m]
                worst = max(worst, float(rel.max()))
        assert worst < 1e-6, worst                                  # band: gamma delta 2 (|x| + |s|) ~ 4e-7; measured far below
        STATS.setdefault("exact_integer_tier_max_rel_err", {})[os.path.basename(model)] = worst
        eng.close()
    monkeypatch.setenv("HAF_GUARD_I8_REL", "1e30")
    eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16)
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    ex = eng.last_exact_tiers()
    assert ex["n_integer"] == ex["n_fp64"] == eng.last_counts()["n_evals"]
    eng.close()
    monkeypatch.delenv("HAF_GUARD_I8_REL")
    big = str(tmp_path / "big.model")
    with open(path) as fh:
        lines = fh.read().splitlines()
    k = lines.index("SV") + 1
    tok = lines[k].split()
    tok[5] = tok[5].split(":")[0] + ":16.5"                     # one component beyond the fixed-point range
    lines[k] = " ".join(tok)
    with open(big, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    eng = make_engine(data_dir, big, capi.FLAG_SPLIT_F16)
    compare_full(eng, O.Oracle(f, r, big), xyz, dict(n_rolls=12), inp)
    ex = eng.last_exact_tiers()
    assert ex["n_integer"] == 0 and ex["n_fp64"] == eng.last_counts()["n_evals"]
    eng.close()


def test_strict_tier_residual_is_decided_with_the_c_librarys_exp(data_dir, surrogate, orc, monkeypatch):
    """Round 3: the strict tier's exp() is the device's; an evaluation whose libsvm-order decision value lies within a last-bit exp
    error of zero (2^-44 sum|coef|) is evaluated once more on the HOST with glibc's exp, the function svm-predict calls.  Nothing
    ever comes that far, so the test forces it: every evaluation through the strict tier (HAF_GUARD_REL / HAF_GUARD2_REL wide
    open), every one of those through the host path (HAF_HOST_EXP_ALL).  Labels, votes and grasp = the oracle's, and the
    decision values the host wrote back are the oracle's BIT FOR BIT (same summation order, same libm)."""
    monkeypatch.setenv("HAF_GUARD_REL", "1e30")
    monkeypatch.setenv("HAF_GUARD2_REL", "1e30")
    monkeypatch.setenv("HAF_HOST_EXP_ALL", "1")
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=32)
    eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16, n_rolls=2)
    got, want = compare_full(eng, orc, xyz, dict(n_rolls=2), inp, check_dec=False)
    cnt = eng.last_counts()
    assert cnt["n_strict"] == cnt["n_evals"] == eng.last_strict_host() == want["n_evals"] > 0
    for roll in range(2):
        m = want["mask"][roll] == 1
        d = eng.debug(capi.DBG_DECISION, 0, roll)
        assert (d[m].view(np.uint64) == want["dec"][roll][m].view(np.uint64)).all()
    eng.close()
    monkeypatch.delenv("HAF_HOST_EXP_ALL")
    eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16, n_rolls=2)      # the normal threshold: nothing is that close to zero
    compare_full(eng, orc, xyz, dict(n_rolls=2), inp)
    assert eng.last_counts()["n_strict"] > 0 and eng.last_strict_host() == 0
    eng.close()


@pytest.mark.parametrize("mode", [pytest.param(capi.FLAG_SPLIT_F16, id="splitf16"), pytest.param(0, id="screen")])
def test_guard_list_overflow_degrades_to_windows(data_dir, surrogate, orc, monkeypatch, mode):
    """More guard-band evaluations than one window of the fp64 tier holds must NOT fail the goal (the reference never fails
    one on this path, server.cpp:778-796): the list is walked window by window.  HAF_GUARD_REL = 1e30 puts every
    evaluation of a dense 56 x 56 cloud (all 42 x 42 cells x 12 rolls) into the band, HAF_FLAG_WINDOW makes the windows
    small (5 of them and a ragged last one); labels, votes and the grasp are the oracle's."""
    monkeypatch.setenv("HAF_GUARD_REL", "1e30")            # every evaluation lands in the band of the three-pass kernel
    monkeypatch.setenv("HAF_FLAG_WINDOW", "4000")
    if mode == 0:
        monkeypatch.setenv("HAF_GUARD0_REL", "1e30")       # default mode: the screening pass hands everything on first
        xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))      # (a request the refinement list can hold)
        inp = dict(grasp_area_length_x=56, grasp_area_length_y=56)
    else:
        xyz = models.synthetic_cloud(grid=56, k=3, seed=1)
        inp = dict(grasp_area_length_x=56, grasp_area_length_y=56)
    eng = make_engine(data_dir, surrogate, mode)
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    cnt = eng.last_counts()
    assert cnt["n_rechecked"] == cnt["n_evals"] > 4000
    if mode:
        assert cnt["n_evals"] > 4 * 4000
    eng.close()


def test_screening_overflow_falls_back_to_three_passes(data_dir, surrogate, orc, monkeypatch):
    """Default mode: when more evaluations fall inside the screening band than the refinement list holds, the SAME call
    redoes the decision stage with the three-pass kernel for every evaluation (identical labels and grasp), and the engine
    stays with that kernel afterwards."""
    monkeypatch.setenv("HAF_GUARD0_REL", "1e30")           # the screening pass decides nothing
    xyz = models.synthetic_cloud(grid=56, k=3, seed=1)     # dense: ~18.7k masked cells over 12 rolls, the list holds 10.7k
    inp = dict(grasp_area_length_x=56, grasp_area_length_y=56)
    eng = make_engine(data_dir, surrogate)
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    cnt = eng.last_counts()
    assert cnt["n_refined"] == 0 and cnt["n_evals"] > 12 * 42 * 42 // 2   # more than the list holds: answered by the fallback
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)                    # and again: no screening pass any more
    assert eng.last_counts()["n_refined"] == 0
    eng.close()


def test_screening_tier_forced_and_reported(data_dir, surrogate, orc, monkeypatch, tmp_path):
    """Default mode: (1) on a well-conditioned model the screening pass decides most evaluations and reports how many it
    passed on; (2) on the surrogate model (C = 512: |dec| is tiny against sum|coef|K, nearly everything sits inside the
    band of the plain variant) the labels are still the oracle's and the engine switches to the kernel variant that measures
    |w|_2 = sqrt(sum (coef K)^2), which decides most evaluations again; (3) with the band forced wide open every evaluation
    goes through the three-pass kernel in list mode and meets that kernel's bar."""
    monkeypatch.setenv("HAF_NO_DIRECT", "1")                  # (a request of this size would go straight to tier 2's arithmetic)
    monkeypatch.setenv("HAF_NO_CALIBRATE", "1")               # the adaptive rule itself, from the plain variant on
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    f, r = _files(data_dir)
    path = str(tmp_path / "rand300.model")
    models.write_random_model(path, 300, seed=2, balanced=True)
    eng = make_engine(data_dir, path)
    compare_full(eng, O.Oracle(f, r, path), xyz, dict(n_rolls=12), inp)
    cnt = eng.last_counts()
    assert 0 < cnt["n_refined"] < 0.5 * cnt["n_evals"] and cnt["n_rechecked"] <= cnt["n_refined"]
    STATS["screen_refined_share_rand300"] = cnt["n_refined"] / max(1, cnt["n_evals"])
    eng.close()
    eng = make_engine(data_dir, surrogate)
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    cnt = eng.last_counts()
    assert cnt["n_refined"] > 0.25 * cnt["n_evals"]            # plain variant: the bound sqrt(max|coef| S) on |w|_2 is hopeless at C = 512
    STATS["screen_refined_share_surrogate_plain"] = cnt["n_refined"] / max(1, cnt["n_evals"])
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)          # the engine has switched to the variant that measures |w|_2
    cnt2 = eng.last_counts()
    # It helps, but not enough for a model of 172 SVs: the fp16 rounding of the operands alone bounds the error by
    # |u^-u| sigma(W^) |w|_2 ~ 0.5 for this model, and most of its decision values are smaller than that.  Round 3 served the
    # model with the three-pass kernel from the third call on; since round 4 the third call runs the centred-remainder form
    # (SCREEN_CR_EXP), whose band is relative to sum|b| psi(z) instead of sum|coef| K: under a third undecided, and the engine stays there.
    assert 0 < cnt2["n_refined"] < cnt["n_refined"], (cnt, cnt2)
    STATS["screen_refined_share_surrogate_sumsq"] = cnt2["n_refined"] / max(1, cnt2["n_evals"])
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    cnt3 = eng.last_counts()
    assert 0 < cnt3["n_refined"] < 0.4 * cnt3["n_evals"] and cnt3["n_refined"] < cnt2["n_refined"], (cnt2, cnt3)
    STATS["screen_refined_share_surrogate_cr_exp"] = cnt3["n_refined"] / max(1, cnt3["n_evals"])
    st = eng.screen_state()
    assert st["active"] and st["variant"] == 2, st
    compare_full(eng, orc, xyz, dict(n_rolls=12), inp)
    assert eng.last_counts() == cnt3 and eng.screen_state()["variant"] == 2
    eng.close()
    monkeypatch.setenv("HAF_GUARD0_REL", "1e30")
    eng = make_engine(data_dir, surrogate)
    want = orc.run(xyz, O.make_cfg(), oracle_input(inp))
    eng.score(xyz, capi.default_input(**inp))
    cnt = eng.last_counts()
    assert cnt["n_refined"] == cnt["n_evals"] == want["n_evals"]
    for roll in range(12):
        assert (eng.debug(capi.DBG_LABELS, 0, roll) == want["labels"][roll]).all()
        m = want["mask"][roll] == 1
        if m.any():
            err = np.abs(eng.debug(capi.DBG_DECISION, 0, roll)[m] - want["dec"][roll][m])
            assert (err <= DEC_REL * want["sabs"][roll][m] + DEC_ABS).all()
    eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_randomised_requests_against_oracle(data_dir, tmp_path, mode):
    """Seeded random requests: random clouds (blobs, planes, outliers, duplicates, points on cell edges), centres,
    approach vectors, gripper widths, search areas, roll counts/steps; stage-by-stage bit parity with the oracle."""
    f, r = _files(data_dir)
    path = str(tmp_path / "rand128.model")
    models.write_random_model(path, 128, seed=77, balanced=True)
    o = O.Oracle(f, r, path)
    rng = np.random.RandomState(2024)
    engines = {}
    for case in range(24):
        n_rolls, step = [(12, 15), (5, 36), (7, 25), (3, 60)][case % 4]
        n = int(rng.choice([50, 500, 4000, 20000]))
        pts = []
        for _ in range(rng.randint(1, 5)):                       # blobs
            c = rng.uniform(-0.15, 0.15, 3) * [1, 1, 0.3] + [0, 0, 0.05]
            pts.append(c + rng.standard_normal((n // 4 + 1, 3)) * rng.uniform(0.005, 0.05, 3))
        xy = rng.uniform(-0.3, 0.3, (n // 2, 2))                  # a tilted plane
        pts.append(np.column_stack([xy, 0.02 + 0.1 * xy[:, 0] - 0.05 * xy[:, 1]]))
        grid_pts = np.round(rng.uniform(-0.28, 0.28, (64, 3)), 2)  # exactly on centimetre edges
        grid_pts[:, 2] = rng.uniform(0, 0.1, 64)
        pts.append(grid_pts)
        xyz = np.concatenate(pts).astype(np.float32)
        xyz = np.concatenate([xyz, xyz[:17]])                     # duplicates
        kw = dict(grasp_area_center=tuple(rng.uniform(-0.05, 0.05, 3) * [1, 1, 0.2]),
                  grasp_area_length_x=float(rng.choice([20, 28.9, 32, 44, 56])),
                  grasp_area_length_y=float(rng.choice([18, 32, 44, 56, 70])),
                  gripper_opening_width=int(rng.choice([1, 1, 1, 2, 3])),
                  show_only_best_grasp=int(rng.rand() < 0.3))
        if rng.rand() < 0.5:
            kw["approach_vector"] = tuple(rng.standard_normal(3) * [0.3, 0.3, 1.0] + [0, 0, 1.0])
        key = (n_rolls, step)
        if key not in engines:
            engines[key] = make_engine(data_dir, path, mode, n_rolls=n_rolls, roll_step_deg=step, max_points=1 << 16)
        compare_full(engines[key], o, xyz, dict(n_rolls=n_rolls, roll_step_deg=step), kw)
    for e in engines.values():
        e.close()


def test_range_file_must_cover_non_constant_attributes(data_dir, surrogate, tmp_path):
    """svm-scale derives the range of an attribute missing from the range file from each roll's data (svm-scale.c:165-198);
    the engine refuses such a configuration instead of silently scaling differently."""
    f, r = _files(data_dir)
    lines = open(r).read().splitlines()
    short = tmp_path / "range_short"
    short.write_text("\n".join(l for l in lines if not l.startswith("17 ")) + "\n")
    with pytest.raises(capi.HafError) as ei:
        capi.Engine(f, str(short), surrogate)
    assert ei.value.code == capi.HAF_E_ARG and "attribute 17" in str(ei.value)


def test_screening_pass_alone_is_accurate_and_stable(data_dir, surrogate, orc, monkeypatch):
    """Regression test for the v_exp_f32 read-after-write hazard on gfx950: with the screening band forced
    to zero every evaluation keeps the single-pass decision value.  Its error must stay below 1e-5 * S (measured 2.3e-6;
    a consumer scheduled too close to its exp showed up as 1e-2 ... 1e-1 on some waves of some launches), on every one of
    several launches, and the launches must agree bit for bit."""
    monkeypatch.setenv("HAF_GUARD0_REL", "0")
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    inp = dict(grasp_area_length_x=32, grasp_area_length_y=32)
    want = orc.run(xyz, O.make_cfg(), oracle_input(inp))
    eng = make_engine(data_dir, surrogate)
    first = None
    for launch in range(6):
        eng.score(xyz, capi.default_input(**inp))
        cnt = eng.last_counts()
        assert cnt["n_refined"] == 0 and cnt["n_evals"] == want["n_evals"]
        decs = []
        for roll in range(12):
            m = want["mask"][roll] == 1
            d = eng.debug(capi.DBG_DECISION, 0, roll)
            if m.any():
                rel = np.abs(d[m] - want["dec"][roll][m]) / want["sabs"][roll][m]
                assert rel.max() < 1e-5, (launch, roll, float(rel.max()))
            decs.append(d[m])
        decs = np.concatenate(decs)
        if first is None:
            first = decs
        else:
            assert (decs.view(np.uint64) == first.view(np.uint64)).all(), launch
    eng.close()


def _bench_model(tmp_path, spec, trained_path):
    """(model file, screening form to pin or None) of a bench-size model: ("rand", nsv, seed) the generator of bench.py's headline,
    ("hard", form): the libsvm-trained surrogate x 24 jittered copies (bench.py's hard_model), ("trained", form): the 8964-SV model."""
    if spec[0] == "rand":
        path = str(tmp_path / ("rand%d_%d.model" % (spec[1], spec[2])))
        models.write_random_model(path, spec[1], seed=spec[2], balanced=True)
        return path, None
    if spec[0] == "hard":
        path = str(tmp_path / "hard.model")
        models.write_replicated_model(path, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surrogate.model"), copies=24,
                                      jitter=0.01, seed=5)
        return path, spec[1]
    return trained_path, spec[1]


@pytest.mark.parametrize("spec,modes", [(("rand", 512, 7), (capi.FLAG_SPLIT_F16, 0, capi.FLAG_FP32_MFMA)), (("rand", 4096, 7), (capi.FLAG_SPLIT_F16, 0)),
                                        (("rand", 4096, 11), (capi.FLAG_SPLIT_F16, 0)), (("hard", 1), (capi.FLAG_SPLIT_F16, 0)),
                                        (("hard", None), (capi.FLAG_SPLIT_F16, 0)), (("trained", None), (capi.FLAG_SPLIT_F16, 0))],
                         ids=["rand512-7", "rand4096-7", "rand4096-11", "hard-sumsq", "hard-auto", "trained-auto"])
def test_contraction_modes_agree_on_every_label_at_full_size(data_dir, tmp_path, trained_model, monkeypatch, spec, modes):
    """All three contraction modes claim libsvm's labels (each tier decides only outside a rigorous error band).  At BASELINE
    config C5 (7.9 M evaluations) the label grids of the screening mode, the three-pass mode and the fp32 mode must be
    identical cell for cell, for a model whose decision values crowd around zero -- a hole in a band would show up here as
    a handful of differing cells out of millions.  Round 4: the same for bench.py's hard_model
    through k_svm_screen<SUMSQ> (pinned) and through the form the engine picks by itself, and for the trained 8964-SV model
    (centred-remainder form; its three-pass reference run decides next to nothing in fp32 and walks the exact tiers window by window)."""
    path, form = _bench_model(tmp_path, spec, trained_model)
    xyz = models.synthetic_cloud(grid=512, k=2, seed=0)
    inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
    ref_labels, ref_rec, counts, forms = None, None, {}, {}
    for mode in modes:
        if form is not None and mode == 0:
            monkeypatch.setenv("HAF_SCREEN_VARIANT", str(form))
        eng = make_engine(data_dir, path, mode, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20)
        monkeypatch.delenv("HAF_SCREEN_VARIANT", raising=False)
        rec = eng.score_rolls([xyz], [inp], 0, 36)[0]
        counts[mode] = eng.last_counts()
        forms[mode] = eng.screen_form()
        labels = np.stack([eng.debug(capi.DBG_LABELS, 0, roll) for roll in range(36)])
        eng.close()
        if ref_labels is None:
            ref_labels, ref_rec = labels, rec
            if spec[0] == "rand":
                assert (labels == 1).sum() > 100000 and (labels == -1).sum() > 100000
        else:
            assert int((labels != ref_labels).sum()) == 0, (mode, int((labels != ref_labels).sum()))
            assert (rec == ref_rec).all(), mode
    assert 0 < counts[0]["n_refined"] < 0.2 * counts[0]["n_evals"]          # the screening pass was really in charge
    if form is not None:
        assert forms[0] == capi.Engine.SCREEN_FORMS[form]
    if spec[0] == "trained":
        assert forms[0] == "centred-remainder/poly" and counts[0]["n_refined"] < 0.02 * counts[0]["n_evals"]
    STATS["c5_%s_tiers" % "_".join(str(t) for t in spec)] = {"forms": {str(k): v for k, v in forms.items()}, "counts": {str(k): v for k, v in counts.items()}}


# ---------------------------------------------------------------------------------------------------------------------
# round 2: the attribute pipeline itself (not only labels): feature -> "%.4g" -> svm-scale -> "%g", bit for bit
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_attr_rows(orc, ii, cells):
    """(features fp32 [n, 324], q4 [n, 324], scaled [n, 323]) of the oracle for the given cells of one integral image."""
    skip = np.zeros(325, np.uint8)
    skip[324] = 1               # phantom attribute: constant -1 in every row -> dropped by svm-scale.c:336-337
    F, Q, S = [], [], []
    for i, j in cells:
        f = orc.feature_values(ii[i - 7:i + 8, j - 7:j + 8])
        q = np.array([O.q4(v) for v in f])
        F.append(f)
        Q.append(q)
        S.append(orc.scale_row(q, 323, skip))
    return np.array(F), np.array(Q), np.array(S)


def _bits32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


ATTR_CASES = [("pcd2", "C2"), ("pcd3", "C4"), ("pcd12", "default"), ("plastic_mug2", "default"), ("pcd2", "tilt"),
              ("table1_mult_obj_rcs_1428580506606673", "C3c")]


@pytest.mark.parametrize("mode", [pytest.param(capi.FLAG_SPLIT_F16, id="splitf16"), pytest.param(capi.FLAG_FP32_MFMA, id="f32mfma")])
@pytest.mark.parametrize("name,cname", ATTR_CASES)
def test_attribute_pipeline_bit_exact_against_oracle(data_dir, surrogate, orc, mode, name, cname):
    """HAF_DBG_ATTR: what the exact-form feature kernels computed for EVERY masked cell -- the fp32 feature
    (fv.cpp:141-199), its "%.4g" round trip (fv.cpp:133 -> svm-scale.c:270) and the scaled value after the "%g" round trip
    (svm-scale.c:333-353 -> svm-predict.c:108) -- against hafo_feature_values / hafo_q4 / hafo_scale_row, bit for bit.
    Small requests take k_features_small, larger ones k_features; the thread-per-evaluation kernel has its own test."""
    import make_fixtures as MF
    spec = MF.CONFIGS[cname]
    xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
    cfg_kw = dict(spec["cfg"])
    in_kw = dict(grasp_area_length_x=spec["inp"]["length_x"], grasp_area_length_y=spec["inp"]["length_y"])
    if "center" in spec["inp"]: in_kw["grasp_area_center"] = spec["inp"]["center"]
    if "approach" in spec["inp"]: in_kw["approach_vector"] = spec["inp"]["approach"]
    eng = make_engine(data_dir, surrogate, mode, **cfg_kw)
    eng.score(xyz, capi.default_input(**in_kw))
    want = orc.run(xyz, O.make_cfg(**cfg_kw), oracle_input(in_kw))
    total = 0
    for roll in range(cfg_kw["n_rolls"]):
        cells, attr, comp = eng.debug_attr(0, roll)
        mask_cells = np.stack(np.nonzero(want["mask"][roll]), 1)
        assert (cells == mask_cells).all() and comp.all()
        if roll % 3 and len(cells) > 64:          # every roll's cell list, every third roll's values (the oracle side is slow)
            continue
        F, Q, S = _oracle_attr_rows(orc, want["integral"][roll], cells)
        assert (_bits32(attr["feature"]) == _bits32(F)).all(), (roll, "feature")
        assert (_bits64(attr["q4"]) == _bits64(Q)).all(), (roll, "q4")
        got_s = attr["scaled"][:, :323]
        assert (_bits64(got_s + 0.0) == _bits64(S + 0.0)).all(), (roll, "scaled")      # (+0.0: -0 and +0 are the same text)
        assert (attr["scaled"][:, 323] == 0).all()             # attribute 324: dropped by svm-scale
        total += len(cells)
    assert total > 0 or want["n_evals"] == 0
    eng.close()


def test_attribute_pipeline_thread_per_evaluation_kernel_and_list_mode(data_dir, surrogate, orc, monkeypatch):
    """The same check for the other two exact-form routes: k_features_serial (HAF_LARGE_EVALS forces it) and the list mode
    behind the screening pass (default mode: only the cells the screening pass handed on are computed, and exactly those)."""
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd3.pcd"))
    in_kw = dict(grasp_area_length_x=32, grasp_area_length_y=44)
    want = orc.run(xyz, O.make_cfg(), oracle_input(in_kw))
    monkeypatch.setenv("HAF_LARGE_EVALS", "1")
    for mode in (capi.FLAG_SPLIT_F16, capi.FLAG_FP32_MFMA):
        eng = make_engine(data_dir, surrogate, mode)
        eng.score(xyz, capi.default_input(**in_kw))
        for roll in (0, 5, 11):
            cells, attr, comp = eng.debug_attr(0, roll)
            assert comp.all()
            F, Q, S = _oracle_attr_rows(orc, want["integral"][roll], cells)
            assert (_bits32(attr["feature"]) == _bits32(F)).all() and (_bits64(attr["q4"]) == _bits64(Q)).all()
            assert (_bits64(attr["scaled"][:, :323] + 0.0) == _bits64(S + 0.0)).all()
        eng.close()
    monkeypatch.delenv("HAF_LARGE_EVALS")
    monkeypatch.setenv("HAF_NO_DIRECT", "1")                    # (a request of this size would otherwise go straight to tier 2's arithmetic)
    monkeypatch.setenv("HAF_NO_CALIBRATE", "1")                 # (and the surrogate would be served without the screening pass)
    eng = make_engine(data_dir, surrogate, 0)                   # default mode: screening pass, three-pass kernel on its list
    eng.score(xyz, capi.default_input(**in_kw))
    n_comp = 0
    for roll in range(12):
        cells, attr, comp = eng.debug_attr(0, roll)
        n_comp += int(comp.sum())
        if comp.any() and roll in (0, 5, 11):
            F, Q, S = _oracle_attr_rows(orc, want["integral"][roll], cells[comp])
            assert (_bits32(attr["feature"][comp]) == _bits32(F)).all() and (_bits64(attr["q4"][comp]) == _bits64(Q)).all()
            assert (_bits64(attr["scaled"][comp][:, :323] + 0.0) == _bits64(S + 0.0)).all()
    assert n_comp == eng.last_counts()["n_refined"] > 0
    eng.close()


@pytest.mark.parametrize("name,roll", [("pcd2", 0), ("pcd2", 5), ("pcd3", 2), ("plastic_mug2", 7)])
def test_attribute_pipeline_against_reference_tool_fixtures(data_dir, golden_dir, surrogate, name, roll):
    """The same records against tests/golden/g23_*.npz: the q4 column is what the REAL svm-scale read from the feature
    text, `scaled` what the REAL svm-predict read from svm-scale's output (tests/golden/make_fixtures.py ran the reference's
    own binaries): for those two columns there is no oracle in between.  The `features` column of the fixture is the ORACLE's
    own fp32 output (make_fixtures.py:172-175 -- fv.cpp cannot be built here), so that comparison is engine against oracle."""
    g = np.load(os.path.join(golden_dir, "g23_%s_r%d.npz" % (name, roll)))
    xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
    eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16)
    eng.score(xyz, capi.default_input())                         # make_fixtures.py: default cfg, O.make_input() = 32 x 32
    cells, attr, comp = eng.debug_attr(0, roll)
    eng.close()
    index = {(int(i), int(j)): k for k, (i, j) in enumerate(cells)}
    rows = [index[(int(i), int(j))] for i, j in g["cells"]]
    assert len(rows) == len(g["cells"]) > 0
    a = attr[rows]
    assert (_bits32(a["feature"]) == _bits32(g["features"])).all()
    assert (_bits64(a["q4"]) == _bits64(g["q4"])).all()
    D = g["scaled"].shape[1]
    assert (_bits64(a["scaled"][:, :D] + 0.0) == _bits64(g["scaled"] + 0.0)).all()


# ---------------------------------------------------------------------------------------------------------------------
# round 2: several GPUs in ONE process behind the C-ABI (csrc/multi.cpp), RCCL collectives
# ---------------------------------------------------------------------------------------------------------------------
def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _multi_devices():
    """[(id, devices)]: two shards sharing the one GPU of the test box (one RCCL rank), and -- on a node with more GPUs --
    one shard per GPU."""
    out = [("1rank_1shard", [0]), ("1rank_2shards", [0, 0]), ("1rank_3shards", [0, 0, 0])]
    n = _n_gpus()
    if n >= 2:
        out.append(("%dranks" % min(n, 4), list(range(min(n, 4)))))
        out.append(("2ranks_2shards_each", [0, 1, 0, 1]))
    return out


@pytest.mark.parametrize("label,devices", _multi_devices() if os.environ.get("HAF_COLLECT_MULTI", "1") == "1" else [])
def test_roll_sharded_request_through_the_c_abi(data_dir, golden_dir, surrogate, label, devices):
    """haf_score_sharded: rolls of one request split over the shards, one ncclAllGather of the 16-byte roll records, the
    sequential cross-roll rule on the gathered set.  Same GraspOutput as the committed oracle goldens (incl. the
    show_only_best early exit, which is why the exchange is an all-gather), and every rank holds identical records."""
    import make_fixtures as MF
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        gold = json.load(f)
    f_, r_ = _files(data_dir)
    for name, cname in [("pcd2", "C2"), ("pcd2", "C2best"), ("plastic_mug2", "C2best"), ("pcd3", "C4"), ("pcd7", "C4")]:
        spec = MF.CONFIGS[cname]
        xyz = pcdio.load_pcd(os.path.join(data_dir, name + ".pcd"))
        me = capi.MultiEngine(f_, r_, surrogate, devices, capi.SHARD_ROLLS, **spec["cfg"])
        info = me.info()
        assert info["n_shards"] == len(devices) and info["n_ranks"] == len(set(devices)) and info["rccl_version"] > 0
        inp = capi.default_input(grasp_area_length_x=spec["inp"]["length_x"], grasp_area_length_y=spec["inp"]["length_y"],
                                 show_only_best_grasp=spec["inp"].get("show_only_best", 0))
        if len(devices) > spec["cfg"]["n_rolls"]:
            continue
        got = me.score_sharded(xyz, inp)
        w = gold["%s/%s" % (name, cname)]
        assert (got["eval"], got["best_row"], got["best_col"], got["best_roll"], got["rolls_done"]) == \
               (w["eval"], w["row"], w["col"], w["roll_idx"], w["rolls_done"]), (name, cname, label)
        assert np.allclose(got["grasp_point1"], w["gp1"], atol=1e-4) and np.allclose(got["grasp_point2"], w["gp2"], atol=1e-4)
        rec0 = me.last_records(0)
        for roll in range(w["rolls_done"]):                      # (rolls behind an early exit: the oracle never ran them)
            assert [int(rec0["row"][roll]), int(rec0["col"][roll]), int(rec0["vote"][roll])] == w["roll_best"][roll], (name, cname, roll)
            assert int(rec0["n_evals"][roll]) == w["masked"][roll]
        for rk in range(1, info["n_ranks"]):
            assert me.last_records(rk).tobytes() == rec0.tobytes()
        # device-resident cloud: lives on devices[0], reaches the other ranks by ncclBroadcast
        import torch
        with torch.cuda.device(devices[0]):
            d_xyz = torch.from_numpy(xyz).cuda(devices[0])
            got2 = me.score_sharded((d_xyz.data_ptr(), xyz.shape[0], 3), inp)
        got2["n_rechecked"] = got["n_rechecked"]      # (tier statistics differ once the engine has stopped screening this model)
        assert got2 == got
        me.close()


@pytest.mark.parametrize("label,devices", _multi_devices() if os.environ.get("HAF_COLLECT_MULTI", "1") == "1" else [])
def test_cloud_sharded_batch_through_the_c_abi(data_dir, golden_dir, surrogate, label, devices):
    """haf_score_batch_sharded (BASELINE config C4): pcd1..8, 20 rolls of 9 degrees, cloud b on shard b % n, ONE
    ncclAllReduce(max) of the packed best-grasp key.  Outputs = the committed goldens; the elected cloud = argmax by hand."""
    import make_fixtures as MF
    with open(os.path.join(golden_dir, "g6_end_to_end.json")) as f:
        gold = json.load(f)
    f_, r_ = _files(data_dir)
    spec = MF.CONFIGS["C4"]
    names = ["pcd%d" % k for k in range(1, 9)]                    # BASELINE config C4: pcd1..pcd8 (all eight have a committed C4 golden)
    clouds = [pcdio.load_pcd(os.path.join(data_dir, n + ".pcd")) for n in names]
    me = capi.MultiEngine(f_, r_, surrogate, devices, capi.SHARD_CLOUDS, max_clouds=8, **spec["cfg"])
    inp = capi.default_input(grasp_area_length_x=32, grasp_area_length_y=44)
    outs, best = me.score_batch_sharded(clouds, [inp] * len(clouds))
    for n, o in zip(names, outs):
        w = gold[n + "/C4"]
        assert (o["eval"], o["best_row"], o["best_col"], o["best_roll"], o["n_evals"]) == (w["eval"], w["row"], w["col"], w["roll_idx"], w["n_evals"]), n
    votes = [o["best_vote"] for o in outs]
    assert best == int(np.argmax(votes))                       # argmax = first maximum = lowest cloud index on ties
    me.close()


def test_multi_rejects_bad_requests(data_dir, surrogate):
    f_, r_ = _files(data_dir)
    with pytest.raises(capi.HafError):
        capi.MultiEngine(f_, r_, surrogate, [0, 0, 1] if _n_gpus() >= 2 else [0] * 13, capi.SHARD_ROLLS)   # uneven shards / more shards than rolls
    me = capi.MultiEngine(f_, r_, surrogate, [0], capi.SHARD_CLOUDS, max_clouds=2)
    with pytest.raises(capi.HafError):
        me.score_sharded(np.zeros((4, 3), np.float32), capi.default_input())      # wrong mode
    me.close()


@pytest.mark.parametrize("spec", [("rand", 4096, 1234), ("rand", 4096, 11), ("hard", 1), ("hard", None), ("trained", None)],
                         ids=["rand4096-1234", "rand4096-11", "hard-sumsq", "hard-auto", "trained-auto"])
def test_bench_configuration_against_the_oracle_where_screening_was_closest(data_dir, tmp_path, trained_model, monkeypatch, spec):
    """The bench's own workload -- C5 (512 x 512, 36 rolls of 5 degrees, 524 288 points), seeded random model nSV = 4096
    (seeds 1234 -- round 2's headline, one-signed -- and 11 -- the hardest of bench.py's five: 37 % positive labels, decision
    values crowding around zero), default mode -- against the oracle's feature / scale / decision chain
    (hafo_feature_values, hafo_q4, hafo_scale_row, hafo_decision: libsvm's fp64 order) on >= 2 000 cells chosen where a
    band hole would show first: the cells the screening tier decided with |dec^| / band closest to 1 (HAF_DBG_SCREEN_MARGIN),
    plus the cells it handed on with the smallest |dec|, plus random ones.  Label identical; decision value inside the
    tier's own band (which the margin makes checkable: |dec^ - dec| < |dec^| / margin).  Round 4: the same for bench.py's hard_model
    through k_svm_screen<SUMSQ> (svm.cpp:2509-2519 on 7.9 M evaluations) and through the form the engine picks,
    and for the trained 8964-SV model through the centred-remainder form."""
    path, form = _bench_model(tmp_path, spec, trained_model)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, path)
    xyz = models.synthetic_cloud(grid=512, k=2, seed=0)
    if form is not None:
        monkeypatch.setenv("HAF_SCREEN_VARIANT", str(form))
    eng = make_engine(data_dir, path, 0, grid_h=512, grid_w=512, n_rolls=36, roll_step_deg=5, max_points=1 << 20)
    monkeypatch.delenv("HAF_SCREEN_VARIANT", raising=False)
    assert not eng.cfg.flags & (capi.FLAG_SPLIT_F16 | capi.FLAG_FP32_MFMA)
    inp = capi.default_input(grasp_area_length_x=512, grasp_area_length_y=512)
    rec = eng.score_rolls([xyz], [inp], 0, 36)[0]
    cnt = eng.last_counts()
    assert cnt["n_evals"] == int(rec["n_evals"].sum()) > 7800000 and 0 < cnt["n_refined"] < 0.1 * cnt["n_evals"]
    # gather (roll, i, j, margin, dec) of every masked cell; keep the closest calls
    close, handed, rnd = [], [], []
    rng = np.random.RandomState(3)
    n_decided = 0
    for roll in range(36):
        mg = eng.debug(capi.DBG_SCREEN_MARGIN, 0, roll)
        dec = eng.debug(capi.DBG_DECISION, 0, roll)
        ok = ~np.isnan(mg)
        assert int(ok.sum()) == int(rec["n_evals"][roll])
        decided = ok & (mg > 0)
        assert (mg[decided] > 1.0).all()                       # the tier only decides outside its band
        n_decided += int(decided.sum())
        ii, jj = np.nonzero(decided)
        order = np.argsort(mg[ii, jj])[:120]
        close += [(roll, int(ii[k]), int(jj[k]), float(mg[ii[k], jj[k]])) for k in order]
        hi, hj = np.nonzero(ok & (mg == 0))
        if len(hi):
            by_abs = np.argsort(np.abs(dec[hi, hj]))
            # the ten closest to zero (decided by the exact tiers), and five each around the 12th, 16th and 25th percentile of |dec|
            # among the handed-on cells: where the three-pass tier's band ends (it passes on ~15 %), its closest calls
            order = list(by_abs[:10])
            for q in (0.12, 0.16, 0.25):
                k0 = int(q * len(by_abs))
                order += list(by_abs[k0:k0 + 5])
            handed += [(roll, int(hi[k]), int(hj[k]), 0.0) for k in dict.fromkeys(order)]
        pick = rng.choice(len(ii), 8, replace=False)
        rnd += [(roll, int(ii[k]), int(jj[k]), float(mg[ii[k], jj[k]])) for k in pick]
    assert n_decided == cnt["n_evals"] - cnt["n_refined"]
    close.sort(key=lambda t: t[3])
    sample = close[:1700] + handed + rnd
    assert len(sample) >= 2000
    m = o.model_arrays()
    skip = np.zeros(325, np.uint8)
    skip[324] = 1
    by_roll = {}
    for s in sample:
        by_roll.setdefault(s[0], []).append(s)
    worst = 0.0
    sv2 = (m["sv"] * m["sv"]).sum(1)
    for roll, items in by_roll.items():
        ii = eng.debug(capi.DBG_INTEGRAL, 0, roll)
        lab = eng.debug(capi.DBG_LABELS, 0, roll)
        dec = eng.debug(capi.DBG_DECISION, 0, roll)
        for _, i, j, mgv in items:
            feats = o.feature_values(ii[i - 7:i + 8, j - 7:j + 8])
            xs = o.scale_row(np.array([O.q4(v) for v in feats]), m["D"], skip)
            d = o.decision(xs)
            want = m["label"][0] if d > 0 else m["label"][1]
            assert lab[i, j] == want, (roll, i, j, d, dec[i, j], mgv)
            if mgv > 0:                                        # decided by the screening tier: its value, inside its band
                assert abs(dec[i, j] - d) < abs(dec[i, j]) / mgv, (roll, i, j, d, dec[i, j], mgv)
                worst = max(worst, abs(dec[i, j] - d) * mgv / abs(dec[i, j]))
            else:                                              # handed on: the value of the tier that decided it
                S = float(np.abs(m["coef"]) @ np.exp(-m["gamma"] * (sv2 - 2.0 * (m["sv"] @ xs) + xs @ xs)))
                assert abs(dec[i, j] - d) <= 1.5e-6 * S + 1e-9, (roll, i, j, d, dec[i, j], S)     # three-pass tier: 2^-20 S + 1e-6, the exact tiers far inside
    STATS["bench_config_oracle_check_%s" % "_".join(str(t) for t in spec)] = {"cells": len(sample), "closest_margin": close[0][3],
                                                        "worst_error_as_fraction_of_band": worst,
                                                        "exact_tiers": eng.last_exact_tiers(), "tiers": cnt, "form": eng.screen_form()}
    eng.close()


@pytest.mark.parametrize("spec,roll", [(("rand", 4096, 42), 0), (("trained", None), 7)], ids=["rand4096-42-roll0", "trained-roll7"])
def test_one_complete_roll_of_the_bench_workload_against_the_oracle(data_dir, tmp_path, trained_model, spec, roll):
    """parity at the bench's OWN size was a sample of ~2 900 cells per model plus cross-mode equality.  Here one
    COMPLETE roll of the C5 request bench.py times -- 512 x 512 grid, 36 rolls of 5 degrees, 524 288 points; the median-seed random
    model (nSV 4096, seed 42) roll 0, and the trained 8964-SV model on a ROLLED grid (roll 7 = 35 degrees) -- is compared with the
    oracle cell for cell: height grid, integral image and mask bit for bit, the label of every one of its ~248 000 evaluations
    (libsvm's fp64 order: svm.cpp:2478-2532), the 29-tap vote grid and the roll record (server.cpp:803-973).  The engine runs the whole
    36-roll request in its default mode, product library, exactly as the bench does (low-rank form of the screening pass + the tiers
    behind it); the oracle scores the one roll on all host cores (HAFO_THREADS; rows are independent, arithmetic untouched)."""
    path, _ = _bench_model(tmp_path, spec, trained_model)
    f, r = _files(data_dir)
    G = 512
    xyz = models.synthetic_cloud(grid=G, k=2, seed=0)
    eng = make_engine(data_dir, path, 0, grid_h=G, grid_w=G, n_rolls=36, roll_step_deg=5, max_points=1 << 20)
    inp = capi.default_input(grasp_area_length_x=G, grasp_area_length_y=G)
    rec = eng.score_rolls([xyz], [inp], 0, 36)[0]
    cnt = eng.last_counts()
    assert eng.screen_low_rank()["last_used"] and 0 < cnt["n_refined"] < 0.1 * cnt["n_evals"]     # the bench's path: low-rank screening + tiers
    got = dict(heights=eng.debug(capi.DBG_HEIGHTS, 0, roll), integral=eng.debug(capi.DBG_INTEGRAL, 0, roll), mask=eng.debug(capi.DBG_MASK, 0, roll),
               labels=eng.debug(capi.DBG_LABELS, 0, roll), vote=eng.roll_grid(0, roll)[0])
    form = eng.screen_form()
    eng.close()
    cores = len(os.sched_getaffinity(0))
    old = os.environ.get("HAFO_THREADS")
    os.environ["HAFO_THREADS"] = str(cores)
    import time
    t0 = time.perf_counter()
    try:
        o = O.Oracle(f, r, path)
        want = o.run(xyz, O.make_cfg(H=G, W=G, n_rolls=roll + 1, roll_step_deg=5), O.make_input(length_x=G, length_y=G), roll_first=roll)
    finally:
        if old is None:
            os.environ.pop("HAFO_THREADS", None)
        else:
            os.environ["HAFO_THREADS"] = old
    dt = time.perf_counter() - t0
    assert want["rolls_done"] == 1 and want["n_evals"] == int(rec["n_evals"][roll]) > 200000
    assert (got["heights"].view(np.uint32) == want["heights"][roll].view(np.uint32)).all()
    assert (got["integral"].view(np.uint32) == want["integral"][roll].view(np.uint32)).all()
    assert (got["mask"] == want["mask"][roll]).all()
    n_diff = int((got["labels"] != want["labels"][roll]).sum())
    assert n_diff == 0, ("labels differ from libsvm's", n_diff)
    assert (got["vote"] == want["graspseval"][roll]).all()
    br, bc, bv = (int(v) for v in want["roll_best"][roll])
    assert (int(rec["row"][roll]), int(rec["col"][roll]), int(rec["vote"][roll])) == (br, bc, bv)
    lab = want["labels"][roll][want["mask"][roll] == 1]
    STATS["full_roll_%s_r%d" % ("_".join(str(t) for t in spec), roll)] = {
        "evaluations": int(want["n_evals"]), "oracle_seconds": dt, "oracle_cores": cores, "labels_differing": n_diff, "form": form,
        "positive_labels": int((lab == lab.max()).sum()), "record": [br, bc, bv]}


def test_integral_image_falls_back_to_the_sequential_order_when_sums_are_inexact(data_dir, surrogate, orc):
    """Beyond ~70 x 70 cells the integral image is built by parallel scans whose every fp64 addition is checked for exactness;
    exact sums are order independent, so the result equals cv::integral's sequential order bit for bit (calc_intimage,
    server.cpp:577-613).  Heights that differ by more than 2^29 in magnitude make partial sums inexact: those grids must be
    flagged and redone in the sequential order -- and still match the oracle bit for bit.  Ordinary clouds never take the
    fallback; small grids (the reference's 56 x 56) are summed sequentially in LDS in the first place."""
    rng = np.random.RandomState(21)
    for grid in (96, 56):
        xyz = models.synthetic_cloud(grid=grid, k=2, seed=3)
        tiny = (-0.15 + rng.randint(1, 4000, size=len(xyz)) * 2.0 ** -26).astype(np.float32)   # heights of a few 2^-26 .. 2^-14
        xyz[:, 2] = tiny
        big = rng.choice(len(xyz), 40, replace=False)
        xyz[big, 2] = (1.0e9 * (1.0 + rng.uniform(0, 1, size=40))).astype(np.float32)           # and some of ~2^30
        inp = dict(grasp_area_length_x=grid, grasp_area_length_y=grid)
        eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16, grid_h=grid, grid_w=grid, n_rolls=6, roll_step_deg=30)
        cfg = dict(n_rolls=6, roll_step_deg=30, grid_h=grid, grid_w=grid)
        compare_full(eng, orc, xyz, cfg, inp, check_dec=False)
        assert (eng.last_prestage()["n_inexact_grids"] > 0) == (grid == 96)
        ordinary = models.synthetic_cloud(grid=grid, k=2, seed=4)
        compare_full(eng, orc, ordinary, cfg, inp)
        assert eng.last_prestage()["n_inexact_grids"] == 0
        eng.close()


@pytest.mark.parametrize("in_kw", [dict(), dict(approach_vector=(0.2, -0.1, 1.0)), dict(gripper_opening_width=2),
                                   dict(grasp_area_center=(0.11, -0.07, 0.02), approach_vector=(-0.3, 0.2, 0.9), gripper_opening_width=3)],
                         ids=["plain", "tilted", "width2", "shifted_tilted_width3"])
def test_bucket_sorted_binning_of_large_grids(data_dir, surrogate, orc, in_kw):
    """Grids beyond LDS size with a sizeable cloud are binned without global atomics: the cloud is counting-sorted into spatial
    buckets once per request, then every (roll, 64 x 64 tile) gathers the buckets that can reach it (generate_grid,
    server.cpp:406-529).  Height grids must equal the oracle's bit for bit for every roll -- with a tilted approach vector
    (the pre-roll position then depends on z), an x-scale and a shifted centre (points outside the grid) as well."""
    grid = 192
    xyz = models.synthetic_cloud(grid=grid, k=2, seed=17)                      # 73 728 points: the bucket path (>= 32 768)
    xyz[::7, 2] += 0.3                                                         # some relief, so that tilts move points across cells
    kw = dict(grasp_area_length_x=grid, grasp_area_length_y=grid)
    kw.update(in_kw)
    rolls, step = (9, 20) if not in_kw else (4, 40)              # (the checker's feature text round trips are the wall time here)
    eng = make_engine(data_dir, surrogate, capi.FLAG_SPLIT_F16, grid_h=grid, grid_w=grid, n_rolls=rolls, roll_step_deg=step, max_points=1 << 17)
    compare_full(eng, orc, xyz, dict(n_rolls=rolls, roll_step_deg=step, grid_h=grid, grid_w=grid), kw, check_dec=False)
    eng.close()


def test_matrix_core_accumulation_stays_inside_the_band_assumption(data_dir, surrogate):
    """The one assumption about undocumented hardware behaviour in the bands of the fp16 tiers: ONE
    v_mfma_f32_16x16x32_f16 deviates from the exact c + sum a_k b_k by at most kappa 2^-24 (|c| + sum |a_k b_k|), so a chain of ten
    that starts from C by at most 10 kappa 2^-24 (|C| + sum |a_k b_k|).  The products are exact in fp32; how the matrix core adds
    them is not documented.  Round 3: kappa is MEASURED at haf_create on the device the engine runs on (screen.hip:
    probe_mfma_rounding, adversarial families) and used with a margin, kappa = max(12, 1.5 x measured) since round 4.  Evidence, not proof:
    (1) what the engine measured here (5.5 on the devices seen so far) and what it uses; (2) the chain of k_svm_screen (same
    builtin, same operand layout, same start-value mechanism) on 2 048 trials x 256 outputs of operands made to hurt --
    magnitudes over the whole fp16 range the operands can take, signs arranged for near-total cancellation, start values like
    t_n -- against an fp64 evaluation.  The worst observed error is reported as a fraction of the budget."""
    L = capi.testlib()
    import ctypes as C
    eng = make_engine(data_dir, surrogate, 0, testing=True)
    meas, used = (C.c_double * 2)(), (C.c_double * 2)()          # [0]: 16x16x32 (the screening kernel's shape), [1]: 16x16x16 (K tail of tier 1)
    L.haf_test_mfma_kappa.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.haf_test_mfma_kappa(eng._h, meas, used) == 0
    eng.close()
    for k in range(2):
        assert 0.5 <= meas[k] <= 12.0 and used[k] == max(12.0, 1.5 * meas[k]), (k, meas[k], used[k])
    budget = 10.0 * used[0] * 2.0 ** -24
    STATS["mfma_rounding_kappa"] = {"measured_32": meas[0], "used_32": used[0], "measured_16": meas[1], "used_16": used[1]}
    rng = np.random.RandomState(77)
    trials = 2048
    a = np.zeros((trials, 16, 320), np.float16)
    b = np.zeros((trials, 320, 16), np.float16)
    c0 = np.zeros((trials, 16), np.float32)
    for t in range(trials):
        kind = t % 4
        if kind == 0:                       # what the engine feeds: |u|, |w| ~ 0.05, start -0.5 .. 0
            a[t] = rng.uniform(-0.1, 0.1, (16, 320)); b[t] = rng.uniform(-0.1, 0.1, (320, 16)); c0[t] = rng.uniform(-0.6, 0, 16)
        elif kind == 1:                     # wide magnitudes: 2^-14 .. 2^3
            a[t] = rng.choice([-1, 1], (16, 320)) * 2.0 ** rng.uniform(-14, 3, (16, 320))
            b[t] = rng.choice([-1, 1], (320, 16)) * 2.0 ** rng.uniform(-14, 3, (320, 16)); c0[t] = rng.uniform(-20, 20, 16)
        elif kind == 2:                     # cancellation: pairs of equal products with opposite signs, then a small rest
            x = rng.uniform(0.5, 4.0, (16, 160)); y = rng.uniform(0.5, 4.0, (160, 16))
            a[t, :, 0::2] = x; a[t, :, 1::2] = -x; b[t, 0::2, :] = y; b[t, 1::2, :] = y
            a[t, :, :8] = rng.uniform(-1e-3, 1e-3, (16, 8)); c0[t] = rng.uniform(-1e-3, 1e-3, 16)
        else:                               # one huge term early, many tiny ones after it
            a[t] = rng.uniform(-2.0 ** -10, 2.0 ** -10, (16, 320)); b[t] = rng.uniform(-1, 1, (320, 16))
            a[t, :, 0] = 60.0; b[t, 0, :] = 60.0; c0[t] = -3600.0 + rng.uniform(-1, 1, 16)
    out = np.zeros((trials, 16, 16), np.float32)
    assert L.haf_test_mfma_accum(a.ctypes.data, b.ctypes.data, c0.ctypes.data, out.ctypes.data, trials) == 0
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    exact = np.einsum("trk,tkc->trc", a64, b64) + c0[:, None, :].astype(np.float64)
    scale = np.einsum("trk,tkc->trc", np.abs(a64), np.abs(b64)) + np.abs(c0[:, None, :].astype(np.float64))
    ratio = np.abs(out.astype(np.float64) - exact) / (budget * scale)
    assert np.isfinite(out).all() and ratio.max() <= 1.0, float(ratio.max())
    STATS["mfma_accumulation_error_as_fraction_of_the_10_kappa_u_budget"] = {"max": float(ratio.max()),
                                                                          "by_kind": [float(ratio[k::4].max()) for k in range(4)]}


def test_matrix_core_rounding_adversarial_search(data_dir, surrogate):
    """the probe at haf_create uses seven fixed families.  Here ONE v_mfma_f32_16x16x32_f16 is attacked by a seeded search
    instead: random exponent spreads (every product's magnitude drawn from 2^-24 .. 2^4 independently), multi-way cancellation (3 to
    8 large products that sum to nearly nothing over many small ones, the accumulator among them or not), ladders, and hill
    climbing from the worst cases found (perturb exponents and signs, keep what raises the ratio).  The largest
    |d - exact| / (2^-24 (|c| + sum|a_k b_k|)) over ~2 M sums must stay below the kappa the bands use and within 15 % of what the
    probe itself measures.  (Round 4: the first version of this search found 6.4, then -- with the "dense" family: one product of
    order 1 over 31 products with 22-bit mantissas -- 9.0, where the probe's seven families reached 5.5 and the bands used 8.2: the
    probe got that family (it measures 7.6 .. 9 with its 64 trials) and the bands use max(12, 1.5 x that) since.  Two guard bits below the largest exponent allow 33 x 0.25 + 1.)"""
    L = capi.testlib()
    eng = make_engine(data_dir, surrogate, 0, testing=True)
    meas, used = (C.c_double * 2)(), (C.c_double * 2)()
    assert L.haf_test_mfma_kappa(eng._h, meas, used) == 0
    eng.close()
    rng = np.random.RandomState(2026)
    u = 2.0 ** -24

    def run(a, b, c):
        T = a.shape[0]
        a16, b16 = np.ascontiguousarray(a.astype(np.float16)), np.ascontiguousarray(b.astype(np.float16))
        c32 = np.ascontiguousarray(c.astype(np.float32))
        d = np.zeros((T, 16, 16), np.float32)
        assert L.haf_test_f16_mfma(a16.ctypes.data, b16.ctypes.data, c32.ctypes.data, d.ctypes.data, T, 1) == 0
        A, B, Cc = a16.astype(np.float64), b16.astype(np.float64), c32.astype(np.float64)
        prod = np.transpose(A[:, :, :, None] * B[:, None, :, :], (0, 1, 3, 2))          # [T][16][16][32], exact in fp64
        terms = np.concatenate([prod, Cc[:, :, :, None]], axis=3).astype(np.longdouble)
        order = np.argsort(np.abs(terms), axis=3)
        s = np.take_along_axis(terms, order, axis=3).sum(axis=3)                       # smallest first, long double
        scale = np.abs(terms).sum(axis=3).astype(np.float64)
        ratio = np.abs(d.astype(np.longdouble) - s).astype(np.float64) / (u * np.maximum(scale, 1e-300))
        assert np.isfinite(d).all()
        return ratio                                                                   # [T][16][16]

    def family(kind, T):
        a = np.ones((T, 16, 32)); b = np.zeros((T, 32, 16)); c = np.zeros((T, 16, 16))
        sg = lambda *sh: rng.choice([-1.0, 1.0], sh)                                   # noqa: E731
        if kind == "spread":                       # every product its own exponent
            a = sg(T, 16, 32) * 2.0 ** rng.uniform(-12, 2, (T, 16, 32)); b = sg(T, 32, 16) * 2.0 ** rng.uniform(-12, 2, (T, 32, 16))
            c = sg(T, 16, 16) * 2.0 ** rng.uniform(-24, 4, (T, 16, 16))
        elif kind == "cancel":                     # k large products (and maybe c) that nearly cancel, the rest small
            b = (0.5 + 0.5 * rng.uniform(size=(T, 32, 16))) * 2.0 ** rng.uniform(-16, -9, (T, 32, 1))
            for t in range(T):
                k = rng.randint(3, 9)
                idx = rng.choice(32, k, replace=False)
                big = 2.0 ** rng.randint(0, 11) * (1.0 + rng.randint(0, 1024, k) / 1024.0)
                sgn = rng.choice([-1.0, 1.0], k)
                b[t, idx, :] = (sgn * big)[:, None]
                if rng.uniform() < 0.5:
                    c[t] = -float((sgn * big).sum()) * (1.0 + rng.uniform(-1e-3, 1e-3, (16, 16)))
        elif kind == "ladder":
            b = sg(T, 32, 16) * 2.0 ** (-np.arange(32)[None, :, None] * rng.uniform(0.3, 1.0, (T, 1, 1))) * (1.0 + rng.randint(0, 1024, (T, 32, 16)) / 1024.0)
            c = sg(T, 16, 16) * 2.0 ** rng.uniform(-30, 2, (T, 16, 16))
        elif kind == "cheavy":                     # accumulator far above or far below the products
            b = sg(T, 32, 16) * (1.0 + rng.uniform(size=(T, 32, 16))) * 2.0 ** rng.uniform(-20, 0, (T, 1, 1))
            c = sg(T, 16, 16) * (1.0 + rng.uniform(size=(T, 16, 16))) * 2.0 ** rng.uniform(-10, 12, (T, 1, 1))
        else:                                      # "dense": one product of order 1, 31 products 2^-14 .. 2^-30 of it with 22-bit mantissas
            e = rng.randint(-30, -13, (T, 1, 1))
            a = (1.5 + rng.randint(0, 512, (T, 16, 32)) / 1024.0) * 2.0 ** (e // 2)
            b = rng.choice([-1.0, 1.0], (T, 1, 1)) * (1.5 + rng.randint(0, 512, (T, 32, 16)) / 1024.0) * 2.0 ** (e - e // 2)
            a[:, :, 0] = 1.0
            b[:, 0, :] = 1.0 + rng.randint(0, 1024, (T, 16)) / 1024.0
        return a, b, c

    worst, by_kind, n_sums = 0.0, {}, 0
    pool = []
    for kind in ("spread", "cancel", "ladder", "cheavy", "dense"):
        a, b, c = family(kind, 768)
        r = run(a, b, c)
        n_sums += r.size
        by_kind[kind] = float(r.max())
        top = np.argsort(r.reshape(len(r), -1).max(1))[-32:]
        pool += [(a[t], b[t], c[t]) for t in top]
    # hill climbing from the worst trials of every family
    for _ in range(6):
        a = np.stack([q[0] for q in pool]); b = np.stack([q[1] for q in pool]); c = np.stack([q[2] for q in pool])
        reps = 6
        a, b, c = np.repeat(a, reps, 0), np.repeat(b, reps, 0), np.repeat(c, reps, 0)
        T = a.shape[0]
        flip = rng.uniform(size=b.shape) < 0.05
        b = np.where(flip, -b, b) * 2.0 ** np.where(rng.uniform(size=b.shape) < 0.1, rng.randint(-3, 4, b.shape), 0)
        c = c * 2.0 ** np.where(rng.uniform(size=c.shape) < 0.2, rng.randint(-2, 3, c.shape), 0) * (1.0 + rng.uniform(-1e-3, 1e-3, c.shape))
        b = np.clip(b, -60000.0, 60000.0)
        r = run(a, b, c)
        n_sums += r.size
        rt = r.reshape(T, -1).max(1)
        keep = np.argsort(rt)[-len(pool):]
        pool = [(a[t], b[t], c[t]) for t in keep]
        by_kind["climb"] = max(by_kind.get("climb", 0.0), float(rt.max()))
    worst = max(by_kind.values())
    STATS["mfma_rounding_adversarial_search"] = {"sums": int(n_sums), "worst_ratio": worst, "by_kind": by_kind, "kappa_used": used[0], "probe_measured": meas[0]}
    assert worst < 0.85 * used[0] and worst <= 1.3 * meas[0], (worst, used[0], meas[0], by_kind)


# ---- probability-output mode (HAF_FLAG_PROBABILITY) -----------------------------------------------------

def _prob_model(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "surrogate_prob.json")) as f:
        pj = json.load(f)
    return models.write_probability_model(str(tmp_path / "surrogate_prob.model"), os.path.join(golden_dir, "surrogate.model"),
                                          pj["probA"], pj["probB"])


def compare_probability(eng, orc, xyz, cfg_kw, in_kw):
    """Probability mode against the oracle: the grids of the common stages bit for bit, the labels svm_predict_probability returns,
    the two "%g" probabilities of every cell, the shifted res*prob grid, the fp32 vote grid, per-roll winners, the grasp."""
    ocfg = O.make_cfg(**{k: v for k, v in cfg_kw.items() if k in ("n_rolls", "roll_step_deg")},
                      H=cfg_kw.get("grid_h", 56), W=cfg_kw.get("grid_w", 56), probability=1)
    want = orc.run(xyz, ocfg, oracle_input(in_kw))
    got = eng.score(xyz, capi.default_input(**in_kw))
    for roll in range(want["rolls_done"]):
        assert (eng.debug(capi.DBG_HEIGHTS, 0, roll).view(np.uint32) == want["heights"][roll].view(np.uint32)).all()
        assert (eng.debug(capi.DBG_MASK, 0, roll) == want["mask"][roll]).all()
        msk = want["mask"][roll] == 1
        d = eng.debug(capi.DBG_DECISION, 0, roll)
        assert np.isnan(d[~msk]).all()
        if msk.any():                                    # strict tier: libsvm's order; device exp vs glibc exp is the residual
            assert (np.abs(d[msk] - want["dec"][roll][msk]) <= 1e-12 * want["sabs"][roll][msk] + 1e-300).all()
        assert (eng.debug(capi.DBG_LABELS, 0, roll) == want["labels"][roll]).all(), ("labels", roll)
        p = eng.debug(capi.DBG_PROBABILITY, 0, roll)
        assert np.isnan(p[~msk]).all()
        assert (p[msk] == want["prob"][roll][msk]).all(), ("probabilities", roll, int((p[msk] != want["prob"][roll][msk]).sum()))
        g = eng.debug(capi.DBG_GRASPSGRID, 0, roll)
        assert (g.view(np.uint32) == want["graspsgrid"][roll].view(np.uint32)).all(), ("graspsgrid", roll)
        ev, _ = eng.roll_grid(0, roll)
        assert (ev.view(np.uint32) == want["graspseval"][roll].view(np.uint32)).all(), ("vote grid", roll)
    for k_e, k_o in [("eval", "eval"), ("best_row", "row"), ("best_col", "col"), ("best_roll", "roll_idx"),
                     ("best_vote", "top"), ("rolls_done", "rolls_done"), ("n_evals", "n_evals")]:
        assert got[k_e] == want[k_o], (k_e, got[k_e], want[k_o])
    np.testing.assert_allclose(got["grasp_point1"], want["gp1"], atol=1e-4)
    np.testing.assert_allclose(got["grasp_point2"], want["gp2"], atol=1e-4)
    assert got["roll"] == want["roll"]
    return got, want


def test_probability_mode_against_oracle(data_dir, golden_dir, tmp_path):
    """svm_with_probability end to end (dead in the reference, server.cpp:383; restated because SURVEY.md §8 lists it): pcd2 at
    C2, a table cloud at C3, a 96 x 96 synthetic grid, and requests with tilted approach vectors."""
    mp = _prob_model(golden_dir, tmp_path)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, mp)
    eng = make_engine(data_dir, mp, capi.FLAG_PROBABILITY)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    got, want = compare_probability(eng, o, xyz, {}, dict(grasp_area_length_x=32, grasp_area_length_y=32))
    assert want["n_evals"] > 3000 and (want["labels"] > 0).any() and (want["graspsgrid"] > 0).any()
    compare_probability(eng, o, xyz, {}, dict(grasp_area_length_x=32, grasp_area_length_y=44, approach_vector=(0.2, -0.1, 1.0)))
    compare_probability(eng, o, pcdio.load_pcd(os.path.join(data_dir, "pcd12.pcd")), {}, dict(show_only_best_grasp=1))
    compare_probability(eng, o, np.zeros((0, 3), np.float32), {}, {})                  # nothing masked: votes all 0
    eng.close()
    eng = make_engine(data_dir, mp, capi.FLAG_PROBABILITY, n_rolls=20, roll_step_deg=9, max_points=1 << 17)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "table1_mult_obj_rcs_1428580506606673.pcd"))
    compare_probability(eng, o, xyz, dict(n_rolls=20, roll_step_deg=9),
                        dict(grasp_area_length_x=56, grasp_area_length_y=56, grasp_area_center=(0.13, 0.25, 0.0)))
    eng.close()
    eng = make_engine(data_dir, mp, capi.FLAG_PROBABILITY, grid_h=96, grid_w=96, n_rolls=5, roll_step_deg=36)
    compare_probability(eng, o, models.synthetic_cloud(grid=96, k=2, seed=11), dict(n_rolls=5, roll_step_deg=36, grid_h=96, grid_w=96),
                        dict(grasp_area_length_x=96, grasp_area_length_y=70))
    eng.close()


def test_probability_mode_estimates_finished_on_the_host(data_dir, golden_dir, tmp_path, monkeypatch):
    """the device's exp is not glibc's, so an estimate within a last-bit exp difference of a six-digit "%g" boundary
    could print differently from svm-predict -b 1 (svm.cpp:1818-1826, svm-predict.c:111-118).  k_prob_eval brackets every estimate
    (decision value and exp result pushed to either side by more than the two libraries can differ) and hands the ones whose label or
    printed digits are not the same on both sides to the host, which finishes them with the C library's exp.  Forced for EVERY
    evaluation here (HAF_PROB_HOST_ALL): labels, both "%g" probabilities, grid, votes and grasp must be the oracle's bit for bit --
    whose probabilities are pinned to the REAL svm-predict -b 1 character for character (g23p fixtures) -- and in the normal run
    next to none take that path."""
    model = _prob_model(golden_dir, tmp_path)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, model)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    kw = dict(grasp_area_length_x=32, grasp_area_length_y=32)
    eng = make_engine(data_dir, model, capi.FLAG_PROBABILITY)
    got, want = compare_probability(eng, o, xyz, dict(n_rolls=12), kw)
    normal = eng.last_strict_host()
    eng.close()
    monkeypatch.setenv("HAF_PROB_HOST_ALL", "1")
    eng = make_engine(data_dir, model, capi.FLAG_PROBABILITY)
    got2, _ = compare_probability(eng, o, xyz, dict(n_rolls=12), kw)
    assert eng.last_strict_host() == want["n_evals"] and normal < 0.01 * want["n_evals"], (normal, eng.last_strict_host())
    assert got2 == got
    STATS["probability_estimates_finished_on_host"] = {"normal": normal, "forced": eng.last_strict_host()}
    eng.close()


def test_probability_mode_needs_a_probability_model(data_dir, surrogate):
    with pytest.raises(capi.HafError, match="probA"):
        make_engine(data_dir, surrogate, capi.FLAG_PROBABILITY)


def test_probability_mode_randomised(data_dir, tmp_path):
    """Random requests x random models with random sigmoid parameters (label order "1 -1", so the header parses to +0 and the
    positive label takes the SECOND probability, that of label -1: the reference's own quirk), probability mode against the oracle."""
    f, r = _files(data_dir)
    rng = np.random.RandomState(4242)
    for case in range(6):
        base = str(tmp_path / ("r%d.model" % case))
        models.write_random_model(base, int(rng.choice([3, 40, 300])), seed=int(rng.randint(1 << 30)), balanced=True,
                                  gamma=float(rng.choice([1.0 / 323, 0.02])))
        mp = models.write_probability_model(str(tmp_path / ("rp%d.model" % case)), base, "%g" % rng.uniform(-40, 40),
                                            "%g" % rng.uniform(-2, 2))
        o = O.Oracle(f, r, mp)
        n_rolls, step = [(12, 15), (5, 36), (3, 60)][case % 3]
        eng = make_engine(data_dir, mp, capi.FLAG_PROBABILITY, n_rolls=n_rolls, roll_step_deg=step, max_points=1 << 16)
        for _ in range(3):
            n = int(rng.choice([200, 5000]))
            pts = [rng.uniform(-0.1, 0.1, 3) * [1, 1, 0.3] + [0, 0, 0.05] + rng.standard_normal((n, 3)) * rng.uniform(0.005, 0.05, 3)
                   for _ in range(rng.randint(1, 4))]
            xy = rng.uniform(-0.3, 0.3, (n, 2))
            pts.append(np.column_stack([xy, 0.02 + 0.1 * xy[:, 0] - 0.05 * xy[:, 1]]))
            xyz = np.concatenate(pts).astype(np.float32)
            kw = dict(grasp_area_center=tuple(rng.uniform(-0.05, 0.05, 3) * [1, 1, 0.2]),
                      grasp_area_length_x=float(rng.choice([20, 32, 44, 56])), grasp_area_length_y=float(rng.choice([18, 32, 44, 56])),
                      gripper_opening_width=int(rng.choice([1, 1, 2])), show_only_best_grasp=int(rng.rand() < 0.3))
            if rng.rand() < 0.5:
                kw["approach_vector"] = tuple(rng.standard_normal(3) * [0.3, 0.3, 1.0] + [0, 0, 1.0])
            compare_probability(eng, o, xyz, dict(n_rolls=n_rolls, roll_step_deg=step), kw)
        eng.close()


def test_probability_mode_through_the_cli(data_dir, golden_dir, tmp_path):
    """haf_grasp_cli --probability prints the oracle's grasp for pcd2 at C2."""
    mp = _prob_model(golden_dir, tmp_path)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, mp)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd2.pcd"))
    want = o.run(xyz, O.make_cfg(probability=1), O.make_input())
    import subprocess
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "haf_grasp_cli")
    out = subprocess.run([cli, "--features", f, "--range", r, "--model", mp, "--search-size", "18", "18", "--probability",
                          os.path.join(data_dir, "pcd2.pcd")], capture_output=True, text=True, check=True)
    line = out.stdout.strip().splitlines()[-1].split()
    assert int(line[0]) == want["eval"], (out.stdout, out.stderr)
    np.testing.assert_allclose([float(v) for v in line[1:4]], want["gp1"], atol=1e-4)


def test_probability_mode_roll_sharded(data_dir, golden_dir, tmp_path):
    """The probability mode behind haf_score_sharded: three shards (on one GPU here) give the oracle's grasp and roll records."""
    mp = _prob_model(golden_dir, tmp_path)
    f, r = _files(data_dir)
    o = O.Oracle(f, r, mp)
    xyz = pcdio.load_pcd(os.path.join(data_dir, "pcd12.pcd"))
    want = o.run(xyz, O.make_cfg(probability=1), O.make_input(length_y=44))
    me = capi.MultiEngine(f, r, mp, [0, 0, 0], capi.SHARD_ROLLS, flags=capi.FLAG_PROBABILITY)
    got = me.score_sharded(xyz, capi.default_input(grasp_area_length_x=32, grasp_area_length_y=44))
    assert (got["eval"], got["best_row"], got["best_col"], got["best_roll"]) == (want["eval"], want["row"], want["col"], want["roll_idx"])
    rec = me.last_records(0)
    for roll in range(12):
        assert [int(rec["row"][roll]), int(rec["col"][roll]), int(rec["vote"][roll])] == list(want["roll_best"][roll])
    me.close()


def test_probability_mode_batch_and_roll_shards(data_dir, golden_dir, tmp_path):
    """Probability mode: a batch of eight clouds == eight single calls (per-cloud grids and records do not leak into each other),
    and roll shards + haf_finalize == the unsharded call."""
    mp = _prob_model(golden_dir, tmp_path)
    names = ["pcd%d" % i for i in range(1, 9)]
    clouds = [capi.load_pcd(os.path.join(data_dir, n + ".pcd")) for n in names]
    inputs = [capi.default_input(grasp_area_length_x=32, grasp_area_length_y=44) for _ in names]
    inputs[6] = capi.default_input(grasp_area_center=(0.30, 0.46, 0.0))
    eng = make_engine(data_dir, mp, capi.FLAG_PROBABILITY, n_rolls=20, roll_step_deg=9, max_clouds=8)
    batch = eng.score_batch(clouds, inputs)
    grids = [eng.debug(capi.DBG_GRASPSGRID, b, 7).copy() for b in range(8)]
    singles = []
    for b, (c, i) in enumerate(zip(clouds, inputs)):
        singles.append(eng.score(c, i))
        assert (eng.debug(capi.DBG_GRASPSGRID, 0, 7).view(np.uint32) == grids[b].view(np.uint32)).all(), b
    for b, s in zip(batch, singles):
        for k in ("eval", "best_row", "best_col", "best_roll", "best_vote", "n_evals", "grasp_point1", "roll"):
            assert b[k] == s[k], k
    full = eng.score_rolls(clouds[:3], inputs[:3], 0, 20)
    parts = [eng.score_rolls(clouds[:3], inputs[:3], a, n) for a, n in ((0, 5), (5, 5), (10, 7), (17, 3))]
    assert (np.concatenate(parts, axis=1) == full).all()
    eng.close()
