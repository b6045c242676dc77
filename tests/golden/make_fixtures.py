#!/usr/bin/env python3
"""Regenerates the committed golden fixtures.  Runs ONLY in the build container (needs /root/reference for
oracle/_ref: the real libsvm-3.12 svm-train / svm-scale / svm-predict / svm.cpp built by oracle/Makefile).

Outputs (all data, no code):
  surrogate.model         genuine libsvm-3.12 C-SVC/RBF model trained by the REFERENCE svm-train on feature rows
                          harvested from tests/golden/data/*.pcd, labelled by a fixed geometric rule.  The real
                          data/all_features.txt.scale.model is missing from the reference checkout
                          (.MISSING_LARGE_BLOBS), so every parity statement is against this surrogate.
  g23_<cloud>_r<roll>.npz reference-tool outputs for one roll's feature file: raw float features (oracle), the
                          values the REAL svm-scale printed (parsed with strtod), the labels the REAL svm-predict
                          printed, and fp64 decision values from the REAL svm_predict_values (libsvm_ref.so).
  heart_scale*, g5_heart.npz  libsvm known-answer: reference svm-train -c 2 -g 0.5 heart_scale -> model, labels,
                          decision values (190 SVs, 259/270).
  heart_scale_prob.model, g5_heart_prob.npz   probability output: reference svm-train -b 1 / svm-predict -b 1
                          on heart_scale: printed labels and probabilities, unrounded svm_predict_probability estimates.
  surrogate_prob.json, g23p_<cloud>_r<roll>.npz  probA/probB of the surrogate from the reference svm-train -b 1 (same SVs), and
                          the reference svm-predict -b 1 output lines for the g23 rows.
  g6_end_to_end.json      per-roll (row, col, val), overall best and GraspOutput of the ORACLE for every cloud x
                          configuration: regression goldens for the GPU engine.
  g6_trained.json         the same for the large trained model (trained.model.npz, made by tools/make_trained_model.py) on six
                          cloud x configuration cases (`g6t`; not part of the default set: the model must exist first).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pcdio  # noqa: E402
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(HERE, "data")
REF = O.ref_dir()
FEATURES = os.path.join(DATA, "Features.txt")
RANGE = os.path.join(DATA, "range21062012_allfeatures")
TMP = "/tmp/haf_fixtures"


def label_rule(win):
    """Fixed geometric surrogate for 'graspable': centre block at least 2 cm above both finger strips."""
    c = win[5:9, 4:10].mean()
    s = max(win[0:3, 4:10].mean(), win[11:14, 4:10].mean())
    return 1 if c - s > 0.02 else -1


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def parse_sparse(path, D):
    rows = []
    labels = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            labels.append(float(t[0]))
            x = np.zeros(D)
            for tok in t[1:]:
                k, v = tok.split(":")
                if int(k) <= D:
                    x[int(k) - 1] = float(v)      # python float() == strtod
            rows.append(x)
    return np.array(labels), np.array(rows).reshape(len(rows), D)


def make_surrogate(train_model=True):
    """train_model=False: only (re)writes the 700 training rows to TMP/train.txt (make_gk trains its own models on them)."""
    os.makedirs(TMP, exist_ok=True)
    orc = O.Oracle(FEATURES, RANGE, None)
    cfg = O.make_cfg()
    inp = O.make_input()
    train = os.path.join(TMP, "train.txt")
    rng = np.random.RandomState(20150817)
    lines = []
    for name in ["pcd1", "pcd3", "pcd12", "plastic_mug2", "pcd11", "pcd2"]:
        xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
        for roll in range(0, 12, 2):
            fpath = os.path.join(TMP, "f_%s_%d.txt" % (name, roll))
            spath = fpath + ".scale"
            n = orc.dump_feature_file(xyz, cfg, inp, roll, fpath)
            if n <= 0:
                continue
            with open(spath, "w") as out:
                run([os.path.join(REF, "svm-scale"), "-r", RANGE, fpath], stdout=out)
            M = np.zeros(16, np.float32)
            O.lib().hafo_transform(C.byref(cfg), C.byref(inp), roll, 0, M.ctypes.data_as(C.c_void_p))
            h = np.zeros((56, 56), np.float32)
            O.lib().hafo_height_grid(C.byref(cfg), xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], 3,
                                     M.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p))
            ii = np.zeros((57, 57), np.float32)
            O.lib().hafo_integral(C.byref(cfg), h.ctypes.data_as(C.c_void_p), ii.ctypes.data_as(C.c_void_p))
            mask = np.zeros((56, 56), np.uint8)
            O.lib().hafo_mask(C.byref(cfg), C.byref(inp), roll, ii.ctypes.data_as(C.c_void_p),
                              mask.ctypes.data_as(C.c_void_p))
            cells = list(zip(*np.nonzero(mask)))
            with open(spath) as f:
                scaled = f.read().splitlines()
            assert len(scaled) == len(cells) == n
            for (i, j), sl in zip(cells, scaled):
                lab = label_rule(h[i - 7:i + 7, j - 7:j + 7])
                body = sl.split(" ", 1)[1]
                lines.append("%+d %s" % (lab, body))
    idx = rng.permutation(len(lines))[:700]
    with open(train, "w") as f:
        for i in sorted(idx):
            f.write(lines[i] + "\n")
    model = os.path.join(HERE, "surrogate.model")
    if not train_model:
        return model
    run([os.path.join(REF, "svm-train"), "-c", "512", "-g", "0.0031", "-q", train, model])
    with open(model) as f:
        head = [next(f) for _ in range(9)]
    print("surrogate:", "".join(head).replace("\n", " | "), os.path.getsize(model), "bytes")
    return model


def ref_decisions(model_path, scaled_path):
    """fp64 decision values straight from the reference svm_predict_values (svm.cpp:2459)."""
    L = C.CDLL(os.path.join(REF, "libsvm_ref.so"))

    class Node(C.Structure):
        _fields_ = [("index", C.c_int), ("value", C.c_double)]
    L.svm_load_model.restype = C.c_void_p
    L.svm_load_model.argtypes = [C.c_char_p]
    L.svm_predict_values.restype = C.c_double
    L.svm_predict_values.argtypes = [C.c_void_p, C.POINTER(Node), C.POINTER(C.c_double)]
    m = L.svm_load_model(model_path.encode())
    assert m
    decs, labs = [], []
    with open(scaled_path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            nodes = (Node * len(t))()
            for i, tok in enumerate(t[1:]):
                k, v = tok.split(":")
                nodes[i].index = int(k)
                nodes[i].value = float(v)
            nodes[len(t) - 1].index = -1
            d = C.c_double()
            labs.append(L.svm_predict_values(m, nodes, C.byref(d)))
            decs.append(d.value)
    return np.array(decs), np.array(labs)


def make_g23(model):
    orc = O.Oracle(FEATURES, RANGE, model)
    cfg = O.make_cfg()
    inp = O.make_input()
    D = orc.model_arrays()["D"]
    for name, roll in [("pcd2", 0), ("pcd2", 5), ("pcd3", 2), ("plastic_mug2", 7)]:
        xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
        fpath = os.path.join(TMP, "g_%s_%d.txt" % (name, roll))
        n = orc.dump_feature_file(xyz, cfg, inp, roll, fpath)
        with open(fpath + ".scale", "w") as out:
            run([os.path.join(REF, "svm-scale"), "-r", RANGE, fpath], stdout=out)
        run([os.path.join(REF, "svm-predict"), fpath + ".scale", model, fpath + ".out"], stdout=subprocess.DEVNULL)
        _, q4 = parse_sparse(fpath, 324)                      # the %.4g values as svm-scale reads them
        _, scaled = parse_sparse(fpath + ".scale", max(D, 323))
        labels = np.loadtxt(fpath + ".out").reshape(-1)
        dec, lab2 = ref_decisions(model, fpath + ".scale")
        assert n == len(q4) == len(scaled) == len(labels) == len(dec)
        assert (labels == lab2).all()
        # raw float features from the oracle, so the fixture is self-contained
        r = orc.run(xyz, cfg, inp)
        ii = r["integral"][roll]
        feats = np.array([orc.feature_values(ii[i - 7:i + 8, j - 7:j + 8]) for i, j in zip(*np.nonzero(r["mask"][roll]))])
        sel = slice(0, None, 3)     # every third row keeps the fixtures small
        np.savez_compressed(os.path.join(HERE, "g23_%s_r%d.npz" % (name, roll)),
                            features=feats.astype(np.float32)[sel], q4=q4[sel], scaled=scaled[sel],
                            labels=labels.astype(np.int8)[sel], dec=dec[sel],
                            cells=np.stack(np.nonzero(r["mask"][roll]), 1).astype(np.int16)[sel])
        print("g23", name, roll, n, "rows; +1:", int((labels > 0).sum()))


def make_g3t(model):
    """g3_trained.npz: for the rows of the four g23 fixtures (the doubles the REAL svm-scale printed and svm-predict parses), the fp64
    decision values of the REAL svm_predict_values (libsvm_ref.so) and the labels of the REAL svm-predict with the large trained
    model: pins the oracle's RBF decision on an 8964-SV model whose decisions are ~1e-7 of sum|coef|K."""
    decs, labs, names = [], [], []
    for name in ["g23_pcd2_r0", "g23_pcd2_r5", "g23_pcd3_r2", "g23_plastic_mug2_r7"]:
        g = np.load(os.path.join(HERE, name + ".npz"))
        fpath = os.path.join(TMP, name + ".scaled.txt")
        with open(fpath, "w") as f:
            for row in g["scaled"]:
                f.write("0 " + " ".join("%d:%r" % (k + 1, float(v)) for k, v in enumerate(row) if v != 0.0) + "\n")   # repr: strtod gives the double back
        run([os.path.join(REF, "svm-predict"), fpath, model, fpath + ".out"], stdout=subprocess.DEVNULL)
        labels = np.loadtxt(fpath + ".out").reshape(-1)
        dec, lab2 = ref_decisions(model, fpath)
        assert (labels == lab2).all() and len(dec) == len(g["scaled"])
        decs.append(dec)
        labs.append(labels.astype(np.int8))
        names.append(name)
        print("g3t", name, len(dec), "rows; +1:", int((labels > 0).sum()), "min |dec| %.3g" % np.abs(dec).min())
    np.savez_compressed(os.path.join(HERE, "g3_trained.npz"), **{n + "_dec": d for n, d in zip(names, decs)},
                        **{n + "_labels": l for n, l in zip(names, labs)})


KERNEL_MODELS = [("linear", ["-t", "0", "-c", "1"]),
                 ("poly", ["-t", "1", "-d", "3", "-g", "0.0031", "-r", "1", "-c", "8"]),
                 ("sigmoid", ["-t", "3", "-g", "0.0031", "-r", "-0.5", "-c", "64"]),
                 ("nu_rbf", ["-s", "1", "-n", "0.25", "-g", "0.0031"])]


def make_gk():
    """Round 5: libsvm's OTHER vector kernels and nu-SVC (svm-predict serves them: Kernel::k_function svm.cpp:318-371).  The reference
    svm-train on the surrogate's own 700 training rows (TMP/train.txt, written by make_surrogate) with -t 0 / 1 / 3 and -s 1; the model
    texts go into kernel_models.npz byte for byte (compressed), and for the rows of the four g23 fixtures the fp64 decision values of
    the REAL svm_predict_values (libsvm_ref.so) and the labels of the REAL svm-predict into gk_kernels.npz."""
    train = os.path.join(TMP, "train.txt")
    if not os.path.exists(train):
        make_surrogate(train_model=False)
    texts, out = {}, {}
    for kname, args in KERNEL_MODELS:
        model = os.path.join(TMP, "k_%s.model" % kname)
        run([os.path.join(REF, "svm-train")] + args + ["-q", train, model])
        with open(model, "rb") as f:
            texts[kname] = np.frombuffer(f.read(), dtype=np.uint8)
        with open(model) as f:
            head = [next(f) for _ in range(10)]
        print("gk", kname, " | ".join(h.strip() for h in head if not h.startswith("SV"))[:200], os.path.getsize(model), "bytes")
        for name in ["g23_pcd2_r0", "g23_pcd2_r5", "g23_pcd3_r2", "g23_plastic_mug2_r7"]:
            g = np.load(os.path.join(HERE, name + ".npz"))
            fpath = os.path.join(TMP, name + ".scaled.txt")
            with open(fpath, "w") as f:
                for row in g["scaled"]:
                    f.write("0 " + " ".join("%d:%r" % (k + 1, float(v)) for k, v in enumerate(row) if v != 0.0) + "\n")
            run([os.path.join(REF, "svm-predict"), fpath, model, fpath + ".out"], stdout=subprocess.DEVNULL)
            labels = np.loadtxt(fpath + ".out").reshape(-1)
            dec, lab2 = ref_decisions(model, fpath)
            assert (labels == lab2).all() and len(dec) == len(g["scaled"])
            out["%s_%s_dec" % (kname, name)] = dec
            out["%s_%s_labels" % (kname, name)] = labels.astype(np.int8)
            print("   ", name, len(dec), "rows; +1:", int((labels > 0).sum()), "min |dec| %.3g" % np.abs(dec).min())
    np.savez_compressed(os.path.join(HERE, "kernel_models.npz"), **texts)
    np.savez_compressed(os.path.join(HERE, "gk_kernels.npz"), **out)


def make_g5():
    src = "/root/reference/libsvm-3.12/heart_scale"
    dst = os.path.join(HERE, "heart_scale")
    with open(src) as f, open(dst, "w") as g:
        g.write(f.read())
    model = os.path.join(HERE, "heart_scale.model")
    run([os.path.join(REF, "svm-train"), "-c", "2", "-g", "0.5", "-q", dst, model])
    out = os.path.join(TMP, "heart.out")
    res = run([os.path.join(REF, "svm-predict"), dst, model, out], stdout=subprocess.PIPE).stdout.decode()
    print("g5:", res.strip())
    dec, lab = ref_decisions(model, dst)
    labels = np.loadtxt(out)
    assert (labels == lab).all()
    y, X = parse_sparse(dst, 13)
    np.savez_compressed(os.path.join(HERE, "g5_heart.npz"), X=X, y=y, labels=labels.astype(np.int8), dec=dec)


def ref_probabilities(model_path, scaled_path):
    """Unrounded probability estimates and labels straight from the reference svm_predict_probability (svm.cpp:2550)."""
    L = C.CDLL(os.path.join(REF, "libsvm_ref.so"))

    class Node(C.Structure):
        _fields_ = [("index", C.c_int), ("value", C.c_double)]
    L.svm_load_model.restype = C.c_void_p
    L.svm_load_model.argtypes = [C.c_char_p]
    L.svm_predict_probability.restype = C.c_double
    L.svm_predict_probability.argtypes = [C.c_void_p, C.POINTER(Node), C.POINTER(C.c_double)]
    m = L.svm_load_model(model_path.encode())
    assert m
    probs, labs = [], []
    with open(scaled_path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            nodes = (Node * len(t))()
            for i, tok in enumerate(t[1:]):
                k, v = tok.split(":")
                nodes[i].index = int(k)
                nodes[i].value = float(v)
            nodes[len(t) - 1].index = -1
            pr = (C.c_double * 2)()
            labs.append(L.svm_predict_probability(m, nodes, pr))
            probs.append((pr[0], pr[1]))
    return np.array(probs), np.array(labs)


def read_prob_output(path):
    """svm-predict -b 1 output: header 'labels a b', then 'label p(a) p(b)' per row (svm-predict.c:60-64, 111-118)."""
    with open(path) as f:
        lines = f.read().splitlines()
    assert lines[0].split()[0] == "labels"
    rows = [ln.split() for ln in lines[1:]]
    return (lines[0], np.array([float(r[0]) for r in rows]), np.array([[float(r[1]), float(r[2])] for r in rows]),
            [" ".join(r) for r in rows])


def make_g5p():
    """f4, libsvm side: reference svm-train -b 1 on heart_scale, svm-predict -b 1 on the same rows."""
    dst = os.path.join(HERE, "heart_scale")
    model = os.path.join(HERE, "heart_scale_prob.model")
    run([os.path.join(REF, "svm-train"), "-b", "1", "-c", "2", "-g", "0.5", "-q", dst, model])
    out = os.path.join(TMP, "heart_prob.out")
    os.makedirs(TMP, exist_ok=True)
    run([os.path.join(REF, "svm-predict"), "-b", "1", dst, model, out], stdout=subprocess.DEVNULL)
    header, labels, prob_text, _ = read_prob_output(out)
    dec, _ = ref_decisions(model, dst)
    raw, lab2 = ref_probabilities(model, dst)
    assert (labels == lab2).all()
    np.savez_compressed(os.path.join(HERE, "g5_heart_prob.npz"), labels=labels.astype(np.int8), prob_text=prob_text, prob_raw=raw,
                        dec=dec, header=np.array(header))
    print("g5p:", header, "| +1:", int((labels > 0).sum()), "of", len(labels))


def make_g23p(model):
    """f4, hot path: the surrogate's training set through the reference svm-train -b 1 gives probA/probB for the SAME support
    vectors (checked); the g23 rows through the reference svm-predict -b 1 give the lines show_predicted_gps would read."""
    train = os.path.join(TMP, "train.txt")
    if not os.path.exists(train):
        sys.exit("run the 'surrogate' step first: it leaves the training rows in " + train)
    pm = os.path.join(TMP, "surrogate_prob.model")
    run([os.path.join(REF, "svm-train"), "-b", "1", "-c", "512", "-g", "0.0031", "-q", train, pm])
    with open(pm) as f:
        ptxt = f.read()
    with open(model) as f:
        mtxt = f.read()
    assert ptxt.split("SV\n", 1)[1] == mtxt.split("SV\n", 1)[1], "svm-train -b 1 ended on other support vectors"
    toks = {ln.split()[0]: ln.split()[1] for ln in ptxt.split("SV\n", 1)[0].splitlines() if ln.startswith("prob")}
    with open(os.path.join(HERE, "surrogate_prob.json"), "w") as f:
        json.dump(dict(probA=toks["probA"], probB=toks["probB"],
                       note="printed by the reference svm-train -b 1 -c 512 -g 0.0031 on the surrogate's training rows; "
                            "tests insert them into surrogate.model (tests/models.py: write_probability_model)"), f, indent=1)
    orc = O.Oracle(FEATURES, RANGE, model)
    cfg = O.make_cfg()
    inp = O.make_input()
    for name, roll in [("pcd2", 0), ("pcd2", 5), ("pcd3", 2), ("plastic_mug2", 7)]:
        xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
        fpath = os.path.join(TMP, "g_%s_%d.txt" % (name, roll))
        orc.dump_feature_file(xyz, cfg, inp, roll, fpath)
        with open(fpath + ".scale", "w") as out:
            run([os.path.join(REF, "svm-scale"), "-r", RANGE, fpath], stdout=out)
        run([os.path.join(REF, "svm-predict"), "-b", "1", fpath + ".scale", pm, fpath + ".pout"], stdout=subprocess.DEVNULL)
        header, labels, prob_text, lines = read_prob_output(fpath + ".pout")
        raw, lab2 = ref_probabilities(pm, fpath + ".scale")
        assert (labels == lab2).all()
        sel = slice(0, None, 3)     # the rows g23 keeps
        np.savez_compressed(os.path.join(HERE, "g23p_%s_r%d.npz" % (name, roll)), labels=labels.astype(np.int8)[sel],
                            prob_text=prob_text[sel], prob_raw=raw[sel], lines=np.array(lines[sel]), header=np.array(header))
        print("g23p", name, roll, len(labels), "rows; +1:", int((labels > 0).sum()), "probA/B", toks)


CONFIGS = {
    "C1": dict(cfg=dict(n_rolls=1), inp=dict(length_x=32, length_y=32)),
    "C2": dict(cfg=dict(n_rolls=12), inp=dict(length_x=32, length_y=32)),
    "C2best": dict(cfg=dict(n_rolls=12), inp=dict(length_x=32, length_y=32, show_only_best=1)),
    "default": dict(cfg=dict(n_rolls=12), inp=dict(length_x=32, length_y=44)),
    "C3": dict(cfg=dict(n_rolls=20, roll_step_deg=9), inp=dict(length_x=56, length_y=56)),
    "C3c": dict(cfg=dict(n_rolls=20, roll_step_deg=9), inp=dict(length_x=56, length_y=56, center=(0.13, 0.25, 0.0))),
    "C4": dict(cfg=dict(n_rolls=20, roll_step_deg=9), inp=dict(length_x=32, length_y=44)),
    "tilt": dict(cfg=dict(n_rolls=12), inp=dict(length_x=32, length_y=44, approach=(0.2, -0.1, 1.0), gripper_width=1)),
}
CLOUD_CONFIGS = [
    ("pcd2", ["C1", "C2", "C2best", "default", "tilt", "C4"]),
    ("pcd1", ["C2", "C4"]), ("pcd3", ["C2", "C4"]), ("pcd4", ["C4"]), ("pcd5", ["C4"]), ("pcd6", ["C4"]),
    ("pcd7", ["C4"]), ("pcd8", ["C4"]), ("pcd9", ["default"]), ("pcd10", ["default"]), ("pcd11", ["default"]),
    ("pcd12", ["default", "C2"]), ("plastic_mug2", ["default", "C2best"]),
    ("table1_mult_obj_rcs_1428580506606673", ["C3", "C3c"]),
    ("table2_mult_obj_rcs_1428580941635676", ["C3", "C3c"]),
    ("table3_mult_obj_rcs_1428581033679923", ["C3", "C3c"]),
]


TRAINED_CLOUD_CONFIGS = [("pcd2", ["C2", "C2best"]), ("pcd12", ["default"]), ("plastic_mug2", ["default"]),
                         ("table1_mult_obj_rcs_1428580506606673", ["C3c"]), ("table3_mult_obj_rcs_1428581033679923", ["C3"])]


def make_g6(model, cloud_configs=None, out_name="g6_end_to_end.json"):
    orc = O.Oracle(FEATURES, RANGE, model)
    res = {}
    for name, cfgs in (cloud_configs or CLOUD_CONFIGS):
        xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
        for cname in cfgs:
            spec = CONFIGS[cname]
            cfg = O.make_cfg(**spec["cfg"])
            inp = O.make_input(**spec["inp"])
            r = orc.run(xyz, cfg, inp)
            key = "%s/%s" % (name, cname)
            res[key] = dict(eval=r["eval"], row=r["row"], col=r["col"], roll_idx=r["roll_idx"], top=r["top"],
                            n_evals=r["n_evals"], rolls_done=r["rolls_done"], gp1=r["gp1"], gp2=r["gp2"],
                            avg=r["avg"], av=r["av"], roll=float(r["roll"]),
                            roll_best=r["roll_best"].tolist(),
                            masked=[int(v) for v in r["mask"].reshape(cfg.n_rolls, -1).sum(1)],
                            positives=[int(v) for v in (r["labels"] > 0).reshape(cfg.n_rolls, -1).sum(1)])
            print("g6", key, res[key]["eval"], res[key]["row"], res[key]["col"], res[key]["roll_idx"], r["n_evals"])
    with open(os.path.join(HERE, out_name), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    O.build()
    what = sys.argv[1:] or ["surrogate", "g23", "g5", "g6", "g5p", "g23p"]
    model = os.path.join(HERE, "surrogate.model")
    if "surrogate" in what:
        make_surrogate()
    if "g23" in what:
        os.makedirs(TMP, exist_ok=True)
        make_g23(model)
    if "gk" in what:
        make_gk()
    if "g5" in what:
        os.makedirs(TMP, exist_ok=True)
        make_g5()
    if "g6" in what:
        make_g6(model)
    if "g6t" in what:
        # the same goldens for the large trained model (tools/make_trained_model.py -> trained.model.npz): ORACLE results, like g6
        import models
        tpath = os.path.join(TMP, "trained.model")
        os.makedirs(TMP, exist_ok=True)
        models.unpack_trained_model(os.path.join(HERE, "trained.model.npz"), tpath)
        make_g6(tpath, TRAINED_CLOUD_CONFIGS, "g6_trained.json")
    if "g3t" in what:
        import models
        tpath = os.path.join(TMP, "trained.model")
        os.makedirs(TMP, exist_ok=True)
        if not os.path.exists(tpath):
            models.unpack_trained_model(os.path.join(HERE, "trained.model.npz"), tpath)
        make_g3t(tpath)
    if "g5p" in what:
        make_g5p()
    if "g23p" in what:
        make_g23p(model)
