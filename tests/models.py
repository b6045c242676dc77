"""Seeded synthetic libsvm-3.12 text models and synthetic clouds (test / bench inputs; data only)."""
import os

import numpy as np


def write_random_model(path, nsv, D=323, seed=0, gamma=None, rho=0.1, density=1.0, balanced=False):
    """C-SVC / RBF model in libsvm's text format (svm.cpp:2599-2691 writer layout).

    SURVEY.md §8(d): SV values U(-1,1) printed %.8g, coef ~ U(0,2) with the class sign (printed %.16g),
    gamma = 1/D, rho 0.1, labels "1 -1", first half of the SVs belong to label 1.
    balanced=True rescales the negative class so that sum(coef) = 0 (as in every real C-SVC solution) and uses
    rho = 0.01: decision values then straddle zero and both labels occur.
    """
    rng = np.random.RandomState(seed)
    gamma = 1.0 / D if gamma is None else gamma
    n0 = nsv // 2
    n1 = nsv - n0
    sv = rng.uniform(-1.0, 1.0, size=(nsv, D))
    keep = rng.uniform(size=(nsv, D)) < density
    coef = rng.uniform(0.0, 2.0, size=nsv)
    coef[n0:] *= -1.0
    if balanced:
        coef[n0:] *= coef[:n0].sum() / -coef[n0:].sum()
        rho = 0.01
    with open(path, "w") as f:
        f.write("svm_type c_svc\nkernel_type rbf\ngamma %g\nnr_class 2\ntotal_sv %d\nrho %g\nlabel 1 -1\nnr_sv %d %d\nSV\n"
                % (gamma, nsv, rho, n0, n1))
        for i in range(nsv):
            parts = ["%.16g " % coef[i]]
            row = sv[i]
            for k in range(D):
                if keep[i, k]:
                    parts.append("%d:%.8g " % (k + 1, row[k]))
            f.write("".join(parts) + "\n")
    return path


def write_clustered_model(path, nsv, D=323, seed=0, gamma=1e-4, coef_scale=1000.0, spread=0.25, rho=0.3):
    """An ILL-CONDITIONED synthetic model of the trained kind: support vectors clustered around one centre
    (every attribute N(centre_k, spread)), coefficients coef_scale * U(0.5, 1) with sum 0, a small gamma: the kernel values are all
    close to one another, sum|coef|K is 1e5..1e7 times the decision values, z = p.q stays small -- what the centred-remainder form of
    the screening pass is for, and nothing an fp32 coefficient sum can decide."""
    rng = np.random.RandomState(seed)
    n0 = nsv // 2
    centre = rng.uniform(-0.6, 0.6, D)
    sv = centre[None, :] + rng.standard_normal((nsv, D)) * spread
    coef = coef_scale * rng.uniform(0.5, 1.0, nsv)
    coef[n0:] *= -1.0
    coef[n0:] *= coef[:n0].sum() / -coef[n0:].sum()
    with open(path, "w") as f:
        f.write("svm_type c_svc\nkernel_type rbf\ngamma %g\nnr_class 2\ntotal_sv %d\nrho %g\nlabel -1 1\nnr_sv %d %d\nSV\n"
                % (gamma, nsv, rho, n0, nsv - n0))
        for i in range(nsv):
            f.write("%.16g " % coef[i] + "".join("%d:%.8g " % (k + 1, sv[i, k]) for k in range(D)) + "\n")
    return path


def synthetic_cloud(grid=512, k=2, seed=0, cell=0.01):
    """SURVEY.md §8(d) C5 cloud: for each 1 cm cell of a grid x grid area centred on the origin, k points at
    uniform-random xy inside the cell, z = 0.05 + 0.20*smooth(x,y) + U(0,0.005) with smooth = mean of 8 seeded
    cosines mapped to [0,1].  Returns float32 [grid*grid*k, 3]."""
    rng = np.random.RandomState(seed)
    half = grid * cell / 2.0
    ii, jj = np.meshgrid(np.arange(grid), np.arange(grid), indexing="ij")
    x = (ii[..., None] + rng.uniform(0.02, 0.98, size=(grid, grid, k))) * cell - half
    y = (jj[..., None] + rng.uniform(0.02, 0.98, size=(grid, grid, k))) * cell - half
    fx = rng.uniform(2.0, 40.0, size=8)
    fy = rng.uniform(2.0, 40.0, size=8)
    ph = rng.uniform(0, 2 * np.pi, size=8)
    s = np.zeros_like(x)
    for a, b, p in zip(fx, fy, ph):
        s += np.cos(a * x + b * y + p)
    s = 0.5 + 0.5 * s / 8.0
    z = 0.05 + 0.20 * s + rng.uniform(0.0, 0.005, size=x.shape)
    pts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    return pts


def write_replicated_model(path, base_model_path, copies=24, jitter=0.01, seed=0):
    """An ILL-CONDITIONED model of bench size from a genuine one: every support vector of `base_model_path` (the committed
    surrogate: libsvm-3.12 svm-train -c 512 on real feature rows, 172 SVs, most coefficients at the bound +-512) appears
    `copies` times with its attribute values jittered by N(0, jitter) (printed %.8g like libsvm does) and its coefficient
    kept.  The decision function is `copies` x the base model's up to the jitter, so the decisions keep crowding around
    zero against sum|coef|K exactly as they do for the trained model -- the case the plain screening band cannot decide.
    rho is scaled with the copies; labels and class order are the base model's."""
    rng = np.random.RandomState(seed)
    with open(base_model_path) as f:
        text = f.read()
    head, body = text.split("SV\n", 1)
    keys = dict(line.split(" ", 1) for line in head.strip().splitlines())
    n0, n1 = [int(t) for t in keys["nr_sv"].split()]
    rows = []
    for line in body.strip().splitlines():
        t = line.split()
        rows.append((float(t[0]), [(int(p.split(":")[0]), float(p.split(":")[1])) for p in t[1:]]))
    assert len(rows) == n0 + n1
    out = []
    for cls_rows in (rows[:n0], rows[n0:]):                 # libsvm keeps the SVs grouped by class
        for coef, pairs in cls_rows:
            for _ in range(copies):
                parts = ["%.16g " % coef]
                for k, v in pairs:
                    parts.append("%d:%.8g " % (k, v + rng.normal(0.0, jitter)))
                out.append("".join(parts))
    with open(path, "w") as f:
        f.write("svm_type c_svc\nkernel_type rbf\ngamma %s\nnr_class 2\ntotal_sv %d\nrho %.9g\nlabel %s\nnr_sv %d %d\nSV\n"
                % (keys["gamma"].strip(), len(out), float(keys["rho"]) * copies, keys["label"].strip(), n0 * copies, n1 * copies))
        f.write("\n".join(out) + "\n")
    return path


def write_probability_model(dst, src, probA, probB):
    """Copy of the libsvm model `src` with `probA` / `probB` header lines (text tokens) in svm_save_model's place for them
    (svm.cpp:2641-2655: behind `label`, before `nr_sv`): what `svm-train -b 1` writes for the same support vectors."""
    with open(src) as f:
        head, body = f.read().split("nr_sv", 1)
    with open(dst, "w") as f:
        f.write(head + "probA %s\nprobB %s\nnr_sv" % (probA, probB) + body)
    return dst


def pack_trained_model(model_path, npz_path):
    """Lossless, compact form of a libsvm-3.12 text model (svm_save_model, svm.cpp:2599-2691) for a committed fixture: the
    header text, the coefficients as doubles ("%.16g" tokens; a token that a double does not print back to is kept as
    text) and the sparse SV values -- which libsvm prints "%.8g" from doubles that svm-scale wrote with six significant
    digits -- as float32 + "%.6g" when every token survives that, else as float64 + "%.8g".  unpack_trained_model()
    rewrites the file byte for byte (the caller checks)."""
    with open(model_path) as f:
        text = f.read()
    head, body = text.split("SV\n", 1)
    coef, coef_txt, indptr, indices, toks = [], {}, [0], [], []
    for r, line in enumerate(body.splitlines()):
        t = line.split()
        c = float(t[0])
        if "%.16g" % c != t[0]:
            coef_txt[r] = t[0]
        coef.append(c)
        for p in t[1:]:
            k, v = p.split(":")
            indices.append(int(k))
            toks.append(v)
        indptr.append(len(indices))
    v64 = np.array([float(v) for v in toks], np.float64)
    v32 = v64.astype(np.float32)
    as32 = all("%.6g" % float(a) == tok for a, tok in zip(v32, toks))
    if not as32:
        assert all("%.8g" % a == tok for a, tok in zip(v64, toks))
    np.savez_compressed(npz_path, head=np.array(head), coef=np.array(coef, np.float64),
                        coef_txt_rows=np.array(sorted(coef_txt), np.int64), coef_txt=np.array([coef_txt[r] for r in sorted(coef_txt)]),
                        indptr=np.array(indptr, np.int64), indices=np.array(indices, np.int16), vals=v32 if as32 else v64)
    keys = dict(line.split(" ", 1) for line in head.strip().splitlines())
    return dict(total_sv=int(keys["total_sv"]), nr_sv=[int(t) for t in keys["nr_sv"].split()], values_as="float32" if as32 else "float64")


def unpack_trained_model(npz_path, model_path):
    """Writes the libsvm text model packed by pack_trained_model()."""
    z = np.load(npz_path)
    vals, fmt = z["vals"], ("%d:%.6g " if z["vals"].dtype == np.float32 else "%d:%.8g ")
    over = dict(zip(z["coef_txt_rows"].tolist(), z["coef_txt"].tolist()))
    indptr, indices, coef = z["indptr"], z["indices"], z["coef"]
    with open(model_path, "w") as f:
        f.write(str(z["head"]) + "SV\n")
        for r in range(len(coef)):
            a, b = indptr[r], indptr[r + 1]
            f.write((over[r] if r in over else "%.16g" % coef[r]) + " " +
                    "".join(fmt % (k, float(v)) for k, v in zip(indices[a:b].tolist(), vals[a:b].tolist())) + "\n")
    return model_path


KERNEL_MODELS = ("linear", "poly", "sigmoid", "nu_rbf")


def unpack_kernel_model(golden_dir, name, model_path):
    """tests/golden/kernel_models.npz: the model texts the REFERENCE svm-train wrote for libsvm's other vector kernels and nu-SVC
    (tests/golden/make_fixtures.py: make_gk), byte for byte."""
    z = np.load(os.path.join(golden_dir, "kernel_models.npz"))
    with open(model_path, "wb") as f:
        f.write(z[name].tobytes())
    return model_path

