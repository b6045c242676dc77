#!/usr/bin/env python3
"""Two C5 requests in flight on ONE GPU: two engines (each its own stream and device state), one host thread each, against one engine
serving the same number of requests one after the other.  A request's tail -- the tiers behind the sweep, pre-stages, vote: ~1.4 ms of
short lists and small launches in a 15 ms step -- leaves most of the chip idle; a second request's feature kernel and sweep fill it.
On a GPU box:  python tools/two_in_flight.py [--steps 12] [--nsv 4096] [--seed 42]"""
import argparse, os, sys, tempfile, threading, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import models
from haf_grasping_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--nsv", type=int, default=4096)
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--rolls", type=int, default=36)
ap.add_argument("--engines", type=int, default=2)
a = ap.parse_args()
D = os.path.join(ROOT, "tests", "golden", "data")
feat, rng = os.path.join(D, "Features.txt"), os.path.join(D, "range21062012_allfeatures")
tmp = tempfile.mkdtemp()
mp = models.write_random_model(os.path.join(tmp, "m.model"), a.nsv, D=323, seed=a.seed, balanced=True)
G = a.grid
xyz = models.synthetic_cloud(grid=G, k=2, seed=0)
d_xyz = torch.from_numpy(xyz).cuda()
cloud = (d_xyz.data_ptr(), xyz.shape[0], 3)
inp = capi.default_input(grasp_area_length_x=G, grasp_area_length_y=G)


def make():
    return capi.Engine(feat, rng, mp, device=0, grid_h=G, grid_w=G, n_rolls=a.rolls, roll_step_deg=5, max_clouds=1, max_points=G * G * 2, flags=0)


def serve(eng, n, out):
    ev = 0
    for _ in range(n):
        rec = eng.score_rolls([cloud], [inp], 0, a.rolls)[0]
        ev += int(rec["n_evals"].sum())
    out.append((ev, eng.finalize(inp, rec)))


engs = [make() for _ in range(a.engines)]
for e in engs:
    serve(e, 2, [])
torch.cuda.synchronize()
# one engine, all the requests one after the other
res = []
t0 = time.perf_counter()
serve(engs[0], a.steps * a.engines, res)
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
ev_seq, best_seq = res[0]
# the same number of requests, one thread per engine
res = []
ths = [threading.Thread(target=serve, args=(e, a.steps, res)) for e in engs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
t_par = time.perf_counter() - t0
ev_par = sum(r[0] for r in res)
same = all(r[1]["best_vote"] == best_seq["best_vote"] and r[1]["best_roll"] == best_seq["best_roll"] for r in res)
n = a.steps * a.engines
print("one engine   : %d requests in %.1f ms = %.2f ms per request, %.3e evals/s" % (n, 1e3 * t_seq, 1e3 * t_seq / n, ev_seq / t_seq))
print("%d in flight  : %d requests in %.1f ms = %.2f ms per request, %.3e evals/s (x %.3f), same best grasp: %s" %
      (a.engines, n, 1e3 * t_par, 1e3 * t_par / n, ev_par / t_par, (ev_par / t_par) / (ev_seq / t_seq), same))
for e in engs: e.close()
