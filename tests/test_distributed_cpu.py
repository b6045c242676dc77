"""world_size-2 gloo test of the multi-GPU logic: roll shards gathered with one all-gather + the cross-roll rule /
pose (host code of the product) reproduce the unsharded answer; one all-reduce(max) elects the best grasp of a batch.
Roll records come from the oracle here (no GPU in this test); on the GPU box test_engine_gpu.py checks that
haf_score_rolls shards compose to the same records."""
import ctypes as C
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

import pcdio
from haf_grasping_amd import capi, distributed
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


def oracle_records(name, n_rolls, step, show_best):
    orc = O.Oracle(os.path.join(DATA, "Features.txt"), os.path.join(DATA, "range21062012_allfeatures"),
                   os.path.join(ROOT, "tests", "golden", "surrogate.model"))
    xyz = pcdio.load_pcd(os.path.join(DATA, name + ".pcd"))
    full = orc.run(xyz, O.make_cfg(n_rolls=n_rolls, roll_step_deg=step), O.make_input(length_x=32, length_y=44))
    rec = np.zeros(n_rolls, capi.ROLL_RECORD_DTYPE)
    for roll in range(n_rolls):
        row, col, val = full["roll_best"][roll]
        win = full["heights"][roll][max(0, row - 4):row + 5, max(0, col - 4):col + 4]
        rec[roll] = (val, row, col, max(np.float32(-10.0), win.max()), int(full["mask"][roll].sum()))
    want = orc.run(xyz, O.make_cfg(n_rolls=n_rolls, roll_step_deg=step),
                   O.make_input(length_x=32, length_y=44, show_only_best=show_best), debug=False)
    return rec, want


def worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n_rolls, step = 9, 20
        for show_best in (0, 1):
            rec, want = oracle_records("pcd2", n_rolls, step, show_best)
            first, count = distributed.roll_shard(n_rolls, world, rank)
            local = rec[None, first:first + count]                      # this rank's rolls only
            full = distributed.gather_roll_records(local, n_rolls)
            assert (full[0] == rec).all()
            out = capi.GraspOutput()
            cfg = capi.default_config(n_rolls=n_rolls, roll_step_deg=step)
            gi = capi.default_input(grasp_area_length_x=32, grasp_area_length_y=44, show_only_best_grasp=show_best)
            assert capi.testlib().haf_test_finalize(C.byref(cfg), C.byref(gi), full[0].ctypes.data, C.byref(out)) == 0
            assert (out.eval, out.best_row, out.best_col, out.best_roll, out.rolls_done) == \
                   (want["eval"], want["row"], want["col"], want["roll_idx"], want["rolls_done"])
            np.testing.assert_allclose(tuple(out.grasp_point1), want["gp1"], atol=1e-6)
        vote, tag = distributed.best_of_batch(100 + rank, tag=rank)
        assert (vote, tag) == (100 + world - 1, world - 1)
        vote, tag = distributed.best_of_batch(77, tag=rank)                 # tie: smaller tag wins
        assert (vote, tag) == (77, 0)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_roll_shards_and_batch_election_world2():
    assert [distributed.roll_shard(36, 8, r) for r in range(8)] == \
           [(0, 5), (5, 5), (10, 5), (15, 5), (20, 4), (24, 4), (28, 4), (32, 4)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus N` without a launcher starts one rank process per GPU itself, before anything in the parent
    touches a GPU, and exits non-zero as soon as any rank fails (here: no GPU in the container) instead of leaving the other
    ranks waiting in a collective."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run the benchmark")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode != 0 and "stopping the other ranks" in p.stderr and time.time() - t0 < 120
    assert p.stdout.strip() == ""                              # no JSON line from a failed run


def _spawn(world, extra):
    """bench.spawn_ranks in a child of its own (it must never share a process with torch), driving tests/stub_rank.py"""
    import subprocess
    import sys
    code = ("import sys, argparse; sys.path.insert(0, %r); import bench; "
            "raise SystemExit(bench.spawn_ranks(argparse.Namespace(gpus=%d), script=%r, argv=%r))"
            % (ROOT, world, os.path.join(ROOT, "tests", "stub_rank.py"), ["--gpus", str(world)] + extra))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420, env=env)


def test_self_spawned_ranks_world8_on_gloo():
    """no 8-GPU run has happened on this pool, so the first one must not fail on plumbing.  bench.py's own launcher
    (spawn_ranks) starts EIGHT rank processes of a stub that does what bench.py's ranks do around the timed region -- rendezvous from
    the launcher's environment, the per-step election collective, the contract's reductions (bench.reduce_over_ranks), rank 0's ONE
    JSON line with the `ranks` block (bench.ranks_block) -- on gloo: rank 0's line is the only stdout, every per-rank field has eight
    entries in rank order, the elapsed time is the MAX over the ranks and the evaluations their SUM."""
    import json
    p = _spawn(8, ["--steps", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    # (the gloo library itself reports its connections on stdout, "[Gloo] Rank r is connected to ..."; RCCL does not)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, p.stdout                                   # rank 0's line and nothing else
    d = json.loads(lines[0])
    rk = d["ranks"]
    assert d["n_gpus"] == 8 and rk["world_size"] == 8 and rk["rccl_world_size"] == 8 and rk["launched_by"] == "bench.py (self-spawned ranks)"
    assert len(rk["per_rank_ms_per_step"]) == len(rk["per_rank_collective_us"]) == len(rk["per_rank_kernel_ms"]) == 8
    assert rk["per_rank_kernel_ms"] == [7.0 + r for r in range(8)]                      # all-gathered in rank order
    np.testing.assert_allclose(rk["per_rank_ms_per_step"], [10.0 * (1.0 + 0.01 * r) for r in range(8)], rtol=1e-12)
    np.testing.assert_allclose(d["ms_per_step"], 10.7, rtol=1e-12)                      # MAX over the ranks
    np.testing.assert_allclose(d["value"], 8 * 3000 / (0.0107 * 3), rtol=1e-12)         # SUM of the evaluations / that time
    assert all(c > 0 for c in rk["per_rank_collective_us"])


def test_self_spawned_ranks_propagate_a_failing_rank():
    """One of eight ranks dies with exit code 7 before the first collective: the launcher returns 7, names the rank, stops the other
    seven at once (they would otherwise wait for the rendezvous until its timeout) and prints no JSON line."""
    import time
    t0 = time.time()
    p = _spawn(8, ["--fail-rank", "5", "--fail-code", "7"])
    assert p.returncode == 7, (p.returncode, p.stderr[-1000:])
    assert "rank 5 exited with 7; stopping the other ranks" in p.stderr
    assert p.stdout.strip() == "" and time.time() - t0 < 120


def test_multi_gpu_partition_without_a_device(data_dir=None):
    """the argument and partition logic of haf_create_multi (csrc/multi.cpp) runs here, on the CPU, through
    haf_multi_plan -- the function create_multi itself uses: 36 rolls over 8 GPUs are 5,5,5,5,4,4,4,4 contiguous ranges
    (SURVEY.md 8e), ranks are the distinct devices in order of first appearance, several shards may share a rank as long as every
    device appears equally often, and bad device lists are refused with the texts a caller of haf_create_multi would get."""
    from haf_grasping_amd import capi
    p = capi.multi_plan(list(range(8)), capi.SHARD_ROLLS, 36)
    assert p["roll_count"] == [5, 5, 5, 5, 4, 4, 4, 4] and p["roll_first"] == [0, 5, 10, 15, 20, 24, 28, 32]
    assert p["rank_of"] == list(range(8)) and p["slot_of"] == [0] * 8 and p["n_ranks"] == 8
    for n_rolls, n in ((12, 5), (20, 8), (36, 7), (1, 1), (12, 12)):
        q = capi.multi_plan(list(range(n)), capi.SHARD_ROLLS, n_rolls)
        assert sum(q["roll_count"]) == n_rolls and max(q["roll_count"]) - min(q["roll_count"]) <= 1
        assert q["roll_first"] == [sum(q["roll_count"][:s]) for s in range(n)] and sorted(q["roll_count"], reverse=True) == q["roll_count"]
    p = capi.multi_plan([2, 2, 5, 5], capi.SHARD_ROLLS, 12)                    # two shards on each of two GPUs
    assert p["rank_of"] == [0, 0, 1, 1] and p["slot_of"] == [0, 1, 0, 1] and p["n_ranks"] == 2 and p["roll_count"] == [3, 3, 3, 3]
    p = capi.multi_plan([3, 1, 3, 1], capi.SHARD_CLOUDS, 12)                   # ranks in order of first appearance
    assert p["rank_of"] == [0, 1, 0, 1] and p["slot_of"] == [0, 0, 1, 1] and p["n_ranks"] == 2
    for devices, mode, n_rolls, text in (([0, 0, 1], capi.SHARD_ROLLS, 12, "same number of times"),
                                         ([0] * 13, capi.SHARD_ROLLS, 12, "more shards than rolls"),
                                         ([0, 1], 7, 12, "unknown shard mode"),
                                         ([0, -1], capi.SHARD_CLOUDS, 12, "negative device"),
                                         ([], capi.SHARD_ROLLS, 12, "bad argument")):
        with pytest.raises(capi.HafError) as ei:
            capi.multi_plan(devices, mode, n_rolls)
        assert ei.value.code == capi.HAF_E_ARG and text in str(ei.value), (devices, str(ei.value))
    assert capi.multi_plan([0] * 13, capi.SHARD_CLOUDS, 12)["n_ranks"] == 1      # (clouds: any number of shards)


def test_create_multi_fails_loudly_without_a_device():
    """No GPU in this container: haf_create_multi gets as far as the first engine and reports HAF_E_DEVICE with the shard and the
    reason -- no CPU fallback, no half-built handle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from haf_grasping_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(root, "tests", "golden", "data")
    with pytest.raises(capi.HafError) as ei:
        capi.MultiEngine(os.path.join(data, "Features.txt"), os.path.join(data, "range21062012_allfeatures"),
                         os.path.join(root, "tests", "golden", "surrogate.model"), [0, 1], capi.SHARD_ROLLS)
    assert ei.value.code == capi.HAF_E_DEVICE and "shard 0" in str(ei.value)
