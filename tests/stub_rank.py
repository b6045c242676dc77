#!/usr/bin/env python3
"""A rank program for tests/test_distributed_cpu.py: what bench.py's ranks do around the timed region -- rendezvous from the
launcher's environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), the per-step election collective, the contract's reductions
(bench.reduce_over_ranks), rank 0 printing the ONE JSON line with the `ranks` block (bench.ranks_block) -- on the gloo backend with
made-up timings instead of a GPU.  --fail-rank R --fail-code C: rank R exits with C before the collectives (the launcher must
report C and stop the others)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--shard", default="clouds")
    ap.add_argument("--fail-rank", type=int, default=-1)
    ap.add_argument("--fail-code", type=int, default=7)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == a.gpus and int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if rank == a.fail_rank:
        sys.stderr.write("stub rank %d: failing on purpose\n" % rank)
        sys.exit(a.fail_code)
    import torch
    import torch.distributed as dist
    import bench
    from haf_grasping_amd import distributed as hd
    dist.init_process_group("gloo")
    coll = []
    for _ in range(a.steps):                                # the step's one collective: the 8-byte election of the best grasp
        t0 = time.perf_counter()
        vote, tag = hd.best_of_batch(100 + rank, tag=rank)
        coll.append(1e6 * (time.perf_counter() - t0))
        assert (vote, tag) == (100 + world - 1, world - 1)
    elapsed = 0.010 * a.steps * (1.0 + 0.01 * rank)          # rank r is r % slower: the MAX must be the last rank's
    evals = 1000 * a.steps
    el, tot, per_rank = bench.reduce_over_ranks(dist, torch, "cpu", world, elapsed, evals, 1e3 * elapsed / a.steps, sorted(coll)[len(coll) // 2], 7.0 + rank)
    if rank == 0:
        line = {"metric": "stub", "value": tot / el, "n_gpus": world, "steps": a.steps, "ms_per_step": 1e3 * el / a.steps,
                "ranks": bench.ranks_block(world, per_rank, a.shard, True, dist, "gloo")}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
