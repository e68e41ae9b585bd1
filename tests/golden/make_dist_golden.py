#!/usr/bin/env python3
"""Golden vector for the data-parallel exchange step from the REFERENCE's own code (build container only).

Run from the repo root:  python tests/golden/make_dist_golden.py
Spawns 2 CPU ranks over gloo (127.0.0.1), imports /root/reference/distributed.py at run time (never copied) and runs
tests/dist_case.py through its ``apply_gradient_allreduce`` / ``reduce_tensor`` (distributed.py:42-46, 95-147).  Writes
tests/golden/dist_allreduce.npz: per rank the parameters after the start-up broadcast, the averaged gradients, the
(unsynchronised) BatchNorm running mean and the reduced scalar.  ``init_distributed`` asserts CUDA (distributed.py:49)
and cannot run here; the process group is created directly.
"""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, REF)
    import dist_case
    import distributed as ref_dist          # the reference's module
    assert ref_dist.__file__.startswith(REF)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = dist_case.run(rank, world, ref_dist.apply_gradient_allreduce, ref_dist.reduce_tensor)
    q.put((rank, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    world, port = 2, 29400 + os.getpid() % 500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    arrs = {}
    for r in range(world):
        for k, v in res[r].items():
            arrs["r%d_%s" % (r, k)] = np.asarray(v)
    path = os.path.join(HERE, "dist_allreduce.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: v.shape for k, v in arrs.items()})
