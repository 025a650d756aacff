"""The N>1 path of bench.py (replicas, one process per GPU): barrier / max-over-ranks timing with world_size 2
over gloo on CPU.  The solver itself needs a GPU, so the step is a stand-in with a rank-dependent duration."""
import os
import socket
import sys
import time

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    steps = 5
    per_step = 0.02 * (rank + 1)  # rank 1 is twice as slow: the reported time must be ITS time
    dt = bench.timed_region(lambda: time.sleep(steps * per_step), lambda: None, dist, 0, backend_device=False)
    out[rank] = (dt, bench.aggregate_rate(world, steps, dt))
    dist.destroy_process_group()


def test_two_rank_timing_takes_the_max_and_aggregates():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (dt0, r0), (dt1, r1) = out[0], out[1]
    assert abs(dt0 - dt1) < 1e-9                       # every rank holds the same (max) time
    assert 0.2 <= dt0 < 0.6                            # 5 steps x 0.04 s of the slow rank, not 0.1 s of the fast one
    assert r0 == pytest.approx(2 * 5 / dt0)            # whole-job rate: all ranks' steps / max time


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    from bdd_amd.instances import set_cover_sizes
    sz = set_cover_sizes(1_000_000, 500_000, 10)
    assert sz["N"] == 10_500_000
    assert bench.algorithmic_bytes_per_pass(sz, 4) == 354_000_000   # B_iter = 708 MB (float)
    assert bench.algorithmic_bytes_per_pass(sz, 8) == 570_000_000   # 1140 MB (double)
