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
    dt = bench.timed_region(lambda: time.sleep(steps * per_step), lambda: None, dist)   # exactly what bench.py does at N > 1: gloo
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
    assert bench.iteration_bytes(sz, 4) == 708_000_000     # B_iter = 708 MB (float)
    assert bench.iteration_bytes(sz, 8) == 1_140_000_000   # 1140 MB (double)
    assert bench.sweep_bytes(sz, 4) == 318_000_000 and bench.exchange_bytes(sz, 4) == 36_000_000   # what each launch processes
    assert bench.sweep_bytes(sz, 8) == 502_000_000 and bench.exchange_bytes(sz, 8) == 68_000_000


def test_labels_follow_the_sizes_and_traffic_is_stamped():
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    from bdd_amd.instances import set_cover_sizes
    a = argparse.Namespace(vars=100_000, rows=50_000, k=10, deterministic=False, pack_width=0, wpb=0, vars_per_bin=0, stage_cap=0)
    assert "configs[1]" in bench.workload_name(set_cover_sizes(100_000, 50_000, 10), a)
    a10 = argparse.Namespace(**{**vars(a), "vars": 1_000_000, "rows": 500_000})
    assert "configs[2]" in bench.workload_name(set_cover_sizes(1_000_000, 500_000, 10), a10)
    odd = argparse.Namespace(**{**vars(a), "vars": 3000, "rows": 2000})
    assert "not a BASELINE.json configuration" in bench.workload_name(set_cover_sizes(3000, 2000, 10), odd)
    assert bench.nodes_label(10_500_000) == "10.5M" and bench.nodes_label(1_050_000) == "1.05M"
    # committed PMC traffic is only quoted while the kernel sources it was measured on are unchanged
    t, t_exch, src, us_rocprof = bench.measured_traffic("forward_mm", a10, "f32")
    import json
    stamped = []
    import glob
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_10m_f32", "traffic.json")):   # every round's committed counters
        stamped.append(json.load(open(path)).get("_source_hash"))
    assert (t is not None) == (bench.source_hash() in stamped)
    if t is not None:
        assert src.startswith("profiles/") and 100e6 < t < 400e6 and (t_exch is None or 30e6 < t_exch < 200e6)
        assert us_rocprof is None or 20.0 < us_rocprof < 200.0   # rocprofv3 average of the same kernel, stored beside its bytes since r6


def test_lbfgs_bytes_and_hbm_only_source():
    """the L-BFGS line carries a roofline on SURVEY §8(d)'s (2m + 6) R L' + sweep bytes; the HBM-only fraction quoted
    in the line comes from the newest round's committed file."""
    sys.path.insert(0, ROOT)
    import glob
    import bench
    from bdd_amd.instances import set_cover_sizes
    sz = set_cover_sizes(1_000_000, 500_000, 10)
    extra = 12 * sz["N_nt"] + 2 * 4 * sz["N"] + 2 * 4 * sz["L_nt"]
    assert bench.lbfgs_bytes(sz, 4, 5, 1.0) == 708_000_000 + 16 * 4 * sz["L_nt"] + 2 * extra
    assert bench.lbfgs_bytes(sz, 8, 5, 1.25) == 1_140_000_000 + 16 * 8 * sz["L_nt"] + 2.25 * (12 * sz["N_nt"] + 16 * sz["N"] + 16 * sz["L_nt"])
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_only_105m.json")))[-1]
    h = bench.hbm_only_fractions()
    assert h["source"] == "profiles/" + os.path.basename(newest) and 0.3 < h["f32"] < 1.0 and 0.3 < h["f64"] < 1.0
