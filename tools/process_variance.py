"""Why do bench.py processes differ (7.0-8.2 k it/s on one box) when a bare solver process is steady at 8.2 k?
One process: [optionally import torch / init its device], build the 10.5 M-node solver, time 1024 iterations, twice.
    python tools/process_variance.py [none|import|device|alloc]"""
import os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode in ("device", "alloc"):
        torch.cuda.set_device(0)
        torch.cuda.synchronize()
    if mode == "alloc":
        x = torch.zeros(1 << 20, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
for rep in range(2):
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    s.iterations(512); s.synchronize()
    t = time.perf_counter(); s.iterations(1024); s.synchronize()
    r = 1024 / (time.perf_counter() - t)
    k = [s.time_kernel(i, 20) * 1e3 for i in (2, 3, 4)]
    del s
    print(f"  {mode} build {rep}: {r:.0f} it/s  fwd {k[0]:.1f} bwd {k[1]:.1f} exch {k[2]:.1f} us", flush=True)
