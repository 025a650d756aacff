"""Iteration rate over the first seconds of the FIRST GPU process on a fresh box (10.5 M nodes, float): blocks of 256 iterations.
    python tools/first_process_ramp.py [seconds=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision="float")
s.synchronize()
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < secs:
    t = time.perf_counter()
    s.iterations(256)
    s.synchronize()
    dt = time.perf_counter() - t
    out.append((time.perf_counter() - t0, 256 / dt))
print(" ".join(f"{t:.2f}s:{r:.0f}" for t, r in out[:: max(1, len(out) // 40)]))
