"""Host CPU spent enqueueing the iteration loop (4 launches per iteration): process / thread CPU time against wall time.
    python tools/host_cpu.py [--vars V --rows B] [--precision float]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=500_000)
ap.add_argument("--precision", default="float")
ap.add_argument("--iters", type=int, default=20000)
a = ap.parse_args()
col, costs = random_set_cover_mt(a.vars, a.rows, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision=a.precision)
s.iterations(500); s.synchronize()
w0, c0, t0 = time.perf_counter(), time.process_time(), time.thread_time()
s.iterations(a.iters); s.synchronize()
w, c, t = time.perf_counter() - w0, time.process_time() - c0, time.thread_time() - t0
print(f"{col.nr_bdd_nodes()} nodes, {a.precision}: {1e6 * w / a.iters:.1f} us wall per iteration ({a.iters / w:.0f} it/s); calling thread {1e6 * t / a.iters:.1f} us CPU per iteration "
      f"= {100 * t / w:.0f} % of a core ({1e6 * t / a.iters / 4:.1f} us per launch); whole process {100 * c / w:.0f} % of a core")
