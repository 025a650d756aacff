"""Is the iteration loop ever short of queued work?  Host time to ENQUEUE n iterations (bddmma_iterations returns when the launches are queued) against
the time until the stream is empty, in windows, with the enqueue thread's CPU time beside it.
    python tools/enqueue_probe.py [n per window = 500] [windows = 12]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision="float")
s.iterations(50); s.synchronize()
for w in range(W):
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    s.iterations(n)
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    print(f"window {w}: enqueue {1e3 * (t1 - t0):7.1f} ms, drained after {1e3 * (t2 - t0):7.1f} ms = {n / (t2 - t0):6.0f} it/s; cpu {1e3 * cpu:6.1f} ms, "
          f"involuntary switches {r1.ru_nivcsw - r0.ru_nivcsw}, voluntary {r1.ru_nvcsw - r0.ru_nvcsw}", flush=True)
