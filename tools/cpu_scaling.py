"""Thread scaling of the CPU leg (oracle/mma_oracle.c, OpenMP over BDDs) on the GPU box's host cores, headline instance:
   python tools/cpu_scaling.py [float|double] [iterations per thread count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from oracle.oracle import Oracle
prec = sys.argv[1] if len(sys.argv) > 1 else "float"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, seed=12345)
ncpu = os.cpu_count()
print("cpus", ncpu, open("/proc/loadavg").read().strip(), flush=True)
o = Oracle(col, costs, prec, threads=min(ncpu, 32))
o.iteration()
for order in ((1, 8, 16, 32, 64, 96, 128, 192, 256), (128, 64, 32, 16, 96, 192)):
    for th in order:
        if th > ncpu:
            continue
        o.set_threads(th)
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); o.iteration(); ts.append(time.perf_counter() - t)
        print(f"threads {th:4d}: " + "  ".join(f"{1 / t:7.2f}" for t in ts) + " it/s", flush=True)
