"""Iteration rate of N solver objects created one after another in one process (the placement of their arrays differs from object to object):
   python tools/objects_rate.py [float|double] [objects] [V]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
prec = sys.argv[1] if len(sys.argv) > 1 else "double"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
V = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
col, costs = random_set_cover_mt(V, V // 2, 10, 12345)
rates = []
for i in range(n):
    s = bdd_hip_parallel_mma(col, costs, precision=prec)
    s.iterations(300); s.synchronize()
    ms = s.time_iterations(1000)
    rates.append(1000 / ms * 1e3)
    s.close()
print(f"{os.environ.get('BDDMMA_LIB', 'shipped'):22s} {prec} V={V}: " + " ".join(f"{r:6.0f}" for r in rates) + f"   mean {sum(rates) / len(rates):6.0f}", flush=True)
