import sys, os, time
sys.path.insert(0, os.getcwd())
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
t00 = time.time()
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision="float")
print("construct done at", round(time.time() - t00, 1), "s")
t0 = time.time()
while time.time() - t0 < 45:
    ms = s.time_iterations(2000)
    print(f"t={time.time() - t0:5.1f}s  {2000 / ms * 1e3:7.0f} it/s")
