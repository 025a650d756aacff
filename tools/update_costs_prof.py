import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision="float")
z = np.zeros(s.nr_variables())
for _ in range(20):
    s.update_costs(z, z * 0 + 1e-9)
print(s.lower_bound())
