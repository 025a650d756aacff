import sys, os
sys.path.insert(0, ".")
from bdd_amd.instances import random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma
import numpy as np
from bdd_amd import BddCollection
rng = np.random.Generator(np.random.PCG64(5))
col, costs = random_set_cover(120, 90, 5, seed=4)
# a few knapsack rows so that the file has wide packs too
for _ in range(3):
    co = rng.integers(1, 40, size=14)
    col.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(120, size=14, replace=False)))
costs = np.concatenate([costs, np.zeros(col.nr_variables() - len(costs))])
s = bdd_hip_parallel_mma(col, costs, precision="double", waves_per_block=2, pack_width=64, pack_stagger=30)
s.iterations(3)
s.save("gpurun_out/checkpoint_small_v07.bin")
print(os.path.getsize("gpurun_out/checkpoint_small_v07.bin"), s.lower_bound(), s.nr_packs())
