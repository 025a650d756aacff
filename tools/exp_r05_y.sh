#!/bin/bash
# round 5, last session: is the float headline's box-to-box spread (7.0-8.8 k it/s, double steady) a property of the box or of an allocation?
# six fresh processes on one box, then six solvers in one process
for i in 1 2 3 4 5 6; do timeout 120 python tools/kbench.py --mt 1 --iters 400 2>&1 | tail -2 | tr '\n' ' '; echo; done
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
keep = []
for i in range(6):
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    s.iterations(20)
    ms = s.time_iterations(400)
    print(f"solver {i} in one process: {400 / ms * 1e3:.0f} it/s  fwd/bwd solve {s.time_kernel(2, 20) * 1e3:.1f} / {s.time_kernel(3, 20) * 1e3:.1f} us", flush=True)
    if i % 2 == 0: keep.append(s)   # every other solver stays alive: the next one gets other addresses
PY
