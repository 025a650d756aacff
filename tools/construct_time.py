"""Construction time of the 10.5 M-node benchmark instance on the GPU box: host layout (BDDMMA_LAYOUT_TIMING=1 prints the phases) +
upload, then checkpoint save / load."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BDDMMA_LAYOUT_TIMING", "1")
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
col, costs = random_set_cover_mt(1_000_000, 500_000, 10)
instr, delims = col.instr, col.delims   # materialise the flat arrays once (not part of construction)
for prec in ("float", "double"):
    for rep in range(2):
        t0 = time.perf_counter()
        s = bdd_hip_parallel_mma(col, costs, precision=prec)
        s.synchronize()
        print(f"{prec}: construct {time.perf_counter() - t0:.3f} s", flush=True)
    s.iterations(5)
    with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
        path = os.path.join(td, "ck.bin")
        t0 = time.perf_counter(); s.save(path); t1 = time.perf_counter()
        t = bdd_hip_parallel_mma.load(path); t.synchronize(); t2 = time.perf_counter()
        print(f"{prec}: save {t1 - t0:.3f} s ({os.path.getsize(path) / 1e6:.0f} MB), load {t2 - t1:.3f} s, lb equal: {t.lower_bound() == s.lower_bound()}", flush=True)
