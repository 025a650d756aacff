"""How long do the solve sweeps / the exchange take when other traffic has passed through the caches since the last iteration?
hipEvent times per kernel class of plain MMA iterations with (a) nothing, (b) a copy of N MB between iterations (torch, device-wide sync on both
sides).  python tools/cold_sweeps.py [float|double]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
prec = sys.argv[1] if len(sys.argv) > 1 else "float"
with_x = len(sys.argv) > 2 and sys.argv[2] == "x"   # an L-BFGS wrapper is attached: the backward solve sweeps also write x in layer order
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision=prec)
if with_x:
    from bdd_amd.solver import bdd_hip_lbfgs
    l = bdd_hip_lbfgs(s)
    l.iteration()
s.iterations(500); s.synchronize()
for mb in (0, 64, 128, 256, 512, 1024):
    a = torch.empty(max(mb, 1) * (1 << 20) // 2, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    s.set_profiling(True, stride=1)
    for _ in range(150):
        s.iteration(); s.synchronize()
        if mb:
            b.copy_(a); torch.cuda.synchronize()
    p = s.get_profile(); s.set_profiling(False)
    avg = [p["total_ms"][i] / max(p["launches"][i], 1) * 1e3 for i in range(3)]
    print(f"{prec}{' + x_layer' if with_x else ''}: {mb:5d} MB of copy traffic between iterations: fwd solve {avg[0]:6.1f} us  bwd solve {avg[1]:6.1f} us  exchange {avg[2]:6.1f} us", flush=True)
