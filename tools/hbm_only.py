"""The HBM-only figure bench.py quotes (roofline.infinity_cache_note.hbm_only_frac_whole_iteration): the 105 M-node instance (random set
cover k = 10, V = 10 M, B = 5 M; 4.5 / 7.2 GB resident: nothing of an iteration is served from the 256 MiB Infinity Cache), both precisions,
on the sources at hand.   python tools/hbm_only.py [out.json] [objects per precision]
Writes the JSON that is committed as profiles/rNN_hbm_only_105m.json (stamped with bench.source_hash())."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bdd_amd.instances import random_set_cover_mt, set_cover_sizes
from bdd_amd.solver import bdd_hip_parallel_mma

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hbm_only_105m.json"
objects = int(sys.argv[2]) if len(sys.argv) > 2 else 2
V, B, K = 10_000_000, 5_000_000, 10
sizes = set_cover_sizes(V, B, K)
col, costs = random_set_cover_mt(V, B, K, seed=12345)
res = {"workload": f"random set cover (mt19937_64 seed 12345), k = {K}, V = {V}, B = {B}: {sizes['N']} BDD nodes",
       "meaning": "frac_whole_iteration = B_iter x it/s / 8 TB/s (SURVEY §8d's algorithmic bytes); the best of `objects` solver objects per "
                  "precision (placement differs from object to object beyond the cache's reach), all listed",
       "_source_hash": bench.source_hash()}
names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exch_reduce"]
for prec, key, R in (("float", "f32", 4), ("double", "f64", 8)):
    rates, kernels, extra = [], None, {}
    for _ in range(objects):
        s = bdd_hip_parallel_mma(col, costs, precision=prec)
        s.iterations(20)
        s.synchronize()
        n = 100
        ms = s.time_iterations(n)
        rates.append(n / ms * 1e3)
        if kernels is None:
            kernels = {nm + " us": round(s.time_kernel(k, 10) * 1e3, 1) for k, nm in enumerate(names)}
            extra = {"hbm_resident_bytes": s.device_bytes(), "solve_sweeps": s.solve_sweep_kind(), "packs": s.nr_packs()}
        s.close()
    b_iter = bench.iteration_bytes(sizes, R)
    res[key] = round(b_iter * max(rates) / 1e9 / bench.HBM_PEAK_GBS, 4)
    res[key + "_detail"] = {"iterations_per_s": [round(r, 1) for r in rates], "B_iter": b_iter, "kernels_back_to_back": kernels, **extra}
    print(prec, rates, res[key], flush=True)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
