"""Per-wave phase timeline of the small-instance kernels from a -DBDDMMA_STAMPS build (tools/build_variant.sh stamps -DBDDMMA_STAMPS):
    BDDMMA_LIB=build/libstamps.so BDDMMA_STAMPS_FILE=gpurun_out/stamps python tools/stamps.py [--vars V --rows B | --matching N]
Prints, per kernel class, when the phases start and end relative to the first wave's entry (s_memtime ticks -> us at 100 MHz)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd.instances import random_set_cover_mt, assignment_ilp
from bdd_amd import to_bdd_collection
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=100_000)
ap.add_argument("--rows", type=int, default=50_000)
ap.add_argument("--matching", type=int, default=0)
ap.add_argument("--precision", default="float")
ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
if a.matching:
    ilp = assignment_ilp(a.matching)
    col, costs = to_bdd_collection(ilp), np.asarray(ilp.objective, float)
else:
    col, costs = random_set_cover_mt(a.vars, a.rows, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision=a.precision, variant_flags=a.variant)
s.iterations(20)
base = os.environ["BDDMMA_STAMPS_FILE"]
names = {2: "fwd_solve", 3: "bwd_solve", 4: "exchange"}
labels = {2: ["entry", "headers", "pack + pairs in LDS", "hop loop done", "flushed"], 4: ["entry", "first chunk arrived", "accumulated", "normalised", "pairs stored"]}
labels[3] = labels[2]
# streaming sweeps (k_fwd_narrow2 / k_bwd_narrow2): stamp 3 is taken before the workgroup's barrier, stamp 2 after it
stream_labels = ["entry", "deltas staged (workgroup)", "hops done, workgroup barrier passed", "hop loop done (this wave)", "flushed"]
for kind in (2, 3, 4):
    ms = s.time_kernel(kind, 50)
    z = np.loadtxt(f"{base}.{kind}", dtype=np.float64, ndmin=2)
    if z.size == 0:
        print(f"{names[kind]}: {ms * 1e3:.2f} us per launch, no stamps in this kernel (streaming sweeps)")
        continue
    t = z[:, 1:]
    t0 = t[:, 0].min()
    tick_us = 0.01   # s_memtime: 100 MHz constant clock
    print(f"{names[kind]}: {ms * 1e3:.2f} us per launch (50 launches back to back), {t.shape[0]} waves stamped")
    streaming = kind in (2, 3) and bool(np.all(t[:, 3] <= t[:, 2]))
    for i, lab in enumerate(stream_labels if streaming else labels[kind]):
        c = (t[:, i][t[:, i] > 0] - t0) * tick_us
        if c.size:
            print(f"   {lab:36s} first {c.min():6.2f}  median {np.median(c):6.2f}  last {c.max():6.2f} us after the first wave's entry")
    if streaming:
        order = [0, 1, 3, 2, 4]
        names_ = ["stage load + first prefetches", "hop loop", "wait for the workgroup", "flush"]
        for a_, b_, nm in zip(order[:-1], order[1:], names_):
            dph = (t[:, b_] - t[:, a_]) * tick_us
            print(f"   per wave: {nm:30s} median {np.median(dph):6.2f}  p10 {np.percentile(dph, 10):6.2f}  p90 {np.percentile(dph, 90):6.2f} us")
        life = (t[:, 4] - t[:, 0]) * tick_us
        span = (t[:, 4].max() - t0) * tick_us
        print(f"   per wave: entry -> flushed median {np.median(life):6.2f} us; all waves {life.sum():.0f} wave-us over {span:.1f} us = {life.sum() / span:.0f} waves in flight on average ({life.sum() / span / 256:.1f} per CU)")
        # in-flight waves over time, 10 samples
        for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
            x = t0 + frac * span / tick_us
            print(f"      at {frac * span:6.1f} us: {int(((t[:, 0] <= x) & (t[:, 4] > x)).sum())} waves resident, {int(((t[:, 1] <= x) & (t[:, 3] > x)).sum())} in their hop loop")
