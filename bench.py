#!/usr/bin/env python
"""bench.py — parallel-MMA iterations/s on the BASELINE.json workload, with roofline + CPU baseline.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one iteration() of the solver = forward_mm + normalize + backward_mm + normalize over
the whole instance (reference: bdd_cuda_parallel_mma.cu:142-153).  Workload: random set cover,
row size 10, V = 1e6, B = 5e5 -> 10.5 M BDD nodes (BASELINE.json configs[2], the configuration the
metric is quoted on).  Multi-GPU: independent instances, one per GPU, no collective on the data
path ("replicas only"); `value` = iterations of all ranks / max-over-ranks time.
Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "HBM3E peak BW")


def algorithmic_bytes_per_pass(sizes, R):
    """12 N' + 2R N + (5R+4) L' + (8R+4) V  (one forward_mm or backward_mm sweep)."""
    return 12 * sizes["N_nt"] + 2 * R * sizes["N"] + (5 * R + 4) * sizes["L_nt"] + (8 * R + 4) * sizes["V"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--precision", default="float", choices=["float", "double"])
    ap.add_argument("--vars", type=int, default=1_000_000)
    ap.add_argument("--rows", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--pack-width", type=int, default=0)
    ap.add_argument("--vars-per-bin", type=int, default=0)
    ap.add_argument("--stage-cap", type=int, default=0)
    ap.add_argument("--wpb", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--event-stride", type=int, default=64)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    from bdd_amd.instances import random_set_cover, set_cover_sizes
    from bdd_amd.solver import bdd_hip_parallel_mma

    sizes = set_cover_sizes(args.vars, args.rows, args.k)
    col, costs = random_set_cover(args.vars, args.rows, args.k, seed=12345 + rank)
    solver = bdd_hip_parallel_mma(col, costs, precision=args.precision, device=local_rank,
                                  pack_width=args.pack_width, deterministic=args.deterministic,
                                  vars_per_bin=args.vars_per_bin, stage_cap=args.stage_cap, waves_per_block=args.wpb)
    R = 4 if args.precision == "float" else 8
    solver.iterations(args.warmup)
    solver.synchronize()
    # hipEvent pairs around the launches of every --event-stride-th (64th) iteration, on the solver's own stream (an event pair
    # per launch costs ~4 us of stream time; at stride 1 the 10.5 M-node iteration is 14 % slower)
    solver.set_profiling(True, stride=args.event_stride)

    dt = timed_region(lambda: solver.iterations(args.steps),
                      lambda: (solver.synchronize(), torch.cuda.synchronize()), dist, local_rank)
    prof = solver.get_profile()
    solver.set_profiling(False)
    lb = solver.lower_bound()

    triad_gbs = copy_gbs = lb_rate = None
    if rank == 0:
        # outside the timed region: (1) the same loop with the lower bound fetched every iteration, as
        # run_solver does (one extra plain backward sweep + reduce + 8-byte D2H per iteration); (2) the
        # STREAM-triad bandwidth of this box (3 x 1 GiB per launch), the measured ceiling next to the 8 TB/s spec
        n_lb = min(args.steps, 100)
        solver.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_lb):
            solver.iteration()
            solver.lower_bound()
        lb_rate = n_lb / (time.perf_counter() - t0)
        triad_gbs = 3 * (1 << 30) / (solver.time_kernel(6, 20) * 1e-3) / 1e9
        copy_gbs = 2 * (1 << 30) / (solver.time_kernel(7, 20) * 1e-3) / 1e9

    if rank == 0:
        its = aggregate_rate(world, args.steps, dt)
        names = ["forward_mm", "backward_mm", "finish_delta", "other"]
        avg_ms = [prof["total_ms"][i] / max(prof["launches"][i], 1) for i in range(4)]
        dom = 0 if avg_ms[0] >= avg_ms[1] else 1
        bytes_pass = algorithmic_bytes_per_pass(sizes, R)
        achieved = bytes_pass / (avg_ms[dom] * 1e-3) / 1e9 if avg_ms[dom] > 0 else 0.0
        out = {
            "metric": "parallel-MMA iterations/sec, 10M BDD nodes, 1 GPU (+ achieved HBM GB/s in roofline)",
            "value": its,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "float" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"random set cover, row size {args.k}, V={args.vars}, B={args.rows}: "
                            f"{sizes['N']} BDD nodes (BASELINE.json configs[2]); one independent instance per GPU",
                "precision": args.precision,
                "omega": 0.5,
                "pack_width": args.pack_width or "auto (128; 64 when fewer than 4000 packs)",
                "waves_per_block": args.wpb or 4,
                "packs": solver.nr_packs(),
                "hops": solver.nr_hops(),
                "delta_exchange": "per-variable gather (deterministic)" if args.deterministic else "binned exchange, LDS accumulators",
                "hbm_resident_bytes": solver.device_bytes(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(names[dom], args),
                "algorithmic_bytes_per_launch": bytes_pass,
                "avg_launch_ms": {names[i]: avg_ms[i] for i in range(3)},
                "timed_launches": {names[i]: prof["launches"][i] for i in range(3)},
                "whole_iteration_GBs": 2 * bytes_pass * its / world / 1e9,
                "stream_triad_GBs": triad_gbs,
                "stream_copy_GBs": copy_gbs,
                "frac_of_stream_triad": achieved / triad_gbs if triad_gbs else None,
            },
            "value_with_lower_bound_every_iteration": lb_rate,
            "lower_bound_after": {"iterations": args.warmup + args.steps, "value": lb},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(col, costs, args, sizes)
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


def timed_region(run_steps, device_sync, dist, local_rank, backend_device=True):
    """barrier + device sync, EXACTLY the K steps, device sync + barrier; returns the MAX over ranks of the
    wall time (the driver's contract).  `dist` is torch.distributed or None (single process)."""
    import torch

    def barrier():
        if dist is not None:
            if backend_device:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    barrier()
    device_sync()
    t0 = time.perf_counter()
    run_steps()
    device_sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_rate(world, steps, dt):
    """whole-job throughput: every rank runs its own instance (replicas, no data-path collective)"""
    return world * steps / dt


def measured_traffic(kernel, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/<tag>/traffic.json, produced by tools/profile.sh + tools/collect_profiles.py: separate --pmc runs,
    bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md §HBM).  None when no profile matches."""
    if (args.vars, args.rows, args.k) != (1_000_000, 500_000, 10) or args.deterministic:
        return None
    tag = "r01_f32" if args.precision == "float" else "r01_f64"
    path = os.path.join(ROOT, "profiles", tag, "traffic.json")
    if not os.path.exists(path):
        return None
    want = "k_fwd_narrow" if kernel == "forward_mm" else "k_bwd_narrow"
    for name, v in json.load(open(path)).items():
        m = re.search(want + r"<\w+, \d+, (\d+), \d+>", name)   # <REAL, R, MODE, waves per block>; MODE 1 = solve
        if m and m.group(1) == "1":
            return v["hbm_bytes"]
    return None


def cpu_baseline(col, costs, args, sizes):
    """The CPU restatement of the reference's `parallel mma` (oracle/, OpenMP over BDDs) timed on this
    box's host cores on the same instance — rank 0, bounded to ~args.cpu_seconds of work."""
    from bdd_amd.solver import bdd_hip_parallel_mma
    from oracle.oracle import Oracle

    # pick the thread count that is fastest on this box (the per-BDD work items are tiny and the delta
    # accumulation uses atomics, so all 256 hardware threads of a 2-socket host are not the optimum)
    ncpu = os.cpu_count() or 1
    o = Oracle(col, costs, args.precision, threads=min(ncpu, 32))
    o.iteration()  # warm-up (first touch, backward_run)
    best, cores = 0.0, min(ncpu, 32)
    for th in sorted({min(ncpu, t) for t in (16, 32, 64, 128, 256)}):
        o.set_threads(th)
        t1 = time.perf_counter()
        o.iteration()
        rate = 1.0 / (time.perf_counter() - t1)
        if rate > best:
            best, cores = rate, th
    o.set_threads(1)
    t1 = time.perf_counter()
    o.iteration()
    single = 1.0 / (time.perf_counter() - t1)
    o.set_threads(cores)
    warm = 2 + len({min(ncpu, t) for t in (16, 32, 64, 128, 256)})
    n, t0 = 0, time.perf_counter()
    while True:
        o.iteration()
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 200:
            break
    cpu_lb = o.lower_bound()
    g = bdd_hip_parallel_mma(col, costs, precision=args.precision, pack_width=args.pack_width,
                             vars_per_bin=args.vars_per_bin, stage_cap=args.stage_cap, waves_per_block=args.wpb)
    g.iterations(n + warm)
    gpu_lb = g.lower_bound()
    return {
        "value": n / el,
        "unit": "iterations/s",
        "cores": cores,
        "kind": "port",
        "single_thread_value": single,
        "value_per_core": n / el / cores,
        "sample": f"{n} iterations of the same {sizes['N']}-node instance after {warm} warm-up / thread-count-probe "
                  f"iterations, oracle/mma_oracle.c with OpenMP over BDDs ({cores} of {ncpu} hardware threads: the "
                  f"fastest of 16..256), {args.precision}",
        "lb_after": {"iterations": n + warm, "cpu": cpu_lb, "gpu": gpu_lb,
                     "rel_diff": abs(cpu_lb - gpu_lb) / max(abs(cpu_lb), 1e-300)},
    }


if __name__ == "__main__":
    main()
