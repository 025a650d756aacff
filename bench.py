#!/usr/bin/env python
"""bench.py — parallel-MMA iterations/s on the BASELINE.json workload, with roofline + CPU baseline.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one iteration() of the solver = forward_mm + normalize + backward_mm + normalize over
the whole instance (reference: bdd_cuda_parallel_mma.cu:142-153).  Default workload: random set cover,
row size 10, V = 1e6, B = 5e5 -> 10.5 M BDD nodes (BASELINE.json configs[2], the configuration the
metric is quoted on), rows and costs from std::mt19937_64(12345 + rank) (bdd_amd/csrc/host/instances.cpp).
`value` is the float run; the double run of the same instance ("double vs float") is timed in the same
invocation and reported as `value_f64` / `roofline_f64`.
Multi-GPU: independent instances, one per GPU, no collective on the data path ("replicas only",
SURVEY.md §8e); the timing barrier and the max-over-ranks go over gloo (CPU), RCCL is never initialised;
`value` = iterations of all ranks / max-over-ranks time.  Inputs are resident in HBM before the timed region.
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "HBM3E peak BW")
KERNEL_SOURCES = ("bdd_amd/csrc/kernels.hpp", "bdd_amd/csrc/kernels/common.hpp", "bdd_amd/csrc/kernels/narrow.hpp", "bdd_amd/csrc/kernels/resident.hpp",
                  "bdd_amd/csrc/kernels/narrow2.hpp", "bdd_amd/csrc/kernels/narrow3.hpp", "bdd_amd/csrc/kernels/wide.hpp", "bdd_amd/csrc/kernels/exchange.hpp",
                  "bdd_amd/csrc/kernels/small.hpp",
                  "bdd_amd/csrc/kernels/elementwise.hpp", "bdd_amd/csrc/solver_impl.hpp", "bdd_amd/csrc/layout.cpp")


def sweep_bytes(sizes, R):
    """Algorithmic bytes ONE sweep launch (k_fwd_narrow / k_bwd_narrow) processes — SURVEY.md §8d's per-pass figure without
    the per-variable term, which the exchange launch moves: 12 N' + 2R N + (5R+4) L'."""
    return 12 * sizes["N_nt"] + 2 * R * sizes["N"] + (5 * R + 4) * sizes["L_nt"]


def exchange_bytes(sizes, R):
    """The per-variable term of SURVEY.md §8d, processed by one k_exchange_reduce launch: (8R+4) V."""
    return (8 * R + 4) * sizes["V"]


def iteration_bytes(sizes, R):
    """B_iter of SURVEY.md §8d = 2 [12 N' + 2R N + (5R+4) L' + (8R+4) V]."""
    return 2 * (sweep_bytes(sizes, R) + exchange_bytes(sizes, R))


def source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def workload_name(sizes, args):
    n = sizes["N"]
    cfg = {1_050_000: "configs[1]", 10_500_000: "configs[2]"}.get(n) if args.k == 10 else None
    label = f"BASELINE.json {cfg}" if cfg else "not a BASELINE.json configuration"
    return (f"random set cover (std::mt19937_64 seed 12345 + rank), row size {args.k}, V={args.vars}, B={args.rows}: "
            f"{n} BDD nodes ({label}); one independent instance per GPU")


def nodes_label(n):
    return f"{n / 1e6:.3g}M".replace(".0M", "M")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--clock-warm", type=float, default=0.3, dest="clock_warm",
                    help="seconds of untimed iterations before the warm-up steps, so that the shader clocks are up (0: none)")
    ap.add_argument("--precision", default="float", choices=["float", "double"], help="precision of `value` (the other one is value_f64 / value_f32)")
    ap.add_argument("--vars", type=int, default=1_000_000)
    ap.add_argument("--rows", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--pack-width", type=int, default=0)
    ap.add_argument("--vars-per-bin", type=int, default=0)
    ap.add_argument("--stage-cap", type=int, default=0)
    ap.add_argument("--wpb", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-precision", action="store_true")
    ap.add_argument("--no-lbfgs", action="store_true", dest="no_lbfgs", help="skip the L-BFGS leg (BASELINE.json configs[3]) after the timed region")
    ap.add_argument("--no-configs", action="store_true", dest="no_configs", help="skip the BASELINE.json configs[1] leg (1.05 M nodes) after the timed region")
    ap.add_argument("--repeats", type=int, default=4, help="further timed regions of exactly K steps after the one `value` is taken from (their rates: repeat_samples)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--event-stride", type=int, default=0, help="hipEvent pairs around every n-th iteration's launches (0: steps // 16)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("BDDMMA_BENCH_SHARE_GPUS") == "1":
        # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks then share devices: the numbers mean nothing)
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:   # the layout of every rank's instance is built on the host cores: each rank takes its share (default: up to 32 threads)
        os.environ.setdefault("BDDMMA_THREADS", str(max(1, min(32, (os.cpu_count() or 1) // world))))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # replicas only: the barrier and the max are host-side, no RCCL communicator is created.  Gloo announces its connections on
        # stdout ("[Gloo] Rank 0 is connected to ..."): stdout is the JSON line's, so it is pointed at stderr while the group forms
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from bdd_amd.instances import random_set_cover_mt, set_cover_sizes
    from bdd_amd.solver import bdd_hip_parallel_mma

    sizes = set_cover_sizes(args.vars, args.rows, args.k)
    col, costs = random_set_cover_mt(args.vars, args.rows, args.k, seed=12345 + rank)
    stride = args.event_stride or max(1, args.steps // 16)

    pre_iterations = {}

    def run(precision):
        """warm-up, the timed K steps (barrier + device sync on both sides), hipEvent profile, lower bound"""
        solver = bdd_hip_parallel_mma(col, costs, precision=precision, device=local_rank,
                                      pack_width=args.pack_width, deterministic=args.deterministic,
                                      vars_per_bin=args.vars_per_bin, stage_cap=args.stage_cap, waves_per_block=args.wpb)
        # Clock ramp: an idle MI355X sits at its lowest shader clock (rocm-smi: sclk 98 MHz) and needs tens of milliseconds of load to
        # come up; W = 5 warm-up steps are 0.6 ms and a K = 20 run 2.5 ms, and such runs measured 6.5 k instead of 7.9 k it/s from time
        # to time.  So the device is kept busy with the same iterations for args.clock_warm seconds first (untimed, counted in
        # lower_bound_after.iterations), then come the W warm-up steps and the K timed ones.
        pre = 0
        t_pre = time.perf_counter()
        if "n" in pre_iterations:           # the second precision of the run: the same number of iterations as the first
            pre = pre_iterations["n"]
            if pre:
                solver.iterations(pre)
                solver.synchronize()
        else:
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < args.clock_warm:
                solver.iterations(64)
                solver.synchronize()
                pre += 64
            pre_iterations["n"] = pre
            pre_iterations["s"] = time.perf_counter() - t_pre
        solver.iterations(args.warmup)
        solver.synchronize()
        dt = timed_region(lambda: solver.iterations(args.steps),
                          lambda: (solver.synchronize(), torch.cuda.synchronize()), dist)
        # `value` is that region.  K = 20 steps are a 2.3 ms sample: the same region again, a few times, shows its spread (repeat_samples)
        repeats = [timed_region(lambda: solver.iterations(args.steps), lambda: (solver.synchronize(), torch.cuda.synchronize()), dist)
                   for _ in range(max(0, args.repeats))]
        # Kernel durations: the same K steps once more with hipEvent pairs on the solver's own stream around the launches of every
        # `stride`-th iteration (>= 16 samples per kernel class for any K >= 16).  An event pair per launch costs ~4 us of stream
        # time (-14 % it/s at stride 1), so the instrumented pass is kept out of `value`; its own rate is reported next to it.
        solver.set_profiling(True, stride=stride)
        dt_ev = timed_region(lambda: solver.iterations(args.steps),
                             lambda: (solver.synchronize(), torch.cuda.synchronize()), dist)
        prof = solver.get_profile()
        prof["ms_per_step_instrumented"] = dt_ev / args.steps * 1e3
        prof["repeat_dt"] = repeats
        solver.set_profiling(False)
        return solver, dt, prof, solver.lower_bound()

    other = "double" if args.precision == "float" else "float"
    # the secondary precision runs first: measured on this box, a double solver created after a float one (freed) has been run is
    # 4 % slower (3 855 vs 4 027 it/s), while the float run does not care about the order (7 787 / 7 791)
    second = None
    if not args.no_second_precision:
        s2, dt2, prof2, lb2 = run(other)
        second = (dt2, prof2, lb2, s2.device_bytes())
        s2.close()
    solver, dt, prof, lb = run(args.precision)

    triad_gbs = copy_gbs = lb_rate = lb_rate_host_loop = None
    if rank == 0:
        # outside the timed region: (1) the same loop with the lower bound fetched every iteration, as run_solver does
        # (one extra plain backward sweep + reduce + 8-byte D2H per iteration); (2) STREAM triad / copy of this box
        n_lb = min(args.steps, 100)
        solver.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_lb):
            solver.iteration()
            solver.lower_bound()
        lb_rate_host_loop = n_lb / (time.perf_counter() - t0)
        # ... and bddmma_run_solver itself (termination tests on the device, the host only reads the published bounds)
        from bdd_amd.solver import run_solver
        n_rs = max(200, min(args.steps, 500))
        run_solver(solver, max_iter=2, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)  # allocates its control block
        rs = run_solver(solver, max_iter=n_rs, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
        lb_rate = rs["iterations"] / rs["seconds"]
        triad_gbs = 3 * (1 << 30) / (solver.time_kernel(6, 20) * 1e-3) / 1e9
        copy_gbs = 2 * (1 << 30) / (solver.time_kernel(7, 20) * 1e-3) / 1e9
    packs, hops, resident = solver.nr_packs(), solver.nr_hops(), solver.device_bytes()
    sweep_kind = solver.solve_sweep_kind()
    solver.close()
    lbfgs = None
    if rank == 0 and not args.no_lbfgs:
        lbfgs = {p: lbfgs_rate(col, costs, p, local_rank, args, sizes) for p in ([args.precision] if args.no_second_precision else [args.precision, other])}
    configs = None
    if rank == 0 and not args.no_configs and (args.vars, args.rows, args.k) == (1_000_000, 500_000, 10):
        configs = {"1m_f32": small_config(local_rank, "float"), "1m_f64": small_config(local_rank, "double"),
                   "matching_8x8_f64": matching_config(local_rank, "double"), "matching_8x8_f32": matching_config(local_rank, "float")}

    if rank == 0:
        its = aggregate_rate(world, args.steps, dt)
        R = 4 if args.precision == "float" else 8
        sfx = "f32" if args.precision == "float" else "f64"
        out = {
            "metric": f"parallel-MMA iterations/sec, {nodes_label(sizes['N'])} BDD nodes, {world} GPU (+ achieved HBM GB/s in roofline)",
            "value": its,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": sfx,
            "data": "synthetic",
            "config": {
                "workload": workload_name(sizes, args),
                "precision": args.precision,
                "omega": 0.5,
                "pack_width": args.pack_width or "auto (128; 64 when fewer than 2048 packs)",
                "waves_per_block": args.wpb or "auto",
                "packs": packs,
                "solve_sweeps": sweep_kind,   # include/bdd_mma.h: BDDMMA_SWEEPS_* (streaming3 = a lane per layer, kernels/narrow3.hpp)
                "hops": hops,
                "delta_exchange": "per-variable gather (deterministic)" if args.deterministic else "binned exchange, LDS accumulators",
                "hbm_resident_bytes": resident,
                "event_stride": stride,
            },
            "roofline": roofline(prof, sizes, R, its / world, args, sfx, triad_gbs, copy_gbs, resident),
            "value_with_lower_bound_every_iteration": lb_rate,
            "value_with_lower_bound_every_iteration_host_loop": lb_rate_host_loop,
            "lower_bound_after": {"iterations": pre_iterations.get("n", 0) + args.warmup + (2 + max(0, args.repeats)) * args.steps, "value": lb},
            # what ran, untimed, before the K timed steps: the clock-warm loop (seconds / iterations of the same step), then the W warm-up steps
            "clock_warm_iterations": pre_iterations.get("n", 0),
            "clock_warm_s": pre_iterations.get("s", 0.0),
            "untimed_iterations_before_timed_region": pre_iterations.get("n", 0) + args.warmup,
            # the timed region of `value` repeated: rates of `repeats` further regions of exactly K steps each (same barriers and syncs)
            "repeat_samples": [aggregate_rate(world, args.steps, t) for t in prof.get("repeat_dt", [])],
        }
        if second is not None:
            dt2, prof2, lb2, _res2 = second
            R2, sfx2 = (8, "f64") if other == "double" else (4, "f32")
            its2 = aggregate_rate(world, args.steps, dt2)
            out["value_" + sfx2] = its2
            out["ms_per_step_" + sfx2] = dt2 / args.steps * 1e3
            out["roofline_" + sfx2] = roofline(prof2, sizes, R2, its2 / world, args, sfx2, triad_gbs, copy_gbs, second[3])
            out["lower_bound_after_" + sfx2] = {"iterations": pre_iterations.get("n", 0) + args.warmup + (2 + max(0, args.repeats)) * args.steps, "value": lb2,
                                               "rel_diff_to_" + sfx: abs(lb2 - lb) / max(abs(lb), 1e-300)}
        if lbfgs is not None:
            out["lbfgs"] = {"what": "BASELINE.json configs[3]: L-BFGS around the same solver on the same instance (history 5, the reference's default "
                                    "parameters), fresh solver, 20 untimed iterations that fill the history, then the timed ones; outside the timed "
                                    "region of `value`", **lbfgs}
        if configs is not None:
            out["configs"] = {"what": "the other single-GPU BASELINE.json configurations, outside the timed region of `value`: configs[1] = random set cover "
                                      "k = 10, V = 1e5, B = 5e4 (1.05 M BDD nodes), same generator and seed; 2 000 timed iterations after 0.1 s of "
                                      "clock-warm iterations; per-kernel times from hipEvent pairs on the solver's stream in a second pass.  configs[0] = "
                                      "the 8 x 8 bipartite matching of test_bdd_bipartite_matching_problem.cpp (16 BDDs, one workgroup: whole iterations "
                                      "inside one launch, csrc/kernels/small.hpp) — iterations(n) and run_solver with its tests on the device", **configs}
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(col, costs, args, sizes)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def lbfgs_bytes(sizes, R, m, trials):
    """Algorithmic bytes of one L-BFGS iteration, SURVEY.md §8(d): the MMA iteration (B_iter) + the vector algebra of the two-loop
    recursion, (2m + 6) R L' (L' = nr_layers: the dual vectors' length) + one extra sweep per lower_bound / bdds_solution evaluation —
    `trials` bound evaluations (one per trial step) and one argmin-path sweep for the subgradient; an extra sweep reads the node
    indices, the {lo, hi} costs and moves the potentials once each way: 12 N' + 2R N + 2R L'."""
    extra = 12 * sizes["N_nt"] + 2 * R * sizes["N"] + 2 * R * sizes["L_nt"]
    return iteration_bytes(sizes, R) + (2 * m + 6) * R * sizes["L_nt"] + (trials + 1.0) * extra


def small_config(device, precision, vars_=100_000, rows=50_000, k=10, steps=2000):
    """BASELINE.json configs[1] (1.05 M nodes) on the same box, outside the timed region: rate, whole-iteration roofline fraction and
    the hipEvent time of each kernel class (BASELINE.md §3 quotes >= 45 k it/s / 0.40 for it; the driver's line never ran it before r6)."""
    from bdd_amd.instances import random_set_cover_mt, set_cover_sizes
    from bdd_amd.solver import bdd_hip_parallel_mma
    sizes = set_cover_sizes(vars_, rows, k)
    col, costs = random_set_cover_mt(vars_, rows, k, seed=12345)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, device=device)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        s.iterations(256)
        s.synchronize()
    s.synchronize()
    t0 = time.perf_counter()
    s.iterations(steps)
    s.synchronize()
    dt = time.perf_counter() - t0
    s.set_profiling(True, stride=max(1, steps // 64))
    s.iterations(steps)
    s.synchronize()
    prof = s.get_profile()
    s.set_profiling(False)
    R = 4 if precision == "float" else 8
    names = ["forward_mm", "backward_mm", "finish_delta"]
    out = {"workload": f"random set cover (std::mt19937_64 seed 12345), row size {k}, V={vars_}, B={rows}: {sizes['N']} BDD nodes (BASELINE.json configs[1])",
           "value": steps / dt, "unit": "iterations/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f32" if R == 4 else "f64",
           "algorithmic_bytes_per_iteration": iteration_bytes(sizes, R),
           "frac_whole_iteration": iteration_bytes(sizes, R) * steps / dt / 1e9 / HBM_PEAK_GBS,
           "avg_launch_us": {names[i]: prof["total_ms"][i] / max(prof["launches"][i], 1) * 1e3 for i in range(3)},
           "solve_sweeps": s.solve_sweep_kind(), "packs": s.nr_packs(), "hbm_resident_bytes": s.device_bytes(),
           "lower_bound_after": {"value": s.lower_bound()}}
    s.close()
    return out


def matching_config(device, precision, n=8):
    """BASELINE.json configs[0]: the n x n assignment problem (-2 on the diagonal, -1 elsewhere: optimum -2 n) — an instance that fits one
    workgroup, so iterations run inside one launch; the rate of iterations(n) and of run_solver (bound and termination tests every iteration)."""
    from bdd_amd import to_bdd_collection
    from bdd_amd.instances import assignment_ilp
    from bdd_amd.solver import bdd_hip_parallel_mma, run_solver
    ilp = assignment_ilp(n)
    col = to_bdd_collection(ilp)
    s = bdd_hip_parallel_mma(col, ilp.objective, precision=precision, device=device)
    s.iterations(5000)
    s.synchronize()
    t0 = time.perf_counter()
    s.iterations(50000)
    s.synchronize()
    dt = time.perf_counter() - t0
    lb = s.lower_bound()
    fused = s.fused_small()
    s.close()
    s = bdd_hip_parallel_mma(col, ilp.objective, precision=precision, device=device)
    rs = run_solver(s, max_iter=5000, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
    s.close()
    return {"workload": f"{n} x {n} bipartite matching, 2n simplex BDDs over n^2 variables (BASELINE.json configs[0])", "dtype": "f64" if precision == "double" else "f32",
            "value": 50000 / dt, "unit": "iterations/s", "us_per_iteration": dt / 50000 * 1e6, "iterations_in_one_launch": fused,
            "run_solver_us_per_iteration": rs["seconds"] / max(rs["iterations"], 1) * 1e6, "lower_bound": lb, "known_optimum": -2.0 * n}


def lbfgs_rate(col, costs, precision, device, args, sizes, iters=100):
    from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma
    s = bdd_hip_parallel_mma(col, costs, precision=precision, device=device, pack_width=args.pack_width, deterministic=args.deterministic,
                             vars_per_bin=args.vars_per_bin, stage_cap=args.stage_cap, waves_per_block=args.wpb)
    l = bdd_hip_lbfgs(s)
    for _ in range(20):
        l.iteration()
    s.lower_bound()
    t0 = time.perf_counter()
    trials = steps = 0
    for _ in range(iters):
        l.iteration()
        st = l.state()
        trials += st["last_trials"]
        steps += st["last_kind"]
    lb = s.lower_bound()
    dt = time.perf_counter() - t0
    l.close()
    s.close()
    R, m = (4 if precision == "float" else 8), 5
    b = lbfgs_bytes(sizes, R, m, trials / iters)
    return {"value": iters / dt, "unit": "iterations/s", "ms_per_iteration": dt / iters * 1e3, "iterations": iters, "lbfgs_steps": steps,
            "trial_steps_per_iteration": trials / iters, "lower_bound_after": {"iterations": 20 + iters, "value": lb},
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_iteration": b,
                         "achieved": b * iters / dt / 1e9, "frac": b * iters / dt / 1e9 / HBM_PEAK_GBS,
                         "basis": "SURVEY §8(d): B_iter + (2m + 6) R L' + (trial steps + 1) extra sweeps of 12 N' + 2R N + 2R L' bytes, m = 5, on "
                                  "the wall clock of the loop (bound read-backs and host decisions included)"}}


INFINITY_CACHE_BYTES = 256 << 20  # MI355X_MICROARCH.md "Infinity Cache (L3)"


def roofline(prof, sizes, R, its_per_gpu, args, sfx, triad_gbs, copy_gbs, resident):
    """Dominant kernel = the slower of the two sweep launches; durations from hipEvent pairs recorded on the solver's stream inside
    a second pass of the same K steps.
      achieved / frac_algorithmic : SURVEY §8(d)'s ALGORITHMIC bytes of one sweep launch / its average duration (the contract figure).
                                    The layout moves fewer bytes than that (4-byte shared node words instead of 12 B of indices, no
                                    terminals), so this rate can exceed what HBM delivers — it is a work rate, not a bandwidth.
      traffic / frac             : bytes the launch really moves, from the committed rocprofv3 PMC passes of this command
                                    ((2 FETCH_SIZE + WRITE_SIZE) x 1024, profiles/<tag>/traffic.json, valid while the kernel sources
                                    hash to what they were taken on) / the same duration / 8 TB/s.  `frac` falls back to the
                                    algorithmic figure (and says so in frac_basis) when no matching counters are committed.
      frac_whole_iteration       : B_iter x value / 8 TB/s on the driver's clock (4 launches + gaps)."""
    names = ["forward_mm", "backward_mm", "finish_delta", "other"]
    avg_ms = [prof["total_ms"][i] / max(prof["launches"][i], 1) for i in range(4)]
    dom = 0 if avg_ms[0] >= avg_ms[1] else 1
    b_sweep, b_exch, b_iter = sweep_bytes(sizes, R), exchange_bytes(sizes, R), iteration_bytes(sizes, R)
    achieved = b_sweep / (avg_ms[dom] * 1e-3) / 1e9 if avg_ms[dom] > 0 else 0.0
    traffic, traffic_exch, traffic_src, us_rocprof = measured_traffic(names[dom], args, sfx)
    counter_gbs = traffic / (avg_ms[dom] * 1e-3) / 1e9 if traffic and avg_ms[dom] > 0 else None
    whole = b_iter * its_per_gpu / 1e9
    hbm_only = hbm_only_fractions()
    return {
        "bound": "hbm",
        "kernel": names[dom],
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": (counter_gbs if counter_gbs is not None else achieved) / HBM_PEAK_GBS,
        "frac_basis": (f"counter bytes of the launch ({traffic_src}) / hipEvent duration" if counter_gbs is not None
                       else "algorithmic bytes (no committed PMC passes match these kernel sources / this workload)"),
        "frac_algorithmic": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "achieved_counter_GBs": counter_gbs,
        "algorithmic_bytes_per_launch": b_sweep,
        "algorithmic_bytes_exchange_launch": b_exch,
        "algorithmic_bytes_per_iteration": b_iter,
        "frac_exchange_launch": (b_exch / (avg_ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms[2] > 0 else None,
        "traffic_exchange_launch": traffic_exch,
        "frac_exchange_launch_counter_bytes": (traffic_exch / (avg_ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_exch and avg_ms[2] > 0 else None,
        "whole_iteration_GBs": whole,
        "frac_whole_iteration": whole / HBM_PEAK_GBS,
        "avg_launch_ms": {names[i]: avg_ms[i] for i in range(3)},
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats run (the hipEvent figure above carries ~2.6 us of
        # event overhead per pair); frac_rocprof_duration = counter bytes / that duration / peak
        "kernel_us_hipevent": avg_ms[dom] * 1e3,
        "kernel_us_rocprof": us_rocprof,
        "frac_rocprof_duration": (traffic / (us_rocprof * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic and us_rocprof else None,
        "ms_per_step_instrumented_pass": prof.get("ms_per_step_instrumented"),
        "timed_launches": {names[i]: prof["launches"][i] for i in range(3)},
        "stream_triad_GBs": triad_gbs,
        "stream_copy_GBs": copy_gbs,
        "infinity_cache_note": {
            "resident_bytes": resident,
            "infinity_cache_bytes": INFINITY_CACHE_BYTES,
            "resident_over_cache": resident / INFINITY_CACHE_BYTES,
            "note": "FETCH_SIZE / WRITE_SIZE count the L2's fabric requests, Infinity-Cache hits included: with the working set within a few "
                    "times the 256 MiB cache part of `traffic` is served on-die, so `frac` is an L2<->fabric rate; the HBM-only figure is "
                    "the whole-iteration fraction of an instance ten times the size (4.5 GB resident), hbm_only_frac_whole_iteration",
            "hbm_only_frac_whole_iteration": hbm_only,
        },
    }


def hbm_only_fractions():
    """frac_whole_iteration of the 105 M-node instance (V = 10 M, B = 5 M, k = 10: 4.5 / 7.2 GB resident, no Infinity-Cache help), measured
    with tools/kbench.py and committed in profiles/ (None if the file is missing)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_only_105m.json")), reverse=True):   # the newest round's
        d = json.load(open(path))
        return {"f32": d.get("f32"), "f64": d.get("f64"), "source": "profiles/" + os.path.basename(path), "sources_hash": d.get("_source_hash"),
                "matches_these_sources": d.get("_source_hash") == source_hash() if d.get("_source_hash") else None}
    return None


def timed_region(run_steps, device_sync, dist):
    """barrier + device sync, EXACTLY the K steps, device sync + barrier; returns the MAX over ranks of the
    wall time (the driver's contract).  `dist` is torch.distributed (gloo) or None (single process)."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    device_sync()
    t0 = time.perf_counter()
    run_steps()
    device_sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_rate(world, steps, dt):
    """whole-job throughput: every rank runs its own instance (replicas, no data-path collective)"""
    return world * steps / dt


def measured_traffic(kernel, args, sfx):
    """(HBM bytes per launch of the dominant sweep kernel, of the exchange kernel, source) from the committed rocprofv3 PMC passes of
    this same command (profiles/<tag>/traffic.json, produced by tools/profile.sh + tools/collect_profiles.py: separate --pmc runs,
    bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md §HBM).  The file is stamped with the hash of the kernel
    sources it was measured on; (None, None, None) when no profile matches the workload or the sources changed since."""
    tag = {(1_000_000, 500_000, 10): "10m", (100_000, 50_000, 10): "1m"}.get((args.vars, args.rows, args.k))
    if tag is None or args.deterministic or args.pack_width or args.wpb or args.vars_per_bin or args.stage_cap:
        return None, None, None, None
    for rnd, dsfx in [(r, x) for r in ("r06", "r05", "r04", "r03", "r02") for x in dict.fromkeys((sfx, "f32"))]:   # a round's f32 directory holds both precisions
        rel = os.path.join("profiles", f"{rnd}_{tag}_{dsfx}", "traffic.json")
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get("_source_hash") != source_hash():
            continue
        want = "k_fwd_narrow" if kernel == "forward_mm" else "k_bwd_narrow"
        real = "float" if sfx == "f32" else "double"   # one profile holds both precisions: the default bench run times both
        sweep = sweep_entry = None

        def take(v):
            nonlocal sweep, sweep_entry
            sweep, sweep_entry = v["hbm_bytes"], v

        for name, v in d.items():   # third-generation streaming solve sweeps (a lane per layer): <REAL, waves per block>
            if isinstance(v, dict) and re.search(want + "3<" + real + r", \d+(?:, \w+)?>", name):   # <REAL, waves per block[, NT]>
                take(v)
        if sweep is None:
            for name, v in d.items():   # second generation: <REAL, R, waves per block, GEN[, NT]>
                if isinstance(v, dict) and re.search(want + "2<" + real + r", \d+, \d+, \w+(?:, \w+)?>", name):
                    take(v)
        if sweep is None:
            for name, v in d.items():
                m = re.search(want + "<" + real + r", \d+, (\d+), \d+(?:, \w+)*>", name)   # <REAL, R, MODE, waves per block[, SEG[, NT]]>; MODE 1 = solve
                if m and m.group(1) == "1":
                    take(v)
        if sweep is None:
            for res in ("_res2", "_res"):   # small instances: the resident sweeps, second / first generation
                for name, v in d.items():
                    if sweep is None and isinstance(v, dict) and re.search(want.replace("_narrow", res) + "<" + real + ",", name):
                        take(v)
        exch = next((v["hbm_bytes"] for name, v in d.items() if "k_exchange_reduce<" + real + "," in name and isinstance(v, dict)), None)
        if sweep is not None:
            return sweep, exch, rel, sweep_entry.get("avg_us_rocprof")   # rocprofv3 --kernel-trace --stats average of the same kernel (collect_profiles.py)
    return None, None, None, None


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
    scaling = {}
    for th in sorted({min(ncpu, t) for t in (16, 32, 64, 128, 256)}):
        o.set_threads(th)
        t1 = time.perf_counter()
        o.iteration()
        rate = 1.0 / (time.perf_counter() - t1)
        scaling[str(th)] = rate
        if rate > best:
            best, cores = rate, th
    o.set_threads(1)
    t1 = time.perf_counter()
    o.iteration()
    single = 1.0 / (time.perf_counter() - t1)
    scaling["1"] = single
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
        # one iteration per thread count (it/s); the oracle's arrays are first touched in parallel over BDDs since r6 (pages on every NUMA node)
        "thread_scaling_one_iteration_each": dict(sorted(scaling.items(), key=lambda kv: int(kv[0]))),
        "sample": f"{n} iterations of the same {sizes['N']}-node instance after {warm} warm-up / thread-count-probe "
                  f"iterations, oracle/mma_oracle.c with OpenMP over BDDs ({cores} of {ncpu} hardware threads: the "
                  f"fastest of 16..256), {args.precision}",
        "lb_after": {"iterations": n + warm, "cpu": cpu_lb, "gpu": gpu_lb,
                     "rel_diff": abs(cpu_lb - gpu_lb) / max(abs(cpu_lb), 1e-300)},
    }


if __name__ == "__main__":
    main()
