"""Full `bdd_solver` JSON path on the GPU, against the known answers of the reference's own end-to-end tests."""
import json
import subprocess
import sys

import numpy as np
import pytest

from bdd_amd.bdd_solver import bdd_solver
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, assignment_ilp, brute_force_optimum, mrf_ilp

pytestmark = pytest.mark.gpu

TC = {"maximum iterations": 200, "improvement slope": 0.0, "minimum improvement": 0.0, "time limit": 1e10}


def cfg(lp, **kw):
    c = {"precision": "double", "relaxation solver": "cuda parallel mma", "termination criteria": dict(TC), "input": lp}
    c.update(kw)
    return c


def test_bipartite_matching_kats():
    # test/test_bdd_bipartite_matching_problem.cpp:8-59 (there with "sequential mma", 20 iterations, 1e-6)
    s = bdd_solver(cfg(assignment_ilp(3).write_lp()), quiet=True).solve()
    assert abs(s.lower_bound() - (-6.0)) <= 1e-6
    c = -np.ones((3, 3)); c[:, 0] = -2
    s = bdd_solver(cfg(assignment_ilp(3, c).write_lp()), quiet=True).solve()
    assert abs(s.lower_bound() - (-4.0)) <= 1e-6
    s = bdd_solver(cfg(assignment_ilp(8).write_lp(), precision="float"), quiet=True).solve()
    assert abs(s.lower_bound() - (-16.0)) <= 1e-4


def test_loose_covering_kat():
    # test/test_loose_covering_problem.cpp:8-88: LB 1.5; the tightened instance has a larger bound
    base = "Minimize\nx1 + x2 + x3 + x4 + x5 + x6\nSubject To\nx1 + x2 + x4 >= 1\nx1 + x3 + x5 >= 1\nx2 + x3 + x6 >= 1\n"
    tail = "Bounds\nBinaries\nx1\nx2\nx3\nx4\nx5\nx6\nEnd\n"
    s = bdd_solver(cfg(base + tail), quiet=True).solve()
    assert abs(s.lower_bound() - 1.5) <= 1e-4
    t = bdd_solver(cfg(base + "x1 + x2 + x3 + x4 + x5 + x6 >= 2\n" + tail), quiet=True).solve()
    assert t.lower_bound() > 1.5 + 1e-4


@pytest.mark.parametrize("solver", ["cuda parallel mma", "lbfgs cuda mma", "cuda lbfgs parallel mma"])
def test_mrf_end_to_end_with_rounding(solver):
    ilp = mrf_ilp(**LONG_CHAIN)
    c = cfg(ilp.write_lp(), **{"relaxation solver": solver,
                               "perturbation rounding": {"initial perturbation": 0.1, "perturbation growth rate": 1.2,
                                                         "inner iterations": 50, "outer iterations": 60}})
    s = bdd_solver(c, quiet=True).solve()
    assert s.result["lb_final"] >= s.result["lb_initial"] - 1e-9
    assert abs(s.result["lb_final"] - (-9.0)) < 1e-6          # test_bdd_cuda_parallel_mma.cu:230 (long_mrf_chain)
    assert s.solution is not None and ilp.feasible(s.solution)
    assert abs(ilp.evaluate(s.solution) - (-9.0)) < 1e-9       # tree MRF: relaxation tight, rounding recovers the optimum


def test_rounding_on_grid_is_feasible_and_bounded():
    ilp = mrf_ilp(**GRID_3X3)
    c = cfg(ilp.write_lp(), **{"perturbation rounding": {"inner iterations": 50, "outer iterations": 100}})
    s = bdd_solver(c, quiet=True).solve()
    assert s.solution is not None and ilp.feasible(s.solution)
    assert ilp.evaluate(s.solution) >= s.result["lb_final"] - 1e-6


def test_min_marginals_and_normalize():
    ilp = assignment_ilp(3)
    s = bdd_solver(cfg(ilp.write_lp(), **{"normalize constraints": True, "print statistics": True}), quiet=True).solve()
    mm = s.min_marginals()
    assert len(mm) == 9 and all(m.shape == (2, 2) for m in mm)
    names, m0, m1 = s.min_marginals_with_variable_names()
    assert names == ilp.var_names


def test_command_line(tmp_path):
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg(assignment_ilp(3).write_lp())))
    out = subprocess.run([sys.executable, "-m", "bdd_amd.bdd_solver_cl", str(p)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "lower bound = -6" in out.stdout


def long_row_lp(n=60, need=5, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.integers(1, 50, n)
    names = [f"x{i}" for i in range(n)]
    lp = "Minimize\n" + " + ".join(f"{int(ci)} {v}" for ci, v in zip(c, names)) + "\nSubject To\n"
    lp += " + ".join(names) + f" >= {need}\n"
    for i in range(0, n - 1, 7):      # a few short rows so that the instance is not a single BDD
        lp += f"{names[i]} + {names[i + 1]} <= 1\n"
    lp += "Bounds\nBinaries\n" + "\n".join(names) + "\nEnd\n"
    return lp, c


def test_split_bdds_same_parity_and_bound():
    """`"split bdds"` (bdd_solver.cpp:112-123, bdd_preprocessor.cpp:372-415): the long row is cut into chunks
    coupled by auxiliary variables; GPU and oracle agree on the split instance, and the split relaxation
    reaches the bound of the un-split one (both decompositions are exact for this instance)."""
    from oracle.oracle import Oracle

    lp, c = long_row_lp()
    tc = dict(TC); tc["maximum iterations"] = 3000
    full = bdd_solver(cfg(lp, **{"termination criteria": tc}), quiet=True).solve()
    split = bdd_solver(cfg(lp, **{"termination criteria": tc, "split bdds": {"split length": 10}}), quiet=True).solve()
    assert split.bdd_col.nr_bdds() == full.bdd_col.nr_bdds() + 5
    assert max(len(split.bdd_col.layer_widths(b)) for b in range(split.bdd_col.nr_bdds())) <= 10 + 2 * 6
    assert split.solver.nr_variables() > full.solver.nr_variables()
    costs = np.zeros(split.solver.nr_variables()); costs[: len(c)] = c
    o = Oracle(split.bdd_col, costs, "double")
    for _ in range(3000):
        o.iteration()
    assert abs(o.lower_bound() - split.lower_bound()) <= 1e-7 * max(1.0, abs(o.lower_bound()))
    assert split.lower_bound() <= full.lower_bound() + 1e-6
    assert split.lower_bound() >= full.lower_bound() - 1e-2 * abs(full.lower_bound())
    # with the implication BDD over the auxiliary variables: one more BDD, still a valid bound, and GPU == oracle
    imp = bdd_solver(cfg(lp, **{"termination criteria": tc, "split bdds": {"split length": 10, "implication bdd": True}}), quiet=True).solve()
    assert imp.bdd_col.nr_bdds() == split.bdd_col.nr_bdds() + 1
    assert imp.lower_bound() <= full.lower_bound() + 1e-6
    o = Oracle(imp.bdd_col, costs, "double")
    for _ in range(3000):
        o.iteration()
    assert abs(o.lower_bound() - imp.lower_bound()) <= 1e-7 * max(1.0, abs(o.lower_bound()))


@pytest.mark.parametrize("front_end", ["python", "pybind"])
def test_variables_that_occur_in_the_objective_only(front_end):
    """an ILP whose last variables appear in no constraint used to fail in bddmma_create (cost vector longer than the
    BDDs' variables).  They are free: min(0, c) goes to the bound, the primal takes their better value."""
    lp = ("Minimize\n2 x1 + 3 x2 + x3 - 4 y1 + 5 y2\nSubject To\nx1 + x2 + x3 >= 1\nx1 + x2 <= 1\nBounds\nBinaries\nx1\nx2\nx3\ny1\ny2\nEnd\n")
    c = cfg(lp, **{"perturbation rounding": {"inner iterations": 20, "outer iterations": 30}})
    if front_end == "python":
        s = bdd_solver(c, quiet=True).solve()
    else:
        from bdd_amd import bdd_solver_py
        s = bdd_solver_py.bdd_solver(c, quiet=True).solve()
    assert abs(s.lower_bound() - (1.0 - 4.0)) <= 1e-6          # x3 = 1 covers the row; y1 = 1 is free money
    assert s.solution is not None and list(s.solution) == [0, 0, 1, 1, 0]


def test_batch_farm_command_line(tmp_path):
    """bdd_solver_cl --batch / --bench-set-cover: independent instances, one host thread per device slot (here two slots on GPU 0)."""
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bdd_amd", "csrc", "bdd_solver_cl")
    cfgs = []
    for i, (ilp, lb) in enumerate([(assignment_ilp(3), -6.0), (mrf_ilp(**LONG_CHAIN), -9.0), (assignment_ilp(8), -16.0), (mrf_ilp(**GRID_3X3), -8.0)]):
        p = tmp_path / f"c{i}.json"
        p.write_text(json.dumps(cfg(ilp.write_lp())))
        cfgs.append((str(p), lb))
    out = subprocess.run([exe, "--batch", *[c for c, _ in cfgs], "--devices", "0,0", "--quiet"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 4 and all(r["ok"] and r["device"] == 0 for r in rows)
    for r, (path, lb) in zip(rows, cfgs):
        assert r["config"] == path and abs(r["lower_bound"] - lb) <= 1e-6
    # a failing config is reported, the others still run
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(cfg("Minimize\nx\nSubject To\nx >= 2\nBounds\nBinaries\nx\nEnd\n")))
    out = subprocess.run([exe, "--batch", cfgs[0][0], str(bad), "--devices", "0", "--quiet"], capture_output=True, text=True, timeout=300)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 1 and rows[0]["ok"] and not rows[1]["ok"] and "error" in rows[1]
    # the benchmark instance of bench.py, two replicas on one GPU, same bound as the Python path
    out = subprocess.run([exe, "--bench-set-cover", "3000", "2500", "8", "--iterations", "40", "--warmup", "10", "--seeds", "12345-12346", "--devices", "0,0",
                          "--precision", "double"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 3 and rows[2]["instances"] == 2 and rows[2]["aggregate_iterations_per_second"] > 0
    from bdd_amd.instances import random_set_cover_mt
    from bdd_amd.solver import bdd_hip_parallel_mma
    col, costs = random_set_cover_mt(3000, 2500, 8, seed=12346)
    ref = bdd_hip_parallel_mma(col, costs, precision="double")
    ref.iterations(50)
    assert abs(rows[1]["lower_bound"] - ref.lower_bound()) <= 1e-9 * abs(ref.lower_bound())


def test_eight_slot_rehearsal_on_one_gpu(tmp_path):
    """BASELINE.json configs[4] (8 instances, one per GPU) cannot be run here: both N = 8 paths are rehearsed with all eight slots on the
    one GPU of this box — the line shapes, the per-rank work and the host-thread budget are what an 8-GPU node would see, the
    numbers are not.  (1) bench.py under torch.distributed.run, 8 ranks, gloo barrier, max over ranks, ONE JSON line from rank 0;
    (2) bdd_solver_cl --bench-set-cover with --devices 0,0,0,0,0,0,0,0: eight host threads, each builds and times its own instance."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BDDMMA_BENCH_SHARE_GPUS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29741", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--vars", "100000",
                          "--rows", "50000", "--no-second-precision", "--clock-warm", "0.05"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    one = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
    assert one <= set(d) and d["n_gpus"] == 8 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 8 * 20 / (d["ms_per_step"] * 20e-3)) <= 1e-6 * d["value"]       # whole-job rate = all ranks' steps / max time
    assert "cpu_baseline" not in d                                                          # the CPU leg is timed at N = 1 only
    exe = os.path.join(root, "bdd_amd", "csrc", "bdd_solver_cl")
    out = subprocess.run([exe, "--bench-set-cover", "100000", "50000", "10", "--iterations", "200", "--warmup", "20", "--seeds", "12345-12352",
                          "--devices", "0,0,0,0,0,0,0,0", "--precision", "float"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 9 and rows[8]["instances"] == 8 and rows[8]["aggregate_iterations_per_second"] > 0
    assert sorted(r["seed"] for r in rows[:8]) == list(range(12345, 12353)) and all(r["ok"] for r in rows[:8])
    # equal work per slot: the eight bounds are those of eight DIFFERENT instances (no slot ran another one's seed twice)
    assert len({round(r["lower_bound"], 6) for r in rows[:8]}) == 8


def test_eight_full_size_instances_share_one_gpu():
    """configs[4] at its own size: eight DIFFERENT 10.5 M-node instances (seeds 12345-12352) built and iterated by the farm's eight host
    threads, all on the one GPU of this box (3.7 GB resident).  What an 8-GPU node would run per device is exactly one of these slots; here
    the eight share the chip, so the aggregate rate is about ONE device's rate — the test pins that nothing in the farm degrades at this
    size (memory, shared layout threads, eight enqueue threads): every slot's bound equals the bound a lone solver reaches on that seed,
    and the aggregate stays above half of the lone solver's rate."""
    import os
    from bdd_amd.instances import random_set_cover_mt
    from bdd_amd.solver import bdd_hip_parallel_mma
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bdd_amd", "csrc", "bdd_solver_cl")
    out = subprocess.run([exe, "--bench-set-cover", "1000000", "500000", "10", "--iterations", "60", "--warmup", "20", "--seeds", "12345-12352",
                          "--devices", "0,0,0,0,0,0,0,0", "--precision", "float"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 9 and rows[8]["instances"] == 8 and all(r["ok"] for r in rows[:8])
    assert sorted(r["seed"] for r in rows[:8]) == list(range(12345, 12353))
    assert len({round(r["lower_bound"], 3) for r in rows[:8]}) == 8
    lone_rate = None
    for seed in (12345, 12352):
        col, costs = random_set_cover_mt(1_000_000, 500_000, 10, seed=seed)
        s = bdd_hip_parallel_mma(col, costs, precision="float")
        s.iterations(20)
        ms = s.time_iterations(60)
        lone_rate = 60 / ms * 1e3
        lb = s.lower_bound()
        row = next(r for r in rows[:8] if r["seed"] == seed)
        assert abs(row["lower_bound"] - lb) <= 1e-5 * abs(lb), (seed, row["lower_bound"], lb)
        del s
    assert rows[8]["aggregate_iterations_per_second"] > 0.5 * lone_rate, (rows[8], lone_rate)


def test_export_keys_of_both_drivers(tmp_path):
    """"export bdd lp" / "export bdd graph" (bdd_solver.cpp:400-410, :432-462) through the C++ driver and the Python driver: the same files,
    equal to what the collection's own emitters give (tests/test_exports.py pins those on the reference's output)."""
    import os
    from bdd_amd import parse_lp, to_bdd_collection
    lp = "Minimize\n1 x_1 + 2 x_2 + 1.5 x_3 - 1 x_4\nSubject To\nx_1 + x_2 + x_3 = 1\nx_2 + x_3 + x_4 >= 1\n2 x_1 + 3 x_3 + 4 x_4 <= 5\nEnd\n"
    src = tmp_path / "p.lp"
    src.write_text(lp)
    ilp = parse_lp(lp)
    col = to_bdd_collection(ilp)
    want_lp = col.write_bdd_lp(ilp.objective)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bdd_amd", "csrc", "bdd_solver_cl")
    for tag, cmd in (("cpp", [exe]), ("py", [sys.executable, "-m", "bdd_amd.bdd_solver_cl"])):
        cfg = {"input": str(src), "relaxation solver": "cuda parallel mma", "termination criteria": {"maximum iterations": 20},
               "export bdd lp": str(tmp_path / f"{tag}.lp"), "export bdd graph": str(tmp_path / f"{tag}_graph.dot")}
        out = subprocess.run(cmd + [json.dumps(cfg)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert (tmp_path / f"{tag}.lp").read_text() == want_lp
        for b in range(col.nr_bdds()):
            got = (tmp_path / f"{tag}_graph_{b}.dot").read_text()
            assert sorted(got.splitlines()) == sorted(col.export_graphviz(b).splitlines())


SHORT_MRF_CHAIN = """Minimize
3 mu_1_0 + 1 mu_1_1
- 1 mu_2_0 + 0 mu_2_1
+ 1 mu_00 + 2 mu_10 + 1 mu_01 + 0 mu_11
Subject To
mu_1_0 + mu_1_1 = 1
mu_2_0 + mu_2_1 = 1
mu_00 + mu_10 + mu_01 + mu_11 = 1
mu_1_0 - mu_00 - mu_01 = 0
mu_1_1 - mu_10 - mu_11 = 0
mu_2_0 - mu_00 - mu_10 = 0
mu_2_1 - mu_01 - mu_11 = 0
"""


def _front_end_bound(front_end, c, tmp_path):
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bdd_amd", "csrc", "bdd_solver_cl")
    if front_end == "python":
        return bdd_solver(c, quiet=True).solve().lower_bound()
    if front_end == "pybind":
        from bdd_amd import bdd_solver_py
        return bdd_solver_py.bdd_solver(c, quiet=True).solve().lower_bound()
    p = tmp_path / "c.json"
    p.write_text(json.dumps(c))
    out = subprocess.run([exe, "--batch", str(p), "--devices", "0", "--quiet"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][0]["lower_bound"]


@pytest.mark.parametrize("front_end", ["python", "pybind", "bdd_solver_cl"])
def test_bounds_fixations_known_answers(front_end, tmp_path):
    """test/test_bdd_solver_fix_variable.cpp:6-48: short_mrf_chain has bound 1; with mu_2_1 fixed to 0 it is 2, with mu_1_1 = 0 on top
    of that 3 (1e-6; the reference states them for its CPU solvers).  The fixations arrive through the `Bounds` section of the .lp text
    here (ILP_parser.cpp:128-131,343-436 -> ILP_input::reduce), which every front end reads with the same two readers.
    The third value is where the reference's GPU rule shows: `mu_1_1 = 0` turns the row `mu_1_1 - mu_10 - mu_11 = 0` into one that forces
    mu_10 and mu_11 to 0, and a layer with a non-finite min-marginal exchanges nothing on the GPU (bdd_cuda_parallel_mma.cu:83-84; the
    CPU solver lets the infinity through), so the cost share parked on the forced arcs stays there: the GPU solver's fixpoint is 2.48538874...,
    a valid but weaker bound, equal to the restated CUDA rule (oracle/cuda_rule_oracle.c) to 1e-9, while the CPU rule on the same BDDs
    gives the reference's 3.  Written with the other variable of each pair fixed to 1 the same model has no such row and the GPU gives 3."""
    from bdd_amd import parse_lp, to_bdd_collection
    from oracle.oracle import CudaRuleOracle, Oracle
    for bounds, want in (("", 1.0), ("Bounds\n mu_2_1 = 0\n", 2.0), ("Bounds\n 1 <= mu_2_0\n 0 <= mu_1_1 <= 0\n", 3.0), ("Bounds\n mu_2_0 = 1\n mu_1_0 >= 1\n", 3.0)):
        lb = _front_end_bound(front_end, cfg(SHORT_MRF_CHAIN + bounds + "End\n"), tmp_path)
        assert abs(lb - want) <= 1e-6, (bounds, lb)
    lp = SHORT_MRF_CHAIN + "Bounds\n mu_2_1 = 0\n mu_1_1 <= 0\n" + "End\n"
    lb = _front_end_bound(front_end, cfg(lp), tmp_path)
    ilp = parse_lp(lp)
    col = to_bdd_collection(ilp)
    cpu, gpu_rule = Oracle(col, np.array(ilp.objective), "double"), CudaRuleOracle(col, np.array(ilp.objective), "double")
    for _ in range(200):
        cpu.iteration()
        gpu_rule.iteration(0.5)
    assert abs(cpu.lower_bound() + ilp.constant - 3.0) <= 1e-6            # test_bdd_solver_fix_variable.cpp:46-48
    assert abs(lb - (gpu_rule.lower_bound() + ilp.constant)) <= 1e-9 and 2.4 < lb <= 3.0 + 1e-9
    # a row of products is refused by every front end (test/test_ILP_parser.cpp:28-33 is the reference's vector for the form)
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bdd_amd", "csrc", "bdd_solver_cl")
    bad = cfg("Minimize\nx1 + x2 + x3\nSubject To\nx1*x2 + x3 >= 1\nEnd\n")
    if front_end == "bdd_solver_cl":
        out = subprocess.run([exe, json.dumps(bad)], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "nonlinear" in (out.stderr + out.stdout)
    else:
        with pytest.raises(Exception, match="nonlinear"):
            if front_end == "python":
                bdd_solver(bad, quiet=True).solve()
            else:
                from bdd_amd import bdd_solver_py
                bdd_solver_py.bdd_solver(bad, quiet=True).solve()
