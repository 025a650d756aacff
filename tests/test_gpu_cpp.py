"""The C++ side of the boundary: the header-only drop-in classes (bdd_amd/csrc/bdd_hip_parallel_mma.hpp), the C++
`bdd_solver` driver and the `bdd_solver_cl` command line, compiled by `make -C bdd_amd/csrc`."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_BIN = os.path.join(ROOT, "tests", "cpp", "test_hip_backend")
CL_BIN = os.path.join(ROOT, "bdd_amd", "csrc", "bdd_solver_cl")

LP = ("Minimize\nx1 + x2 + x3 + x4 + x5 + x6\nSubject To\nx1 + x2 + x4 >= 1\nx1 + x3 + x5 >= 1\nx2 + x3 + x6 >= 1\n"
      "Bounds\nBinaries\nx1\nx2\nx3\nx4\nx5\nx6\nEnd\n")


def test_cpp_binaries_are_built():
    assert os.access(TEST_BIN, os.X_OK) and os.access(CL_BIN, os.X_OK), "run `make -C bdd_amd/csrc` (__graft_entry__.build())"


def test_command_line_reports_errors_without_touching_the_gpu(tmp_path):
    r = subprocess.run([CL_BIN], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([CL_BIN, '{"relaxation solver": "cuda parallel mma"}'], capture_output=True, text=True)
    assert r.returncode == 1 and "no input specified" in r.stderr
    r = subprocess.run([CL_BIN, '{"input": "Minimize\\nx\\nSubject To\\nx + y >= 3\\nEnd\\n"}'], capture_output=True, text=True)
    assert r.returncode == 1 and "infeasible" in r.stderr


@pytest.mark.gpu
def test_cpp_test_program():
    r = subprocess.run([TEST_BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 failure(s)" in r.stdout


@pytest.mark.gpu
def test_cpp_command_line(tmp_path):
    lp = tmp_path / "cover.lp"
    lp.write_text(LP)
    cfg = tmp_path / "cfg.json"
    cfg.write_text('{"input": "%s", "relaxation solver": "cuda parallel mma", "precision": "double", '
                   '"termination criteria": {"maximum iterations": 300, "improvement slope": 0.0, "minimum improvement": 0.0}, '
                   '"perturbation rounding": {"inner iterations": 50, "outer iterations": 50}}' % lp)
    r = subprocess.run([CL_BIN, str(cfg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lb = [ln for ln in r.stdout.splitlines() if "final lower bound" in ln]
    assert lb and abs(float(lb[0].split("=")[1]) - 1.5) <= 1e-3
    obj = [ln for ln in r.stdout.splitlines() if "primal objective" in ln]
    assert obj and abs(float(obj[0].split("=")[1]) - 2.0) <= 1e-9


OPB = ("* #variable= 6 #constraint= 3\nmin: +1 x1 +1 x2 +1 x3 +1 x4 +1 x5 +1 x6 ;\n+1 x1 +1 x2 +1 x4 >= 1 ;\n+1 x1 +1 x3 +1 x5 >= 1 ;\n"
       "+1 x2 +1 x3 +1 x6 >= 1 ;\n")


@pytest.mark.gpu
def test_cpp_command_line_opb_input_statistics_and_split(tmp_path):
    """OPB file input (by extension), the `print statistics` block and `split bdds` with the implication BDD."""
    opb = tmp_path / "cover.opb"
    opb.write_text(OPB)
    r = subprocess.run([CL_BIN, '{"input": "%s", "print statistics": true, "termination criteria": {"maximum iterations": 300, '
                                '"improvement slope": 0.0, "minimum improvement": 0.0}}' % opb], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "#variables = 6" in r.stdout and "#BDDs = 3" in r.stdout and "maximum num. constraints per var = 2" in r.stdout
    lb = [ln for ln in r.stdout.splitlines() if "final lower bound" in ln]
    assert lb and abs(float(lb[0].split("=")[1]) - 1.5) <= 1e-3
    rows = " + ".join(f"y{i}" for i in range(30))
    lp = tmp_path / "long.lp"
    lp.write_text("Minimize\n" + " + ".join(f"{1 + (7 * i) % 11} y{i}" for i in range(30)) + f"\nSubject To\n{rows} >= 3\nEnd\n")
    r = subprocess.run([CL_BIN, '{"input": "%s", "split bdds": {"split length": 6, "implication bdd": true}, "termination criteria": '
                                '{"maximum iterations": 3000, "improvement slope": 0.0, "minimum improvement": 0.0}}' % lp],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "final #BDDs = 6" in r.stdout                     # 5 chunks + the implication BDD
    lb = [ln for ln in r.stdout.splitlines() if "final lower bound" in ln]
    assert lb and float(lb[0].split("=")[1]) <= 1 + 1 + 1 + 1e-6 and float(lb[0].split("=")[1]) >= 2.9   # costs 1 at i = 0, 11, 22
