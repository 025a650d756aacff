"""bdd_solver_cl — command line front-end: one argument, a JSON config file or JSON string
(reference: src/bdd_solver/bdd_solver_cl.cpp:3-10).

    python -m bdd_amd.bdd_solver_cl config.json
    python -m bdd_amd.bdd_solver_cl '{"input": "problem.lp", "relaxation solver": "cuda parallel mma"}'
"""
import sys

from .bdd_solver import bdd_solver


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: bdd_solver_cl <config.json | json string>", file=sys.stderr)
        return 2
    s = bdd_solver(argv[0])
    s.solve()
    print(f"[bdd solver] lower bound = {s.lower_bound():.10g}")
    if s.solution is not None:
        print("[bdd solver] primal solution: " + " ".join(f"{n}={v}" for n, v in zip(s.ilp.var_names, s.solution)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
