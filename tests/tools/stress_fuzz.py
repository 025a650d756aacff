"""Race hunt: run the differential fuzz test for a few seeds over and over (launch several copies at once on one GPU):
    python tests/tools/stress_fuzz.py REPEATS SEED [SEED ...]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from oracle import oracle
oracle.build()
import test_gpu_parity as T
reps, seeds = int(sys.argv[1]), [int(x) for x in sys.argv[2:]]
fails = 0
for r in range(reps):
    for s in seeds:
        try:
            T.test_randomised_instances_and_layout_options_vs_oracle.__wrapped__(s) if hasattr(T.test_randomised_instances_and_layout_options_vs_oracle, "__wrapped__") else T.test_randomised_instances_and_layout_options_vs_oracle(s)
        except AssertionError as e:
            fails += 1
            msg = str(e).splitlines()
            print(f"FAIL rep {r} seed {s}: " + " | ".join(m.strip() for m in msg[:12])[:900], flush=True)
        except KeyboardInterrupt:
            raise
        except BaseException as e:  # pytest.skip raises a BaseException
            if "skip" in type(e).__name__.lower():
                continue
            fails += 1
            traceback.print_exc()
print(f"pid {os.getpid()}: {fails} failures in {reps * len(seeds)} runs")
