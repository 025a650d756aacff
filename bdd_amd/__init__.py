"""bdd_amd — MI355X-native parallel deferred min-marginal averaging over BDDs.

Host-side mirror of the reference's solver interface for the `cuda parallel mma` path.
The compute lives in bdd_amd/csrc/libbdd_mma_hip.so (hand-written HIP for gfx950 behind the
C-ABI of include/bdd_mma.h); there is no CPU fallback.
"""
from .bdd_collection import BddCollection, TOPSINK, BOTSINK  # noqa: F401
from .ilp import ILP, parse_lp, to_bdd_collection  # noqa: F401
