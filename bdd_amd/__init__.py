"""bdd_amd — MI355X-native parallel deferred min-marginal averaging over BDDs.

Host-side mirror of the reference's solver interface for the `cuda parallel mma` path.
The compute lives in bdd_amd/csrc/libbdd_mma_hip.so (hand-written HIP for gfx950 behind the
C-ABI of include/bdd_mma.h); there is no CPU fallback.
"""
import ctypes as _C
import importlib.util as _ilu
import os as _os
import sys as _sys


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, file name without
    the version); libbdd_mma_hip.so asks for libamdhip64.so.7.  Loaded after torch, the library binds to torch's copy by soname and
    both share devices, streams and allocations (the tests hand torch device buffers to the C-ABI).  Loaded BEFORE torch, the loader
    would take /opt/rocm's copy, torch would later load its own by file name, and the second runtime to initialise finds no device
    ("no ROCm-capable device is detected").  So when a torch installation exists, its HIP runtime is mapped first — without
    importing torch.  BDDMMA_SYSTEM_HIP_RUNTIME=1 keeps the system runtime (processes that never import torch do not care)."""
    if "torch" in _sys.modules or _os.environ.get("BDDMMA_SYSTEM_HIP_RUNTIME") == "1":
        return
    try:
        spec = _ilu.find_spec("torch")
        for loc in (spec.submodule_search_locations or []) if spec else []:
            path = _os.path.join(loc, "lib", "libamdhip64.so")
            if _os.path.exists(path):
                _C.CDLL(path, mode=_C.RTLD_GLOBAL)
                return
    except Exception:
        pass  # no torch, or not a ROCm build: the system runtime is the only one


_share_hip_runtime_with_torch()

from .bdd_collection import BddCollection, TOPSINK, BOTSINK  # noqa: F401
from .ilp import ILP, parse_lp, to_bdd_collection  # noqa: F401
