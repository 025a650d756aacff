"""ISA lint of the built library (CPU only): no unpadded VMEM-store write-data hazard — tools/isa_lint.py (which the Makefile also runs
after linking) on libbdd_mma_hip.so; see there for the hazard."""
import os
import sys

import pytest

from bdd_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import isa_lint  # noqa: E402


@pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_unpadded_wide_store_followed_by_a_valu_write_of_its_data():
    wide_sgpr, hits, stores = isa_lint.lint(capi.LIB_PATH)
    assert not hits, f"{len(hits)} unpadded store / VALU-write pairs, e.g. {hits[:3]}"
    assert stores > 100   # the scan did see the kernels
    print(f"{wide_sgpr} wide buffer stores with an SGPR soffset, all padded")
