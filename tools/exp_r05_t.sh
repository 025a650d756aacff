#!/bin/bash
# round 5, last session: wide sweeps skip the node slots of a wavefront that lie past the hop's last slot (scalar branch).
# shipped (forward solve sweeps only) / build/libprev.so (none) / build/libskipall.so (every mode, both directions), same box
for rep in 1 2; do for lib in "" build/libprev.so build/libskipall.so; do
  echo "== 25000 rows of 18 lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 2>&1 | grep -E "iteration|fwd_plain"
  echo "== 30000 rows of 16 lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 30000 --k 16 --iters 200 2>&1 | grep -E "iteration|fwd_plain"
  echo "== 4000 rows of 18 lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 4000 --k 18 --iters 200 2>&1 | grep -E "iteration|fwd_plain"
  echo "== 40000 knapsack rows lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 40000 --iters 200 2>&1 | grep -E "iteration|fwd_plain"
done; done
