#!/bin/bash
# round 5, last session: register caps of the general-row solve sweeps — shipped (mixed: 6 waves per SIMD for 64-slot packs, 5 for 128; first-generation
# narrow: 6 for 64-slot packs; float only) against uncapped (build/libmw4.so) and 7 waves (build/libmw7.so), same box
for rep in 1 2; do for lib in "" build/libmw4.so build/libmw7.so; do
  echo "== 40000 knapsack rows lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 40000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 100000 rows of 11 (narrow only) lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 100000 --k 11 --iters 200 2>&1 | grep -E "iteration"
done; done
for lib in "" build/libmw4.so build/libmw7.so; do
  echo "== 20000 + 250000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 20000 --cover-rows 250000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 30000 + 100000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 30000 --cover-rows 100000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 10000 + 400000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 10000 --cover-rows 400000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 4000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 4000 --iters 300 2>&1 | grep -E "iteration"
  echo "== 25000 rows of 18 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 2>&1 | grep -E "iteration"
done
