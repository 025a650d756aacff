#!/bin/bash
# round 5, last session: register cap of k_*_mixed (amdgpu_waves_per_eu 4 = 101-106 VGPRs as before, 5 = 96, 6 = 80 with 2-5 spills for 64-slot packs)
for rep in 1 2; do for lib in "" build/libmw5.so build/libmw6.so; do
  echo "== 40000 knapsack rows lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 40000 --iters 200 2>&1 | grep -E "iteration|fwd_plain"
done; done
for lib in "" build/libmw5.so build/libmw6.so; do
  echo "== 20000 + 250000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 20000 --cover-rows 250000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 30000 + 100000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 30000 --cover-rows 100000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 10000 + 400000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 10000 --cover-rows 400000 --iters 200 2>&1 | grep -E "iteration"
  echo "== 4000 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 4000 --iters 300 2>&1 | grep -E "iteration"
done
