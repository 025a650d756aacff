#!/bin/bash
# round 5, last session: register cap of k_*_wide2<float, SOLVE, 2 nodes per thread> (80 / 75 VGPRs = 6 waves per SIMD; 72 = 7 with 3-4 spills; 64 = 8 with 14-17)
for rep in 1 2; do for lib in "" build/libww7.so build/libww8.so; do
  echo "== 25000 rows of 18 lib=$lib rep=$rep"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 --precision float 2>&1 | grep -E "iteration|fwd_plain"
done; done
for lib in "" build/libww7.so build/libww8.so; do
  echo "== 4000 rows of 18 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 4000 --k 18 --iters 200 --precision float 2>&1 | grep -E "iteration|fwd_plain"
  echo "== 8000 rows of 16 lib=$lib"; BDDMMA_LIB=$lib timeout 300 python tools/widebench.py --rows 30000 --k 16 --iters 200 --precision float 2>&1 | grep -E "layout|iteration|fwd_plain"
done
