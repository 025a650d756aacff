#!/bin/bash
# round 5, last session: general rows without wide packs — 128-slot packs (first-generation narrow sweeps, R = 2: 109-122 VGPRs = 4 waves per SIMD),
# the same capped at 96 (build/libn1r2.so), and 64-slot packs (R = 1: capped at 80 since this session)
for a in "--rows 100000 --k 11" "--rows 60000 --k 12" "--rows 150000 --k 10"; do
for rep in 1 2; do
  echo "== $a lib= rep=$rep"; timeout 300 python tools/widebench.py $a --iters 200 --precision float 2>&1 | grep -E "layout|iteration|fwd_plain"
  echo "== $a lib=build/libn1r2.so rep=$rep"; BDDMMA_LIB=build/libn1r2.so timeout 300 python tools/widebench.py $a --iters 200 --precision float 2>&1 | grep -E "iteration|fwd_plain"
  echo "== $a --pack-width 64 lib= rep=$rep"; timeout 300 python tools/widebench.py $a --pack-width 64 --iters 200 --precision float 2>&1 | grep -E "layout|iteration|fwd_plain"
done; done
