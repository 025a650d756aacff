#!/bin/bash
# round 5, last session: hops per staggered pack against the number of waves the chip holds (k_*_mixed<float,1,1,2>: 101 VGPRs = 4 waves per SIMD = 4 096 waves)
for st in 0 30 36 42 48 56 64 70 84; do
  echo "== 40000 knapsack rows stagger=$st"; timeout 300 python tools/widebench.py --rows 40000 --iters 200 --stagger $st 2>&1 | grep -E "layout|iteration"
done
for st in 0 56 70; do
  echo "== 20000 + 250000 stagger=$st"; timeout 300 python tools/widebench.py --rows 20000 --cover-rows 250000 --iters 200 --stagger $st 2>&1 | grep -E "layout|iteration"
done
