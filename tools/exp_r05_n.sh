#!/bin/bash
# round 5, last session: diamond-shaped BDDs widest first (keep_bdd_order 0) against the order before (2), same box
mkdir -p gpurun_out/n
for rep in 1 2; do
for k in 0 2; do
  echo "== 40000 knapsack rows keep=$k rep=$rep"; timeout 300 python tools/widebench.py --rows 40000 --iters 200 --keep-order $k 2>&1 | grep -E "layout|iteration|fwd_plain"
done; done
for k in 0 2; do
  echo "== 20000 knapsack + 250000 covering rows keep=$k"; timeout 300 python tools/widebench.py --rows 20000 --cover-rows 250000 --iters 200 --keep-order $k 2>&1 | grep -E "layout|iteration"
  echo "== 30000 + 100000 keep=$k"; timeout 300 python tools/widebench.py --rows 30000 --cover-rows 100000 --iters 200 --keep-order $k 2>&1 | grep -E "layout|iteration"
  echo "== 25000 rows of 18 keep=$k"; timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 --keep-order $k 2>&1 | grep -E "layout|iteration"
  echo "== 4000 rows keep=$k"; timeout 300 python tools/widebench.py --rows 4000 --iters 300 --keep-order $k 2>&1 | grep -E "layout|iteration"
done
