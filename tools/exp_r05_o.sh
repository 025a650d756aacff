#!/bin/bash
# round 5, last session: classes longest first (keep_bdd_order 0) against first appearance (2), same box
for rep in 1 2; do for k in 0 2; do
  echo "== mixed set cover 3..16 keep=$k rep=$rep"; timeout 300 python tools/mixedcover.py --keep-order $k 2>&1 | grep -E "packs|iteration"
done; done
for k in 0 2; do
  echo "== mixed set cover 2..40 keep=$k"; timeout 300 python tools/mixedcover.py --kmin 2 --kmax 40 --rows 250000 --keep-order $k 2>&1 | grep -E "BDDs|packs|iteration"
  echo "== 20000 knapsack + 250000 covering rows keep=$k"; timeout 300 python tools/widebench.py --rows 20000 --cover-rows 250000 --iters 200 --keep-order $k 2>&1 | grep -E "iteration"
  echo "== 25000 rows of 18 keep=$k"; timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 --keep-order $k 2>&1 | grep -E "iteration"
done
echo "== headline"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
