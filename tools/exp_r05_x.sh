#!/bin/bash
# round 5, last session: set cover with rows of 3-16 variables — which sweeps run, packs per workgroup, second generation
for o in "" "--wpb 8" "--wpb 2" "--variant 0x40000" "--pack-width 64"; do
  echo "== mixedcover 3..16 $o"; timeout 300 python tools/mixedcover.py $o --precision float 2>&1 | grep -E "packs|iteration"
done
echo "== uniform k=10 same size"; timeout 300 python tools/kbench.py --vars 1000000 --rows 533000 --iters 200 2>&1 | tail -2
