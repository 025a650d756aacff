#!/bin/bash
# round 5, last session: 64-slot packs for every instance with chained (staggered) packs?  mixed instances, automatic width (128) against --pack-width 64
for a in "--rows 20000 --cover-rows 250000" "--rows 10000 --cover-rows 400000" "--rows 30000 --cover-rows 100000" "--rows 40000" "--rows 5000 --cover-rows 450000"; do
for rep in 1 2; do
  echo "== $a auto rep=$rep"; timeout 300 python tools/widebench.py $a --iters 200 2>&1 | grep -E "layout|iteration"
  echo "== $a --pack-width 64 rep=$rep"; timeout 300 python tools/widebench.py $a --pack-width 64 --iters 200 2>&1 | grep -E "layout|iteration"
done; done
