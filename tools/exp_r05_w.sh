#!/bin/bash
# round 5, last session: the shipped rule (64-slot packs where chained BDDs hold most narrow nodes; 96 VGPRs for 128-slot packs swept one per workgroup)
# against --pack-width 128, both precisions
for a in "--rows 100000 --k 11" "--rows 150000 --k 10" "--rows 200000 --k 9"; do
  echo "== $a auto"; timeout 300 python tools/widebench.py $a --iters 200 2>&1 | grep -E "layout|iteration"
  echo "== $a --pack-width 128"; timeout 300 python tools/widebench.py $a --pack-width 128 --iters 200 2>&1 | grep -E "layout|iteration"
done
