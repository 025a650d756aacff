#!/bin/bash
# round 6: double beyond the caches' reach (k = 10, 21 M / 42 M nodes) — resident waves per CU are LDS-limited (eight packs per workgroup: 121 KB, one workgroup of 8 waves
# per CU); settings that hold 12-16 waves per CU instead, alternating with the rules' choice, two rounds
mkdir -p gpurun_out/r06z
for round in 1 2; do
for cfg in "2000000 150" "4000000 80"; do
  set -- $cfg
  for opt in "" "--wpb 4" "--wpb 4 --stage-cap 512" "--wpb 4 --stage-cap 448" "--wpb 4 --stage-cap 320" "--pack-width 64 --wpb 4" "--pack-width 64 --wpb 8" "--pack-width 64 --wpb 2"; do
    echo "V=$1 double [$opt]: $(timeout 600 python tools/kbench.py --mt 1 --precision double --vars $1 --rows $(($1/2)) --iters $2 $opt 2>/dev/null | tail -2 | tr '\n' ' ' | cut -c1-200)"
  done
done
done > gpurun_out/r06z/occ.txt 2>&1
wc -l gpurun_out/r06z/occ.txt
