#!/bin/bash
# round 6: the one-round rule for the stage groups (layout.cpp) against the size it replaces (stage_cap = 640 given explicitly), alternating, both precisions
mkdir -p gpurun_out/r06z
for nodes in 10500000 5250000 21000000; do
for k in 24 28 32 40 44 50 56 64 80 100; do
  rows=$((nodes / (2 * k + 1))); vars=$((2 * rows))
  for prec in double float; do
  for sc in 640 0; do
    echo "nodes=$nodes k=$k $prec stage_cap=$sc: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $k --vars $vars --rows $rows --stage-cap $sc --iters 200 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
  done
  done
done
done > gpurun_out/r06z/stage_cap5.txt 2>&1
wc -l gpurun_out/r06z/stage_cap5.txt
