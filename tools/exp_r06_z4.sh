#!/bin/bash
# round 6: longer rows, double — a denser grid (row length x stage_cap x instance size) for a cost model of the stage groups (profiles/r06_stage_groups.txt)
mkdir -p gpurun_out/r06z
for nodes in 10500000 5250000 21000000; do
for k in 24 28 32 36 40 44 50 56 64 80 100; do
  rows=$((nodes / (2 * k + 1))); vars=$((2 * rows))
  for sc in 320 384 448 512 576 640; do
    echo "nodes=$nodes k=$k stage_cap=$sc: $(timeout 300 python tools/kbench.py --mt 1 --precision double --k $k --vars $vars --rows $rows --stage-cap $sc --iters 200 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
  done
done
done > gpurun_out/r06z/stage_cap4.txt 2>&1
wc -l gpurun_out/r06z/stage_cap4.txt
