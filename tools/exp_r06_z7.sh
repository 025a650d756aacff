#!/bin/bash
# round 6: double, rows of 24-36 variables (several rounds of workgroups): packs per workgroup x group size around the shipped choice (four packs, 576 layers)
mkdir -p gpurun_out/r06z
for nodes in 7500000 10500000 15000000; do
for k in 24 28 32 36; do
  rows=$((nodes / (2 * k + 1))); vars=$((2 * rows))
  for opt in "" "--wpb 2 --stage-cap 640" "--wpb 2 --stage-cap 576" "--wpb 2 --stage-cap 448" "--wpb 1 --stage-cap 448" "--wpb 1 --stage-cap 384" "--wpb 4 --stage-cap 512"; do
    echo "nodes=$nodes k=$k [$opt]: $(timeout 300 python tools/kbench.py --mt 1 --precision double --k $k --vars $vars --rows $rows --iters 200 $opt 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
  done
done
done > gpurun_out/r06z/stage_cap7.txt 2>&1
wc -l gpurun_out/r06z/stage_cap7.txt
