#!/bin/bash
# round 6: the stage-group rules as shipped (stage_cap = 0: one round of workgroups, else a third workgroup per CU) against stage_cap = 640, alternating, two rounds
mkdir -p gpurun_out/r06z
for round in 1 2; do
for nodes in 7500000 10500000 15000000; do
for k in 24 28 32 36 44; do
  rows=$((nodes / (2 * k + 1))); vars=$((2 * rows))
  for prec in double; do
  for sc in 640 0; do
    echo "nodes=$nodes k=$k $prec stage_cap=$sc: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $k --vars $vars --rows $rows --stage-cap $sc --iters 200 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
  done
  done
done
done
done > gpurun_out/r06z/stage_cap6.txt 2>&1
wc -l gpurun_out/r06z/stage_cap6.txt
