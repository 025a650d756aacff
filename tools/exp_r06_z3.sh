#!/bin/bash
# round 6: longer rows, double — stage_cap in steps of 64 (which group sizes the second-generation sweeps like)
mkdir -p gpurun_out/r06z
for k in 24 32 40 50 64 80; do
  rows=$((10500000 / (2 * k + 1))); vars=$((2 * rows))
  for prec in double; do
    for sc in 320 384 448 512 576 640; do
      echo "k=$k $prec stage_cap=$sc: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $k --vars $vars --rows $rows --stage-cap $sc --iters 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
    done
  done
done > gpurun_out/r06z/stage_cap3.txt 2>&1
cat gpurun_out/r06z/stage_cap3.txt
