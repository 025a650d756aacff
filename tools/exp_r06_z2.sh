#!/bin/bash
# round 6: longer rows — stage groups of EQUAL size (same number of rounds, less LDS than the 640-layer cap) and two packs per workgroup
mkdir -p gpurun_out/r06z
for cfg in "32 512" "50 576" "64 512" "24 384" "100 640"; do
  set -- $cfg
  k=$1; bal=$2; rows=$((10500000 / (2 * k + 1))); vars=$((2 * rows))
  for prec in double float; do
    for opt in "0 0" "$bal 0" "0 2" "$bal 2" "$bal 1"; do
      set -- $opt
      echo "k=$k $prec stage_cap=$1 wpb=$2: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $k --vars $vars --rows $rows --stage-cap $1 --wpb $2 --iters 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
    done
  done
done > gpurun_out/r06z/stage_cap2.txt 2>&1
cat gpurun_out/r06z/stage_cap2.txt
