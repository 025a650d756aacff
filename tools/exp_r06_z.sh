#!/bin/bash
# round 6: longer rows (k = 32, 100; 64-slot packs, second generation) — stage groups of fewer layers (less LDS per pack, more rounds) and packs per workgroup
mkdir -p gpurun_out/r06z
for cfg in "32 323076 161538" "100 104476 52238"; do
  set -- $cfg
  for prec in double float; do
    for sc in 0 64 128 256 384; do
      echo "k=$1 $prec stage_cap=$sc wpb=0: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $1 --vars $2 --rows $3 --stage-cap $sc --iters 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
    done
    for wpb in 1 2 8; do
      echo "k=$1 $prec stage_cap=0 wpb=$wpb: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $1 --vars $2 --rows $3 --wpb $wpb --iters 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
    done
  done
done > gpurun_out/r06z/stage_cap.txt 2>&1
cat gpurun_out/r06z/stage_cap.txt
