#!/bin/bash
# round 6: does the one-round boundary show in the third-generation sweeps (k = 10, double: 4 workgroups of 4 packs per CU = 1 024 workgroups = 4 096 packs = V 524 k)?
mkdir -p gpurun_out/r06z
for v in 400000 440000 480000 500000 520000 530000 540000 560000 600000 700000 800000; do
  for prec in double float; do
    echo "V=$v $prec: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --vars $v --rows $((v/2)) --iters 300 2>/dev/null | tail -2 | tr '\n' ' ' | cut -c1-200)"
  done
done > gpurun_out/r06z/quant.txt 2>&1
