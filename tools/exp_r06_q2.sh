#!/bin/bash
# round 6: the cliff between V = 520 k and 530 k (k = 10): the bin size of the exchange crosses 2 048 variables there (auto = V / 256)
mkdir -p gpurun_out/r06z
for v in 500000 520000 530000 560000 600000 700000; do
  for prec in double float; do
    for vb in 0 1024 1536 2048; do
      echo "V=$v $prec vars_per_bin=$vb: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --vars $v --rows $((v/2)) --iters 300 --vars-per-bin $vb 2>/dev/null | tail -2 | tr '\n' ' ' | cut -c1-200)"
    done
  done
done > gpurun_out/r06z/quant2.txt 2>&1
