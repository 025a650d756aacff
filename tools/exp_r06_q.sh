#!/bin/bash
mkdir -p gpurun_out/r06q
for i in 1 2; do for v in base n3la3; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  echo "[$v] $(timeout 300 python tools/kbench.py --mt 1 --precision double --iters 1000 2>&1 | tail -1)"
done; done > gpurun_out/r06q/la3.txt 2>&1
cat gpurun_out/r06q/la3.txt
