#!/bin/bash
# round 6: long BDDs (hundreds of hops, few packs): look-ahead of the streaming sweeps' hop pipeline 1 (shipped) / 2 / 4 as build variants
mkdir -p gpurun_out/r06o
for v in base la2 la4; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  for a in "cover 100" "cover 300" "assign 1000" "assign 1000 0" "cover 10"; do echo "[$v] $(timeout 600 python tools/shape_bench.py $a 2>&1 | grep -v amdgpu | cut -c1-200)"; done
done > gpurun_out/r06o/lookahead.txt 2>&1
cat gpurun_out/r06o/lookahead.txt
