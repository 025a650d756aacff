#!/bin/bash
mkdir -p gpurun_out/r06r
for i in 1 2 3; do for v in base n3la1; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  timeout 300 python tools/objects_rate.py double 6 2>&1 | grep -v amdgpu
done; done > gpurun_out/r06r/la_objects.txt 2>&1
cat gpurun_out/r06r/la_objects.txt
