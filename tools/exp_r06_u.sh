#!/bin/bash
# round 6: neighbouring potentials with one load in the third-generation sweeps (forward: the copy of the next hop's costs-from-terminal; backward: the
# layer's two costs-from-root): parity, then same-box A/B against the library built from the sources before (build/libn3old.so), six solver objects per process
mkdir -p gpurun_out/r06u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cuda_rule.py tests/test_gpu_lbfgs_rounding.py -x -q 2>&1 | tail -3 > gpurun_out/r06u/tests.txt
for i in 1 2 3; do for v in base n3old; do for p in float double; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  timeout 300 python tools/objects_rate.py $p 4 2>&1 | grep -v amdgpu
done; done; done > gpurun_out/r06u/ab.txt 2>&1
for v in base n3old; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  echo "[$v] $(timeout 300 python tools/kbench.py --mt 1 --iters 1000 2>&1 | tail -2 | tr '\n' ' ')"
  echo "[$v] 4.2M: $(timeout 300 python tools/kbench.py --mt 1 --vars 400000 --rows 200000 --iters 1000 2>&1 | tail -2 | tr '\n' ' ')"
done >> gpurun_out/r06u/ab.txt 2>&1
cat gpurun_out/r06u/tests.txt gpurun_out/r06u/ab.txt
