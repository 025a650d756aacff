#!/bin/bash
# round 6: wide solve sweeps with two-hop prefetch distances (kernels/wide3.hpp) against the wide2 form (variant_flags bit 20 = 1048576)
mkdir -p gpurun_out/r06s
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06s/tests.txt
{
for a in "--rows 25000 --k 18 --iters 100" "--rows 4000" "--rows 40000" "--rows 20000 --cover-rows 250000"; do
  for v in 0 1048576 0 1048576; do echo "== widebench $a --variant $v"; timeout 300 python tools/widebench.py $a --variant $v 2>&1 | grep -E "iteration|fwd_plain"; done
done
} > gpurun_out/r06s/widebench.txt 2>&1
cat gpurun_out/r06s/tests.txt gpurun_out/r06s/widebench.txt
