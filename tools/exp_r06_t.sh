#!/bin/bash
mkdir -p gpurun_out/r06t
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cuda_rule.py -x -q 2>&1 | tail -6 > gpurun_out/r06t/tests.txt
{
for v in 0 1048576 0 1048576; do echo "== widebench --rows 25000 --k 18 --variant $v"; timeout 300 python tools/widebench.py --rows 25000 --k 18 --iters 100 --variant $v 2>&1 | grep -E "iteration|fwd_plain"; done
} > gpurun_out/r06t/widebench.txt 2>&1
cat gpurun_out/r06t/tests.txt gpurun_out/r06t/widebench.txt
