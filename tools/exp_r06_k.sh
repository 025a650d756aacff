#!/bin/bash
mkdir -p gpurun_out/r06k
timeout 600 python -m pytest tests/test_gpu_small_fused.py -x -q 2>&1 | tail -25 > gpurun_out/r06k/tests.txt
timeout 300 python tools/small_rate.py > gpurun_out/r06k/rates.txt 2>&1
python tools/small_phases.py 2>&1 | grep -v amdgpu > gpurun_out/r06k/phases.txt
cat gpurun_out/r06k/tests.txt gpurun_out/r06k/rates.txt gpurun_out/r06k/phases.txt
