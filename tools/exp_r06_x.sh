#!/bin/bash
mkdir -p gpurun_out/r06x
timeout 600 python -m pytest tests/test_gpu_small_fused.py -x -q 2>&1 | tail -15 > gpurun_out/r06x/tests.txt
cat gpurun_out/r06x/tests.txt
