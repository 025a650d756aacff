#!/bin/bash
mkdir -p gpurun_out/r06m
timeout 900 python -m pytest tests/test_gpu_small_fused.py tests/test_gpu_run_solver.py -x -q 2>&1 | tail -15 > gpurun_out/r06m/tests.txt
cat gpurun_out/r06m/tests.txt
