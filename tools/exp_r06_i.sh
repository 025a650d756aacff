#!/bin/bash
# round 6: whole iterations in one launch for instances that fit one workgroup (k_iterate_small): parity tests, then iteration rates
mkdir -p gpurun_out/r06i
timeout 600 python -m pytest tests/test_gpu_small_fused.py -x -q 2>&1 | tail -25 > gpurun_out/r06i/tests.txt
timeout 300 python tools/small_rate.py > gpurun_out/r06i/rates.txt 2>&1
cat gpurun_out/r06i/tests.txt gpurun_out/r06i/rates.txt
