#!/bin/bash
mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06l/tests.txt
cat gpurun_out/r06l/tests.txt
