#!/bin/bash
mkdir -p gpurun_out/r06e
for p in float double; do timeout 300 python tools/cold_sweeps.py $p; done > gpurun_out/r06e/cold_sweeps.txt 2>&1
cat gpurun_out/r06e/cold_sweeps.txt
