#!/bin/bash
mkdir -p gpurun_out/r06g
for a in "float" "float x" "double x"; do timeout 300 python tools/cold_sweeps.py $a; done > gpurun_out/r06g/cold_sweeps_x.txt 2>&1
cat gpurun_out/r06g/cold_sweeps_x.txt
