#!/bin/bash
# round 6, baseline on the round-5 sources: GPU tests, default bench line, L-BFGS rates
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/r06a/gputest.txt
python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
for p in float double; do for i in 1 2; do python tools/lbfgs_prof.py $p 200; done; done > gpurun_out/r06a/lbfgs.txt 2>&1
cat gpurun_out/r06a/gputest.txt gpurun_out/r06a/lbfgs.txt; cut -c1-400 gpurun_out/r06a/bench_default.json
