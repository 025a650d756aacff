#!/bin/bash
# round 6: the one run_solver mismatch of the soak (defaults-double: (53, 0) vs (400, 0)) — eight processes looping that case next to four fuzz processes,
# everything printed on a mismatch
mkdir -p gpurun_out/r06v
for p in 1 2 3 4; do python tests/tools/stress_fuzz.py 10 $(seq 0 39) > gpurun_out/r06v/fuzz$p.txt 2>&1 & done
for p in 1 2 3 4 5 6 7 8; do python tests/tools/stress_run_solver.py 400 double > gpurun_out/r06v/p$p.txt 2>&1 & done
wait
grep -h "MISMATCH\|mismatches\|failures in" gpurun_out/r06v/*.txt | cut -c1-900
