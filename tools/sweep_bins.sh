#!/bin/bash
# variables per exchange bin at 10.5 M nodes, float and double (gpurun_out/s2/sweep_bins.txt)
mkdir -p gpurun_out/s2
out=gpurun_out/s2/sweep_bins.txt; : > $out
for p in float double; do for vb in 0 2048 3072 6144 8192 9728; do
  echo "## $p vars_per_bin $vb" >> $out
  python tools/kbench.py --precision $p --vars-per-bin $vb 2>&1 | tail -2 >> $out
done; done
