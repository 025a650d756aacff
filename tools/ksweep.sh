#!/bin/bash
# Hop-depth sensitivity: random set cover with row size k, ~10.5 M BDD nodes at every k.  GPU box: bash tools/ksweep.sh > gpurun_out/k_sweep.txt
for cfg in "4 1166666 2333332" "10 500000 1000000" "32 161538 323076" "100 52238 104476"; do
  set -- $cfg
  for p in float double; do
    echo "k=$1 B=$2 V=$3 $p: $(python tools/kbench.py --k $1 --rows $2 --vars $3 --precision $p 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
