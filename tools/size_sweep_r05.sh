#!/bin/bash
# instance-size sweep on the round-5 sources (random set cover, k = 10, V = 2 B): it/s and the kernels that ran (tools/kbench.py)
for cfg in "100000 400" "400000 400" "1000000 400" "2000000 200" "4000000 100" "10000000 40"; do
  set -- $cfg
  for prec in float double; do
    echo "V=$1 $prec: $(timeout 900 python tools/kbench.py --mt 1 --precision $prec --vars $1 --rows $(($1/2)) --iters $2 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
