#!/bin/bash
# run-to-run spread of the 10.5 M-node float iteration time in fresh processes: bash tools/repeat_float.sh N [ENV=VALUE ...]
n=$1; shift
for i in $(seq 1 $n); do
  env "$@" python tools/kbench.py 2>&1 | tail -n 1 | sed "s/^/[$*] /"
done
