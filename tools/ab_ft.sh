#!/bin/bash
# non-temporal potentials stores (double, footprint > 640 MB) on / off (variant_flags bit 2) across instance sizes: gpurun_out/s14/ab.txt
for rep in 1 2; do
for cfg in "10 500000 1000000" "32 161538 323076" "4 1166666 2333332" "10 200000 400000" "10 50000 100000" "10 1000000 2000000"; do
  set -- $cfg
  for v in 0 4; do
    echo "k=$1 B=$2 double variant $v: $(python tools/kbench.py --k $1 --rows $2 --vars $3 --precision double --variant $v 2>&1 | tail -n 1)"
  done
done
done
for v in 0 4; do echo "float 10.5M variant $v: $(python tools/kbench.py --variant $v 2>&1 | tail -n 1)"; done
