#!/bin/bash
# it/s and per-kernel times of the random set-cover family around the Infinity-Cache size (256 MiB): is the 10.5 M-node headline helped by it?
# usage: tools/size_curve.sh [precision] ; run through gpurun
P=${1:-float}
for v in 400000 600000 700000 800000 900000 1000000 1100000 1200000 1400000 1600000 2000000; do
  PYTHONPATH=. python tools/kbench.py --precision $P --mt 1 --vars $v --rows $((v/2)) --iters 300 2>&1 | tail -2 | tr '\n' ' '
  echo " V=$v"
done
