#!/bin/bash
# same-box A/B: the scheduled exchange (default) against the LDS-atomic kernels (variant_flags bit 17)
for P in float double; do
 for V in 100000 1000000; do
  for VAR in 0 131072 0 131072; do
   echo "== $P V=$V variant=$VAR"; PYTHONPATH=. python tools/kbench.py --precision $P --mt 1 --vars $V --rows $((V/2)) --variant $VAR --iters 400 2>&1 | tail -2
  done
 done
done
