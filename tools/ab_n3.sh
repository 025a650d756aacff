#!/bin/bash
# A/B on one box: third-generation streaming sweeps (a lane per layer) against the second / first generation (--variant 262144 = bit 18)
for rep in 1 2; do
for prec in float double; do
  for V in ${VS:-1000000}; do
    for var in 0 262144; do
      echo "$prec V=$V variant=$var: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --vars $V --rows $((V/2)) --iters ${ITERS:-400} --variant $var 2>/dev/null | tail -2 | tr '\n' ' ')"
    done
  done
done
done
