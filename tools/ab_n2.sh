#!/bin/bash
# A/B on one box, three rounds: first-generation streaming sweeps (variant_flags 0x1000), second generation with the register cap (shipped), without
for rep in 1 2 3; do
for prec in float double; do
  a=$(python tools/kbench.py --mt 1 --precision $prec --variant 4096 --iters 400 2>/dev/null | tail -1)
  b=$(python tools/kbench.py --mt 1 --precision $prec --iters 400 2>/dev/null | tail -1)
  c=$(BDDMMA_LIB=build/libn2nocap.so python tools/kbench.py --mt 1 --precision $prec --iters 400 2>/dev/null | tail -1)
  echo "$prec  gen1: $a | gen2 capped: $b | gen2 uncapped: $c"
done
done
