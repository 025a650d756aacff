#!/bin/bash
# round 6: longer rows (k = 20, 32, 100) — packs of 64 slots (the rule below 2048 x hops / 16 packs: second generation) against packs of 128 (third generation)
mkdir -p gpurun_out/r06w
for cfg in "20 512195 256097" "32 323076 161538" "100 104476 52238"; do
  set -- $cfg
  for prec in float double; do for pw in 0 64 128; do
    echo "k=$1 $prec pack_width=$pw: $(timeout 300 python tools/kbench.py --mt 1 --precision $prec --k $1 --vars $2 --rows $3 --pack-width $pw --iters 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
  done; done
done > gpurun_out/r06w/pack_width.txt 2>&1
cat gpurun_out/r06w/pack_width.txt
