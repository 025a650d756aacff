#!/bin/bash
# layout-option sweeps on the final round-3 kernels (gpurun_out/s8/*.txt): 1.05 M nodes float (pack width x packs per workgroup x resident,
# variables per bin), 10.5 M nodes float / double (packs per workgroup, pack width, stage cap)
mkdir -p gpurun_out/s8
out=gpurun_out/s8/sweep_1m.txt; : > $out
for pw in 64 128; do for wpb in 1 2 4; do for res in 1 2; do
  echo "## pw $pw wpb $wpb res $res" >> $out
  python tools/kbench.py --vars 100000 --rows 50000 --pack-width $pw --wpb $wpb --res $res 2>&1 | tail -2 >> $out
done; done; done
for vb in 0 256 384 512 768 1024; do
  echo "## default vb $vb" >> $out
  python tools/kbench.py --vars 100000 --rows 50000 --vars-per-bin $vb 2>&1 | tail -2 >> $out
done
out=gpurun_out/s8/sweep_10m.txt; : > $out
for p in float double; do for a in "" "--wpb 2" "--wpb 8" "--pack-width 64" "--pack-width 256" "--stage-cap 320" "--stage-cap 384"; do
  echo "## $p $a" >> $out
  python tools/kbench.py --precision $p --mt 1 $a 2>&1 | tail -2 >> $out
done; done
