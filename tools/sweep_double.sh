#!/bin/bash
# layout options of the 10.5 M-node instance in double (gpurun_out/s2/sweep_10m_d.txt)
mkdir -p gpurun_out/s2
out=gpurun_out/s2/sweep_10m_d.txt; : > $out
for a in "" "--pack-width 64" "--pack-width 64 --wpb 8" "--pack-width 64 --wpb 2" "--wpb 2" "--wpb 8" "--pack-width 256" "--pack-width 256 --wpb 2"; do
  echo "## double $a" >> $out
  python tools/kbench.py --precision double $a 2>&1 | tail -2 >> $out
done
