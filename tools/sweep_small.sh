mkdir -p gpurun_out/s2
out=gpurun_out/s2/sweep_1m.txt; : > $out
for pw in 64 128; do for wpb in 1 2 4 8; do for res in 1 2; do
  echo "## pw $pw wpb $wpb res $res" >> $out
  python tools/kbench.py --vars 100000 --rows 50000 --pack-width $pw --wpb $wpb --res $res 2>&1 | tail -2 >> $out
done; done; done
echo "## default" >> $out
python tools/kbench.py --vars 100000 --rows 50000 2>&1 | tail -2 >> $out
