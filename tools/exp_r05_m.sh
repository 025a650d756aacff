# run length of the staging in the real solver, 42 M nodes (V = 4 M): bins of 2 048 / 4 096 (rule) / 8 192 variables, four / eight packs per workgroup
for rep in 1 2; do
for opt in "--vars-per-bin 2048" "" "--vars-per-bin 8192" "--vars-per-bin 8192 --wpb 4" "--wpb 4"; do
  echo "float V=4M [$opt]: $(timeout 600 python tools/kbench.py --mt 1 --precision float --vars 4000000 --rows 2000000 --iters 100 $opt 2>/dev/null | tail -2 | tr '\n' ' ')"
done
done
