# bins of 4096 (rule) / 2048 / 1024 variables with the third-generation sweeps: two workgroups per CU overlap the exchange's phases, the sweeps' staged runs halve
for rep in 1 2; do
for vpb in 0 2048 1024; do
  echo "float vars-per-bin=$vpb: $(timeout 600 python tools/kbench.py --mt 1 --precision float --iters 300 --vars-per-bin $vpb 2>/dev/null | tail -2 | tr '\n' ' ')"
done
done
