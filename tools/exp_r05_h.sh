# third generation on longer rows (several stage groups per pack, no resident headers): k = 20 / 32 / 100 at ~10.5 M nodes
for rep in 1 2; do
for cfg in "20 250000" "32 160000" "100 52000"; do
  set -- $cfg
  for prec in float double; do
    for var in 0 262144; do
      echo "$prec k=$1 rows=$2 variant=$var: $(timeout 600 python tools/kbench.py --precision $prec --vars 1000000 --rows $2 --k $1 --iters 200 --variant $var 2>/dev/null | tail -2 | tr '\n' ' ')"
    done
  done
done
done
