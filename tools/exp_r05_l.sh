# third generation, two stage groups per pack (half the staging LDS per wave -> more waves per CU, two staging rounds per quad)
for rep in 1 2; do
for prec in double float; do
  for opt in "" "--stage-cap 320" "--stage-cap 320 --wpb 8" "--wpb 8"; do
    echo "$prec [$opt]: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --iters 300 $opt 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
done
