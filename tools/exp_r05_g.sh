# third-generation sweeps: look-ahead 1 (shipped) / 2 / 3 hops
for rep in 1 2; do
for cfg in "float 400000" "float 1000000" "float 4000000" "double 1000000" "double 4000000"; do
  set -- $cfg
  for lib in "" n3la2 n3la3; do
    if [ -z "$lib" ]; then e=""; else e="BDDMMA_LIB=build/lib$lib.so"; fi
    echo "$1 V=$2 lib=[$lib]: $(timeout 600 env $e python tools/kbench.py --mt 1 --precision $1 --vars $2 --rows $(($2/2)) --iters 200 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
done
