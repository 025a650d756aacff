# third-generation sweeps: hops per trip of the hop loop (BDDMMA_HOP_UNROLL 1 / 2 (shipped) / 4)
for rep in 1 2 3; do
for cfg in "float 400000" "float 1000000" "double 1000000"; do
  set -- $cfg
  for lib in "" hu1 hu4; do
    if [ -z "$lib" ]; then e=""; else e="BDDMMA_LIB=build/lib$lib.so"; fi
    echo "$1 V=$2 lib=[$lib]: $(timeout 600 env $e python tools/kbench.py --mt 1 --precision $1 --vars $2 --rows $(($2/2)) --iters 300 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
done
