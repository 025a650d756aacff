# -DBDDMMA_EXP_INT_ATOMIC (adds the doubles' bit patterns with integer LDS atomics).  NOT a measurement of integer atomics: the garbage sums turn
# every cost into NaN within an iteration, the sweeps then defer 0 everywhere and the exchange skips all its atomics — what this measures is
# the exchange WITHOUT its accumulation (15.5 instead of 19.5 us at 10.5 M nodes).  A real fixed-point exchange (ds_add_u64 of converted
# floats, round 5) accumulated in exactly the time of the ds_add_f64 form (per-wave stamps: 9.70 vs 9.66 us) and was removed.
for rep in 1 2 3; do
for cfg in "float 1000000" "float 400000" "float 4000000"; do
  set -- $cfg
  for lib in "" intat; do
    if [ -z "$lib" ]; then e=""; else e="BDDMMA_LIB=build/lib$lib.so"; fi
    echo "$1 V=$2 lib=[$lib]: $(timeout 600 env $e python tools/kbench.py --mt 1 --precision $1 --vars $2 --rows $(($2/2)) --iters 300 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
done
