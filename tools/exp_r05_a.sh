./build/atomics_rate 1563 320 100000
./build/atomics_rate 7813 640 1000000
for V in 2000000 4000000; do
 for opt in "" "--pack-width 256" "--pack-width 64" "--wpb 8" "--wpb 8 --pack-width 256" "--wpb 2 --pack-width 256"; do
  echo "V=$V [$opt] $(python tools/kbench.py --mt 1 --vars $V --rows $((V/2)) --iters 100 $opt 2>/dev/null | tail -2 | tr '\n' ' ')"
 done
done
