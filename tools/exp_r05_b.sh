# HBM-only regime (V = 2 M / 4 M): second-generation streaming sweeps with deeper look-ahead and the register cap that LDS allows anyway; resident sweeps forced
run() { echo "$1 [$2] $(timeout 300 env $3 python tools/kbench.py --mt 1 --iters 100 $2 2>/dev/null | tail -2 | tr '\n' ' ')"; }
for V in 2000000 4000000; do
  A="--vars $V --rows $((V/2))"
  run "V=$V first-gen" "$A" ""
  run "V=$V narrow2" "$A --variant 8192" ""
  run "V=$V narrow2 wpb8" "$A --variant 8192 --wpb 8" ""
  run "V=$V la2w4" "$A --variant 8192" "BDDMMA_LIB=build/libla2w4.so"
  run "V=$V la2w4 wpb8" "$A --variant 8192 --wpb 8" "BDDMMA_LIB=build/libla2w4.so"
  run "V=$V la3w3" "$A --variant 8192" "BDDMMA_LIB=build/libla3w3.so"
  run "V=$V la3w3 wpb8" "$A --variant 8192 --wpb 8" "BDDMMA_LIB=build/libla3w3.so"
  run "V=$V res2 pw64 wpb1" "$A --res 2 --pack-width 64 --wpb 1" ""
  run "V=$V res2 pw64 wpb2" "$A --res 2 --pack-width 64 --wpb 2" ""
  run "V=$V res2 pw64 wpb4" "$A --res 2 --pack-width 64 --wpb 4" ""
  run "V=$V res pw128 wpb4" "$A --res 2 --pack-width 128 --wpb 4" ""
done
run "V=1M la2w4" "" "BDDMMA_LIB=build/libla2w4.so"
run "V=1M default" "" ""
run "V=1M double la2w4" "--precision double" "BDDMMA_LIB=build/libla2w4.so"
run "V=1M double default" "--precision double" ""
