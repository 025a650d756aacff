#!/bin/bash
# A/B on one box, three rounds: lookahead depth / register cap of the second-generation streaming sweeps (libraries from tools/build_variant.sh)
for rep in 1 2 3; do
for prec in ${PRECS:-double}; do
  for lib in "" $LIBS; do
    if [ -z "$lib" ]; then r=$(python tools/kbench.py --mt 1 --precision $prec --iters 400 $KARGS 2>/dev/null | tail -1); lib=shipped
    else r=$(BDDMMA_LIB=build/lib$lib.so python tools/kbench.py --mt 1 --precision $prec --iters 400 $KARGS 2>/dev/null | tail -1); fi
    echo "$prec $lib: $r"
  done
done
done
