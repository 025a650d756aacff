#!/bin/bash
# A/B on one box, three alternating rounds: the library of the last commit (build/libhead.so, built from a git worktree of HEAD) against the working tree's
for rep in 1 2 3; do
for prec in ${PRECS:-float double}; do
  a=$(BDDMMA_LIB=build/libhead.so python tools/kbench.py --mt 1 --precision $prec --iters 400 $KARGS 2>/dev/null | tail -1)
  b=$(python tools/kbench.py --mt 1 --precision $prec --iters 400 $KARGS 2>/dev/null | tail -1)
  echo "$prec  HEAD: $a"
  echo "$prec  tree: $b"
done
done
