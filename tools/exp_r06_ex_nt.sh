#!/bin/bash
# round 6: cache policy of the exchange beyond the caches' reach — non-temporal loads of the differences / variable indices (BDDMMA_EXP_EX_LD_AUX=2) and
# non-temporal stores of the broadcast pairs (BDDMMA_EX_ST_AUX=2), build knobs, alternating with the shipped library
mkdir -p gpurun_out/r06nt
{
for round in 1 2; do
for cfg in "1000000 300" "2000000 200" "4000000 100" "10000000 40"; do
  set -- $cfg
  for prec in float double; do
    for lib in shipped exld exst exboth; do
      if [ $lib = shipped ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$lib.so; fi
      echo "V=$1 $prec $lib: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --vars $1 --rows $(($1/2)) --iters $2 2>/dev/null | tail -2 | tr '\n' ' ')"
    done
  done
done
done
} > gpurun_out/r06nt/ex_nt.txt 2>&1
cat gpurun_out/r06nt/ex_nt.txt
