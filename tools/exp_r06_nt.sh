#!/bin/bash
# round 6: non-temporal loads in the first / second generation of streaming sweeps beyond the caches' reach — build knobs, alternating with the shipped library
mkdir -p gpurun_out/r06nt
{
for round in 1 2; do
for cfg in "2000000 200" "4000000 100" "10000000 40"; do
  set -- $cfg
  for prec in float double; do
    for lib in shipped ntpot nttab ntboth; do
      if [ $lib = shipped ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$lib.so; fi
      echo "V=$1 $prec $lib: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --vars $1 --rows $(($1/2)) --iters $2 2>/dev/null | tail -2 | tr '\n' ' ')"
    done
  done
done
done
} > gpurun_out/r06nt/nt.txt 2>&1
cat gpurun_out/r06nt/nt.txt
