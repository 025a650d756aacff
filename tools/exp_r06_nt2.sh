#!/bin/bash
# round 6: the non-temporal instantiations of the first / second generation as shipped (rule: beyond 640 MiB) against a build without them, alternating
mkdir -p gpurun_out/r06nt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "nontemporal or lane_per_layer_sweeps_are_the_rule or long_bdds" 2>&1 | tail -3 > gpurun_out/r06nt/tests.txt
{
for round in 1 2; do
for cfg in "2000000 200" "4000000 100" "10000000 40"; do
  set -- $cfg
  for prec in float double; do
    for lib in without shipped; do
      if [ $lib = shipped ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/libnont.so; fi
      echo "V=$1 $prec $lib: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --vars $1 --rows $(($1/2)) --iters $2 2>/dev/null | tail -2 | tr '\n' ' ')"
    done
  done
done
done
} > gpurun_out/r06nt/nt2.txt 2>&1
cat gpurun_out/r06nt/tests.txt gpurun_out/r06nt/nt2.txt
