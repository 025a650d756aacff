#!/bin/bash
# round 6: (1) L-BFGS with the streamed vectors (history, previous iterate) in uncached / fine-grained device memory — do the solver's sweeps get
# their out-of-loop speed back when the history does not allocate in the caches?  kernel stats of the loop per variant; (2) CPU leg thread scaling
mkdir -p gpurun_out/r06c
for v in base lb_uc lb_fg; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  for p in float double; do echo "== $v $p: $(timeout 300 python tools/lbfgs_prof.py $p 200 2>&1 | tail -1)"; done
done > gpurun_out/r06c/rates.txt 2>&1
for v in base lb_uc; do
  if [ $v = base ]; then unset BDDMMA_LIB; else export BDDMMA_LIB=build/lib$v.so; fi
  bash tools/kstats.sh r06c_$v tools/lbfgs_prof.py float 200 > gpurun_out/r06c/kstats_$v.txt 2>&1
done
unset BDDMMA_LIB
timeout 300 python tools/cpu_scaling.py float 3 > gpurun_out/r06c/cpu_scaling.txt 2>&1
cat gpurun_out/r06c/rates.txt gpurun_out/r06c/cpu_scaling.txt
