#!/bin/bash
# round 6: previous iterate by swapping buffers (no prev <- cur copies in the store pass): parity tests + rates; CPU leg: schedule / binding of the oracle's loops
mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_lbfgs_rounding.py -x -q 2>&1 | tail -3 > gpurun_out/r06d/lbfgs_tests.txt
for i in 1 2; do for p in float double; do echo "== $p: $(timeout 300 python tools/lbfgs_prof.py $p 200 2>&1 | tail -1)"; done; done > gpurun_out/r06d/rates.txt 2>&1
bash tools/kstats.sh r06d tools/lbfgs_prof.py float 200 > gpurun_out/r06d/kstats.txt 2>&1
{
echo "== dynamic,64"; timeout 200 python tools/cpu_scaling.py float 4 2>&1 | grep -E "cpus|threads  *(32|64|128):"
echo "== static,512"; MMA_ORACLE_SCHEDULE=static timeout 200 python tools/cpu_scaling.py float 4 2>&1 | grep -E "threads  *(32|64|128):"
echo "== static,512 bind spread"; MMA_ORACLE_SCHEDULE=static OMP_PROC_BIND=spread OMP_PLACES=cores timeout 200 python tools/cpu_scaling.py float 4 2>&1 | grep -E "threads  *(32|64|128):"
echo "== dynamic bind spread"; OMP_PROC_BIND=spread OMP_PLACES=cores timeout 200 python tools/cpu_scaling.py float 4 2>&1 | grep -E "threads  *(32|64|128):"
} > gpurun_out/r06d/cpu_sched.txt 2>&1
cat gpurun_out/r06d/lbfgs_tests.txt gpurun_out/r06d/rates.txt gpurun_out/r06d/cpu_sched.txt; head -12 gpurun_out/r06d/kstats.txt | cut -c1-140
