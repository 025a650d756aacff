#!/bin/bash
# round 6: k_project_entries keeps a bin's entries in registers (one read of the vector instead of two): L-BFGS parity, rates, kernel stats
mkdir -p gpurun_out/r06y
timeout 900 python -m pytest tests/test_gpu_lbfgs_rounding.py tests/test_gpu_small_fused.py -x -q 2>&1 | tail -3 > gpurun_out/r06y/tests.txt
for i in 1 2; do for p in float double; do echo "== $p: $(timeout 300 python tools/lbfgs_prof.py $p 200 2>&1 | tail -1)"; done; done > gpurun_out/r06y/rates.txt 2>&1
bash tools/kstats.sh r06y_f32 tools/lbfgs_prof.py float 200 > gpurun_out/r06y/kstats_f32.txt 2>&1
bash tools/kstats.sh r06y_f64 tools/lbfgs_prof.py double 200 > gpurun_out/r06y/kstats_f64.txt 2>&1
cat gpurun_out/r06y/tests.txt gpurun_out/r06y/rates.txt; grep -h "project_entries" gpurun_out/r06y/kstats_f32.txt gpurun_out/r06y/kstats_f64.txt | cut -c1-140
