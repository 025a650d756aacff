#!/bin/bash
# round 6: the last exchange of the MMA iteration on a second stream (L-BFGS loop): L-BFGS parity tests, rates, kernel stats
mkdir -p gpurun_out/r06h
python -m pytest tests/test_gpu_lbfgs_rounding.py -x -q 2>&1 | tail -3 > gpurun_out/r06h/lbfgs_tests.txt
for i in 1 2; do for p in float double; do echo "== $p: $(timeout 300 python tools/lbfgs_prof.py $p 200 2>&1 | tail -1)"; done; done > gpurun_out/r06h/rates.txt 2>&1
bash tools/kstats.sh r06h_f32 tools/lbfgs_prof.py float 200 > gpurun_out/r06h/kstats_f32.txt 2>&1
bash tools/kstats.sh r06h_f64 tools/lbfgs_prof.py double 200 > gpurun_out/r06h/kstats_f64.txt 2>&1
cat gpurun_out/r06h/lbfgs_tests.txt gpurun_out/r06h/rates.txt; head -12 gpurun_out/r06h/kstats_f32.txt | cut -c1-140; head -12 gpurun_out/r06h/kstats_f64.txt | cut -c1-140
