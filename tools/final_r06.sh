#!/bin/bash
# round 6, final collection on the shipped sources, part 1: GPU tests, rocprofv3 kernel stats + PMC passes of bench.py (10.5 M and 1.05 M nodes), kernel stats
# of the L-BFGS loop.  Part 2 (tools/final_r06b.sh): the HBM-only figure, the bench lines, the size sweep, small-instance rates.
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/final/gputest.txt
bash tools/profile.sh r06_10m_f32 > gpurun_out/final/profile_10m.log 2>&1
bash tools/profile.sh r06_1m_f32 --vars 100000 --rows 50000 > gpurun_out/final/profile_1m.log 2>&1
bash tools/kstats.sh r06_lbfgs_f32 tools/lbfgs_prof.py float 200 > gpurun_out/final/kstats_lbfgs_f32.txt 2>&1
bash tools/kstats.sh r06_lbfgs_f64 tools/lbfgs_prof.py double 200 > gpurun_out/final/kstats_lbfgs_f64.txt 2>&1
# does a --pmc pass over the fused small-instance kernel return at all?  (bounded: 90 s)
timeout 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_small -o pmc -- python tools/small_phases.py > gpurun_out/final/pmc_small.txt 2>&1; echo "pmc over k_iterate_small: exit $?" >> gpurun_out/final/pmc_small.txt
rm -rf gpurun_out/final/pmc_small
cat gpurun_out/final/gputest.txt; tail -3 gpurun_out/final/pmc_small.txt; ls gpurun_out/prof_r06_10m_f32 gpurun_out/prof_r06_1m_f32
