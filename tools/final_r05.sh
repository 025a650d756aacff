#!/bin/bash
# round 5, final collection on the shipped sources: GPU tests, rocprofv3 kernel stats + PMC passes of bench.py (10.5 M and 1.05 M nodes), kernel stats of
# the general-row benchmark and of the L-BFGS loop, the default bench line, the general-row table
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/final/gputest.txt
bash tools/profile.sh r05_10m_f32 > gpurun_out/final/profile_10m.log 2>&1
bash tools/profile.sh r05_1m_f32 --vars 100000 --rows 50000 > gpurun_out/final/profile_1m.log 2>&1
bash tools/kstats.sh r05_wide10m tools/widebench.py --rows 40000 --iters 100 > gpurun_out/final/kstats_wide10m.txt 2>&1
bash tools/kstats.sh r05_lbfgs_f32 tools/lbfgs_prof.py float 200 > gpurun_out/final/kstats_lbfgs_f32.txt 2>&1
bash tools/kstats.sh r05_lbfgs_f64 tools/lbfgs_prof.py double 200 > gpurun_out/final/kstats_lbfgs_f64.txt 2>&1
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --vars 100000 --rows 50000 --no-cpu-baseline > gpurun_out/final/bench_1m.json 2>/dev/null
{
for a in "--rows 4000" "--rows 40000" "--rows 20000 --cover-rows 250000" "--rows 30000 --cover-rows 100000" "--rows 10000 --cover-rows 400000" "--rows 25000 --k 18 --iters 100" "--rows 100000 --k 11"; do
  echo "== widebench $a"; timeout 300 python tools/widebench.py $a 2>&1 | grep -E "built|layout|iteration|fwd_plain"
done
echo "== mixedcover 3..16"; timeout 300 python tools/mixedcover.py 2>&1 | grep -E "BDDs|packs|iteration"
echo "== mixedcover 2..40"; timeout 300 python tools/mixedcover.py --kmin 2 --kmax 40 --rows 250000 2>&1 | grep -E "BDDs|packs|iteration"
} > gpurun_out/final/widebench.txt 2>&1
cat gpurun_out/final/gputest.txt; tail -1 gpurun_out/final/bench_default.json | cut -c1-600
