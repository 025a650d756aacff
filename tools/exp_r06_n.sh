#!/bin/bash
# round 6, shape evidence on the final kernels: k-sweep, structured instances, general rows
mkdir -p gpurun_out/r06n
{
for k in 4 10 32 100 300; do timeout 600 python tools/shape_bench.py cover $k 2>&1 | grep -v amdgpu; done
} > gpurun_out/r06n/k_sweep.txt
{
timeout 900 python tools/shape_bench.py assign 1000 2>&1 | grep -v amdgpu
timeout 900 python tools/shape_bench.py assign 1000 0 2>&1 | grep -v amdgpu
timeout 900 python tools/shape_bench.py assign 300 2>&1 | grep -v amdgpu
echo "== mrf grid 300 x 300"; timeout 900 python tools/mrf_grid.py 300 2>&1 | grep -v amdgpu
} > gpurun_out/r06n/structured.txt
{
for a in "--rows 4000" "--rows 40000" "--rows 20000 --cover-rows 250000" "--rows 30000 --cover-rows 100000" "--rows 10000 --cover-rows 400000" "--rows 25000 --k 18 --iters 100" "--rows 100000 --k 11"; do
  echo "== widebench $a"; timeout 300 python tools/widebench.py $a 2>&1 | grep -E "built|layout|iteration|fwd_plain"
done
echo "== mixedcover 3..16"; timeout 300 python tools/mixedcover.py 2>&1 | grep -E "BDDs|packs|iteration"
echo "== mixedcover 2..40"; timeout 300 python tools/mixedcover.py --kmin 2 --kmax 40 --rows 250000 2>&1 | grep -E "BDDs|packs|iteration"
} > gpurun_out/r06n/widebench.txt 2>&1
cat gpurun_out/r06n/k_sweep.txt gpurun_out/r06n/structured.txt; tail -30 gpurun_out/r06n/widebench.txt
