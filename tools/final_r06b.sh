#!/bin/bash
# round 6, final collection, part 2
mkdir -p gpurun_out/final
timeout 900 python tools/hbm_only.py gpurun_out/final/hbm_only_105m.json 3 > gpurun_out/final/hbm_only.txt 2>&1
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_20steps.json 2>/dev/null
python bench.py --vars 100000 --rows 50000 --no-cpu-baseline > gpurun_out/final/bench_1m.json 2>/dev/null
{
for cfg in "100000 400" "400000 400" "1000000 400" "2000000 200" "4000000 100" "10000000 40"; do
  set -- $cfg
  for prec in float double; do
    echo "V=$1 $prec: $(timeout 600 python tools/kbench.py --mt 1 --precision $prec --vars $1 --rows $(($1/2)) --iters $2 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
} > gpurun_out/final/size_sweep.txt 2>&1
timeout 300 python tools/small_rate.py > gpurun_out/final/small_rate.txt 2>&1
cat gpurun_out/final/hbm_only.txt; tail -1 gpurun_out/final/bench_default.json | cut -c1-300; cat gpurun_out/final/size_sweep.txt
