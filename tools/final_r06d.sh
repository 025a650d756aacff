#!/bin/bash
# round 6, final collection, last part: the bench lines on the committed profiles (traffic.json and hbm_only stamped with the final sources' hash)
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_20steps.json 2>/dev/null
python bench.py --vars 100000 --rows 50000 --no-cpu-baseline > gpurun_out/final/bench_1m.json 2>/dev/null
tail -1 gpurun_out/final/bench_default.json | cut -c1-200
