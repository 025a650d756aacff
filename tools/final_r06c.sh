#!/bin/bash
# round 6, final collection, part 3 (after the last source change): the HBM-only figure and the bench lines again, stamped with the final sources' hash
mkdir -p gpurun_out/final
timeout 900 python tools/hbm_only.py gpurun_out/final/hbm_only_105m.json 3 > gpurun_out/final/hbm_only.txt 2>&1
cp gpurun_out/final/hbm_only_105m.json profiles/r06_hbm_only_105m.json
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_20steps.json 2>/dev/null
python bench.py --vars 100000 --rows 50000 --no-cpu-baseline > gpurun_out/final/bench_1m.json 2>/dev/null
cat gpurun_out/final/hbm_only.txt; tail -1 gpurun_out/final/bench_default.json | cut -c1-200
