#!/bin/bash
# round 6: new C-ABI tests, the bench line with the configs[1] / L-BFGS-roofline / repeat keys, the 105 M-node HBM-only figure
mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06b/gputest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06b/bench_driver_cmd.json 2> gpurun_out/r06b/bench_driver_cmd.err
timeout 900 python tools/hbm_only.py gpurun_out/r06b/hbm_only_105m.json 2 > gpurun_out/r06b/hbm_only.txt 2>&1
cat gpurun_out/r06b/gputest.txt gpurun_out/r06b/hbm_only.txt; tail -c 1500 gpurun_out/r06b/bench_driver_cmd.json
