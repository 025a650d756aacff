#!/bin/bash
# round 6: double third-generation sweeps with a look-ahead of two hops: the GPU tests, the default bench line twice
mkdir -p gpurun_out/r06p
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06p/tests.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-lbfgs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value']), 'f64', round(d['value_f64']), 'repeat', [round(x) for x in d['repeat_samples']], 'configs', {k:(round(v['value']) if isinstance(v,dict) else '') for k,v in d['configs'].items()})"; done > gpurun_out/r06p/bench.txt
cat gpurun_out/r06p/tests.txt gpurun_out/r06p/bench.txt
