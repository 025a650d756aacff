#!/bin/bash
# usage: bash tools/quick.sh "<bench args>" ...   -> one line per config
for a in "$@"; do
  r=$(timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['roofline']['avg_launch_ms']; print(round(d['value']), 'frac', round(d['roofline']['frac'],3), 'us', round(m['forward_mm']*1e3), round(m['backward_mm']*1e3), round(m['finish_delta']*1e3))")
  echo "[$a] -> $r"
done
