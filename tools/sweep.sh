#!/bin/bash
# config sweep on the GPU box: prints it/s and per-kernel ms
for prec in float double; do
for pw in 64 128; do
for cap in 320 384 640; do
for vb in 2048 4096 8192; do
  [ $cap -lt $pw ] && continue
  r=$(timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --precision $prec --pack-width $pw --stage-cap $cap --vars-per-bin $vb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['roofline']['avg_launch_ms']; print(round(d['value']), round(m['forward_mm']*1e3), round(m['backward_mm']*1e3), round(m['finish_delta']*1e3))")
  echo "$prec pw=$pw cap=$cap vb=$vb -> $r"
done; done; done; done
