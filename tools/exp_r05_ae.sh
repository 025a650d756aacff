#!/bin/bash
# round 5, last session: placement of the double 10.5 M-node instance (0.72 GB) — one hipMalloc per array / arena with several spacings (build/libexp.so = EXPERIMENTAL build)
export BDDMMA_LIB=build/libexp.so
for ar in "0,0" "2048,0" "2048,4096" "2048,65536" "2048,1048576" "2048,1114112" "2048,2097152" "2048,8388608" "256,0" "64,0"; do
  echo "== double 10.5 M nodes BDDMMA_EXP_ARENA=$ar"; BDDMMA_EXP_ARENA=$ar python tools/placement_probe.py 4 double 2>&1 | grep solver | cut -c1-44,96-200
done
