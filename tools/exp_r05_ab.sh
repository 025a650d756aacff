#!/bin/bash
# round 5, last session: arena (one allocation, arrays 4 KiB-aligned back to back) against one hipMalloc per array, other sizes and double
for cfg in "double 1000000" "float 400000" "float 2000000" "float 100000" "double 400000" "float 4000000"; do
  set -- $cfg
  for ar in "" "1024,0"; do
    echo "== $1 V=$2 BDDMMA_EXP_ARENA=$ar"; BDDMMA_EXP_ARENA=$ar python tools/placement_probe.py 8 $1 $2 2>&1 | grep solver | cut -c1-62,100-200
  done
done
