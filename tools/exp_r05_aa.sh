#!/bin/bash
# round 5, last session: the solver's arrays out of one allocation (arena) — does the iteration rate still depend on the placement?
for ar in "" "1024,0" "1024,65536" "1024,2097152" "1024,1114112" "1024,4096"; do
  echo "== BDDMMA_EXP_ARENA=$ar"; BDDMMA_EXP_ARENA=$ar python tools/placement_probe.py 10 2>&1 | grep solver | cut -c1-60,100-200
done
