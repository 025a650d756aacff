#!/bin/bash
# round 5, last session: arena chunk size at 42 M and 21 M nodes float (several solvers per setting: is the rate the same for every instantiation?)
for V in 4000000 2000000; do for ch in 64 256 1024 8192; do
  echo "== V=$V chunk $ch MiB"; BDDMMA_EXP_ARENA=$ch,0 python tools/placement_probe.py 5 float $V 2>&1 | grep solver | cut -c1-44,96-200
done; done
