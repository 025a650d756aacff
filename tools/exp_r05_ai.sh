#!/bin/bash
# round 5, last session: the non-temporal STORES of the third-generation sweeps as part of the template parameter NT instead of a uniform branch per store —
# shipped against build/libprev.so, two alternating rounds, double
for lib in "" build/libprev.so "" build/libprev.so; do
  echo "== lib=$lib"
  BDDMMA_LIB=$lib python tools/placement_probe.py 4 double 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 double 400000 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 double 1400000 2>&1 | grep solver | cut -c1-44,96-200
done
