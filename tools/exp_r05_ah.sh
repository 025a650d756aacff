#!/bin/bash
# round 5, last session: third-generation sweeps instantiated with non-temporal loads (template parameter NT; double, footprint > 640 MiB) — shipped against
# build/libprev.so (the library before), two alternating rounds
for lib in "" build/libprev.so "" build/libprev.so; do
  echo "== lib=$lib"
  BDDMMA_LIB=$lib python tools/placement_probe.py 4 double 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 float 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 double 1400000 2>&1 | grep solver | cut -c1-44,96-200
done
