#!/bin/bash
# round 5, last session: run-time non-temporal loads (PackDev::nt_potentials bit 1, footprint > 640 MiB) — shipped against build/libprev.so (the library before)
for lib in "" build/libprev.so "" build/libprev.so; do
  echo "== lib=$lib"
  BDDMMA_LIB=$lib python tools/placement_probe.py 3 float 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 3 double 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 3 float 2000000 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 float 4000000 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 double 2000000 2>&1 | grep solver | cut -c1-44,96-200
done
