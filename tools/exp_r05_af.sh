#!/bin/bash
# round 5, last session: cache policy of the loads that are read once (potentials F / T: BDDMMA_LD_POT_AUX; staging tables: BDDMMA_LD_TAB_AUX; 2 = nt,
# 18 = sc1 | nt) — does keeping them out of the caches leave more of the working set in the Infinity Cache?  headline instance, float and double; 21 M float
for lib in "" build/libp2.so build/libt2.so build/libp2t2.so build/libp18.so; do
  echo "== lib=$lib"
  BDDMMA_LIB=$lib python tools/placement_probe.py 3 float 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 double 2>&1 | grep solver | cut -c1-44,96-200
  BDDMMA_LIB=$lib python tools/placement_probe.py 2 float 2000000 2>&1 | grep solver | cut -c1-44,96-200
done
