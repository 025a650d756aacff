#!/bin/bash
# round 5, last session: arena skew sweep (extra bytes between consecutive arrays) at 42 M nodes float, one solver per setting
for sk in 0 256 1024 4096 8192 16384 32768 65536 131072 262144 524288 1048576 1114112 2097152 2101248 4194304 3145728 6291456 8388608 12582912; do
  echo "skew $sk: $(BDDMMA_EXP_ARENA=4096,$sk python tools/placement_probe.py 1 float 4000000 2>&1 | grep solver | cut -c1-40,90-200)"
done
