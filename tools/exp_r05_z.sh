#!/bin/bash
# round 5, last session: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) against the runtime's default — iteration rate of the headline
# instance in fresh processes, alternating, with the host's load beside it
cat /proc/loadavg
for i in 1 2 3 4 5 6; do for v in default 0 1; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "kernarg=$v run $i: $(timeout 120 python tools/kbench.py --mt 1 --iters 1000 2>&1 | tail -1)"
done; done
cat /proc/loadavg
