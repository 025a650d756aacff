#!/bin/bash
# round-4 option sweeps on the second-generation sweeps: 1.05 M nodes (resident), 10.5 M nodes (streaming)
k() { python tools/kbench.py --mt 1 "$@" 2>/dev/null | tail -2 | tr '\n' ' '; echo; }
echo "## 1.05 M nodes, float"
for o in "" "--byvar 2" "--vars-per-bin 256" "--vars-per-bin 1024" "--wpb 2" "--wpb 4"; do echo "[$o] $(k --vars 100000 --rows 50000 --iters 400 $o)"; done
echo "## 10.5 M nodes, float"
for o in "" "--wpb 2" "--wpb 8" "--vars-per-bin 2048" "--vars-per-bin 3072" "--vars-per-bin 6144" "--pack-width 64" "--pack-width 256" "--stage-cap 320"; do echo "[$o] $(k --iters 300 $o)"; done
echo "## 10.5 M nodes, double"
for o in "" "--wpb 2" "--vars-per-bin 1024" "--vars-per-bin 4096" "--pack-width 64"; do echo "[$o] $(k --precision double --iters 200 $o)"; done
