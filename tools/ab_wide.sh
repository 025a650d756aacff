#!/bin/bash
# general linear rows: first generation (variant_flags 0x1000) vs second generation narrow sweeps, same box.  usage: ab_wide.sh
run() { python tools/widebench.py "$@" 2>/dev/null | grep -E "iteration|layout:" | tr '\n' ' '; echo; }
for v in 4096 0 512 0; do echo "1M knapsack variant=$v: $(run --variant $v)"; done
for v in 4096 0; do echo "10M knapsack (40k rows) variant=$v: $(run --rows 40000 --variant $v)"; done
for v in 4096 0; do echo "20k knapsack + 250k cover variant=$v: $(run --rows 20000 --cover-rows 250000 --variant $v)"; done
for v in 4096 0; do echo "25k rows of 18 (wide only) variant=$v: $(run --rows 25000 --k 18 --variant $v)"; done
