#!/bin/bash
# Soak of the shipped kernels with several processes sharing the one GPU (VERDICT r2 #1b): P copies of the differential fuzz
# (tests/tools/stress_fuzz.py) and P copies of tests/test_gpu_run_solver.py + tests/test_gpu_small_fused.py (round 6: whole iterations in one launch) in a loop, all at once.
#   tools/soak.sh TAG [P=4] [FUZZ_REPEATS=25] [RUN_SOLVER_LOOPS=8] [ENV=VALUE ...]
# Summary -> gpurun_out/soak_TAG/summary.txt (copy to profiles/).
set -u
TAG=${1:?tag}; P=${2:-4}; REPS=${3:-25}; LOOPS=${4:-8}
shift; shift 2>/dev/null; shift 2>/dev/null; shift 2>/dev/null
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/soak_$TAG
mkdir -p "$OUT"
SEEDS=$(seq 0 39)
t0=$(date +%s)
pids=()
for p in $(seq 1 "$P"); do
    python tests/tools/stress_fuzz.py "$REPS" $SEEDS > "$OUT/fuzz_$p.txt" 2>&1 &
    pids+=($!)
    (for i in $(seq 1 "$LOOPS"); do python -m pytest tests/test_gpu_run_solver.py tests/test_gpu_small_fused.py -q -x -rf -p no:cacheprovider 2>&1 | grep -E "^(FAILED|E  )|passed|failed" | cut -c1-300; done) > "$OUT/runsolver_$p.txt" 2>&1 &
    pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid"; done
t1=$(date +%s)
{
    echo "soak $TAG: $P fuzz processes x $REPS repeats x 40 seeds + $P x $LOOPS loops of tests/test_gpu_run_solver.py + tests/test_gpu_small_fused.py, concurrently on one GPU; $((t1 - t0)) s; env: $*"
    echo "kernel sources: $(cat bdd_amd/csrc/kernels.hpp bdd_amd/csrc/kernels/*.hpp bdd_amd/csrc/solver_impl.hpp bdd_amd/csrc/layout.cpp | sha256sum | cut -c1-16)"
    grep -h "failures in" "$OUT"/fuzz_*.txt
    grep -h "^FAIL" "$OUT"/fuzz_*.txt | head -40
    echo "run_solver loops (one line per pytest run):"
    cat "$OUT"/runsolver_*.txt | sed -E 's/ in [0-9.]+s//' | sort | uniq -c | sort -rn | head -30
} > "$OUT/summary.txt"
cat "$OUT/summary.txt"
