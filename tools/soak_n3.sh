#!/bin/bash
# Soak of the third-generation sweeps under multi-process load: P processes loop over the lane-per-layer parity tests (bit-equality with the
# second generation, oracle comparisons) while the standard soak (tools/soak.sh: differential fuzz + run_solver loops) runs beside them.
#   tools/soak_n3.sh TAG [P=4] [LOOPS=10]
TAG=${1:?tag}; P=${2:-4}; LOOPS=${3:-10}
OUT=gpurun_out/soak_$TAG
mkdir -p "$OUT"
pids=()
for p in $(seq 1 "$P"); do
  (for i in $(seq 1 "$LOOPS"); do python -m pytest tests/test_gpu_parity.py -q -x -rf -p no:cacheprovider -k "lane_per_layer" 2>&1 | grep -E "^(FAILED|E  )|passed|failed" | cut -c1-300; done) > "$OUT/n3_$p.txt" 2>&1 &
  pids+=($!)
done
bash tools/soak.sh $TAG 4 6 4 > /dev/null 2>&1
for pid in "${pids[@]}"; do wait "$pid"; done
{
  cat "$OUT/summary.txt"
  echo "lane-per-layer parity tests, $P processes x $LOOPS loops beside it (one line per pytest run):"
  cat "$OUT"/n3_*.txt | sed -E 's/ in [0-9.]+s//' | sort | uniq -c | sort -rn | head
} > "$OUT/summary_n3.txt"
cat "$OUT/summary_n3.txt"
