# what a hop costs without its global streams (timing experiments, wrong results): forward solve sweep, second generation, stamped builds
# (tools/build_variant.sh noload -DBDDMMA_STAMPS -DBDDMMA_EXP_NO_HOP_LOADS; nostore ... -DBDDMMA_EXP_NO_HOP_STORES; noboth: both)
for lib in stamps noload nostore noboth; do
  for args in "--vars 400000 --rows 200000" "--vars 1000000 --rows 500000" "--vars 4000000 --rows 2000000 --variant 8192"; do
    echo "=== $lib $args"
    BDDMMA_LIB=build/lib$lib.so BDDMMA_STAMPS_FILE=gpurun_out/stamps timeout 300 python tools/stamps.py $args 2>&1 | grep -B0 -A12 "^fwd_solve:" | grep -E "solve:|per wave"
  done
done
rm -f gpurun_out/stamps.*
