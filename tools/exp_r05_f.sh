# per-wave timeline of the third-generation sweeps against the second / first generation (stamped build)
for args in "--vars 1000000 --rows 500000" "--vars 1000000 --rows 500000 --variant 262144" "--vars 4000000 --rows 2000000" "--vars 4000000 --rows 2000000 --variant 270336"; do
  echo "=== $args"
  BDDMMA_LIB=build/libstamps.so BDDMMA_STAMPS_FILE=gpurun_out/stamps timeout 300 python tools/stamps.py $args 2>&1 | grep -E "solve:|per wave|waves resident"
done
rm -f gpurun_out/stamps.*
