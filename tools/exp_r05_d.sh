# hop-loop time against the waves a SIMD holds (second-generation sweeps capped at 5 (shipped) / 2 / 1 waves per SIMD by amdgpu_waves_per_eu)
for lib in stamps stamps2 stamps1; do
  for args in "--vars 1000000 --rows 500000" "--vars 4000000 --rows 2000000 --variant 8192"; do
    echo "=== $lib $args"
    BDDMMA_LIB=build/lib$lib.so BDDMMA_STAMPS_FILE=gpurun_out/stamps timeout 300 python tools/stamps.py $args 2>&1 | grep -E "solve:|per wave|waves resident" | grep -v exchange
  done
done
rm -f gpurun_out/stamps.*
