export BDDMMA_LIB=build/libstamps.so BDDMMA_STAMPS_FILE=gpurun_out/stamps
for args in "--vars 1000000 --rows 500000" "--vars 4000000 --rows 2000000 --variant 8192" "--vars 4000000 --rows 2000000" "--vars 400000 --rows 200000" "--vars 1000000 --rows 500000 --precision double"; do
  echo "=== $args"; timeout 300 python tools/stamps.py $args 2>&1 | grep -v amdgpu.ids
done
rm -f gpurun_out/stamps.*
