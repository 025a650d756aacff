#!/bin/bash
# Experimental build of the library with extra -D flags (kernel A/B experiments on one GPU box):
#   tools/build_variant.sh NAME -DEXP_EX=1      ->  build/libNAME.so, selected at run time with BDDMMA_LIB=build/libNAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../bdd_amd/csrc"
mkdir -p ../../build/$name
for p in f32 f64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c solver_$p.hip -o ../../build/$name/solver_$p.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib$name.so layout.o capi.o solver_base.o ../../build/$name/solver_f32.o ../../build/$name/solver_f64.o lbfgs.o host/bdd_store.o host/ilp.o host/ilp_capi.o host/instances.o
echo built build/lib$name.so
