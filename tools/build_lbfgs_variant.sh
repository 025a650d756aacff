#!/bin/bash
# Experimental build of the library with extra -D flags for lbfgs.hip only (the L-BFGS wrapper's passes; the solver objects are the shipped ones):
#   tools/build_lbfgs_variant.sh NAME -DBDDMMA_EXP_X=1      ->  build/libNAME.so, selected at run time with BDDMMA_LIB=build/libNAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../bdd_amd/csrc"
mkdir -p ../../build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c lbfgs.hip -o ../../build/$name/lbfgs.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib$name.so layout.o capi.o solver_base.o solver_f32.o solver_f64.o ../../build/$name/lbfgs.o host/bdd_store.o host/ilp.o host/ilp_capi.o host/instances.o
echo built build/lib$name.so
