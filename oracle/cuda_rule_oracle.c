/*
 * cuda_rule_oracle.c — second CPU oracle: the reference's GPU solver semantics, with omega and the GPU's non-finite rule.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/ may load it (it lives in oracle/libmma_oracle.so next to
 * mma_oracle.c); the product path (bdd_amd/, the C-ABI library) never links, imports or calls anything in oracle/.
 *
 * Why a second oracle.  mma_oracle.c restates the reference's CPU solver (bdd_parallel_mma_base.cpp), which
 *   (a) ignores the omega argument of forward_mm / backward_mm and hard-codes 0.5 (bdd_parallel_mma_base.cpp:975,1001), and
 *   (b) turns a non-finite min-marginal into +inf arc costs (:844-863),
 * while the path under test follows the reference's GPU solver
 *   LPMP::bdd_cuda_parallel_mma<REAL> (src/bdd_solver/bdd_cuda_parallel_mma.cu:29-42,59-139,142-153,164-346,358-430) over
 *   LPMP::bdd_cuda_base<REAL> (src/bdd_solver/bdd_cuda_base.cu:217-227,439-503,560-713,1243-1251,1396-1445),
 * which scales by the omega it is given and applies NO update to a layer unless both of its min-marginals are finite.  This file
 * restates those CUDA sources (SURVEY.md §8 a') on the CPU, in BDD-local form, so that omega != 0.5 and forced variables /
 * single-variable BDDs / bot-only arcs are checked value for value instead of by properties.
 *
 * Parity pinning: the finite branch at omega = 0.5 is PINNED — tests/test_cuda_rule_oracle.py checks this file against the pinned
 * CPU oracle (mma_oracle.c: reference KATs + bit-exact oracle/_ref traces) on every golden instance, per pass, and against the
 * reference tests' known answers.  The omega-scaled and non-finite branches are restated from the CUDA sources line by line and
 * are UNPINNED BY NECESSITY: the only code of the reference that takes those branches is CUDA, which cannot run in this image
 * (no NVIDIA toolchain or device), and the reference's tests hold no golden vector for them.  What pins them indirectly: both are
 * a single select / multiply on top of the pinned arithmetic, the lower bound must stay below the brute-force optimum, and at
 * omega = 0.5 on all-finite instances the two oracles agree to the last bit in double.
 *
 * Build: make -C oracle   ->  oracle/libmma_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CR_TOPSINK UINT64_MAX
#define CR_BOTSINK (UINT64_MAX - 1)

#define REAL float
#define SUFFIX _f32
#define RINF INFINITY
#include "cuda_rule_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "cuda_rule_oracle_impl.h"
#undef REAL
#undef SUFFIX
