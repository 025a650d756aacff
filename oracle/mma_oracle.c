/*
 * mma_oracle.c — CPU oracle for the parallel deferred min-marginal-averaging path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (bdd_amd/, the C-ABI
 * library) never links, imports or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the reference's CPU solver
 *   LPMP::bdd_parallel_mma_base<bdd_branch_instruction<REAL,uint16_t>>
 *   (src/bdd_solver/bdd_parallel_mma_base.cpp, include/bdd_solver/bdd_branch_instruction.h)
 * — the solver the reference selects with "relaxation solver": "parallel mma" and the one
 * BASELINE.json names as the parity target for the HIP path.  Function-level citations are
 * in mma_oracle_impl.h.
 *
 * Parity pinning: PINNED.  The restatement is checked (tests/test_oracle_kat.py) against
 *  (a) the known-answer values of the reference's own tests for this path
 *      (test/test_bdd_cuda_base.cpp:49-115, test/test_bdd_cuda_min_marginals.cpp:17-36,
 *       test/test_bdd_cuda_parallel_mma.cu:197-247, test/test_bdd_bipartite_matching_problem.cpp:8-59,
 *       test/test_loose_covering_problem.cpp:8-88, test/test_bdd_parallel_mma.cpp:19-125);
 *  (b) golden trajectories under tests/golden/ produced by oracle/_ref (reference sources
 *      compiled where they lie: bdd_collection + bdd_manager + bdd_branch_instruction.h;
 *      recipe oracle/Makefile, generator oracle/make_golden.py);
 *  (c) the lower-bound trajectories recorded in BASELINE.md §2 from the unmodified reference.
 * The reference translation unit bdd_parallel_mma_base.cpp itself is UNBUILDABLE in this
 * image (it includes <Eigen/SparseCore>, which is absent; no stand-in headers are written),
 * see DESIGN.md.
 *
 * Build: make -C oracle   ->  oracle/libmma_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TOPSINK UINT64_MAX
#define BOTSINK (UINT64_MAX - 1)
/* bdd_branch_instruction.h:32-33 (uint32 here, uint16 in the reference) */
#define TERM0 0xFFFFFFFFu
#define TERM1 0xFFFFFFFEu

#define REAL float
#define SUFFIX _f32
#define RINF INFINITY
#define RMIN(a, b) fminf_like_std((a), (b))
static inline float fminf_like_std(float a, float b) { return b < a ? b : a; } /* std::min semantics */
#ifdef _OPENMP
#include <omp.h>
#endif
/* The loops over BDDs of the mm passes use schedule(runtime): dynamic, 64 unless the environment says otherwise (the reference:
 * schedule(dynamic), bdd_parallel_mma_base.cpp:973,999).  MMA_ORACLE_SCHEDULE=static makes them static, 512 — the schedule the arrays
 * were first touched with (oracle_create), for the timing leg of bench.py on multi-socket hosts. */
static void oracle_default_schedule(void)
{
#ifdef _OPENMP
    static int done = 0;
    if (done) return;
    done = 1;
    const char* e = getenv("MMA_ORACLE_SCHEDULE");
    if (e && e[0] == 's') omp_set_schedule(omp_sched_static, 512);
    else omp_set_schedule(omp_sched_dynamic, 64);
#endif
}
#include "mma_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RMIN

#define REAL double
#define SUFFIX _f64
#define RMIN(a, b) fmin_like_std((a), (b))
static inline double fmin_like_std(double a, double b) { return b < a ? b : a; }
#include "mma_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RMIN
