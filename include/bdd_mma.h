/*
 * bdd_mma.h — C-ABI of the MI355X-native parallel deferred min-marginal-averaging
 * (MMA) solver over BDDs.
 *
 * This is the drop-in boundary for the reference's GPU relaxation solver
 * `LPMP::bdd_cuda_parallel_mma<REAL>` and its base `LPMP::bdd_cuda_base<REAL>`
 * (reference: include/bdd_solver/bdd_cuda_parallel_mma.h:7-52,
 * include/bdd_solver/bdd_cuda_base.h:58-229).  The reference has no FFI for this
 * path; the boundary is a compile-time "solver concept" selected by
 * `"relaxation solver": "cuda parallel mma"` (src/bdd_solver/bdd_solver.cpp:164-176).
 * Every entry point below names the reference member function it replaces.
 *
 * Conventions
 *  - Nothing but POD crosses the ABI.  All device memory is owned by the handle.
 *  - Every call returns BDDMMA_OK (0) or a negative error code; the message is
 *    available from bddmma_last_error().  No C++ exception crosses the ABI.
 *  - `precision` is BDDMMA_F32 or BDDMMA_F64; "REAL" below means that type.
 *    Buffers declared `void*` hold REAL elements of the handle's precision.
 *  - `on_device` != 0 means the buffer is a device pointer on the handle's GPU
 *    (the reference's thrust::device_vector overloads); 0 means host memory.
 *  - One handle per problem; handles are independent (own stream, own buffers)
 *    so one host thread/process per GPU is safe (reference: single stream,
 *    device 0 hard-coded, include/cuda_utils.h:111-114).
 *  - Layers: one layer per (BDD, variable) pair = one dual variable.  Terminal
 *    layers carry no information and are not exposed: nr_layers() equals the
 *    reference CPU solver's nr_layers() (bdd_parallel_mma_base.cpp:1398-1402),
 *    i.e. the reference GPU nr_layers() minus nr_bdds().  Per-layer vectors are
 *    in the solver's internal layer order; bddmma_layer_variables /
 *    bddmma_layer_bdds give the (variable, BDD) of every entry.
 */
#ifndef BDD_MMA_H
#define BDD_MMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDDMMA_OK 0
#define BDDMMA_ERR_INVALID_ARGUMENT (-1)
#define BDDMMA_ERR_INVALID_BDD (-2)   /* not a QBDD / malformed collection */
#define BDDMMA_ERR_DEVICE (-3)        /* HIP runtime error */
#define BDDMMA_ERR_STATE (-4)         /* call not valid in current solver state */
#define BDDMMA_ERR_UNSUPPORTED (-5)
#define BDDMMA_ERR_IO (-6)

#define BDDMMA_F32 0
#define BDDMMA_F64 1

/* Bit-identical to BDD::bdd_instruction {size_t lo, hi, index}
 * (reference: include/bdd_collection/bdd_collection.h:14-36).  `lo`/`hi` are
 * ABSOLUTE indices into the instruction array; terminals have
 * index == BDDMMA_TOPSINK / BDDMMA_BOTSINK. */
typedef struct bddmma_instruction {
    uint64_t lo;
    uint64_t hi;
    uint64_t index;
} bddmma_instruction;

#define BDDMMA_TOPSINK UINT64_MAX
#define BDDMMA_BOTSINK (UINT64_MAX - 1)

typedef struct bddmma_solver bddmma_solver;

/* Tunables of the device layout (0 = default). */
typedef struct bddmma_options {
    uint32_t pack_width;       /* max #nodes of one hop inside a wave-sized BDD pack: 64, 128 (default) or 256 */
    uint32_t wide_pack_width;  /* max #nodes of one hop inside a workgroup-sized pack whose frontier lives in LDS (default 2048; <= 2048 for
                                  F64, <= 4096 for F32); BDDs with wider layers use the same kernels with the frontier in global memory */
    uint32_t deterministic;    /* 1: the per-variable delta sums in a fixed order — (variable, bdd), the CPU solver's — instead of LDS atomics:
                                  bit-reproducible; one launch like the default exchange, ~10 % slower than it at 10 M nodes */
    uint32_t vars_per_bin;     /* variables per exchange bin, <= 9728 (16 B of LDS accumulators each); default: ~V/256 rounded, 1024..9728 */
    uint32_t stage_cap;        /* max layers of one stage group of a narrow pack (default 640) */
    uint32_t waves_per_block;  /* narrow packs swept by one workgroup with cooperative staging: 1, 2, 4 or 8 (default 4) */
    uint32_t keep_bdd_order;   /* 1: keep the input order of the BDDs when forming packs (default: BDDs of equal shape are grouped, diamond-shaped
                                  BDDs of small shape classes — general linear rows — follow widest first; 2: grouped by shape only) */
    uint32_t resident_sweeps;  /* narrow packs copied to LDS up front and swept from there: 0 automatic (when every workgroup of a sweep can
                                  be in flight at once: small / medium instances), 1 off, 2 on whenever the packs fit */
    uint32_t exchange_by_variable;  /* 2: entry arrays ordered by (variable, bdd) — the per-variable delta reduction becomes one thread
                                  per variable over a contiguous run: deterministic, no LDS accumulators, a shorter exchange launch; but
                                  the sweeps' accesses to the entry arrays lose their locality (slower overall on every instance
                                  measured).  Default (0 / 1): binned order */
    uint32_t variant_flags;    /* TEST HOOKS, default 0.  Every bit forces a code path of the shipped library that the automatic rules choose on
                                  other shapes or sizes, so that the differential tests reach it on small instances; none changes results
                                  beyond floating-point summation order.  (The A/B-only switches of rounds 2-4 — bits 2-8 and 16 — are gone:
                                  the paths they selected were deleted or are selected by rule alone; those bits are ignored.)
                                  bit 0 / bit 1: narrow and wide backward / forward sweeps as two launches (rule: one launch, k_*_mixed, except
                                                 where the wide share is large)
                                  bit 9 / bit 10: make_dual_feasible of the L-BFGS direction through the staging tables / by gathers
                                                  (rule: staged from 500 000 layers on)
                                  bit 11: first-generation resident sweeps (rule: packs of 128 / 256 slots or layers wider than two nodes)
                                  bit 12: first-generation streaming solve sweeps (rule: packs that share no records; float above 16 M slots)
                                  bit 13: per-lane records for the streaming solve sweeps also where packs do not share them
                                  bit 14: staging transfers with 64-bit addresses (rule: arrays of 4 GiB and more)
                                  bit 15: the L-BFGS direction as its own pass (rule: wide packs present, or the projection by gathers)
                                  bit 17: `deterministic` exchanges by per-variable gathers, two launches (rule: bins that do not fit the
                                          one-launch schedule k_exchange_seg; same sums, same order: bit-equal results)
                                  bit 18: no third-generation streaming solve sweeps (a lane per layer; rule: packs of 128 slots with layers of
                                          <= 2 nodes and <= 64 layers per hop that share their records, up to 16 M slots): the second / first
                                          generation instead (bit 13 lifts the sharing and size conditions of the third generation too)
                                  bit 19: four launches per iteration also for instances that fit one workgroup (rule: whole iterations in one
                                          launch, k_iterate_small, for <= 16 narrow packs of 64 slots with layers of <= 2 nodes)
                                  bit 20: the streaming solve sweeps' instantiations that load potentials and staging tables non-temporally
                                          whatever the footprint (rule: arrays beyond 640 MiB; first generation, and second in double,
                                          packs of 128 slots, 4 / 8 per workgroup; the third generation's own: double beyond 640 MiB) */
    uint32_t pack_fill;        /* slots of a narrow pack's hop that further BDDs are packed into, in [2, pack_width] (default 0 = pack_width).
                                  Smaller values give more, emptier packs (more wavefronts for the same nodes); measured slower on every
                                  instance (NOTES.md section 6: the sweeps are bound by instructions issued, not by latency), kept for experiments */
    uint32_t pack_stagger;     /* narrow packs whose BDDs start at different hops ("staggered"): a BDD that no longer fits next to the ones
                                  of the open pack hop by hop is tried a few hops further down, where those have become narrow again — BDDs
                                  of general linear rows are narrow at both ends and wide in the middle, and side by side from hop 0 they
                                  fill ~30 % of a pack's lanes.  Value = most hops a pack may have; 0: automatic (on when the instance is
                                  large enough to keep ~2 700 packs of three BDD lengths), 1: off.  Applies to the wide packs too: a chained
                                  wide pack is one BDD wide (automatic while >= ~500 packs and enough wavefronts for the chip remain). */
} bddmma_options;

/* ---- construction ------------------------------------------------------- */

/* Replaces bdd_cuda_parallel_mma<REAL>(const BDD::bdd_collection&, const std::vector<double>& costs_hi)
 * (src/bdd_solver/bdd_cuda_parallel_mma.cu:7-17, bdd_cuda_base.cu:31-53).
 * `bdd_delims` has n_bdds+1 entries: BDD b occupies instr[bdd_delims[b] .. bdd_delims[b+1]),
 * nodes grouped by variable in BDD order, the two terminals last (either order).
 * Every BDD must be a reordered QBDD (bdd_cuda_base.cu:98-99).
 * costs_hi may be NULL (n_costs = 0): all costs zero. */
int bddmma_create(bddmma_solver** out, int precision, int device,
                  const bddmma_instruction* instr, const uint64_t* bdd_delims, uint64_t n_bdds,
                  const double* costs_hi, uint64_t n_costs, const bddmma_options* opts);
void bddmma_destroy(bddmma_solver* s);

/* Number of HIP devices visible to the process (0 when there is none / no driver).  One host thread per device, each with its own
 * handles, is the multi-GPU model: independent instances, no collective (reference: device 0 hard-coded, include/cuda_utils.h:111-114). */
int bddmma_device_count(void);
/* What the layout rules and the input stage ask the chip (hipDeviceProp of `device`): compute units, LDS bytes per compute unit, and
 * multiProcessorCount * maxThreadsPerMultiProcessor — the figure the reference's getMaximumOccupancy() derives its split length from
 * (src/bdd_conversion/bdd_preprocessor.cpp:21-30: cudaGetDeviceProperties, / 10 there).  Any out pointer may be NULL.
 * BDDMMA_ERR_DEVICE when there is no such device. */
int bddmma_device_chip(int device, uint32_t* n_cus, uint32_t* lds_bytes_per_cu, uint64_t* max_resident_threads);
/* Host threads one layout build (bddmma_create) may use: 0 = automatic (environment BDDMMA_THREADS, else min(cores, 32)).  A process that
 * builds one instance per GPU at the same time gives every build cores / #GPUs (the reference builds its layout on the device,
 * bdd_cuda_base.cu:146-391; here it is host work). */
int bddmma_set_layout_threads(int n);
/* The same limit for the builds started by the calling host thread only (0 = follow the process-wide setting).  One-solver-per-GPU host
 * threads (include/cuda_utils.h:111-114 of the reference knows one device; the batch farm here has a thread per device) set it for
 * themselves, so concurrent batches and the application's own process-wide value do not overwrite each other. */
int bddmma_set_thread_layout_threads(int n);
/* Error text of the last failed call on `s`; with s == NULL the last failed bddmma_create. */
const char* bddmma_last_error(const bddmma_solver* s);

/* ---- sizes (bdd_cuda_base.h:98-116) -------------------------------------- */
uint64_t bddmma_nr_variables(const bddmma_solver* s);
uint64_t bddmma_nr_bdds(const bddmma_solver* s);
uint64_t bddmma_nr_layers(const bddmma_solver* s);      /* non-terminal layers = #dual variables */
uint64_t bddmma_nr_bdd_nodes(const bddmma_solver* s);   /* incl. 2 terminals per BDD, as the reference counts */
uint64_t bddmma_nr_hops(const bddmma_solver* s);        /* length of the longest BDD */
uint64_t bddmma_nr_packs(const bddmma_solver* s);
/* Which kernels run the solve sweeps (forward_mm / backward_mm) of the narrow packs — chosen by rule at creation from the instance's shape and
 * size; diagnostics for benchmarks and for the tests that must know which path they compare with the oracle. */
enum {
    BDDMMA_SWEEPS_NONE = 0,        /* no narrow packs (wide / huge packs only) */
    BDDMMA_SWEEPS_MIXED = 1,       /* narrow and wide packs in one launch (k_fwd_mixed / k_bwd_mixed) */
    BDDMMA_SWEEPS_STREAMING1 = 2,  /* streaming, node words (k_fwd_narrow / k_bwd_narrow) */
    BDDMMA_SWEEPS_STREAMING2 = 3,  /* streaming, a 16-byte record per lane and hop (k_*_narrow2) */
    BDDMMA_SWEEPS_STREAMING3 = 4,  /* streaming, a lane per layer (k_*_narrow3) */
    BDDMMA_SWEEPS_RESIDENT1 = 5,   /* pack resident in LDS (k_*_res) */
    BDDMMA_SWEEPS_RESIDENT2 = 6    /* pack resident in LDS, records (k_*_res2) */
};
int bddmma_solve_sweep_kind(const bddmma_solver* s);
/* 1 when the instance fits one workgroup and bddmma_iterations / bddmma_run_solver run whole iterations inside one launch (sweeps, exchanges
 * and run_solver's tests with workgroup barriers in between, csrc/kernels/small.hpp; variant_flags bit 19 turns it off), else 0. */
int bddmma_fused_small(const bddmma_solver* s);
/* 1 when the narrow packs' solve sweeps run in the instantiation that loads what a sweep reads once (potentials, staging tables) non-temporally. */
int bddmma_nontemporal_loads(const bddmma_solver* s);
int bddmma_precision(const bddmma_solver* s);
int bddmma_device(const bddmma_solver* s);
/* nr_bdds(var): int32[nr_variables] (get_num_bdds_per_var, bdd_cuda_base.h:166) */
int bddmma_num_bdds_per_var(const bddmma_solver* s, int32_t* out);
/* (variable, bdd) of every layer in internal layer order (get_primal_variable_index / get_bdd_index) */
int bddmma_layer_variables(const bddmma_solver* s, int32_t* out);
int bddmma_layer_bdds(const bddmma_solver* s, int32_t* out);
/* nodes / layers per hop (get_cum_nr_bdd_nodes_per_hop_dist etc., non-cumulative, nr_hops entries) */
int bddmma_nodes_per_hop(const bddmma_solver* s, uint64_t* out);
int bddmma_layers_per_hop(const bddmma_solver* s, uint64_t* out);

/* ---- costs -------------------------------------------------------------- */
/* update_costs(cost_delta_0, cost_delta_1) (bdd_cuda_base.cu:476-558): cost[layer] += c[var]/nr_bdds(var).
 * n_lo / n_hi may be 0 (that side untouched) or <= nr_variables; layers of variables >= n are
 * SET to 0 as in the reference (bdd_cuda_base.cu:465-469).  elem_precision: type of lo/hi buffers. */
int bddmma_update_costs(bddmma_solver* s, const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi,
                        int elem_precision, int on_device);
/* set_cost(c, var) (bdd_cuda_base.cu:441-455): hi cost of all layers of `var` += c/nr_bdds(var). */
int bddmma_set_cost(bddmma_solver* s, double c, uint64_t var);
/* get_solver_costs / set_solver_costs (bdd_cuda_base.cu:1308-1344): REAL[nr_layers] each. */
int bddmma_get_solver_costs(const bddmma_solver* s, void* lo, void* hi, void* deferred_mm_diff, int on_device);
int bddmma_set_solver_costs(bddmma_solver* s, const void* lo, const void* hi, const void* deferred_mm_diff, int on_device);
/* compute_primal_objective_vec (bdd_cuda_base.cu:1352-1362): out[var] = sum over layers of var (hi - lo). */
int bddmma_primal_objective_vec(bddmma_solver* s, void* out, int on_device);

/* ---- plain sweeps ------------------------------------------------------- */
int bddmma_forward_run(bddmma_solver* s);   /* bdd_cuda_base.cu:588-612 */
int bddmma_backward_run(bddmma_solver* s);  /* bdd_cuda_base.cu:669-713 (without path costs) */
/* lower_bound() (bdd_cuda_base.cu:1243-1251): sum of root costs-from-terminal, accumulated in double. */
int bddmma_lower_bound(bddmma_solver* s, double* lb);
/* lower_bound_per_bdd (bdd_cuda_base.cu:1253-1259): REAL[nr_bdds], indexed by BDD number. */
int bddmma_lower_bound_per_bdd(bddmma_solver* s, void* out, int on_device);

/* ---- parallel MMA (bdd_cuda_parallel_mma.cu) ------------------------------ */
/* iteration(omega) (:142-153): forward_mm, normalize_delta, backward_mm, normalize_delta on the
 * solver's own deferred delta. Asynchronous: returns once queued on the handle's stream. */
int bddmma_iteration(bddmma_solver* s, double omega);
/* n iterations back to back without host synchronisation. */
int bddmma_iterations(bddmma_solver* s, double omega, uint64_t n);
/* forward_mm(omega, delta_lo_hi) / backward_mm (:207-257, :301-346): delta is REAL[2*nr_variables],
 * interleaved {lo,hi}; read as the values to add, overwritten with the un-normalised sums. */
int bddmma_forward_mm(bddmma_solver* s, double omega, void* delta_lo_hi, int on_device);
int bddmma_backward_mm(bddmma_solver* s, double omega, void* delta_lo_hi, int on_device);
/* normalize_delta (:410-430): delta[i] /= nr_bdds(i/2). */
int bddmma_normalize_delta(const bddmma_solver* s, void* delta_lo_hi, int on_device);
/* distribute_delta() (bdd_cuda_base.cu:1396-1436). */
int bddmma_distribute_delta(bddmma_solver* s);
/* the solver's deferred delta_lo_hi_ (REAL[2*nr_variables]) */
int bddmma_get_delta(const bddmma_solver* s, void* out, int on_device);
int bddmma_set_delta(bddmma_solver* s, const void* in, int on_device);

/* ---- min-marginals and per-BDD solutions -------------------------------- */
/* min_marginals_cuda(get_sorted) (bdd_cuda_base.cu:716-749): var int32[nr_layers], mm0/mm1 REAL[nr_layers].
 * sorted != 0: ordered by (variable, bdd) as primal_variable_sorting_order_ (bdd_cuda_base.cu:379-391). */
int bddmma_min_marginals(bddmma_solver* s, int sorted, int32_t* var, void* mm0, void* mm1, int on_device);
/* mm1 - mm0 per layer, internal layer order, REAL[nr_layers] (compute_and_set_min_marginal_diff of the reference's Python module,
 * src/bdd_solver/bdd_cuda_parallel_mma_py.cu:56-72: min_marginals_cuda(false) followed by thrust::minus into the caller's buffer). */
int bddmma_min_marginal_diff(bddmma_solver* s, void* out, int on_device);
/* bdds_solution_vec() (bdd_cuda_base.cu:1139-1202): char[nr_layers] argmin path per BDD, internal
 * layer order (sorted = 0) or (variable,bdd) order (sorted = 1, as bdds_solution(), :1204-1233). */
int bddmma_bdds_solution(bddmma_solver* s, int sorted, char* sol, int on_device);

/* ---- L-BFGS support (lbfgs.h:22-27) --------------------------------------- */
/* net_solver_costs() (bdd_cuda_parallel_mma.cu:432-463): hi - lo + deferred mm diff, REAL[nr_layers]. */
int bddmma_net_solver_costs(const bddmma_solver* s, void* out, int on_device);
/* make_dual_feasible(g) (bdd_cuda_base.cu:1261-1303): g[layer] -= mean over layers of the same variable. */
int bddmma_make_dual_feasible(const bddmma_solver* s, void* g, int on_device);
/* gradient_step(g, step) (bdd_cuda_parallel_mma.h:62-77): hi += step * g. */
int bddmma_gradient_step(bddmma_solver* s, const void* g, double step_size, int on_device);

/* L-BFGS outer solver wrapping a handle (lbfgs.h:35-111, lbfgs_impl.h). */
typedef struct bddmma_lbfgs bddmma_lbfgs;
typedef struct bddmma_lbfgs_params {
    int32_t history_size;                   /* "history size", default 5 */
    double init_step_size;                  /* "initial step size", default 1e-6 */
    double req_rel_lb_increase;             /* "required relative lb increase", default 1e-6 */
    double step_size_decrease_factor;       /* default 0.8 */
    double step_size_increase_factor;       /* default 1.1 */
} bddmma_lbfgs_params;
int bddmma_lbfgs_create(bddmma_lbfgs** out, bddmma_solver* s, const bddmma_lbfgs_params* p);
void bddmma_lbfgs_destroy(bddmma_lbfgs* l);
int bddmma_lbfgs_iteration(bddmma_lbfgs* l);       /* lbfgs::iteration(), lbfgs_impl.h:137-157 */
int bddmma_lbfgs_update_costs(bddmma_lbfgs* l, const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi,
                              int elem_precision, int on_device);  /* lbfgs_impl.h:353-364: also drops history */
int bddmma_lbfgs_flush(bddmma_lbfgs* l);           /* flush_lbfgs_states(), lbfgs_impl.h:318-326 */
/* What the state machine did in the last bddmma_lbfgs_iteration (the reference only logs these; exposed so that the
 * parity tests can compare the decision sequence with the CPU restatement oracle/lbfgs_oracle.py). */
typedef struct bddmma_lbfgs_state {
    double step_size;                 /* lbfgs::step_size after the last iteration */
    double last_applied_step;         /* step left applied by search_step_size_and_apply (0: none / rolled back) */
    uint64_t mma_iterations;          /* lbfgs::mma_iterations */
    uint64_t lbfgs_iterations;        /* lbfgs::lbfgs_iterations */
    int32_t history_entries;          /* history.size() */
    int32_t num_unsuccessful_updates; /* num_unsuccessful_lbfgs_updates_ */
    int32_t last_kind;                /* choose_solver() of the last iteration: 0 mma, 1 lbfgs */
    int32_t last_trials;              /* gradient steps taken by the last step-size search */
} bddmma_lbfgs_state;
int bddmma_lbfgs_get_state(const bddmma_lbfgs* l, bddmma_lbfgs_state* out);

/* ---- run_solver (include/run_solver_util.h:10-77) -------------------------
 * Same criteria, same order, same result as the reference's loop (iteration(); lower_bound(); time limit, minimum improvement,
 * improvement slope, infeasibility).  For the plain solver (lbfgs_or_null == NULL) the three tests on the bound run on the device,
 * in the launch that ends each iteration; the host keeps a few iterations queued and never synchronises inside the loop, and the
 * launches queued behind the stopping iteration return without doing anything — the solver is left in exactly the state after the
 * iteration that met the criterion.  The wall-clock limit is tested on the host after every iteration it sees complete. */
typedef struct bddmma_run_result {
    uint64_t iterations;
    double lb_initial;
    double lb_final;
    double seconds;
    int32_t stop_reason; /* 0 max iter, 1 time limit, 2 min improvement, 3 improvement slope, 4 infeasible */
} bddmma_run_result;
int bddmma_run_solver(bddmma_solver* s, bddmma_lbfgs* lbfgs_or_null, uint64_t max_iter, double tolerance,
                      double improvement_slope, double time_limit, int verbose, bddmma_run_result* res);
/* The same loop literally as the reference writes it — iteration(); lower_bound() with a host round trip; the tests on the host — for the
 * plain solver too (with an L-BFGS wrapper both entry points run this loop).  Same iteration count, same state, same bound. */
int bddmma_run_solver_host_loop(bddmma_solver* s, bddmma_lbfgs* lbfgs_or_null, uint64_t max_iter, double tolerance,
                                double improvement_slope, double time_limit, int verbose, bddmma_run_result* res);

/* ---- primal rounding (src/bdd_solver/incremental_mm_agreement_rounding_cuda.cu:333-372) ------------------
 * incremental_mm_agreement_rounding_cuda(s, init_delta, delta_growth_rate, num_itr_lb, verbose, num_rounds):
 * perturbs the costs towards the sign of the min-marginal differences until they agree in every BDD.
 * sol: char[nr_variables]; *found = 1 if a solution was reconstructed.  The solver's costs stay perturbed
 * afterwards, as in the reference (bdd_solver.cpp:368 "TODO: reset solver state"). */
int bddmma_incremental_mm_agreement_rounding(bddmma_solver* s, bddmma_lbfgs* lbfgs_or_null, double init_delta,
                                             double delta_growth_rate, uint64_t num_itr_lb, uint64_t num_rounds,
                                             uint32_t seed, int verbose, char* sol, int* found);

/* One round of perturb_primal_costs (incremental_mm_agreement_rounding_cuda.cu:262-331): distribute_delta, min-marginals,
 * per-variable sign agreement (:29-65,76-108), sums (:110-134) and the perturbation {delta,0} / {0,delta} / random
 * (:136-205), then update_costs — through the L-BFGS wrapper (which drops its history, lbfgs_impl.h:343-364) when
 * `lbfgs_or_null` is given.  counts = #one, #zero, #equal, #inconsistent.  When all variables are `one` or `zero` the
 * solution is written to sol (char[nr_variables]) and the costs are left alone.  cost_delta_0 / cost_delta_1 (REAL[nr_variables],
 * host, may be NULL) receive the perturbation that was applied.  The random draws of the `equal` / `inconsistent` types
 * come from a counter-based hash of (variable, round, seed) instead of thrust::default_random_engine. */
int bddmma_perturb_primal_costs(bddmma_solver* s, bddmma_lbfgs* lbfgs_or_null, double cur_delta, uint32_t round_index, uint32_t seed,
                                uint32_t counts[4], char* sol, void* cost_delta_0, void* cost_delta_1);

/* ---- checkpoint (bdd_cuda_base.cu:1486-1550) ------------------------------ */
int bddmma_save(const bddmma_solver* s, const char* path);
int bddmma_load(bddmma_solver** out, int device, const char* path);

/* ---- measurement ---------------------------------------------------------- */
int bddmma_synchronize(bddmma_solver* s);
/* Kernel classes timed with hipEvents on the handle's stream when profiling is on. */
#define BDDMMA_K_FORWARD_MM 0
#define BDDMMA_K_BACKWARD_MM 1
#define BDDMMA_K_FINISH_DELTA 2
#define BDDMMA_K_OTHER 3
#define BDDMMA_K_COUNT 4
typedef struct bddmma_profile {
    uint64_t launches[BDDMMA_K_COUNT];
    double total_ms[BDDMMA_K_COUNT];
} bddmma_profile;
/* on = 0: off; on = n > 0: record events for every n-th iteration() (an event pair per launch costs ~4 us of
 * stream time, so n = 1 slows a 10.5 M-node iteration by ~14 %).  Resets the counters. */
int bddmma_set_profiling(bddmma_solver* s, int on);
int bddmma_get_profile(bddmma_solver* s, bddmma_profile* out);  /* synchronises */
/* Run n iterations bracketed by hipEvents on the handle's stream; *ms = elapsed device time. */
int bddmma_time_iterations(bddmma_solver* s, double omega, uint64_t n, double* ms);
/* Time `reps` back-to-back launches of one kernel class with hipEvents on the handle's stream
 * (kernel-level benchmarking; leaves the sweep state invalid).  kind: 0 forward_run sweep, 1 backward_run
 * sweep, 2 forward_mm sweep, 3 backward_mm sweep, 4 exchange reduce, 5 exchange broadcast, 6 STREAM triad
 * a = b + s*c over three temporary arrays of BDDMMA_TRIAD_BYTES each (3 * BDDMMA_TRIAD_BYTES of HBM traffic
 * per launch), 7 STREAM copy a = b (2 * BDDMMA_TRIAD_BYTES per launch): the measured bandwidth ceilings of the
 * box the roofline is quoted next to. */
#define BDDMMA_TRIAD_BYTES (1ull << 30)
int bddmma_time_kernel(bddmma_solver* s, int kind, uint64_t reps, double* ms);
/* HBM bytes of the arrays the handle holds (its working set), and the bytes it has allocated for them: the
 * arrays are carved out of few large allocations, so allocated >= held (what a farm of many small solvers
 * must budget with). */
uint64_t bddmma_device_bytes(const bddmma_solver* s);
uint64_t bddmma_device_allocated_bytes(const bddmma_solver* s);

/* ---- host-only layout inspection (no GPU needed; used by the CPU test-suite) ------------- */
typedef struct bddmma_layout bddmma_layout;
int bddmma_layout_create(bddmma_layout** out, const bddmma_instruction* instr, const uint64_t* bdd_delims,
                         uint64_t n_bdds, const bddmma_options* opts);
/* The same for values of real_size bytes (4 / 8) on a chip with n_cus compute units and lds_bytes_per_cu of LDS each (0: MI355X, what
 * bddmma_layout_create assumes): the layout bddmma_create builds on such a device (bddmma_device_chip), e.g. a CPX / DPX partition. */
int bddmma_layout_create_for_chip(bddmma_layout** out, const bddmma_instruction* instr, const uint64_t* bdd_delims,
                                  uint64_t n_bdds, const bddmma_options* opts, int real_size, uint32_t n_cus, uint32_t lds_bytes_per_cu);
void bddmma_layout_destroy(bddmma_layout* l);
uint64_t bddmma_layout_size(const bddmma_layout* l, int what);
int bddmma_layout_copy(const bddmma_layout* l, int which, void* out);
/* The per-lane records of the second-generation resident sweeps (derived data, csrc/layout.hpp: Res2Records) for values of real_size
 * bytes: info[0] = usable, [1] = number of 32-bit words (4 per record), [2] / [3] = the slot / layer capacity of a wave's LDS region the
 * offsets assume, [4] = hops of the longest pack.  words (info[1] entries) and rec_off (one per narrow pack) may be NULL to query sizes. */
int bddmma_layout_res2_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off);
/* The same for the records of the second-generation streaming sweeps (csrc/layout.hpp: StreamRecords): info[0] = usable, [1] = words. */
int bddmma_layout_stream_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off);
/* ... and the per-layer records of the third-generation streaming sweeps (csrc/layout.hpp: LayerRecords; 64 records of 4 words per hop):
 * info[0] = usable (packs of 128 slots, layers of <= 2 nodes, <= 64 layers per hop, no staggered packs), [1] = number of 32-bit words. */
int bddmma_layout_layer_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off);
/* The schedule of the atomic-free exchange (csrc/layout.hpp: SegExchange) for workgroups of `threads` and values of real_size bytes:
 * info[0] = usable, [1] = 32-bit words of `bin` (4 per bin), [2] = 16-bit words of `perm`, [3] = 32-bit words of `thr` (2 per bin and
 * thread), [4] / [5] / [6] = entries / slots / groups of the largest bin.  The arrays may be NULL to query sizes. */
int bddmma_layout_seg_exchange(const bddmma_layout* l, int threads, int real_size, uint32_t* info, uint32_t* bin, uint16_t* perm, uint32_t* thr);

#ifdef __cplusplus
}
#endif
#endif /* BDD_MMA_H */
