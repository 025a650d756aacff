// solver.hpp — precision-erased solver interface behind the C-ABI (include/bdd_mma.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bdd_mma.h"
#include "layout.hpp"

namespace bddmma {

struct SolverBase {
    int precision = BDDMMA_F32, device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    uint64_t n_vars = 0, n_bdds = 0, n_layers = 0, n_hops = 0, n_input_nodes = 0, n_slots = 0;
    int solve_sweep_kind = 0;  // bddmma_solve_sweep_kind (include/bdd_mma.h): which kernels run the narrow packs' solve sweeps
    uint32_t pack_width = 0, wide_pack_width = 0, wide_slot_base = 0;
    uint64_t dev_bytes = 0;          // bytes of the arrays the solver holds
    uint64_t dev_alloc_bytes = 0;    // bytes hipMalloc'd for them (arena capacity included: >= dev_bytes)
    bool fwd_valid = false, bwd_valid = false;  // forward_state_valid_ / backward_state_valid_ (bdd_cuda_base.h:205-206)
    uint64_t cost_epoch = 0;                    // counts the calls that changed arc costs other than through a solve sweep (update_costs, set_cost, gradient steps, ...)
    bool deterministic = false;
    std::vector<uint64_t> nodes_per_hop, layers_per_hop;
    std::vector<int32_t> h_nbdds, h_layer_var, h_layer_bdd;
    std::vector<uint32_t> h_var_ptr;
    uint32_t n_packs_narrow = 0, n_packs_wide = 0;
    // checkpoint support (bdd_cuda_base.cu:1486-1550): the layout itself is archived — scalars here, arrays fetched from the device
    LayoutScalars lay_scalars{};
    bddmma_options saved_opts{};
    virtual int download_layout(HostLayout& H) = 0;  // rebuilds a HostLayout from what the device holds
    virtual int init_from_layout(const HostLayout& L, const bddmma_options* opts) = 0;  // device buffers + upload (create_solver)

    // profiling: one hipEvent pair per launch group on `stream`
    bool profiling = false;        // hipEvent pairs are recorded for every `prof_stride`-th iteration only: an event
    uint32_t prof_stride = 1;      // pair per launch costs ~4 us of stream time (measured: -14 % it/s at stride 1)
    uint64_t prof_iter = 0;
    bool prof_active = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_class;
    size_t ev_used = 0;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;

    virtual ~SolverBase() {}
    virtual int forward_run() = 0;
    virtual int backward_run() = 0;
    virtual int lower_bound(double* lb) = 0;
    // The same bound without the host waiting for it: enqueue the backward run + reduction into one of two pinned slots, fetch it
    // later (the fetch waits for the stream).  The value is the bound of the costs at enqueue time, whatever was enqueued behind it.
    virtual int lower_bound_enqueue(int slot) = 0;
    virtual int lower_bound_fetch(int slot, double* lb) = 0;
    virtual int lower_bound_per_bdd(void* out, int on_device) = 0;
    virtual int iteration(double omega) = 0;
    // n iterations; instances that fit one workgroup run them inside one launch (kernels/small.hpp), everything else calls iteration() n times
    virtual int iterations(double omega, uint64_t n)
    {
        for (uint64_t i = 0; i < n; ++i)
            if (int rc = iteration(omega)) return rc;
        return BDDMMA_OK;
    }
    bool fused_small = false;   // whole iterations in one launch (diagnostics: bddmma_fused_small)
    bool nt_loads = false;      // the solve sweeps' non-temporal instantiation (diagnostics: bddmma_nontemporal_loads)
    // run_solver (include/run_solver_util.h:10-77) around iteration(): termination tests on the device, see solver_impl.hpp
    virtual int run_plain(uint64_t max_iter, double tolerance, double slope, double time_limit, int verbose, bddmma_run_result* res) = 0;
    virtual int forward_mm(double omega, void* delta, int on_device) = 0;
    virtual int backward_mm(double omega, void* delta, int on_device) = 0;
    virtual int normalize_delta(void* delta, int on_device) = 0;
    virtual int distribute_delta() = 0;
    virtual int get_delta(void* out, int on_device) = 0;
    virtual int set_delta(const void* in, int on_device) = 0;
    virtual int update_costs(const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi, int elem_precision, int on_device) = 0;
    virtual int set_cost(double c, uint64_t var) = 0;
    virtual int get_solver_costs(void* lo, void* hi, void* mm, int on_device) = 0;
    virtual int set_solver_costs(const void* lo, const void* hi, const void* mm, int on_device) = 0;
    virtual int primal_objective_vec(void* out, int on_device) = 0;
    virtual int min_marginals(int sorted, int32_t* var, void* mm0, void* mm1, int on_device) = 0;
    virtual int min_marginal_diff(void* out, int on_device) = 0;
    virtual int bdds_solution(int sorted, char* sol, int on_device) = 0;
    virtual int net_solver_costs(void* out, int on_device) = 0;
    virtual int make_dual_feasible(void* g, int on_device) = 0;
    virtual int gradient_step(const void* g, double step, int on_device) = 0;
    virtual int projection_means(const void* g_dev) = 0;                       // per-variable means of a device vector, kept inside the solver
    virtual int gradient_step_projected(const void* g_dev, double step) = 0;   // costs += step * (g - its per-variable mean)
    // projection_means of a vector that is a linear combination of stored ones (layout.hpp: LinComb), formed on the fly; `tag` is what
    // gradient_step_projected will be called with.  Only where projection_fuses_lincomb() (staged projection, narrow packs only).
    virtual bool projection_fuses_lincomb() const = 0;
    virtual int projection_means_lincomb(const LinComb& lc, const void* tag) = 0;
    virtual void* stream_handle() = 0;
    // L-BFGS wrapper (lbfgs.hip): the argmin paths straight into a device buffer with nothing but stream order (bdds_solution() also
    // copies and synchronises; `prezeroed`: the caller has left the buffer all 0, the sweep only writes the layers on a path), and
    // net_solver_costs() as a view: x = (hi - lo) + deferred mm per layer.
    struct LbfgsViews {
        const void* x_layer;       // REAL per layer
    };
    virtual int bdds_solution_async(char* dev_out, int prezeroed) = 0;
    // From the first call on the backward solve sweeps also write x in layer order (one coalesced 4-8 byte store per layer, formed from
    // the new arc costs and the deferred difference while both are in the sweep's hands), so that the wrapper reads x without the 5 M random
    // gathers of net_solver_costs() and without re-reading {lo, hi}; rebuilt by net_solver_costs()'s gather when something else has changed
    // costs or deferred values since the last backward solve sweep.
    virtual int lbfgs_views(LbfgsViews* out) = 0;
    virtual int time_kernel(int kind, uint64_t reps, double* ms) = 0;
    // one round of perturb_primal_costs (incremental_mm_agreement_rounding_cuda.cu:262-331); counts = #one,#zero,#equal,#inconsistent
    // `applied` = 0 when the solution was read off (all variables one / zero) and the costs were left alone; c0_host / c1_host
    // (REAL[n_vars], may be null) receive the perturbation.  With apply_update = false the perturbation is computed but not applied.
    virtual int rounding_round(double delta, uint32_t round, uint32_t seed, uint32_t counts[4], char* sol_host, void* c0_host, void* c1_host,
                               bool apply_update, int* applied) = 0;
    virtual int rounding_scratch(void** c0_dev, void** c1_dev) = 0;  // device vectors holding the last perturbation (REAL[n_vars] each)

    int synchronize();
    void prof_begin(int kclass);
    void prof_end(int kclass);
    int set_profiling(int on);
    int get_profile(bddmma_profile* out);
    int time_iterations(double omega, uint64_t n, double* ms);
};

int device_count();
int query_chip(int device, ChipInfo* out, std::string& err);  // CU count and LDS per CU of `device` (hipDeviceProp)
int create_solver(SolverBase** out, int precision, int device, const HostLayout& L, const bddmma_options* opts, std::string& err);

}  // namespace bddmma

// the opaque C handle
struct bddmma_solver {
    bddmma::SolverBase* impl = nullptr;
};
