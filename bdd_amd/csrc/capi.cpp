// capi.cpp — extern "C" entry points of include/bdd_mma.h (everything except the L-BFGS ones,
// which live in lbfgs.hip).  No exception crosses this boundary.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/bdd_mma.h"
#include "layout.hpp"
#include "solver.hpp"

using namespace bddmma;

struct bddmma_lbfgs;
extern "C" int bddmma_lbfgs_iteration(bddmma_lbfgs* l);
extern "C" int bddmma_lbfgs_update_costs(bddmma_lbfgs* l, const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi, int elem_precision,
                                         int on_device);

namespace {
thread_local std::string g_err;

template <typename F>
int guarded(bddmma_solver* s, F&& f)
{
    if (!s || !s->impl) return BDDMMA_ERR_INVALID_ARGUMENT;
    try {
        return f(s->impl);
    } catch (const std::bad_alloc&) {
        s->impl->err = "out of host memory";
        return BDDMMA_ERR_DEVICE;
    } catch (const std::exception& e) {
        s->impl->err = e.what();
        return BDDMMA_ERR_DEVICE;
    }
}
template <typename F>
int guarded(const bddmma_solver* s, F&& f) { return guarded(const_cast<bddmma_solver*>(s), f); }

template <typename T>
int copy_host(const std::vector<T>& v, T* out)
{
    if (!out) return BDDMMA_ERR_INVALID_ARGUMENT;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(T));
    return BDDMMA_OK;
}
}  // namespace

extern "C" {

int bddmma_create(bddmma_solver** out, int precision, int device, const bddmma_instruction* instr,
                  const uint64_t* bdd_delims, uint64_t n_bdds, const double* costs_hi, uint64_t n_costs,
                  const bddmma_options* opts)
{
    if (!out) return BDDMMA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
#ifdef BDDMMA_EXPERIMENTAL  // make EXPERIMENTAL=1: phase timings of the construction on stderr (tools/construct_time.py)
        static const bool timing = std::getenv("BDDMMA_LAYOUT_TIMING") != nullptr;
#else
        constexpr bool timing = false;
#endif
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[create] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        HostLayout L;
        ChipInfo chip;
        int rc = query_chip(device, &chip, g_err);
        if (rc) return rc;
        rc = build_layout(instr, bdd_delims, n_bdds, opts, L, g_err, false, precision == BDDMMA_F64 ? 8 : 4, chip);
        if (rc) return rc;
        lap("host layout");
        SolverBase* impl = nullptr;
        rc = create_solver(&impl, precision, device, L, opts, g_err);
        if (rc) return rc;
        lap("device buffers + upload");
        impl->n_packs_narrow = L.narrow.n_packs();
        impl->n_packs_wide = L.wide.n_packs() + L.huge.n_packs();
        if (costs_hi && n_costs) {
            rc = impl->update_costs(nullptr, 0, costs_hi, n_costs, BDDMMA_F64, 0);
            if (rc) {
                g_err = impl->err;
                delete impl;
                return rc;
            }
        }
        lap("update_costs");
        *out = new bddmma_solver{impl};
        return BDDMMA_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return BDDMMA_ERR_DEVICE;
    }
}

int bddmma_device_count(void) { return device_count(); }
int bddmma_device_chip(int device, uint32_t* n_cus, uint32_t* lds_bytes_per_cu, uint64_t* max_resident_threads)
{
    if (device < 0 || device >= device_count()) { g_err = "no such HIP device"; return BDDMMA_ERR_DEVICE; }
    ChipInfo chip;
    const int rc = query_chip(device, &chip, g_err);
    if (rc) return rc;
    if (n_cus) *n_cus = chip.n_cus;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = chip.lds_bytes;
    if (max_resident_threads) *max_resident_threads = chip.max_resident_threads;
    return BDDMMA_OK;
}
int bddmma_set_layout_threads(int n)
{
    if (n < 0) return BDDMMA_ERR_INVALID_ARGUMENT;
    set_layout_threads((unsigned)n);
    return BDDMMA_OK;
}
int bddmma_set_thread_layout_threads(int n)
{
    if (n < 0) return BDDMMA_ERR_INVALID_ARGUMENT;
    set_thread_layout_threads((unsigned)n);
    return BDDMMA_OK;
}

void bddmma_destroy(bddmma_solver* s)
{
    if (!s) return;
    delete s->impl;
    delete s;
}

const char* bddmma_last_error(const bddmma_solver* s) { return (s && s->impl) ? s->impl->err.c_str() : g_err.c_str(); }

uint64_t bddmma_nr_variables(const bddmma_solver* s) { return s && s->impl ? s->impl->n_vars : 0; }
uint64_t bddmma_nr_bdds(const bddmma_solver* s) { return s && s->impl ? s->impl->n_bdds : 0; }
uint64_t bddmma_nr_layers(const bddmma_solver* s) { return s && s->impl ? s->impl->n_layers : 0; }
uint64_t bddmma_nr_bdd_nodes(const bddmma_solver* s) { return s && s->impl ? s->impl->n_input_nodes : 0; }
uint64_t bddmma_nr_hops(const bddmma_solver* s) { return s && s->impl ? s->impl->n_hops : 0; }
uint64_t bddmma_nr_packs(const bddmma_solver* s) { return s && s->impl ? s->impl->n_packs_narrow + s->impl->n_packs_wide : 0; }
int bddmma_solve_sweep_kind(const bddmma_solver* s) { return s && s->impl ? s->impl->solve_sweep_kind : -1; }
int bddmma_fused_small(const bddmma_solver* s) { return s && s->impl ? (s->impl->fused_small ? 1 : 0) : -1; }
int bddmma_nontemporal_loads(const bddmma_solver* s) { return s && s->impl ? (s->impl->nt_loads ? 1 : 0) : -1; }
int bddmma_precision(const bddmma_solver* s) { return s && s->impl ? s->impl->precision : -1; }
int bddmma_device(const bddmma_solver* s) { return s && s->impl ? s->impl->device : -1; }
uint64_t bddmma_device_bytes(const bddmma_solver* s) { return s && s->impl ? s->impl->dev_bytes : 0; }
uint64_t bddmma_device_allocated_bytes(const bddmma_solver* s) { return s && s->impl ? s->impl->dev_alloc_bytes : 0; }

int bddmma_num_bdds_per_var(const bddmma_solver* s, int32_t* out)
{
    return guarded(s, [&](SolverBase* b) { return copy_host(b->h_nbdds, out); });
}
int bddmma_layer_variables(const bddmma_solver* s, int32_t* out)
{
    return guarded(s, [&](SolverBase* b) { return copy_host(b->h_layer_var, out); });
}
int bddmma_layer_bdds(const bddmma_solver* s, int32_t* out)
{
    return guarded(s, [&](SolverBase* b) { return copy_host(b->h_layer_bdd, out); });
}
int bddmma_nodes_per_hop(const bddmma_solver* s, uint64_t* out)
{
    return guarded(s, [&](SolverBase* b) { return copy_host(b->nodes_per_hop, out); });
}
int bddmma_layers_per_hop(const bddmma_solver* s, uint64_t* out)
{
    return guarded(s, [&](SolverBase* b) { return copy_host(b->layers_per_hop, out); });
}

int bddmma_update_costs(bddmma_solver* s, const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi,
                        int elem_precision, int on_device)
{
    return guarded(s, [&](SolverBase* b) {
        if (elem_precision != BDDMMA_F32 && elem_precision != BDDMMA_F64) {
            b->err = "elem_precision must be BDDMMA_F32 or BDDMMA_F64";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        return b->update_costs(lo, n_lo, hi, n_hi, elem_precision, on_device);
    });
}
int bddmma_set_cost(bddmma_solver* s, double c, uint64_t var)
{
    return guarded(s, [&](SolverBase* b) { return b->set_cost(c, var); });
}
int bddmma_get_solver_costs(const bddmma_solver* s, void* lo, void* hi, void* mm, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return b->get_solver_costs(lo, hi, mm, on_device); });
}
int bddmma_set_solver_costs(bddmma_solver* s, const void* lo, const void* hi, const void* mm, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return b->set_solver_costs(lo, hi, mm, on_device); });
}
int bddmma_primal_objective_vec(bddmma_solver* s, void* out, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return b->primal_objective_vec(out, on_device); });
}
int bddmma_forward_run(bddmma_solver* s) { return guarded(s, [&](SolverBase* b) { return b->forward_run(); }); }
int bddmma_backward_run(bddmma_solver* s) { return guarded(s, [&](SolverBase* b) { return b->backward_run(); }); }
int bddmma_lower_bound(bddmma_solver* s, double* lb)
{
    return guarded(s, [&](SolverBase* b) { return lb ? b->lower_bound(lb) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_lower_bound_per_bdd(bddmma_solver* s, void* out, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return b->lower_bound_per_bdd(out, on_device); });
}
int bddmma_iteration(bddmma_solver* s, double omega)
{
    return guarded(s, [&](SolverBase* b) { return b->iteration(omega); });
}
int bddmma_iterations(bddmma_solver* s, double omega, uint64_t n)
{
    return guarded(s, [&](SolverBase* b) { return b->iterations(omega, n); });
}
int bddmma_forward_mm(bddmma_solver* s, double omega, void* d, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return d ? b->forward_mm(omega, d, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_backward_mm(bddmma_solver* s, double omega, void* d, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return d ? b->backward_mm(omega, d, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_normalize_delta(const bddmma_solver* s, void* d, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return d ? b->normalize_delta(d, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_distribute_delta(bddmma_solver* s) { return guarded(s, [&](SolverBase* b) { return b->distribute_delta(); }); }
int bddmma_get_delta(const bddmma_solver* s, void* out, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return out ? b->get_delta(out, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_set_delta(bddmma_solver* s, const void* in, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return in ? b->set_delta(in, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_min_marginals(bddmma_solver* s, int sorted, int32_t* var, void* mm0, void* mm1, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return b->min_marginals(sorted, var, mm0, mm1, on_device); });
}
int bddmma_min_marginal_diff(bddmma_solver* s, void* out, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return out ? b->min_marginal_diff(out, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_bdds_solution(bddmma_solver* s, int sorted, char* sol, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return sol ? b->bdds_solution(sorted, sol, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_net_solver_costs(const bddmma_solver* s, void* out, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return out ? b->net_solver_costs(out, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_make_dual_feasible(const bddmma_solver* s, void* g, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return g ? b->make_dual_feasible(g, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_gradient_step(bddmma_solver* s, const void* g, double step, int on_device)
{
    return guarded(s, [&](SolverBase* b) { return g ? b->gradient_step(g, step, on_device) : BDDMMA_ERR_INVALID_ARGUMENT; });
}

// run_solver, include/run_solver_util.h:10-77.  host_loop: the reference's literal loop (iteration, lower_bound, tests on the host) for the
// plain solver too — bddmma_run_solver_host_loop, the twin the device-resident loop is tested against.
static int run_solver_impl(bddmma_solver* s, bddmma_lbfgs* lbfgs, uint64_t max_iter, double tolerance, double improvement_slope, double time_limit,
                           int verbose, bddmma_run_result* res, bool host_loop)
{
    return guarded(s, [&](SolverBase* b) {
        if (improvement_slope < 0.0 || improvement_slope >= 1.0 || time_limit < 0.0 || tolerance < 0.0) {
            b->err = "run_solver: invalid termination criteria";  // asserts at run_solver_util.h:13-15
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        // plain MMA: the loop runs with its termination tests on the device (no host round trip per iteration); the L-BFGS wrapper
        // decides its steps on the host anyway and keeps the sequential loop below
        if (!lbfgs && !host_loop) return b->run_plain(max_iter, tolerance, improvement_slope, time_limit, verbose, res);
        const auto t0 = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        double lb_initial;
        int rc = b->lower_bound(&lb_initial);
        if (rc) return rc;
        double lb_first = std::numeric_limits<double>::max(), lb_prev = lb_initial, lb_post = lb_initial;
        if (verbose) std::printf("[bdd solver] initial lower bound = %.10g, time = %.3f s\n", lb_prev, elapsed());
        uint64_t iter = 0;
        int reason = 0;
        for (; iter < max_iter; ++iter) {
            rc = lbfgs ? bddmma_lbfgs_iteration(lbfgs) : b->iteration(0.5);
            if (rc) return rc;
            lb_prev = lb_post;
            if ((rc = b->lower_bound(&lb_post))) return rc;
            if (iter == 0) lb_first = lb_post;
            const double t = elapsed();
            if (verbose) std::printf("[bdd solver] iteration %llu, lower bound = %.10g, time = %.3f s\n", (unsigned long long)iter, lb_post, t);
            if (t > time_limit) { reason = 1; ++iter; break; }
            if (std::abs(lb_prev - lb_post) < std::abs(tolerance * lb_prev)) { reason = 2; ++iter; break; }
            if (std::abs(lb_prev - lb_post) < improvement_slope * std::abs(lb_initial - lb_first)) { reason = 3; ++iter; break; }
            if (lb_post == std::numeric_limits<double>::infinity()) { reason = 4; ++iter; break; }
        }
        if (verbose) std::printf("[bdd solver] final lower bound = %.10g\n", lb_post);
        if (res) {
            res->iterations = iter;
            res->lb_initial = lb_initial;
            res->lb_final = lb_post;
            res->seconds = elapsed();
            res->stop_reason = reason;
        }
        return BDDMMA_OK;
    });
}
int bddmma_run_solver(bddmma_solver* s, bddmma_lbfgs* lbfgs, uint64_t max_iter, double tolerance, double improvement_slope, double time_limit,
                      int verbose, bddmma_run_result* res)
{
    return run_solver_impl(s, lbfgs, max_iter, tolerance, improvement_slope, time_limit, verbose, res, false);
}
int bddmma_run_solver_host_loop(bddmma_solver* s, bddmma_lbfgs* lbfgs, uint64_t max_iter, double tolerance, double improvement_slope,
                                double time_limit, int verbose, bddmma_run_result* res)
{
    return run_solver_impl(s, lbfgs, max_iter, tolerance, improvement_slope, time_limit, verbose, res, true);
}

// perturb_primal_costs (incremental_mm_agreement_rounding_cuda.cu:262-331).  The cost update goes through the solver type the
// caller holds: with an L-BFGS wrapper that is lbfgs::update_costs, which drops the (s, y) history first (lbfgs_impl.h:343-364).
static int perturb_round(SolverBase* b, bddmma_lbfgs* lbfgs, double cur_delta, uint32_t round, uint32_t seed, uint32_t counts[4], char* sol,
                         void* c0_host, void* c1_host, int* applied)
{
    int rc = b->rounding_round(cur_delta, round, seed, counts, sol, c0_host, c1_host, lbfgs == nullptr, applied);
    if (rc || !*applied || !lbfgs) return rc;
    void *c0 = nullptr, *c1 = nullptr;
    if ((rc = b->rounding_scratch(&c0, &c1))) return rc;
    return bddmma_lbfgs_update_costs(lbfgs, c0, b->n_vars, c1, b->n_vars, b->precision, 1);
}

int bddmma_perturb_primal_costs(bddmma_solver* s, bddmma_lbfgs* lbfgs, double cur_delta, uint32_t round_index, uint32_t seed,
                                uint32_t counts[4], char* sol, void* cost_delta_0, void* cost_delta_1)
{
    if (!sol || !counts) return BDDMMA_ERR_INVALID_ARGUMENT;
    return guarded(s, [&](SolverBase* b) {
        int applied = 0;
        return perturb_round(b, lbfgs, cur_delta, round_index, seed, counts, sol, cost_delta_0, cost_delta_1, &applied);
    });
}

// incremental_mm_agreement_rounding_cuda (incremental_mm_agreement_rounding_cuda.cu:333-372)
int bddmma_incremental_mm_agreement_rounding(bddmma_solver* s, bddmma_lbfgs* lbfgs, double init_delta, double delta_growth_rate,
                                             uint64_t num_itr_lb, uint64_t num_rounds, uint32_t seed, int verbose, char* sol, int* found)
{
    if (!sol || !found) return BDDMMA_ERR_INVALID_ARGUMENT;
    *found = 0;
    return guarded(s, [&](SolverBase* b) {
        if (!(delta_growth_rate > 0) || !(init_delta > 0)) {
            b->err = "rounding: init_delta and delta_growth_rate must be positive";  // asserts at :336-337
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        int rc = b->distribute_delta();
        if (rc) return rc;
        double lb;
        if ((rc = b->lower_bound(&lb))) return rc;
        if (verbose) std::printf("[incremental primal rounding] lower bound after distributing delta: %.10g\n", lb);
        double cur_delta = init_delta / delta_growth_rate;
        for (uint64_t round = 0; round < num_rounds; ++round) {
            cur_delta = std::min(cur_delta * delta_growth_rate, 1e6);
            uint32_t counts[4];
            int applied = 0;
            if ((rc = perturb_round(b, lbfgs, cur_delta, (uint32_t)round, seed, counts, sol, nullptr, nullptr, &applied))) return rc;
            if (verbose)
                std::printf("[incremental primal rounding] round %llu, cost delta %g: #ones %u, #zeros %u, #equal %u, #inconsistent %u\n",
                            (unsigned long long)round, cur_delta, counts[0], counts[1], counts[2], counts[3]);
            if (!applied) {
                *found = 1;
                return BDDMMA_OK;
            }
            bddmma_run_result rr;
            rc = bddmma_run_solver(s, lbfgs, num_itr_lb, 1e-7, 0.0001, std::numeric_limits<double>::max(), 0, &rr);  // :366
            if (rc) return rc;
            if (verbose) std::printf("[incremental primal rounding] lower bound = %.10g\n", rr.lb_final);
        }
        return BDDMMA_OK;
    });
}

// ---- checkpoint (bdd_cuda_base.cu:1486-1550: every index array of the layout + the costs are archived) -------------------------
// File: magic, header {precision, #arrays, sizeof(LayoutScalars), sizeof(bddmma_options)}, LayoutScalars, options, the layout arrays
// as {id, element size, count, data} records (layout.hpp: visit_layout_arrays), then lo / hi / deferred mm / delta.  Loading
// uploads the arrays as they are: build_layout does not run again.
static const char kMagic[8] = {'B', 'D', 'D', 'M', 'M', 'A', '0', '7'};  // 07: hop_root of the wide packs too; 06: hop_root (staggered packs); 05: checksum of the layout section; 04: narrow node words carry the layer index

namespace {
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) std::fclose(f); }
};
// Checksum of the layout section: four independent multiply-rotate lanes over 8-byte words, so that a 240 MB section costs
// ~20 ms; byte order is the host's (little endian on every target of this library), a foreign-endian file fails the magic / header test.
struct Checksum {
    uint64_t lane[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    uint64_t n = 0;
    static uint64_t mix(uint64_t h, uint64_t w)
    {
        h ^= w * 0x9FB21C651E98DF25ull;
        h = (h << 27) | (h >> 37);
        return h * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    }
    void add(const void* p, size_t bytes)
    {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        n += bytes;
        size_t i = 0;
        for (; i + 32 <= bytes; i += 32) {
            uint64_t w[4];
            std::memcpy(w, c + i, 32);
            for (int k = 0; k < 4; ++k) lane[k] = mix(lane[k], w[k]);
        }
        for (int k = 0; i < bytes; i += 8, ++k) {
            uint64_t w = 0;
            std::memcpy(&w, c + i, std::min<size_t>(8, bytes - i));
            lane[k & 3] = mix(lane[k & 3], w ^ 0xA5A5A5A5A5A5A5A5ull);
        }
    }
    uint64_t value() const { return mix(mix(mix(mix(n, lane[0]), lane[1]), lane[2]), lane[3]); }
};

// Consistency of a layout read from a file.  The arrays go to kernels that index LDS and global memory with them, and to host code
// that copies them into caller buffers, so everything that is used as an index, an offset or a size is checked against its target
// before a solver is built from it (one linear pass per array).  Tables that are functions of other tables (the resident
// headers) are recomputed and compared.
bool layout_plausible(const HostLayout& L, std::string& why)
{
    auto fail = [&](const std::string& m) { why = m; return false; };
    auto mono_to = [&](const auto& v, uint64_t first, uint64_t last, const char* name) {
        if (v.empty()) { why = std::string(name) + " is empty"; return false; }
        if (v.front() != first || v.back() != last) { why = std::string(name) + " does not span its range"; return false; }
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i] < v[i - 1]) { why = std::string(name) + " is not monotone"; return false; }
        return true;
    };
    auto all_below = [&](const auto& v, uint64_t bound, const char* name) {
        for (auto x : v)
            if ((uint64_t)x >= bound) { why = std::string(name) + " out of range"; return false; }
        return true;
    };
    const Exchange& X = L.ex;
    if (L.n_layers == 0 || L.n_vars == 0 || L.n_bdds == 0) return fail("empty layout");
    // ---- scalars the launch configuration and the LDS carving are computed from
    if (L.pack_width != 64 && L.pack_width != 128 && L.pack_width != 256) return fail("pack_width");
    if (X.waves_per_block != 1 && X.waves_per_block != 2 && X.waves_per_block != 4 && X.waves_per_block != 8) return fail("waves_per_block");
    if (X.stage_cap == 0 || X.stage_cap > 640) return fail("stage_cap");  // kernels.hpp: stage_cap <= 64 * STAGE_ITERS
    if (L.wide_pack_width > 4096) return fail("wide_pack_width");
    if (X.vars_per_bin == 0 || X.vars_per_bin > 65536 || (uint64_t)X.n_bins != (L.n_vars + X.vars_per_bin - 1) / X.vars_per_bin) return fail("vars_per_bin / n_bins");
    if (L.n_slots >= (1ull << 32) || L.n_layers >= (1ull << 31) || L.n_vars >= (1ull << 31) || L.narrow_slots > L.n_slots) return fail("sizes");
    // ---- array sizes
    if (L.layer_var.size() != L.n_layers || L.layer_bdd.size() != L.n_layers || L.var_layers.size() != L.n_layers ||
        X.lpos.size() != L.n_layers || X.evar.size() != L.n_layers || X.bvar.size() != L.n_layers || X.vpos.size() != L.n_layers ||
        L.num_bdds_per_var.size() != L.n_vars || L.var_ptr.size() != L.n_vars + 1 || L.bdd_root_slot.size() != L.n_bdds ||
        X.bin_ptr.size() != (size_t)X.n_bins + 1) return fail("array sizes do not match the header");
    if (L.nodes_per_hop.size() != L.n_hops || L.layers_per_hop.size() != L.n_hops) return fail("per-hop statistics do not have n_hops entries");
    if (!mono_to(L.var_ptr, 0, L.n_layers, "var_ptr") || !mono_to(X.bin_ptr, 0, L.n_layers, "bin_ptr")) return false;
    // ---- packs: offsets monotone and contiguous over the three sets, hops no wider than the set's pack width
    uint64_t slots = 0, layers = 0;
    const uint32_t widths[3] = {L.pack_width, L.wide_pack_width, L.huge_pack_width};
    int set = 0;
    for (const PackSet* ps : {&L.narrow, &L.wide, &L.huge}) {
        const uint32_t P = ps->n_packs(), maxw = widths[set++];
        if (P == 0) {
            if (!ps->hop_node_off.empty() && ps->hop_node_off.size() != 1) return fail("hop records without packs");
            continue;
        }
        if (ps->pack_steps.size() != P) return fail("pack_steps size");
        if (ps->hop_node_off.empty() || !mono_to(ps->pack_hop_ptr, 0, ps->hop_node_off.size() - 1, "pack_hop_ptr")) return fail("pack_hop_ptr");
        if (ps->hop_layer_off.size() != ps->hop_node_off.size()) return fail("hop offset sizes");
        if (!mono_to(ps->hop_node_off, slots, ps->hop_node_off.back(), "hop_node_off") ||
            !mono_to(ps->hop_layer_off, layers, ps->hop_layer_off.back(), "hop_layer_off")) return false;
        for (size_t q = 0; q + 1 < ps->hop_node_off.size(); ++q)
            if (ps->hop_node_off[q + 1] - ps->hop_node_off[q] > maxw || ps->hop_layer_off[q + 1] - ps->hop_layer_off[q] > maxw) return fail("hop wider than its pack");
        slots = ps->hop_node_off.back();
        layers = ps->hop_layer_off.back();
    }
    if (slots != L.n_slots || layers != L.n_layers) return fail("pack offsets do not cover the slots / layers");
    if (L.narrow.n_packs() ? L.narrow.hop_node_off.back() != L.narrow_slots : L.narrow_slots != 0) return fail("narrow_slots");
    // ---- per-layer / per-variable / per-entry index arrays
    if (!all_below(X.lpos, L.n_layers, "lpos") || !all_below(X.vpos, L.n_layers, "vpos") || !all_below(L.var_layers, L.n_layers, "var_layers") ||
        !all_below(X.evar, L.n_vars, "evar") || !all_below(X.bvar, X.vars_per_bin, "bvar") || !all_below(L.bdd_root_slot, L.n_slots, "bdd_root_slot")) return false;
    for (int32_t v : L.layer_var) if (v < 0 || (uint64_t)v >= L.n_vars) return fail("layer variable out of range");
    for (int32_t b : L.layer_bdd) if (b < 0 || (uint64_t)b >= L.n_bdds) return fail("layer BDD out of range");
    for (int32_t c : L.num_bdds_per_var) if (c < 0) return fail("negative BDD count");
    for (uint32_t b = 0; b < X.n_bins; ++b)  // an entry's variable lies in the bin that holds the entry
        for (uint32_t e = X.bin_ptr[b]; e < X.bin_ptr[b + 1]; ++e)
            if (X.evar[e] / X.vars_per_bin != b || X.evar[e] % X.vars_per_bin != X.bvar[e]) return fail("entry outside its variable's bin");
    // ---- narrow packs: words, stage groups, cooperative staging, resident headers
    const uint32_t Pn = L.narrow.n_packs();
    if (Pn) {
        const PackSet& N = L.narrow;
        const uint32_t nl = N.hop_layer_off.back(), n_rec = (uint32_t)N.hop_node_off.size() - 1;
        const uint32_t WPB = X.waves_per_block, n_quads = (Pn + WPB - 1) / WPB;
        if (L.narrow_word_off.size() != Pn || X.pack_group_ptr.size() != (size_t)Pn + 1 || L.res.pack_hdr.size() != (size_t)Pn * 8 ||
            L.res.quad_hdr.size() != (size_t)n_quads * 4 || X.quad_round_ptr.size() != (size_t)n_quads + 1) return fail("narrow pack tables");
        if (X.cs_entry.size() != nl || X.cs_slot.size() != nl) return fail("staging tables");
        if (N.hop_root.size() != n_rec) return fail("hop_root size");
        for (uint32_t p = 0; p < Pn; ++p)
            for (uint32_t q = N.pack_hop_ptr[p]; q < N.pack_hop_ptr[p + 1]; ++q) {
                const uint16_t r = N.hop_root[q];
                if (r == NO_ROOT) continue;
                if (q == N.pack_hop_ptr[p] || r >= N.hop_node_off[q + 1] - N.hop_node_off[q]) return fail("hop_root out of range");
                if (L.res.ok) return fail("resident sweeps flagged for a staggered pack");
            }
        if (X.cs_ptr.empty() || !mono_to(X.cs_ptr, 0, nl, "cs_ptr") || !mono_to(X.quad_round_ptr, 0, X.cs_ptr.size() - 1, "quad_round_ptr")) return fail("staging pointers");
        for (size_t r = 0; r + 1 < X.cs_ptr.size(); ++r)
            if (X.cs_ptr[r + 1] - X.cs_ptr[r] > 64u * WPB * 10u) return fail("staging round larger than a workgroup can hold");
        if (!all_below(X.cs_entry, L.n_layers, "cs_entry") || !all_below(X.cs_slot, (uint64_t)WPB * X.stage_cap, "cs_slot")) return false;
        const size_t G = X.grp_hop_end.size();
        if (X.grp_layer_off.size() != G + 1 || !mono_to(X.pack_group_ptr, 0, G, "pack_group_ptr") || !mono_to(X.grp_layer_off, 0, nl, "grp_layer_off")) return fail("stage groups");
        for (size_t g = 0; g < G; ++g)
            if (X.grp_hop_end[g] > n_rec || X.grp_layer_off[g + 1] - X.grp_layer_off[g] > X.stage_cap) return fail("stage group out of range");
        for (uint32_t p = 0; p < Pn; ++p) {
            const uint32_t s0 = N.hop_node_off[N.pack_hop_ptr[p]], s1 = N.hop_node_off[N.pack_hop_ptr[p + 1]];
            if ((uint64_t)L.narrow_word_off[p] + (s1 - s0) > L.narrow_words_unique.size()) return fail("word offsets out of range");
            for (uint32_t g = X.pack_group_ptr[p]; g < X.pack_group_ptr[p + 1]; ++g)  // a pack's groups end inside the pack, in order
                if (X.grp_hop_end[g] <= N.pack_hop_ptr[p] || X.grp_hop_end[g] > N.pack_hop_ptr[p + 1] || (g > X.pack_group_ptr[p] && X.grp_hop_end[g] < X.grp_hop_end[g - 1]))
                    return fail("stage group outside its pack");
        }
        for (uint32_t w : L.narrow_words_unique)  // children index the W + 2 entries of the LDS frontier arrays
            if ((w & NW_CHILD_MASK) > L.pack_width + 1 || ((w >> NW_CHILD_BITS) & NW_CHILD_MASK) > L.pack_width + 1) return fail("narrow node word: child out of range");
        // the resident headers are functions of the tables above (layout.cpp): recompute and compare
        uint32_t max_slots = 0, max_layers = 0;
        for (uint32_t p = 0; p < Pn; ++p) {
            const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1];
            const uint32_t want[8] = {N.hop_node_off[q0], N.hop_node_off[q1] - N.hop_node_off[q0], N.hop_layer_off[q0], N.hop_layer_off[q1] - N.hop_layer_off[q0], q0,
                                      (q1 - q0) | ((uint32_t)N.pack_steps[p] << 16), L.narrow_word_off[p], 0};
            if (q1 - q0 > 0xFFFFu || std::memcmp(want, &L.res.pack_hdr[(size_t)p * 8], sizeof(want)) != 0) return fail("resident pack header");
            max_slots = std::max(max_slots, want[1]);
            max_layers = std::max(max_layers, want[3]);
            if (L.res.ok && (q1 - q0 > 63 || X.pack_group_ptr[p + 1] - X.pack_group_ptr[p] != 1)) return fail("resident sweeps flagged for a pack they cannot hold");
        }
        if (L.res.max_slots != max_slots || L.res.max_layers != max_layers) return fail("resident sizes");
        for (uint32_t Q = 0; Q < n_quads; ++Q) {
            const uint32_t r0 = X.quad_round_ptr[Q], r1 = X.quad_round_ptr[Q + 1];
            const uint32_t want[4] = {r1 > r0 ? X.cs_ptr[r0] : 0, r1 > r0 ? X.cs_ptr[r0 + 1] - X.cs_ptr[r0] : 0, r1 - r0, 0};
            if (std::memcmp(want, &L.res.quad_hdr[(size_t)Q * 4], sizeof(want)) != 0) return fail("resident workgroup header");
            if (L.res.ok && r1 - r0 != 1) return fail("resident sweeps flagged for a workgroup with several rounds");
        }
    } else if (L.res.ok || !L.narrow_words_unique.empty()) {
        return fail("narrow tables without narrow packs");
    }
    // ---- wide / huge packs: children and layer fields index arrays of the pack's width (+ 2 sink entries)
    if (L.wide_words.size() != L.n_slots - L.narrow_slots) return fail("wide words");
    {
        size_t w0 = 0;
        int k = 1;
        for (const PackSet* ps : {&L.wide, &L.huge}) {
            const uint32_t maxw = widths[k++];
            // staggered wide packs (format 07): a root below a pack's first hop is a slot of that hop; huge packs have no such table
            if (ps == &L.wide) {
                if (ps->hop_root.size() + 1 != ps->hop_node_off.size() && !(ps->hop_node_off.empty() && ps->hop_root.empty())) return fail("wide hop_root size");
                for (uint32_t p = 0; p < ps->n_packs(); ++p)
                    for (uint32_t q = ps->pack_hop_ptr[p]; q < ps->pack_hop_ptr[p + 1]; ++q) {
                        const uint16_t r = ps->hop_root[q];
                        if (r == NO_ROOT) continue;
                        if (q == ps->pack_hop_ptr[p] || r >= ps->hop_node_off[q + 1] - ps->hop_node_off[q]) return fail("wide hop_root out of range");
                    }
            }
            if (ps->n_packs() == 0) continue;
            const size_t w1 = w0 + (ps->hop_node_off.back() - ps->hop_node_off.front());
            for (size_t i = w0; i < w1; ++i) {
                const uint64_t w = L.wide_words[i], lo = w & WW_CHILD_MASK, hi = (w >> WW_CHILD_BITS) & WW_CHILD_MASK, ly = (w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK;
                if ((lo < WW_TOP && lo >= maxw) || (hi < WW_TOP && hi >= maxw) || ly >= maxw) return fail("wide node word out of range");
            }
            w0 = w1;
        }
    }
    return true;
}
}  // namespace

int bddmma_save(const bddmma_solver* s, const char* path)
{
    return guarded(s, [&](SolverBase* b) {
        if (!path) return BDDMMA_ERR_INVALID_ARGUMENT;
        const size_t R = b->precision == BDDMMA_F64 ? 8 : 4;
        HostLayout H;
        int rc = b->download_layout(H);
        if (rc) return rc;
        std::vector<char> lo(b->n_layers * R), hi(b->n_layers * R), mm(b->n_layers * R), delta(2 * b->n_vars * R);
        if ((rc = b->get_solver_costs(lo.data(), hi.data(), mm.data(), 0))) return rc;
        if ((rc = b->get_delta(delta.data(), 0))) return rc;
        FileCloser fc{std::fopen(path, "wb")};
        FILE* f = fc.f;
        if (!f) { b->err = std::string("cannot open ") + path; return BDDMMA_ERR_IO; }
        bool ok = true;
        auto w = [&](const void* p, size_t n) { ok = ok && (n == 0 || std::fwrite(p, 1, n, f) == n); };
        uint64_t n_arrays = 0;
        visit_layout_arrays(H, [&](int, auto&) { ++n_arrays; });
        const uint64_t hdr[4] = {(uint64_t)b->precision, n_arrays, sizeof(LayoutScalars), sizeof(bddmma_options)};
        const LayoutScalars sc = layout_scalars(H);
        w(kMagic, 8); w(hdr, sizeof(hdr)); w(&sc, sizeof(sc)); w(&b->saved_opts, sizeof(bddmma_options));
        Checksum cs;
        cs.add(&sc, sizeof(sc));
        cs.add(&b->saved_opts, sizeof(bddmma_options));
        visit_layout_arrays(H, [&](int id, auto& vec) {
            const uint64_t rec[3] = {(uint64_t)id, sizeof(vec[0]), vec.size()};
            w(rec, sizeof(rec));
            w(vec.data(), vec.size() * sizeof(vec[0]));
            cs.add(rec, sizeof(rec));
            cs.add(vec.data(), vec.size() * sizeof(vec[0]));
        });
        const uint64_t sum = cs.value();
        w(&sum, sizeof(sum));
        w(lo.data(), lo.size()); w(hi.data(), hi.size()); w(mm.data(), mm.size()); w(delta.data(), delta.size());
        fc.f = nullptr;
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) { b->err = std::string("write failed: ") + path; return BDDMMA_ERR_IO; }
        return BDDMMA_OK;
    });
}

int bddmma_load(bddmma_solver** out, int device, const char* path)
{
    if (!out || !path) return BDDMMA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    FileCloser fc{std::fopen(path, "rb")};
    FILE* f = fc.f;
    if (!f) { g_err = std::string("cannot open ") + path; return BDDMMA_ERR_IO; }
    SolverBase* impl = nullptr;
    try {
        bool ok = true;
        auto r = [&](void* p, size_t n) { ok = ok && (n == 0 || std::fread(p, 1, n, f) == n); };
        if (std::fseek(f, 0, SEEK_END) != 0) { g_err = "cannot seek"; return BDDMMA_ERR_IO; }
        const uint64_t file_size = (uint64_t)std::ftell(f);
        std::rewind(f);
        char magic[8];
        uint64_t hdr[4] = {0, 0, 0, 0};
        r(magic, 8); r(hdr, sizeof(hdr));
        if (!ok || std::memcmp(magic, kMagic, 8) != 0 || hdr[0] > 1 || hdr[2] != sizeof(LayoutScalars) || hdr[3] != sizeof(bddmma_options) ||
            hdr[1] > 1000) {
            g_err = std::string("not a bdd_mma checkpoint (or one of another version): ") + path;
            return BDDMMA_ERR_IO;
        }
        LayoutScalars sc{};
        bddmma_options opts{};
        r(&sc, sizeof(sc)); r(&opts, sizeof(opts));
        HostLayout H;
        set_layout_scalars(H, sc);
        bool bad = false;
        Checksum cs;
        cs.add(&sc, sizeof(sc));
        cs.add(&opts, sizeof(opts));
        for (uint64_t a = 0; a < hdr[1] && ok && !bad; ++a) {
            uint64_t rec[3] = {0, 0, 0};
            r(rec, sizeof(rec));
            if (!ok) break;
            cs.add(rec, sizeof(rec));
            // a count the rest of the file cannot hold means a truncated or corrupt record: refuse before allocating
            if (rec[1] == 0 || rec[1] > 8 || rec[2] > (file_size - (uint64_t)std::ftell(f)) / rec[1]) { bad = true; break; }
            bool found = false;
            visit_layout_arrays(H, [&](int id, auto& vec) {
                if ((uint64_t)id != rec[0] || found) return;
                found = true;
                if (sizeof(vec[0]) != rec[1]) { bad = true; return; }
                vec.resize(rec[2]);
                r(vec.data(), rec[2] * rec[1]);
                if (ok) cs.add(vec.data(), rec[2] * rec[1]);
            });
            if (!found) bad = true;
        }
        std::string why;
        if (ok && !bad) {
            uint64_t sum = 0;
            r(&sum, sizeof(sum));
            if (ok && sum != cs.value()) { bad = true; why = "layout checksum mismatch"; }
        }
        if (!ok || bad || !layout_plausible(H, why)) {
            g_err = "corrupt or truncated checkpoint" + (why.empty() ? std::string() : " (" + why + ")");
            return BDDMMA_ERR_IO;
        }
        {   // the cost section must be complete before a device is touched
            const uint64_t Rb = hdr[0] == (uint64_t)BDDMMA_F64 ? 8 : 4, need = (3 * H.n_layers + 2 * H.n_vars) * Rb;
            if (file_size - (uint64_t)std::ftell(f) < need) { g_err = "corrupt or truncated checkpoint (cost section incomplete)"; return BDDMMA_ERR_IO; }
        }
        int rc = create_solver(&impl, (int)hdr[0], device, H, &opts, g_err);
        if (rc) return rc;
        impl->n_packs_narrow = H.narrow.n_packs();
        impl->n_packs_wide = H.wide.n_packs() + H.huge.n_packs();
        const size_t R = impl->precision == BDDMMA_F64 ? 8 : 4;
        std::vector<char> lo(impl->n_layers * R), hi(impl->n_layers * R), mm(impl->n_layers * R), delta(2 * impl->n_vars * R);
        r(lo.data(), lo.size()); r(hi.data(), hi.size()); r(mm.data(), mm.size()); r(delta.data(), delta.size());
        if (!ok) { g_err = "truncated checkpoint"; delete impl; return BDDMMA_ERR_IO; }
        rc = impl->set_solver_costs(lo.data(), hi.data(), mm.data(), 0);
        if (rc == BDDMMA_OK) rc = impl->set_delta(delta.data(), 0);
        if (rc) { g_err = impl->err; delete impl; return rc; }
        *out = new bddmma_solver{impl};
        return BDDMMA_OK;
    } catch (const std::exception& e) {
        delete impl;
        g_err = e.what();
        return BDDMMA_ERR_IO;
    }
}

int bddmma_synchronize(bddmma_solver* s) { return guarded(s, [&](SolverBase* b) { return b->synchronize(); }); }
int bddmma_set_profiling(bddmma_solver* s, int on) { return guarded(s, [&](SolverBase* b) { return b->set_profiling(on); }); }
int bddmma_get_profile(bddmma_solver* s, bddmma_profile* out)
{
    return guarded(s, [&](SolverBase* b) { return out ? b->get_profile(out) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_time_kernel(bddmma_solver* s, int kind, uint64_t reps, double* ms)
{
    return guarded(s, [&](SolverBase* b) { return ms ? b->time_kernel(kind, reps, ms) : BDDMMA_ERR_INVALID_ARGUMENT; });
}
int bddmma_time_iterations(bddmma_solver* s, double omega, uint64_t n, double* ms)
{
    return guarded(s, [&](SolverBase* b) { return ms ? b->time_iterations(omega, n, ms) : BDDMMA_ERR_INVALID_ARGUMENT; });
}

// ---- host-only debug ABI (no GPU needed): lets CPU tests inspect the device layout ---------------
struct bddmma_layout {
    HostLayout L;
};
int bddmma_layout_create(bddmma_layout** out, const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                         const bddmma_options* opts)
{
    return bddmma_layout_create_for_chip(out, instr, delims, n_bdds, opts, 4, 0, 0);
}
int bddmma_layout_create_for_chip(bddmma_layout** out, const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                                  const bddmma_options* opts, int real_size, uint32_t n_cus, uint32_t lds_bytes_per_cu)
{
    if (!out || (real_size != 4 && real_size != 8)) return BDDMMA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
        auto* l = new bddmma_layout();
        ChipInfo chip;
        if (n_cus) chip.n_cus = n_cus;
        if (lds_bytes_per_cu) chip.lds_bytes = lds_bytes_per_cu;
        int rc = build_layout(instr, delims, n_bdds, opts, l->L, g_err, true, (uint32_t)real_size, chip);
        if (rc) { delete l; return rc; }
        *out = l;
        return BDDMMA_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return BDDMMA_ERR_DEVICE;
    }
}
void bddmma_layout_destroy(bddmma_layout* l) { delete l; }
// what: 0 n_slots, 1 narrow_slots, 2 n_layers, 3 narrow packs, 4 wide packs, 5 n_hops, 6 n_vars,
//       7 narrow (pack,hop) records, 8 wide (pack,hop) records, 9 bins, 10 vars per bin, 11 stage groups,
//       12 narrow layers, 13 stage cap, 14 waves per block, 15 (quad, round) records, 16 pack width,
//       17 huge packs, 18 huge (pack,hop) records, 19 huge pack width, 20 distinct narrow words stored on the device,
//       21 entries ordered by (variable, bdd), 22 resident sweeps possible, 23 / 24 largest narrow pack in slots / layers
uint64_t bddmma_layout_size(const bddmma_layout* l, int what)
{
    const HostLayout& L = l->L;
    switch (what) {
        case 0: return L.n_slots;
        case 1: return L.narrow_slots;
        case 2: return L.n_layers;
        case 3: return L.narrow.n_packs();
        case 4: return L.wide.n_packs();
        case 5: return L.n_hops;
        case 6: return L.n_vars;
        case 7: return L.narrow.hop_node_off.empty() ? 0 : L.narrow.hop_node_off.size() - 1;
        case 8: return L.wide.hop_node_off.empty() ? 0 : L.wide.hop_node_off.size() - 1;
        case 9: return L.ex.n_bins;
        case 10: return L.ex.vars_per_bin;
        case 11: return L.ex.grp_hop_end.size();
        case 12: return L.ex.grp_layer_off.empty() ? 0 : L.ex.grp_layer_off.back();
        case 13: return L.ex.stage_cap;
        case 14: return L.ex.waves_per_block;
        case 16: return L.pack_width;
        case 17: return L.huge.n_packs();
        case 18: return L.huge.hop_node_off.empty() ? 0 : L.huge.hop_node_off.size() - 1;
        case 19: return L.huge_pack_width;
        case 20: return L.narrow_words_unique.size();
        case 15: return L.ex.cs_ptr.empty() ? 0 : L.ex.cs_ptr.size() - 1;
        case 21: return L.ex.entry_by_var ? 1 : 0;
        case 22: return L.res.ok ? 1 : 0;
        case 23: return L.res.max_slots;
        case 24: return L.res.max_layers;
        default: return 0;
    }
}
// which: 0 narrow_words(u32) 1 wide_words(u64) 2 slot_to_instr(u64) 3 layer_var(i32) 4 layer_bdd(i32)
//        5/6/7/8 narrow pack_hop_ptr/hop_node_off/hop_layer_off(u32)/pack_steps(u8)   9/10/11/12 wide ...
//        13 var_ptr(u32) 14 var_layers(u32) 15 bdd_root_slot(u32)
//        16 bin_ptr(u32) 17 evar(u32) 18 lpos(u32) 19 vpos(u32) 20 pack_group_ptr 21 grp_layer_off 22 grp_hop_end
//        23 quad_round_ptr 24 cs_ptr 25 cs_entry(u32) 26 cs_slot(u16)   27/28/29/30 huge pack_hop_ptr/hop_node_off/hop_layer_off/pack_steps
//        31 narrow_words_unique(u32) 32 narrow_word_off(u32, per narrow pack)
//        35 pack_hdr(u32 x 8 per narrow pack) 36 quad_hdr(u32 x 4 per quad)
int bddmma_layout_copy(const bddmma_layout* l, int which, void* out)
{
    const HostLayout& L = l->L;
    auto cp = [&](const auto& v) {
        if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(v[0]));
        return BDDMMA_OK;
    };
    switch (which) {
        case 0: return cp(L.narrow_words);
        case 1: return cp(L.wide_words);
        case 2: return cp(L.slot_to_instr);
        case 3: return cp(L.layer_var);
        case 4: return cp(L.layer_bdd);
        case 5: return cp(L.narrow.pack_hop_ptr);
        case 6: return cp(L.narrow.hop_node_off);
        case 7: return cp(L.narrow.hop_layer_off);
        case 8: return cp(L.narrow.pack_steps);
        case 9: return cp(L.wide.pack_hop_ptr);
        case 10: return cp(L.wide.hop_node_off);
        case 11: return cp(L.wide.hop_layer_off);
        case 12: return cp(L.wide.pack_steps);
        case 13: return cp(L.var_ptr);
        case 14: return cp(L.var_layers);
        case 15: return cp(L.bdd_root_slot);
        case 16: return cp(L.ex.bin_ptr);
        case 17: return cp(L.ex.evar);
        case 18: return cp(L.ex.lpos);
        case 19: return cp(L.ex.vpos);
        case 20: return cp(L.ex.pack_group_ptr);
        case 21: return cp(L.ex.grp_layer_off);
        case 22: return cp(L.ex.grp_hop_end);
        case 23: return cp(L.ex.quad_round_ptr);
        case 24: return cp(L.ex.cs_ptr);
        case 25: return cp(L.ex.cs_entry);
        case 26: return cp(L.ex.cs_slot);
        case 27: return cp(L.huge.pack_hop_ptr);
        case 28: return cp(L.huge.hop_node_off);
        case 29: return cp(L.huge.hop_layer_off);
        case 30: return cp(L.huge.pack_steps);
        case 31: return cp(L.narrow_words_unique);
        case 32: return cp(L.narrow_word_off);
        case 35: return cp(L.res.pack_hdr);
        case 36: return cp(L.res.quad_hdr);
        default: return BDDMMA_ERR_INVALID_ARGUMENT;
    }
}
int bddmma_layout_res2_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off)
{
    if (!l || !info || (real_size != 4 && real_size != 8)) return BDDMMA_ERR_INVALID_ARGUMENT;
    const HostLayout& L = l->L;
    const uint32_t ns = res2_slot_capacity((uint32_t)real_size, L.res.max_slots), nl = res2_layer_capacity((uint32_t)real_size, L.res.max_layers);  // as SolverT::init
    Res2Records R;
    build_res2_records(L, (uint32_t)real_size, ns, nl, R);
    info[0] = R.ok ? 1u : 0u; info[1] = (uint32_t)R.rec.size(); info[2] = ns; info[3] = nl; info[4] = R.max_hops;
    if (words && !R.rec.empty()) std::memcpy(words, R.rec.data(), R.rec.size() * sizeof(uint32_t));
    if (rec_off && !R.rec_off.empty()) std::memcpy(rec_off, R.rec_off.data(), R.rec_off.size() * sizeof(uint32_t));
    return BDDMMA_OK;
}

int bddmma_layout_seg_exchange(const bddmma_layout* l, int threads, int real_size, uint32_t* info, uint32_t* bin, uint16_t* perm, uint32_t* thr)
{
    if (!l || !info || (real_size != 4 && real_size != 8) || threads <= 0) return BDDMMA_ERR_INVALID_ARGUMENT;
    SegExchange S;
    build_seg_exchange(l->L, (uint32_t)threads, (uint32_t)real_size, S);
    info[0] = S.ok ? 1u : 0u; info[1] = (uint32_t)S.bin.size(); info[2] = (uint32_t)S.perm.size(); info[3] = (uint32_t)S.thr.size();
    info[4] = S.max_entries; info[5] = S.max_slots; info[6] = S.max_groups;
    if (bin && !S.bin.empty()) std::memcpy(bin, S.bin.data(), S.bin.size() * sizeof(uint32_t));
    if (perm && !S.perm.empty()) std::memcpy(perm, S.perm.data(), S.perm.size() * sizeof(uint16_t));
    if (thr && !S.thr.empty()) std::memcpy(thr, S.thr.data(), S.thr.size() * sizeof(uint32_t));
    return BDDMMA_OK;
}

int bddmma_layout_stream_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off)
{
    if (!l || !info || (real_size != 4 && real_size != 8)) return BDDMMA_ERR_INVALID_ARGUMENT;
    StreamRecords R;
    build_stream_records(l->L, (uint32_t)real_size, R);
    info[0] = R.ok ? 1u : 0u; info[1] = (uint32_t)R.rec.size();
    if (words && !R.rec.empty()) std::memcpy(words, R.rec.data(), R.rec.size() * sizeof(uint32_t));
    if (rec_off && !R.rec_off.empty()) std::memcpy(rec_off, R.rec_off.data(), R.rec_off.size() * sizeof(uint32_t));
    return BDDMMA_OK;
}

int bddmma_layout_layer_records(const bddmma_layout* l, int real_size, uint32_t* info, uint32_t* words, uint32_t* rec_off)
{
    if (!l || !info || (real_size != 4 && real_size != 8)) return BDDMMA_ERR_INVALID_ARGUMENT;
    LayerRecords R;
    build_layer_records(l->L, (uint32_t)real_size, R);
    info[0] = R.ok ? 1u : 0u; info[1] = (uint32_t)R.rec.size();
    if (words && !R.rec.empty()) std::memcpy(words, R.rec.data(), R.rec.size() * sizeof(uint32_t));
    if (rec_off && !R.rec_off.empty()) std::memcpy(rec_off, R.rec_off.data(), R.rec_off.size() * sizeof(uint32_t));
    return BDDMMA_OK;
}

}  // extern "C"
