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
        static const bool timing = std::getenv("BDDMMA_LAYOUT_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[create] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        HostLayout L;
        int rc = build_layout(instr, bdd_delims, n_bdds, opts, L, g_err, false, precision == BDDMMA_F64 ? 8 : 4);
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
int bddmma_precision(const bddmma_solver* s) { return s && s->impl ? s->impl->precision : -1; }
int bddmma_device(const bddmma_solver* s) { return s && s->impl ? s->impl->device : -1; }
uint64_t bddmma_device_bytes(const bddmma_solver* s) { return s && s->impl ? s->impl->dev_bytes : 0; }

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
    return guarded(s, [&](SolverBase* b) {
        for (uint64_t i = 0; i < n; ++i) {
            int rc = b->iteration(omega);
            if (rc) return rc;
        }
        return BDDMMA_OK;
    });
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

// BDDMMA_SEQUENTIAL_RUN_SOLVER=1 keeps the reference's literal loop (iteration, lower_bound, tests on the host) for the plain solver
// too: the twin the device-resident loop is tested against.
static bool sequential_run_solver()
{
    const char* e = std::getenv("BDDMMA_SEQUENTIAL_RUN_SOLVER");
    return e && e[0] == '1';
}

// run_solver, include/run_solver_util.h:10-77
int bddmma_run_solver(bddmma_solver* s, bddmma_lbfgs* lbfgs, uint64_t max_iter, double tolerance,
                      double improvement_slope, double time_limit, int verbose, bddmma_run_result* res)
{
    return guarded(s, [&](SolverBase* b) {
        if (improvement_slope < 0.0 || improvement_slope >= 1.0 || time_limit < 0.0 || tolerance < 0.0) {
            b->err = "run_solver: invalid termination criteria";  // asserts at run_solver_util.h:13-15
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        // plain MMA: the loop runs with its termination tests on the device (no host round trip per iteration); the L-BFGS wrapper
        // decides its steps on the host anyway and keeps the sequential loop below
        if (!lbfgs && !sequential_run_solver()) return b->run_plain(max_iter, tolerance, improvement_slope, time_limit, verbose, res);
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
static const char kMagic[8] = {'B', 'D', 'D', 'M', 'M', 'A', '0', '4'};  // 04: narrow node words carry the layer index (layout.hpp)

namespace {
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) std::fclose(f); }
};
// Consistency of a layout read from a file: the offsets every kernel trusts must be monotone and end where the sizes say.
bool layout_plausible(const HostLayout& L, std::string& why)
{
    auto mono_to = [&](const auto& v, uint64_t first, uint64_t last, const char* name) {
        if (v.empty()) { why = std::string(name) + " is empty"; return false; }
        if (v.front() != first || v.back() != last) { why = std::string(name) + " does not span its range"; return false; }
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i] < v[i - 1]) { why = std::string(name) + " is not monotone"; return false; }
        return true;
    };
    if (L.n_layers == 0 || L.n_vars == 0 || L.n_bdds == 0) { why = "empty layout"; return false; }
    if (L.layer_var.size() != L.n_layers || L.layer_bdd.size() != L.n_layers || L.var_layers.size() != L.n_layers ||
        L.ex.lpos.size() != L.n_layers || L.ex.evar.size() != L.n_layers || L.ex.bvar.size() != L.n_layers || L.ex.vpos.size() != L.n_layers ||
        L.num_bdds_per_var.size() != L.n_vars || L.var_ptr.size() != L.n_vars + 1 || L.bdd_root_slot.size() != L.n_bdds ||
        L.ex.bin_ptr.size() != (size_t)L.ex.n_bins + 1) { why = "array sizes do not match the header"; return false; }
    if (!mono_to(L.var_ptr, 0, L.n_layers, "var_ptr") || !mono_to(L.ex.bin_ptr, 0, L.n_layers, "bin_ptr")) return false;
    uint64_t slots = 0, layers = 0;
    for (const PackSet* ps : {&L.narrow, &L.wide, &L.huge}) {
        const uint32_t P = ps->n_packs();
        if (P == 0) continue;
        if (ps->pack_steps.size() != P) { why = "pack_steps size"; return false; }
        if (!mono_to(ps->pack_hop_ptr, 0, ps->hop_node_off.size() - 1, "pack_hop_ptr")) return false;
        if (ps->hop_layer_off.size() != ps->hop_node_off.size()) { why = "hop offset sizes"; return false; }
        if (!mono_to(ps->hop_node_off, slots, ps->hop_node_off.back(), "hop_node_off") ||
            !mono_to(ps->hop_layer_off, layers, ps->hop_layer_off.back(), "hop_layer_off")) return false;
        slots = ps->hop_node_off.back();
        layers = ps->hop_layer_off.back();
    }
    if (slots != L.n_slots || layers != L.n_layers) { why = "pack offsets do not cover the slots / layers"; return false; }
    for (uint32_t e : L.ex.lpos) if (e >= L.n_layers) { why = "lpos out of range"; return false; }
    for (int32_t v : L.layer_var) if (v < 0 || (uint64_t)v >= L.n_vars) { why = "layer variable out of range"; return false; }
    const uint32_t Pn = L.narrow.n_packs();
    if (Pn) {
        if (L.narrow_word_off.size() != Pn || L.ex.pack_group_ptr.size() != (size_t)Pn + 1 || L.res.pack_hdr.size() != (size_t)Pn * 8) { why = "narrow pack tables"; return false; }
        const uint32_t nl = L.narrow.hop_layer_off.back();
        if (L.ex.cs_entry.size() != nl || L.ex.cs_slot.size() != nl) { why = "staging tables"; return false; }
        if (!mono_to(L.ex.cs_ptr, 0, nl, "cs_ptr") || !mono_to(L.ex.quad_round_ptr, 0, L.ex.cs_ptr.size() - 1, "quad_round_ptr")) return false;
        for (uint32_t e : L.ex.cs_entry) if (e >= L.n_layers) { why = "cs_entry out of range"; return false; }
        for (uint32_t p = 0; p < Pn; ++p) {
            const uint32_t s0 = L.narrow.hop_node_off[L.narrow.pack_hop_ptr[p]], s1 = L.narrow.hop_node_off[L.narrow.pack_hop_ptr[p + 1]];
            if ((uint64_t)L.narrow_word_off[p] + (s1 - s0) > L.narrow_words_unique.size()) { why = "word offsets out of range"; return false; }
        }
    }
    if (L.wide_words.size() != L.n_slots - L.narrow_slots) { why = "wide words"; return false; }
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
        visit_layout_arrays(H, [&](int id, auto& vec) {
            const uint64_t rec[3] = {(uint64_t)id, sizeof(vec[0]), vec.size()};
            w(rec, sizeof(rec));
            w(vec.data(), vec.size() * sizeof(vec[0]));
        });
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
        for (uint64_t a = 0; a < hdr[1] && ok && !bad; ++a) {
            uint64_t rec[3] = {0, 0, 0};
            r(rec, sizeof(rec));
            if (!ok) break;
            // a count the rest of the file cannot hold means a truncated or corrupt record: refuse before allocating
            if (rec[1] == 0 || rec[1] > 8 || rec[2] > (file_size - (uint64_t)std::ftell(f)) / rec[1]) { bad = true; break; }
            bool found = false;
            visit_layout_arrays(H, [&](int id, auto& vec) {
                if ((uint64_t)id != rec[0] || found) return;
                found = true;
                if (sizeof(vec[0]) != rec[1]) { bad = true; return; }
                vec.resize(rec[2]);
                r(vec.data(), rec[2] * rec[1]);
            });
            if (!found) bad = true;
        }
        std::string why;
        if (!ok || bad || !layout_plausible(H, why)) {
            g_err = "corrupt or truncated checkpoint" + (why.empty() ? std::string() : " (" + why + ")");
            return BDDMMA_ERR_IO;
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
    if (!out) return BDDMMA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
        auto* l = new bddmma_layout();
        int rc = build_layout(instr, delims, n_bdds, opts, l->L, g_err, true);
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

}  // extern "C"
