// capi.cpp — extern "C" entry points of include/bdd_mma.h (everything except the L-BFGS ones,
// which live in lbfgs.hip).  No exception crosses this boundary.
#include <chrono>
#include <cmath>
#include <cstdio>
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
        HostLayout L;
        int rc = build_layout(instr, bdd_delims, n_bdds, opts, L, g_err, false, precision == BDDMMA_F64 ? 8 : 4);
        if (rc) return rc;
        SolverBase* impl = nullptr;
        rc = create_solver(&impl, precision, device, L, opts, g_err);
        if (rc) return rc;
        impl->n_packs_narrow = L.narrow.n_packs();
        impl->n_packs_wide = L.wide.n_packs() + L.huge.n_packs();
        impl->saved_instr.assign(instr, instr + bdd_delims[n_bdds]);
        impl->saved_delims.assign(bdd_delims, bdd_delims + n_bdds + 1);
        if (opts) impl->saved_opts = *opts;
        if (costs_hi && n_costs) {
            rc = impl->update_costs(nullptr, 0, costs_hi, n_costs, BDDMMA_F64, 0);
            if (rc) {
                g_err = impl->err;
                delete impl;
                return rc;
            }
        }
        *out = new bddmma_solver{impl};
        return BDDMMA_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return BDDMMA_ERR_DEVICE;
    }
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

// run_solver, include/run_solver_util.h:10-77
int bddmma_run_solver(bddmma_solver* s, bddmma_lbfgs* lbfgs, uint64_t max_iter, double tolerance,
                      double improvement_slope, double time_limit, int verbose, bddmma_run_result* res)
{
    return guarded(s, [&](SolverBase* b) {
        if (improvement_slope < 0.0 || improvement_slope >= 1.0 || time_limit < 0.0 || tolerance < 0.0) {
            b->err = "run_solver: invalid termination criteria";  // asserts at run_solver_util.h:13-15
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
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

// ---- checkpoint: the layout is a pure function of (collection, options), so the file holds the
// collection + options + the mutable state (costs, deferred mm, deferred delta).  Mirrors what the
// reference archives (bdd_cuda_base.cu:1486-1550; cost_from_root/terminal are not saved there either).
static const char kMagic[8] = {'B', 'D', 'D', 'M', 'M', 'A', '0', '2'};

int bddmma_save(const bddmma_solver* s, const char* path)
{
    return guarded(s, [&](SolverBase* b) {
        if (!path) return BDDMMA_ERR_INVALID_ARGUMENT;
        const size_t R = b->precision == BDDMMA_F64 ? 8 : 4;
        std::vector<char> lo(b->n_layers * R), hi(b->n_layers * R), mm(b->n_layers * R), delta(2 * b->n_vars * R);
        int rc = b->get_solver_costs(lo.data(), hi.data(), mm.data(), 0);
        if (rc) return rc;
        if ((rc = b->get_delta(delta.data(), 0))) return rc;
        FILE* f = std::fopen(path, "wb");
        if (!f) { b->err = std::string("cannot open ") + path; return BDDMMA_ERR_IO; }
        bool ok = true;
        auto w = [&](const void* p, size_t n) { ok = ok && (n == 0 || std::fwrite(p, 1, n, f) == n); };
        const uint64_t hdr[4] = {(uint64_t)b->precision, b->n_bdds, (uint64_t)b->saved_instr.size(), b->n_layers};
        w(kMagic, 8); w(hdr, sizeof(hdr)); w(&b->saved_opts, sizeof(bddmma_options));
        w(b->saved_delims.data(), b->saved_delims.size() * 8);
        w(b->saved_instr.data(), b->saved_instr.size() * sizeof(bddmma_instruction));
        w(lo.data(), lo.size()); w(hi.data(), hi.size()); w(mm.data(), mm.size()); w(delta.data(), delta.size());
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) { b->err = std::string("write failed: ") + path; return BDDMMA_ERR_IO; }
        return BDDMMA_OK;
    });
}

int bddmma_load(bddmma_solver** out, int device, const char* path)
{
    if (!out || !path) return BDDMMA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) { g_err = std::string("cannot open ") + path; return BDDMMA_ERR_IO; }
    bool ok = true;
    auto r = [&](void* p, size_t n) { ok = ok && (n == 0 || std::fread(p, 1, n, f) == n); };
    char magic[8];
    uint64_t hdr[4] = {0, 0, 0, 0};
    bddmma_options opts{};
    r(magic, 8); r(hdr, sizeof(hdr)); r(&opts, sizeof(opts));
    if (!ok || std::memcmp(magic, kMagic, 8) != 0 || hdr[1] == 0 || hdr[1] > (1ull << 40) || hdr[2] > (1ull << 40)) {
        std::fclose(f);
        g_err = std::string("not a bdd_mma checkpoint: ") + path;
        return BDDMMA_ERR_IO;
    }
    int rc = BDDMMA_OK;
    try {
        std::vector<uint64_t> delims(hdr[1] + 1);
        std::vector<bddmma_instruction> instr(hdr[2]);
        r(delims.data(), delims.size() * 8);
        r(instr.data(), instr.size() * sizeof(bddmma_instruction));
        if (!ok) { std::fclose(f); g_err = "truncated checkpoint"; return BDDMMA_ERR_IO; }
        rc = bddmma_create(out, (int)hdr[0], device, instr.data(), delims.data(), hdr[1], nullptr, 0, &opts);
        if (rc) { std::fclose(f); return rc; }
        SolverBase* b = (*out)->impl;
        const size_t R = b->precision == BDDMMA_F64 ? 8 : 4;
        if (b->n_layers != hdr[3]) ok = false;
        std::vector<char> lo(b->n_layers * R), hi(b->n_layers * R), mm(b->n_layers * R), delta(2 * b->n_vars * R);
        r(lo.data(), lo.size()); r(hi.data(), hi.size()); r(mm.data(), mm.size()); r(delta.data(), delta.size());
        std::fclose(f);
        if (!ok) { g_err = "truncated checkpoint"; bddmma_destroy(*out); *out = nullptr; return BDDMMA_ERR_IO; }
        rc = b->set_solver_costs(lo.data(), hi.data(), mm.data(), 0);
        if (rc == BDDMMA_OK) rc = b->set_delta(delta.data(), 0);
        if (rc) { g_err = b->err; bddmma_destroy(*out); *out = nullptr; }
        return rc;
    } catch (const std::exception& e) {
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
