// C++ tests of the drop-in classes (bdd_hip_parallel_mma<REAL>, bdd_hip_lbfgs_mma<REAL>) and of the C++ bdd_solver
// driver, in the style of the reference's own tests:
//   test/test_bdd_cuda_base.cpp:8-116, test/test_bdd_cuda_min_marginals.cpp, test/test_cuda_parallel_mma.cu:197-247,
//   test/test_bdd_bipartite_matching_problem.cpp:8-59, test/test_loose_covering_problem.cpp:8-88.
// Needs a GPU; run by tests/test_gpu_cpp.py.  Exit code 0 = all passed.
#include <cmath>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../bdd_amd/csrc/bdd_hip_parallel_mma.hpp"
#include "../../bdd_amd/csrc/host/bdd_solver.hpp"
#include "../../bdd_amd/csrc/host/bdd_store.hpp"

using namespace LPMP;
using bddmma_host::bdd_store;

static int failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)
#define CHECK_NEAR(a, b, tol)                                                                              \
    do {                                                                                                   \
        const double a_ = (a), b_ = (b);                                                                   \
        if (!(std::fabs(a_ - b_) <= (tol))) { std::printf("  FAILED %s:%d: %s = %.12g, expected %.12g\n", __FILE__, __LINE__, #a, a_, b_); ++failures; } \
    } while (0)

// n x n assignment: 2n simplex rows over n^2 variables (test_bdd_bipartite_matching_problem.cpp)
static bdd_store matching(size_t n)
{
    bdd_store col;
    for (size_t i = 0; i < n; ++i) {
        std::vector<size_t> row, colv;
        for (size_t j = 0; j < n; ++j) { row.push_back(i * n + j); colv.push_back(j * n + i); }
        col.add_simplex(row);
        col.add_simplex(colv);
    }
    return col;
}

template <typename REAL>
static void test_matching_kats()
{
    const double tol = sizeof(REAL) == 8 ? 1e-9 : 1e-4;
    {   // diagonal costs -2, others -1: LB -6 from the start (test_bdd_cuda_base.cpp:114)
        std::vector<double> c(9, -1.0);
        c[0] = c[4] = c[8] = -2.0;
        bdd_hip_parallel_mma<REAL> s(matching(3), c);
        CHECK(s.nr_variables() == 9 && s.nr_bdds() == 6 && s.nr_layers() == 18);
        CHECK(s.nr_bdds(4) == 2);
        CHECK_NEAR(s.lower_bound(), -6.0, tol);
        for (int i = 0; i < 5; ++i) s.iteration();
        CHECK_NEAR(s.lower_bound(), -6.0, tol);
    }
    {   // first column -2: the exact dyadic trajectory of the CPU parallel mma, optimum -4
        std::vector<double> c(9, -1.0);
        c[0] = c[3] = c[6] = -2.0;
        bdd_hip_parallel_mma<REAL> s(matching(3), c);
        CHECK_NEAR(s.lower_bound(), -5.0, tol);
        const double want[] = {-4.78125, -4.3671875, -4.187866210938, -4.103347778320, -4.059029579163};
        for (double w : want) {
            s.iteration();
            CHECK_NEAR(s.lower_bound(), w, sizeof(REAL) == 8 ? 1e-11 : 1e-4);
        }
        for (int i = 0; i < 95; ++i) s.iteration();
        CHECK_NEAR(s.lower_bound(), -4.0, 1e-6);
    }
}

template <typename REAL>
static void test_costs_marginals_solution()
{
    // one simplex over 3 variables, costs (1, 2, 3): min-marginals per variable (test_bdd_cuda_min_marginals.cpp pattern)
    bdd_store col;
    col.add_simplex({0, 1, 2});
    bdd_hip_parallel_mma<REAL> s(col);
    s.set_cost(1.0, 0);
    s.update_costs(std::vector<REAL>{}, std::vector<REAL>{0, 2, 3});
    CHECK_NEAR(s.lower_bound(), 1.0, 1e-6);
    const auto obj = s.get_primal_objective_vector_host();
    CHECK(obj.size() == 3);
    CHECK_NEAR(obj[0], 1, 1e-6); CHECK_NEAR(obj[1], 2, 1e-6); CHECK_NEAR(obj[2], 3, 1e-6);
    const auto mm = s.min_marginals();
    CHECK(mm.size() == 3 && mm[0].size() == 1);
    CHECK_NEAR(mm[0][0][0], 2, 1e-6); CHECK_NEAR(mm[0][0][1], 1, 1e-6);   // x0 = 0 -> best is x1 (2); x0 = 1 -> 1
    CHECK_NEAR(mm[1][0][0], 1, 1e-6); CHECK_NEAR(mm[1][0][1], 2, 1e-6);
    CHECK_NEAR(mm[2][0][0], 1, 1e-6); CHECK_NEAR(mm[2][0][1], 3, 1e-6);
    const auto sol = s.bdds_solution_vec_host();
    CHECK(sol.size() == 3);
    int ones = 0;
    for (char x : sol) ones += x;
    CHECK(ones == 1);
}

// test/test_bdd_cuda_base_sol.cpp:30-86, both blocks: two simplex rows sharing x_4, costs set per variable, bdds_solution() as
// [variable][bdd]; the first block's answer is unique up to the tie x_2 / x_4 in BDD 0, which the argmin rule (`cost_diff > 0 -> 0`,
// bdd_cuda_base.cu:1121-1131) resolves towards the earlier variable
template <typename REAL>
static void test_bdds_solution_reference_kat()
{
    auto build = [](const std::vector<double>& obj) {
        bdd_store col;
        col.add_simplex({0, 1, 2, 3});        // x_1 + x_2 + x_3 + x_4 = 1
        col.add_linear({1, 1, 1}, bddmma_host::ineq_t::eq, 2, {3, 4, 5});  // x_4 + x_5 + x_6 = 2
        bdd_hip_parallel_mma<REAL> s(col);
        for (size_t i = 0; i < obj.size(); ++i) s.set_cost(obj[i], i);
        return s;
    };
    {
        auto s = build({2, 1, 1.5, 2, 2, 3});
        CHECK(s.nr_variables() == 6 && s.nr_bdds() == 2);
        const auto sol = s.bdds_solution();
        CHECK(sol.size() == 6);
        for (size_t i = 0; i < 6; ++i) CHECK(sol[i].size() == (i == 3 ? 2u : 1u));
        CHECK(sol[0][0] == 0); CHECK(sol[1][0] == 1); CHECK(sol[2][0] == 0); CHECK(sol[3][0] == 0);
        CHECK(sol[3][1] == 1); CHECK(sol[4][0] == 1); CHECK(sol[5][0] == 0);
    }
    {
        auto s = build({1, 1, 1, 2, 1, 1});
        const auto sol = s.bdds_solution();
        CHECK(sol.size() == 6);
        for (size_t i = 0; i < 6; ++i) CHECK(sol[i].size() == (i == 3 ? 2u : 1u));
        CHECK(sol[0][0] + sol[1][0] + sol[2][0] + sol[3][0] == 1);
        CHECK(sol[3][1] + sol[4][0] + sol[5][0] == 2);
    }
}

template <typename REAL>
static void test_explicit_mm_and_distribute()
{
    // forward_mm / backward_mm with a caller-held delta, then distribute_delta keeps the sum of costs (test_cuda_parallel_mma.cu:13-103)
    std::vector<double> c(9, -1.0);
    c[0] = c[3] = c[6] = -2.0;
    bdd_hip_parallel_mma<REAL> s(matching(3), c);
    std::vector<REAL> delta(2 * s.nr_variables(), REAL(0));
    const double lb0 = s.lower_bound();
    for (int it = 0; it < 10; ++it) {
        s.forward_mm(REAL(0.5), delta);
        for (size_t v = 0; v < s.nr_variables(); ++v) { delta[2 * v] /= s.nr_bdds(v); delta[2 * v + 1] /= s.nr_bdds(v); }
        s.backward_mm(REAL(0.5), delta);
        for (size_t v = 0; v < s.nr_variables(); ++v) { delta[2 * v] /= s.nr_bdds(v); delta[2 * v + 1] /= s.nr_bdds(v); }
    }
    CHECK(s.lower_bound() >= lb0 - 1e-6);
    s.distribute_delta();
    const auto obj = s.get_primal_objective_vector_host();
    for (size_t v = 0; v < 9; ++v) CHECK_NEAR(obj[v], c[v], sizeof(REAL) == 8 ? 1e-9 : 1e-4);
}

// the thrust::device_vector overloads of the reference (test/test_cuda_parallel_mma.cu:71-99 passes delta_lo_hi as a device vector):
// raw device pointers here; same protocol on a host vector and on a device buffer, results must agree
template <typename REAL>
static void test_device_vectors()
{
    std::vector<double> c(9, -1.0);
    c[0] = c[3] = c[6] = -2.0;
    bdd_hip_parallel_mma<REAL> a(matching(3), c), b(matching(3), c);
    const size_t n = 2 * a.nr_variables();
    std::vector<REAL> ha(n, REAL(0)), hb(n);
    REAL* dev = nullptr;
    CHECK(hipMalloc((void**)&dev, n * sizeof(REAL)) == hipSuccess);
    CHECK(hipMemset(dev, 0, n * sizeof(REAL)) == hipSuccess);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    for (int it = 0; it < 5; ++it) {
        a.forward_mm(REAL(0.5), ha);
        b.forward_mm(REAL(0.5), dev);
        CHECK(hipMemcpy(hb.data(), dev, n * sizeof(REAL), hipMemcpyDeviceToHost) == hipSuccess);
        for (size_t i = 0; i < n; ++i) CHECK_NEAR(hb[i], ha[i], sizeof(REAL) == 8 ? 1e-12 : 1e-5);
        for (size_t v = 0; v < a.nr_variables(); ++v) { ha[2 * v] /= a.nr_bdds(v); ha[2 * v + 1] /= a.nr_bdds(v); }
        b.normalize_delta(dev);
        a.backward_mm(REAL(0.5), ha);
        b.backward_mm(REAL(0.5), dev);
        CHECK(hipMemcpy(hb.data(), dev, n * sizeof(REAL), hipMemcpyDeviceToHost) == hipSuccess);
        for (size_t i = 0; i < n; ++i) CHECK_NEAR(hb[i], ha[i], sizeof(REAL) == 8 ? 1e-12 : 1e-5);
        for (size_t v = 0; v < a.nr_variables(); ++v) { ha[2 * v] /= a.nr_bdds(v); ha[2 * v + 1] /= a.nr_bdds(v); }
        b.normalize_delta(dev);
    }
    CHECK_NEAR(a.lower_bound(), b.lower_bound(), 1e-6);
    // L-BFGS support on device vectors: net_solver_costs -> make_dual_feasible -> gradient_step
    REAL* g = nullptr;
    CHECK(hipMalloc((void**)&g, a.nr_layers() * sizeof(REAL)) == hipSuccess);
    b.net_solver_costs(g);
    b.make_dual_feasible(g);
    const double lb0 = b.lower_bound();
    b.gradient_step(g, 0.0);
    CHECK_NEAR(b.lower_bound(), lb0, 1e-9);
    (void)hipFree(g);
    (void)hipFree(dev);
}

template <typename REAL>
static void test_lbfgs_and_move()
{
    bdd_store col;
    for (auto row : std::vector<std::vector<size_t>>{{0, 1, 3}, {0, 2, 4}, {1, 2, 5}}) col.add_covering(row);
    bdd_hip_parallel_mma<REAL> base(col, std::vector<double>(6, 1.0));
    bdd_hip_parallel_mma<REAL> moved(std::move(base));   // the variant in bdd_solver.h:64-69 needs movability
    bdd_hip_lbfgs_mma<REAL> s(std::move(moved));
    double prev = s.lower_bound();
    for (int i = 0; i < 60; ++i) {
        s.iteration();
        const double lb = s.lower_bound();
        CHECK(lb >= prev - 1e-5);   // lbfgs_impl.h:403
        prev = lb;
    }
    CHECK_NEAR(prev, 1.5, 1e-3);    // test_loose_covering_problem.cpp: LP bound 1.5
}

// the embedding façades of include/bdd_cuda.h / include/bdd_lbfgs_cuda_mma.h
template <typename REAL>
static void test_facades()
{
    bdd_store col;
    for (auto row : std::vector<std::vector<size_t>>{{0, 1, 3}, {0, 2, 4}, {1, 2, 5}}) col.add_covering(row);
    const std::vector<double> costs(6, 1.0);
    bdd_hip<REAL> a(col, costs.begin(), costs.end());
    CHECK(a.nr_variables() == 6);
    for (int i = 0; i < 300; ++i) a.iteration();
    CHECK_NEAR(a.lower_bound(), 1.5, 1e-3);
    CHECK(a.min_marginals().size() == 6);
    const std::vector<char> sol = a.incremental_mm_agreement_rounding(0.1, 1.1, 50, 100);
    CHECK(sol.size() == 6);
    if (sol.size() == 6) {
        CHECK(col.evaluate(0, sol) && col.evaluate(1, sol) && col.evaluate(2, sol));
        int ones = 0;
        for (char x : sol) ones += x;
        CHECK(ones == 2);   // optimum of the loose covering instance
    }
    bdd_hip<REAL> moved(std::move(a));
    moved.backward_run();
    bdd_lbfgs_hip_mma<REAL> l(col, costs.begin(), costs.end(), 5);
    double prev = l.lower_bound();
    for (int i = 0; i < 60; ++i) {
        l.iteration();
        CHECK(l.lower_bound() >= prev - 1e-5);
        prev = l.lower_bound();
    }
    CHECK_NEAR(prev, 1.5, 1e-3);
    const std::vector<double> zero(6, 0.0), two(6, 2.0);
    l.update_costs(zero.begin(), zero.begin(), two.begin(), two.end());   // hi += 2: costs 3 per variable, bound 4.5
    for (int i = 0; i < 300; ++i) l.iteration();
    CHECK_NEAR(l.lower_bound(), 4.5, 3e-3);
    CHECK(l.incremental_mm_agreement_rounding(0.1, 1.1, 50, 100).size() == 6);
}

// The remaining public members of bdd_cuda_base (bdd_cuda_base.h:70,84,106-135,167-170): lower_bound_per_bdd, min_marginals_cuda(get_sorted),
// nr_layers(hop) / nr_bdd_nodes(hop), var_constraint_indices, get / set_solver_costs, load — one check each, host and device forms.
template <typename REAL>
static void test_base_members()
{
    const double tol = sizeof(REAL) == 8 ? 1e-9 : 1e-4;
    std::vector<double> c(9, -1.0);
    c[0] = c[3] = c[6] = -2.0;
    bdd_hip_parallel_mma<REAL> s(matching(3), c);
    for (int i = 0; i < 3; ++i) s.iteration();
    const size_t L = s.nr_layers(), B = s.nr_bdds();
    // per-hop sizes: six simplex BDDs over three variables each -> hop 0: 6 one-node layers, hops 1, 2: 6 two-node layers
    CHECK(s.nr_hops() == 3);
    CHECK(s.nr_layers(0) == 6 && s.nr_layers(1) == 6 && s.nr_layers(2) == 6);
    CHECK(s.nr_bdd_nodes(0) == 6 && s.nr_bdd_nodes(1) == 12 && s.nr_bdd_nodes(2) == 12);
    // (variable, BDD) of every dual variable: every BDD three times, every variable twice, consistent with nr_bdds(var)
    const auto vc = s.var_constraint_indices();
    CHECK(vc.first.size() == L && vc.second.size() == L);
    std::vector<int> per_var(9, 0), per_bdd(B, 0);
    for (size_t l = 0; l < L; ++l) { ++per_var[(size_t)vc.first[l]]; ++per_bdd[(size_t)vc.second[l]]; }
    for (size_t v = 0; v < 9; ++v) CHECK(per_var[v] == 2 && s.nr_bdds(v) == 2 && s.get_num_bdds_per_var()[v] == 2);
    for (size_t b = 0; b < B; ++b) CHECK(per_bdd[b] == 3);
    // lower_bound_per_bdd: host form, device form, and the sum is lower_bound()
    const auto lbh = s.lower_bound_per_bdd_host();
    REAL* dlb = nullptr;
    CHECK(hipMalloc((void**)&dlb, B * sizeof(REAL)) == hipSuccess);
    s.lower_bound_per_bdd(dlb);
    std::vector<REAL> lbd(B);
    CHECK(hipMemcpy(lbd.data(), dlb, B * sizeof(REAL), hipMemcpyDeviceToHost) == hipSuccess);
    double sum = 0;
    for (size_t b = 0; b < B; ++b) { CHECK(lbd[b] == lbh[b]); sum += lbh[b]; }
    CHECK_NEAR(sum, s.lower_bound(), tol);
    (void)hipFree(dlb);
    // min_marginals_cuda: sorted by variable = the nested min_marginals(); device form = host form; unsorted follows var_constraint_indices
    const auto sorted = s.min_marginals_cuda(true);
    const auto nested = s.min_marginals();
    size_t k = 0;
    for (size_t v = 0; v < 9; ++v)
        for (const auto& m : nested[v]) {
            CHECK(std::get<0>(sorted)[k] == (int)v);
            CHECK(double(std::get<1>(sorted)[k]) == m[0] && double(std::get<2>(sorted)[k]) == m[1]);
            ++k;
        }
    CHECK(k == L);
    const auto unsorted = s.min_marginals_cuda(false);
    for (size_t l = 0; l < L; ++l) CHECK(std::get<0>(unsorted)[l] == vc.first[l]);
    int32_t* dv = nullptr;
    REAL *d0 = nullptr, *d1 = nullptr;
    CHECK(hipMalloc((void**)&dv, L * sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&d0, L * sizeof(REAL)) == hipSuccess &&
          hipMalloc((void**)&d1, L * sizeof(REAL)) == hipSuccess);
    s.min_marginals_cuda(dv, d0, d1, true);
    std::vector<int32_t> hv(L);
    std::vector<REAL> h0(L), h1(L);
    CHECK(hipMemcpy(hv.data(), dv, L * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(h0.data(), d0, L * sizeof(REAL), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(h1.data(), d1, L * sizeof(REAL), hipMemcpyDeviceToHost) == hipSuccess);
    for (size_t l = 0; l < L; ++l) CHECK(hv[l] == std::get<0>(sorted)[l] && h0[l] == std::get<1>(sorted)[l] && h1[l] == std::get<2>(sorted)[l]);
    // get / set_solver_costs: a second solver that takes the first one's costs continues identically (device and host forms)
    const auto costs = s.get_solver_costs();
    CHECK(std::get<0>(costs).size() == L);
    bdd_hip_parallel_mma<REAL> t(matching(3), std::vector<double>(9, 0.0)), u(matching(3), std::vector<double>(9, 0.0));
    t.set_solver_costs(costs);
    {
        REAL* d2 = nullptr;
        CHECK(hipMalloc((void**)&d2, L * sizeof(REAL)) == hipSuccess);
        s.get_solver_costs(d0, d1, d2);
        u.set_solver_costs(d0, d1, d2);
        (void)hipFree(d2);
    }
    CHECK_NEAR(t.lower_bound(), s.lower_bound(), 0.0);
    CHECK_NEAR(u.lower_bound(), s.lower_bound(), 0.0);
    bool threw = false;
    try {
        auto bad = costs;
        std::get<2>(bad).pop_back();
        t.set_solver_costs(bad);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    (void)hipFree(dv); (void)hipFree(d0); (void)hipFree(d1);
    // save / load: the loaded object is a solver of the same type in the same state; the other precision is refused
    const std::string path = std::string("/tmp/bddmma_cpp_members_") + (sizeof(REAL) == 8 ? "f64" : "f32") + ".ckpt";
    s.save(path);
    auto r = bdd_hip_parallel_mma<REAL>::load(path);
    CHECK(r.nr_layers() == L && r.nr_bdds() == B);
    CHECK_NEAR(r.lower_bound(), s.lower_bound(), 0.0);
    s.iteration();
    r.iteration();
    CHECK_NEAR(r.lower_bound(), s.lower_bound(), sizeof(REAL) == 8 ? 1e-12 : 1e-5);
    threw = false;
    try {
        using OTHER = typename std::conditional<sizeof(REAL) == 8, float, double>::type;
        (void)bdd_hip_parallel_mma<OTHER>::load(path);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
        (void)bdd_hip_parallel_mma<REAL>::load("/tmp/bddmma_no_such_file.ckpt");
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    std::remove(path.c_str());
}

static void test_checkpoint()
{
    std::vector<double> c(16, -1.0);
    bdd_hip_parallel_mma<double> s(matching(4), c);
    for (int i = 0; i < 3; ++i) s.iteration();
    const std::string path = "/tmp/bddmma_cpp_test.ckpt";
    s.save(path);
    bddmma_solver* h = nullptr;
    CHECK(bddmma_load(&h, 0, path.c_str()) == BDDMMA_OK);
    double lb = 0;
    CHECK(bddmma_lower_bound(h, &lb) == BDDMMA_OK);
    CHECK_NEAR(lb, s.lower_bound(), 1e-12);
    bddmma_destroy(h);
    std::remove(path.c_str());
}

static void test_driver()
{
    const std::string lp = "Minimize\\nx1 + x2 + x3 + x4 + x5 + x6\\nSubject To\\nx1 + x2 + x4 >= 1\\nx1 + x3 + x5 >= 1\\nx2 + x3 + x6 >= 1\\nEnd\\n";
    for (const char* solver : {"cuda parallel mma", "lbfgs cuda mma"}) {
        const std::string cfg = std::string("{\"input\": \"") + lp + "\", \"relaxation solver\": \"" + solver +
                                "\", \"precision\": \"double\", \"termination criteria\": {\"maximum iterations\": 300, \"improvement slope\": 0.0, "
                                "\"minimum improvement\": 0.0}, \"perturbation rounding\": {\"inner iterations\": 50, \"outer iterations\": 50}}";
        bddmma_host::bdd_solver s(cfg, true);
        s.solve();
        // (the costs stay perturbed after the rounding, as in the reference: bdd_solver.cpp:368; the bound is checked below)
        CHECK(s.solution().size() == 6);
        if (s.solution().size() == 6) {
            CHECK(s.ilp().feasible(s.solution()));
            CHECK_NEAR(s.solution_objective(), 2.0, 1e-9);   // optimum of the loose covering instance
        }
        CHECK(s.min_marginals().size() == 6);
        const std::string dual_only = std::string("{\"input\": \"") + lp + "\", \"relaxation solver\": \"" + solver +
                                      "\", \"termination criteria\": {\"maximum iterations\": 300, \"improvement slope\": 0.0, \"minimum improvement\": 0.0}}";
        bddmma_host::bdd_solver d(dual_only, true);
        d.solve();
        CHECK_NEAR(d.lower_bound(), 1.5, 1e-3);   // test_loose_covering_problem.cpp
    }
    // a long row cut by "split bdds" gives a valid bound and more BDDs
    std::string rows = "Minimize\\n";
    for (int i = 0; i < 40; ++i) rows += (i ? " + " : "") + std::to_string(1 + (i * 7) % 11) + " y" + std::to_string(i);
    rows += "\\nSubject To\\n";
    for (int i = 0; i < 40; ++i) rows += (i ? " + y" : "y") + std::to_string(i);
    rows += " >= 4\\nEnd\\n";
    const std::string tail = "\", \"termination criteria\": {\"maximum iterations\": 2000, \"improvement slope\": 0.0, \"minimum improvement\": 0.0}";
    bddmma_host::bdd_solver full("{\"input\": \"" + rows + tail + "}", true), split("{\"input\": \"" + rows + tail + ", \"split bdds\": {\"split length\": 8}}", true);
    full.solve();
    split.solve();
    CHECK(full.bdds().nr_bdds() == 1 && split.bdds().nr_bdds() == 5);
    CHECK_NEAR(full.lower_bound(), 4.0, 1e-6);                // the four cheapest: costs 1 + (7 i mod 11), i = 0, 11, 22, 33
    CHECK(split.lower_bound() <= full.lower_bound() + 1e-6);
    CHECK(split.lower_bound() >= full.lower_bound() - 0.05);
    // option errors of the reference (bdd_solver.cpp:93, :265, :142)
    auto throws = [](const std::string& cfg, const std::string& what) {
        try {
            bddmma_host::bdd_solver s(cfg, true);
            s.solve();
        } catch (const std::exception& e) {
            return std::string(e.what()).find(what) != std::string::npos;
        }
        return false;
    };
    CHECK(throws("{\"input\": \"" + lp + "\", \"variable order\": \"spiral\"}", "Variable order spiral unknown"));
    CHECK(throws("{\"input\": \"" + lp + "\", \"relaxation solver\": \"quantum mma\"}", "relaxation solver quantum mma unknown"));
    CHECK(throws("{\"input\": \"" + lp + "\", \"precision\": \"half\"}", "precision must be"));
    CHECK(throws("{}", "no input specified"));
}

int main()
{
    const std::pair<const char*, std::function<void()>> tests[] = {
        {"matching KATs <double>", test_matching_kats<double>},
        {"matching KATs <float>", test_matching_kats<float>},
        {"costs / min-marginals / solution <double>", test_costs_marginals_solution<double>},
        {"costs / min-marginals / solution <float>", test_costs_marginals_solution<float>},
        {"bdds_solution() reference KAT (test_bdd_cuda_base_sol.cpp) <double>", test_bdds_solution_reference_kat<double>},
        {"bdds_solution() reference KAT (test_bdd_cuda_base_sol.cpp) <float>", test_bdds_solution_reference_kat<float>},
        {"explicit forward_mm / backward_mm / distribute_delta <double>", test_explicit_mm_and_distribute<double>},
        {"explicit forward_mm / backward_mm / distribute_delta <float>", test_explicit_mm_and_distribute<float>},
        {"device-vector overloads <double>", test_device_vectors<double>},
        {"device-vector overloads <float>", test_device_vectors<float>},
        {"L-BFGS wrapper, move construction <double>", test_lbfgs_and_move<double>},
        {"L-BFGS wrapper, move construction <float>", test_lbfgs_and_move<float>},
        {"facades bdd_hip / bdd_lbfgs_hip_mma <double>", test_facades<double>},
        {"facades bdd_hip / bdd_lbfgs_hip_mma <float>", test_facades<float>},
        {"remaining bdd_cuda_base members <double>", test_base_members<double>},
        {"remaining bdd_cuda_base members <float>", test_base_members<float>},
        {"checkpoint", test_checkpoint},
        {"bdd_solver driver", test_driver},
    };
    for (const auto& [name, fn] : tests) {
        const int before = failures;
        try {
            fn();
        } catch (const std::exception& e) {
            std::printf("  EXCEPTION: %s\n", e.what());
            ++failures;
        }
        std::printf("[%s] %s\n", failures == before ? " OK " : "FAIL", name);
    }
    std::printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
