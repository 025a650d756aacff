// bdd_hip_parallel_mma.hpp — header-only C++ class over the C-ABI (include/bdd_mma.h) that satisfies the
// reference's relaxation-solver concept, so it can be listed as one more alternative of
// `bdd_solver::solver_type` (reference: include/bdd_solver/bdd_solver.h:64-69) and constructed where
// `"relaxation solver": "cuda parallel mma"` is handled (src/bdd_solver/bdd_solver.cpp:164-176).
// Member names, argument meaning and error behaviour (std::runtime_error) follow
// LPMP::bdd_cuda_parallel_mma<REAL> / bdd_cuda_base<REAL>.  Needs no HIP headers: device vectors are
// raw device pointers (`device_ptr<REAL>`) owned by the caller or by the handle.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/bdd_mma.h"

namespace LPMP {

template <typename REAL>
class bdd_hip_parallel_mma {
    static_assert(std::is_same<REAL, float>::value || std::is_same<REAL, double>::value, "REAL must be float or double");

   public:
    using value_type = REAL;
    static constexpr int precision = std::is_same<REAL, double>::value ? BDDMMA_F64 : BDDMMA_F32;

    bdd_hip_parallel_mma() = default;

    // From the flat storage of BDD::bdd_collection: `instr` = bdd_instructions.data() (bit-identical layout),
    // `delims` = bdd_delimiters.data().  Mirrors bdd_cuda_parallel_mma(const bdd_collection&, const std::vector<double>&).
    bdd_hip_parallel_mma(const bddmma_instruction* instr, const uint64_t* delims, size_t nr_bdds,
                         const std::vector<double>& costs_hi = {}, int device = 0, const bddmma_options* opts = nullptr)
    {
        check(bddmma_create(&h_, precision, device, instr, delims, nr_bdds, costs_hi.data(), costs_hi.size(), opts), nullptr);
    }
    // Any type with the accessors of BDD::bdd_collection used by the reference constructor
    // (bdd_cuda_base.cu:55-144): nr_bdds(), nr_bdd_nodes(b), offset(b), operator()(b, i).
    template <typename BDD_COLLECTION, typename = decltype(std::declval<const BDD_COLLECTION&>().nr_bdds())>
    explicit bdd_hip_parallel_mma(const BDD_COLLECTION& bdd_col, const std::vector<double>& costs_hi = {}, int device = 0)
    {
        std::vector<bddmma_instruction> instr;
        std::vector<uint64_t> delims;
        flatten(bdd_col, instr, delims);
        check(bddmma_create(&h_, precision, device, instr.data(), delims.data(), bdd_col.nr_bdds(), costs_hi.data(), costs_hi.size(), nullptr), nullptr);
    }
    // The collection's BDDs as the dense arrays bddmma_create takes (arc targets re-based from the collection's storage offsets to the
    // dense array).  Needs no GPU: oracle/ref_driver.cpp instantiates it with the reference's own BDD::bdd_collection
    // (include/bdd_collection/bdd_collection.h:206 `operator()(bdd_nr, offset)`) and tests/test_bdd_builders.py compares the result with
    // the reference-side export.
    template <typename BDD_COLLECTION>
    static void flatten(const BDD_COLLECTION& bdd_col, std::vector<bddmma_instruction>& instr, std::vector<uint64_t>& delims)
    {
        instr.clear();
        delims.assign(1, 0);
        for (size_t b = 0; b < bdd_col.nr_bdds(); ++b) {
            const size_t off = bdd_col.offset(b), base = instr.size();
            for (size_t i = 0; i < bdd_col.nr_bdd_nodes(b); ++i) {
                const auto in = bdd_col(b, off + i);
                const bool term = in.index >= BDDMMA_BOTSINK;
                instr.push_back({term ? in.lo : in.lo - off + base, term ? in.hi : in.hi - off + base, in.index});
            }
            delims.push_back(instr.size());
        }
    }
    ~bdd_hip_parallel_mma() { bddmma_destroy(h_); }
    bdd_hip_parallel_mma(bdd_hip_parallel_mma&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    bdd_hip_parallel_mma& operator=(bdd_hip_parallel_mma&& o) noexcept
    {
        if (this != &o) { bddmma_destroy(h_); h_ = o.h_; o.h_ = nullptr; }
        return *this;
    }
    bdd_hip_parallel_mma(const bdd_hip_parallel_mma&) = delete;
    bdd_hip_parallel_mma& operator=(const bdd_hip_parallel_mma&) = delete;

    // ---- sizes (bdd_cuda_base.h:98-116)
    size_t nr_variables() const { return bddmma_nr_variables(h_); }
    size_t nr_bdds() const { return bddmma_nr_bdds(h_); }
    size_t nr_layers() const { return bddmma_nr_layers(h_); }
    size_t nr_bdd_nodes() const { return bddmma_nr_bdd_nodes(h_); }
    size_t nr_hops() const { return bddmma_nr_hops(h_); }
    size_t nr_bdds(const size_t var) const
    {
        std::vector<int32_t> n(nr_variables());
        check(bddmma_num_bdds_per_var(h_, n.data()));
        return n.at(var);
    }
    // nr_layers(hop) / nr_bdd_nodes(hop), bdd_cuda_base.h:106-113, for hop < nr_hops(): non-terminal layers / nodes at that distance from
    // the roots (the reference counts the terminal layers of BDDs that end at the hop as well; nr_layers() here is non-terminal)
    size_t nr_layers(const int hop_index) const { return per_hop(bddmma_layers_per_hop).at((size_t)hop_index); }
    size_t nr_bdd_nodes(const int hop_index) const { return per_hop(bddmma_nodes_per_hop).at((size_t)hop_index); }
    // var_constraint_indices(), bdd_cuda_base.h:121 / bdd_cuda_base.cu:1450-1461: (primal variable, BDD) of every dual variable, hop-major
    std::pair<std::vector<int>, std::vector<int>> var_constraint_indices() const
    {
        std::vector<int32_t> v(nr_layers()), b(nr_layers());
        check(bddmma_layer_variables(h_, v.data()));
        check(bddmma_layer_bdds(h_, b.data()));
        return {std::vector<int>(v.begin(), v.end()), std::vector<int>(b.begin(), b.end())};
    }
    std::vector<int> get_primal_variable_index() const { return var_constraint_indices().first; }
    std::vector<int> get_bdd_index() const { return var_constraint_indices().second; }
    std::vector<int> get_num_bdds_per_var() const
    {
        std::vector<int32_t> n(nr_variables());
        check(bddmma_num_bdds_per_var(h_, n.data()));
        return std::vector<int>(n.begin(), n.end());
    }

    // ---- costs (bdd_cuda_base.cu:439-558)
    void update_costs(const std::vector<REAL>& cost_delta_0, const std::vector<REAL>& cost_delta_1)
    {
        check(bddmma_update_costs(h_, cost_delta_0.data(), cost_delta_0.size(), cost_delta_1.data(), cost_delta_1.size(), precision, 0));
    }
    void update_costs(const REAL* dev_cost_delta_0, size_t n0, const REAL* dev_cost_delta_1, size_t n1)  // device_vector overload
    {
        check(bddmma_update_costs(h_, dev_cost_delta_0, n0, dev_cost_delta_1, n1, precision, 1));
    }
    void set_cost(const double c, const size_t var) { check(bddmma_set_cost(h_, c, var)); }
    std::vector<REAL> get_primal_objective_vector_host()
    {
        std::vector<REAL> v(nr_variables());
        check(bddmma_primal_objective_vec(h_, v.data(), 0));
        return v;
    }
    void compute_primal_objective_vec(REAL* dev_primal_obj) { check(bddmma_primal_objective_vec(h_, dev_primal_obj, 1)); }
    // get_solver_costs / set_solver_costs, bdd_cuda_base.h:124-135 (bdd_cuda_base.cu:1308-1344): {lo, hi, deferred mm difference}, nr_layers() each.
    // Device pointers as in the reference; the tuple-returning form hands out host vectors (no device_vector type here).
    void get_solver_costs(REAL* dev_lo, REAL* dev_hi, REAL* dev_deferred_mm_diff) const
    {
        check(bddmma_get_solver_costs(h_, dev_lo, dev_hi, dev_deferred_mm_diff, 1));
    }
    using SOLVER_COSTS_VECS = std::tuple<std::vector<REAL>, std::vector<REAL>, std::vector<REAL>>;
    SOLVER_COSTS_VECS get_solver_costs() const
    {
        const size_t L = nr_layers();
        SOLVER_COSTS_VECS c{std::vector<REAL>(L), std::vector<REAL>(L), std::vector<REAL>(L)};
        check(bddmma_get_solver_costs(h_, std::get<0>(c).data(), std::get<1>(c).data(), std::get<2>(c).data(), 0));
        return c;
    }
    void set_solver_costs(const REAL* dev_lo, const REAL* dev_hi, const REAL* dev_deferred_mm_diff)
    {
        check(bddmma_set_solver_costs(h_, dev_lo, dev_hi, dev_deferred_mm_diff, 1));
    }
    void set_solver_costs(const SOLVER_COSTS_VECS& c)
    {
        if (std::get<0>(c).size() != nr_layers() || std::get<1>(c).size() != nr_layers() || std::get<2>(c).size() != nr_layers())
            throw std::runtime_error("bdd_hip_parallel_mma: set_solver_costs needs nr_layers() entries per vector");
        check(bddmma_set_solver_costs(h_, std::get<0>(c).data(), std::get<1>(c).data(), std::get<2>(c).data(), 0));
    }

    // ---- sweeps and bounds
    void forward_run() { check(bddmma_forward_run(h_)); }
    void backward_run() { check(bddmma_backward_run(h_)); }
    double lower_bound()
    {
        double lb;
        check(bddmma_lower_bound(h_, &lb));
        return lb;
    }
    // lower_bound_per_bdd(device_ptr), bdd_cuda_base.h:70 (bdd_cuda_base.cu:1253-1259): nr_bdds() values
    void lower_bound_per_bdd(REAL* dev_lb_per_bdd) { check(bddmma_lower_bound_per_bdd(h_, dev_lb_per_bdd, 1)); }
    std::vector<REAL> lower_bound_per_bdd_host()
    {
        std::vector<REAL> lb(nr_bdds());
        check(bddmma_lower_bound_per_bdd(h_, lb.data(), 0));
        return lb;
    }

    // ---- parallel mma (bdd_cuda_parallel_mma.cu:142-153, 207-257, 301-346, 410-430)
    void iteration(const REAL omega = 0.5) { check(bddmma_iteration(h_, omega)); }
    void forward_mm(const REAL omega, std::vector<REAL>& delta_lo_hi) { check(bddmma_forward_mm(h_, omega, delta_lo_hi.data(), 0)); }
    void backward_mm(const REAL omega, std::vector<REAL>& delta_lo_hi) { check(bddmma_backward_mm(h_, omega, delta_lo_hi.data(), 0)); }
    void forward_mm(const REAL omega, REAL* dev_delta_lo_hi) { check(bddmma_forward_mm(h_, omega, dev_delta_lo_hi, 1)); }
    void backward_mm(const REAL omega, REAL* dev_delta_lo_hi) { check(bddmma_backward_mm(h_, omega, dev_delta_lo_hi, 1)); }
    void normalize_delta(REAL* dev_delta_lo_hi) const { check(bddmma_normalize_delta(h_, dev_delta_lo_hi, 1)); }
    void distribute_delta() { check(bddmma_distribute_delta(h_)); }

    // ---- min-marginals [var][bdd] -> {mm0, mm1} (bdd_cuda_base.cu:751-786)
    std::vector<std::vector<std::array<double, 2>>> min_marginals()
    {
        const size_t L = nr_layers();
        std::vector<int32_t> var(L);
        std::vector<REAL> m0(L), m1(L);
        check(bddmma_min_marginals(h_, 1, var.data(), m0.data(), m1.data(), 0));
        std::vector<std::vector<std::array<double, 2>>> out(nr_variables());
        for (size_t k = 0; k < L; ++k) out[var[k]].push_back({double(m0[k]), double(m1[k])});
        return out;
    }
    // min_marginals_cuda(get_sorted), bdd_cuda_base.h:84 (bdd_cuda_base.cu:716-749): (primal variable, mm_lo, mm_hi) per dual variable — what
    // incremental_mm_agreement_rounding_cuda.cu:262-362 consumes; sorted by variable (then BDD) or in the solver's layer order.
    // Device buffers of nr_layers() entries each, owned by the caller ...
    void min_marginals_cuda(int32_t* dev_var, REAL* dev_mm_lo, REAL* dev_mm_hi, bool get_sorted = true)
    {
        check(bddmma_min_marginals(h_, get_sorted ? 1 : 0, dev_var, dev_mm_lo, dev_mm_hi, 1));
    }
    // ... or as host vectors
    std::tuple<std::vector<int>, std::vector<REAL>, std::vector<REAL>> min_marginals_cuda(bool get_sorted = true)
    {
        const size_t L = nr_layers();
        std::vector<int32_t> var(L);
        std::vector<REAL> m0(L), m1(L);
        check(bddmma_min_marginals(h_, get_sorted ? 1 : 0, var.data(), m0.data(), m1.data(), 0));
        return {std::vector<int>(var.begin(), var.end()), std::move(m0), std::move(m1)};
    }
    // two_dim_variable_array<REAL> bdds_solution() (bdd_cuda_base.cu:1204-1233): [variable][bdd] -> 0 / 1, the argmin path of every BDD,
    // BDDs of a variable in ascending order (primal_variable_sorting_order_, :379-391); nested vectors instead of two_dim_variable_array
    std::vector<std::vector<REAL>> bdds_solution()
    {
        const size_t L = nr_layers(), V = nr_variables();
        std::vector<char> sorted(L);
        check(bddmma_bdds_solution(h_, 1, sorted.data(), 0));
        std::vector<int32_t> n(V);
        check(bddmma_num_bdds_per_var(h_, n.data()));
        std::vector<std::vector<REAL>> out(V);
        size_t k = 0;
        for (size_t v = 0; v < V; ++v) {
            out[v].resize((size_t)n[v]);
            for (int32_t b = 0; b < n[v]; ++b, ++k) out[v][(size_t)b] = REAL(sorted[k]);
        }
        return out;
    }
    std::vector<char> bdds_solution_vec_host()
    {
        std::vector<char> s(nr_layers());
        check(bddmma_bdds_solution(h_, 0, s.data(), 0));
        return s;
    }

    // ---- L-BFGS support on device vectors (lbfgs.h:22-27)
    void net_solver_costs(REAL* dev_out) const { check(bddmma_net_solver_costs(h_, dev_out, 1)); }
    void bdds_solution_vec(char* dev_out) { check(bddmma_bdds_solution(h_, 0, dev_out, 1)); }
    void make_dual_feasible(REAL* dev_g) const { check(bddmma_make_dual_feasible(h_, dev_g, 1)); }
    void gradient_step(const REAL* dev_g, double step_size) { check(bddmma_gradient_step(h_, dev_g, step_size, 1)); }

    // cereal save / load of the reference (bdd_cuda_base.h:167-170, bdd_cuda_base.cu:1486-1550; pickled by bdd_cuda_parallel_mma_py.cu:15-38):
    // the archive is a file; it holds the device layout and the costs, load() rebuilds nothing
    void save(const std::string& path) const { check(bddmma_save(h_, path.c_str())); }
    static bdd_hip_parallel_mma load(const std::string& path, int device = 0)
    {
        bddmma_solver* h = nullptr;
        const int rc = bddmma_load(&h, device, path.c_str());
        if (rc != BDDMMA_OK) throw std::runtime_error(std::string("bdd_hip_parallel_mma: ") + bddmma_last_error(nullptr));
        if (bddmma_precision(h) != precision) {
            bddmma_destroy(h);
            throw std::runtime_error("bdd_hip_parallel_mma: the checkpoint was written by a solver of the other precision");
        }
        bdd_hip_parallel_mma s;
        s.h_ = h;
        return s;
    }
    bddmma_solver* handle() { return h_; }

   private:
    std::vector<uint64_t> per_hop(int (*f)(const bddmma_solver*, uint64_t*)) const
    {
        std::vector<uint64_t> n(nr_hops());
        check(f(h_, n.data()));
        return n;
    }
    void check(int rc) const { check(rc, h_); }
    static void check(int rc, const bddmma_solver* h)
    {
        if (rc != BDDMMA_OK) throw std::runtime_error(std::string("bdd_hip_parallel_mma: ") + bddmma_last_error(h));
    }
    bddmma_solver* h_ = nullptr;
};

// lbfgs<bdd_cuda_parallel_mma<REAL>, ...> (include/bdd_solver/lbfgs.h:35-111) over the HIP solver.
template <typename REAL>
class bdd_hip_lbfgs_mma {
   public:
    bdd_hip_lbfgs_mma(bdd_hip_parallel_mma<REAL>&& s, const bddmma_lbfgs_params* p = nullptr) : solver_(std::move(s))
    {
        if (bddmma_lbfgs_create(&l_, solver_.handle(), p) != BDDMMA_OK)
            throw std::runtime_error(std::string("bdd_hip_lbfgs_mma: ") + bddmma_last_error(solver_.handle()));
    }
    ~bdd_hip_lbfgs_mma() { bddmma_lbfgs_destroy(l_); }
    bdd_hip_lbfgs_mma(const bdd_hip_lbfgs_mma&) = delete;
    void iteration()
    {
        if (bddmma_lbfgs_iteration(l_) != BDDMMA_OK)
            throw std::runtime_error(std::string("bdd_hip_lbfgs_mma: ") + bddmma_last_error(solver_.handle()));
    }
    double lower_bound() { return solver_.lower_bound(); }
    bdd_hip_parallel_mma<REAL>& solver() { return solver_; }
    bddmma_lbfgs* lbfgs_handle() { return l_; }

   private:
    bdd_hip_parallel_mma<REAL> solver_;
    bddmma_lbfgs* l_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------
// The two CUDA-free façades the reference ships for embedding the GPU solver in other code
// (include/bdd_cuda.h:9-31 `bdd_cuda<REAL>`, include/bdd_lbfgs_cuda_mma.h:9-36 `bdd_lbfgs_cuda_mma<REAL>`; their
// implementations throw "not compiled with CUDA support" without CUDA, src/bdd_cuda.cpp:27,38).  Same public
// members, same argument meaning; move-only.  min_marginals() returns [variable][bdd] -> {mm0, mm1} as nested vectors
// (the reference's two_dim_variable_array has the same indexing).
template <typename REAL>
class bdd_hip {
   public:
    template <typename BDD_COLLECTION>
    explicit bdd_hip(const BDD_COLLECTION& bdd_col) : s_(bdd_col)
    {
    }
    template <typename BDD_COLLECTION, typename ITERATOR>
    bdd_hip(const BDD_COLLECTION& bdd_col, ITERATOR cost_begin, ITERATOR cost_end) : s_(bdd_col)
    {
        update_costs(cost_begin, cost_begin, cost_begin, cost_end);  // as bdd_cuda.h:36-41
    }
    bdd_hip(bdd_hip&&) = default;
    bdd_hip& operator=(bdd_hip&&) = default;

    template <typename ITERATOR>
    void update_costs(ITERATOR cost_lo_begin, ITERATOR cost_lo_end, ITERATOR cost_hi_begin, ITERATOR cost_hi_end)
    {
        s_.update_costs(std::vector<REAL>(cost_lo_begin, cost_lo_end), std::vector<REAL>(cost_hi_begin, cost_hi_end));
    }
    double lower_bound() { return s_.lower_bound(); }
    size_t nr_variables() const { return s_.nr_variables(); }
    std::vector<std::vector<std::array<double, 2>>> min_marginals() { return s_.min_marginals(); }
    void iteration() { s_.iteration(); }
    void backward_run() { s_.backward_run(); }
    // incremental_mm_agreement_rounding_cuda (incremental_mm_agreement_rounding_cuda.cu:333-372); empty: no solution found
    std::vector<char> incremental_mm_agreement_rounding(const double init_delta, const double delta_growth_rate, const int num_itr_lb,
                                                        const int num_rounds = 500)
    {
        return round(s_.handle(), nullptr, init_delta, delta_growth_rate, num_itr_lb, num_rounds);
    }
    bdd_hip_parallel_mma<REAL>& solver() { return s_; }

    static std::vector<char> round(bddmma_solver* h, bddmma_lbfgs* l, double init_delta, double growth, int num_itr_lb, int num_rounds)
    {
        std::vector<char> sol(bddmma_nr_variables(h), 0);
        int found = 0;
        if (bddmma_incremental_mm_agreement_rounding(h, l, init_delta, growth, (uint64_t)num_itr_lb, (uint64_t)num_rounds, 0, 0, sol.data(), &found) != BDDMMA_OK)
            throw std::runtime_error(std::string("incremental_mm_agreement_rounding: ") + bddmma_last_error(h));
        if (!found) sol.clear();
        return sol;
    }

   private:
    bdd_hip_parallel_mma<REAL> s_;
};

template <typename REAL>
class bdd_lbfgs_hip_mma {
   public:
    template <typename BDD_COLLECTION>
    bdd_lbfgs_hip_mma(const BDD_COLLECTION& bdd_col, const int history_size, const double init_step_size = 1e-6,
                      const double req_rel_lb_increase = 1e-6, const double step_size_decrease_factor = 0.8,
                      const double step_size_increase_factor = 1.1)
        : l_(bdd_hip_parallel_mma<REAL>(bdd_col), params(history_size, init_step_size, req_rel_lb_increase, step_size_decrease_factor, step_size_increase_factor))
    {
    }
    template <typename BDD_COLLECTION, typename ITERATOR>
    bdd_lbfgs_hip_mma(const BDD_COLLECTION& bdd_col, ITERATOR cost_begin, ITERATOR cost_end, const int history_size,
                      const double init_step_size = 1e-6, const double req_rel_lb_increase = 1e-6,
                      const double step_size_decrease_factor = 0.8, const double step_size_increase_factor = 1.1)
        : bdd_lbfgs_hip_mma(bdd_col, history_size, init_step_size, req_rel_lb_increase, step_size_decrease_factor, step_size_increase_factor)
    {
        update_costs(cost_begin, cost_begin, cost_begin, cost_end);  // as bdd_lbfgs_cuda_mma.h:41-52
    }
    template <typename ITERATOR>
    void update_costs(ITERATOR cost_lo_begin, ITERATOR cost_lo_end, ITERATOR cost_hi_begin, ITERATOR cost_hi_end)
    {
        const std::vector<REAL> lo(cost_lo_begin, cost_lo_end), hi(cost_hi_begin, cost_hi_end);
        constexpr int prec = std::is_same<REAL, double>::value ? BDDMMA_F64 : BDDMMA_F32;
        // lbfgs::update_costs also drops the history (lbfgs_impl.h:343-364)
        if (bddmma_lbfgs_update_costs(l_.lbfgs_handle(), lo.data(), lo.size(), hi.data(), hi.size(), prec, 0) != BDDMMA_OK)
            throw std::runtime_error(std::string("bdd_lbfgs_hip_mma: ") + bddmma_last_error(l_.solver().handle()));
    }
    double lower_bound() { return l_.lower_bound(); }
    size_t nr_variables() { return l_.solver().nr_variables(); }
    std::vector<std::vector<std::array<double, 2>>> min_marginals() { return l_.solver().min_marginals(); }
    void iteration() { l_.iteration(); }
    void backward_run() { l_.solver().backward_run(); }
    std::vector<char> incremental_mm_agreement_rounding(const double init_delta, const double delta_growth_rate, const int num_itr_lb,
                                                        const int num_rounds = 500)
    {
        return bdd_hip<REAL>::round(l_.solver().handle(), l_.lbfgs_handle(), init_delta, delta_growth_rate, num_itr_lb, num_rounds);
    }

   private:
    static const bddmma_lbfgs_params* params(int m, double s, double r, double d, double i)
    {
        static thread_local bddmma_lbfgs_params p;
        p = bddmma_lbfgs_params{m, s, r, d, i};
        return &p;
    }
    bdd_hip_lbfgs_mma<REAL> l_;
};

}  // namespace LPMP
