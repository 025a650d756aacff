// bdd_solver.hpp — the reference's orchestrator (LPMP::bdd_solver, include/bdd_solver/bdd_solver.h:45-103,
// src/bdd_solver/bdd_solver.cpp:36-527) re-hosted over the C-ABI of include/bdd_mma.h.
//
// Same pipeline and JSON keys: read_ILP -> process_ILP -> transform_to_BDDs -> construct_solver -> solve_dual ->
// perturbation_rounding (bdd_solver::solve, :477-495).  Only the relaxation solvers of the hot path exist
// ("cuda parallel mma" and the GPU L-BFGS names); asking for a CPU solver, an ILP re-ordering or an exporter
// other than .lp throws std::runtime_error like the reference does for an unknown option.
// bdd_solver_py.cpp binds this class for Python (the reference's pybind module, src/bdd_solver/bdd_solver_py.cpp:9-20);
// bdd_amd/bdd_solver.py is the same driver written in Python, kept as the readable specification.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "../../../include/bdd_mma.h"
#include "ilp.hpp"
#include "json_min.hpp"

namespace bddmma_host {

class bdd_solver {
public:
    // config: JSON text or the path of a JSON file (bdd_solver.cpp:468-475)
    // device >= 0 overrides the config's "device" (the batch farm places one solver per GPU)
    explicit bdd_solver(const std::string& config, bool quiet = false, int device = -1);
    ~bdd_solver();
    bdd_solver(const bdd_solver&) = delete;
    bdd_solver& operator=(const bdd_solver&) = delete;

    ilp_input read_ILP() const;                                  // :44-66
    void process_ILP(ilp_input& ilp) const;                      // :71-103
    bdd_store transform_to_BDDs(const ilp_input& ilp) const;     // :112-123
    void construct_solver(const bdd_store& col, const std::vector<double>& costs);  // :130-267
    bddmma_run_result solve_dual();                              // :277-309
    std::vector<char> perturbation_rounding();                   // :318-380 (empty: none found / not requested)
    void solve();                                                // :477-495

    double lower_bound();                                        // incl. the objective constant
    std::vector<std::vector<std::array<double, 2>>> min_marginals();  // [var][bdd] -> {mm0, mm1}

    const ilp_input& ilp() const { return ilp_; }
    const bdd_store& bdds() const { return col_; }
    const std::vector<char>& solution() const { return solution_; }
    double solution_objective() const { return solution_objective_; }
    const bddmma_run_result& result() const { return result_; }  // of the last solve_dual()
    bddmma_solver* handle() { return solver_; }

private:
    void log(const std::string& s) const;
    void check(int rc) const;
    json config_;
    bool quiet_;
    int device_ = -1;
    bool constructed_ = false;
    // variables that occur in the objective only (in no constraint): fixed to their better value, min(0, c) goes to the bound
    double free_constant_ = 0;
    std::vector<signed char> free_ones_;  // per ILP variable: -1 in a BDD, 0 / 1 objective-only (its better value)
    ilp_input ilp_;
    bdd_store col_;
    bddmma_solver* solver_ = nullptr;
    bddmma_lbfgs* lbfgs_ = nullptr;
    std::vector<char> solution_;
    double solution_objective_ = 0;
    bddmma_run_result result_{};
};

// ---- batch farm: independent instances over the GPUs of one node, one host thread per device, no collective
// (BASELINE.json north_star / configs[4]; the reference hard-codes device 0, include/cuda_utils.h:111-114).
struct batch_result {
    std::string config;
    int device = -1;
    bool ok = false;
    std::string error;
    double lower_bound = 0, primal = 0, seconds = 0;
    bool has_primal = false;
    uint64_t iterations = 0;
};
// configs are handed to the device threads dynamically (a thread takes the next one when its GPU is free); devices may repeat
std::vector<batch_result> solve_batch(const std::vector<std::string>& configs, const std::vector<int>& devices, bool quiet);

struct bench_result {
    int device = -1;
    uint64_t seed = 0;
    bool ok = false;
    std::string error;
    double construct_seconds = 0, iterations_per_second = 0, lower_bound = 0;
};
// random set cover (bddilp_random_set_cover) with seeds[i] on devices[i % devices.size()], all started together; *aggregate =
// iterations of all instances / the slowest one's time (the weak-scaling figure of the benchmark)
std::vector<bench_result> bench_set_cover(uint64_t n_vars, uint64_t n_rows, uint64_t k, const std::vector<uint64_t>& seeds, const std::vector<int>& devices,
                                          const std::string& precision, uint64_t warmup, uint64_t iterations, double* aggregate);

}  // namespace bddmma_host
