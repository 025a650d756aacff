// bdd_solver.hpp — the reference's orchestrator (LPMP::bdd_solver, include/bdd_solver/bdd_solver.h:45-103,
// src/bdd_solver/bdd_solver.cpp:36-527) re-hosted over the C-ABI of include/bdd_mma.h.
//
// Same pipeline and JSON keys: read_ILP -> process_ILP -> transform_to_BDDs -> construct_solver -> solve_dual ->
// perturbation_rounding (bdd_solver::solve, :477-495).  Only the relaxation solvers of the hot path exist
// ("cuda parallel mma" and the GPU L-BFGS names); asking for a CPU solver, an ILP re-ordering or an exporter
// other than .lp throws std::runtime_error like the reference does for an unknown option.
// bdd_amd/bdd_solver.py is the same driver in Python (the pybind module's role, bdd_solver_py.cpp:9-20).
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
    explicit bdd_solver(const std::string& config, bool quiet = false);
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
    bddmma_solver* handle() { return solver_; }

private:
    void log(const std::string& s) const;
    void check(int rc) const;
    json config_;
    bool quiet_;
    bool constructed_ = false;
    ilp_input ilp_;
    bdd_store col_;
    bddmma_solver* solver_ = nullptr;
    bddmma_lbfgs* lbfgs_ = nullptr;
    std::vector<char> solution_;
    double solution_objective_ = 0;
};

}  // namespace bddmma_host
