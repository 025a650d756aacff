// bdd_solver.cpp — see bdd_solver.hpp.
#include "bdd_solver.hpp"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../../include/bdd_ilp.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <set>
#include <sstream>

namespace bddmma_host {

namespace {
const std::set<std::string> GPU_MMA{"cuda parallel mma", "hip parallel mma"};
// bdd_solver.cpp:222,251 and README.md:30,59 of the reference use both spellings
const std::set<std::string> GPU_LBFGS{"lbfgs cuda mma", "cuda lbfgs parallel mma", "lbfgs hip mma", "hip lbfgs parallel mma"};
const std::set<std::string> CPU_ONLY{"sequential mma", "parallel mma", "lbfgs parallel mma", "subgradient"};

bool file_exists(const std::string& p)
{
    if (p.size() > 4096 || p.find('\n') != std::string::npos) return false;
    std::ifstream f(p);
    return f.good();
}
std::string slurp(const std::string& p)
{
    std::ifstream f(p);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
std::string extension(const std::string& p)
{
    const size_t d = p.find_last_of('.'), s = p.find_last_of('/');
    return d == std::string::npos || (s != std::string::npos && d < s) ? "" : p.substr(d);
}
}  // namespace

bdd_solver::bdd_solver(const std::string& config, bool quiet, int device) : quiet_(quiet), device_(device)
{
    config_ = json::parse(file_exists(config) ? slurp(config) : config);
    if (!config_.is_object()) throw std::runtime_error("configuration must be a JSON object");
}

bdd_solver::~bdd_solver()
{
    if (lbfgs_) bddmma_lbfgs_destroy(lbfgs_);
    if (solver_) bddmma_destroy(solver_);
}

void bdd_solver::log(const std::string& s) const
{
    if (!quiet_) std::cout << s << std::endl;
}

void bdd_solver::check(int rc) const
{
    if (rc != BDDMMA_OK) throw std::runtime_error(bddmma_last_error(solver_));
}

ilp_input bdd_solver::read_ILP() const
{
    if (!config_.contains("input")) throw std::runtime_error("no input specified");
    const std::string inp = config_["input"].str;
    if (file_exists(inp)) {
        log("[bdd_solver] Read input file " + inp);
        return extension(inp) == ".opb" ? parse_opb(slurp(inp)) : parse_lp_or_opb(slurp(inp));
    }
    log("[bdd_solver] Read input string");
    return parse_lp_or_opb(inp);  // the reference tries the .lp grammar, then OPB (:59-63)
}

void bdd_solver::process_ILP(ilp_input& ilp) const
{
    const std::string order = config_.string_or("variable order", "input");
    if (order == "bfs" || order == "cuthill" || order == "minimum degree")
        throw std::runtime_error("Variable order " + order + " is not available in this backend (ILP re-orderings are outside the hot path)");
    if (order != "input") throw std::runtime_error("Variable order " + order + " unknown");
    if (config_.bool_or("normalize constraints", false)) {
        log("[bdd_solver] Normalize constraints");
        ilp.normalize();
    }
}

bdd_store bdd_solver::transform_to_BDDs(const ilp_input& ilp) const
{
    log("[bdd solver] Compute BDDs");
    bdd_store col = to_bdds(ilp);
    if (config_.contains("split bdds")) {
        const json& sb = config_["split bdds"];
        // the reference tests contains("implication bdd") and then reads key "implication" (:119); accept both
        const bool implication = sb.bool_or("implication bdd", sb.bool_or("implication", false));
        const size_t len = (size_t)sb.number_or("split length", 0);
        // getMaximumOccupancy() of the reference (bdd_preprocessor.cpp:21-30): a tenth of the device's resident threads, asked of the
        // device the solver will run on (MI355X: 256 x 2048 / 10; that figure stands in when no device is visible, e.g. "export bdd lp" on a CPU box)
        uint64_t threads = 0;
        const int device = device_ >= 0 ? device_ : (int)config_.number_or("device", 0);
        if (bddmma_device_chip(device, nullptr, nullptr, &threads) != BDDMMA_OK || threads == 0) threads = 256ull * 2048;
        const auto [n, nv] = col.split_long_bdds(std::max(col.nr_variables(), ilp.nr_variables()), len, (size_t)(threads / 10), implication);
        (void)nv;
        log("[bdd preprocessor] Split " + std::to_string(n) + " BDDs");
        log("[bdd preprocessor] final #BDDs = " + std::to_string(col.nr_bdds()));
    }
    return col;
}

void bdd_solver::construct_solver(const bdd_store& col, const std::vector<double>& costs)
{
    const std::string precision = config_.string_or("precision", "double");
    if (precision != "double" && precision != "single" && precision != "float") throw std::runtime_error("precision must be double|single|float");
    const std::string name = config_.string_or("relaxation solver", "cuda parallel mma");
    if (CPU_ONLY.count(name))
        throw std::runtime_error("relaxation solver " + name + " is a CPU solver of the reference; this backend provides the GPU parallel mma and L-BFGS solvers");
    if (!GPU_MMA.count(name) && !GPU_LBFGS.count(name)) throw std::runtime_error("relaxation solver " + name + " unknown");
    // NB: the reference constructs the <float> GPU solver for "double" and vice versa (:167-174); here
    // "precision" means what it says.
    // The solver's variables are those of the BDDs.  A variable that occurs in the objective only (in no constraint) is free: its
    // better value contributes min(0, c) to the bound and is fixed in the primal.  Such a variable may have ANY index — the LP reader
    // numbers variables by first appearance and reads the objective first — so it is recognised by its BDD count, not by its position
    // (ADVICE r2).  (The reference only asserts on such costs, bdd_parallel_mma_base.cpp:685-695; release builds drop them.)
    const size_t nv = col.nr_variables();
    std::vector<double> c(nv, 0.0);
    std::copy(costs.begin(), costs.begin() + (long)std::min(nv, costs.size()), c.begin());
    const int device = device_ >= 0 ? device_ : (int)config_.number_or("device", 0);
    const int rc = bddmma_create(&solver_, precision == "double" ? BDDMMA_F64 : BDDMMA_F32, device,
                                 col.instructions.data(), col.delimiters.data(), col.nr_bdds(), c.data(), c.size(), nullptr);
    if (rc != BDDMMA_OK) throw std::runtime_error(bddmma_last_error(nullptr));
    std::vector<int32_t> nbdds(bddmma_nr_variables(solver_), 0);
    check(bddmma_num_bdds_per_var(solver_, nbdds.data()));
    free_constant_ = 0;
    free_ones_.assign(costs.size(), -1);  // -1: the variable is in a BDD; 0 / 1: objective-only, its better value
    for (size_t v = 0; v < costs.size(); ++v) {
        if (v < nbdds.size() && nbdds[v] > 0) continue;
        free_ones_[v] = costs[v] < 0 ? 1 : 0;
        if (costs[v] < 0) free_constant_ += costs[v];
    }
    if (GPU_LBFGS.count(name)) {
        const json& p = config_["lbfgs"];  // :179-199
        bddmma_lbfgs_params lp;
        lp.history_size = (int32_t)p.number_or("history size", 5);
        lp.init_step_size = p.number_or("initial step size", 1e-6);
        lp.req_rel_lb_increase = p.number_or("required relative lb increase", 1e-6);
        lp.step_size_decrease_factor = p.number_or("step size decrease factor", 0.8);
        lp.step_size_increase_factor = p.number_or("step size increase factor", 1.1);
        check(bddmma_lbfgs_create(&lbfgs_, solver_, &lp));
    }
    constructed_ = true;
}

bddmma_run_result bdd_solver::solve_dual()
{
    const json& tc = config_["termination criteria"];
    bddmma_run_result res{};
    check(bddmma_run_solver(solver_, lbfgs_, (uint64_t)tc.number_or("maximum iterations", 1000), tc.number_or("minimum improvement", 1e-6),
                            tc.number_or("improvement slope", 1e-9), tc.number_or("time limit", 3600), quiet_ ? 0 : 1, &res));
    log("[bdd solver] Terminated dual optimization");
    result_ = res;
    return res;
}

std::vector<char> bdd_solver::perturbation_rounding()
{
    if (!config_.contains("perturbation rounding")) return {};
    const json& pr = config_["perturbation rounding"];
    std::vector<char> sol(bddmma_nr_variables(solver_), 0);
    int found = 0;
    check(bddmma_incremental_mm_agreement_rounding(solver_, lbfgs_, pr.number_or("initial perturbation", 0.1),
                                                   pr.number_or("perturbation growth rate", 1.1), (uint64_t)pr.number_or("inner iterations", 100),
                                                   (uint64_t)pr.number_or("outer iterations", 100), (uint32_t)pr.number_or("seed", 0),
                                                   quiet_ ? 0 : 1, sol.data(), &found));
    if (!found) {
        log("[incremental primal rounding] No solution found");
        return {};
    }
    sol.resize(ilp_.nr_variables(), 0);  // auxiliary split variables are not part of the answer
    for (size_t v = 0; v < sol.size() && v < free_ones_.size(); ++v)
        if (free_ones_[v] >= 0) sol[v] = free_ones_[v];  // variables of the objective only: their better value (construct_solver)
    solution_ = sol;
    solution_objective_ = ilp_.feasible(sol) ? ilp_.evaluate(sol) : std::numeric_limits<double>::infinity();
    std::ostringstream o;
    o.precision(12);
    o << "[incremental primal rounding] solution objective = " << solution_objective_;
    log(o.str());
    return solution_;
}

void bdd_solver::solve()
{
    if (!constructed_) {
        const auto t0 = std::chrono::steady_clock::now();
        ilp_ = read_ILP();
        process_ILP(ilp_);
        if (config_.contains("export lp")) {
            const std::string path = config_["export lp"].str;
            if (extension(path) != ".lp") throw std::runtime_error("Cannot recognize file extension " + extension(path) + " for exporting problem file");
            std::ofstream(path) << ilp_.write_lp();
        }
        col_ = transform_to_BDDs(ilp_);
        if (config_.contains("print statistics")) {
            std::vector<size_t> per_var(ilp_.nr_variables(), 0);
            for (const auto& c : ilp_.constraints)
                for (size_t v : std::set<size_t>(c.variables.begin(), c.variables.end())) ++per_var[v];
            size_t mn = per_var.empty() ? 0 : per_var[0], mx = 0, sum = 0;
            for (size_t c : per_var) { mn = std::min(mn, c); mx = std::max(mx, c); sum += c; }
            std::cout << "[print_statistics] #variables = " << ilp_.nr_variables() << "\n[print_statistics] #constraints = " << ilp_.constraints.size()
                      << "\n[print_statistics] #BDDs = " << col_.nr_bdds() << "\n[print_statistics] minimum num. constraints per var = " << mn
                      << "\n[print_statistics] maximum num. constraints per var = " << mx << "\n[print_statistics] mean num. constraints per var = "
                      << (per_var.empty() ? 0.0 : (double)sum / (double)per_var.size()) << std::endl;
        }
        if (config_.contains("export bdd lp")) {   // bdd_solver.cpp:400-410
            std::ofstream f(config_["export bdd lp"].str);
            col_.write_bdd_lp(f, ilp_.objective);
        }
        if (config_.contains("export bdd graph")) {   // :432-462: <name>_<bdd>.dot per BDD (the reference also shells out to `dot -Tpng`; not done here)
            const std::string file = config_["export bdd graph"].str;
            const size_t dot = file.rfind('.');
            const std::string stem = dot != std::string::npos ? file.substr(0, dot) : file;
            for (size_t b = 0; b < col_.nr_bdds(); ++b) {
                std::ofstream f(stem + "_" + std::to_string(b) + ".dot");
                col_.export_graphviz(b, f);
            }
        }
        construct_solver(col_, ilp_.objective);
        char buf[64];
        std::snprintf(buf, sizeof buf, "%.3f", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        log(std::string("[bdd solver] set-up time = ") + buf + " s");
    }
    solve_dual();
    perturbation_rounding();
}

double bdd_solver::lower_bound()
{
    double lb = 0;
    check(bddmma_lower_bound(solver_, &lb));
    return lb + ilp_.constant + free_constant_;
}

std::vector<std::vector<std::array<double, 2>>> bdd_solver::min_marginals()
{
    const size_t L = bddmma_nr_layers(solver_), V = bddmma_nr_variables(solver_);
    std::vector<int32_t> var(L);
    const bool f64 = bddmma_precision(solver_) == BDDMMA_F64;
    std::vector<double> m0d(f64 ? L : 0), m1d(f64 ? L : 0);
    std::vector<float> m0f(f64 ? 0 : L), m1f(f64 ? 0 : L);
    check(bddmma_min_marginals(solver_, 1, var.data(), f64 ? (void*)m0d.data() : (void*)m0f.data(), f64 ? (void*)m1d.data() : (void*)m1f.data(), 0));
    std::vector<std::vector<std::array<double, 2>>> out(V);
    for (size_t l = 0; l < L; ++l) out[var[l]].push_back({f64 ? m0d[l] : (double)m0f[l], f64 ? m1d[l] : (double)m1f[l]});
    out.resize(ilp_.nr_variables() ? std::min(V, ilp_.nr_variables()) : V);
    return out;
}

// ---------------------------------------------------------------------------------------------- batch farm
// one layout build per device slot at a time: each gets its share of the host cores (8 slots x 32 builder threads would oversubscribe them)
// (set per worker thread: the process-wide value, which an application or a second concurrent batch may own, is left alone)
static void share_layout_threads(size_t slots)
{
    const unsigned hw = std::thread::hardware_concurrency();
    bddmma_set_thread_layout_threads((int)std::max<size_t>(1, std::min<size_t>(32, (hw ? hw : 1) / std::max<size_t>(1, slots))));
}

std::vector<batch_result> solve_batch(const std::vector<std::string>& configs, const std::vector<int>& devices, bool quiet)
{
    std::vector<batch_result> out(configs.size());
    if (devices.empty()) throw std::runtime_error("solve_batch: no devices");
    std::atomic<size_t> next{0};
    auto worker = [&](int device) {
        share_layout_threads(devices.size());
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= configs.size()) return;
            batch_result& r = out[i];
            r.config = configs[i];
            r.device = device;
            const auto t0 = std::chrono::steady_clock::now();
            try {
                bdd_solver s(configs[i], quiet, device);
                s.solve();
                r.lower_bound = s.lower_bound();
                r.iterations = s.result().iterations;
                r.has_primal = !s.solution().empty();
                r.primal = s.solution_objective();
                r.ok = true;
            } catch (const std::exception& e) {
                r.error = e.what();
            }
            r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    };
    std::vector<std::thread> th;
    for (int d : devices) th.emplace_back(worker, d);
    for (auto& t : th) t.join();
    return out;
}

std::vector<bench_result> bench_set_cover(uint64_t n_vars, uint64_t n_rows, uint64_t k, const std::vector<uint64_t>& seeds, const std::vector<int>& devices,
                                          const std::string& precision, uint64_t warmup, uint64_t iterations, double* aggregate)
{
    if (devices.empty() || seeds.empty()) throw std::runtime_error("bench_set_cover: no devices / seeds");
    std::vector<bench_result> out(seeds.size());
    std::vector<double> ms(seeds.size(), 0.0);
    // every instance is built first; the timed loops start together (a barrier over the threads) and run concurrently
    std::mutex m;
    std::condition_variable cv;
    size_t ready = 0;
    auto worker = [&](size_t i) {
        share_layout_threads(seeds.size());
        bench_result& r = out[i];
        r.device = devices[i % devices.size()];
        r.seed = seeds[i];
        bddmma_solver* s = nullptr;
        try {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<uint64_t> rows(n_rows * k);
            std::vector<double> costs(n_vars);
            if (bddilp_random_set_cover(n_vars, n_rows, k, seeds[i], rows.data(), costs.data()) != BDDILP_OK) throw std::runtime_error("invalid instance sizes");
            bdd_store col;
            std::vector<size_t> row(k);
            for (uint64_t b = 0; b < n_rows; ++b) {
                for (uint64_t j = 0; j < k; ++j) row[j] = (size_t)rows[b * k + j];
                col.add_covering(row);
            }
            costs.resize(col.nr_variables());
            const int rc = bddmma_create(&s, (precision == "float" || precision == "single") ? BDDMMA_F32 : BDDMMA_F64, r.device, col.instructions.data(),
                                         col.delimiters.data(), col.nr_bdds(), costs.data(), costs.size(), nullptr);
            if (rc != BDDMMA_OK) throw std::runtime_error(bddmma_last_error(nullptr));
            if (bddmma_iterations(s, 0.5, warmup) != BDDMMA_OK || bddmma_synchronize(s) != BDDMMA_OK) throw std::runtime_error(bddmma_last_error(s));
            r.construct_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            r.ok = true;
        } catch (const std::exception& e) {
            r.error = e.what();
        }
        {
            std::unique_lock<std::mutex> l(m);
            if (++ready == seeds.size()) cv.notify_all();
            else cv.wait(l, [&] { return ready == seeds.size(); });
        }
        if (r.ok) {
            const auto t0 = std::chrono::steady_clock::now();
            if (bddmma_iterations(s, 0.5, iterations) != BDDMMA_OK || bddmma_synchronize(s) != BDDMMA_OK) {
                r.ok = false;
                r.error = bddmma_last_error(s);
            }
            ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r.ok) {
                r.iterations_per_second = iterations / (ms[i] * 1e-3);
                bddmma_lower_bound(s, &r.lower_bound);
            }
        }
        if (s) bddmma_destroy(s);
    };
    std::vector<std::thread> th;
    for (size_t i = 0; i < seeds.size(); ++i) th.emplace_back(worker, i);
    for (auto& t : th) t.join();
    if (aggregate) {
        double worst = 0;
        size_t n_ok = 0;
        for (size_t i = 0; i < seeds.size(); ++i)
            if (out[i].ok) { worst = std::max(worst, ms[i]); ++n_ok; }
        *aggregate = worst > 0 ? n_ok * iterations / (worst * 1e-3) : 0;
    }
    return out;
}

}  // namespace bddmma_host
