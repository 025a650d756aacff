// ilp_capi.cpp — extern "C" glue of include/bdd_ilp.h over ilp.hpp / bdd_store.hpp.  No exception crosses the ABI.
#include <cstring>
#include <fstream>
#include <vector>
#include <set>
#include <stdexcept>
#include <string>

#include "../../../include/bdd_ilp.h"
#include "ilp.hpp"

using namespace bddmma_host;

struct bddilp { ilp_input ilp; };
struct bddilp_bdds { bdd_store col; };

static thread_local std::string g_err;

// getMaximumOccupancy() of the reference (bdd_preprocessor.cpp:21-30: device 0's resident threads / 10); MI355X's figure when no
// device is visible — these entry points are host code and must work on a box without a GPU
static size_t default_split_parallelism()
{
    uint64_t threads = 0;
    if (bddmma_device_chip(0, nullptr, nullptr, &threads) != BDDMMA_OK || threads == 0) threads = 256ull * 2048;
    return (size_t)(threads / 10);
}

template <typename F>
static int guarded(int fail_code, F&& f)
{
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        return g_err.find("infeasible") != std::string::npos ? BDDILP_ERR_INFEASIBLE : fail_code;
    }
}

extern "C" {

const char* bddilp_last_error(void) { return g_err.c_str(); }

int bddilp_parse_lp(const char* lp_text, bddilp** out)
{
    if (!lp_text || !out) { g_err = "null argument"; return BDDILP_ERR_INVALID_ARGUMENT; }
    return guarded(BDDILP_ERR_PARSE, [&] { *out = new bddilp{parse_lp(lp_text)}; return BDDILP_OK; });
}
int bddilp_parse_opb(const char* text, bddilp** out)
{
    if (!text || !out) { g_err = "null argument"; return BDDILP_ERR_INVALID_ARGUMENT; }
    return guarded(BDDILP_ERR_PARSE, [&] { *out = new bddilp{parse_opb(text)}; return BDDILP_OK; });
}
int bddilp_parse(const char* text, bddilp** out)
{
    if (!text || !out) { g_err = "null argument"; return BDDILP_ERR_INVALID_ARGUMENT; }
    return guarded(BDDILP_ERR_PARSE, [&] { *out = new bddilp{parse_lp_or_opb(text)}; return BDDILP_OK; });
}
void bddilp_destroy(bddilp* ilp) { delete ilp; }
uint64_t bddilp_nr_variables(const bddilp* ilp) { return ilp->ilp.nr_variables(); }
uint64_t bddilp_nr_constraints(const bddilp* ilp) { return ilp->ilp.constraints.size(); }
const char* bddilp_variable_name(const bddilp* ilp, uint64_t v) { return v < ilp->ilp.var_names.size() ? ilp->ilp.var_names[v].c_str() : ""; }
int bddilp_objective(const bddilp* ilp, double* objective, double* constant)
{
    if (objective) std::memcpy(objective, ilp->ilp.objective.data(), ilp->ilp.objective.size() * sizeof(double));
    if (constant) *constant = ilp->ilp.constant;
    return BDDILP_OK;
}
uint64_t bddilp_constraint_size(const bddilp* ilp, uint64_t c) { return c < ilp->ilp.constraints.size() ? ilp->ilp.constraints[c].variables.size() : 0; }
int bddilp_constraint(const bddilp* ilp, uint64_t c, int64_t* coeffs, uint64_t* vars, int* ineq, int64_t* rhs)
{
    if (c >= ilp->ilp.constraints.size()) { g_err = "constraint index out of range"; return BDDILP_ERR_INVALID_ARGUMENT; }
    const constraint& k = ilp->ilp.constraints[c];
    for (size_t i = 0; i < k.variables.size(); ++i) {
        if (coeffs) coeffs[i] = k.coefficients[i];
        if (vars) vars[i] = k.variables[i];
    }
    if (ineq) *ineq = (int)k.ineq;
    if (rhs) *rhs = k.rhs;
    return BDDILP_OK;
}
const char* bddilp_constraint_name(const bddilp* ilp, uint64_t c) { return c < ilp->ilp.constraints.size() ? ilp->ilp.constraints[c].name.c_str() : ""; }
int bddilp_normalize(bddilp* ilp) { ilp->ilp.normalize(); return BDDILP_OK; }

int bddilp_to_bdds(const bddilp* ilp, int split, uint64_t split_length, bddilp_bdds** out)
{
    if (!ilp || !out) { g_err = "null argument"; return BDDILP_ERR_INVALID_ARGUMENT; }
    return guarded(BDDILP_ERR_INVALID_ARGUMENT, [&] {
        auto* b = new bddilp_bdds{to_bdds(ilp->ilp)};
        if (split) b->col.split_long_bdds(std::max(b->col.nr_variables(), ilp->ilp.nr_variables()), split_length, default_split_parallelism(), split == 2);
        *out = b;
        return BDDILP_OK;
    });
}
int bddilp_bdds_create(bddilp_bdds** out) { *out = new bddilp_bdds{}; return BDDILP_OK; }
int bddilp_bdds_add_row(bddilp_bdds* b, const int64_t* coeffs, const uint64_t* vars, uint64_t n, int ineq, int64_t rhs, int* status)
{
    if (!b || !coeffs || !vars || n == 0 || ineq < -1 || ineq > 1) { g_err = "invalid row"; return BDDILP_ERR_INVALID_ARGUMENT; }
    return guarded(BDDILP_ERR_INVALID_ARGUMENT, [&] {
        constraint c;
        c.coefficients.assign(coeffs, coeffs + n);
        c.variables.assign(vars, vars + n);
        c.ineq = (ineq_t)ineq;
        c.rhs = rhs;
        if (std::set<size_t>(c.variables.begin(), c.variables.end()).size() != c.variables.size()) throw std::runtime_error("constraint repeats a variable");
        row_status st = row_status::ok;
        if (c.is_simplex()) b->col.add_simplex(c.variables);
        else st = b->col.add_linear(c.coefficients, c.ineq, c.rhs, c.variables);
        if (status) *status = (int)st;
        return BDDILP_OK;
    });
}
int bddilp_bdds_split(bddilp_bdds* b, uint64_t nr_variables, uint64_t split_length, int with_implication_bdd, uint64_t* nr_split,
                      uint64_t* nr_variables_after)
{
    return guarded(BDDILP_ERR_INVALID_ARGUMENT, [&] {
        const auto [n, nv] = b->col.split_long_bdds(nr_variables, split_length, default_split_parallelism(), with_implication_bdd != 0);
        if (nr_split) *nr_split = n;
        if (nr_variables_after) *nr_variables_after = nv;
        return BDDILP_OK;
    });
}
void bddilp_bdds_destroy(bddilp_bdds* b) { delete b; }
uint64_t bddilp_bdds_nr_bdds(const bddilp_bdds* b) { return b->col.nr_bdds(); }
uint64_t bddilp_bdds_nr_instructions(const bddilp_bdds* b) { return b->col.instructions.size(); }
uint64_t bddilp_bdds_nr_variables(const bddilp_bdds* b) { return b->col.nr_variables(); }
const bddmma_instruction* bddilp_bdds_instructions(const bddilp_bdds* b) { return b->col.instructions.data(); }
const uint64_t* bddilp_bdds_delimiters(const bddilp_bdds* b) { return b->col.delimiters.data(); }

// The emitters index instructions[lo / hi] and the per-BDD `incoming` tables directly: check what bddmma_create checks (monotone
// delimiters, >= 1 node + 2 terminals per BDD, terminals last, arcs of a non-terminal node point forward inside their BDD) before
// anything is written (out-of-bounds reads and writes on malformed input through a public C ABI).
static bddmma_host::bdd_store store_of(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds)
{
    for (uint64_t b = 0; b < n_bdds; ++b) {
        const uint64_t d0 = delims[b], d1 = delims[b + 1];
        if (d1 < d0 || d1 - d0 < 3) throw std::runtime_error("BDD " + std::to_string(b) + ": fewer than 3 instructions or delimiters not ascending");
        for (uint64_t i = d0; i < d1; ++i) {
            const bool term = instr[i].index >= BDDMMA_BOTSINK;
            if (term != (i + 2 >= d1)) throw std::runtime_error("BDD " + std::to_string(b) + ": the two terminals must be its last two instructions");
            if (!term && (instr[i].lo <= i || instr[i].lo >= d1 || instr[i].hi <= i || instr[i].hi >= d1))
                throw std::runtime_error("BDD " + std::to_string(b) + ": arc of instruction " + std::to_string(i) + " does not point forward inside the BDD");
        }
        if (instr[d1 - 2].index == instr[d1 - 1].index) throw std::runtime_error("BDD " + std::to_string(b) + ": needs one top and one bot sink");
    }
    bddmma_host::bdd_store c;
    c.delimiters.assign(delims, delims + n_bdds + 1);
    c.instructions.assign(instr, instr + delims[n_bdds]);
    return c;
}
int bddilp_write_bdd_lp(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds, const double* costs, uint64_t n_costs,
                        const char* path)
{
    return guarded(BDDILP_ERR_INVALID_ARGUMENT, [&] {
        if (!instr || !delims || !path || (!costs && n_costs)) throw std::runtime_error("null argument");
        const bdd_store store = store_of(instr, delims, n_bdds);  // validated before the file is opened (opening truncates it)
        std::ofstream f(path);
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        store.write_bdd_lp(f, std::vector<double>(costs, costs + n_costs));
        return BDDILP_OK;
    });
}
int bddilp_export_graphviz(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds, uint64_t bdd_nr, const char* path)
{
    return guarded(BDDILP_ERR_INVALID_ARGUMENT, [&] {
        if (!instr || !delims || !path) throw std::runtime_error("null argument");
        if (bdd_nr >= n_bdds) throw std::runtime_error("bdd_nr out of range");
        const bdd_store store = store_of(instr, delims, n_bdds);
        std::ofstream f(path);
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        store.export_graphviz(bdd_nr, f);
        return BDDILP_OK;
    });
}

}  // extern "C"
