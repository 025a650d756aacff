// ilp.hpp — 0/1 ILP model, reader for the .lp subset the reference parses, ILP -> QBDDs (host C++17).
//
// Mirrors LPMP::ILP_input and the PEGTL grammar of src/ILP/ILP_parser.cpp:24-140 (Minimize, objective terms,
// Subject To, optionally named rows, Bounds = variable fixations, Binaries / Generals lists skipped, End; products
// of variables in a row are refused); variable indices are assigned
// in order of first appearance, objective first (ILP_parser.cpp:246-254, :316-327).  to_bdds() is
// bdd_preprocessor::add_ilp for linear rows (src/bdd_conversion/bdd_preprocessor.cpp:123-336).
// bdd_amd/ilp.py is the same reader in Python; tests/test_native_host.py checks they agree.
#pragma once
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "bdd_store.hpp"

namespace bddmma_host {

struct constraint {
    std::vector<long> coefficients;
    std::vector<size_t> variables;
    ineq_t ineq = ineq_t::le;
    long rhs = 0;
    std::string name;
    // ILP_input::constraint::is_simplex, src/ILP/ILP_input.cpp:81-93
    bool is_simplex() const;
};

class ilp_input {
public:
    std::vector<std::string> var_names;
    std::vector<double> objective;
    std::vector<constraint> constraints;
    double constant = 0.0;

    size_t nr_variables() const { return var_names.size(); }
    size_t var(const std::string& name);  // index of `name`, created on first use
    bool has_var(const std::string& name) const { return index_.count(name) != 0; }
    size_t var_index(const std::string& name) const { return index_.at(name); }
    // ILP_input::reduce (src/ILP/ILP_input.cpp:508-591): the model without the fixed variables — a variable fixed to 1
    // moves its objective coefficient into the constant and its row coefficients to the right-hand sides; a row that
    // loses every term is checked (`0 <rel> rhs`, throws if violated) and dropped.  Remaining variables keep their order.
    ilp_input reduce(const std::set<size_t>& zeros, const std::set<size_t>& ones) const;
    double evaluate(const std::vector<char>& x) const;
    bool feasible(const std::vector<char>& x) const;
    std::string write_lp() const;
    void normalize();  // ILP_input::constraint::normalize: monomials sorted by variable

private:
    std::unordered_map<std::string, size_t> index_;
};

// throws std::runtime_error with the offending text on malformed input
ilp_input parse_lp(const std::string& text);
// The OPB (pseudo-Boolean) subset of src/ILP/OPB_parser.cpp:23-60: leading `* comment` lines, `min: <terms> ;`, then
// `<terms> {<=,>=,=} <integer> ;` rows (a row may span lines); whatever follows the last `;` is ignored.
ilp_input parse_opb(const std::string& text);
// bdd_solver::read_ILP (bdd_solver.cpp:44-66): try the .lp grammar first, then OPB
ilp_input parse_lp_or_opb(const std::string& text);

// Rows that are trivially true are skipped (bdd_preprocessor.cpp:213-214); an infeasible row throws (:215-216).
bdd_store to_bdds(const ilp_input& ilp);

}  // namespace bddmma_host
