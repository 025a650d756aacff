// bdd_store.hpp — host-side flat QBDD storage and constraint -> QBDD builders (C++17, no dependencies).
//
// The storage is the reference's BDD::bdd_collection interchange format
// (include/bdd_collection/bdd_collection.h:14-36,122-288): a flat vector of {lo, hi, index} with ABSOLUTE
// child indices, one delimiter per BDD, nodes grouped by variable in BDD order and the two terminals last.
// `bdd_store` offers the accessors the solver constructor uses (nr_bdds, nr_bdd_nodes(b), offset(b),
// operator()(b, i); bdd_cuda_base.cu:55-144), so it can be handed to bdd_hip_parallel_mma<REAL> exactly like
// a BDD::bdd_collection.  The builders restate, in closed form, what the reference produces with bdd_mgr +
// make_qbdd (bdd_collection.cpp:2039-2134, :1670-1812; bdd_preprocessor.cpp:165-226) and split_qbdd
// (bdd_collection.cpp:507-949); bdd_amd/bdd_collection.py is the same code in Python and both are pinned
// node-for-node against oracle/_ref by the tests.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <ostream>
#include <vector>

#include "../../../include/bdd_mma.h"

namespace bddmma_host {

enum class ineq_t : int { le = -1, eq = 0, ge = 1 };

enum class row_status : int { ok = 0, trivially_true = 1, infeasible = 2 };

class bdd_store {
public:
    std::vector<bddmma_instruction> instructions;
    std::vector<uint64_t> delimiters{0};

    // --- the accessors of BDD::bdd_collection the solver constructor needs
    size_t nr_bdds() const { return delimiters.size() - 1; }
    size_t nr_bdd_nodes() const { return instructions.size(); }
    size_t nr_bdd_nodes(size_t b) const { return delimiters[b + 1] - delimiters[b]; }
    size_t offset(size_t b) const { return delimiters[b]; }
    // as bdd_collection::operator()(bdd_nr, offset): `offset` is ABSOLUTE, offset(b) <= offset < offset(b + 1) (bdd_collection.cpp:1538-1544)
    const bddmma_instruction& operator()(size_t /*b*/, size_t offset) const { return instructions[offset]; }

    size_t nr_variables() const;                      // 1 + largest variable index
    std::vector<size_t> variables(size_t b) const;    // in BDD order
    std::vector<size_t> layer_widths(size_t b) const; // nodes per variable layer (bdd_collection.h:195)
    bool evaluate(size_t b, const std::vector<char>& x) const;  // bdd_collection.h:293-311

    // --- builders; every one appends one BDD and returns its number
    size_t add_simplex(const std::vector<size_t>& vars);   // sum x = 1   (bdd_collection.cpp:2039-2103)
    size_t add_covering(const std::vector<size_t>& vars);  // sum x >= 1  (:2105-2134 followed by make_qbdd)
    // sum a_i x_i {<=,=,>=} rhs as the canonical QBDD over `vars` in the given order; variables the function
    // does not depend on are dropped, as in the reference's reduced BDDs.  `bdd_nr` is set when status == ok.
    row_status add_linear(const std::vector<long>& coeffs, ineq_t ineq, long rhs, const std::vector<size_t>& vars, size_t* bdd_nr = nullptr);

    // --- long-BDD splitting (bdd_collection.cpp:507-949, bdd_preprocessor.cpp:372-415)
    // Appends the chunks of BDD b (the original stays); returns the new BDD numbers and the next free aux variable.
    // with_implication_bdd: additionally append the BDD over the auxiliary variables that encodes which nodes of different
    // cuts are connected (bdd_collection.cpp:801-941) when the split has more than two chunks and such implications exist
    std::pair<std::vector<size_t>, size_t> split_qbdd(size_t b, size_t chunk_size, size_t aux_var_start, bool with_implication_bdd = false);
    void remove(std::vector<size_t> bdd_nrs);
    // splits every BDD with more than split_length variables (0: compute_split_length); returns {#split, #variables afterwards}
    std::pair<size_t, size_t> split_long_bdds(size_t nr_variables, size_t split_length, size_t parallelism = 256 * 2048 / 10,
                                              bool with_implication_bdd = false);
    size_t compute_split_length(size_t parallelism) const;  // bdd_preprocessor.cpp:32-121

    // --- text exports ("export bdd lp" / "export bdd graph" of the driver, bdd_solver.cpp:400-410, :432-462)
    // the network-flow LP over the arcs of all BDDs, linked by the original variables (bdd_collection.h:731-830); `costs[v]` = objective of x_v
    void write_bdd_lp(std::ostream& s, const std::vector<double>& costs) const;
    // one BDD as a Graphviz digraph, one cluster per variable (bdd_collection.h:663-729)
    void export_graphviz(size_t b, std::ostream& s) const;

private:
    // append one BDD given LOCAL child indices; TOP_LOCAL / BOT_LOCAL mark the sinks
    static constexpr long TOP_LOCAL = -1, BOT_LOCAL = -2;
    size_t append_local(const std::vector<long>& lo, const std::vector<long>& hi, const std::vector<size_t>& var, bool top_first);
    // returns false when there is no implication between the cuts (nothing appended)
    bool append_implication_bdd(const std::vector<bddmma_instruction>& src, size_t off, size_t top_abs, size_t bot_abs,
                                const std::vector<size_t>& widths, const std::vector<size_t>& loff, size_t chunk, size_t n_chunks,
                                const std::vector<size_t>& aux);
};

}  // namespace bddmma_host
