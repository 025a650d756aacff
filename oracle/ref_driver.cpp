/*
 * ref_driver.cpp — driver around the REFERENCE's own code, compiled where it lies.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libref_driver.so
 * together with these unmodified reference translation units:
 *     /root/reference/src/bdd_collection/{bdd_collection,transitive_closure_dag}.cpp
 *     /root/reference/src/bdd_manager/*.cpp
 * and the reference header include/bdd_solver/bdd_branch_instruction.h (node arithmetic:
 * forward_step / backward_step / min_marginals).
 *
 * What comes from the reference here:
 *   - BDD construction: simplex_constraint, not_all_false_constraint (+make_qbdd),
 *     cardinality_constraint, all_equal_constraint, bdd_mgr synthesis + add_bdd + reorder +
 *     make_qbdd + rebase, exactly the call sequence of bdd_preprocessor::add_ilp
 *     (src/bdd_conversion/bdd_preprocessor.cpp:172-226);
 *   - per-node arithmetic of the CPU parallel-mma solver (bdd_branch_instruction.h:98-198).
 * What does NOT come from the reference: the loops over BDDs / layers in ref_mma below restate
 * src/bdd_solver/bdd_parallel_mma_base.cpp (add_bdds :75-170, update_costs :626-696,
 * forward_mm :814-889, backward_mm :891-956, iteration :1011-1044), because that translation
 * unit includes <Eigen/SparseCore>, which this image lacks, and is therefore unbuildable here.
 */
#include "bdd_collection/bdd_collection.h"
#include "bdd_manager/bdd_mgr.h"
#include "bdd_solver/bdd_branch_instruction.h"
// the drop-in class of the product, for ONE purpose: its template constructor's flattening loop is instantiated below with the
// reference's real BDD::bdd_collection; no bddmma_* entry point is called from here (nothing of the product is linked)
#include "../bdd_amd/csrc/bdd_hip_parallel_mma.hpp"

#include <sstream>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <vector>

using namespace BDD;

namespace {

size_t finish_bdd(bdd_collection& col, size_t bdd_nr, const size_t* vars, size_t n)
{
    // bdd_preprocessor.cpp:217-221
    if (!col.is_reordered(bdd_nr)) col.reorder(bdd_nr);
    if (!col.is_qbdd(bdd_nr)) {
        col.make_qbdd(bdd_nr);
        col.remove(bdd_nr);
    }
    col.rebase(bdd_nr, vars, vars + n);
    return bdd_nr;
}

template <typename REAL>
struct ref_mma {
    using node_t = LPMP::bdd_branch_instruction<REAL, uint16_t>;
    std::vector<node_t> nodes;
    std::vector<size_t> bdd_layer_ptr, layer_node_ptr, layer_var, nr_bdds_per_var;
    std::vector<REAL> delta_in, delta_out;
    bool after_backward = false;

    explicit ref_mma(const bdd_collection& col)
    {
        size_t nr_vars = 0;
        for (size_t b = 0; b < col.nr_bdds(); ++b) nr_vars = std::max(nr_vars, col.min_max_variables(b)[1] + 1);
        nr_bdds_per_var.assign(nr_vars, 0);
        for (size_t b = 0; b < col.nr_bdds(); ++b) {
            bdd_layer_ptr.push_back(layer_var.size());
            size_t prev = std::numeric_limits<size_t>::max() - 5;
            for (auto it = col.cbegin(b); it != col.cend(b); ++it) {
                const bdd_instruction& st = *it;
                if (st.is_terminal()) continue;
                node_t n;
                const auto& lo = col.get_bdd_instruction(st.lo);
                const auto& hi = col.get_bdd_instruction(st.hi);
                if (lo.is_botsink()) n.offset_low = node_t::terminal_0_offset;
                else if (lo.is_topsink()) n.offset_low = node_t::terminal_1_offset;
                else n.offset_low = st.lo - col.offset(st);
                if (hi.is_botsink()) n.offset_high = node_t::terminal_0_offset;
                else if (hi.is_topsink()) n.offset_high = node_t::terminal_1_offset;
                else n.offset_high = st.hi - col.offset(st);
                if (n.offset_low == node_t::terminal_0_offset) n.low_cost = std::numeric_limits<REAL>::infinity();
                if (n.offset_high == node_t::terminal_0_offset) n.high_cost = std::numeric_limits<REAL>::infinity();
                if (st.index != prev) {
                    layer_node_ptr.push_back(nodes.size());
                    layer_var.push_back(st.index);
                    nr_bdds_per_var[st.index]++;
                    prev = st.index;
                }
                nodes.push_back(n);
            }
        }
        bdd_layer_ptr.push_back(layer_var.size());
        layer_node_ptr.push_back(nodes.size());
        delta_in.assign(2 * nr_vars, 0);
        delta_out.assign(2 * nr_vars, 0);
    }

    void update_costs(const double* lo, size_t n_lo, const double* hi, size_t n_hi)
    {
        after_backward = false;
        for (size_t l = 0; l < layer_var.size(); ++l) {
            const size_t var = layer_var[l];
            const double lc = (var < n_lo) ? lo[var] / double(nr_bdds_per_var[var]) : 0.0;
            const double hc = (var < n_hi) ? hi[var] / double(nr_bdds_per_var[var]) : 0.0;
            for (size_t i = layer_node_ptr[l]; i < layer_node_ptr[l + 1]; ++i) {
                if (nodes[i].offset_low != node_t::terminal_0_offset) nodes[i].low_cost += lc;
                if (nodes[i].offset_high != node_t::terminal_0_offset) nodes[i].high_cost += hc;
            }
        }
    }

    void backward_run()
    {
        if (after_backward) return;
        for (size_t b = 0; b + 1 < bdd_layer_ptr.size(); ++b) {
            const size_t first = layer_node_ptr[bdd_layer_ptr[b]], last = layer_node_ptr[bdd_layer_ptr[b + 1]];
            for (std::ptrdiff_t i = last - 1; i >= std::ptrdiff_t(first); --i) nodes[i].backward_step();
        }
        after_backward = true;
    }

    double lower_bound()
    {
        backward_run();
        double lb = 0.0;
        for (size_t b = 0; b + 1 < bdd_layer_ptr.size(); ++b) lb += nodes[layer_node_ptr[bdd_layer_ptr[b]]].m;
        return lb;
    }

    void layer_update(size_t l, REAL omega, std::vector<REAL>& dout)
    {
        const size_t first = layer_node_ptr[l], last = layer_node_ptr[l + 1], var = layer_var[l];
        std::array<REAL, 2> cur = {std::numeric_limits<REAL>::infinity(), std::numeric_limits<REAL>::infinity()};
        for (size_t i = first; i < last; ++i) {
            const auto mm = nodes[i].min_marginals();
            cur[0] = std::min(mm[0], cur[0]);
            cur[1] = std::min(mm[1], cur[1]);
        }
        const bool f0 = std::isfinite(cur[0]), f1 = std::isfinite(cur[1]);
        if (!f0) dout[2 * var] = std::numeric_limits<REAL>::infinity();
        if (!f1) dout[2 * var + 1] = std::numeric_limits<REAL>::infinity();
        if (f0 && f1) {
            if (cur[0] < cur[1]) dout[2 * var + 1] += omega * (cur[1] - cur[0]);
            else dout[2 * var] += omega * (cur[0] - cur[1]);
        }
        for (size_t i = first; i < last; ++i) {
            if (!f0) nodes[i].low_cost = std::numeric_limits<REAL>::infinity();
            if (!f1) nodes[i].high_cost = std::numeric_limits<REAL>::infinity();
            if (f0 && f1) {
                if (cur[0] < cur[1]) nodes[i].high_cost += omega * (cur[0] - cur[1]);
                else nodes[i].low_cost += omega * (cur[1] - cur[0]);
            }
        }
    }

    // delta: in = values to add, out = raw sums (the std::swap of :979)
    void forward_mm(REAL omega, REAL* delta)
    {
        backward_run();
        std::fill(delta_out.begin(), delta_out.end(), REAL(0));
        for (size_t b = 0; b + 1 < bdd_layer_ptr.size(); ++b) {
            nodes[layer_node_ptr[bdd_layer_ptr[b]]].m = 0.0;
            for (size_t l = bdd_layer_ptr[b]; l < bdd_layer_ptr[b + 1]; ++l) {
                layer_update(l, omega, delta_out);
                const size_t first = layer_node_ptr[l], last = layer_node_ptr[l + 1], var = layer_var[l];
                if (l + 1 < bdd_layer_ptr[b + 1])
                    for (size_t i = layer_node_ptr[l + 1]; i < layer_node_ptr[l + 2]; ++i)
                        nodes[i].m = std::numeric_limits<REAL>::infinity();
                for (size_t i = first; i < last; ++i) {
                    nodes[i].low_cost += delta[2 * var];
                    nodes[i].high_cost += delta[2 * var + 1];
                    nodes[i].forward_step();
                }
            }
        }
        std::memcpy(delta, delta_out.data(), delta_out.size() * sizeof(REAL));
        after_backward = false;
    }

    double backward_mm(REAL omega, REAL* delta)
    {
        std::fill(delta_out.begin(), delta_out.end(), REAL(0));
        double lb = 0.0;
        for (size_t b = 0; b + 1 < bdd_layer_ptr.size(); ++b) {
            for (std::ptrdiff_t l = bdd_layer_ptr[b + 1] - 1; l >= std::ptrdiff_t(bdd_layer_ptr[b]); --l) {
                layer_update(l, omega, delta_out);
                const size_t first = layer_node_ptr[l], last = layer_node_ptr[l + 1], var = layer_var[l];
                for (std::ptrdiff_t i = last - 1; i >= std::ptrdiff_t(first); --i) {
                    nodes[i].low_cost += delta[2 * var];
                    nodes[i].high_cost += delta[2 * var + 1];
                    nodes[i].backward_step();
                }
            }
            lb += nodes[layer_node_ptr[bdd_layer_ptr[b]]].m;
        }
        std::memcpy(delta, delta_out.data(), delta_out.size() * sizeof(REAL));
        after_backward = true;
        return lb;
    }

    void average(REAL* d)
    {
        for (size_t v = 0; v < nr_bdds_per_var.size(); ++v)
            if (nr_bdds_per_var[v] > 0) {
                d[2 * v] /= REAL(nr_bdds_per_var[v]);
                d[2 * v + 1] /= REAL(nr_bdds_per_var[v]);
            }
    }

    double iteration()
    {
        backward_run();
        forward_mm(0.5, delta_in.data());
        average(delta_in.data());
        const double lb = backward_mm(0.5, delta_in.data());
        average(delta_in.data());
        return lb;
    }
};

}  // namespace

extern "C" {

void* ref_col_new() { return new bdd_collection(); }
void ref_col_free(void* c) { delete static_cast<bdd_collection*>(c); }

long ref_col_add_simplex(void* c, size_t n, const size_t* vars)
{
    auto& col = *static_cast<bdd_collection*>(c);
    return finish_bdd(col, col.simplex_constraint(n), vars, n);
}
long ref_col_add_covering(void* c, size_t n, const size_t* vars)
{
    auto& col = *static_cast<bdd_collection*>(c);
    return finish_bdd(col, col.not_all_false_constraint(n), vars, n);
}
long ref_col_add_cardinality(void* c, size_t n, size_t k, const size_t* vars)
{
    auto& col = *static_cast<bdd_collection*>(c);
    return finish_bdd(col, col.cardinality_constraint(n, k), vars, n);
}
long ref_col_add_all_equal(void* c, size_t n, const size_t* vars)
{
    auto& col = *static_cast<bdd_collection*>(c);
    return finish_bdd(col, col.all_equal_constraint(n), vars, n);
}

// sum coeffs[i] x_i  (ineq: -1 "<=", 0 "=", 1 ">=")  rhs, synthesised with the reference's bdd_mgr.
// returns -1 if the constraint is trivially true, -2 if infeasible.
long ref_col_add_linear(void* c, size_t n, const int* coeffs, int ineq, int rhs, const size_t* vars)
{
    auto& col = *static_cast<bdd_collection*>(c);
    bdd_mgr mgr;
    for (size_t i = 0; i < n; ++i) mgr.add_variable();
    std::map<std::pair<size_t, long>, node_ref> memo;
    std::function<node_ref(size_t, long)> rec = [&](size_t i, long s) -> node_ref {
        if (i == n) {
            const bool ok = ineq < 0 ? (s <= rhs) : (ineq == 0 ? (s == rhs) : (s >= rhs));
            return ok ? mgr.topsink() : mgr.botsink();
        }
        auto key = std::make_pair(i, s);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        node_ref lo = rec(i + 1, s);
        node_ref hi = rec(i + 1, s + coeffs[i]);
        node_ref r = mgr.ite_rec(mgr.projection(i), hi, lo);
        memo.insert({key, r});
        return r;
    };
    node_ref f = rec(0, 0);
    if (f.is_topsink()) return -1;
    if (f.is_botsink()) return -2;
    const size_t bdd_nr = col.add_bdd(f);
    return finish_bdd(col, bdd_nr, vars, n);
}

// bdd_collection::split_qbdd (bdd_collection.cpp:507-949) on BDD `bdd_nr`, then remove the original as
// bdd_preprocessor.cpp:393-412 does.  Returns the number of BDDs the split produced (1 = not split, nothing
// removed); *next_aux receives the next free auxiliary variable.
long ref_col_split_qbdd(void* c, size_t bdd_nr, size_t chunk_size, size_t aux_var_start, int with_implication,
                        size_t* next_aux)
{
    auto& col = *static_cast<bdd_collection*>(c);
    const auto [new_nrs, na] = col.split_qbdd(bdd_nr, chunk_size, aux_var_start, with_implication != 0);
    *next_aux = na;
    if (new_nrs.size() > 1) col.remove(bdd_nr);
    return (long)new_nrs.size();
}

size_t ref_col_nr_bdds(void* c) { return static_cast<bdd_collection*>(c)->nr_bdds(); }
size_t ref_col_nr_instructions(void* c)
{
    auto& col = *static_cast<bdd_collection*>(c);
    size_t n = 0;
    for (size_t b = 0; b < col.nr_bdds(); ++b) n += col.nr_bdd_nodes(b);
    return n;
}
int ref_col_is_qbdd(void* c, size_t b) { return static_cast<bdd_collection*>(c)->is_qbdd(b); }

void ref_col_export(void* c, uint64_t* instr, uint64_t* delims)
{
    auto& col = *static_cast<bdd_collection*>(c);
    size_t k = 0;
    for (size_t b = 0; b < col.nr_bdds(); ++b) {
        delims[b] = k;
        const size_t off = col.offset(b);
        const bdd_instruction* it = col.cbegin(b);  // cend() stops before the two terminals
        for (size_t i = 0; i < col.nr_bdd_nodes(b); ++i, ++it, ++k) {
            // re-base absolute indices to the exported (dense) array
            instr[3 * k + 0] = it->is_terminal() ? it->lo : it->lo - off + delims[b];
            instr[3 * k + 1] = it->is_terminal() ? it->hi : it->hi - off + delims[b];
            instr[3 * k + 2] = it->index;
        }
    }
    delims[col.nr_bdds()] = k;
}

// the same export through the product's own template code: LPMP::bdd_hip_parallel_mma<REAL>::flatten(const BDD::bdd_collection&), the loop
// its constructor runs before bddmma_create.  Returns the number of instructions written (instr: 3 words each; delims: nr_bdds + 1).
size_t ref_col_flatten_dropin(void* c, uint64_t* instr, uint64_t* delims)
{
    const bdd_collection& col = *static_cast<bdd_collection*>(c);
    std::vector<bddmma_instruction> in;
    std::vector<uint64_t> de;
    LPMP::bdd_hip_parallel_mma<float>::flatten(col, in, de);
    static_assert(sizeof(bddmma_instruction) == 3 * sizeof(uint64_t), "bddmma_instruction is {lo, hi, index}");
    std::memcpy(instr, in.data(), in.size() * sizeof(bddmma_instruction));
    std::memcpy(delims, de.data(), de.size() * sizeof(uint64_t));
    return in.size();
}

// text exports of the reference's bdd_collection (bdd_collection.h:663-830), for tests/golden/exports.json: the text is copied into
// `out` (capacity `cap`); returns the length the whole text has
static size_t copy_text(const std::string& t, char* out, size_t cap)
{
    if (out && cap) {
        const size_t n = std::min(cap - 1, t.size());
        std::memcpy(out, t.data(), n);
        out[n] = 0;
    }
    return t.size();
}
size_t ref_col_write_bdd_lp(void* c, const double* costs, size_t n, char* out, size_t cap)
{
    std::ostringstream s;
    static_cast<bdd_collection*>(c)->write_bdd_lp(s, costs, costs + n);
    return copy_text(s.str(), out, cap);
}
size_t ref_col_export_graphviz(void* c, size_t bdd_nr, char* out, size_t cap)
{
    std::ostringstream s;
    static_cast<bdd_collection*>(c)->export_graphviz(bdd_nr, s);
    return copy_text(s.str(), out, cap);
}

#define MMA_API(SUF, REAL)                                                                                   \
    void* ref_mma_new_##SUF(void* c) { return new ref_mma<REAL>(*static_cast<bdd_collection*>(c)); }          \
    void ref_mma_free_##SUF(void* m) { delete static_cast<ref_mma<REAL>*>(m); }                               \
    size_t ref_mma_nr_variables_##SUF(void* m) { return static_cast<ref_mma<REAL>*>(m)->nr_bdds_per_var.size(); } \
    void ref_mma_update_costs_##SUF(void* m, const double* lo, size_t nlo, const double* hi, size_t nhi)     \
    { static_cast<ref_mma<REAL>*>(m)->update_costs(lo, nlo, hi, nhi); }                                       \
    double ref_mma_lower_bound_##SUF(void* m) { return static_cast<ref_mma<REAL>*>(m)->lower_bound(); }       \
    double ref_mma_iteration_##SUF(void* m) { return static_cast<ref_mma<REAL>*>(m)->iteration(); }           \
    void ref_mma_forward_mm_##SUF(void* m, REAL omega, REAL* d) { static_cast<ref_mma<REAL>*>(m)->forward_mm(omega, d); } \
    double ref_mma_backward_mm_##SUF(void* m, REAL omega, REAL* d) { return static_cast<ref_mma<REAL>*>(m)->backward_mm(omega, d); }

MMA_API(f32, float)
MMA_API(f64, double)

}  // extern "C"
