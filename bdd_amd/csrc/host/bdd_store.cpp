// bdd_store.cpp — see bdd_store.hpp.  Host-only C++17.
#include "bdd_store.hpp"
#include <cmath>
#include <string>
#include <limits>
#include <array>
#include <unordered_map>

#include <algorithm>
#include <cassert>
#include <map>
#include <set>

namespace bddmma_host {

static inline bool is_terminal(const bddmma_instruction& i) { return i.index >= BDDMMA_BOTSINK; }

size_t bdd_store::nr_variables() const
{
    uint64_t m = 0;
    bool any = false;
    for (const auto& i : instructions)
        if (!is_terminal(i)) {
            m = std::max(m, i.index);
            any = true;
        }
    return any ? m + 1 : 0;
}

std::vector<size_t> bdd_store::variables(size_t b) const
{
    std::vector<size_t> out;
    for (size_t i = delimiters[b]; i + 2 < delimiters[b + 1]; ++i)
        if (out.empty() || out.back() != instructions[i].index) out.push_back(instructions[i].index);
    return out;
}

std::vector<size_t> bdd_store::layer_widths(size_t b) const
{
    std::vector<size_t> out;
    uint64_t prev = BDDMMA_TOPSINK;
    for (size_t i = delimiters[b]; i + 2 < delimiters[b + 1]; ++i) {
        if (out.empty() || instructions[i].index != prev) out.push_back(0);
        ++out.back();
        prev = instructions[i].index;
    }
    return out;
}

bool bdd_store::evaluate(size_t b, const std::vector<char>& x) const
{
    size_t i = delimiters[b];
    for (;;) {
        const auto& n = instructions[i];
        if (n.index == BDDMMA_TOPSINK) return true;
        if (n.index == BDDMMA_BOTSINK) return false;
        i = x[n.index] ? n.hi : n.lo;
    }
}

size_t bdd_store::append_local(const std::vector<long>& lo, const std::vector<long>& hi, const std::vector<size_t>& var, bool top_first)
{
    const size_t n = var.size(), base = instructions.size();
    const size_t top = base + n + (top_first ? 0 : 1), bot = base + n + (top_first ? 1 : 0);
    auto abs = [&](long c) -> uint64_t { return c == TOP_LOCAL ? top : c == BOT_LOCAL ? bot : base + (size_t)c; };
    for (size_t i = 0; i < n; ++i) instructions.push_back({abs(lo[i]), abs(hi[i]), var[i]});
    const bddmma_instruction t{BDDMMA_TOPSINK, BDDMMA_TOPSINK, BDDMMA_TOPSINK}, f{BDDMMA_BOTSINK, BDDMMA_BOTSINK, BDDMMA_BOTSINK};
    instructions.push_back(top_first ? t : f);
    instructions.push_back(top_first ? f : t);
    delimiters.push_back(instructions.size());
    return nr_bdds() - 1;
}

// Two states per level after the first variable: {sum == 0, sum == 1} (simplex) or {uncovered, covered}.
static void two_state_chain(size_t n, bool covering, std::vector<long>& lo, std::vector<long>& hi, std::vector<size_t>& layer)
{
    constexpr long T = -1, B = -2;
    const size_t nn = 2 * n - 1;
    lo.assign(nn, 0); hi.assign(nn, 0); layer.assign(nn, 0);
    lo[0] = 1; hi[0] = 2;
    for (size_t i = 1; i + 1 < n; ++i) {
        const size_t a = 2 * i - 1, c = 2 * i;
        lo[a] = 2 * i + 1; hi[a] = 2 * i + 2;
        lo[c] = 2 * i + 2; hi[c] = covering ? (long)(2 * i + 2) : B;
        layer[a] = layer[c] = i;
    }
    const size_t a = 2 * n - 3, c = 2 * n - 2;
    lo[a] = B; hi[a] = T;
    lo[c] = T; hi[c] = covering ? T : B;
    layer[a] = layer[c] = n - 1;
}

size_t bdd_store::add_simplex(const std::vector<size_t>& vars)
{
    assert(!vars.empty());
    if (vars.size() == 1) return append_local({BOT_LOCAL}, {TOP_LOCAL}, vars, false);
    std::vector<long> lo, hi;
    std::vector<size_t> layer, var;
    two_state_chain(vars.size(), false, lo, hi, layer);
    for (size_t l : layer) var.push_back(vars[l]);
    return append_local(lo, hi, var, false);
}

size_t bdd_store::add_covering(const std::vector<size_t>& vars)
{
    assert(!vars.empty());
    // a single-node not_all_false BDD already is a QBDD: make_qbdd is skipped and the sinks stay (bot, top)
    if (vars.size() == 1) return append_local({BOT_LOCAL}, {TOP_LOCAL}, vars, false);
    std::vector<long> lo, hi;
    std::vector<size_t> layer, var;
    two_state_chain(vars.size(), true, lo, hi, layer);
    for (size_t l : layer) var.push_back(vars[l]);
    return append_local(lo, hi, var, true);  // make_qbdd output has the top sink first
}

row_status bdd_store::add_linear(const std::vector<long>& coeffs, ineq_t ineq, long rhs, const std::vector<size_t>& vars, size_t* bdd_nr)
{
    std::vector<long> a = coeffs;
    std::vector<size_t> vs = vars;
    assert(a.size() == vs.size() && !a.empty());
    auto sat = [&](long s) { return ineq == ineq_t::le ? s <= rhs : ineq == ineq_t::eq ? s == rhs : s >= rhs; };
    constexpr long T = TOP_LOCAL, B = BOT_LOCAL;
    for (;;) {
        const size_t n = a.size();
        // reachable partial sums per level (sorted)
        std::vector<std::vector<long>> levels(n + 1);
        levels[0] = {0};
        for (size_t i = 0; i < n; ++i) {
            std::vector<long> nxt;
            nxt.reserve(2 * levels[i].size());
            for (long s : levels[i]) { nxt.push_back(s); nxt.push_back(s + a[i]); }
            std::sort(nxt.begin(), nxt.end());
            nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
            levels[i + 1] = std::move(nxt);
        }
        bool all_true = true;
        for (long s : levels[n]) all_true = all_true && sat(s);
        // bottom-up canonical ids per partial sum: T / B or an index into the level's table of (lo, hi) pairs
        std::map<long, long> cur;
        for (long s : levels[n]) cur[s] = sat(s) ? T : B;
        std::vector<std::vector<std::pair<long, long>>> tables(n);
        for (size_t i = n; i-- > 0;) {
            std::map<std::pair<long, long>, long> uniq;
            std::map<long, long> nw;
            for (long s : levels[i]) {  // ascending partial sums: first-seen order of the pairs is deterministic
                const std::pair<long, long> key{cur.at(s), cur.at(s + a[i])};
                if (key.first == B && key.second == B) { nw[s] = B; continue; }
                auto it = uniq.find(key);
                if (it == uniq.end()) {
                    it = uniq.emplace(key, (long)tables[i].size()).first;
                    tables[i].push_back(key);
                }
                nw[s] = it->second;
            }
            cur = std::move(nw);
        }
        const long root = cur.at(0);
        if (root == B) return row_status::infeasible;
        if (all_true) return row_status::trivially_true;
        // nodes reachable from the root
        std::vector<std::set<long>> reach(n);
        reach[0].insert(root);
        for (size_t i = 0; i + 1 < n; ++i)
            for (long k : reach[i]) {
                if (tables[i][k].first >= 0) reach[i + 1].insert(tables[i][k].first);
                if (tables[i][k].second >= 0) reach[i + 1].insert(tables[i][k].second);
            }
        // variables the function does not depend on (lo == hi on every node of the level) are not in the reference's BDD
        std::vector<size_t> keep;
        for (size_t i = 0; i < n; ++i) {
            bool dead = true;
            for (long k : reach[i]) dead = dead && tables[i][k].first == tables[i][k].second;
            if (!dead) keep.push_back(i);
        }
        if (keep.size() != n) {
            std::vector<long> a2;
            std::vector<size_t> v2;
            for (size_t i : keep) { a2.push_back(a[i]); v2.push_back(vs[i]); }
            a.swap(a2); vs.swap(v2);
            if (a.empty()) return row_status::trivially_true;  // cannot happen for a non-constant function; defensive
            continue;
        }
        std::vector<std::map<long, size_t>> remap(n);
        std::vector<size_t> offs(n + 1, 0);
        for (size_t i = 0; i < n; ++i) {
            size_t j = 0;
            for (long k : reach[i]) remap[i][k] = j++;
            offs[i + 1] = offs[i] + reach[i].size();
        }
        std::vector<long> lo(offs[n]), hi(offs[n]);
        std::vector<size_t> var(offs[n]);
        for (size_t i = 0; i < n; ++i)
            for (const auto& [k, j] : remap[i]) {
                const auto [l, h] = tables[i][k];
                lo[offs[i] + j] = l < 0 ? l : (long)(offs[i + 1] + remap[i + 1].at(l));
                hi[offs[i] + j] = h < 0 ? h : (long)(offs[i + 1] + remap[i + 1].at(h));
                var[offs[i] + j] = vs[i];
            }
        const size_t nr = append_local(lo, hi, var, false);
        if (bdd_nr) *bdd_nr = nr;
        return row_status::ok;
    }
}

// ------------------------------------------------------------------------------------------ splitting
void bdd_store::remove(std::vector<size_t> bdd_nrs)
{
    std::sort(bdd_nrs.begin(), bdd_nrs.end());
    bdd_nrs.erase(std::unique(bdd_nrs.begin(), bdd_nrs.end()), bdd_nrs.end());
    if (bdd_nrs.empty()) return;
    std::vector<bddmma_instruction> out;
    std::vector<uint64_t> nd{0};
    out.reserve(instructions.size());
    size_t next = 0;
    for (size_t b = 0; b < nr_bdds(); ++b) {
        if (next < bdd_nrs.size() && bdd_nrs[next] == b) { ++next; continue; }
        const int64_t shift = (int64_t)out.size() - (int64_t)delimiters[b];
        for (size_t i = delimiters[b]; i < delimiters[b + 1]; ++i) {
            bddmma_instruction x = instructions[i];
            if (!is_terminal(x)) { x.lo = (uint64_t)((int64_t)x.lo + shift); x.hi = (uint64_t)((int64_t)x.hi + shift); }
            out.push_back(x);
        }
        nd.push_back(out.size());
    }
    instructions.swap(out);
    delimiters.swap(nd);
}

std::pair<std::vector<size_t>, size_t> bdd_store::split_qbdd(size_t b, size_t chunk, size_t aux0, bool with_implication_bdd)
{
    assert(chunk > 0);
    const size_t off = delimiters[b], n_nodes = delimiters[b + 1] - off - 2;
    const std::vector<size_t> widths = layer_widths(b);
    const size_t n_layers = widths.size();
    if (n_layers <= chunk) return {{b}, aux0};
    std::vector<size_t> loff(n_layers + 1, 0);
    for (size_t l = 0; l < n_layers; ++l) loff[l + 1] = loff[l] + widths[l];
    const size_t n_chunks = (n_layers + chunk - 1) / chunk;
    std::vector<size_t> aux{aux0};
    for (size_t c = 1; c + 1 < n_chunks; ++c) aux.push_back(aux.back() + widths[c * chunk]);
    const std::vector<bddmma_instruction> src(instructions.begin() + off, instructions.begin() + off + n_nodes + 2);  // appends below reallocate
    const size_t top_abs = src[n_nodes].index == BDDMMA_TOPSINK ? off + n_nodes : off + n_nodes + 1;
    const size_t bot_abs = top_abs == off + n_nodes ? off + n_nodes + 1 : off + n_nodes;

    std::vector<size_t> new_nrs;
    for (size_t c = 0; c < n_chunks; ++c) {
        const size_t first = c * chunk, last = std::min((c + 1) * chunk, n_layers) - 1;
        const size_t w_head = c > 0 ? widths[first] : 0;
        const size_t w_tail = c + 1 < n_chunks ? widths[last + 1] : 0;
        const size_t n_head = w_head * (w_head + 1) / 2;
        const size_t n_chunk = loff[last + 1] - loff[first];
        const size_t n_tail = w_tail ? w_tail * (w_tail + 1) / 2 + w_tail - 1 : 0;
        std::vector<long> lo, hi;
        std::vector<size_t> var;
        auto emit = [&](size_t v, long l, long h) { var.push_back(v); lo.push_back(l); hi.push_back(h); };
        // head: row i has i+1 nodes; node (i, j) = "the j-th candidate is still open after a_0 .. a_{i-1}"
        auto head = [](size_t i, size_t j) { return (long)(i * (i + 1) / 2 + j); };
        for (size_t i = 0; i < w_head; ++i) {
            const size_t a = aux[c - 1] + i;
            const bool last_row = i + 1 == w_head;
            for (size_t j = 0; j <= i; ++j) {
                if (!last_row) {
                    if (j == 0) emit(a, head(i + 1, 0), head(i + 1, 1));
                    else emit(a, head(i + 1, j + 1), BOT_LOCAL);
                } else {
                    if (j == 0) emit(a, BOT_LOCAL, (long)(n_head + j));
                    else emit(a, (long)(n_head + j), BOT_LOCAL);
                }
            }
        }
        // the chunk's own nodes; children in layer last+1 land on the first row of the tail
        const long shift = (long)n_head - (long)loff[first];
        for (size_t u = loff[first]; u < loff[last + 1]; ++u) {
            auto child = [&](uint64_t x) -> long {
                return x == top_abs ? TOP_LOCAL : x == bot_abs ? BOT_LOCAL : (long)(x - off) + shift;
            };
            emit(src[u].index, child(src[u].lo), child(src[u].hi));
        }
        // tail: behind node j of the cut layer only the matching one-hot assignment of the cut's variables survives
        if (w_tail == 1) {
            emit(aux[c], BOT_LOCAL, TOP_LOCAL);  // (the reference asserts w_tail > 1, bdd_collection.cpp:583)
        } else if (w_tail > 1) {
            const size_t W = w_tail, base = n_head + n_chunk;
            auto tail = [&](size_t i, size_t j) {
                return (long)(base + (i == 0 ? j : W + W * (i - 1) + j - (i - 1) * (i - 2) / 2));
            };
            for (size_t j = 0; j < W; ++j) {
                if (j + 1 == W) emit(aux[c], BOT_LOCAL, tail(1, W - 1));
                else emit(aux[c], tail(1, j), BOT_LOCAL);
            }
            for (size_t i = 1; i + 1 < W; ++i) {
                const size_t n_row = W - i + 1;
                for (size_t j = 0; j < n_row; ++j) {
                    if (j + 1 == n_row) emit(aux[c] + i, tail(i + 1, W - i - 1), BOT_LOCAL);
                    else if (j + 2 == n_row) emit(aux[c] + i, BOT_LOCAL, tail(i + 1, j));
                    else emit(aux[c] + i, tail(i + 1, j), BOT_LOCAL);
                }
            }
            emit(aux[c] + W - 1, BOT_LOCAL, TOP_LOCAL);
            emit(aux[c] + W - 1, TOP_LOCAL, BOT_LOCAL);
        }
        assert(var.size() == n_head + n_chunk + n_tail);
        (void)n_tail;
        new_nrs.push_back(append_local(lo, hi, var, false));
    }
    if (with_implication_bdd && new_nrs.size() > 2 && append_implication_bdd(src, off, top_abs, bot_abs, widths, loff, chunk, n_chunks, aux))
        new_nrs.push_back(nr_bdds() - 1);
    return {new_nrs, aux.back() + widths[(n_chunks - 1) * chunk]};
}

// The optional extra BDD of split_qbdd (bdd_collection.cpp:801-941): over the auxiliary variables only, the conjunction of
// one simplex per cut and, for every node of a cut and every other cut, the clause "this node is crossed => one of the
// nodes connected to it by a directed path is crossed" (skipped when all nodes of the other cut are connected).  The
// reference forms the conjunction with bdd_mgr (bdd_and, reorder, make_qbdd); here it is the product automaton of the
// constraints over the variables in ascending order, reduced bottom-up: the same canonical quasi-reduced BDD, nodes of a
// layer in make_qbdd's order (first discovery, hi child before lo child).  bdd_amd/bdd_collection.py: _implication_bdd.
bool bdd_store::append_implication_bdd(const std::vector<bddmma_instruction>& src, size_t off, size_t top_abs, size_t bot_abs,
                                       const std::vector<size_t>& widths, const std::vector<size_t>& loff, size_t chunk, size_t n_chunks,
                                       const std::vector<size_t>& aux)
{
    const size_t n_cuts = n_chunks - 1;
    auto cut_layer = [&](size_t c) { return c * chunk; };  // cuts are numbered 1 .. n_cuts
    auto aux_of = [&](size_t c, size_t i) { return aux[c - 1] + (widths[cut_layer(c)] - 1 - i); };
    using bits = std::vector<uint64_t>;
    auto test = [](const bits& m, size_t i) { return (m[i >> 6] >> (i & 63)) & 1u; };
    // sources[(c, c2)][j]: the nodes of cut c that reach node j of the later cut c2
    std::map<std::pair<size_t, size_t>, std::vector<bits>> sources;
    for (size_t c = 1; c <= n_cuts; ++c) {
        const size_t w = widths[cut_layer(c)], words = (w + 63) / 64;
        std::vector<bits> mask(w, bits(words, 0));
        for (size_t i = 0; i < w; ++i) mask[i][i >> 6] |= 1ull << (i & 63);
        for (size_t l = cut_layer(c); l < cut_layer(n_cuts); ++l) {
            std::vector<bits> nxt(widths[l + 1], bits(words, 0));
            for (size_t k = 0; k < widths[l]; ++k) {
                const bddmma_instruction& u = src[loff[l] + k];
                for (uint64_t ch : {u.lo, u.hi}) {
                    if (ch == top_abs || ch == bot_abs) continue;
                    bits& t = nxt[ch - off - loff[l + 1]];
                    for (size_t x = 0; x < words; ++x) t[x] |= mask[k][x];
                }
            }
            mask.swap(nxt);
            if ((l + 1) % chunk == 0) sources[{c, (l + 1) / chunk}] = mask;
        }
    }
    struct con_t { bool simplex; size_t neg; std::vector<size_t> vars; size_t last; };  // clause: neg = the negated variable, vars = the positive ones
    std::vector<con_t> cons;
    for (size_t c = 1; c <= n_cuts; ++c) {
        con_t k{true, 0, {}, 0};
        for (size_t i = 0; i < widths[cut_layer(c)]; ++i) k.vars.push_back(aux_of(c, i));
        cons.push_back(k);
    }
    size_t n_clauses = 0;
    auto add_clause = [&](size_t neg, std::vector<size_t> pos, size_t w2) {
        if (pos.size() == w2) return;
        cons.push_back(con_t{false, neg, std::move(pos), 0});
        ++n_clauses;
    };
    for (size_t c = 1; c < n_cuts; ++c)  // forward implications (:826-857)
        for (size_t c2 = c + 1; c2 <= n_cuts; ++c2) {
            const size_t w1 = widths[cut_layer(c)], w2 = widths[cut_layer(c2)];
            const auto& sm = sources[{c, c2}];
            for (size_t i1 = 0; i1 < w1; ++i1) {
                std::vector<size_t> pos;
                for (size_t i2 = 0; i2 < w2; ++i2)
                    if (test(sm[i2], i1)) pos.push_back(aux_of(c2, i2));
                add_clause(aux_of(c, i1), std::move(pos), w2);
            }
        }
    for (size_t c1 = 2; c1 <= n_cuts; ++c1)  // reverse implications (:861-893)
        for (size_t c2 = 1; c2 < c1; ++c2) {
            const size_t w1 = widths[cut_layer(c1)], w2 = widths[cut_layer(c2)];
            const auto& sm = sources[{c2, c1}];
            for (size_t i1 = 0; i1 < w1; ++i1) {
                std::vector<size_t> pos;
                for (size_t i2 = 0; i2 < w2; ++i2)
                    if (test(sm[i1], i2)) pos.push_back(aux_of(c2, i2));
                add_clause(aux_of(c1, i1), std::move(pos), w2);
            }
        }
    if (n_clauses == 0) return false;
    const size_t v_first = aux[0], v_last = aux.back() + widths[cut_layer(n_cuts)] - 1, n_lev = v_last - v_first + 1;
    std::vector<std::vector<size_t>> by_var(n_lev);
    for (size_t k = 0; k < cons.size(); ++k) {
        std::vector<size_t> vs = cons[k].vars;
        if (!cons[k].simplex) vs.push_back(cons[k].neg);
        cons[k].last = *std::max_element(vs.begin(), vs.end());
        for (size_t v : vs) by_var[v - v_first].push_back(k);
    }
    // product automaton: local states 0 / 1 (simplex: ones seen; clause: satisfied), 2 = finished and accepted
    using state = std::vector<uint8_t>;
    constexpr long T = TOP_LOCAL, B = BOT_LOCAL;
    std::vector<std::vector<std::pair<long, long>>> trans(n_lev);
    std::vector<state> cur{state(cons.size(), 0)};
    for (size_t l = 0; l < n_lev; ++l) {
        const size_t v = v_first + l;
        std::map<state, long> next_id;
        std::vector<state> nxt;
        trans[l].resize(cur.size());
        for (size_t sid = 0; sid < cur.size(); ++sid) {
            long out[2];
            for (int val = 0; val < 2; ++val) {
                state ns = cur[sid];
                bool dead = false;
                for (size_t k : by_var[l]) {
                    const con_t& c = cons[k];
                    uint8_t x;
                    if (c.simplex) {
                        x = (uint8_t)(ns[k] + val);
                        if (x > 1 || (v == c.last && x != 1)) { dead = true; break; }
                    } else {
                        x = (ns[k] == 1 || (v == c.neg && val == 0) || (v != c.neg && val == 1)) ? 1 : 0;
                        if (v == c.last && x != 1) { dead = true; break; }
                    }
                    ns[k] = v == c.last ? 2 : x;
                }
                if (dead) out[val] = B;
                else if (v == v_last) out[val] = T;
                else {
                    auto it = next_id.find(ns);
                    if (it == next_id.end()) {
                        it = next_id.emplace(ns, (long)nxt.size()).first;
                        nxt.push_back(ns);
                    }
                    out[val] = it->second;
                }
            }
            trans[l][sid] = {out[0], out[1]};
        }
        cur.swap(nxt);
    }
    // bottom-up reduction to the canonical quasi-reduced form
    std::vector<std::vector<long>> canon(n_lev);
    std::vector<std::vector<std::pair<long, long>>> tables(n_lev);
    for (size_t l = n_lev; l-- > 0;) {
        std::map<std::pair<long, long>, long> uniq;
        for (const auto& [lo0, hi0] : trans[l]) {
            const long lo = lo0 < 0 ? lo0 : canon[l + 1][lo0], hi = hi0 < 0 ? hi0 : canon[l + 1][hi0];
            if (lo == B && hi == B) { canon[l].push_back(B); continue; }
            auto it = uniq.find({lo, hi});
            if (it == uniq.end()) {
                it = uniq.emplace(std::make_pair(lo, hi), (long)tables[l].size()).first;
                tables[l].push_back({lo, hi});
            }
            canon[l].push_back(it->second);
        }
    }
    assert(canon[0][0] != B);
    std::vector<std::vector<long>> order(n_lev);
    order[0].push_back(canon[0][0]);
    for (size_t l = 0; l + 1 < n_lev; ++l) {
        std::set<long> seen;
        for (long k : order[l])
            for (long ch : {tables[l][k].second, tables[l][k].first})
                if (ch >= 0 && seen.insert(ch).second) order[l + 1].push_back(ch);
    }
    std::vector<std::map<long, size_t>> remap(n_lev);
    std::vector<size_t> offs(n_lev + 1, 0);
    for (size_t l = 0; l < n_lev; ++l) {
        for (size_t j = 0; j < order[l].size(); ++j) remap[l][order[l][j]] = j;
        offs[l + 1] = offs[l] + order[l].size();
    }
    std::vector<long> lo(offs[n_lev]), hi(offs[n_lev]);
    std::vector<size_t> var(offs[n_lev]);
    for (size_t l = 0; l < n_lev; ++l)
        for (const auto& [k, j] : remap[l]) {
            const auto [a, b2] = tables[l][k];
            lo[offs[l] + j] = a < 0 ? a : (long)(offs[l + 1] + remap[l + 1].at(a));
            hi[offs[l] + j] = b2 < 0 ? b2 : (long)(offs[l + 1] + remap[l + 1].at(b2));
            var[offs[l] + j] = v_first + l;
        }
    append_local(lo, hi, var, true);  // make_qbdd leaves the top sink first
    return true;
}

size_t bdd_store::compute_split_length(size_t parallelism) const
{
    std::vector<size_t> widths;
    for (size_t b = 0; b < nr_bdds(); ++b) {
        const auto w = layer_widths(b);
        if (w.size() > widths.size()) widths.resize(w.size(), 0);
        for (size_t i = 0; i < w.size(); ++i) widths[i] += w[i];
    }
    if (widths.empty()) return 0;
    for (size_t i = widths.size() - 1; i-- > 0;) widths[i] = std::max(widths[i], widths[i + 1]);
    auto occupancy = [&](const std::vector<size_t>& w) {
        double s = 0;
        for (size_t x : w) s += (double)std::min(x, parallelism) / (double)parallelism;
        return s / (double)w.size();
    };
    size_t length = widths.size();
    for (; length >= 200; --length) {
        std::vector<size_t> folded(length, 0);
        for (size_t i = 0; i < widths.size(); ++i) folded[i % length] += widths[i];
        if (occupancy(folded) >= 0.5) break;
    }
    return length;
}

std::pair<size_t, size_t> bdd_store::split_long_bdds(size_t nr_vars, size_t split_length, size_t parallelism, bool with_implication_bdd)
{
    if (split_length == 0) split_length = compute_split_length(parallelism);
    if (split_length == 0) return {0, nr_vars};
    size_t next = nr_vars;
    std::vector<size_t> removed;
    const size_t nb = nr_bdds();
    for (size_t b = 0; b < nb; ++b) {
        if (layer_widths(b).size() > split_length) {
            auto [nrs, na] = split_qbdd(b, split_length, next, with_implication_bdd);
            next = na;
            if (nrs.size() > 1) removed.push_back(b);
        }
    }
    const size_t n = removed.size();
    remove(std::move(removed));
    return {n, next};
}


// ---- text exports ---------------------------------------------------------------------------------------------------------------
// Both follow the reference's emitters statement by statement (the files are meant to be read by the same downstream tools): the same
// identifiers, the same order of rows, the same quirks (a row whose lo arc goes to the bot sink starts with " + ").
namespace {
inline bool top_sink(const bddmma_instruction& i) { return i.index == BDDMMA_TOPSINK; }
inline bool bot_sink(const bddmma_instruction& i) { return i.index == BDDMMA_BOTSINK; }
inline bool sink(const bddmma_instruction& i) { return top_sink(i) || bot_sink(i); }
}  // namespace

void bdd_store::write_bdd_lp(std::ostream& s, const std::vector<double>& costs) const
{
    auto arc = [&](size_t b, size_t idx, size_t value) {
        return std::string("arc_") + std::to_string(b) + "_" + std::to_string(idx - delimiters[b]) + "_" + std::to_string(value);
    };
    auto var = [](size_t v) { return std::string("x_") + std::to_string(v); };
    s << "Minimize\n";
    for (size_t i = 0; i < costs.size(); ++i) s << (costs[i] < 0 ? "-" : "+") << std::abs(costs[i]) << " " << var(i) << "\n";
    s << "Subject To\n";
    for (size_t b = 0; b < nr_bdds(); ++b) {  // flow through every BDD: one unit leaves the root, conservation at the inner nodes
        const size_t d0 = delimiters[b], d1 = delimiters[b + 1];
        const bddmma_instruction& root = instructions[d0];
        s << "R_" << b << ": ";
        if (!bot_sink(instructions[root.lo])) s << arc(b, d0, 0);
        if (!bot_sink(instructions[root.hi])) s << " + " << arc(b, d0, 1);
        s << " = 1\n";
        std::vector<std::vector<std::array<size_t, 2>>> incoming(d1 - d0);
        if (!sink(instructions[root.lo])) incoming[root.lo - d0].push_back({d0, 0});
        if (!sink(instructions[root.hi])) incoming[root.hi - d0].push_back({d0, 1});
        for (size_t i = d0 + 1; i + 2 < d1; ++i) {
            const bddmma_instruction& in = instructions[i];
            s << "FC_" << b << "_" << i - d0 << ": ";
            if (!bot_sink(instructions[in.lo])) s << arc(b, i, 0);
            if (!bot_sink(instructions[in.hi])) s << " + " << arc(b, i, 1);
            for (const auto& nv : incoming[i - d0]) s << " - " << arc(b, nv[0], nv[1]);
            s << " = 0\n";
            if (!sink(instructions[in.lo])) incoming[in.lo - d0].push_back({i, 0});
            if (!sink(instructions[in.hi])) incoming[in.hi - d0].push_back({i, 1});
        }
    }
    for (size_t b = 0; b < nr_bdds(); ++b) {  // x_v = flow over the hi arcs of the layer of v
        const size_t d0 = delimiters[b], d1 = delimiters[b + 1];
        size_t cur = instructions[d0].index;
        for (size_t i = d0; i + 2 < d1; ++i) {
            const bddmma_instruction& in = instructions[i];
            if (in.index != cur) {
                s << " - " << var(cur) << " = 0\n";
                cur = in.index;
            }
            if (!bot_sink(instructions[in.hi])) s << " + " << arc(b, i, 1);
        }
        s << " - " << var(cur) << " = 0\n";
    }
    s << "Bounds\n";
    s << "Binaries\n";
    for (size_t b = 0; b < nr_bdds(); ++b)
        for (size_t i = delimiters[b]; i + 2 < delimiters[b + 1]; ++i) {
            const bddmma_instruction& in = instructions[i];
            if (!bot_sink(instructions[in.lo])) s << arc(b, i, 0) << "\n";
            if (!bot_sink(instructions[in.hi])) s << arc(b, i, 1) << "\n";
        }
    s << "End\n";
}

void bdd_store::export_graphviz(size_t b, std::ostream& s) const
{
    const size_t d0 = delimiters[b], d1 = delimiters[b + 1];
    s << "digraph BDD\n";
    s << "{\n";
    // (unordered containers as in the reference: the order of the clusters in the file is theirs)
    std::unordered_map<size_t, std::string> clusters;
    std::unordered_map<size_t, size_t> cluster_nodes;
    for (size_t i = d0; i < d1; ++i) {
        const bddmma_instruction& in = instructions[i];
        if (sink(in)) {
            std::string& str = clusters[std::numeric_limits<size_t>::max()];
            str += std::to_string(i - d0) + (top_sink(in) ? " [label=\"top\"];\n" : " [label=\"bot\"];\n");
        } else {
            std::string& str = clusters[in.index];
            str += std::to_string(i - d0) + " [label=\"" + std::to_string(in.index) + "\"];\n";
            cluster_nodes[in.index] = i - d0;
        }
    }
    for (auto& [idx, str] : clusters) {
        s << "subgraph cluster_" << idx << " {\n";
        s << str;
        s << "color = blue\n";
        s << "}\n";
    }
    std::vector<std::array<size_t, 2>> order;   // invisible arrows keep the clusters in variable order
    for (auto [x, y] : cluster_nodes) order.push_back({x, y});
    std::sort(order.begin(), order.end(), [](const auto& a, const auto& c) { return a[0] < c[0]; });
    for (size_t c = 0; c + 1 < order.size(); ++c) s << order[c][1] << " -> " << order[c + 1][1] << " [style=invis];\n";
    for (size_t i = d0; i < d1; ++i) {
        const bddmma_instruction& in = instructions[i];
        if (sink(in)) continue;
        s << i - d0 << " -> " << in.hi - d0 << ";\n";
        s << i - d0 << " -> " << in.lo - d0 << "[style=\"dashed\"];\n";
    }
    s << "}\n";
}

}  // namespace bddmma_host
