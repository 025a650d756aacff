// layout.cpp — see layout.hpp.
#include "layout.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <cstring>
#include <limits>

namespace bddmma {

namespace {

// Host threads for the layout build (std::thread: no OpenMP runtime to link).  The reference builds its layout with ~10 device
// sorts (bdd_cuda_base.cu:146-391); here it is host work, spread over the cores: f(begin, end, thread) over contiguous chunks.
std::atomic<unsigned> g_layout_threads{0};  // bddmma_set_layout_threads: 0 = BDDMMA_THREADS, else min(cores, 32)
thread_local unsigned t_layout_threads = 0; // bddmma_set_thread_layout_threads: the calling thread's builds only; wins over the process-wide value
struct Par {
    unsigned nt = 1;
    Par()
    {
        const char* e = std::getenv("BDDMMA_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned set = t_layout_threads ? t_layout_threads : g_layout_threads.load(std::memory_order_relaxed);
        nt = set ? set : (e ? (unsigned)std::atoi(e) : std::min(hw ? hw : 1u, 32u));
        if (nt < 1) nt = 1;
    }
    template <typename F>
    void run(uint64_t n, F&& f, uint64_t min_parallel = 2048) const
    {
        if (nt == 1 || n < min_parallel) {
            if (n) f((uint64_t)0, n, 0u);
            return;
        }
        const uint64_t chunk = (n + nt - 1) / nt;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) {
            const uint64_t b = t * chunk, e = std::min(n, b + chunk);
            if (b >= e) break;
            th.emplace_back([&f, b, e, t] { f(b, e, t); });
        }
        for (auto& x : th) x.join();
    }
};
// first error (lowest item index) found by any thread
struct FirstError {
    std::mutex m;
    uint64_t at = std::numeric_limits<uint64_t>::max();
    int code = 0;
    std::string msg;
    void set(uint64_t item, int c, std::string text)
    {
        std::lock_guard<std::mutex> g(m);
        if (item < at) { at = item; code = c; msg = std::move(text); }
    }
};

inline bool is_top(const bddmma_instruction& i) { return i.index == BDDMMA_TOPSINK; }
inline bool is_bot(const bddmma_instruction& i) { return i.index == BDDMMA_BOTSINK; }
inline bool is_term(const bddmma_instruction& i) { return is_top(i) || is_bot(i); }

struct PackBuilder {
    // Greedy first-fit in input order: a BDD joins the current pack if, at every hop, its layer
    // fits below `width` slots without straddling a `group`-slot boundary (group = 64 for narrow
    // packs so that a layer lives inside one wavefront group; 0 = no such rule).
    uint32_t width, group;
    uint32_t fill = 0;  // a BDD joins an OPEN pack only below `fill` slots (0: = width); an empty pack takes anything up to `width`
    std::vector<uint32_t> used;     // slots used per hop in the open pack
    std::vector<uint32_t> nlayers;  // layers per hop in the open pack
    uint32_t maxw = 0;
    bool open = false;
    // Staggered packing (bddmma_options.pack_stagger): a BDD that does not fit the open pack from hop 0 may start `d` hops further down,
    // where the BDDs already placed have become narrow again, as long as the pack stays within `max_hops` hops and no other BDD starts
    // at that hop (the kernels know one root slot per hop below the first).  0: every BDD starts at hop 0.
    uint32_t max_hops = 0;
    bool chain_any = false;         // explicit pack_stagger: chain whatever fits (automatic mode: only BDDs that narrow again, see add)
    std::vector<uint16_t> root;     // per hop of the open pack: local slot of the BDD that starts there (hops > 0), or NO_ROOT

    // closed packs
    std::vector<uint32_t> pack_first_bdd;      // index into `order`
    std::vector<uint32_t> pack_hop_ptr;        // into flat_used / flat_nlayers
    std::vector<uint32_t> flat_used, flat_nlayers;
    std::vector<uint16_t> flat_root;
    std::vector<uint8_t> pack_steps;

    static uint8_t steps_for(uint32_t w)
    {
        uint8_t s = 0;
        while ((1u << s) < w) ++s;
        return s;
    }
    void close()
    {
        if (!open) return;
        pack_hop_ptr.push_back((uint32_t)flat_used.size());
        flat_used.insert(flat_used.end(), used.begin(), used.end());
        flat_nlayers.insert(flat_nlayers.end(), nlayers.begin(), nlayers.end());
        root.resize(used.size(), NO_ROOT);
        flat_root.insert(flat_root.end(), root.begin(), root.end());
        pack_steps.push_back(steps_for(maxw));
        used.clear();
        nlayers.clear();
        root.clear();
        maxw = 0;
        open = false;
    }
    uint32_t place(uint32_t u, uint32_t w) const
    {
        // narrow packs: a two-node layer starts at an even lane, so that its minimum is one DPP swap of neighbouring lanes
        // (quad_perm [1, 0, 3, 2]; kernels.hpp: k_fwd_res2 / k_bwd_res2)
        if (group && w == 2 && (u & 1u)) ++u;
        if (group && (u % group) + w > group) u = (u + group - 1) / group * group;
        return u;
    }
    bool fits_at(uint32_t d, const uint32_t* widths, uint32_t n) const
    {
        for (uint32_t h = 0; h < n; ++h) {
            const uint32_t u = d + h < used.size() ? used[d + h] : 0;
            if (place(u, widths[h]) + widths[h] > (fill ? fill : width)) return false;
        }
        return true;
    }
    static bool chainable(const uint32_t* widths, uint32_t n)
    {
        uint64_t sum = 0;
        uint32_t mx = 0;
        for (uint32_t h = 0; h < n; ++h) { sum += widths[h]; mx = std::max(mx, widths[h]); }
        return mx >= 4 && sum * 10 <= (uint64_t)mx * n * 6;  // mean layer width <= 0.6 of the widest layer
    }
    // widths[0..n) = layer widths of the BDD.  Writes the slot position of every layer to pos[]; returns the hop of the pack at which
    // the BDD starts.
    uint32_t add(uint32_t order_idx, const uint32_t* widths, uint32_t n, uint32_t* pos)
    {
        uint32_t d = 0;
        if (open) {
            bool fits = fits_at(0, widths, n);
            // Only BDDs that are narrow at their ends and wide in the middle are chained (general linear rows: mean layer width about a
            // third of the widest).  Flat BDDs — covering, simplex, cardinality rows — never become narrow again: chained, each of them
            // lengthens the pack by a hop for two lanes' worth of nodes (a staircase), and a staggered pack also costs the launch its four
            // packs per workgroup and the resident sweeps.  Seen with keep_bdd_order or shape classes of < 256 members, which bypass the
            // closed-form packing of form_narrow (300 k cover rows, 46 880 -> 112 492 wave-hops in automatic mode).
            if (!fits && max_hops > n && (chain_any || chainable(widths, n))) {
                for (uint32_t dd = 1; dd + n <= max_hops && dd <= used.size() && !fits; ++dd) {
                    if (dd < root.size() && root[dd] != NO_ROOT) continue;  // one BDD may start per hop
                    if (fits_at(dd, widths, n)) { fits = true; d = dd; }
                }
            }
            if (!fits) close();
        }
        if (!open) {
            open = true;
            d = 0;
            pack_first_bdd.push_back(order_idx);
        }
        if (used.size() < d + n) {
            used.resize(d + n, 0);
            nlayers.resize(d + n, 0);
        }
        for (uint32_t h = 0; h < n; ++h) {
            const uint32_t p = place(used[d + h], widths[h]);
            pos[h] = p;
            used[d + h] = p + widths[h];
            nlayers[d + h]++;
            maxw = std::max(maxw, widths[h]);
        }
        if (d > 0) {
            root.resize(std::max<size_t>(root.size(), d + 1), NO_ROOT);
            root[d] = (uint16_t)pos[0];
        }
        return d;
    }
    void finish(uint32_t n_order)
    {
        close();
        pack_first_bdd.push_back(n_order);
        pack_hop_ptr.push_back((uint32_t)flat_used.size());
    }
    uint32_t n_packs() const { return (uint32_t)pack_first_bdd.size() - 1; }
};

}  // namespace

static int build_layout_w(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                          const bddmma_options* opts, HostLayout& L, std::string& err, bool keep_debug_maps, uint32_t real_size, ChipInfo chip);

int build_layout(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                 const bddmma_options* opts, HostLayout& L, std::string& err, bool keep_debug_maps, uint32_t real_size, ChipInfo chip)
{
    int rc = build_layout_w(instr, delims, n_bdds, opts, L, err, keep_debug_maps, real_size, chip);
    // Few packs (small instance, or few but long BDDs): a sweep is then bound by the latency of one pack's
    // hop chain, so prefer more, narrower packs.  Only when the caller left pack_width open.
    // (threshold measured on random set cover: 2.1 M nodes = 1 563 packs of 128: 64-wide 18.5 / 19.1 us per sweep vs 19.7 / 18.3; 4.2 M nodes =
    // 3 125 packs: 29.4 / 29.6 vs 27.3 / 25.5 — so 128 stays from ~2 000 packs on)
    // Longer packs need more of them: 161 538 rows of 32 variables (10.5 M nodes) are 2 524 packs of 128 with 32 hops and run at 6 455 it/s, as
    // 64-wide packs at 7 319 (float); 40 000 knapsack rows in staggered packs of 42 hops: 3 767 packs of 128 4 901 / 3 790 it/s (float / double),
    // 6 048 of 64 5 333 / 4 015.  So the pack count below which 64-wide packs are tried grows with the hops of the longest pack (x hops / 16,
    // between 1 and 3).
    uint32_t longest_pack = 0;
    if (rc == BDDMMA_OK)
        for (uint32_t p = 0; p < L.narrow.n_packs(); ++p) longest_pack = std::max(longest_pack, L.narrow.pack_hop_ptr[p + 1] - L.narrow.pack_hop_ptr[p]);
    const uint32_t few_packs = 2048u * std::min(48u, std::max(16u, longest_pack)) / 16u;
    // General linear rows (round 5): where the BDDs chained into staggered packs hold most of the narrow nodes, 64-slot packs at any pack
    // count — a staggered pack is swept by a workgroup of one wave either way, a 128-slot pack is two 64-lane groups filled independently
    // (lane utilisation 0.45-0.58 against 0.63) and twice the work per wave.  it/s float, 128 / 64 slots (tools/exp_r05_u.sh, _v.sh):
    // 100 000 rows of 11 variables 6 080 / 7 550, 150 000 rows of 10 5 530 / 6 990; with covering rows in the majority 128 stays ahead
    // (10 000 knapsack + 400 000 covering rows 6 920 / 6 520; 20 000 + 250 000, 47 % of the nodes: 6 570 / 6 660, double 3 470 / 3 350).
    bool staggered = false;
    if (rc == BDDMMA_OK)
        for (uint16_t r : L.narrow.hop_root)
            if (r != NO_ROOT) { staggered = true; break; }
    const bool mostly_chained = staggered && L.diamond_nodes * 2 > L.narrow_nodes;
    if (rc == BDDMMA_OK && !(opts && opts->pack_width) && (L.narrow.n_packs() < few_packs || mostly_chained) && L.narrow.n_packs() > 0) {
        bddmma_options o = opts ? *opts : bddmma_options{};
        o.pack_width = 64;
        HostLayout L2;
        std::string err2;
        if (build_layout_w(instr, delims, n_bdds, &o, L2, err2, keep_debug_maps, real_size, chip) == BDDMMA_OK) L = std::move(L2);
    }
    return rc;
}

static int build_layout_w(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                          const bddmma_options* opts, HostLayout& L, std::string& err, bool keep_debug_maps, uint32_t real_size, ChipInfo chip)
{
    L = HostLayout();
    // BDDMMA_LAYOUT_TIMING=1: phase times on stderr
#ifdef BDDMMA_EXPERIMENTAL  // make EXPERIMENTAL=1
    static const bool timing = std::getenv("BDDMMA_LAYOUT_TIMING") != nullptr;
#else
    constexpr bool timing = false;
#endif
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[layout] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (!instr || !delims || n_bdds == 0) {
        err = "empty BDD collection";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    uint32_t W = opts && opts->pack_width ? opts->pack_width : 128;
    uint32_t WW = opts && opts->wide_pack_width ? opts->wide_pack_width : 2048;
    if (W != 64 && W != 128 && W != 256) {
        err = "pack_width must be 64, 128 or 256";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    if (WW < 64 || WW > 4096) {
        err = "wide_pack_width must be in [64, 4096]";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    if (opts && opts->pack_stagger > 0xFFFFu) {  // hop counts of a pack are 16-bit in the resident pack headers
        err = "pack_stagger must be <= 65535";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    if (opts && opts->pack_fill && (opts->pack_fill > W || opts->pack_fill < 2)) {
        err = "pack_fill must be in [2, pack_width]";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    L.pack_width = W;
    L.wide_pack_width = WW;
    L.n_bdds = n_bdds;
    L.n_input_nodes = delims[n_bdds] - delims[0];

    // ---- pass 1: parse + validate every BDD (bdd_cuda_base.cu:86-144) ------------------------
    const Par par;
    std::vector<uint64_t> lay_first;   // first instruction of every input layer (BDD-major)
    std::vector<uint32_t> bdd_lay_ptr(n_bdds + 1, 0);
    std::vector<uint32_t> bdd_maxw(n_bdds, 0);
    std::vector<uint64_t> bdd_shape(n_bdds, 0);  // hash of the BDD's structure (arcs and layer boundaries relative to its first entry)
    uint64_t max_var = 0;
    {
        // 1a: terminals where they belong, layers per BDD, largest variable
        FirstError fe;
        std::vector<uint64_t> tmax(par.nt, 0);
        par.run(n_bdds, [&](uint64_t b0, uint64_t b1, unsigned t) {
            uint64_t mv = 0;
            for (uint64_t b = b0; b < b1; ++b) {
                const uint64_t d0 = delims[b], d1 = delims[b + 1];
                if (d1 < d0 + 3) { fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + " has fewer than 3 entries"); return; }
                const bddmma_instruction& t0 = instr[d1 - 2];
                const bddmma_instruction& t1 = instr[d1 - 1];
                if (!((is_top(t0) && is_bot(t1)) || (is_bot(t0) && is_top(t1)))) {
                    fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + ": the last two entries must be the top and bot sinks");
                    return;
                }
                uint64_t prev = BDDMMA_BOTSINK - 7;
                uint32_t nl = 0;
                for (uint64_t i = d0; i < d1 - 2; ++i) {
                    if (is_term(instr[i])) { fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + ": terminal entry before the end"); return; }
                    if (instr[i].index != prev) {
                        ++nl;
                        prev = instr[i].index;
                        mv = std::max(mv, prev);
                    }
                }
                bdd_lay_ptr[b + 1] = nl;
            }
            tmax[t] = mv;
        });
        if (fe.code) { err = fe.msg; return fe.code; }
        for (uint64_t v : tmax) max_var = std::max(max_var, v);
        uint64_t total = 0;
        for (uint64_t b = 0; b < n_bdds; ++b) {
            total += bdd_lay_ptr[b + 1];
            if (total > std::numeric_limits<uint32_t>::max() - 2) { err = "too many layers"; return BDDMMA_ERR_UNSUPPORTED; }
            bdd_lay_ptr[b + 1] = (uint32_t)total;
        }
    }
    const uint32_t Lin = bdd_lay_ptr[n_bdds];
    lay_first.assign((size_t)Lin + 1, 0);  // + sentinel slot
    par.run(n_bdds, [&](uint64_t b0, uint64_t b1, unsigned) {
        for (uint64_t b = b0; b < b1; ++b) {
            uint64_t prev = BDDMMA_BOTSINK - 7;
            uint32_t l = bdd_lay_ptr[b];
            for (uint64_t i = delims[b]; i < delims[b + 1] - 2; ++i)
                if (instr[i].index != prev) {
                    lay_first[l++] = i;
                    prev = instr[i].index;
                }
        }
    });
    if (max_var >= (uint64_t)std::numeric_limits<int32_t>::max()) {
        err = "variable index too large";
        return BDDMMA_ERR_UNSUPPORTED;
    }
    L.n_vars = max_var + 1;
    L.n_layers = Lin;
    L.num_bdds_per_var.assign(L.n_vars, 0);

    auto layer_end = [&](uint64_t b, uint32_t l) -> uint64_t {
        return (l + 1 < bdd_lay_ptr[b + 1]) ? lay_first[l + 1] : delims[b + 1] - 2;
    };
    {
        // 1b: one root, every variable in one layer, arcs go to the next layer (QBDD), widths, shape hash, BDDs per variable
        FirstError fe;
        std::vector<uint64_t> thops(par.nt, 0);
        par.run(n_bdds, [&](uint64_t b0, uint64_t b1, unsigned t) {
            std::vector<uint64_t> vars;
            uint64_t hops = 0;
            for (uint64_t b = b0; b < b1; ++b) {
                const uint64_t d1 = delims[b + 1];
                const uint32_t l0 = bdd_lay_ptr[b], l1 = bdd_lay_ptr[b + 1];
                if (layer_end(b, l0) - lay_first[l0] != 1) {
                    fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + " does not have exactly one root node");
                    return;
                }
                vars.clear();
                uint32_t maxw = 0;
                uint64_t shape = 0;
                for (uint32_t l = l0; l < l1; ++l) {
                    const uint64_t f = lay_first[l], e = layer_end(b, l);
                    const uint64_t v = instr[f].index;
                    vars.push_back(v);
                    __atomic_fetch_add(&L.num_bdds_per_var[v], 1, __ATOMIC_RELAXED);
                    maxw = std::max<uint32_t>(maxw, (uint32_t)(e - f));
                    shape = (shape ^ (e - f)) * 1099511628211ull + 0x9e3779b97f4a7c15ull;
                    const bool last = (l + 1 == l1);
                    const uint64_t nf = last ? 0 : lay_first[l + 1], ne = last ? 0 : layer_end(b, l + 1);
                    for (uint64_t i = f; i < e; ++i) {
                        for (int side = 0; side < 2; ++side) {
                            const uint64_t c = side ? instr[i].hi : instr[i].lo;
                            if (c >= d1 || c < delims[b]) {
                                fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + ": child index out of range");
                                return;
                            }
                            shape = (shape ^ (is_term(instr[c]) ? (is_top(instr[c]) ? ~0ull : ~1ull) : c - delims[b])) * 1099511628211ull;
                            if (is_bot(instr[c])) continue;
                            if (is_top(instr[c])) {
                                if (!last) {
                                    fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + " is not a QBDD: arc to the top sink skips variables");
                                    return;
                                }
                                continue;
                            }
                            if (last || c < nf || c >= ne) {
                                fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + " is not a QBDD: arc does not go to the next variable's layer");
                                return;
                            }
                        }
                    }
                }
                // is_reordered (bdd_cuda_base.cu:98-99): no variable in two layers of one BDD
                std::sort(vars.begin(), vars.end());
                const auto dup = std::adjacent_find(vars.begin(), vars.end());
                if (dup != vars.end()) {
                    fe.set(b, BDDMMA_ERR_INVALID_BDD, "BDD " + std::to_string(b) + " is not reordered: variable " + std::to_string(*dup) + " appears in two layers");
                    return;
                }
                bdd_maxw[b] = maxw;
                bdd_shape[b] = shape;
                hops = std::max<uint64_t>(hops, l1 - l0);
            }
            thops[t] = std::max(thops[t], hops);
        });
        if (fe.code) { err = fe.msg; return fe.code; }
        for (uint64_t h : thops) L.n_hops = std::max(L.n_hops, h);
    }

    lap("parse + validate");
    // ---- pass 2: narrow / wide classification and greedy pack formation ----------------------
    const uint32_t narrow_limit = std::min<uint32_t>(NARROW_MAX_LAYER_WIDTH, W);
    std::vector<uint32_t> order_n, order_w, order_h;
    uint32_t HW = 0;  // huge packs: one workgroup per pack too, but the frontier lives in global memory
    for (uint64_t b = 0; b < n_bdds; ++b) {
        if (bdd_maxw[b] <= narrow_limit) order_n.push_back((uint32_t)b);
        else if (bdd_maxw[b] <= WW) order_w.push_back((uint32_t)b);
        else if (bdd_maxw[b] < WW_TOP) {
            order_h.push_back((uint32_t)b);
            HW = std::max<uint32_t>(HW, (uint32_t)bdd_maxw[b]);
        } else {
            err = "BDD " + std::to_string(b) + " has a layer of " + std::to_string(bdd_maxw[b]) +
                  " nodes; the node word addresses " + std::to_string(WW_TOP - 1) + " nodes per hop";
            return BDDMMA_ERR_UNSUPPORTED;
        }
    }
    HW = (HW + 7u) / 8u * 8u;  // per-pack scratch regions stay 8-byte aligned
    L.huge_pack_width = HW;
    // Narrow BDDs of the same shape are packed together (stable: input order inside a shape class), so that packs
    // are structurally identical and share one stored word sequence (layout.hpp: narrow_words_unique).  Classes are
    // ordered by first appearance.  The result does not depend on the order of the BDDs.
    std::vector<uint32_t> order_cls;  // shape class of order_n[k] (empty when the input order is kept)
    if (!(opts && opts->keep_bdd_order == 1)) {
        std::unordered_map<uint64_t, uint32_t> cls;
        std::vector<uint32_t> cls_of(order_n.size());
        for (size_t k = 0; k < order_n.size(); ++k) cls_of[k] = cls.emplace(bdd_shape[order_n[k]], (uint32_t)cls.size()).first->second;
        std::vector<uint32_t> idx(order_n.size());
        for (size_t k = 0; k < idx.size(); ++k) idx[k] = (uint32_t)k;
        // Diamond-shaped BDDs of small classes (general linear rows: every shape its own class) are chained into staggered packs below;
        // neighbours of similar peak width chain at shorter offsets than neighbours in input order — 4 000 / 40 000 knapsack rows of 14
        // variables: lane utilisation 0.57 -> 0.60 with at most 42 hops per pack (a window of 16-64 candidates per placement or several
        // open packs give no more: 0.60-0.61; the chain itself ends at ~0.68).  Widest first, and all of them before the other classes:
        // their chained packs are the longest of the launch and must not start last (behind 250 000 covering rows: 5 640 -> 5 060 it/s);
        // classes stay together (same shape = same peak).
        std::vector<uint32_t> cls_size(cls.size(), 0);
        for (uint32_t c : cls_of) ++cls_size[c];
        std::vector<uint32_t> peak_key(order_n.size(), 0);  // 0: keep the class order
        if (!(opts && opts->keep_bdd_order == 2))
            for (size_t k = 0; k < order_n.size(); ++k) {
                const uint32_t b = order_n[k];
                const uint64_t nodes = (delims[b + 1] - delims[b]) - 2, nl = bdd_lay_ptr[b + 1] - bdd_lay_ptr[b];
                const bool diamond = bdd_maxw[b] >= 4 && nodes * 10 <= (uint64_t)bdd_maxw[b] * nl * 6;  // = PackBuilder::chainable
                if (diamond && cls_size[cls_of[k]] < 256) peak_key[k] = bdd_maxw[b];
            }
        for (size_t k = 0; k < order_n.size(); ++k) {
            const uint64_t nodes = (delims[order_n[k] + 1] - delims[order_n[k]]) - 2;
            L.narrow_nodes += nodes;
            if (peak_key[k]) L.diamond_nodes += nodes;
        }
        // ... and the other classes longest BDDs first (the blocks of a launch start in pack order: long packs must not start last)
        std::vector<uint32_t> len_key(order_n.size(), 0);
        if (!(opts && opts->keep_bdd_order == 2))
            for (size_t k = 0; k < order_n.size(); ++k) len_key[k] = bdd_lay_ptr[order_n[k] + 1] - bdd_lay_ptr[order_n[k]];
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            if (peak_key[a] != peak_key[b]) return peak_key[a] > peak_key[b];
            if (len_key[a] != len_key[b]) return len_key[a] > len_key[b];
            return cls_of[a] < cls_of[b];
        });
        std::vector<uint32_t> grouped(order_n.size());
        order_cls.resize(order_n.size());
        for (size_t k = 0; k < idx.size(); ++k) {
            grouped[k] = order_n[idx[k]];
            order_cls[k] = cls_of[idx[k]];
        }
        order_n.swap(grouped);
    }
    std::vector<uint32_t> lay_pos(Lin);  // slot position of every input layer inside its (pack,hop)
    std::vector<uint32_t> bdd_hop0(n_bdds, 0);  // hop of its pack at which a BDD starts (> 0: staggered narrow packs)
    // Wide packs are swept by one workgroup each, two barriers per hop whatever the width, so for a given amount of wide BDDs
    // narrower packs mean more workgroups in flight.  The pack width adapts: aim at >= 1024 packs (4 workgroups per CU), but a
    // pack must hold the widest of these BDDs; wide_pack_width (the option) stays the upper limit.
    uint32_t WWe = WW;
    if (!order_w.empty()) {
        uint64_t sum_w = 0;
        uint32_t max_w = 0;
        for (uint32_t b : order_w) {
            sum_w += bdd_maxw[b];
            max_w = std::max(max_w, bdd_maxw[b]);
        }
        // ... and not beyond 1024 slots unless a BDD needs it (or the option asks): 25 000 rows of 18 variables (15 M nodes, layers up to ~500
        // nodes) as 2 554 packs of 1024 slots run at 2 205 it/s, as 1 245 packs of 2048 at 1 867 (512 threads x 2 nodes either way)
        const uint64_t cap = opts && opts->wide_pack_width ? WW : 1024;
        const uint64_t want = std::max<uint64_t>(max_w, std::min<uint64_t>(cap, (sum_w + 1023) / 1024));
        WWe = (uint32_t)std::min<uint64_t>(WW, (want + 63) / 64 * 64);
        WWe = std::max(WWe, (max_w + 0u));  // WW itself need not be a multiple of 64
    }
    L.wide_pack_width = WWe;
    PackBuilder pn{W, 64}, pw{WWe, 0}, ph{std::max(HW, 1u), 0};
    pn.fill = opts && opts->pack_fill ? opts->pack_fill : 0;
    std::vector<uint32_t> widths;
    auto form = [&](PackBuilder& pb, const std::vector<uint32_t>& order) {
        for (uint32_t k = 0; k < order.size(); ++k) {
            const uint64_t b = order[k];
            const uint32_t l0 = bdd_lay_ptr[b], n = bdd_lay_ptr[b + 1] - l0;
            widths.resize(n);
            for (uint32_t h = 0; h < n; ++h) widths[h] = (uint32_t)(layer_end(b, l0 + h) - lay_first[l0 + h]);
            bdd_hop0[b] = pb.add(k, widths.data(), n, &lay_pos[l0]);
        }
        pb.finish((uint32_t)order.size());
    };
    // Narrow BDDs arrive grouped by shape.  A long run of one shape (every row of a constraint family) is packed in closed form:
    // first-fit of identical BDDs repeats the same pack, so one pack is simulated and stamped out; the per-BDD work (checking
    // that the widths really are the representative's, writing the slot positions) is spread over the host threads.
    auto form_narrow = [&]() {
        if (order_cls.empty()) { form(pn, order_n); return; }
        PackBuilder& pb = pn;
        const uint32_t n_order = (uint32_t)order_n.size();
        std::vector<uint32_t> rep_w;
        std::vector<std::vector<uint32_t>> pos;  // pos[j][h]: slot position of layer h of the j-th BDD of a pack
        uint32_t k = 0;
        while (k < n_order) {
            uint32_t k1 = k;
            while (k1 < n_order && order_cls[k1] == order_cls[k]) ++k1;
            const uint32_t m = k1 - k;
            const uint64_t b_rep = order_n[k];
            const uint32_t l_rep = bdd_lay_ptr[b_rep], n = bdd_lay_ptr[b_rep + 1] - l_rep;
            bool uniform = m >= 256;
            if (uniform) {
                rep_w.resize(n);
                for (uint32_t h = 0; h < n; ++h) rep_w[h] = (uint32_t)(layer_end(b_rep, l_rep + h) - lay_first[l_rep + h]);
                // the class id comes from a hash: make sure the widths really agree
                std::vector<char> bad(par.nt, 0);
                par.run(m, [&](uint64_t i0, uint64_t i1, unsigned t) {
                    for (uint64_t i = i0; i < i1 && !bad[t]; ++i) {
                        const uint64_t b = order_n[k + i];
                        const uint32_t l0 = bdd_lay_ptr[b];
                        if (bdd_lay_ptr[b + 1] - l0 != n) { bad[t] = 1; break; }
                        for (uint32_t h = 0; h < n; ++h)
                            if ((uint32_t)(layer_end(b, l0 + h) - lay_first[l0 + h]) != rep_w[h]) { bad[t] = 1; break; }
                    }
                });
                for (char c : bad) uniform = uniform && !c;
            }
            if (!uniform) {
                for (uint32_t kk = k; kk < k1; ++kk) {
                    const uint64_t b = order_n[kk];
                    const uint32_t l0 = bdd_lay_ptr[b], nn = bdd_lay_ptr[b + 1] - l0;
                    widths.resize(nn);
                    for (uint32_t h = 0; h < nn; ++h) widths[h] = (uint32_t)(layer_end(b, l0 + h) - lay_first[l0 + h]);
                    bdd_hop0[b] = pb.add(kk, widths.data(), nn, &lay_pos[l0]);
                }
                k = k1;
                continue;
            }
            pb.close();  // a uniform run starts with an empty pack
            // simulate one pack
            pos.clear();
            std::vector<uint32_t> used(n, 0);
            uint32_t maxw = 0;
            for (uint32_t h = 0; h < n; ++h) maxw = std::max(maxw, rep_w[h]);
            for (;;) {
                bool fits = true;
                const uint32_t limit = pos.empty() || !pb.fill ? pb.width : pb.fill;  // the first BDD of a pack may use the whole width
                for (uint32_t h = 0; h < n && fits; ++h)
                    if (pb.place(used[h], rep_w[h]) + rep_w[h] > limit) fits = false;
                if (!fits) break;
                pos.emplace_back(n);
                for (uint32_t h = 0; h < n; ++h) {
                    pos.back()[h] = pb.place(used[h], rep_w[h]);
                    used[h] = pos.back()[h] + rep_w[h];
                }
            }
            const uint32_t c = (uint32_t)pos.size();  // >= 1: a narrow BDD always fits an empty pack
            for (uint32_t first = 0; first < m; first += c) {
                const uint32_t cnt = std::min(c, m - first);
                pb.pack_first_bdd.push_back(k + first);
                pb.pack_hop_ptr.push_back((uint32_t)pb.flat_used.size());
                for (uint32_t h = 0; h < n; ++h) {
                    pb.flat_used.push_back(pos[cnt - 1][h] + rep_w[h]);
                    pb.flat_nlayers.push_back(cnt);
                    pb.flat_root.push_back(NO_ROOT);
                }
                pb.pack_steps.push_back(PackBuilder::steps_for(maxw));
            }
            par.run(m, [&](uint64_t i0, uint64_t i1, unsigned) {
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint32_t l0 = bdd_lay_ptr[order_n[k + i]];
                    const std::vector<uint32_t>& pj = pos[i % c];
                    for (uint32_t h = 0; h < n; ++h) lay_pos[l0 + h] = pj[h];
                }
            });
            k = k1;
        }
        pb.finish(n_order);
    };
    {   // staggered narrow packs: explicit limit, or automatic when the instance is large enough that the longer (hence fewer) packs
        // still fill the GPU — below ~4096 packs a sweep is bound by the length of one pack's hop chain, not by lane utilisation
        uint64_t narrow_nodes = 0;
        uint32_t longest = 0;
        for (uint32_t b : order_n) {
            narrow_nodes += (delims[b + 1] - delims[b]) - 2;
            longest = std::max(longest, bdd_lay_ptr[b + 1] - bdd_lay_ptr[b]);
        }
        const uint32_t opt = opts ? opts->pack_stagger : 0;
        if (opt >= 2) { pn.max_hops = opt; pn.chain_any = true; }
        else if (opt == 0) {
            // (the pack count to keep was 4096 until the segmented minimum got its DPP folds; on those kernels 10 M knapsack nodes run at
            // 4 320 / 4 223 / 4 674 / 4 377 / 4 446 it/s with at most 28 / 36 / 42 / 48 / 56 hops per pack (6 724 ... 2 552 packs), the mixed
            // instance at 5 255 / 5 394 / 5 451 / 5 101 with 28 / 36 / 42 / 48: three BDD lengths, ~3 600 packs = 3.5 waves per SIMD)
            const uint64_t hops_for_packs = narrow_nodes / ((uint64_t)W * 2730 * 6 / 10 + 1);
            pn.max_hops = hops_for_packs >= longest + 2 ? (uint32_t)std::min<uint64_t>(hops_for_packs, 3ull * longest) : 0;
        }
    }
    form_narrow();
    // Staggered narrow packs get one pack per workgroup (see waves_per_block below), and the wide packs of the same launch then have one
    // wavefront each: narrower wide packs (two hop slots per lane, ~one BDD per pack) keep more of them in flight.  With 128 instead of the
    // 256-512 the rule above picks (10 M nodes, it/s float / double): 30 k knapsack + 100 k covering rows 5 892 -> 6 012 / 3 530 -> 3 911, 20 k +
    // 250 k 6 084 -> 6 209 / 3 256 -> 3 438, 40 k knapsack rows 5 344 -> 5 295 / 3 881 -> 3 835.
    if (!order_w.empty() && !(opts && (opts->wide_pack_width || opts->waves_per_block))) {
        bool staggered = false;
        for (uint16_t r : pn.flat_root)
            if (r != NO_ROOT) { staggered = true; break; }
        if (staggered) {
            uint32_t max_w = 0;
            for (uint32_t b : order_w) max_w = std::max(max_w, bdd_maxw[b]);
            const uint32_t narrow_ww = std::max(128u, (max_w + 63u) / 64u * 64u);
            if (narrow_ww < WWe) {
                WWe = narrow_ww;
                L.wide_pack_width = WWe;
                pw.width = WWe;
            }
        }
    }
    {   // staggered wide packs: BDDs of general linear rows are diamond-shaped (mean layer width ~ a third of the widest), so side by side from
        // hop 0 a wide pack fills a third of its slots.  Staggered, a pack is ONE BDD wide and chains them — the next one starts where the
        // previous one narrows — over up to three BDD lengths, while >= ~500 packs remain.  25 000 rows of 18 variables (15 M nodes, layers up
        // to ~510 nodes), it/s float: packs of 512 / 640 / 768 / 1024 slots side by side 2 136 / 1 898 / 1 851 / 2 194; staggered over 36 hops
        // 2 628 / 2 264 / 2 149 / 2 089, over 54 (three lengths) 2 844 / 2 865 / 2 603 / 2 205, over 90 2 371 / 2 161 / 1 995 / 1 787.
        uint64_t wide_nodes = 0;
        uint32_t longest = 0, max_w = 0;
        for (uint32_t b : order_w) {
            wide_nodes += (delims[b + 1] - delims[b]) - 2;
            longest = std::max(longest, bdd_lay_ptr[b + 1] - bdd_lay_ptr[b]);
            max_w = std::max(max_w, bdd_maxw[b]);
        }
        const uint32_t opt = opts ? opts->pack_stagger : 0;
        if (opt >= 2) { pw.max_hops = opt; pw.chain_any = true; }
        else if (opt == 0 && !order_w.empty()) {
            const uint32_t one_wide = std::min(WW, std::max(128u, (max_w + 63u) / 64u * 64u));
            const uint32_t w_st = opts && opts->wide_pack_width ? pw.width : std::max(one_wide, max_w);
            // packs to keep: ~500, and enough wavefronts for the chip when the narrow packs of the launch do not provide them (a wide pack has
            // w_st / 128 wavefronts: 4 000 such rows = 2.5 M nodes are 994 packs side by side, 7 747 it/s, and 582 chained ones, 6 315)
            const uint64_t waves_per_pack = std::max<uint64_t>(1, w_st / 128);
            const uint64_t narrow_waves = pn.n_packs();
            const uint64_t keep = std::max<uint64_t>(512, narrow_waves < 2048 ? (2048 - narrow_waves) / waves_per_pack : 0);
            const uint64_t hops_for_packs = wide_nodes / ((uint64_t)w_st * keep * 6 / 10 + 1);
            if (hops_for_packs >= longest + 2) {
                pw.max_hops = (uint32_t)std::min<uint64_t>(hops_for_packs, 3ull * longest);
                pw.width = w_st;
                WWe = w_st;
                L.wide_pack_width = WWe;
            }
        }
    }
    // (chained wide packs in the same order — widest first — save 0.6 % of their hops and were measured 1-3 % slower: input order stays)
    form(pw, order_w);
    form(ph, order_h);

    lap("classify + pack formation");
    // ---- pass 3: emit ----------------------------------------------------------------------
    uint64_t total_slots = 0;
    for (uint32_t u : pn.flat_used) total_slots += u;
    const uint64_t narrow_slots = total_slots;
    for (uint32_t u : pw.flat_used) total_slots += u;
    for (uint32_t u : ph.flat_used) total_slots += u;
    if (total_slots >= std::numeric_limits<uint32_t>::max()) {
        err = "more than 2^32 node slots";
        return BDDMMA_ERR_UNSUPPORTED;
    }
    L.n_slots = total_slots;
    L.narrow_slots = (uint32_t)narrow_slots;
    L.narrow_words.assign(narrow_slots, nw_pad_word(W));
    L.wide_words.assign(total_slots - narrow_slots, 0);
    L.layer_var.assign(Lin, 0);
    L.layer_bdd.assign(Lin, 0);
    L.bdd_root_slot.assign(n_bdds, 0);
    L.nodes_per_hop.assign(L.n_hops, 0);
    L.layers_per_hop.assign(L.n_hops, 0);
    if (keep_debug_maps) L.slot_to_instr.assign(total_slots, std::numeric_limits<uint64_t>::max());
    std::vector<uint32_t> in_layer_to_internal(Lin);

    uint32_t slot_cursor = 0, layer_cursor = 0;
    std::vector<std::vector<uint64_t>> t_nodes(par.nt, std::vector<uint64_t>(L.n_hops, 0)), t_layers(par.nt, std::vector<uint64_t>(L.n_hops, 0));
    auto emit = [&](PackBuilder& pb, const std::vector<uint32_t>& order, PackSet& ps, bool wide) {
        const uint32_t P = pb.n_packs();
        ps.pack_hop_ptr.resize(P + 1);
        ps.pack_steps = pb.pack_steps;
        ps.hop_root = pb.flat_root;  // wide packs: staggered too (k_*_wide2); huge packs: all NO_ROOT
        // offsets of every (pack, hop) record first (a running sum), then the packs are emitted independently
        const uint32_t n_rec = P ? pb.pack_hop_ptr[P] : 0;
        ps.hop_node_off.resize((size_t)n_rec + 1);
        ps.hop_layer_off.resize((size_t)n_rec + 1);
        for (uint32_t p = 0; p < P; ++p) ps.pack_hop_ptr[p] = pb.pack_hop_ptr[p];
        ps.pack_hop_ptr[P] = n_rec;
        for (uint32_t r = 0; r < n_rec; ++r) {
            ps.hop_node_off[r] = slot_cursor;
            ps.hop_layer_off[r] = layer_cursor;
            slot_cursor += pb.flat_used[r];
            layer_cursor += pb.flat_nlayers[r];
        }
        ps.hop_node_off[n_rec] = slot_cursor;
        ps.hop_layer_off[n_rec] = layer_cursor;
        par.run(P, [&](uint64_t p0, uint64_t p1, unsigned t) {  // min_parallel below: a pack is hundreds of nodes
            std::vector<uint32_t> lcount, gcur, gfirst;
            for (uint32_t p = (uint32_t)p0; p < (uint32_t)p1; ++p) {
                const uint32_t base_q = pb.pack_hop_ptr[p];
                const uint32_t H = pb.pack_hop_ptr[p + 1] - base_q;
                lcount.assign(H, 0);
                gcur.assign(H, 0);    // lane group of the hop's latest layer, and the hop-local index of that group's first layer
                gfirst.assign(H, 0);
                for (uint32_t k = pb.pack_first_bdd[p]; k < pb.pack_first_bdd[p + 1]; ++k) {
                    const uint64_t b = order[k];
                    const uint32_t l0 = bdd_lay_ptr[b], n = bdd_lay_ptr[b + 1] - l0;
                    const uint32_t d0 = bdd_hop0[b];  // hop of the pack at which this BDD starts
                    for (uint32_t hb = 0; hb < n; ++hb) {
                        const uint32_t h = d0 + hb;   // hop of the pack; hb = depth inside the BDD
                        const uint32_t l = l0 + hb;
                        const uint64_t f = lay_first[l], e = layer_end(b, l);
                        const uint32_t lloc = lcount[h]++;
                        // index of the layer inside its 64-lane group of the hop (the BDDs of a pack are placed left to right)
                        if (!wide) {
                            const uint32_t grp = lay_pos[l] / 64;
                            if (grp != gcur[h]) { gcur[h] = grp; gfirst[h] = lloc; }
                        }
                        const uint32_t lgrp = wide ? 0u : lloc - gfirst[h];
                        const uint32_t lg = ps.hop_layer_off[base_q + h] + lloc;
                        in_layer_to_internal[l] = lg;
                        L.layer_var[lg] = (int32_t)instr[f].index;
                        L.layer_bdd[lg] = (int32_t)b;
                        t_nodes[t][hb] += e - f;   // per-hop statistics count by depth inside the BDD, as the reference's hops do
                        t_layers[t][hb] += 1;
                        const bool last = (hb + 1 == n);
                        const uint64_t nf = last ? 0 : lay_first[l + 1];
                        const uint32_t npos = last ? 0 : lay_pos[l + 1];
                        for (uint64_t i = f; i < e; ++i) {
                            const uint32_t j = lay_pos[l] + (uint32_t)(i - f);
                            const uint32_t slot = ps.hop_node_off[base_q + h] + j;
                            if (hb == 0) L.bdd_root_slot[b] = slot;
                            if (keep_debug_maps) L.slot_to_instr[slot] = i;
                            uint64_t ch[2];
                            for (int side = 0; side < 2; ++side) {
                                const uint64_t c = side ? instr[i].hi : instr[i].lo;
                                if (is_bot(instr[c])) ch[side] = wide ? WW_BOT : nw_bot(W);
                                else if (is_top(instr[c])) ch[side] = wide ? WW_TOP : nw_top(W);
                                else ch[side] = npos + (c - nf);
                            }
                            if (wide) {
                                L.wide_words[slot - narrow_slots] =
                                    ch[0] | (ch[1] << WW_CHILD_BITS) | ((uint64_t)lloc << (2 * WW_CHILD_BITS)) | (i == f ? WW_HEAD : 0);
                            } else {
                                L.narrow_words[slot] = (uint32_t)ch[0] | ((uint32_t)ch[1] << NW_CHILD_BITS) |
                                                       ((uint32_t)(i - f) << NW_POS_SHIFT) | (lgrp << NW_LIDX_SHIFT) | (e - f == 2 ? NW_TWO : 0u);
                            }
                        }
                    }
                }
            }
        }, 64);
    };
    emit(pn, order_n, L.narrow, false);
    emit(pw, order_w, L.wide, true);
    emit(ph, order_h, L.huge, true);
    for (unsigned t = 0; t < par.nt; ++t)
        for (uint64_t h = 0; h < L.n_hops; ++h) {
            L.nodes_per_hop[h] += t_nodes[t][h];
            L.layers_per_hop[h] += t_layers[t][h];
        }
    L.n_nodes = L.n_input_nodes - 2 * n_bdds;
    lap("emit words");
    {   // structure templates: store every distinct pack word sequence once
        const PackSet& N = L.narrow;
        const uint32_t P = N.n_packs();
        L.narrow_word_off.assign(P, 0);
        std::unordered_map<uint64_t, std::vector<uint32_t>> seen;  // hash -> packs already stored with that hash
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t s0 = N.hop_node_off[N.pack_hop_ptr[p]], s1 = N.hop_node_off[N.pack_hop_ptr[p + 1]];
            uint64_t h = 1469598103934665603ull ^ (s1 - s0);
            for (uint32_t i = s0; i < s1; ++i) h = (h ^ L.narrow_words[i]) * 1099511628211ull;
            bool found = false;
            for (uint32_t o : seen[h]) {
                const uint32_t o0 = N.hop_node_off[N.pack_hop_ptr[o]], o1 = N.hop_node_off[N.pack_hop_ptr[o + 1]];
                if (o1 - o0 == s1 - s0 && std::equal(L.narrow_words.begin() + s0, L.narrow_words.begin() + s1, L.narrow_words.begin() + o0)) {
                    // the hop boundaries must coincide too: the kernels address words by (slot - first slot of the pack)
                    bool same_hops = N.pack_hop_ptr[p + 1] - N.pack_hop_ptr[p] == N.pack_hop_ptr[o + 1] - N.pack_hop_ptr[o];
                    for (uint32_t q = 0; same_hops && q <= N.pack_hop_ptr[p + 1] - N.pack_hop_ptr[p]; ++q)
                        same_hops = N.hop_node_off[N.pack_hop_ptr[p] + q] - s0 == N.hop_node_off[N.pack_hop_ptr[o] + q] - o0;
                    if (same_hops) {
                        L.narrow_word_off[p] = L.narrow_word_off[o];
                        found = true;
                        break;
                    }
                }
            }
            if (!found) {
                L.narrow_word_off[p] = (uint32_t)L.narrow_words_unique.size();
                L.narrow_words_unique.insert(L.narrow_words_unique.end(), L.narrow_words.begin() + s0, L.narrow_words.begin() + s1);
                seen[h].push_back(p);
            }
        }
    }

    lap("structure templates");
    // ---- variable -> layers CSR, sorted by (variable, bdd) (bdd_cuda_base.cu:379-391) ---------
    L.var_ptr.assign(L.n_vars + 1, 0);
    for (uint64_t v = 0; v < L.n_vars; ++v) L.var_ptr[v + 1] = L.var_ptr[v] + (uint32_t)L.num_bdds_per_var[v];
    L.var_layers.assign(Lin, 0);
    {
        std::vector<uint32_t> cursor(L.var_ptr.begin(), L.var_ptr.end() - 1);
        for (uint32_t l = 0; l < Lin; ++l) {  // input layers are BDD-major => stable by bdd
            const uint32_t lg = in_layer_to_internal[l];
            L.var_layers[cursor[L.layer_var[lg]]++] = lg;
        }
    }
    lap("variable CSR");
    // ---- variable <-> layer exchange tables (see layout.hpp, struct Exchange) --------------------
    {
        Exchange& X = L.ex;
        // default bin: as many variables as a 128 KiB LDS tile holds (2 REAL each), but at least ~256 bins
        // so that the exchange kernel (one workgroup per bin) has enough workgroups to spread over the CUs
        const uint32_t max_vb = 9728u;  // k_exchange_reduce: 2 * vars_per_bin <= 19 * 1024 threads; 152 KiB of double accumulators
        // ~256 bins (one workgroup per CU), but at least 512 variables per bin — unless the variables have so many
        // layers that such a bin would hold several 12 K-entry chunks (long rows: V small, L large): then smaller bins
        uint32_t auto_vb = (uint32_t)(((L.n_vars + 255) / 256 + 63) / 64 * 64);
        const uint64_t one_chunk_vars = Lin ? 12288ull * L.n_vars / Lin : 1024;
        // (512: measured at V = 100 k — exchange 5.2 / 4.6 / 7.1 us with 256 / 512 / 1024 variables per bin, while the sweeps, whose runs
        // in the entry arrays shrink with the bins, take 23.8 / 22.9 / 22.1 us)
        const uint32_t min_vb = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(64, one_chunk_vars / 64 * 64));
        auto_vb = std::min(std::max(auto_vb, min_vb), max_vb);
        // up to 1024 variables per bin the exchange runs its 256-thread variant (kernels.hpp: EXS_*), which is also the better choice
        // a little beyond (V = 300-400 k: 1024 per bin 9.1-9.7 us, the 1024-thread kernel on 1216-1600 per bin 10.8-11.2 us)
        if (auto_vb > 1024 && auto_vb <= 2048) auto_vb = 1024;
        // double, large V: bins of 2048 (512-thread exchange workgroups, two per CU, whose load / accumulate / store phases overlap) instead of
        // ~256 bins of one 1024-thread workgroup per CU: the exchange 25.9 -> 22.3 us while the sweeps lose 2-5 us to shorter runs in the entry
        // arrays: 10.5 M nodes 4 165 / 4 263 -> 4 315 / 4 334 it/s (two alternating runs on one box).  Float loses 1 % with the same change.
        if (real_size == 8 && auto_vb > 2048) auto_vb = 2048;
        X.vars_per_bin = opts && opts->vars_per_bin ? opts->vars_per_bin : auto_vb;
        // stage groups hold <= stage_cap layers of one pack; the default is the largest pack's layer count (one group per pack) up to
        // 640, so that small packs do not reserve LDS staging space they never use
        uint32_t auto_cap = W;
        for (uint32_t p = 0; p < L.narrow.n_packs(); ++p)
            auto_cap = std::max(auto_cap, L.narrow.hop_layer_off[L.narrow.pack_hop_ptr[p + 1]] - L.narrow.hop_layer_off[L.narrow.pack_hop_ptr[p]]);
        const uint32_t raw_cap = auto_cap;   // the largest pack's layers: beyond 640 its packs have several stage groups
        auto_cap = std::min<uint32_t>(640, (auto_cap + 63) / 64 * 64);
        X.stage_cap = opts && opts->stage_cap ? opts->stage_cap : auto_cap;
        if (X.vars_per_bin < 64 || X.vars_per_bin > max_vb) {
            err = "vars_per_bin must be in [64, 9728]";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        if (X.stage_cap < W || X.stage_cap > 640) {
            err = "stage_cap must be in [pack_width, 640]";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        X.n_bins = (uint32_t)((L.n_vars + X.vars_per_bin - 1) / X.vars_per_bin);
        // stage groups: runs of hops of a narrow pack holding <= stage_cap layers
        const PackSet& N = L.narrow;
        const uint32_t Pn = N.n_packs();
        // packs per workgroup (the rules read like this since rounds 2-5; a function of the stage groups' size because the rule for that size below asks
        // what they would choose)
        auto choose_wpb = [&](uint32_t stage_cap) -> uint32_t {
            uint32_t wpb = opts && opts->waves_per_block ? opts->waves_per_block : 4;
            // small instances: keep at least ~512 workgroups so that every CU has work
            if (!(opts && opts->waves_per_block))
                while (wpb > 1 && Pn / wpb < 512) wpb /= 2;
            // large instances: 8.  The entries a workgroup stages per bin form one run in the entry arrays; with four packs per workgroup a run is
            // (layers of four packs) / bins entries long — 10 at 10.5 M nodes float, 1 at 105 M, where the sweeps fall from 5.7 to 4.2 TB/s on their
            // counter bytes.  Eight packs double it.  Measured (tools/kbench.py --wpb 4 / 8, one box each, it/s): 10.5 M nodes float (run 10.4) 8 546 /
            // 8 488, double (5.2) 4 225 / 4 251; 15.8 M 4 437 / 4 414 and 2 659 / 2 707; 21 M 3 516 / 3 557 and 1 832 / 1 868; 42 M 1 620 / 1 632 and
            // 884 / 906; 105 M 596 / 630 and 314 / 331.  With the sweeps starting from the resident headers (PackDev::hdr_pack; later in the round)
            // the double 10.5 M case is a tie (4 276 / 4 273, 4 319 / 4 263: four packs keep the per-lane records, eight exceed their 64 KB), 105 M
            // still 599 / 640 and 318 / 332.  Rule: 8 when the run of four packs is shorter than 4.5 entries.
            if (!(opts && opts->waves_per_block) && wpb == 4 && Pn >= 8 * 512 && X.n_bins > 0) {
                const double run4 = (double)L.n_layers / ((double)Pn / 4.0) / (double)X.n_bins;
                // (LDS of a sweep workgroup of eight, as SolverT::init adds it up: staging pairs + frontier / potentials / hop window per wave + the
                // segmented-minimum scratch)
                const uint64_t lds8 = 8ull * stage_cap * 2 * real_size + 8ull * (3ull * (W + 2) * real_size + 2 * 64 * 4 + 2) + 8ull * 128 * real_size;
                if (run4 < 4.5 && lds8 <= 150 * 1024) wpb = 8;
            }
            // Instances of a few thousand narrow packs and nothing else are candidates for the resident sweeps, whose workgroups should be all in
            // flight at once: one pack per workgroup packs the CUs' LDS best (a wave's 12-21 KB region; k_fwd_res2).  The entry arrays of such an
            // instance stay in cache, so the longer runs of cooperative staging buy nothing there (1.05 M nodes, streaming: 28.0 k it/s with 1, 2 or 4).
            // (64-slot packs, up to ~1.45 x what the chip holds at once in float; beyond that the streaming sweeps run, which want their 4: 3.1 M
            // nodes 16.3 k it/s with 4, 16.0 k with 1; 4.2 M, packs of 128: 14.6 k / 13.8 k)
            // Only where those sweeps can be chosen at all (SolverT::init: layers of at most two nodes, resident sweeps not switched off) — the
            // streaming sweeps of any other instance keep their four packs per workgroup and the staging runs that go with them.  3 700 packs
            // on 256 CUs, scaled with the CU count.
            bool two_node_layers = true;
            for (uint8_t st : L.narrow.pack_steps) two_node_layers = two_node_layers && st < 2;
            if (!(opts && opts->waves_per_block) && W == 64 && (uint64_t)Pn * 256 <= 3700ull * chip.n_cus && L.wide.n_packs() == 0 && L.huge.n_packs() == 0 &&
                two_node_layers && !(opts && opts->resident_sweeps == 1))
                wpb = 1;
            // Instances with a sizeable share of wide packs (>= 10 % of the node slots): their solve sweeps share the narrow packs' launch
            // (k_fwd_mixed / k_bwd_mixed), so a wide pack gets 64 * waves_per_block threads — more than its hop width leaves threads idle
            // behind every barrier.  Knapsack benchmark (wide packs of 65-77 nodes): 4 -> 2 packs per workgroup 16.1 k -> 18.1 k it/s
            // (float), 14.5 k -> 16.5 k (double); 1: 18.0 k / 15.4 k; 8: 12.8 k / 10.7 k.
            if (!(opts && opts->waves_per_block) && L.wide.n_packs() > 0 && !L.wide.hop_node_off.empty()) {
                const uint64_t wide_slots = L.wide.hop_node_off.back() - L.wide.hop_node_off.front();
                if (wide_slots * 10 >= (uint64_t)L.n_slots) {
                    uint32_t fit = 1;
                    while (fit < 4 && 64u * fit < L.wide_pack_width) fit *= 2;
                    wpb = std::min(wpb, fit);
                }
            }
            // Staggered packs: one pack per workgroup.  The waves of a workgroup share its LDS and meet at the barriers of the staging rounds, and
            // staggered packs differ in length and in what their hops cost (BDDs start and end anywhere inside them), so a workgroup of four lives as
            // long as its slowest pack; uniform packs finish together.  Measured on the round-3 kernels, 10 M nodes, 4 -> 1 packs per workgroup, it/s
            // float / double: 20 k knapsack + 250 k covering rows 5 458 -> 6 110 / 2 862 -> 3 293, 10 k + 400 k 5 641 -> 6 886 / 2 961 -> 3 512, 30 k +
            // 100 k 5 342 -> 5 234 / 3 152 -> 3 418, 40 k knapsack rows of 14 variables 4 734 -> 4 890 / 3 300 -> 3 740, of 10 variables 11 595 ->
            // 12 972 / 9 885 -> 10 173.  Not staggered: random set cover keeps 4 (8 127 vs 7 717 / 4 351 vs 3 926 with one), the 1 M-node knapsack
            // instance its 2 (19 189 vs 18 475).  (The hop counts alone do not tell: grouping staggered packs by four pads them by < 10 %.)
            if (!(opts && opts->waves_per_block) && wpb > 1) {
                bool staggered = false;
                for (uint16_t r : N.hop_root)
                    if (r != NO_ROOT) { staggered = true; break; }
                if (staggered) wpb = 1;
            }
            return wpb;
        };
        // Several stage groups per pack (rows of more than ~20 variables in 64-slot packs) in ONE round of workgroups where a smaller staging area allows
        // it.  The sweeps of such instances are latency-bound and bimodal: ~105 us when every workgroup of the launch is resident at once, ~135 us when a
        // second, nearly empty round follows (double, 10.5 M nodes; profiles/r06_stage_groups.txt: 33 shapes x 6 group sizes).  What decides is the LDS
        // of the forward sweep's workgroup — packs per workgroup x (stage_cap pairs + the second generation's static arrays, (4 (W + 2) + 2 W) values + 768 B)
        // + ~0.6 KB the hardware keeps per workgroup (fitted: 14 workgroups of 11 072 B share a CU, 13 of 12 096 B do not) — against the CU's LDS, and the
        // register budget's 16 (double) / 20 (float) waves per CU.  Fewer, larger groups win whenever the launch fits one round anyway (fewer staging
        // rounds per pack), and a change from three rounds to two measured as a loss (-3..-5 %), so: only when the rule's size needs more than one round and
        // a smaller one needs exactly one — double 10.5 M nodes, rows of 40 / 44 / 50 / 56 variables +11 / +14 / +22 / +11 %, 5.25 M nodes rows of 24 / 28
        // +19 / +11 %, 21 M nodes rows of 80 / 100 +6 / +16 %; float is never LDS-limited here and keeps its 640.  64-slot packs, narrow packs only.
        if (!(opts && opts->stage_cap) && Pn && W == 64 && L.wide.n_packs() == 0 && L.huge.n_packs() == 0 && raw_cap > X.stage_cap) {
            const uint32_t wpb = choose_wpb(X.stage_cap);
            const uint64_t n_wg = ((uint64_t)Pn + wpb - 1) / wpb;
            auto per_cu = [&](uint32_t cap) {
                const uint64_t per_wg = (uint64_t)wpb * ((uint64_t)cap * 2 * real_size + (4ull * (W + 2) + 2ull * W) * real_size + 768) + 640;
                return std::min<uint64_t>(chip.lds_bytes / per_wg, std::max<uint32_t>(1, (real_size == 8 ? 16u : 20u) / wpb));
            };
            auto one_round = [&](uint32_t cap) { return n_wg <= per_cu(cap) * chip.n_cus; };
            const uint32_t cap0 = X.stage_cap;
            if (!one_round(cap0))
                for (uint32_t cap = cap0 - 64; cap >= 256 && cap >= W; cap -= 64)
                    if (one_round(cap)) {
                        if (choose_wpb(cap) == wpb) X.stage_cap = cap;
                        break;
                    }
            // ... and where no size gets the launch into one round: one step down (640 -> 576 layers) if that is what lets a CU hold a THIRD workgroup
            // instead of two — four packs per workgroup in double are 57 KB with groups of 640 layers, 53 KB with 576 — half as many resident waves
            // again for groups 10 % smaller (10.5 M nodes, rows of 28 / 32 / 36 variables: +7 %; rows of 24: 0)
            if (X.stage_cap == cap0 && !one_round(cap0) && cap0 >= 128 + W && per_cu(cap0) == 2 && per_cu(cap0 - 64) == 3 && choose_wpb(cap0 - 64) == wpb)
                X.stage_cap = cap0 - 64;
        }
        X.pack_group_ptr.assign(Pn + 1, 0);
        std::vector<uint32_t> layer_group(Lin, 0);
        const uint32_t narrow_layers = Pn ? N.hop_layer_off.back() : 0;  // narrow layers come first
        for (uint32_t p = 0; p < Pn; ++p) {
            X.pack_group_ptr[p] = (uint32_t)X.grp_hop_end.size();
            const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1];
            uint32_t acc = 0;
            for (uint32_t q = q0; q < q1; ++q) {
                const uint32_t nl = N.hop_layer_off[q + 1] - N.hop_layer_off[q];
                if (acc + nl > X.stage_cap && acc > 0) {
                    X.grp_hop_end.push_back(q);
                    acc = 0;
                }
                acc += nl;
                const uint32_t g = (uint32_t)X.grp_hop_end.size();
                for (uint32_t l = N.hop_layer_off[q]; l < N.hop_layer_off[q + 1]; ++l) layer_group[l] = g;
            }
            X.grp_hop_end.push_back(q1);
        }
        X.pack_group_ptr[Pn] = (uint32_t)X.grp_hop_end.size();
        const uint32_t G = (uint32_t)X.grp_hop_end.size();
        for (uint32_t l = narrow_layers; l < Lin; ++l) layer_group[l] = G;  // wide layers: one pseudo group, last in every bin

        // a group's layers are contiguous in layer order; entry order: by (bin, group, layer)
        X.grp_layer_off.assign(G + 1, 0);
        for (uint32_t l = 0; l < narrow_layers; ++l) X.grp_layer_off[layer_group[l] + 1] = l + 1;
        for (uint32_t g = 0; g < G; ++g)
            if (X.grp_layer_off[g + 1] < X.grp_layer_off[g]) X.grp_layer_off[g + 1] = X.grp_layer_off[g];
        // entries by (variable, bdd) (layout.hpp): only on request.  Measured at 1 / 2 / 4 M nodes: the exchange launch drops from 9.7 to
        // 5.3 us, but every sweep gains 6 / 14 / 35 us because its gathers and scatters in the entry arrays lose their runs — even
        // with the arrays in L2, a sweep's staging time follows the number of cache lines it touches.
        X.entry_by_var = opts && opts->exchange_by_variable == 2;
        // Entry order (bin, group, layer): the layers are already in (group, layer) order (narrow packs first, wide layers form the
        // last pseudo group), so this is a stable counting sort by bin — chunks of the sequence are counted and scattered by the host
        // threads independently.  With entry_by_var the sequence is var_layers ((variable, bdd) order; bins are ranges of variables).
        (void)layer_group;
        X.bin_ptr.assign(X.n_bins + 1, 0);
        X.evar.assign(Lin, 0);
        X.bvar.assign(Lin, 0);
        X.lpos.assign(Lin, 0);
        {
            const uint32_t VB = X.vars_per_bin, NB = X.n_bins;
            auto layer_at = [&](uint64_t k) -> uint32_t { return X.entry_by_var ? L.var_layers[k] : (uint32_t)k; };
            std::vector<std::vector<uint32_t>> hist(par.nt, std::vector<uint32_t>(NB, 0));
            std::vector<uint64_t> chunk_begin(par.nt + 1, Lin);
            par.run(Lin, [&](uint64_t k0, uint64_t k1, unsigned t) {
                chunk_begin[t] = k0;
                for (uint64_t k = k0; k < k1; ++k) hist[t][(uint32_t)L.layer_var[layer_at(k)] / VB]++;
            });
            for (uint32_t bb = 0; bb < NB; ++bb) {
                uint32_t run = X.bin_ptr[bb];
                for (unsigned t = 0; t < par.nt; ++t) {
                    const uint32_t c = hist[t][bb];
                    hist[t][bb] = run;  // first entry of (chunk t, bin bb)
                    run += c;
                }
                X.bin_ptr[bb + 1] = run;
            }
            par.run(Lin, [&](uint64_t k0, uint64_t k1, unsigned t) {
                for (uint64_t k = k0; k < k1; ++k) {
                    const uint32_t l = layer_at(k);
                    const uint32_t v = (uint32_t)L.layer_var[l], bb = v / VB;
                    const uint32_t e = hist[t][bb]++;
                    X.evar[e] = v;
                    X.bvar[e] = (uint16_t)(v - bb * VB);
                    X.lpos[l] = e;
                }
            });
        }
        lap("entry tables");
        // cooperative staging tables
        // default: 4 packs per workgroup (measured after the node words became shared: 4 beats 8 in float by 5-8 %:
        // 28 KB of LDS per workgroup instead of 57 KB, i.e. 5 instead of 4 waves per SIMD)
        X.waves_per_block = choose_wpb(X.stage_cap);
        if (X.waves_per_block != 1 && X.waves_per_block != 2 && X.waves_per_block != 4 && X.waves_per_block != 8) {
            err = "waves_per_block must be 1, 2, 4 or 8";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        if ((uint64_t)X.waves_per_block * X.stage_cap > 65535) {
            err = "waves_per_block * stage_cap must be < 65536";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        {
            const uint32_t WPB = X.waves_per_block;
            const uint32_t n_quads = (Pn + WPB - 1) / WPB;
            // record offsets first (rounds per quad, items per round), then the quads fill their ranges independently
            X.quad_round_ptr.assign(n_quads + 1, 0);
            for (uint32_t Q = 0; Q < n_quads; ++Q) {
                uint32_t rounds = 0;
                for (uint32_t w = 0; w < WPB && Q * WPB + w < Pn; ++w) {
                    const uint32_t p = Q * WPB + w;
                    rounds = std::max(rounds, X.pack_group_ptr[p + 1] - X.pack_group_ptr[p]);
                }
                X.quad_round_ptr[Q + 1] = X.quad_round_ptr[Q] + rounds;
            }
            const uint32_t n_rounds = X.quad_round_ptr[n_quads];
            X.cs_ptr.assign((size_t)n_rounds + 1, 0);
            for (uint32_t Q = 0; Q < n_quads; ++Q)
                for (uint32_t r = X.quad_round_ptr[Q]; r < X.quad_round_ptr[Q + 1]; ++r) {
                    const uint32_t k = r - X.quad_round_ptr[Q];
                    uint32_t items = 0;
                    for (uint32_t w = 0; w < WPB && Q * WPB + w < Pn; ++w) {
                        const uint32_t p = Q * WPB + w;
                        const uint32_t g = X.pack_group_ptr[p] + k;
                        if (g < X.pack_group_ptr[p + 1]) items += X.grp_layer_off[g + 1] - X.grp_layer_off[g];
                    }
                    X.cs_ptr[r + 1] = X.cs_ptr[r] + items;
                }
            X.cs_entry.assign(narrow_layers, 0);
            X.cs_slot.assign(narrow_layers, 0);
            par.run(n_quads, [&](uint64_t Q0, uint64_t Q1, unsigned) {  // (a quad is ~2 500 items to gather and sort)
                std::vector<std::pair<uint32_t, uint16_t>> items;
                for (uint32_t Q = (uint32_t)Q0; Q < (uint32_t)Q1; ++Q)
                    for (uint32_t r = X.quad_round_ptr[Q]; r < X.quad_round_ptr[Q + 1]; ++r) {
                        const uint32_t k = r - X.quad_round_ptr[Q];
                        items.clear();
                        for (uint32_t w = 0; w < WPB && Q * WPB + w < Pn; ++w) {
                            const uint32_t p = Q * WPB + w;
                            const uint32_t g = X.pack_group_ptr[p] + k;
                            if (g >= X.pack_group_ptr[p + 1]) continue;
                            for (uint32_t l = X.grp_layer_off[g]; l < X.grp_layer_off[g + 1]; ++l)
                                items.push_back({X.lpos[l], (uint16_t)(w * X.stage_cap + (l - X.grp_layer_off[g]))});
                        }
                        std::sort(items.begin(), items.end());
                        uint32_t o = X.cs_ptr[r];
                        for (const auto& it : items) {
                            X.cs_entry[o] = it.first;
                            X.cs_slot[o] = it.second;
                            ++o;
                        }
                    }
            }, 16);
        }
        lap("cooperative staging tables");
        X.vpos.assign(Lin, 0);
        par.run(Lin, [&](uint64_t k0, uint64_t k1, unsigned) {
            for (uint64_t k = k0; k < k1; ++k) X.vpos[k] = X.lpos[L.var_layers[k]];
        });

        // headers of the resident sweeps (layout.hpp: struct Resident)
        {
            Resident& Rz = L.res;
            const uint32_t WPB = X.waves_per_block;
            const uint32_t n_quads = (Pn + WPB - 1) / WPB;
            Rz.ok = Pn > 0;
            Rz.pack_hdr.assign((size_t)Pn * 8, 0);
            Rz.quad_hdr.assign((size_t)n_quads * 4, 0);
            for (uint32_t p = 0; p < Pn; ++p) {
                const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1];
                const uint32_t s0 = N.hop_node_off[q0], s1 = N.hop_node_off[q1];
                const uint32_t l0 = N.hop_layer_off[q0], l1 = N.hop_layer_off[q1];
                uint32_t* h = &Rz.pack_hdr[(size_t)p * 8];
                h[0] = s0; h[1] = s1 - s0; h[2] = l0; h[3] = l1 - l0; h[4] = q0;
                h[5] = (q1 - q0) | ((uint32_t)N.pack_steps[p] << 16);
                h[6] = L.narrow_word_off[p];
                Rz.max_slots = std::max(Rz.max_slots, s1 - s0);
                Rz.max_layers = std::max(Rz.max_layers, l1 - l0);
                if (q1 - q0 > 63 || X.pack_group_ptr[p + 1] - X.pack_group_ptr[p] != 1) Rz.ok = false;
                for (uint32_t q = q0; q < q1; ++q)
                    if (N.hop_root[q] != NO_ROOT) Rz.ok = false;  // the resident sweeps know roots at a pack's first hop only
            }
            for (uint32_t Q = 0; Q < n_quads; ++Q) {
                const uint32_t r0 = X.quad_round_ptr[Q], r1 = X.quad_round_ptr[Q + 1];
                uint32_t* h = &Rz.quad_hdr[(size_t)Q * 4];
                h[0] = r1 > r0 ? X.cs_ptr[r0] : 0;
                h[1] = r1 > r0 ? X.cs_ptr[r0 + 1] - X.cs_ptr[r0] : 0;
                h[2] = r1 - r0;
                if (r1 - r0 != 1) Rz.ok = false;
            }
        }
    }
    lap("vpos + resident headers");
    return BDDMMA_OK;
}

// layout.hpp: struct Res2Records
void build_res2_records(const HostLayout& L, uint32_t S, uint32_t ns, uint32_t nl, Res2Records& out)
{
    out = Res2Records();
    const PackSet& N = L.narrow;
    const uint32_t P = N.n_packs(), W = L.pack_width;
    if (!L.res.ok || P == 0 || W != 64 || L.narrow_word_off.size() != P) return;
    if (res2_wave_bytes(S, ns, nl) > 0xFFFFu || (uint64_t)nl * 2 * S >= RES2_NO_STORE || (uint64_t)(ns + 64) * S >= 0xFFFFu) return;
    const uint32_t T_OFF = res2_t_off(), F_OFF = res2_f_off(S, ns);
    out.rec_off.assign(P, 0);
    std::unordered_map<uint32_t, uint32_t> seen;  // word offset of a structure template -> its first record
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1], nh = q1 - q0;
        const uint32_t s0 = N.hop_node_off[q0], l0 = N.hop_layer_off[q0];
        out.max_hops = std::max(out.max_hops, nh);
        if (N.hop_node_off[q1] - s0 > ns || N.hop_layer_off[q1] - l0 > nl || N.pack_steps[p] > 1) { out = Res2Records(); return; }
        const auto it = seen.find(L.narrow_word_off[p]);
        if (it != seen.end()) { out.rec_off[p] = it->second; continue; }
        const uint32_t first = (uint32_t)(out.rec.size() / 4);
        seen.emplace(L.narrow_word_off[p], first);
        out.rec_off[p] = first;
        out.rec.resize(out.rec.size() + (size_t)nh * 64 * 4);
        uint32_t* r = &out.rec[(size_t)first * 4];
        const uint32_t* words = &L.narrow_words_unique[L.narrow_word_off[p]];
        for (uint32_t h = 0; h < nh; ++h) {
            const uint32_t nb = N.hop_node_off[q0 + h] - s0, n = N.hop_node_off[q0 + h + 1] - N.hop_node_off[q0 + h];
            const uint32_t nb_next = nb + n, lb = N.hop_layer_off[q0 + h] - l0;
            if (n > 64) { out = Res2Records(); return; }
            for (uint32_t j = 0; j < 64; ++j, r += 4) {
                const uint32_t w = j < n ? words[nb + j] : nw_pad_word(W);
                const uint32_t dummy = F_OFF + S * (ns + j);  // the lane's own entry behind the costs-from-root
                if (w & NW_PAD) {
                    const uint32_t bot = T_OFF + S * (ns + 1);
                    r[0] = bot | (bot << 16);
                    r[1] = dummy | (dummy << 16);
                    r[2] = 0u | ((S * (ns + j)) << 16);  // own slot: past the pack's slots, so the global store drops it
                    r[3] = RES2_PAD;
                    continue;
                }
                uint32_t t[2], f[2];
                for (int side = 0; side < 2; ++side) {
                    const uint32_t c = side ? (w >> NW_CHILD_BITS) & NW_CHILD_MASK : w & NW_CHILD_MASK;
                    if (c < W) {
                        t[side] = T_OFF + S * (nb_next + c);
                        f[side] = F_OFF + S * (nb_next + c);
                    } else {
                        t[side] = T_OFF + S * (ns + (c == nw_top(W) ? 0u : 1u));
                        f[side] = dummy;
                    }
                }
                const uint32_t ll = lb + ((w >> NW_LIDX_SHIFT) & NW_FIELD6);
                const bool head = ((w >> NW_POS_SHIFT) & NW_FIELD6) == 0;
                // a layout from before the pair alignment of PackBuilder::place (an older checkpoint): first-generation kernels
                if ((w & NW_TWO) && head && (j & 1u)) { out = Res2Records(); return; }
                r[0] = t[0] | (t[1] << 16);
                r[1] = f[0] | (f[1] << 16);
                r[2] = (ll * 2 * S) | ((S * (nb + j)) << 16);
                r[3] = (head ? ll * 2 * S : RES2_NO_STORE) | ((w & NW_TWO ? 1u : 0u) << 16);
            }
        }
    }
    out.ok = true;
}

// layout.hpp: struct SegExchange
void build_seg_exchange(const HostLayout& L, uint32_t T, uint32_t real_size, SegExchange& out)
{
    const uint32_t VEC = 16 / real_size;
    out = SegExchange();
    const Exchange& X = L.ex;
    const uint32_t NB = X.n_bins, VB = X.vars_per_bin;
    if (NB == 0 || X.entry_by_var || X.vpos.size() != L.n_layers || L.var_ptr.size() != L.n_vars + 1 || T == 0 || T % 64 != 0) return;
    out.threads = T;
    out.bin.assign(4 * (size_t)NB, 0);
    out.thr.assign(2 * (size_t)NB * T, 0);
    // pass 1 (per bin, independent): the deal — which variables a thread owns, in which order — and with it the bin's group count
    struct Deal { std::vector<uint32_t> order; uint32_t groups = 0, slots = 0; };  // order: the bin's variables with entries, thread-major
    std::vector<Deal> deal(NB);
    std::vector<uint32_t> thr_first((size_t)NB * T + 1, 0);  // index of a thread's first variable in its bin's order
    std::atomic<bool> fail{false};
    Par par;
    par.run(NB, [&](uint64_t b0, uint64_t b1, unsigned) {
        std::vector<uint32_t> vars, cnt_sorted;
        std::vector<std::vector<uint32_t>> own(T);
        for (uint64_t b = b0; b < b1; ++b) {
            const uint32_t v0 = (uint32_t)b * VB, v1 = (uint32_t)std::min<uint64_t>(L.n_vars, (uint64_t)v0 + VB);
            const uint32_t E = X.bin_ptr[b + 1] - X.bin_ptr[b];
            if (L.var_ptr[v1] - L.var_ptr[v0] != E || E + VEC >= 0xFFFFu) { fail = true; return; }
            vars.clear();
            for (uint32_t v = v0; v < v1; ++v)
                if (L.var_ptr[v + 1] > L.var_ptr[v]) vars.push_back(v);
            // most entries first (ties: by variable), dealt 0 .. T-1, T-1 .. 0, 0 .. T-1, ...
            std::stable_sort(vars.begin(), vars.end(), [&](uint32_t a, uint32_t c) { return L.var_ptr[a + 1] - L.var_ptr[a] > L.var_ptr[c + 1] - L.var_ptr[c]; });
            for (auto& o : own) o.clear();
            for (size_t i = 0; i < vars.size(); ++i) {
                const uint32_t round = (uint32_t)(i / T), pos = (uint32_t)(i % T);
                own[(round & 1u) ? T - 1 - pos : pos].push_back(vars[i]);
            }
            Deal& D = deal[b];
            D.order.reserve(vars.size());
            uint32_t longest = 0;
            for (uint32_t t = 0; t < T; ++t) {
                thr_first[b * T + t] = (uint32_t)D.order.size();
                uint32_t run = 0;
                for (uint32_t v : own[t]) { D.order.push_back(v); run += L.var_ptr[v + 1] - L.var_ptr[v]; }
                longest = std::max(longest, run);
            }
            if (longest > SEG_MAX_RUN) { fail = true; return; }
            D.groups = (longest + 7) / 8;
            D.slots = (uint32_t)vars.size();
        }
    }, 1);
    if (fail) { out = SegExchange(); return; }
    uint64_t first = 0;
    for (uint32_t b = 0; b < NB; ++b) out.max_groups = std::max(out.max_groups, deal[b].groups);
    out.max_groups = out.max_groups <= 2 ? 2 : 4;  // the two instantiations of k_exchange_seg<.., G>
    for (uint32_t b = 0; b < NB; ++b) deal[b].groups = out.max_groups;  // every run padded to it
    for (uint32_t b = 0; b < NB; ++b) {
        const uint32_t E = X.bin_ptr[b + 1] - X.bin_ptr[b];
        out.bin[4 * (size_t)b + 0] = (uint32_t)first;
        if (deal[b].slots >= (1u << 24)) { out = SegExchange(); return; }
        out.bin[4 * (size_t)b + 1] = deal[b].groups | (deal[b].slots << 8);
        out.bin[4 * (size_t)b + 2] = X.bin_ptr[b];
        out.bin[4 * (size_t)b + 3] = E;
        first += (uint64_t)deal[b].groups * T;
        out.max_entries = std::max(out.max_entries, E);
        out.max_slots = std::max(out.max_slots, deal[b].slots);
        out.max_groups = std::max(out.max_groups, deal[b].groups);
    }
    if (first * 16 >= (1ull << 32)) { out = SegExchange(); return; }
    out.perm.assign((size_t)first * 8, 0);
    // pass 2: the runs
    par.run(NB, [&](uint64_t b0, uint64_t b1, unsigned) {
        for (uint64_t b = b0; b < b1; ++b) {
            const Deal& D = deal[b];
            const uint32_t e0 = X.bin_ptr[b], E = X.bin_ptr[b + 1] - e0, G = D.groups;
            uint16_t* pb = &out.perm[(size_t)out.bin[4 * b] * 8];
            for (uint32_t t = 0; t < T; ++t) {
                const uint32_t i0 = thr_first[b * T + t], i1 = t + 1 < T ? thr_first[b * T + t + 1] : (uint32_t)D.order.size();
                uint32_t k = 0, ends = 0;
                for (uint32_t i = i0; i < i1; ++i) {
                    const uint32_t v = D.order[i];
                    for (uint32_t q = L.var_ptr[v]; q < L.var_ptr[v + 1]; ++q, ++k) pb[((size_t)(k / 8) * T + t) * 8 + k % 8] = (uint16_t)(X.vpos[q] - e0);
                    ends |= 1u << (k - 1);
                }
                for (; k < G * 8; ++k) pb[((size_t)(k / 8) * T + t) * 8 + k % 8] = (uint16_t)((E + VEC - 1) / VEC * VEC);  // holds 0 (k_exchange_seg)
                out.thr[2 * ((size_t)b * T + t) + 0] = ends;
                out.thr[2 * ((size_t)b * T + t) + 1] = i0;  // slots are numbered along `order`
            }
        }
    }, 1);
    out.ok = true;
}

// layout.hpp: struct StreamRecords
void build_stream_records(const HostLayout& L, uint32_t S, StreamRecords& out)
{
    out = StreamRecords();
    const PackSet& N = L.narrow;
    const uint32_t P = N.n_packs(), W = L.pack_width;
    if (P == 0 || L.narrow_word_off.size() != P || N.hop_root.size() + 1 != N.hop_node_off.size()) return;
    if ((2u * W + 2u) * S >= RES2_NO_STORE || (uint64_t)W * 2 * S >= RES2_NO_STORE) return;
    out.rec_off.assign(P, 0);
    std::unordered_map<uint32_t, uint32_t> seen;  // word offset of a structure template -> its first record
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1], nh = q1 - q0;
        const uint32_t s0 = N.hop_node_off[q0];
        const auto it = seen.find(L.narrow_word_off[p]);
        if (it != seen.end()) { out.rec_off[p] = it->second; continue; }
        const uint64_t first = out.rec.size() / 4;
        if ((first + (uint64_t)nh * W) * 16 >= (1ull << 31)) { out = StreamRecords(); return; }
        seen.emplace(L.narrow_word_off[p], (uint32_t)first);
        out.rec_off[p] = (uint32_t)first;
        out.rec.resize(out.rec.size() + (size_t)nh * W * 4);
        uint32_t* r = &out.rec[(size_t)first * 4];
        const uint32_t* words = &L.narrow_words_unique[L.narrow_word_off[p]];
        for (uint32_t h = 0; h < nh; ++h) {
            const uint32_t nb = N.hop_node_off[q0 + h] - s0, n = N.hop_node_off[q0 + h + 1] - N.hop_node_off[q0 + h];
            if (n > W) { out = StreamRecords(); return; }
            uint32_t grp_first = 0;  // layers of the hop's lower 64-lane groups
            for (uint32_t j = 0; j < W; ++j, r += 4) {
                if (j > 0 && j % 64 == 0) {  // hop-local index of this lane group's first layer = number of heads before it
                    grp_first = 0;
                    for (uint32_t i = 0; i < j && i < n; ++i)
                        if (!(words[nb + i] & NW_PAD) && ((words[nb + i] >> NW_POS_SHIFT) & NW_FIELD6) == 0) ++grp_first;
                }
                const uint32_t w = j < n ? words[nb + j] : nw_pad_word(W);
                const uint32_t dummy = (W + 2 + j) * S;
                if (w & NW_PAD) {
                    r[0] = ((W + 1) * S) | (((W + 1) * S) << 16);  // cost to terminal +inf on both sides
                    r[1] = dummy | (dummy << 16);
                    r[2] = 0u | (RES2_NO_STORE << 16);
                    r[3] = SREC_PAD;
                    continue;
                }
                const uint32_t lo = w & NW_CHILD_MASK, hi = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
                const uint32_t lq = (grp_first + ((w >> NW_LIDX_SHIFT) & NW_FIELD6)) * 2 * S;
                const bool head = ((w >> NW_POS_SHIFT) & NW_FIELD6) == 0;
                // packs of <= 2-node layers take the DPP swap of aligned pairs: a layout from before the pair alignment keeps the first generation
                if (N.pack_steps[p] <= 1 && (w & NW_TWO) && head && (j & 1u)) { out = StreamRecords(); return; }
                r[0] = (lo * S) | ((hi * S) << 16);                                        // sinks: W, W + 1 are the constant entries
                r[1] = (lo < W ? lo * S : dummy) | ((hi < W ? hi * S : dummy) << 16);
                r[2] = lq | ((head ? lq : RES2_NO_STORE) << 16);
                r[3] = ((w & NW_TWO) ? 1u : 0u) | (((w >> NW_POS_SHIFT) & NW_FIELD6) << 8);
            }
        }
    }
    out.ok = true;
}

void build_layer_records(const HostLayout& L, uint32_t S, LayerRecords& out)
{
    out = LayerRecords();
    const PackSet& N = L.narrow;
    const uint32_t P = N.n_packs(), W = L.pack_width;
    if (P == 0 || W != 128 || L.narrow_word_off.size() != P || N.hop_root.size() + 1 != N.hop_node_off.size()) return;
    if ((uint64_t)(W + 128u) * S >= LREC_NO_STORE) return;
    for (uint16_t r : N.hop_root)
        if (r != NO_ROOT) return;  // staggered packs
    out.rec_off.assign(P, 0);
    std::unordered_map<uint32_t, uint32_t> seen;  // word offset of a structure template -> its first record
    for (uint32_t p = 0; p < P; ++p) {
        if (N.pack_steps[p] > 1) { out = LayerRecords(); return; }  // a layer wider than two nodes
        const uint32_t q0 = N.pack_hop_ptr[p], q1 = N.pack_hop_ptr[p + 1], nh = q1 - q0;
        const uint32_t s0 = N.hop_node_off[q0];
        const auto it = seen.find(L.narrow_word_off[p]);
        if (it != seen.end()) { out.rec_off[p] = it->second; continue; }
        const uint64_t first = out.rec.size() / 4;
        if ((first + (uint64_t)nh * 64) * 16 >= (1ull << 31)) { out = LayerRecords(); return; }
        seen.emplace(L.narrow_word_off[p], (uint32_t)first);
        out.rec_off[p] = (uint32_t)first;
        out.rec.resize(out.rec.size() + (size_t)nh * 64 * 4);
        uint32_t* r = &out.rec[(size_t)first * 4];
        const uint32_t* words = &L.narrow_words_unique[L.narrow_word_off[p]];
        for (uint32_t h = 0; h < nh; ++h, r += 64 * 4) {
            const uint32_t nb = N.hop_node_off[q0 + h] - s0, n = N.hop_node_off[q0 + h + 1] - N.hop_node_off[q0 + h];
            const uint32_t nl = N.hop_layer_off[q0 + h + 1] - N.hop_layer_off[q0 + h];
            if (n > W || nl > 64) { out = LayerRecords(); return; }
            for (uint32_t l = 0; l < 64; ++l) {  // idle lanes
                const uint32_t bot = (W + 2 * l + 1) * S;
                r[4 * l + 0] = r[4 * l + 1] = bot | (bot << 16);
                r[4 * l + 2] = (W + 2 * l) * S;
                r[4 * l + 3] = LREC_NO_STORE | (LREC_NO_STORE << 16);
            }
            uint32_t l = 0;  // hop-local layer index = heads before this slot = the lane
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t w = words[nb + j];
                if (w & NW_PAD) continue;
                const uint32_t pos = (w >> NW_POS_SHIFT) & NW_FIELD6;
                const uint32_t lo = w & NW_CHILD_MASK, hi = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
                auto child = [&](uint32_t c, uint32_t lane) { return (c < W ? c : W + 2 * lane + (c - W)) * S; };  // W = TOP, W + 1 = BOT (nw_pad_word)
                if (pos == 0) {
                    if (l >= nl) { out = LayerRecords(); return; }
                    r[4 * l + 0] = child(lo, l) | (child(hi, l) << 16);
                    r[4 * l + 2] = (j * S) | (LREC_REAL << 16);
                    r[4 * l + 3] = (j * S) | (LREC_NO_STORE << 16);
                    ++l;
                } else {
                    // second node of the layer whose head is the slot before (two-node layers are placed in neighbouring slots)
                    if (pos != 1 || l == 0 || !(w & NW_TWO) || (r[4 * (l - 1) + 2] & 0xFFFFu) != (j - 1) * S) { out = LayerRecords(); return; }
                    const uint32_t k = l - 1;
                    r[4 * k + 1] = child(lo, k) | (child(hi, k) << 16);
                    r[4 * k + 2] |= LREC_TWO << 16;
                    r[4 * k + 3] = (r[4 * k + 3] & 0xFFFFu) | ((j * S) << 16);
                }
            }
            if (l != nl) { out = LayerRecords(); return; }
        }
    }
    out.ok = true;
}

void set_layout_threads(unsigned n) { g_layout_threads.store(n, std::memory_order_relaxed); }
void set_thread_layout_threads(unsigned n) { t_layout_threads = n; }

}  // namespace bddmma
