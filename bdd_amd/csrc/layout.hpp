// layout.hpp — host-side device-layout builder for the parallel-MMA hot path.
//
// Replaces the layout stage of the reference's GPU solver constructor
// (bdd_cuda_base.cu:31-46,55-391: initialize / populate_bdd_nodes / reorder_bdd_nodes /
// compress_bdd_nodes_to_layer / reorder_within_bdd_layers / set_special_nodes_* /
// find_primal_variable_ordering).  The reference sorts all nodes of all BDDs hop-major so
// that one kernel launch per hop can sweep them; that forces a grid-wide barrier (a kernel
// boundary) per hop.  Here BDDs are grouped into *packs*: a pack is a set of BDDs that one
// wavefront (narrow pack) or one workgroup (wide pack) walks hop by hop with the frontier in
// LDS, so a whole pass is ONE launch.  Inside a pack the nodes are stored hop-major SoA, so
// the 64 lanes of a wave read consecutive addresses at every hop.
//
// Pure C++ (no HIP): unit-testable on a CPU-only box through the bddmma_layout_* debug ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bdd_mma.h"

#ifndef __host__   // pure C++ translation units (layout.cpp, capi.cpp): the few helpers shared with the kernels are plain inline functions
#define __host__
#define __device__
#endif

namespace bddmma {

// ---- narrow node word (uint32) -------------------------------------------------------------
//  bits  0..8   lo child: local index inside the NEXT hop of the same pack, or the pack's TOP / BOT code
//  bits  9..17  hi child
//  bits 18..23  position of the node inside its layer (0 = head)
//  bits 24..29  index of the node's layer among the layers of its 64-lane group of the hop
//  bit  30      the layer has exactly two nodes (the DPP-pair segmented minimum)
//  bit  31      padding slot (no node)
// The sink codes are TOP = pack_width and BOT = pack_width + 1: the kernels keep two constant entries
// behind the LDS frontier arrays (cost-from-terminal 0 / +inf), so sink children need no branch.  The
// layer of a node is hop_layer_off[q] + (layers of the lower lane groups) + bits 24..29: two VALU; the first
// format stored the layer width instead and derived the index from the head ballot (v_mbcnt), nine.
constexpr uint32_t NW_CHILD_BITS = 9;
constexpr uint32_t NW_CHILD_MASK = (1u << NW_CHILD_BITS) - 1;
constexpr uint32_t NW_POS_SHIFT = 18, NW_LIDX_SHIFT = 24, NW_FIELD6 = 63;
constexpr uint32_t NW_TWO = 1u << 30;
constexpr uint32_t NW_PAD = 1u << 31;
constexpr uint32_t nw_top(uint32_t pack_width) { return pack_width; }
constexpr uint32_t nw_bot(uint32_t pack_width) { return pack_width + 1; }
constexpr uint32_t nw_pad_word(uint32_t pack_width)
{
    return NW_PAD | (nw_bot(pack_width) << NW_CHILD_BITS) | nw_bot(pack_width);
}
constexpr uint32_t NARROW_MAX_LAYER_WIDTH = 64;  // a layer never straddles a 64-lane group
constexpr uint32_t NARROW_MAX_PACK_WIDTH = 256;  // 64 * R, R <= 4
constexpr uint16_t NO_ROOT = 0xFFFFu;            // PackSet::hop_root

// ---- wide node word (uint64) ---------------------------------------------------------------
//  bits  0..20  lo child (local index in next hop) or WW_BOT / WW_TOP
//  bits 21..41  hi child
//  bits 42..62  layer index local to (pack, hop)
//  bit  63      head
constexpr uint32_t WW_CHILD_BITS = 21;
constexpr uint64_t WW_CHILD_MASK = (1ull << WW_CHILD_BITS) - 1;
constexpr uint64_t WW_BOT = WW_CHILD_MASK;
constexpr uint64_t WW_TOP = WW_CHILD_MASK - 1;
constexpr uint64_t WW_HEAD = 1ull << 63;

struct PackSet {
    // one entry per pack (+1): index of the pack's first (pack,hop) record
    std::vector<uint32_t> pack_hop_ptr;
    // one entry per (pack,hop) (+1): global offsets into the node / layer arrays
    std::vector<uint32_t> hop_node_off;
    std::vector<uint32_t> hop_layer_off;
    // per pack: number of shuffle-halving steps a segmented min needs = ceil(log2(max layer width))
    std::vector<uint8_t> pack_steps;
    // one entry per (pack,hop): local slot of the root of a BDD that STARTS at this hop of its pack, or NO_ROOT.  Every slot of a
    // pack's first hop is a root (the BDDs placed side by side from hop 0); further down at most one BDD starts per hop (staggered
    // packing, bddmma_options.pack_stagger).  Narrow packs only; empty for the other sets.
    std::vector<uint16_t> hop_root;
    uint32_t n_packs() const { return pack_hop_ptr.empty() ? 0 : (uint32_t)pack_hop_ptr.size() - 1; }
};

// Variable <-> layer exchange ("propagation blocking").  Every pass has to broadcast delta[var] to the
// layers of the variable and reduce mm[layer] back per variable.  Done naively these are two random
// 4-8 byte accesses per layer into multi-MB arrays — on the 10.5 M-node benchmark they move 3x the
// bytes of the streamed sweep data.  Instead both directions go through one array in *binned* order:
//   entry e = (bin of the variable, stage group of the layer, position inside the group)
// A bin is a contiguous range of `vars_per_bin` variables whose 2 REAL accumulators fit in LDS; a
// stage group is a run of hops of one narrow pack with <= stage_cap layers.  The sweep kernel loads
// the delta pairs of the group's layers into LDS (entry index per layer = lpos; the entries of one
// (bin, group) sit next to each other and next to those of the neighbouring groups, which run on the
// same XCD at the same time, so the lines are shared in L2), reads / overwrites them by the layer's
// position inside the group, and writes the min-marginal differences back the same way; the
// exchange kernel (one workgroup per bin) streams the bin's entries, accumulates per variable in LDS
// and streams the normalised delta pairs back.  No random global access is left on the hot path.
struct Exchange {
    uint32_t vars_per_bin = 0, n_bins = 0, stage_cap = 0;
    std::vector<uint32_t> bin_ptr;         // [n_bins+1] first entry of each bin
    std::vector<uint32_t> evar;            // [L] entry -> variable
    std::vector<uint16_t> bvar;            // [L] entry -> variable index local to its bin (exchange kernel: 2 B instead of 4)
    std::vector<uint32_t> lpos;            // [L] layer -> entry
    std::vector<uint32_t> vpos;            // [L] (variable,bdd)-sorted position -> entry (deterministic gather)
    std::vector<uint32_t> pack_group_ptr;  // [narrow packs + 1] first stage group of each narrow pack
    std::vector<uint32_t> grp_layer_off;   // [G+1] first layer of each group (a group's layers are contiguous)
    std::vector<uint32_t> grp_hop_end;     // [G] (pack,hop) record index one past the group's last hop
    // cooperative staging: `waves_per_block` consecutive narrow packs (a "quad") are swept by one workgroup;
    // in round k the workgroup stages the k-th stage groups of its packs together, so that the runs it
    // reads / writes in the entry arrays are (bin, quad)-contiguous instead of (bin, pack)-contiguous
    uint32_t waves_per_block = 0;
    std::vector<uint32_t> quad_round_ptr;  // [quads+1] first round record of each quad
    std::vector<uint32_t> cs_ptr;          // [rounds+1] first staged item of each (quad, round)
    std::vector<uint32_t> cs_entry;        // [narrow layers] staged item -> entry (ascending inside a round)
    std::vector<uint16_t> cs_slot;         // [narrow layers] staged item -> LDS slot = wave * stage_cap + (layer - group's first layer)
    // entry_by_var: entries ordered by (variable, bdd) instead of (bin, group, layer).  The entries of one variable are then
    // consecutive, and compute_delta + normalize_delta + the broadcast become one thread per variable over a contiguous run
    // (kernels.hpp: k_exchange_byvar): no LDS accumulators, no barriers, a third of the binned kernel's latency chain.  The price is
    // that the sweeps' accesses to the entry arrays lose their (bin, quad) runs, which only matters when those arrays do not stay in
    // cache — so this order is for small and medium instances.
    bool entry_by_var = false;
};

// Resident sweeps (kernels.hpp: k_fwd_res / k_bwd_res): a narrow pack whose node words, opposite-direction potentials and arc
// costs fit in its wave's LDS slice is loaded with a handful of 1 KiB direct-to-LDS copies in ONE memory round trip and swept out
// of LDS; the streaming kernels pay a dependent round trip per table and a latency-bound software pipeline per hop, which is
// what bounds small and medium instances.  Everything a wave needs to start is in one 32-byte header.
struct Resident {
    bool ok = false;                 // every narrow pack has <= 63 hops and a single stage group
    uint32_t max_slots = 0;          // largest pack, in node slots / layers
    uint32_t max_layers = 0;
    std::vector<uint32_t> pack_hdr;  // [8 per narrow pack] first slot, #slots, first layer, #layers, first hop record, #hops | steps << 16, word offset, 0
    std::vector<uint32_t> quad_hdr;  // [4 per quad] first staged item, #staged items, #rounds, 0
};

struct HostLayout {
    uint64_t n_bdds = 0, n_vars = 0, n_hops = 0;
    uint64_t n_input_nodes = 0;   // incl. terminals (reference nr_bdd_nodes())
    uint64_t n_slots = 0;         // node slots in the device arrays (non-terminal nodes + padding)
    uint64_t n_nodes = 0;         // non-terminal nodes
    uint64_t n_layers = 0;        // non-terminal layers
    uint32_t pack_width = 0, wide_pack_width = 0;

    PackSet narrow, wide, huge;   // huge: BDDs with a layer wider than wide_pack_width (frontier in global memory)
    uint32_t huge_pack_width = 0;  // largest hop of a huge pack (0: none)
    uint32_t narrow_slots = 0;        // slots [0, narrow_slots) belong to narrow packs
    uint64_t narrow_nodes = 0, diamond_nodes = 0;  // nodes of the narrow BDDs / of those among them that are chained into staggered packs (layout.cpp)
    std::vector<uint32_t> narrow_words;  // [narrow_slots]
    // Packs with the same structure (same BDD shapes at the same slots: every row of one constraint family) have
    // identical word sequences.  The device holds each distinct sequence once; pack p reads its words at
    // narrow_word_off[p] + (slot - first slot of the pack), so the copies stay L2-resident instead of streaming from HBM.
    std::vector<uint32_t> narrow_words_unique;  // concatenated distinct pack word sequences
    std::vector<uint32_t> narrow_word_off;      // [narrow packs] offset of the pack's sequence in narrow_words_unique
    std::vector<uint64_t> wide_words;    // [n_slots - narrow_slots]: wide packs, then huge packs

    // per layer (internal order: pack-major, hop-major, BDD order inside the pack)
    std::vector<int32_t> layer_var, layer_bdd;
    // per variable
    std::vector<int32_t> num_bdds_per_var;
    // CSR variable -> layers, sorted by (variable, bdd): var_ptr[V+1], var_layers[L]
    // (= primal_variable_sorting_order_, bdd_cuda_base.cu:379-391)
    std::vector<uint32_t> var_ptr, var_layers;
    // per BDD: internal layer index of its first layer is not contiguous; root slot:
    std::vector<uint32_t> bdd_root_slot;
    // per hop statistics (over all packs)
    std::vector<uint64_t> nodes_per_hop, layers_per_hop;
    Exchange ex;
    Resident res;
    // slot -> input instruction index (debug / round-trip tests), UINT64_MAX for padding
    std::vector<uint64_t> slot_to_instr;
};

// ---- checkpointing the layout ------------------------------------------------------------------
// The reference archives every index array of the device layout (bdd_cuda_base.cu:1486-1550) so that a solver can be restored
// without redoing the construction.  Same here: the scalars below + every array visit_layout_arrays names are what the device
// solver is built from (create_solver), so writing them and reading them back skips build_layout entirely.
struct LayoutScalars {
    uint64_t n_bdds, n_vars, n_hops, n_input_nodes, n_slots, n_nodes, n_layers;
    uint32_t pack_width, wide_pack_width, huge_pack_width, narrow_slots;
    uint32_t vars_per_bin, n_bins, stage_cap, waves_per_block, entry_by_var;
    uint32_t res_ok, res_max_slots, res_max_layers;
    uint32_t reserved[4];
};
inline LayoutScalars layout_scalars(const HostLayout& L)
{
    LayoutScalars s{};
    s.n_bdds = L.n_bdds; s.n_vars = L.n_vars; s.n_hops = L.n_hops; s.n_input_nodes = L.n_input_nodes; s.n_slots = L.n_slots;
    s.n_nodes = L.n_nodes; s.n_layers = L.n_layers;
    s.pack_width = L.pack_width; s.wide_pack_width = L.wide_pack_width; s.huge_pack_width = L.huge_pack_width; s.narrow_slots = L.narrow_slots;
    s.vars_per_bin = L.ex.vars_per_bin; s.n_bins = L.ex.n_bins; s.stage_cap = L.ex.stage_cap; s.waves_per_block = L.ex.waves_per_block;
    s.entry_by_var = L.ex.entry_by_var ? 1 : 0;
    s.res_ok = L.res.ok ? 1 : 0; s.res_max_slots = L.res.max_slots; s.res_max_layers = L.res.max_layers;
    return s;
}
inline void set_layout_scalars(HostLayout& L, const LayoutScalars& s)
{
    L.n_bdds = s.n_bdds; L.n_vars = s.n_vars; L.n_hops = s.n_hops; L.n_input_nodes = s.n_input_nodes; L.n_slots = s.n_slots;
    L.n_nodes = s.n_nodes; L.n_layers = s.n_layers;
    L.pack_width = s.pack_width; L.wide_pack_width = s.wide_pack_width; L.huge_pack_width = s.huge_pack_width; L.narrow_slots = s.narrow_slots;
    L.ex.vars_per_bin = s.vars_per_bin; L.ex.n_bins = s.n_bins; L.ex.stage_cap = s.stage_cap; L.ex.waves_per_block = s.waves_per_block;
    L.ex.entry_by_var = s.entry_by_var != 0;
    L.res.ok = s.res_ok != 0; L.res.max_slots = s.res_max_slots; L.res.max_layers = s.res_max_layers;
}
// every array create_solver consumes, with a stable id (the checkpoint format); v(id, vector&)
template <typename LAYOUT, typename V>
void visit_layout_arrays(LAYOUT& L, V&& v)
{
    v(1, L.narrow_words_unique); v(2, L.narrow_word_off); v(3, L.wide_words);
    v(4, L.layer_var); v(5, L.layer_bdd); v(6, L.num_bdds_per_var); v(7, L.var_ptr); v(8, L.var_layers); v(9, L.bdd_root_slot);
    v(10, L.narrow.pack_hop_ptr); v(11, L.narrow.hop_node_off); v(12, L.narrow.hop_layer_off); v(13, L.narrow.pack_steps);
    v(14, L.wide.pack_hop_ptr); v(15, L.wide.hop_node_off); v(16, L.wide.hop_layer_off); v(17, L.wide.pack_steps);
    v(18, L.huge.pack_hop_ptr); v(19, L.huge.hop_node_off); v(20, L.huge.hop_layer_off); v(21, L.huge.pack_steps);
    v(22, L.ex.evar); v(23, L.ex.bvar); v(24, L.ex.lpos); v(25, L.ex.vpos); v(26, L.ex.bin_ptr);
    v(27, L.ex.pack_group_ptr); v(28, L.ex.grp_layer_off); v(29, L.ex.grp_hop_end);
    v(30, L.ex.quad_round_ptr); v(31, L.ex.cs_ptr); v(32, L.ex.cs_entry); v(33, L.ex.cs_slot);
    v(34, L.res.pack_hdr); v(35, L.res.quad_hdr);
    v(36, L.nodes_per_hop); v(37, L.layers_per_hop);
    v(38, L.narrow.hop_root); v(39, L.wide.hop_root);
}
constexpr int LAYOUT_ARRAY_IDS = 40;

// Host threads of the layout build: 0 = automatic (BDDMMA_THREADS, else min(cores, 32)).  Processes that build several layouts at once
// (one per device slot: bddmma_host::solve_batch / bench_set_cover) share the cores through this.
void set_layout_threads(unsigned n);
// ... the same for the builds the calling thread starts (0 = follow the process-wide setting): what the batch farms use, so that two
// concurrent batches, or an application's own bddmma_set_layout_threads, are not overwritten
void set_thread_layout_threads(unsigned n);

// ---- records of the resident sweeps, second generation (kernels.hpp: k_fwd_res2 / k_bwd_res2) -------------------------------------
// The hop loop of the first resident sweeps spends ~165 instructions per 64 slots, ~25 of them floating point: the rest unpacks the
// 4-byte node word and turns hop-local indices into LDS addresses (profiles/r03_1m_latency.txt, profiles/r04_hop_isa.txt).  A record
// is everything a lane needs at one hop as ready-made 16-bit byte offsets into its wave's LDS region, one 16-byte load per lane and
// hop, coalesced, for EVERY lane of the hop (hops are dense here: 64 records per hop, padding lanes included):
//   .x  lo child: offset of its cost-from-terminal  |  hi child << 16        (sinks: the region's two constant entries)
//   .y  lo child: offset of its cost-from-root (push target)  |  hi << 16    (sinks and padding: the lane's own dummy entry)
//   .z  offset of the layer's {lo, hi} pair inside the cost / staging arrays  |  the node's own slot << 16 (same scale as the potentials)
//   .w  the pair offset again if the lane is its layer's head, else RES2_NO_STORE  |  flags << 16 (bit 0: two-node layer); all ones: padding lane
// Region of a wave (byte offsets, S = sizeof(REAL)): [T: (ns + 4) S | F: (ns + 64) S | {lo, hi}: nl 2 S]; ns / nl = the solver's
// resident slot / layer capacity.  Records depend on S and on ns, so they are derived data, built when a solver is created (not
// part of the checkpoint).  Packs that share their node words (structure templates) share their records.
constexpr uint32_t RES2_NO_STORE = 0xFFF0u;
constexpr uint32_t RES2_PAD = 0xFFFFFFFFu;
struct Res2Records {
    bool ok = false;                 // false: some offset does not fit 16 bits, packs are not 64 wide, ... -> first-generation kernels
    std::vector<uint32_t> rec;       // 4 words per record
    std::vector<uint32_t> rec_off;   // [narrow packs] first record of the pack (records, not words)
    uint32_t max_hops = 0;
};
__host__ __device__ inline uint32_t res2_t_off() { return 0; }
__host__ __device__ inline uint32_t res2_f_off(uint32_t real_size, uint32_t ns) { return (ns + 4u) * real_size; }
__host__ __device__ inline uint32_t res2_c_off(uint32_t real_size, uint32_t ns) { return (ns + 4u) * real_size + (ns + 64u) * real_size; }
__host__ __device__ inline uint32_t res2_wave_bytes(uint32_t real_size, uint32_t ns, uint32_t nl) { return res2_c_off(real_size, ns) + nl * 2u * real_size; }
// capacities of a wave's region: slots to a multiple of 1 KB of potentials, layers to a multiple of 512 B (the ONE place that rounds
// them: SolverT::init and the record builder's test entry bddmma_layout_res2_records use it, so the 16-bit offset guards are
// exercised at the capacities the GPU runs with)
inline uint32_t res2_slot_capacity(uint32_t real_size, uint32_t max_slots) { const uint32_t q = 1024u / real_size; return (max_slots + q - 1) / q * q; }
inline uint32_t res2_layer_capacity(uint32_t real_size, uint32_t max_layers) { const uint32_t q = 512u / real_size; return (max_layers + q - 1) / q * q; }
void build_res2_records(const HostLayout& L, uint32_t real_size, uint32_t ns, uint32_t nl, Res2Records& out);

// ---- records of the streaming sweeps, second generation (kernels.hpp: k_fwd_narrow2 / k_bwd_narrow2) --------------------------------
// The same idea for the streaming kernels, whose hop works out of per-hop LDS buffers (frontier, children's costs-from-terminal) instead of
// the whole pack: a lane's record at a hop holds byte offsets into THOSE buffers.  pack_width records per hop (dense), 16 bytes each:
//   .x  lo child: offset of its cost-from-terminal in the hop buffer (index * S; sinks: the two constant entries behind the slots)  |  hi << 16
//   .y  lo child: offset of its cost-from-root in the next frontier  |  hi << 16       (sinks and padding: the lane's own dummy entry, (W + 2 + j) S)
//   .z  2 S * (index of the node's layer among the layers of its hop)  |  the same if the lane is its layer's head, else RES2_NO_STORE, << 16
//   .w  bit 0: two-node layer; bits 8..13: position of the node inside its layer (the LDS segmented minimum of wider layers); bit 31: padding lane
// Packs of one structure template share their records.
constexpr uint32_t SREC_PAD = 0x80000000u;
struct StreamRecords {
    bool ok = false;
    std::vector<uint32_t> rec;      // 4 words per record
    std::vector<uint32_t> rec_off;  // [narrow packs] first record of the pack
};
void build_stream_records(const HostLayout& L, uint32_t real_size, StreamRecords& out);

// ---- records of the streaming sweeps, third generation (kernels/narrow3.hpp: k_fwd_narrow3 / k_bwd_narrow3; round 5) ---------------------
// A lane per LAYER instead of a lane per node.  Round 5's measurements (profiles/r05_hbm_only.txt) showed the streaming solve sweeps bound
// by what a hop costs in issue slots and in bytes through the CU's vector-memory path (5 KB per wave and hop, 2 KB of them records),
// not by HBM: with the hop's global loads AND stores removed a sweep still took 65 % of its time.  In packs whose layers have at most two
// nodes a lane that owns a whole layer (its one or two nodes sit in neighbouring slots, layout.cpp: PackBuilder::place) does every
// per-layer step once instead of once per node — arc costs, staged pair, the difference, the new costs, their store — needs no
// cross-lane minimum, reads and writes {lo, hi} and the staged values at lane * size (contiguous, no offsets from a record), and one
// 16-byte record per lane and hop instead of two:
//   .x  node a: lo child | hi child << 16     byte offsets into a hop buffer of W + 128 values: a child's hop-local slot index * S; the
//   .y  node b: lo child | hi child << 16     sinks are per-lane entries behind the slots, (W + 2 lane) S = TOP and (W + 2 lane + 1) S = BOT —
//                                             constants 0 / +inf in the costs-from-terminal buffers, dummy push targets in the frontier
//                                             buffers, so ONE offset serves the gather and the push (one-node layers / idle lanes: b -> BOT)
//   .z  slot of node a (hop-local) * S | flags << 16     flags bit 0: the layer has two nodes (b = a + 1), bit 1: the lane has a layer
//   .w  store offset of a's potential | b's << 16        slot * S inside the hop's slice, LREC_NO_STORE (past any slice: dropped) otherwise
// 64 records per hop (dense), shared by the packs of a structure template.  Usable where every hop of every narrow pack has <= 64 layers of
// <= 2 nodes, packs are 128 slots wide and none is staggered; everything else keeps the second / first generation.
constexpr uint32_t LREC_NO_STORE = 0xFFF0u;
constexpr uint32_t LREC_TWO = 1u, LREC_REAL = 2u;
struct LayerRecords {
    bool ok = false;
    std::vector<uint32_t> rec;      // 4 words per record
    std::vector<uint32_t> rec_off;  // [narrow packs] first record of the pack
};
void build_layer_records(const HostLayout& L, uint32_t real_size, LayerRecords& out);

// A per-layer vector given as a linear combination of stored ones (the L-BFGS direction, lbfgs.hip: q = g + sum_k cy[order[k]] Y[order[k]]
// + sum_k cs[order[k]] S[order[k]], accumulated in double in that order and rounded to REAL once): what the wrapper hands to
// SolverBase::projection_means_lincomb so that the direction is formed inside the first pass of its projection instead of being written
// and read back.  All pointers are device pointers; S (REAL) and Y (char) hold slots of `slot` elements (a multiple of 64, so every slot
// is 16-byte aligned and may be read up to the next multiple of 4 past its last element); g is readable up to the same bound.
struct LinComb {
    const char* g;
    const void* S;
    const char* Y;
    uint64_t slot;
    const double* cy;       // by physical slot
    const double* cs;
    const uint32_t* order;  // physical slots of the ns kept vectors, oldest first
    int ns;                 // <= LINCOMB_MAX
};
constexpr int LINCOMB_MAX = 8;

// ---- exchange without LDS atomics (kernels.hpp: k_exchange_seg; round 5) ---------------------------------------------------------------
// The binned exchange reduces a bin's deferred differences per variable with one LDS float atomic per entry — 7 of the launch's 19 us at
// 10.5 M nodes, at a rate (about one lane per clock) that neither conflict-free addresses nor integer adds change (profiles/
// r04_exchange_stamps.txt).  The structure is static, so the reduction can be a fixed schedule instead: the variables of a bin are dealt to
// the workgroup's threads (largest first, boustrophedon, so every thread gets the same number of entries to within one or two), a thread's
// entries — those of its variables, one variable after the other, (variable, bdd) order inside — are its RUN, and all it needs is where
// they are: `perm`, the entry offsets (inside the bin) of the positions of its run.  The kernel copies the bin's differences to LDS with
// coalesced loads, every thread walks its run with plain LDS reads and sums each variable in the order of the deterministic path
// (k_delta_gather), writes the pair to the variable's SLOT (slots are numbered along the runs), and writes the slot number back to its
// entries' LDS places; the broadcast then streams entry -> slot -> pair.  No atomics, no dependent global loads, fixed summation order.
//   bin  [4 per bin]                first 16-byte group of the bin in `perm`, groups (of 8 positions) per thread | slots << 8, first entry, entries
//   perm [8 per (bin, group, thread)]  u16 entry offsets, thread-minor: group g of thread t is the 16 bytes at (first + g * threads + t);
//                                   past the end of a run: the bin's entry count rounded up to 16 bytes of REAL (an LDS place that holds 0)
//   thr  [2 per (bin, thread)]      bit k set: position k of the run is the last entry of its variable; the thread's first slot
struct SegExchange {
    bool ok = false;
    uint32_t threads = 0;
    uint32_t max_entries = 0, max_slots = 0, max_groups = 0;  // largest bin
    std::vector<uint32_t> bin;
    std::vector<uint16_t> perm;
    std::vector<uint32_t> thr;
};
constexpr uint32_t SEG_MAX_RUN = 32;  // positions per thread: bits of the end mask
void build_seg_exchange(const HostLayout& L, uint32_t threads, uint32_t real_size, SegExchange& out);

// What the automatic layout rules and the kernel selection need to know about the chip (bddmma_create reads it from hipDeviceProp:
// query_chip, solver_base.hip; the defaults are MI355X and are what the CPU-side layout entry points use).
struct ChipInfo {
    uint32_t n_cus = 256;
    uint32_t lds_bytes = 160 * 1024;  // per CU
    uint64_t max_resident_threads = 256ull * 2048;   // multiProcessorCount * maxThreadsPerMultiProcessor (the input stage's split-length rule)
};

// Returns BDDMMA_OK or an error code; `err` receives the message.
int build_layout(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds,
                 const bddmma_options* opts, HostLayout& out, std::string& err, bool keep_debug_maps,
                 uint32_t real_size = 4, ChipInfo chip = ChipInfo{});

}  // namespace bddmma
