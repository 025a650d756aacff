// kernels/narrow3.hpp — narrow packs, streaming solve sweeps, third generation (k_fwd_narrow3 / k_bwd_narrow3): a lane per LAYER.
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// narrow packs, streaming sweeps, third generation: one lane owns one layer (round 5)
// =============================================================================================
// What round 5 measured on the second generation (profiles/r05_hbm_only.txt): with the hop's global loads AND stores compiled out a sweep
// still takes 65 % of its time, the hop loop's duration follows the instructions a SIMD has to issue for the waves it holds, and the
// loads and stores of a hop ADD to that time (5 KB per wave and hop through the CU's 64 B/clk vector-memory path, 2 KB of them records)
// instead of hiding behind it.  So a hop has to cost fewer issue slots and fewer bytes through that path — HBM is not the limit.
// In packs of <= 2-node layers a lane that owns a whole LAYER (layout.hpp: LayerRecords) does every per-layer step once instead of
// once per node: one record (1 KB per wave and hop instead of 2), {lo, hi} and the staged pair read at lane * size (no offset from the
// record, 512 B instead of 1 KB of {lo, hi} per hop), one difference, one pair of new costs, one store; the layer's minimum is a plain
// minimum of the lane's two candidates (no DPP, no wait states); the two potentials of the layer are one ds_read2.  Same arithmetic in
// the same order as k_fwd_narrow2 / k_bwd_narrow2 (SURVEY.md §8 a'), bit-equal results.
// Hop buffers in LDS: [0, W) the hop's slots, behind them per lane l the entries W + 2 l (TOP) and W + 2 l + 1 (BOT): constants 0 / +inf
// in the costs-from-terminal buffers, dummy push targets in the frontier buffers.
// Packs of 128 slots with <= 64 layers of <= 2 nodes per hop, none staggered; everything else: second / first generation
// (SolverT::use_narrow3).  Stage groups, staging rounds and the start from the resident headers as in k_fwd_narrow2.
constexpr int N3_W = 128;
template <typename REAL>
__device__ __forceinline__ void lds_ld2(REAL& a, REAL& b, const unsigned char* lds, uint32_t off)  // two neighbouring values (ds_read2_b32 / ds_read2_b64: 4- / 8-byte alignment)
{
    const REAL* p = reinterpret_cast<const REAL*>(lds + off);
    a = p[0];
    b = p[1];
}

// Two neighbouring potentials with ONE load (round 6): the slots 2 lane, 2 lane + 1 of a hop (the forward sweep's copy of the next hop's
// costs-from-terminal into LDS) or the nodes a, a + 1 of the lane's layer (the backward sweep's costs-from-root; LayerRecords: b = a + 1).  A hop
// of these sweeps costs what its vector-memory instructions cost (a 4-byte-per-layer store added to the backward hop's seven instructions cost
// 15 % of the sweep, profiles/r06_lbfgs_experiments.txt), and two 4-byte accesses of neighbouring addresses are one 8-byte access.  What lies
// past the hop's last slot / behind a one-node layer comes back as whatever is there (or 0 past the descriptor's end) and is never used.
template <typename REAL, bool NT>
__device__ __forceinline__ void pot_pair_load(typename Pair<REAL>::type& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    constexpr int AUX = NT ? 2 : BDDMMA_LD_POT_AUX;
    if constexpr (sizeof(REAL) == 4) v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rh, voff, soff, AUX));
    else v = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rh, voff, soff, AUX));
}

// store of one potential (see hop_store: double-precision instances far beyond the Infinity Cache store F / T non-temporally)
__device__ __forceinline__ void pot_store(float v, rsrc_t rh, uint32_t voff, uint32_t soff, uint32_t /*nt*/) { hop_store(v, rh, voff, soff); }
__device__ __forceinline__ void pot_store(double v, rsrc_t rh, uint32_t voff, uint32_t soff, uint32_t nt)
{
    if (nt) hop_store<2>(v, rh, voff, soff);  // uniform (a kernel argument)
    else hop_store<>(v, rh, voff, soff);
}

// The layer's one or two potentials (nodes a, b = a + 1) with as little store work as its shape allows: two-node layers write {a, b} with one
// 8- / 16-byte store, one-node layers a with a 4- / 8-byte one — two instructions with complementary offsets (the other lanes' lie past the
// slice and are dropped); in a uniform family a hop's layers are all of one kind, so one of the two has nothing to do.  (Merging the two stores of
// every lane is not possible: a one-node layer's second value would land in the next layer's slot.)
template <typename REAL>
__device__ __forceinline__ void pot_store_layer(REAL a, REAL b, bool two, rsrc_t rh, uint32_t off_a, uint32_t soff, uint32_t nt)
{
    using P2 = typename Pair<REAL>::type;
    const uint32_t off2 = two ? off_a : (uint32_t)LREC_NO_STORE, off1 = two ? (uint32_t)LREC_NO_STORE : off_a;
    P2 v;
    v.x = a;
    v.y = b;
    if constexpr (sizeof(REAL) == 4) {
        using u2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(rh, 0, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rh, off2, soff, BDDMMA_ST_FT_AUX_F32);
        pot_store(a, rh, off1, soff, nt);
    } else {
        using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(rh, 0, 0, 0));
        const u4 data = __builtin_bit_cast(u4, v);
        if (nt) __builtin_amdgcn_raw_buffer_store_b128(data, rh, off2, soff, 2);  // uniform (a kernel argument)
        else __builtin_amdgcn_raw_buffer_store_b128(data, rh, off2, soff, BDDMMA_ST_AUX);
        asm volatile("s_nop 0" ::"v"(data) : "memory");  // wide store with an SGPR offset: the write-data hazard the compiler does not pad (hop_store(double2))
        pot_store(a, rh, off1, soff, nt);
    }
}

// Hops the prefetches of the hop pipeline run ahead.  One for float (two and three: nothing, round 5).  Two for double: +0.8 % in a same-box A/B
// over 36 solver objects (tools/exp_r06_r.sh: 4 612 / 4 620 / 4 482 against 4 576 / 4 589 / 4 481 it/s at 10.5 M nodes, process by process — the
// process-to-process spread of 3 % is larger than the effect; a first single-sample reading of +4-6 % was that spread), three: -1 %, four: spills.
#ifndef BDDMMA_N3_LOOKAHEAD
#define BDDMMA_N3_LOOKAHEAD(REAL) (sizeof(REAL) == 8 ? 2 : BDDMMA_LOOKAHEAD)
#endif
template <typename REAL, int WPB, bool NT = false, int LA = BDDMMA_LOOKAHEAD>
__device__ __forceinline__ void fwd_narrow3_body(const DevPtrs<REAL>& d, const PackDev& pk, const uint32_t* __restrict__ lrec,
                                                 const uint32_t* __restrict__ lrec_off, uint32_t lrec_words, REAL omega, uint32_t block_id)
{
    constexpr int W = N3_W;
    constexpr uint32_t S = sizeof(REAL);
    constexpr uint32_t BUF = (W + 128) * S;  // one hop buffer
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    __shared__ __attribute__((aligned(16))) unsigned char sF_[WPB][2][BUF];  // frontier of the current / next hop
    __shared__ __attribute__((aligned(16))) unsigned char sT_[WPB][2][BUF];  // costs-from-terminal of the next hop (written one hop ahead)
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    unsigned char* sFw = &sF_[wave][0][0];
    unsigned char* sTw = &sT_[wave][0][0];
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    // resident headers (given when every pack has one stage group and every quad one round, see fwd_narrow2_body): the pack's hop / slot /
    // layer ranges and the quad's range of the staging tables in one round trip
    const bool hdr = pk.hdr_pack != nullptr;  // uniform
    const uint32_t* const hp = hdr ? pk.hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t rbase = has_pack ? lrec_off[p] : 0;
    const uint32_t c0_h = hdr ? pk.hdr_quad[4 * (size_t)quad] : 0, cnt_h = hdr ? pk.hdr_quad[4 * (size_t)quad + 1] : 0;
    BDDMMA_STAMP(p, 0);
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);  // on their way while the pipeline is set up
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    const rsrc_t rr = make_rsrc(lrec, lrec_words);
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    constexpr int D = LA;
    uint32_t o[2 * D + 3];  // first slot of hops q .. q + 2D + 2
    uint32_t lb[D + 2];     // first layer of hops q .. q + D + 1
    u4v rc[2 * D + 1];      // records of hops q .. q + 2D
    P2 tr[D + 1];           // costs-from-terminal of hops q + 2 .. q + D + 2, slots (2 lane, 2 lane + 1): copied to LDS one hop before they are read
    P2 Lr[D + 1];           // {lo, hi} of the lane's layer in hops q .. q + D
#pragma unroll
    for (int i = 0; i < 2 * D + 3; ++i) o[i] = 0;
#pragma unroll
    for (int i = 0; i < D + 2; ++i) lb[i] = 0;
    auto ldrec = [&](uint32_t h) { return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + h * 64u) * 16u, 0); };
    const uint32_t sink = (W + 2 * (uint32_t)lane) * S;  // this lane's TOP entry; BOT follows it
    if (has_pack) {
        hw.fill(pk, q0, lane);
#pragma unroll
        for (int i = 0; i < 2 * D + 3; ++i) o[i] = hw.node_off(q0 + i);
#pragma unroll
        for (int i = 0; i < D + 2; ++i) lb[i] = hw.layer_off(q0 + i);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t j = lane + 64 * r;
            lds_st<REAL>(sFw, j * S, (j < o[1] - o[0]) ? REAL(0) : INF);  // every slot of hop 0 is a root (flush_costs_from_root)
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {  // the sink constants of both costs-from-terminal buffers
            lds_st<REAL>(sTw, b * BUF + sink, REAL(0));
            lds_st<REAL>(sTw, b * BUF + sink + S, INF);
        }
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = ldrec((uint32_t)i);  // (past the last hop: some other records, never used)
        {
            P2 t1;
            pot_pair_load<REAL, NT>(t1, hop_rsrc(Tp, o[1], o[2] - o[1]), 2u * (uint32_t)lane * S, o[1] * S);  // T of hop q0+1: straight into LDS
#pragma unroll
            for (int i = 0; i < D; ++i) pot_pair_load<REAL, NT>(tr[i], hop_rsrc(Tp, o[i + 2], o[i + 3] - o[i + 2]), 2u * (uint32_t)lane * S, o[i + 2] * S);  // T of hop q0+2+i
            if (2u * (uint32_t)lane < o[2] - o[1]) lds_st<P2>(sTw, 2u * (uint32_t)lane * S, t1);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) hop_load(Lr[i], rs.lohi, (uint32_t)lane * (uint32_t)sizeof(P2), lb[i] * (uint32_t)sizeof(P2));
        wave_sync();
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = u4v{0u, 0u, 0u, 0u};
    }
    uint32_t cur = 0;
    uint32_t q = q0;
    const uint32_t g0 = (has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = !has_pack ? 0 : (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0);
    const uint32_t r0 = hdr ? 0 : pk.quad_round_ptr[quad];
    const uint32_t n_rounds = hdr ? 1u : pk.quad_round_ptr[quad + 1] - r0;
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);  // this wave's slots of the staging area
    for (uint32_t k = 0; k < n_rounds; ++k) {
    uint32_t gl0 = 0, cnt = 0, qe = q1;
    if (hdr) {
        cnt = cnt_h;
        stage_load_pairs<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
        qe = has_pack ? q1 : q;  // one group: the whole pack
    } else {
        const uint32_t c0 = pk.cs_ptr[r0 + k];
        cnt = pk.cs_ptr[r0 + k + 1] - c0;
        stage_load<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(sD, ent, esl, rs, c0, cnt, tid);  // the delta pairs of the quad's k-th groups -> LDS
        if (k < ng) {
            gl0 = pk.grp_layer_off[g0 + k] - l0;
            qe = pk.grp_hop_end[g0 + k];
        } else {
            qe = q;  // this pack has no k-th group: no hops in this round
        }
    }
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 1);
    auto hop = [&]() {
        if (q + 2 * D + 3 >= hw.base + HOP_WIN && hw.base + HOP_WIN <= q1) hw.fill(pk, q, lane);
        const uint32_t nb = o[0];
        const uint32_t n3 = o[3] - o[2];  // slots of hop q+2
        const uint32_t fc = cur * BUF, fn = (cur ^ 1u) * BUF;
        const uint32_t stg = db + (lb[0] - gl0 + (uint32_t)lane) * (uint32_t)sizeof(P2);  // the lane's layer inside the wave's staging slots
        // ---- global prefetch: record of hop q+2D, T of hop q+D+2, arc costs of hop q+D
        rc[2 * D] = ldrec(q - q0 + 2 * D);
#ifndef BDDMMA_EXP_NO_HOP_LOADS
        pot_pair_load<REAL, NT>(tr[D], hop_rsrc(Tp, o[D + 2], o[D + 3] - o[D + 2]), 2u * (uint32_t)lane * S, o[D + 2] * S);
        hop_load(Lr[D], rs.lohi, (uint32_t)lane * (uint32_t)sizeof(P2), lb[D] * (uint32_t)sizeof(P2));
#endif
        const u4v ra = rc[0];
        const P2 c = Lr[0];
        // ---- the hop's LDS reads, one batch
        const uint32_t flags = ra[2] >> 16;
        REAL fa, fb;
        lds_ld2(fa, fb, sFw, fc + (ra[2] & 0xFFFFu));
        const REAL tla = lds_ld<REAL>(sTw, fc + (ra[0] & 0xFFFFu)), tha = lds_ld<REAL>(sTw, fc + (ra[0] >> 16));
        const REAL tlb = lds_ld<REAL>(sTw, fc + (ra[1] & 0xFFFFu)), thb = lds_ld<REAL>(sTw, fc + (ra[1] >> 16));
        const P2 dd = lds_ld<P2>(dyn_lds, stg);
        const uint32_t o_new = hw.node_off(q + 2 * D + 3);
        const uint32_t l_next = hw.layer_off(q + D + 2);
        // ---- set-up of the next hop's buffers (nothing above depends on it)
        if (2u * (uint32_t)lane < n3) lds_st<P2>(sTw, fn + 2u * (uint32_t)lane * S, tr[0]);  // T of hop q+2, gathered by hop q+1
        lds_st<P2>(sFw, fn + 2u * (uint32_t)lane * S, P2{INF, INF});
        wave_sync();
        // ---- arithmetic: the layer's two candidates per side, their minimum, the deferred difference, the new arc costs
        fa = (flags & LREC_REAL) ? fa : INF;
        fb = (flags & LREC_TWO) ? fb : INF;
        const REAL m0 = rmin((fa + c.x) + tla, (fb + c.x) + tlb);
        const REAL m1 = rmin((fa + c.y) + tha, (fb + c.y) + thb);
        const REAL mm = mm_diff1(m0, m1, omega);
        P2 nc;
        nc.x = (c.x + min0(mm)) + dd.x;
        nc.y = (c.y + min0_neg(mm)) + dd.y;
        // ---- writes: new arc costs, staged difference, pushes into the next frontier, costs-from-root
#ifndef BDDMMA_EXP_NO_HOP_STORES
        const rsrc_t rl = hop_rsrc(reinterpret_cast<const P2*>(lohi_p), lb[0], lb[1] - lb[0]);  // ends with the hop's layers: idle lanes are dropped
        hop_store(nc, rl, (uint32_t)lane * (uint32_t)sizeof(P2), lb[0] * (uint32_t)sizeof(P2));
#endif
        if (flags & LREC_REAL) lds_st<REAL>(dyn_lds, stg, mm);
        lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[0] & 0xFFFFu)), fa + nc.x);  // sinks / idle lanes: the lane's own dummy entries
        lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[0] >> 16)), fa + nc.y);
        lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[1] & 0xFFFFu)), fb + nc.x);
        lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[1] >> 16)), fb + nc.y);
#ifndef BDDMMA_EXP_NO_HOP_STORES
        {
            const rsrc_t rf = hop_rsrc(Fp, nb, o[1] - o[0]);  // LREC_NO_STORE lies past the slice
            pot_store_layer<REAL>(fa, fb, (flags & LREC_TWO) != 0, rf, ra[3] & 0xFFFFu, nb * S, pk.nt_potentials);
        }
#endif
        wave_sync();
        cur ^= 1u;
        // ---- rotate the pipeline registers
#pragma unroll
        for (int i = 0; i < 2 * D + 2; ++i) o[i] = o[i + 1];
        o[2 * D + 2] = o_new;
#pragma unroll
        for (int i = 0; i < D + 1; ++i) lb[i] = lb[i + 1];
        lb[D + 1] = l_next;
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = rc[i + 1];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            Lr[i] = Lr[i + 1];
            tr[i] = tr[i + 1];
        }
        ++q;
    };
    while (q + HOP_UNROLL <= qe) {
#pragma unroll
        for (int u = 0; u < HOP_UNROLL; ++u) hop();
    }
    while (q < qe) hop();
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);  // min-marginal differences of the round -> entry array
    BDDMMA_STAMP(p, 4);
    if (WPB > 1 && k + 1 < n_rounds) __syncthreads();    // the next round overwrites the staging area
    }
}

#ifndef BDDMMA_N3_WAVES
#define BDDMMA_N3_WAVES(REAL) (sizeof(REAL) == 4 ? 5 : 4)
#endif
template <typename REAL, int WPB, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N3_WAVES(REAL)))) k_fwd_narrow3(DevPtrs<REAL> d, PackDev pk, const uint32_t* __restrict__ lrec, const uint32_t* __restrict__ lrec_off,
                                                          uint32_t lrec_words, REAL omega)
{
    fwd_narrow3_body<REAL, WPB, NT, BDDMMA_N3_LOOKAHEAD(REAL)>(d, pk, lrec, lrec_off, lrec_words, omega, blockIdx.x);
}

template <typename REAL, int WPB, bool NT = false, int LA = BDDMMA_LOOKAHEAD>
__device__ __forceinline__ void bwd_narrow3_body(const DevPtrs<REAL>& d, const PackDev& pk, const uint32_t* __restrict__ lrec,
                                                 const uint32_t* __restrict__ lrec_off, uint32_t lrec_words, REAL omega, uint32_t block_id)
{
    constexpr int W = N3_W;
    constexpr uint32_t S = sizeof(REAL);
    constexpr uint32_t BUF = (W + 128) * S;
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    __shared__ __attribute__((aligned(16))) unsigned char sT_[WPB][2][BUF];  // costs-from-terminal of the hop below (read) / of this hop (written)
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    unsigned char* sTw = &sT_[wave][0][0];
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    const bool hdr = pk.hdr_pack != nullptr;  // uniform: resident headers, see fwd_narrow3_body
    const uint32_t* const hp = hdr ? pk.hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t rbase = has_pack ? lrec_off[p] : 0;
    const uint32_t c0_h = hdr ? pk.hdr_quad[4 * (size_t)quad] : 0, cnt_h = hdr ? pk.hdr_quad[4 * (size_t)quad + 1] : 0;
    BDDMMA_STAMP(p, 0);
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    REAL* const x_p = d.x_layer != nullptr ? d.x_layer + l0 : nullptr;
    const rsrc_t rr = make_rsrc(lrec, lrec_words);
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    auto nb_of = [&](uint32_t qq) { return hw.node_off(qq); };
    // before hop q is processed (q counts down) the wave holds the records of hops q .. q-2D, the costs-from-root of the layers of hops
    // q .. q-D and their arc costs.  o[i] = first slot of hop q+1-i, lb[i] = first layer of hop q+1-i (hops below q0: those of q0).
    constexpr int D = LA;
    uint32_t o[2 * D + 2];
    uint32_t lb[D + 2];
    u4v rc[2 * D + 1];
    P2 fr[D + 1];       // costs-from-root of the lane's layer (nodes a, b = a + 1) in hops q .. q-D
    P2 Lr[D + 1];
#pragma unroll
    for (int i = 0; i < 2 * D + 2; ++i) o[i] = 0;
#pragma unroll
    for (int i = 0; i < D + 2; ++i) lb[i] = 0;
    uint32_t q = q1;
    // record of hop h of the pack (below the first hop: any record, never used)
    auto ldrec = [&](uint32_t qq) { return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + (qq >= q0 ? qq - q0 : 0u) * 64u) * 16u, 0); };
    // costs-from-root of the two nodes of the lane's layer: slice of the hop [nb, nb + n), the record's store offsets (idle lanes / no second node: past the slice -> 0)
    auto ldf = [&](P2& f, const u4v& r, uint32_t nb, uint32_t n) {   // node a's store offset; idle lanes: LREC_NO_STORE, past the slice -> 0
        pot_pair_load<REAL, NT>(f, hop_rsrc(Fp, nb, n), r[3] & 0xFFFFu, nb * S);
    };
    const uint32_t sink = (W + 2 * (uint32_t)lane) * S;
    if (has_pack) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            lds_st<REAL>(sTw, b * BUF + sink, REAL(0));
            lds_st<REAL>(sTw, b * BUF + sink + S, INF);
        }
        hw.fill(pk, q1 + 1 > q0 + HOP_WIN ? q1 + 1 - HOP_WIN : q0, lane);  // window ends at record q1
#pragma unroll
        for (int i = 0; i < 2 * D + 2; ++i) o[i] = nb_of(q1 >= q0 + i ? q1 - i : q0);
#pragma unroll
        for (int i = 0; i < D + 2; ++i) lb[i] = hw.layer_off(q1 >= q0 + i ? q1 - i : q0);
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = ldrec(q1 >= q0 + i + 1 ? q1 - 1 - i : q0);  // hop q1-1-i
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ldf(fr[i], rc[i], o[i + 1], o[i] - o[i + 1]);                                                                 // hop q1-1-i
            hop_load(Lr[i], rs.lohi, (uint32_t)lane * (uint32_t)sizeof(P2), lb[i + 1] * (uint32_t)sizeof(P2));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = u4v{0u, 0u, 0u, 0u};
    }
    uint32_t cur = 0;
    const uint32_t g0 = (has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = !has_pack ? 0 : (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0);
    const uint32_t r0 = hdr ? 0 : pk.quad_round_ptr[quad];
    const uint32_t n_rounds = hdr ? 1u : pk.quad_round_ptr[quad + 1] - r0;
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);
    for (uint32_t k = n_rounds; k-- > 0;) {  // same rounds as the forward sweep, in reverse
    uint32_t gl0 = 0, cnt = 0, qs = q0;
    if (hdr) {
        cnt = cnt_h;
        stage_load_pairs<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
        qs = has_pack ? q0 : q;  // one group: the whole pack
    } else {
        const uint32_t c0 = pk.cs_ptr[r0 + k];
        cnt = pk.cs_ptr[r0 + k + 1] - c0;
        stage_load<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(sD, ent, esl, rs, c0, cnt, tid);
        if (k < ng) {
            gl0 = pk.grp_layer_off[g0 + k] - l0;
            qs = (k == 0) ? q0 : pk.grp_hop_end[g0 + k - 1];
        } else {
            qs = q;  // no k-th group in this pack
        }
    }
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 1);
    auto hop = [&]() {
        --q;
        if (q < hw.base + 2 * D + 1 && hw.base > q0) hw.fill(pk, q + 1 > q0 + HOP_WIN ? q + 1 - HOP_WIN : q0, lane);
        const uint32_t nb = o[1];
        const uint32_t tc = cur * BUF, tn = (cur ^ 1u) * BUF;
        const uint32_t stg = db + (lb[1] - gl0 + (uint32_t)lane) * (uint32_t)sizeof(P2);  // hop q starts at layer lb[1]
        // ---- prefetch: record of hop q-2D, costs-from-root and arc costs of hop q-D
        rc[2 * D] = ldrec(q >= q0 + 2 * D ? q - 2 * D : q0);
#ifndef BDDMMA_EXP_NO_HOP_LOADS
        ldf(fr[D], rc[D], o[D + 1], o[D] - o[D + 1]);
        hop_load(Lr[D], rs.lohi, (uint32_t)lane * (uint32_t)sizeof(P2), lb[D + 1] * (uint32_t)sizeof(P2));
#endif
        const u4v ra = rc[0];
        const P2 c = Lr[0];
        const uint32_t flags = ra[2] >> 16;
        // ---- LDS reads
        const REAL tla = lds_ld<REAL>(sTw, tc + (ra[0] & 0xFFFFu)), tha = lds_ld<REAL>(sTw, tc + (ra[0] >> 16));
        const REAL tlb = lds_ld<REAL>(sTw, tc + (ra[1] & 0xFFFFu)), thb = lds_ld<REAL>(sTw, tc + (ra[1] >> 16));
        const P2 dd = lds_ld<P2>(dyn_lds, stg);
        const uint32_t o_new = (q >= q0 + 2 * D + 1) ? nb_of(q - 1 - 2 * D) : o[2 * D + 1];
        const uint32_t l_next = hw.layer_off(q >= q0 + D + 1 ? q - 1 - D : q0);
        // ---- arithmetic
        const REAL fa = (flags & LREC_REAL) ? fr[0].x : INF;
        const REAL fb = (flags & LREC_TWO) ? fr[0].y : INF;
        const REAL m0 = rmin((fa + c.x) + tla, (fb + c.x) + tlb);
        const REAL m1 = rmin((fa + c.y) + tha, (fb + c.y) + thb);
        const REAL mm = mm_diff1(m0, m1, omega);
        P2 nc;
        nc.x = (c.x + min0(mm)) + dd.x;
        nc.y = (c.y + min0_neg(mm)) + dd.y;
        const REAL ta = rmin(nc.y + tha, nc.x + tla);
        const REAL tb = rmin(nc.y + thb, nc.x + tlb);
        // ---- writes
#ifndef BDDMMA_EXP_NO_HOP_STORES
        const rsrc_t rl = hop_rsrc(reinterpret_cast<const P2*>(lohi_p), lb[1], lb[0] - lb[1]);
        hop_store(nc, rl, (uint32_t)lane * (uint32_t)sizeof(P2), lb[1] * (uint32_t)sizeof(P2));
#endif
        if (flags & LREC_REAL) {
            lds_st<REAL>(dyn_lds, stg, mm);
            lds_st<REAL>(sTw, tn + (ra[2] & 0xFFFFu), ta);
        }
        if (flags & LREC_TWO) lds_st<REAL>(sTw, tn + (ra[2] & 0xFFFFu) + S, tb);
        if (x_p != nullptr) {  // uniform: net_solver_costs x = (hi' - lo') + mm in layer order for an L-BFGS wrapper
            const rsrc_t rx = hop_rsrc(x_p, lb[1], lb[0] - lb[1]);
            hop_store((nc.y - nc.x) + mm, rx, (uint32_t)lane * S, lb[1] * S);
        }
#ifndef BDDMMA_EXP_NO_HOP_STORES
        {
            const rsrc_t rt = hop_rsrc(Tp, nb, o[0] - o[1]);
            pot_store_layer<REAL>(ta, tb, (flags & LREC_TWO) != 0, rt, ra[3] & 0xFFFFu, nb * S, pk.nt_potentials);
        }
#endif
        wave_sync();
        cur ^= 1u;
#pragma unroll
        for (int i = 0; i < 2 * D + 1; ++i) o[i] = o[i + 1];
        o[2 * D + 1] = o_new;
#pragma unroll
        for (int i = 0; i < D + 1; ++i) lb[i] = lb[i + 1];
        lb[D + 1] = l_next;
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) rc[i] = rc[i + 1];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            Lr[i] = Lr[i + 1];
            fr[i] = fr[i + 1];
        }
    };
    while (q >= qs + HOP_UNROLL) {
#pragma unroll
        for (int u = 0; u < HOP_UNROLL; ++u) hop();
    }
    while (q > qs) hop();
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
    BDDMMA_STAMP(p, 4);
    if (WPB > 1 && k > 0) __syncthreads();
    }
    if (!has_pack) return;
    // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251); every slot of the first hop is a root
    const uint32_t n0 = nb_of(q0 + 1) - nb_of(q0);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t j = lane + 64 * r;
        if (j < n0) s += (double)lds_ld<REAL>(sTw, cur * BUF + j * S);
    }
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = s;
}

template <typename REAL, int WPB, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N3_WAVES(REAL)))) k_bwd_narrow3(DevPtrs<REAL> d, PackDev pk, const uint32_t* __restrict__ lrec, const uint32_t* __restrict__ lrec_off,
                                                          uint32_t lrec_words, REAL omega)
{
    bwd_narrow3_body<REAL, WPB, NT, BDDMMA_N3_LOOKAHEAD(REAL)>(d, pk, lrec, lrec_off, lrec_words, omega, blockIdx.x);
}

}  // namespace bddmma
