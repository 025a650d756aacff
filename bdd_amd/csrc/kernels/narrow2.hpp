// kernels/narrow2.hpp — narrow packs, streaming solve sweeps, second generation (k_fwd_narrow2 / k_bwd_narrow2): per-lane records.
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// narrow packs, streaming sweeps, second generation: the solve sweeps on per-lane records
// =============================================================================================
// k_fwd_narrow / k_bwd_narrow spend 136 instructions per hop and 64 slots, 72 of them VALU of which 14 are floating point
// (profiles/r04_hop_isa.txt): on the headline instance the vector ALUs are busy half of the sweep's duration with address arithmetic.
// These kernels are the same sweeps — same pipeline (records two hops ahead, arc costs and potentials one), same staging rounds, same
// arithmetic in the same order — with the 4-byte node word replaced by a 16-byte record of ready-made byte offsets into the hop's LDS
// buffers (layout.hpp: StreamRecords): no unpacking, no sink / padding selects (constant entries; a private dummy entry per lane behind
// the frontier), the layer index inside the hop instead of a ballot count per lane group, the head-only store as an offset past the
// hop's layers, the two-node minimum as one DPP swap.  Records of a structure template are shared by its packs (L2 hits).
// Solve sweeps of packs whose layers have <= 2 nodes and that are not staggered (SolverT::use_narrow2); everything else: first generation.
__device__ __forceinline__ void hop_store(float2 v, rsrc_t rh, uint32_t voff, uint32_t soff);   // defined with the exchange kernels below
__device__ __forceinline__ void hop_store(double2 v, rsrc_t rh, uint32_t voff, uint32_t soff);
template <int R>
__device__ __forceinline__ void load_recs(u4v (&r)[R], rsrc_t rr, uint32_t first_rec, int lane)
{
#pragma unroll
    for (int g = 0; g < R; ++g) r[g] = __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)(lane + 64 * g) * 16u, first_rec * 16u, 0);
}
// {lo, hi} of the lanes' layers: the layer's offset inside the hop is the record's, the hop's first layer goes into the scalar offset
template <typename REAL, int R>
__device__ __forceinline__ void load_costs(typename Pair<REAL>::type (&c)[R], const u4v (&r)[R], rsrc_t lohi, uint32_t lbase)
{
    using P2 = typename Pair<REAL>::type;
#pragma unroll
    for (int g = 0; g < R; ++g) hop_load(c[g], lohi, r[g][2] & 0xFFFFu, lbase * (uint32_t)sizeof(P2));
}

template <typename REAL, int R, int WPB, bool GEN, int LA = BDDMMA_LOOKAHEAD, bool NT = false>
__device__ __forceinline__ void fwd_narrow2_body(const DevPtrs<REAL>& d, const PackDev& pk, const uint32_t* __restrict__ srec,
                                                 const uint32_t* __restrict__ srec_off, uint32_t srec_words, REAL omega, uint32_t block_id,
                                                 const uint32_t* __restrict__ hdr_pack = nullptr, const uint32_t* __restrict__ hdr_quad = nullptr)
{
    constexpr int W = 64 * R;
    constexpr uint32_t S = sizeof(REAL);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    __shared__ __attribute__((aligned(16))) REAL sF_[WPB][2][2 * W + 2];  // frontier of the current / next hop; [W], [W + 1] unused, [W + 2 + j]: lane slot j's dummy push target
    __shared__ __attribute__((aligned(16))) REAL sT_[WPB][2][W + 2];      // costs-from-terminal of the next hop (written one hop ahead); [W] = 0 (top), [W + 1] = +inf (bot)
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    unsigned char* sFw = reinterpret_cast<unsigned char*>(&sF_[wave][0][0]);
    unsigned char* sTw = reinterpret_cast<unsigned char*>(&sT_[wave][0][0]);
    constexpr uint32_t F_STRIDE = (2 * W + 2) * S, T_STRIDE = (W + 2) * S;
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    // Resident headers (layout.hpp: struct Resident; given when every pack has one stage group and every quad one round): the pack's hop /
    // slot / layer ranges and the quad's range of the staging tables in ONE round trip — without them q0 -> {first slot, first layer} and
    // quad -> round -> item range are two dependent round trips each, and all workgroups of a launch walk those chains at the same time
    // (the first ~10 us of a sweep move little: profiles/r04_sweep_timeline.txt).
    const bool hdr = hdr_pack != nullptr;  // uniform
    const uint32_t* const hp = hdr ? hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t rbase = has_pack ? srec_off[p] : 0;
    const uint32_t c0_h = hdr ? hdr_quad[4 * (size_t)quad] : 0, cnt_h = hdr ? hdr_quad[4 * (size_t)quad + 1] : 0;
    BDDMMA_STAMP(p, 0);
    // GEN: packs with layers wider than two nodes (LDS segmented minimum, seg_min2: per-wave scratch behind the rest of the dynamic LDS) and
    // staggered packs (a BDD root below the pack's first hop, PackDev::hop_root)
    const int steps = GEN ? (has_pack ? pk.pack_steps[p] : 0) : 1;
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);  // the pack's first slot / layer: everything below is relative to them (HopWindow)
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);  // on their way while the pipeline is set up
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    (void)lohi_p;
    const rsrc_t rr = make_rsrc(srec, srec_words);
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    auto off = [&](uint32_t q) { return hw.node_off(q); };
    constexpr int D = LA;
    uint32_t o[2 * D + 3];
    uint32_t lb[D + 2];  // first layer of hops q .. q + D + 1
    u4v rc[2 * D + 1][R];
    REAL tr[D + 1][R];
    P2 Lr[D + 1][R];
#pragma unroll
    for (int i = 0; i < 2 * D + 3; ++i) o[i] = 0;
#pragma unroll
    for (int i = 0; i < D + 2; ++i) lb[i] = 0;
    if (has_pack) {
        hw.fill(pk, q0, lane);
#pragma unroll
        for (int i = 0; i < 2 * D + 3; ++i) o[i] = off(q0 + i);
#pragma unroll
        for (int i = 0; i < D + 2; ++i) lb[i] = hw.layer_off(q0 + i);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            lds_st<REAL>(sFw, j * S, (j < o[1] - o[0]) ? REAL(0) : INF);  // every slot of hop 0 is a root (flush_costs_from_root)
        }
        if (lane < 4) lds_st<REAL>(sTw, (uint32_t)(lane >> 1) * T_STRIDE + (W + (lane & 1)) * S, (lane & 1) ? INF : REAL(0));
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) load_recs<R>(rc[i], rr, rbase + (uint32_t)i * W, lane);  // (past the last hop: some other records, never used)
        {
            REAL t1[R];
            load_vals_p<REAL, R, NT>(t1, Tp, o[1], o[2] - o[1], lane);  // T of hop q0+1: straight into LDS
#pragma unroll
            for (int i = 0; i < D; ++i) load_vals_p<REAL, R, NT>(tr[i], Tp, o[i + 2], o[i + 3] - o[i + 2], lane);  // T of hop q0+2+i
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                if (j < o[2] - o[1]) lds_st<REAL>(sTw, j * S, t1[r]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) load_costs<REAL, R>(Lr[i], rc[i], rs.lohi, lb[i]);
        wave_sync();
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) rc[i][r] = u4v{0u, 0u, 0u, SREC_PAD};
    }
    uint32_t cur = 0;
    uint32_t q = q0;
    uint32_t rt = NO_ROOT;  // GEN: root slot of hop q when a BDD starts there; the first hop's roots are set up above
    const uint32_t g0 = (has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = !has_pack ? 0 : (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0);
    const uint32_t r0 = hdr ? 0 : pk.quad_round_ptr[quad];
    const uint32_t n_rounds = hdr ? 1u : pk.quad_round_ptr[quad + 1] - r0;
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);  // this wave's slots of the staging area
    for (uint32_t k = 0; k < n_rounds; ++k) {
        uint32_t gl0 = 0, cnt = 0, qe = q1;
        {
            if (hdr) {
                cnt = cnt_h;
                stage_load_pairs<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
            } else {
                const uint32_t c0 = pk.cs_ptr[r0 + k];
                cnt = pk.cs_ptr[r0 + k + 1] - c0;
                stage_load<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(sD, ent, esl, rs, c0, cnt, tid);  // the delta pairs of the quad's k-th groups -> LDS
            }
            if (hdr) {
                qe = has_pack ? q1 : q;  // one group: the whole pack
            } else if (k < ng) {
                gl0 = pk.grp_layer_off[g0 + k] - l0;
                qe = pk.grp_hop_end[g0 + k];
            } else {
                qe = q;  // this pack has no k-th group: no hops in this round
            }
            if (WPB > 1) __syncthreads(); else wave_sync();
            BDDMMA_STAMP(p, 1);
        }
        auto hop = [&]() {
            if (q + 2 * D + 3 >= hw.base + HOP_WIN && hw.base + HOP_WIN <= q1) hw.fill(pk, q, lane);
            const uint32_t nb = o[0];
            const uint32_t n3 = o[3] - o[2];  // slots of hop q+2
            const uint32_t fc = cur * F_STRIDE, fn = (cur ^ 1u) * F_STRIDE, tc = cur * T_STRIDE, tn = (cur ^ 1u) * T_STRIDE;
            const uint32_t stg = db + (lb[0] - gl0) * (uint32_t)sizeof(P2);  // the hop's first layer inside the wave's staging slots
            // ---- global prefetch: records of hop q+2D, T of hop q+D+2, arc costs of hop q+D
            load_recs<R>(rc[2 * D], rr, rbase + (q - q0 + 2 * D) * W, lane);
#ifndef BDDMMA_EXP_NO_HOP_LOADS  // timing experiments only (wrong results): the hop loop without its streams from / to global memory
            load_vals_p<REAL, R, NT>(tr[D], Tp, o[D + 2], o[D + 3] - o[D + 2], lane);
            load_costs<REAL, R>(Lr[D], rc[D], rs.lohi, lb[D]);
#endif
            u4v (&ra)[R] = rc[0];
            P2 (&La)[R] = Lr[0];
            // ---- the hop's LDS reads, one batch
            REAL f[R], tl[R], th[R];
            P2 dd[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                f[r] = lds_ld<REAL>(sFw, fc + j * S);
                if (GEN && j == rt) f[r] = REAL(0);  // a BDD that starts at this hop: its root has no parents (flush_costs_from_root)
                tl[r] = lds_ld<REAL>(sTw, tc + (ra[r][0] & 0xFFFFu));  // sinks: [W] = 0, [W+1] = +inf; padding lanes: +inf
                th[r] = lds_ld<REAL>(sTw, tc + (ra[r][0] >> 16));
                dd[r] = lds_ld<P2>(dyn_lds, stg + (ra[r][2] & 0xFFFFu));
            }
            const uint32_t o_new = off(q + 2 * D + 3);
            const uint32_t l_next = hw.layer_off(q + D + 2);
            const uint32_t rt_next = GEN ? hw.root_of(q + 1) : (uint32_t)NO_ROOT;
            // ---- set-up of the next hop's buffers (nothing above depends on it)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                if (j < n3) lds_st<REAL>(sTw, tn + j * S, tr[0][r]);  // T of hop q+2, gathered by hop q+1
                lds_st<REAL>(sFw, fn + j * S, INF);
            }
            wave_sync();
            // ---- arithmetic
            P2 nc[R];
            REAL mmv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const REAL lc = La[r].x, hc = La[r].y;
                REAL m0 = (f[r] + lc) + tl[r], m1 = (f[r] + hc) + th[r];
                if (!GEN || steps <= 1) pair_min_aligned(m0, m1, (ra[r][3] & 1u) != 0);
                else seg_min2(m0, m1, lane, (ra[r][3] >> 8) & 63u, 0u, steps, sM);
                const REAL mm = mm_diff1(m0, m1, omega);
                mmv[r] = mm;
                nc[r].x = (lc + min0(mm)) + dd[r].x;
                nc[r].y = (hc + min0_neg(mm)) + dd[r].y;
            }
            // ---- writes: new arc costs (heads), staged min-marginal differences, pushes into the next frontier, costs-from-root
            const rsrc_t rl = hop_rsrc(reinterpret_cast<const P2*>(lohi_p), lb[0], lb[1] - lb[0]);  // ends with the hop's layers: RES2_NO_STORE is dropped
#pragma unroll
            for (int r = 0; r < R; ++r) {
#ifndef BDDMMA_EXP_NO_HOP_STORES
                hop_store(nc[r], rl, ra[r][2] >> 16, lb[0] * (uint32_t)sizeof(P2));
#endif
                if (!(ra[r][3] & SREC_PAD)) lds_st<REAL>(dyn_lds, stg + (ra[r][2] & 0xFFFFu), mmv[r]);  // every lane of a layer holds the same value
                lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[r][1] & 0xFFFFu)), f[r] + nc[r].x);  // sinks / padding: the lane's own dummy entry
                lds_min(reinterpret_cast<REAL*>(sFw + fn + (ra[r][1] >> 16)), f[r] + nc[r].y);
            }
#ifndef BDDMMA_EXP_NO_HOP_STORES
            store_vals<R>(f, Fp, nb, o[1] - o[0], lane, pk.nt_potentials);
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
            rt = rt_next;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) rc[i][r] = rc[i + 1][r];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    Lr[i][r] = Lr[i + 1][r];
                    tr[i][r] = tr[i + 1][r];
                }
            ++q;
        };
        while (q + HOP_UNROLL <= qe) {
#pragma unroll
            for (int u = 0; u < HOP_UNROLL; ++u) hop();
        }
        while (q < qe) hop();
        {
            BDDMMA_STAMP(p, 3);
            if (WPB > 1) __syncthreads(); else wave_sync();
            BDDMMA_STAMP(p, 2);
            stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);  // min-marginal differences of the round -> entry array
            BDDMMA_STAMP(p, 4);
            if (WPB > 1) __syncthreads();                        // the next round overwrites the staging area
        }
    }
}

// register budget: the records' ring costs 18 VGPRs more than the node words' (109 / 131 instead of 92 / 123 in the forward sweep, float / double,
// R = 2); asking for 5 / 4 waves per SIMD makes the allocator stay at 96 / 128 without spilling
#ifndef BDDMMA_N2_WAVES
#define BDDMMA_N2_WAVES(REAL, R) ((R) <= 2 ? (sizeof(REAL) == 4 ? 5 : 4) : 1)
#endif
template <typename REAL, int R, int WPB, bool GEN, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N2_WAVES(REAL, R)))) k_fwd_narrow2(DevPtrs<REAL> d, PackDev pk, const uint32_t* __restrict__ srec, const uint32_t* __restrict__ srec_off,
                                                          uint32_t srec_words, REAL omega)
{
    fwd_narrow2_body<REAL, R, WPB, GEN, BDDMMA_LOOKAHEAD, NT>(d, pk, srec, srec_off, srec_words, omega, blockIdx.x, pk.hdr_pack, pk.hdr_quad);
}

template <typename REAL, int R, int WPB, bool GEN, int LA = BDDMMA_LOOKAHEAD, bool NT = false>
__device__ __forceinline__ void bwd_narrow2_body(const DevPtrs<REAL>& d, const PackDev& pk, const uint32_t* __restrict__ srec,
                                                 const uint32_t* __restrict__ srec_off, uint32_t srec_words, REAL omega, uint32_t block_id,
                                                 const uint32_t* __restrict__ hdr_pack = nullptr, const uint32_t* __restrict__ hdr_quad = nullptr)
{
    constexpr int W = 64 * R;
    constexpr uint32_t S = sizeof(REAL);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    __shared__ __attribute__((aligned(16))) REAL sT_[WPB][2][W + 2];  // per wave; +2: sink entries TOP = W (0) and BOT = W + 1 (+inf)
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    unsigned char* sTw = reinterpret_cast<unsigned char*>(&sT_[wave][0][0]);
    constexpr uint32_t T_STRIDE = (W + 2) * S;
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    const bool hdr = hdr_pack != nullptr;  // uniform: resident headers, see fwd_narrow2_body
    const uint32_t* const hp = hdr ? hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t rbase = has_pack ? srec_off[p] : 0;
    const uint32_t c0_h = hdr ? hdr_quad[4 * (size_t)quad] : 0, cnt_h = hdr ? hdr_quad[4 * (size_t)quad + 1] : 0;
    BDDMMA_STAMP(p, 0);
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);  // the pack's first slot / layer: everything below is relative to them (HopWindow)
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);  // on their way while the pipeline is set up
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    (void)lohi_p;
    REAL* const x_p = d.x_layer != nullptr ? d.x_layer + l0 : nullptr;  // DevPtrs::x_layer, from the pack's first layer on
    const rsrc_t rr = make_rsrc(srec, srec_words);
    const int steps = GEN ? (has_pack ? pk.pack_steps[p] : 0) : 1;  // see k_fwd_narrow2
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    double lb_stag = 0.0;  // GEN: costs-to-terminal of the roots below the pack's first hop (staggered packs), for the lower bound
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    auto nb_of = [&](uint32_t q) { return hw.node_off(q); };
    // pipeline mirrored from k_fwd_narrow2 (see k_bwd_narrow): before hop q is processed (q counts down) the wave holds the records of hops
    // q .. q-2D+1, the costs-from-root of hops q .. q-D and the arc costs of hops q .. q-D+1.  o[i] = first slot of hop q+1-i,
    // lb[i] = first layer of hop q+1-i (hops below q0: those of q0).
    constexpr int D = LA;
    uint32_t o[2 * D + 2];
    uint32_t lb[D + 2];
    u4v rc[2 * D + 1][R];
    REAL fr[D + 2][R];
    P2 Lr[D + 1][R];
#pragma unroll
    for (int i = 0; i < 2 * D + 2; ++i) o[i] = 0;
#pragma unroll
    for (int i = 0; i < D + 2; ++i) lb[i] = 0;
    uint32_t q = q1;
    // record of hop h of the pack (h may run below 0 at the pipeline's end: any record, never used)
    auto rec_of = [&](uint32_t qq) { return rbase + (qq >= q0 ? qq - q0 : 0u) * W; };
    if (has_pack) {
        if (lane < 4) lds_st<REAL>(sTw, (uint32_t)(lane >> 1) * T_STRIDE + (W + (lane & 1)) * S, (lane & 1) ? INF : REAL(0));
        hw.fill(pk, q1 + 1 > q0 + HOP_WIN ? q1 + 1 - HOP_WIN : q0, lane);  // window ends at record q1
#pragma unroll
        for (int i = 0; i < 2 * D + 2; ++i) o[i] = nb_of(q1 >= q0 + i ? q1 - i : q0);
#pragma unroll
        for (int i = 0; i < D + 2; ++i) lb[i] = hw.layer_off(q1 >= q0 + i ? q1 - i : q0);
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) load_recs<R>(rc[i], rr, rec_of(q1 >= q0 + i + 1 ? q1 - 1 - i : q0), lane);  // hop q1-1-i
#pragma unroll
        for (int i = 0; i < D + 1; ++i) load_vals_p<REAL, R, NT>(fr[i], Fp, o[i + 1], o[i] - o[i + 1], lane);            // F of hop q1-1-i
#pragma unroll
        for (int i = 0; i < D; ++i) load_costs<REAL, R>(Lr[i], rc[i], rs.lohi, lb[i + 1]);                          // hop q1-1-i starts at layer lb[i+1]
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) rc[i][r] = u4v{0u, 0u, 0u, SREC_PAD};
    }
    uint32_t cur = 0;
    const uint32_t g0 = (has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = !has_pack ? 0 : (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0);
    const uint32_t r0 = hdr ? 0 : pk.quad_round_ptr[quad];
    const uint32_t n_rounds = hdr ? 1u : pk.quad_round_ptr[quad + 1] - r0;
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);
    for (uint32_t k = n_rounds; k-- > 0;) {  // same rounds as the forward sweep, in reverse
        uint32_t gl0 = 0, cnt = 0, qs = q0;
        {
            if (hdr) {
                cnt = cnt_h;
                stage_load_pairs<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
            } else {
                const uint32_t c0 = pk.cs_ptr[r0 + k];
                cnt = pk.cs_ptr[r0 + k + 1] - c0;
                stage_load<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(sD, ent, esl, rs, c0, cnt, tid);
            }
            if (hdr) {
                qs = has_pack ? q0 : q;  // one group: the whole pack
            } else if (k < ng) {
                gl0 = pk.grp_layer_off[g0 + k] - l0;
                qs = (k == 0) ? q0 : pk.grp_hop_end[g0 + k - 1];
            } else {
                qs = q;  // no k-th group in this pack
            }
            if (WPB > 1) __syncthreads(); else wave_sync();
            BDDMMA_STAMP(p, 1);
        }
        auto hop = [&]() {
            --q;
            if (q < hw.base + 2 * D + 1 && hw.base > q0) hw.fill(pk, q + 1 > q0 + HOP_WIN ? q + 1 - HOP_WIN : q0, lane);
            const uint32_t nb = o[1];
            const uint32_t tc = cur * T_STRIDE, tn = (cur ^ 1u) * T_STRIDE;
            const uint32_t stg = db + (lb[1] - gl0) * (uint32_t)sizeof(P2);  // hop q starts at layer lb[1]
            // ---- prefetch: records of hop q-2D, F of hop q-D-1, arc costs of hop q-D
            load_recs<R>(rc[2 * D], rr, rec_of(q >= q0 + 2 * D ? q - 2 * D : q0), lane);
            load_vals_p<REAL, R, NT>(fr[D + 1], Fp, o[D + 2], o[D + 1] - o[D + 2], lane);
            load_costs<REAL, R>(Lr[D], rc[D], rs.lohi, lb[D + 1]);
            u4v (&ra)[R] = rc[0];
            REAL (&fa)[R] = fr[0];
            P2 (&La)[R] = Lr[0];
            // ---- LDS reads
            REAL tl[R], th[R];
            P2 dd[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tl[r] = lds_ld<REAL>(sTw, tc + (ra[r][0] & 0xFFFFu));  // sinks: [W] = 0, [W+1] = +inf
                th[r] = lds_ld<REAL>(sTw, tc + (ra[r][0] >> 16));
                dd[r] = lds_ld<P2>(dyn_lds, stg + (ra[r][2] & 0xFFFFu));
            }
            const uint32_t o_new = (q >= q0 + 2 * D + 1) ? nb_of(q - 1 - 2 * D) : o[2 * D + 1];
            const uint32_t l_next = hw.layer_off(q >= q0 + D + 1 ? q - 1 - D : q0);
            const uint32_t rt = (GEN && q > q0) ? hw.root_of(q) : (uint32_t)NO_ROOT;  // the first hop's roots are summed behind the loop
            // ---- arithmetic
            REAL t[R], mmv[R];
            P2 nc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const REAL lc = La[r].x, hc = La[r].y;
                REAL m0 = (fa[r] + lc) + tl[r], m1 = (fa[r] + hc) + th[r];
                if (!GEN || steps <= 1) pair_min_aligned(m0, m1, (ra[r][3] & 1u) != 0);
                else seg_min2(m0, m1, lane, (ra[r][3] >> 8) & 63u, 0u, steps, sM);
                const REAL mm = mm_diff1(m0, m1, omega);
                mmv[r] = mm;
                nc[r].x = (lc + min0(mm)) + dd[r].x;
                nc[r].y = (hc + min0_neg(mm)) + dd[r].y;
                t[r] = rmin(nc[r].y + th[r], nc[r].x + tl[r]);
            }
            // ---- writes
            const rsrc_t rl = hop_rsrc(reinterpret_cast<const P2*>(lohi_p), lb[1], lb[0] - lb[1]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                hop_store(nc[r], rl, ra[r][2] >> 16, lb[1] * (uint32_t)sizeof(P2));
                if (!(ra[r][3] & SREC_PAD)) {
                    lds_st<REAL>(dyn_lds, stg + (ra[r][2] & 0xFFFFu), mmv[r]);
                    lds_st<REAL>(sTw, tn + j * S, t[r]);
                }
                if (GEN && j == rt) lb_stag += (double)t[r];
            }
            if (x_p != nullptr) {  // uniform: net_solver_costs x = (hi' - lo') + mm in layer order for an L-BFGS wrapper, heads only, straight from the hop
                // (as a pass over the staging area behind the round it cost 13 us of a 41 us sweep at 10.5 M nodes, tools/xlayer_cost.py)
                const rsrc_t rx = hop_rsrc(x_p, lb[1], lb[0] - lb[1]);  // ends with the hop's layers: half of RES2_NO_STORE is dropped as well
#pragma unroll
                for (int r = 0; r < R; ++r) hop_store((nc[r].y - nc[r].x) + mmv[r], rx, (ra[r][2] >> 16) >> 1, lb[1] * S);
            }
            store_vals<R>(t, Tp, nb, o[0] - o[1], lane, pk.nt_potentials);
            wave_sync();
            cur ^= 1u;
#pragma unroll
            for (int i = 0; i < 2 * D + 1; ++i) o[i] = o[i + 1];
            o[2 * D + 1] = o_new;
#pragma unroll
            for (int i = 0; i < D + 1; ++i) lb[i] = lb[i + 1];
            lb[D + 1] = l_next;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) rc[i][r] = rc[i + 1][r];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) Lr[i][r] = Lr[i + 1][r];
#pragma unroll
            for (int i = 0; i < D + 1; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) fr[i][r] = fr[i + 1][r];
        };
        while (q >= qs + HOP_UNROLL) {
#pragma unroll
            for (int u = 0; u < HOP_UNROLL; ++u) hop();
        }
        while (q > qs) hop();
        {
            BDDMMA_STAMP(p, 3);
            if (WPB > 1) __syncthreads(); else wave_sync();
            BDDMMA_STAMP(p, 2);
            stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
            BDDMMA_STAMP(p, 4);
            if (WPB > 1) __syncthreads();
        }
    }
    if (!has_pack) return;
    // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251)
    const uint32_t n0 = nb_of(q0 + 1) - nb_of(q0);
    double s = lb_stag;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        if (j < n0) s += (double)lds_ld<REAL>(sTw, cur * T_STRIDE + j * S);
    }
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = s;
}

template <typename REAL, int R, int WPB, bool GEN, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N2_WAVES(REAL, R)))) k_bwd_narrow2(DevPtrs<REAL> d, PackDev pk, const uint32_t* __restrict__ srec, const uint32_t* __restrict__ srec_off,
                                                          uint32_t srec_words, REAL omega)
{
    bwd_narrow2_body<REAL, R, WPB, GEN, BDDMMA_LOOKAHEAD, NT>(d, pk, srec, srec_off, srec_words, omega, blockIdx.x, pk.hdr_pack, pk.hdr_quad);
}

}  // namespace bddmma
