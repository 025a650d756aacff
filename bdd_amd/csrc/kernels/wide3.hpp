// kernels/wide3.hpp — wide packs, solve sweeps with every prefetch two hops ahead of its first use (fwd_wide3_body / bwd_wide3_body).
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// wide packs, solve sweeps, third form: a two-hop distance for every dependent load
// =============================================================================================
// k_fwd_wide2 / k_bwd_wide2 ask for a hop's data in three dependent levels — node word -> entry index (lpos[layer of the word]) ->
// delta pair — one hop apart each, and rotate their registers at the end of every hop: a rotation reads what the hop itself requested,
// so every level has ONE hop to arrive, and a hop of ~0.3 us of work takes a memory round trip (~1.1 us per hop and workgroup on the
// 25 000-row instance of 18-variable knapsack rows: 0.23 of the roofline, profiles/r05_wide_skip.txt).  Here
//   * the entry index is a per-SLOT array (DevPtrs::wide_ent, built once per solver: lpos[layer of the slot's word]) — same bytes, same
//     request count as the per-node gather of lpos it replaces, but not behind the word: two levels instead of three;
//   * the loop runs in TRIPS of two hops.  Every load of a trip is issued at its start: {word, entry} of the two hops two trips ahead,
//     {arc costs, delta pair, potential} of the two hops of the next trip (their addresses need the words / entries requested a trip
//     earlier), the hop offsets likewise.  Nothing is consumed before the next trip, so whatever wait the compiler puts at the loop's back
//     edge (it drains everything there: the registers change roles) finds loads that are two hops old.  (First attempt: per-hop requests
//     into two alternating landing sets — the back edge then waited for loads issued one hop earlier, and offsets the compiler knew to be
//     uniform were moved to SGPRs right behind their load, a round trip per hop: 259 / 224 us per sweep instead of wide2's 136 / 185.)
// Same arithmetic, same order, same LDS protocol (frontier / potentials / per-layer minima, two barriers per hop) as the wide2 solve
// sweeps: bit-equal results.  Solve modes only; the plain / marginal / solution sweeps stay wide2.
template <typename REAL, int NPT>
struct WideHop {       // what a thread holds of one hop: its nodes' words and entry indices, their layers' arc costs and delta pairs, and a potential
    using P2 = typename Pair<REAL>::type;
    uint64_t W[NPT];
    uint32_t E[NPT];
    P2 C[NPT], D[NPT];
    REAL V[NPT];       // backward: cost from root of the hop's nodes; forward: cost to terminal of the nodes TWO hops on (LDS-bound, see fwd)
};
template <typename REAL, int NPT>
__device__ __forceinline__ void wide_hop_clear(WideHop<REAL, NPT>& h)
{
    using P2 = typename Pair<REAL>::type;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        h.W[i] = WW_PAD_WORD;
        h.E[i] = 0;
        h.C[i] = h.D[i] = P2{};
        h.V[i] = REAL(0);
    }
}

template <typename REAL, int NPT>
__device__ __forceinline__ void bwd_wide3_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t ww, uint32_t p)
{
    using P2 = typename Pair<REAL>::type;
    using Hop = WideHop<REAL, NPT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[16];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    const uint32_t S = ww + 2;
    REAL* const lds = reinterpret_cast<REAL*>(smem);  // integer offsets instead of rotating pointers, see k_fwd_wide2
    const uint32_t oM0 = 4 * S, oM1 = 6 * S;
    const REAL INF = inf_v<REAL>();
    const WideRs<REAL> rs(d);
    const rsrc_t rent = make_rsrc(d.wide_ent, (uint64_t)d.n_slots - d.wide_slot_base);
    const int64_t q0 = (int64_t)pk.pack_hop_ptr[p], q1 = (int64_t)pk.pack_hop_ptr[p + 1];
    const uint32_t wsb = d.wide_slot_base;
    // Hop offsets, clamped to the pack: hop h has the nodes [rn(h), rn(h + 1)) and the layers [rl(h), rl(h + 1)) — empty outside [q0, q1).
    // Requested a trip ahead like everything else.  (The index carries a zero the compiler cannot see through: a load it KNOWS to be uniform
    // is moved to an SGPR by a v_readfirstlane right behind the load — a wait for a load just issued.  As "per-lane" values the offsets stay
    // in VGPRs until they are used and are made scalar there.)
    uint32_t vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto clampq = [&](int64_t h) { return (uint32_t)(h < q0 ? q0 : (h > q1 ? q1 : h)); };
    auto rn = [&](int64_t h) { return pk.hop_node_off[clampq(h) + vz]; };
    auto rl = [&](int64_t h) { return pk.hop_layer_off[clampq(h) + vz]; };
    auto root_at = [&](int64_t h) -> uint32_t { return (h > q0 && h < q1) ? (uint32_t)pk.hop_root[(uint32_t)h + vz] : (uint32_t)NO_ROOT; };
    for (uint32_t j = tid; j < S; j += T) { lds[oM0 + j] = INF; lds[oM0 + S + j] = INF; lds[oM1 + j] = INF; lds[oM1 + S + j] = INF; }
    if (tid < 4) lds[(tid >> 1) * S + ww + (tid & 1)] = (tid & 1) ? INF : REAL(0);
    __syncthreads();
    uint32_t tc = 0, cur = 0;  // T buffer tc: hop q+1 (children), tc^1: hop q (being written)
    double lb_stag = 0.0;
    // one hop of the sweep on data that has arrived; n / nb / lb: the hop's nodes, first slot, first layer; nl1: layers of the hop processed next
    auto compute = [&](const Hop& X, uint32_t n, uint32_t nb, uint32_t lb, uint32_t nl1, uint32_t rt0) {
        const uint32_t oTc = tc * S, oTn = (tc ^ 1) * S;
        const uint32_t oMa = oM0 + cur * S, oMb = oM1 + cur * S, oMa_n = oM0 + (cur ^ 1) * S, oMb_n = oM1 + (cur ^ 1) * S;
        // ---- phase A: per-layer minima of the two min-marginals
        REAL tl[NPT], th[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            tl[i] = lds[oTc + ww_lo(X.W[i], ww)];
            th[i] = lds[oTc + ww_hi(X.W[i], ww)];
            const uint32_t l = ww_layer(X.W[i]);
            REAL a = (X.V[i] + X.C[i].x) + tl[i], b = (X.V[i] + X.C[i].y) + th[i];
            const bool lead = seg_fold_by_key(a, b, act ? l : 0xFFFFFFFFu, (int)(tid & 63u));
            if (act && lead) {
                lds_min(&lds[oMa + l], a);
                lds_min(&lds[oMb + l], b);
            }
        }
        __syncthreads();
        // ---- phase B
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            const uint64_t w = X.W[i];
            const uint32_t l = ww_layer(w);
            const bool head = act && (w & WW_HEAD);
            const REAL m0 = lds[oMa + l], m1 = lds[oMb + l];
            const REAL mm = mm_diff(m0, m1, omega);
            const REAL nlo = (X.C[i].x + min0(mm)) + X.D[i].x;
            const REAL nhi = (X.C[i].y + min0_neg(mm)) + X.D[i].y;
            const REAL t = rmin(nhi + th[i], nlo + tl[i]);
            P2 nc;
            nc.x = nlo;
            nc.y = nhi;
            bstore(nc, rs.lohi, head ? (lb + l) * (uint32_t)sizeof(P2) : OOB);
            bstore(mm, rs.mm, head ? X.E[i] * (uint32_t)sizeof(REAL) : OOB);
            if (d.x_layer != nullptr && head) d.x_layer[lb + l] = (nhi - nlo) + mm;
            if (act) lds[oTn + j] = t;
            if (j == rt0) lb_stag += (double)t;
            bstore(t, rs.T, act ? (nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        for (uint32_t l = tid; l < nl1; l += T) { lds[oMa_n + l] = INF; lds[oMb_n + l] = INF; }
        __syncthreads();
        tc ^= 1;
        cur ^= 1;
    };
    // requests of one hop: {arc costs, delta pair, cost from root} through the hop's words / entries (arrived), or the {words, entries} themselves
    auto request_cdv = [&](Hop& H, uint32_t c, uint32_t nb, uint32_t lb) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            bload(H.C[i], rs.lohi, j < c ? (lb + ww_layer(H.W[i])) * (uint32_t)sizeof(P2) : OOB);
            bload(H.D[i], rs.dlay, j < c ? H.E[i] * (uint32_t)sizeof(P2) : OOB);
            bload(H.V[i], rs.F, j < c ? (nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
    };
    auto request_we = [&](Hop& G, uint32_t c, uint32_t nb) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            const uint64_t x = bload_u64(rs.words, j < c ? (nb - wsb + j) * 8u : OOB);
            G.W[i] = j < c ? x : WW_PAD_WORD;
            G.E[i] = bload_u32(rent, j < c ? (nb - wsb + j) * 4u : OOB);
        }
    };
    // Register sets: R0 / R1 the trip's two hops (q, q-1), complete; H0 / H1 the next trip's (q-2, q-3): words and entries arrived, the rest
    // requested at this trip's start; G0 / G1 the words and entries of the trip after (q-4, q-5), requested at this trip's start.
    Hop R0, R1, H0, H1, G0, G1;
    wide_hop_clear(R0); wide_hop_clear(R1); wide_hop_clear(H0); wide_hop_clear(H1); wide_hop_clear(G0); wide_hop_clear(G1);
    // offsets of the hops q+1 .. q-5 / q+1 .. q-3 (RN[j] = rn(q + 1 - j), RL[j] = rl(q + 1 - j)), roots of q, q-1; two more of each per trip
    int64_t q = q1 + 5;
    uint32_t RN[7], RL[5], RT[2];
#pragma unroll
    for (int j = 0; j < 7; ++j) RN[j] = rn(q + 1 - j);
#pragma unroll
    for (int j = 0; j < 5; ++j) RL[j] = rl(q + 1 - j);
    RT[0] = RT[1] = NO_ROOT;
    for (; q >= q0; q -= 2) {   // three priming trips (requests only: hops >= q1), then the pack's hops
        // ---- the trip's requests (consumed from the next trip on)
        const uint32_t rn_a = rn(q - 6), rn_b = rn(q - 7), rl_a = rl(q - 4), rl_b = rl(q - 5), rt_a = root_at(q - 2), rt_b = root_at(q - 3);
        request_cdv(H0, uni(RN[2] - RN[3]), uni(RN[3]), uni(RL[3]));   // hop q-2
        request_cdv(H1, uni(RN[3] - RN[4]), uni(RN[4]), uni(RL[4]));   // hop q-3
        request_we(G0, uni(RN[4] - RN[5]), uni(RN[5]));                // hop q-4
        request_we(G1, uni(RN[5] - RN[6]), uni(RN[6]));                // hop q-5
        // ---- the trip's two hops
        if (q < q1) compute(R0, uni(RN[0] - RN[1]), uni(RN[1]), uni(RL[1]), uni(RL[1] - RL[2]), uni(RT[0]));
        if (q - 1 < q1 && q - 1 >= q0) compute(R1, uni(RN[1] - RN[2]), uni(RN[2]), uni(RL[2]), uni(RL[2] - RL[3]), uni(RT[1]));
        // ---- everything requested at the trip's start has had two hops to arrive: the sets move up
        R0 = H0; R1 = H1;
#pragma unroll
        for (int i = 0; i < NPT; ++i) { H0.W[i] = G0.W[i]; H0.E[i] = G0.E[i]; H1.W[i] = G1.W[i]; H1.E[i] = G1.E[i]; }
#pragma unroll
        for (int j = 0; j < 5; ++j) RN[j] = RN[j + 2];
        RN[5] = rn_a; RN[6] = rn_b;
#pragma unroll
        for (int j = 0; j < 3; ++j) RL[j] = RL[j + 2];
        RL[3] = rl_a; RL[4] = rl_b;
        RT[0] = rt_a; RT[1] = rt_b;
    }
    // lower bound contribution of this pack (bdd_cuda_base.cu:1243-1251)
    const uint32_t n0 = pk.hop_node_off[q0 + 1] - pk.hop_node_off[q0];
    double acc = lb_stag;
    for (uint32_t j = tid; j < n0; j += T) acc += (double)lds[tc * S + j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < T / 64; ++i) t += red[i];
        d.lb_partial[pk.lb_base + p] = t;
    }
}

template <typename REAL, int NPT>
__device__ __forceinline__ void fwd_wide3_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t ww, uint32_t p)
{
    using P2 = typename Pair<REAL>::type;
    using Hop = WideHop<REAL, NPT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    const uint32_t S = ww + 2;
    REAL* const lds = reinterpret_cast<REAL*>(smem);           // 8 arrays of S values: F x3, T x1 (+1 unused), minima x4
    const uint32_t oT0 = 3 * S;
    const uint32_t oM0 = 4 * S, oM1 = 6 * S;  // minima of lo / hi: [oM0 + cur * S + l], [oM1 + cur * S + l]
    const REAL INF = inf_v<REAL>();
    const WideRs<REAL> rs(d);
    const rsrc_t rent = make_rsrc(d.wide_ent, (uint64_t)d.n_slots - d.wide_slot_base);
    const int64_t q0 = (int64_t)pk.pack_hop_ptr[p], q1 = (int64_t)pk.pack_hop_ptr[p + 1];
    const uint32_t wsb = d.wide_slot_base;
    uint32_t vz;   // an opaque zero: the offset loads stay "per-lane" values until they are used (see bwd_wide3_body)
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto clampq = [&](int64_t h) { return (uint32_t)(h < q0 ? q0 : (h > q1 ? q1 : h)); };
    auto rn = [&](int64_t h) { return pk.hop_node_off[clampq(h) + vz]; };
    auto rl = [&](int64_t h) { return pk.hop_layer_off[clampq(h) + vz]; };
    auto root_at = [&](int64_t h) -> uint32_t { return (h > q0 && h < q1) ? (uint32_t)pk.hop_root[(uint32_t)h + vz] : (uint32_t)NO_ROOT; };
    // LDS: roots of hop q0, empty next frontiers, empty minima; the costs to terminal of hop q0 + 1 arrive with the priming hop q0 - 1
    {
        const uint32_t n_first = uni(rn(q0 + 1) - rn(q0));
        for (uint32_t j = tid; j < S; j += T) {
            lds[j] = j < n_first ? REAL(0) : INF;  // every node of hop q0 is a root (flush_costs_from_root)
            lds[S + j] = INF;
            lds[2 * S + j] = INF;
            lds[oM0 + j] = INF; lds[oM0 + S + j] = INF; lds[oM1 + j] = INF; lds[oM1 + S + j] = INF;
        }
        if (tid < 2) lds[oT0 + ww + tid] = tid ? INF : REAL(0);   // sink children: cost to terminal 0 (top) / +inf (bot)
    }
    __syncthreads();
    constexpr bool SKIP_IDLE = BDDMMA_WIDE_SKIP_FWD(REAL, FWD_SOLVE);
    const uint32_t wv0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid);  // the wavefront's first thread (uniform)
    uint32_t fc = 0, cur = 0;  // frontier buffer fc: current, (fc+1)%3: next, (fc+2)%3: being cleared for the hop after
    // the costs to terminal of the hop two on (X.V, this thread's slots) go to LDS: phase A of the NEXT hop reads them (its children's)
    auto stage_t = [&](const Hop& X, uint32_t c2, uint32_t oFx, bool clear_frontier) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            if (j < c2) {
                if (clear_frontier) lds[oFx + j] = INF;
                lds[oT0 + j] = X.V[i];
            }
        }
    };
    // one hop: n / nb / lb its nodes, first slot, first layer; c2 = nodes of the hop two on; nl1 = layers of the next hop
    auto compute = [&](const Hop& X, uint32_t n, uint32_t nb, uint32_t lb, uint32_t c2, uint32_t nl1, uint32_t rt0) {
        const uint32_t oFc = fc * S, oFn = (fc == 2 ? 0 : fc + 1) * S, oFx = (fc == 0 ? 2 : fc - 1) * S;
        const uint32_t oMa = oM0 + cur * S, oMb = oM1 + cur * S, oMa_n = oM0 + (cur ^ 1) * S, oMb_n = oM1 + (cur ^ 1) * S;
        // ---- phase A: per-layer minima of the two min-marginals (children's costs to terminal: LDS, staged behind the previous hop's phase B)
        REAL f[NPT], tl[NPT], th[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            f[i] = act ? lds[oFc + j] : INF;
            if (j == rt0) f[i] = REAL(0);  // a BDD that starts at this hop: its root has no parents (flush_costs_from_root)
            tl[i] = lds[oT0 + ww_lo(X.W[i], ww)];
            th[i] = lds[oT0 + ww_hi(X.W[i], ww)];
            const uint32_t l = ww_layer(X.W[i]);
            REAL a = (f[i] + X.C[i].x) + tl[i], b = (f[i] + X.C[i].y) + th[i];
            const bool lead = seg_fold_by_key(a, b, act ? l : 0xFFFFFFFFu, (int)(tid & 63u));
            if (act && lead) {  // inactive lanes have nothing to contribute (and would all hit one address)
                lds_min(&lds[oMa + l], a);
                lds_min(&lds[oMb + l], b);
            }
        }
        __syncthreads();
        // ---- phase B: cost update, pushes into the next frontier
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            const uint64_t w = X.W[i];
            const uint32_t l = ww_layer(w), lo_i = ww_lo(w, ww), hi_i = ww_hi(w, ww);
            const REAL m0 = lds[oMa + l], m1 = lds[oMb + l];
            const REAL mm = mm_diff(m0, m1, omega);
            const REAL nlo = (X.C[i].x + min0(mm)) + X.D[i].x;
            const REAL nhi = (X.C[i].y + min0_neg(mm)) + X.D[i].y;
            const bool head = act && (w & WW_HEAD);
            P2 nc;
            nc.x = nlo;
            nc.y = nhi;
            bstore(nc, rs.lohi, head ? (lb + l) * (uint32_t)sizeof(P2) : OOB);
            bstore(mm, rs.mm, head ? X.E[i] * (uint32_t)sizeof(REAL) : OOB);
            const bool plo = lo_i < ww, phi = hi_i < ww;  // sink children and inactive lanes: no-op on a slot of their own (see k_fwd_narrow)
            if (act) {
                lds_min(&lds[oFn + (plo ? lo_i : j)], plo ? f[i] + nlo : INF);
                lds_min(&lds[oFn + (phi ? hi_i : j)], phi ? f[i] + nhi : INF);
            }
            bstore(f[i], rs.F, act ? (nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        // set-up of later hops: the frontier after next is cleared, the costs to terminal of the hop two on go to LDS, the minima of the next hop are reset
        stage_t(X, c2, oFx, true);
        for (uint32_t l = tid; l < nl1; l += T) { lds[oMa_n + l] = INF; lds[oMb_n + l] = INF; }
        __syncthreads();
        fc = fc == 2 ? 0 : fc + 1;
        cur ^= 1;
    };
    // requests of one hop h: {arc costs, delta pair} through its words / entries (arrived) and the costs to terminal of hop h + 2; or its {words, entries}
    auto request_cdv = [&](Hop& H, uint32_t c, uint32_t lb, uint32_t c2, uint32_t nb2) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            bload(H.C[i], rs.lohi, j < c ? (lb + ww_layer(H.W[i])) * (uint32_t)sizeof(P2) : OOB);
            bload(H.D[i], rs.dlay, j < c ? H.E[i] * (uint32_t)sizeof(P2) : OOB);
            bload(H.V[i], rs.T, j < c2 ? (nb2 + j) * (uint32_t)sizeof(REAL) : OOB);
        }
    };
    auto request_we = [&](Hop& G, uint32_t c, uint32_t nb) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            const uint64_t x = bload_u64(rs.words, j < c ? (nb - wsb + j) * 8u : OOB);
            G.W[i] = j < c ? x : WW_PAD_WORD;
            G.E[i] = bload_u32(rent, j < c ? (nb - wsb + j) * 4u : OOB);
        }
    };
    // register sets as in bwd_wide3_body: R the trip's hops (q, q+1), H the next trip's (q+2, q+3), G the words / entries of (q+4, q+5)
    Hop R0, R1, H0, H1, G0, G1;
    wide_hop_clear(R0); wide_hop_clear(R1); wide_hop_clear(H0); wide_hop_clear(H1); wide_hop_clear(G0); wide_hop_clear(G1);
    // offsets of the hops q .. q+7 / q .. q+5 (RN[j] = rn(q + j), RL[j] = rl(q + j)), roots of q, q+1; two more of each per trip
    int64_t q = q0 - 6;
    uint32_t RN[8], RL[6], RT[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) RN[j] = rn(q + j);
#pragma unroll
    for (int j = 0; j < 6; ++j) RL[j] = rl(q + j);
    RT[0] = RT[1] = NO_ROOT;
    for (; q < q1; q += 2) {   // three priming trips (requests only; the last stages the costs to terminal of hop q0 + 1), then the pack's hops
        const uint32_t rn_a = rn(q + 8), rn_b = rn(q + 9), rl_a = rl(q + 6), rl_b = rl(q + 7), rt_a = root_at(q + 2), rt_b = root_at(q + 3);
        request_cdv(H0, uni(RN[3] - RN[2]), uni(RL[2]), uni(RN[5] - RN[4]), uni(RN[4]));   // hop q+2 (costs to terminal: hop q+4)
        request_cdv(H1, uni(RN[4] - RN[3]), uni(RL[3]), uni(RN[6] - RN[5]), uni(RN[5]));   // hop q+3 (hop q+5)
        request_we(G0, uni(RN[5] - RN[4]), uni(RN[4]));                                    // hop q+4
        request_we(G1, uni(RN[6] - RN[5]), uni(RN[5]));                                    // hop q+5
        if (q >= q0) compute(R0, uni(RN[1] - RN[0]), uni(RN[0]), uni(RL[0]), uni(RN[3] - RN[2]), uni(RL[2] - RL[1]), uni(RT[0]));
        if (q + 1 >= q0 && q + 1 < q1) compute(R1, uni(RN[2] - RN[1]), uni(RN[1]), uni(RL[1]), uni(RN[4] - RN[3]), uni(RL[3] - RL[2]), uni(RT[1]));
        else if (q + 1 == q0 - 1) {   // the hop before the first: its "two on" is hop q0 + 1, the children of the first hop
            stage_t(R1, uni(RN[4] - RN[3]), 0, false);
            __syncthreads();
        }
        R0 = H0; R1 = H1;
#pragma unroll
        for (int i = 0; i < NPT; ++i) { H0.W[i] = G0.W[i]; H0.E[i] = G0.E[i]; H1.W[i] = G1.W[i]; H1.E[i] = G1.E[i]; }
#pragma unroll
        for (int j = 0; j < 6; ++j) RN[j] = RN[j + 2];
        RN[6] = rn_a; RN[7] = rn_b;
#pragma unroll
        for (int j = 0; j < 4; ++j) RL[j] = RL[j + 2];
        RL[4] = rl_a; RL[5] = rl_b;
        RT[0] = rt_a; RT[1] = rt_b;
    }
}

template <typename REAL, int NPT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) k_fwd_wide3(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    fwd_wide3_body<REAL, NPT>(d, pk, omega, ww, blockIdx.x);
}
template <typename REAL, int NPT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) k_bwd_wide3(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    bwd_wide3_body<REAL, NPT>(d, pk, omega, ww, blockIdx.x);
}

}  // namespace bddmma
