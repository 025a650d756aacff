// kernels/wide.hpp — wide and huge packs (k_*_wide, k_*_wide2) and the mixed launches (k_*_mixed).
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// wide packs: one workgroup per pack; layers may span waves, so the layer min goes through LDS
// =============================================================================================
constexpr int WIDE_THREADS = 256;

template <typename REAL>
struct WideLds {
    REAL* a;  // fwd: F cur/next ; bwd: T cur/next
    REAL* b;
    REAL* t;   // fwd: T of next hop
    REAL* m0;  // per-layer min-marginals
    REAL* m1;
    REAL* lc;  // per-layer costs staged so that the in-place cost update cannot race with readers
    REAL* hc;
    unsigned char* act0;
    unsigned char* act1;
};

__host__ __device__ inline size_t wide_lds_bytes(size_t real_size, uint32_t ww, bool solution)
{
    return 7 * real_size * ww + (solution ? 2 * ww : 0);
}

template <typename REAL>
__device__ __forceinline__ WideLds<REAL> carve_lds(unsigned char* base, uint32_t ww)
{
    WideLds<REAL> l;
    REAL* r = reinterpret_cast<REAL*>(base);
    l.a = r; l.b = r + ww; l.t = r + 2 * ww; l.m0 = r + 3 * ww; l.m1 = r + 4 * ww; l.lc = r + 5 * ww; l.hc = r + 6 * ww;
    l.act0 = base + 7 * sizeof(REAL) * ww;
    l.act1 = l.act0 + ww;
    return l;
}

// GLOBAL: the frontier arrays of pack p live at scratch + p * wide_lds_bytes(ww) in global memory (huge packs)
template <typename REAL, int MODE, bool GLOBAL = false>
__global__ void __launch_bounds__(WIDE_THREADS) k_fwd_wide(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww, unsigned char* scratch = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    WideLds<REAL> s = carve_lds<REAL>(GLOBAL ? scratch + (size_t)p * wide_lds_bytes(sizeof(REAL), ww, true) : smem, ww);
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    const REAL INF = inf_v<REAL>();
    REAL* Fc = s.a;
    REAL* Fn = s.b;
    unsigned char* Ac = s.act0;
    unsigned char* An = s.act1;
    uint32_t nb = pk.hop_node_off[q0], ne = pk.hop_node_off[q0 + 1];
    for (uint32_t j = tid; j < ne - nb; j += WIDE_THREADS) {
        Fc[j] = REAL(0);
        if (MODE == FWD_SOLUTION) Ac[j] = 1;
    }
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t n = ne - nb;
        const bool last = (q + 1 == q1);
        const uint32_t ne2 = last ? ne : pk.hop_node_off[q + 2];
        const uint32_t n2 = ne2 - ne;
        const uint32_t lbase = pk.hop_layer_off[q];
        const uint32_t nl = pk.hop_layer_off[q + 1] - lbase;
        for (uint32_t j = tid; j < n2; j += WIDE_THREADS) {
            if (MODE != FWD_PLAIN) s.t[j] = d.T[ne + j];
            Fn[j] = INF;
            if (MODE == FWD_SOLUTION) An[j] = 0;
        }
        for (uint32_t l = tid; l < nl; l += WIDE_THREADS) {
            s.m0[l] = INF;
            s.m1[l] = INF;
            s.lc[l] = d.lohi[2 * (size_t)(lbase + l)];
            s.hc[l] = d.lohi[2 * (size_t)(lbase + l) + 1];
        }
        __syncthreads();
        if (MODE == FWD_SOLVE) {
            for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
                const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
                const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
                const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
                const REAL f = frontier_load<GLOBAL>(&Fc[j]);
                const REAL tl = lo_i == WW_BOT ? INF : (lo_i == WW_TOP ? REAL(0) : s.t[lo_i]);
                const REAL th = hi_i == WW_BOT ? INF : (hi_i == WW_TOP ? REAL(0) : s.t[hi_i]);
                frontier_min<GLOBAL>(&s.m0[l], (f + s.lc[l]) + tl);
                frontier_min<GLOBAL>(&s.m1[l], (f + s.hc[l]) + th);
            }
            __syncthreads();
        }
        for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
            const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
            const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
            const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
            const REAL f = frontier_load<GLOBAL>(&Fc[j]);
            REAL nlo = s.lc[l], nhi = s.hc[l];
            if (MODE == FWD_SOLVE) {
                const REAL m0 = frontier_load<GLOBAL>(&s.m0[l]), m1 = frontier_load<GLOBAL>(&s.m1[l]);
                const uint32_t e = d.lpos[lbase + l];
                const REAL mm = mm_diff(m0, m1, omega);
                nlo = (nlo + min0(mm)) + d.delta_lay[2 * (size_t)e];
                nhi = (nhi + min0_neg(mm)) + d.delta_lay[2 * (size_t)e + 1];
                if (w & WW_HEAD) {
                    d.lohi[2 * (size_t)(lbase + l)] = nlo;
                    d.lohi[2 * (size_t)(lbase + l) + 1] = nhi;
                    d.mm_binned[e] = mm;
                }
            } else if (MODE == FWD_SOLUTION) {
                if (Ac[j]) {
                    const REAL tl = lo_i == WW_BOT ? INF : (lo_i == WW_TOP ? REAL(0) : s.t[lo_i]);
                    const REAL th = hi_i == WW_BOT ? INF : (hi_i == WW_TOP ? REAL(0) : s.t[hi_i]);
                    const REAL hi_path = f + (th + nhi);
                    const REAL lo_path = f + (tl + nlo);
                    const bool take_lo = (hi_path - lo_path) > 0;
                    d.sol_out[lbase + l] = take_lo ? 0 : 1;
                    const uint32_t c = take_lo ? lo_i : hi_i;
                    if (c < WW_TOP) An[c] = 1;
                }
            }
            if (lo_i < WW_TOP) frontier_min<GLOBAL>(&Fn[lo_i], f + nlo);
            if (hi_i < WW_TOP) frontier_min<GLOBAL>(&Fn[hi_i], f + nhi);
            d.F[nb + j] = f;
        }
        __syncthreads();
        REAL* tmp = Fc; Fc = Fn; Fn = tmp;
        unsigned char* ta = Ac; Ac = An; An = ta;
        nb = ne;
        ne = ne2;
    }
}

template <typename REAL, int MODE, bool GLOBAL = false>
__global__ void __launch_bounds__(WIDE_THREADS) k_bwd_wide(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww, unsigned char* scratch = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[WIDE_THREADS / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    WideLds<REAL> s = carve_lds<REAL>(GLOBAL ? scratch + (size_t)p * wide_lds_bytes(sizeof(REAL), ww, true) : smem, ww);
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    const REAL INF = inf_v<REAL>();
    REAL* Tc = s.a;  // T of hop q+1
    REAL* Tn = s.b;  // T of hop q (being written)
    for (uint32_t q = q1; q-- > q0;) {
        const uint32_t nb = pk.hop_node_off[q], ne = pk.hop_node_off[q + 1];
        const uint32_t n = ne - nb;
        const uint32_t lbase = pk.hop_layer_off[q];
        const uint32_t nl = pk.hop_layer_off[q + 1] - lbase;
        for (uint32_t l = tid; l < nl; l += WIDE_THREADS) {
            s.m0[l] = INF;
            s.m1[l] = INF;
            s.lc[l] = d.lohi[2 * (size_t)(lbase + l)];
            s.hc[l] = d.lohi[2 * (size_t)(lbase + l) + 1];
        }
        __syncthreads();
        if (MODE != BWD_PLAIN) {
            for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
                const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
                const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
                const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
                const REAL f = d.F[nb + j];
                const REAL tl = lo_i == WW_BOT ? INF : (lo_i == WW_TOP ? REAL(0) : Tc[lo_i]);
                const REAL th = hi_i == WW_BOT ? INF : (hi_i == WW_TOP ? REAL(0) : Tc[hi_i]);
                if (MODE == BWD_SOLVE) {
                    frontier_min<GLOBAL>(&s.m0[l], (f + s.lc[l]) + tl);
                    frontier_min<GLOBAL>(&s.m1[l], (f + s.hc[l]) + th);
                } else {
                    frontier_min<GLOBAL>(&s.m0[l], f + (tl + s.lc[l]));
                    frontier_min<GLOBAL>(&s.m1[l], f + (th + s.hc[l]));
                }
            }
            __syncthreads();
        }
        for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
            const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
            const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
            const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
            const REAL tl = lo_i == WW_BOT ? INF : (lo_i == WW_TOP ? REAL(0) : Tc[lo_i]);
            const REAL th = hi_i == WW_BOT ? INF : (hi_i == WW_TOP ? REAL(0) : Tc[hi_i]);
            REAL t;
            if (MODE == BWD_SOLVE) {
                const REAL m0 = frontier_load<GLOBAL>(&s.m0[l]), m1 = frontier_load<GLOBAL>(&s.m1[l]);
                const uint32_t e = d.lpos[lbase + l];
                const REAL mm = mm_diff(m0, m1, omega);
                const REAL nlo = (s.lc[l] + min0(mm)) + d.delta_lay[2 * (size_t)e];
                const REAL nhi = (s.hc[l] + min0_neg(mm)) + d.delta_lay[2 * (size_t)e + 1];
                t = rmin(nhi + th, nlo + tl);
                if (w & WW_HEAD) {
                    d.lohi[2 * (size_t)(lbase + l)] = nlo;
                    d.lohi[2 * (size_t)(lbase + l) + 1] = nhi;
                    d.mm_binned[e] = mm;
                    if (d.x_layer != nullptr) d.x_layer[lbase + l] = (nhi - nlo) + mm;
                }
            } else {
                t = rmin(th + s.hc[l], tl + s.lc[l]);
                if (MODE == BWD_MARGINALS && (w & WW_HEAD)) {
                    d.mm0_out[lbase + l] = frontier_load<GLOBAL>(&s.m0[l]);
                    d.mm1_out[lbase + l] = frontier_load<GLOBAL>(&s.m1[l]);
                }
            }
            Tn[j] = t;
            d.T[nb + j] = t;
        }
        __syncthreads();
        REAL* tmp = Tc; Tc = Tn; Tn = tmp;
    }
    const uint32_t n0 = pk.hop_node_off[q0 + 1] - pk.hop_node_off[q0];
    double acc = 0.0;
    for (uint32_t j = tid; j < n0; j += WIDE_THREADS) acc += (double)Tc[j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < WIDE_THREADS / 64; ++i) t += red[i];
        d.lb_partial[pk.lb_base + p] = t;
    }
}

// =============================================================================================
// wide packs, LDS frontier: register-resident rewrite of the workgroup-per-pack sweeps
// =============================================================================================
// One workgroup of T = blockDim.x threads (a multiple of 64, <= 1024) sweeps one wide pack; thread `tid` owns the nodes
// tid + i * T, i < NPT, of every hop (NPT = 1 for packs up to 1024 nodes per hop), so everything a node needs between the
// min-marginal phase and the update phase stays in registers and the node words are read once per hop.
//   * global loads are branch-free raw buffer ops issued three hops ahead (words, potentials), two hops ahead (the layer's
//     entry index, which needs the word) and one hop ahead (arc costs, delta pair), so a hop never waits for memory it asked
//     for in the same hop;
//   * the per-layer minimum goes through LDS (ds_min per node into the layer's slot: layers may span wavefronts);
//   * two workgroup barriers per hop in the solve / marginal modes (after the minima, after the pushes), one in the plain and
//     solution modes (three rotating frontier buffers make the second one unnecessary);
//   * sink children are ordinary LDS indices (ww = cost-to-terminal 0 / dummy push target, ww + 1 = +inf), as in the narrow kernels.
// The old k_*_wide kernels above remain for huge packs (frontier in global memory).
__host__ __device__ inline size_t wide2_lds_bytes(size_t real_size, uint32_t ww, bool solution)
{
    return 8 * real_size * (size_t)(ww + 2) + (solution ? 3 * (size_t)(ww + 2) : 0);
}
constexpr uint64_t WW_PAD_WORD = WW_BOT | (WW_BOT << WW_CHILD_BITS);  // inactive lane: children = bot sink, layer 0, not a head

template <typename REAL>
struct WideRs {
    rsrc_t words, T, F, lohi, lpos, dlay, mm;
    __device__ __forceinline__ explicit WideRs(const DevPtrs<REAL>& d)
    {
        words = make_rsrc(d.wwords, (uint64_t)d.n_slots - d.wide_slot_base);
        T = make_rsrc(d.T, d.n_slots);
        F = make_rsrc(d.F, d.n_slots);
        lohi = make_rsrc(d.lohi, 2ull * d.n_layers);
        lpos = make_rsrc(d.lpos, d.n_layers);
        dlay = make_rsrc(d.delta_lay, 2ull * d.n_layers);
        mm = make_rsrc(d.mm_binned, d.n_layers);
    }
};
__device__ __forceinline__ uint64_t bload_u64(rsrc_t r, uint32_t off)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return ((uint64_t)v[1] << 32) | (uint64_t)v[0];
}
__device__ __forceinline__ uint32_t ww_child(uint32_t c, uint32_t ww) { return c >= (uint32_t)WW_TOP ? ww + (c - (uint32_t)WW_TOP) : c; }
__device__ __forceinline__ uint32_t ww_lo(uint64_t w, uint32_t ww) { return ww_child((uint32_t)(w & WW_CHILD_MASK), ww); }
__device__ __forceinline__ uint32_t ww_hi(uint64_t w, uint32_t ww) { return ww_child((uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK), ww); }
__device__ __forceinline__ uint32_t ww_layer(uint64_t w) { return (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK); }

template <int NPT>
__device__ __forceinline__ void wide_load_words(uint64_t (&w)[NPT], rsrc_t words, uint32_t wb, uint32_t n, uint32_t tid, uint32_t T)
{
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const uint32_t j = tid + i * T;
        const uint64_t x = bload_u64(words, j < n ? (wb + j) * 8u : OOB);
        w[i] = j < n ? x : WW_PAD_WORD;
    }
}
template <typename REAL, int NPT>
__device__ __forceinline__ void wide_load_vals(REAL (&v)[NPT], rsrc_t src, uint32_t nb, uint32_t n, uint32_t tid, uint32_t T)
{
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const uint32_t j = tid + i * T;
        bload(v[i], src, j < n ? (nb + j) * (uint32_t)sizeof(REAL) : OOB);
    }
}

// Which wide sweeps skip the node slots of a wavefront that lie past the hop's last slot (phase A below).  Measured per mode (tools/exp_r05_t.sh,
// profiles/r05_wide_skip.txt, same box, two rounds): forward solve sweeps in float 25 000 rows of 18 variables 156 -> 147 us (2 883 -> 2 968 it/s),
// 4 000 rows 58 -> 49.5 us (7 630 -> 8 100 it/s), rows of 16 and the knapsack benchmark +-1 %; in double rows of 18 gain 5 % and rows of 16 lose 4 %
// (forward solve sweep 124 -> 141 us, reproducibly); plain sweeps lose 7-18 % (the branch costs them their counted waits), backward solve
// sweeps gain on the small instance only.  (Round 5 read this as "bound by the latency of their own prefetches, whose registers the rotation at
// the end of a hop touches".  Round 6 built that fix — every load two hops old at its first use, three forms, `git show 1eaaed8:bdd_amd/csrc/
// kernels/wide3.hpp`, profiles/r06_wide3.txt — and it was SLOWER (193 / 185 us against 134 / 183): the sweeps do not wait for memory either.  What
// is left is the hop itself: two workgroup barriers, per-layer LDS minima and frontier pushes for 120 active of 512 slots — the packing.)
#ifndef BDDMMA_WIDE_SKIP_FWD
#define BDDMMA_WIDE_SKIP_FWD(REAL, MODE) ((MODE) == FWD_SOLVE && sizeof(REAL) == 4)
#endif
#ifndef BDDMMA_WIDE_SKIP_BWD
#define BDDMMA_WIDE_SKIP_BWD(REAL, MODE) false
#endif
template <typename REAL, int MODE, int NPT>
__device__ __forceinline__ void fwd_wide2_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t ww, uint32_t p)
{
    using P2 = typename Pair<REAL>::type;
    constexpr bool NEED_T = (MODE != FWD_PLAIN);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    const uint32_t S = ww + 2;
    // All LDS arrays are addressed as lds[offset + index] with integer offsets that rotate from hop to hop: with rotating POINTERS the
    // compiler loses the address space and emits flat loads / a flat compare-and-swap loop for the float minimum (seen in the ISA).
    REAL* const lds = reinterpret_cast<REAL*>(smem);           // 8 arrays of S values: F x3, T, minima x4
    unsigned char* const ldsA = smem + 8 * sizeof(REAL) * S;   // 3 arrays of S flags (solution mode)
    // T of the next hop.  Solve mode: one buffer, rewritten in phase B (its readers are behind the phase-A barrier).  Solution mode has
    // no such barrier, so it alternates between two buffers (the second one is the space of the minima, unused there).
    const uint32_t oT0 = 3 * S, oT1 = MODE == FWD_SOLUTION ? 4 * S : 3 * S;
    const uint32_t oM0 = 4 * S, oM1 = 6 * S;  // minima of lo / hi: [oM0 + cur * S + l], [oM1 + cur * S + l]
    const REAL INF = inf_v<REAL>();
    const WideRs<REAL> rs(d);
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    auto noff = [&](uint32_t q) { return pk.hop_node_off[min(q, q1)]; };
    auto loff = [&](uint32_t q) { return pk.hop_layer_off[min(q, q1)]; };
    const uint32_t wsb = d.wide_slot_base;
    // node / layer offsets of hops q .. q+4 / q .. q+3
    uint32_t nv[5], lv[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) nv[i] = noff(q0 + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) lv[i] = loff(q0 + i);
    // ---- prologue: three dependent round trips, once per pack
    uint64_t W0[NPT], W1[NPT], W2[NPT];
    wide_load_words<NPT>(W0, rs.words, nv[0] - wsb, nv[1] - nv[0], tid, T);
    wide_load_words<NPT>(W1, rs.words, nv[1] - wsb, nv[2] - nv[1], tid, T);
    wide_load_words<NPT>(W2, rs.words, nv[2] - wsb, nv[3] - nv[2], tid, T);
    REAL T1[NPT], T2[NPT];
    if (NEED_T) {
        wide_load_vals<REAL, NPT>(T1, rs.T, nv[1], nv[2] - nv[1], tid, T);
        wide_load_vals<REAL, NPT>(T2, rs.T, nv[2], nv[3] - nv[2], tid, T);
    }
    uint32_t E0[NPT], E1[NPT];
    P2 C0[NPT], D0[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const uint32_t j = tid + i * T;
        const bool a0 = j < nv[1] - nv[0], a1 = j < nv[2] - nv[1];
        bload(C0[i], rs.lohi, a0 ? (lv[0] + ww_layer(W0[i])) * (uint32_t)sizeof(P2) : OOB);
        E0[i] = E1[i] = 0;
        if (MODE == FWD_SOLVE) {
            E0[i] = bload_u32(rs.lpos, a0 ? (lv[0] + ww_layer(W0[i])) * 4u : OOB);
            E1[i] = bload_u32(rs.lpos, a1 ? (lv[1] + ww_layer(W1[i])) * 4u : OOB);
        }
    }
    if (MODE == FWD_SOLVE) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) bload(D0[i], rs.dlay, (tid + i * T) < nv[1] - nv[0] ? E0[i] * (uint32_t)sizeof(P2) : OOB);
    }
    // LDS: roots, empty next frontiers, T of hop q0+1, empty minima
    for (uint32_t j = tid; j < S; j += T) {
        lds[j] = j < nv[1] - nv[0] ? REAL(0) : INF;  // every node of hop 0 is a root (flush_costs_from_root)
        lds[S + j] = INF;
        lds[2 * S + j] = INF;
        if (MODE == FWD_SOLVE) { lds[oM0 + j] = INF; lds[oM0 + S + j] = INF; lds[oM1 + j] = INF; lds[oM1 + S + j] = INF; }
        if (MODE == FWD_SOLUTION) { ldsA[j] = j < nv[1] - nv[0] ? 1 : 0; ldsA[S + j] = 0; ldsA[2 * S + j] = 0; }
    }
    if (NEED_T) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            if (j < nv[2] - nv[1]) lds[oT0 + j] = T1[i];
        }
        if (tid < 4) lds[((tid >> 1) ? oT1 : oT0) + ww + (tid & 1)] = (tid & 1) ? INF : REAL(0);
    }
    __syncthreads();
    const uint32_t wv0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid);  // the wavefront's first thread (uniform)
    constexpr bool SKIP_IDLE = BDDMMA_WIDE_SKIP_FWD(REAL, MODE);
    uint32_t fc = 0, cur = 0;  // frontier buffer fc: current, (fc+1)%3: next, (fc+2)%3: being cleared for the hop after
    // staggered wide packs: slot of the BDD that starts at hop q (below the pack's first hop), read two hops ahead like the offsets
    auto root_at = [&](uint32_t q) -> uint32_t { return (q > q0 && q < q1) ? (uint32_t)pk.hop_root[q] : (uint32_t)NO_ROOT; };
    uint32_t rt0 = NO_ROOT, rt1 = root_at(q0 + 1);
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t rt2 = root_at(q + 2);
        const uint32_t n = nv[1] - nv[0];
        const uint32_t oFc = fc * S, oFn = (fc == 2 ? 0 : fc + 1) * S, oFx = (fc == 0 ? 2 : fc - 1) * S;
        const uint32_t oT = cur ? oT1 : oT0, oTn = cur ? oT0 : oT1;
        const uint32_t oMa = oM0 + cur * S, oMb = oM1 + cur * S, oMa_n = oM0 + (cur ^ 1) * S, oMb_n = oM1 + (cur ^ 1) * S;
        // ---- prefetch (consumed in later hops): words / T of hop q+3, entry indices of hop q+2, arc costs and delta pairs of hop q+1
        uint64_t W3[NPT];
        REAL T3[NPT];
        uint32_t E2[NPT];
        P2 C1[NPT], D1[NPT];
        const uint32_t nv5 = noff(q + 5), lv4 = loff(q + 4);
        wide_load_words<NPT>(W3, rs.words, nv[3] - wsb, nv[4] - nv[3], tid, T);
        if (NEED_T) wide_load_vals<REAL, NPT>(T3, rs.T, nv[3], nv[4] - nv[3], tid, T);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            const bool a1 = j < nv[2] - nv[1], a2 = j < nv[3] - nv[2];
            bload(C1[i], rs.lohi, a1 ? (lv[1] + ww_layer(W1[i])) * (uint32_t)sizeof(P2) : OOB);
            E2[i] = 0;
            if (MODE == FWD_SOLVE) {
                E2[i] = bload_u32(rs.lpos, a2 ? (lv[2] + ww_layer(W2[i])) * 4u : OOB);
                bload(D1[i], rs.dlay, a1 ? E1[i] * (uint32_t)sizeof(P2) : OOB);
            }
        }
        // ---- phase A: per-layer minima of the two min-marginals
        // SKIP_IDLE: a wavefront whose i-th nodes all lie past the hop's last slot skips them in both phases (a scalar branch): a chained wide
        // pack is as wide as its widest BDD and BDDs of general linear rows are diamonds — the mean hop of the 25 000-row instance fills a
        // quarter of its 512 slots (everything the skipped code does is predicated on `act`).
        REAL f[NPT], tl[NPT], th[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            f[i] = act ? lds[oFc + j] : INF;
            if (j == rt0) f[i] = REAL(0);  // a BDD that starts at this hop: its root has no parents (flush_costs_from_root)
            if (NEED_T) {
                tl[i] = lds[oT + ww_lo(W0[i], ww)];
                th[i] = lds[oT + ww_hi(W0[i], ww)];
            }
            if (MODE == FWD_SOLVE) {
                const uint32_t l = ww_layer(W0[i]);
                REAL a = (f[i] + C0[i].x) + tl[i], b = (f[i] + C0[i].y) + th[i];
                const bool lead = seg_fold_by_key(a, b, act ? l : 0xFFFFFFFFu, (int)(tid & 63u));
                if (act && lead) {  // inactive lanes have nothing to contribute (and would all hit one address)
                    lds_min(&lds[oMa + l], a);
                    lds_min(&lds[oMb + l], b);
                }
            }
        }
        if (MODE == FWD_SOLVE) __syncthreads();
        // ---- phase B: cost update, pushes into the next frontier
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            const uint64_t w = W0[i];
            const uint32_t l = ww_layer(w), lo_i = ww_lo(w, ww), hi_i = ww_hi(w, ww);
            REAL nlo = C0[i].x, nhi = C0[i].y;
            if (MODE == FWD_SOLVE) {
                const REAL m0 = lds[oMa + l], m1 = lds[oMb + l];
                const REAL mm = mm_diff(m0, m1, omega);
                nlo = (nlo + min0(mm)) + D0[i].x;
                nhi = (nhi + min0_neg(mm)) + D0[i].y;
                const bool head = act && (w & WW_HEAD);
                P2 nc;
                nc.x = nlo;
                nc.y = nhi;
                bstore(nc, rs.lohi, head ? (lv[0] + l) * (uint32_t)sizeof(P2) : OOB);
                bstore(mm, rs.mm, head ? E0[i] * (uint32_t)sizeof(REAL) : OOB);
            } else if (MODE == FWD_SOLUTION) {
                if (act && (ldsA[oFc + j] || j == rt0)) {
                    const REAL hi_path = f[i] + (th[i] + nhi);  // backward_step_with_path_costs, bdd_cuda_base.cu:633-640
                    const REAL lo_path = f[i] + (tl[i] + nlo);
                    const bool take_lo = (hi_path - lo_path) > 0;
                    d.sol_out[lv[0] + l] = take_lo ? 0 : 1;
                    ldsA[oFn + (take_lo ? lo_i : hi_i)] = 1;  // sink entries are dummies
                }
            }
            const bool plo = lo_i < ww, phi = hi_i < ww;  // sink children and inactive lanes: no-op on a slot of their own (see k_fwd_narrow)
            if (act) {
                lds_min(&lds[oFn + (plo ? lo_i : j)], plo ? f[i] + nlo : INF);
                lds_min(&lds[oFn + (phi ? hi_i : j)], phi ? f[i] + nhi : INF);
            }
            if (MODE != FWD_SOLUTION) bstore(f[i], rs.F, act ? (nv[0] + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        // set-up of later hops: the frontier after next is cleared, T of hop q+2 goes to LDS (phase A of the next hop reads it),
        // the minima of hop q+1 are reset
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            if (j < nv[3] - nv[2]) {
                lds[oFx + j] = INF;
                if (MODE == FWD_SOLUTION) ldsA[oFx + j] = 0;
                if (NEED_T) lds[oTn + j] = T2[i];
            }
        }
        if (MODE == FWD_SOLVE) {
            const uint32_t nl1 = lv[2] - lv[1];
            for (uint32_t l = tid; l < nl1; l += T) { lds[oMa_n + l] = INF; lds[oMb_n + l] = INF; }
        }
        __syncthreads();
        // rotate
        fc = fc == 2 ? 0 : fc + 1;
        cur ^= 1;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            W0[i] = W1[i]; W1[i] = W2[i]; W2[i] = W3[i];
            E0[i] = E1[i]; E1[i] = E2[i];
            C0[i] = C1[i];
            if (MODE == FWD_SOLVE) D0[i] = D1[i];
            if (NEED_T) T2[i] = T3[i];
        }
        nv[0] = nv[1]; nv[1] = nv[2]; nv[2] = nv[3]; nv[3] = nv[4]; nv[4] = nv5;
        lv[0] = lv[1]; lv[1] = lv[2]; lv[2] = lv[3]; lv[3] = lv4;
        rt0 = rt1; rt1 = rt2;
    }
}

// Register budget of the wide sweeps: 4 waves per SIMD = what a 1024-thread workgroup implies (the solve sweeps with two nodes per thread
// hold 80 / 75 VGPRs = 6 waves).  Measured and not kept (tools/exp_r05_s.sh, 25 000 rows of 18 variables, float): 72 VGPRs (7 waves, 3-4
// spills) 2 890 -> 2 550 it/s, 64 (8 waves, 14-17 spills) 1 420.
#ifndef BDDMMA_WIDE2_WAVES
#define BDDMMA_WIDE2_WAVES(REAL, MODE, NPT) 4
#endif
template <typename REAL, int MODE, int NPT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(BDDMMA_WIDE2_WAVES(REAL, MODE, NPT)))) k_fwd_wide2(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    fwd_wide2_body<REAL, MODE, NPT>(d, pk, omega, ww, blockIdx.x);
}

template <typename REAL, int MODE, int NPT>
__device__ __forceinline__ void bwd_wide2_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t ww, uint32_t p)
{
    using P2 = typename Pair<REAL>::type;
    constexpr bool NEED_F = (MODE != BWD_PLAIN);
    constexpr bool NEED_M = (MODE != BWD_PLAIN);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[16];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    BDDMMA_EXIT_IF(p >= pk.n_packs, d)
    const uint32_t S = ww + 2;
    REAL* const lds = reinterpret_cast<REAL*>(smem);  // integer offsets instead of rotating pointers, see k_fwd_wide2
    const uint32_t oM0 = 4 * S, oM1 = 6 * S;
    const REAL INF = inf_v<REAL>();
    const WideRs<REAL> rs(d);
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    const uint32_t wsb = d.wide_slot_base;
    // hop h below the current one: node range [nb(h), nb(h+1)), empty below q0
    auto nb_of = [&](int64_t h) { return pk.hop_node_off[h < (int64_t)q0 ? q0 : (uint32_t)h]; };
    auto cnt_of = [&](int64_t h) { return h < (int64_t)q0 ? 0u : pk.hop_node_off[h + 1] - pk.hop_node_off[h]; };
    auto lb_of = [&](int64_t h) { return pk.hop_layer_off[h < (int64_t)q0 ? q0 : (uint32_t)h]; };
    auto nl_of = [&](int64_t h) { return h < (int64_t)q0 ? 0u : pk.hop_layer_off[h + 1] - pk.hop_layer_off[h]; };
    int64_t q = (int64_t)q1 - 1;
    // ---- prologue
    uint64_t W0[NPT], W1[NPT], W2[NPT];
    REAL F0[NPT], F1[NPT], F2[NPT];
    wide_load_words<NPT>(W0, rs.words, nb_of(q) - wsb, cnt_of(q), tid, T);
    wide_load_words<NPT>(W1, rs.words, nb_of(q - 1) - wsb, cnt_of(q - 1), tid, T);
    wide_load_words<NPT>(W2, rs.words, nb_of(q - 2) - wsb, cnt_of(q - 2), tid, T);
    if (NEED_F) {
        wide_load_vals<REAL, NPT>(F0, rs.F, nb_of(q), cnt_of(q), tid, T);
        wide_load_vals<REAL, NPT>(F1, rs.F, nb_of(q - 1), cnt_of(q - 1), tid, T);
        wide_load_vals<REAL, NPT>(F2, rs.F, nb_of(q - 2), cnt_of(q - 2), tid, T);
    }
    uint32_t E0[NPT], E1[NPT];
    P2 C0[NPT], D0[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const uint32_t j = tid + i * T;
        const bool a0 = j < cnt_of(q), a1 = j < cnt_of(q - 1);
        bload(C0[i], rs.lohi, a0 ? (lb_of(q) + ww_layer(W0[i])) * (uint32_t)sizeof(P2) : OOB);
        E0[i] = E1[i] = 0;
        if (MODE == BWD_SOLVE) {
            E0[i] = bload_u32(rs.lpos, a0 ? (lb_of(q) + ww_layer(W0[i])) * 4u : OOB);
            E1[i] = bload_u32(rs.lpos, a1 ? (lb_of(q - 1) + ww_layer(W1[i])) * 4u : OOB);
        }
    }
    if (MODE == BWD_SOLVE) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) bload(D0[i], rs.dlay, (tid + i * T) < cnt_of(q) ? E0[i] * (uint32_t)sizeof(P2) : OOB);
    }
    for (uint32_t j = tid; j < S; j += T) {
        if (NEED_M) { lds[oM0 + j] = INF; lds[oM0 + S + j] = INF; lds[oM1 + j] = INF; lds[oM1 + S + j] = INF; }
    }
    if (tid < 4) lds[(tid >> 1) * S + ww + (tid & 1)] = (tid & 1) ? INF : REAL(0);
    __syncthreads();
    const uint32_t wv0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid);  // the wavefront's first thread (uniform)
    constexpr bool SKIP_IDLE = BDDMMA_WIDE_SKIP_BWD(REAL, MODE);
    uint32_t tc = 0, cur = 0;  // T buffer tc: hop q+1 (children), tc^1: hop q (being written)
    // staggered wide packs: the root that sits at hop q below the pack's first hop contributes its cost-to-terminal to the lower bound
    auto root_at = [&](int64_t h) -> uint32_t { return h > (int64_t)q0 ? (uint32_t)pk.hop_root[h] : (uint32_t)NO_ROOT; };
    uint32_t rt0 = root_at(q), rt1 = root_at(q - 1);
    double lb_stag = 0.0;
    for (; q >= (int64_t)q0; --q) {
        const uint32_t rt2 = root_at(q - 2);
        const uint32_t n = cnt_of(q), nb = nb_of(q), lb = lb_of(q);
        const uint32_t oTc = tc * S, oTn = (tc ^ 1) * S;
        const uint32_t oMa = oM0 + cur * S, oMb = oM1 + cur * S, oMa_n = oM0 + (cur ^ 1) * S, oMb_n = oM1 + (cur ^ 1) * S;
        // ---- prefetch: words / F of hop q-3, entry indices of hop q-2, arc costs and delta pairs of hop q-1
        uint64_t W3[NPT];
        REAL F3[NPT];
        uint32_t E2[NPT];
        P2 C1[NPT], D1[NPT];
        const uint32_t c1 = cnt_of(q - 1), c2 = cnt_of(q - 2), c3 = cnt_of(q - 3);
        const uint32_t lb1 = lb_of(q - 1), lb2 = lb_of(q - 2);
        wide_load_words<NPT>(W3, rs.words, nb_of(q - 3) - wsb, c3, tid, T);
        if (NEED_F) wide_load_vals<REAL, NPT>(F3, rs.F, nb_of(q - 3), c3, tid, T);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const uint32_t j = tid + i * T;
            bload(C1[i], rs.lohi, j < c1 ? (lb1 + ww_layer(W1[i])) * (uint32_t)sizeof(P2) : OOB);
            E2[i] = 0;
            if (MODE == BWD_SOLVE) {
                E2[i] = bload_u32(rs.lpos, j < c2 ? (lb2 + ww_layer(W2[i])) * 4u : OOB);
                bload(D1[i], rs.dlay, j < c1 ? E1[i] * (uint32_t)sizeof(P2) : OOB);
            }
        }
        // ---- phase A
        REAL tl[NPT], th[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;  // no node of this wavefront in the hop: see k_fwd_wide2
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            tl[i] = lds[oTc + ww_lo(W0[i], ww)];
            th[i] = lds[oTc + ww_hi(W0[i], ww)];
            if (NEED_M) {
                const uint32_t l = ww_layer(W0[i]);
                REAL a, b;
                if (MODE == BWD_SOLVE) {
                    a = (F0[i] + C0[i].x) + tl[i];
                    b = (F0[i] + C0[i].y) + th[i];
                } else {  // backward_step_with_path_costs, bdd_cuda_base.cu:633-641
                    a = F0[i] + (tl[i] + C0[i].x);
                    b = F0[i] + (th[i] + C0[i].y);
                }
                const bool lead = seg_fold_by_key(a, b, act ? l : 0xFFFFFFFFu, (int)(tid & 63u));
                if (act && lead) {
                    lds_min(&lds[oMa + l], a);
                    lds_min(&lds[oMb + l], b);
                }
            }
        }
        if (NEED_M) __syncthreads();
        // ---- phase B
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (SKIP_IDLE && !(wv0 + i * T < n)) continue;
            const uint32_t j = tid + i * T;
            const bool act = j < n;
            const uint64_t w = W0[i];
            const uint32_t l = ww_layer(w);
            const bool head = act && (w & WW_HEAD);
            REAL t;
            if (MODE == BWD_SOLVE) {
                const REAL m0 = lds[oMa + l], m1 = lds[oMb + l];
                const REAL mm = mm_diff(m0, m1, omega);
                const REAL nlo = (C0[i].x + min0(mm)) + D0[i].x;
                const REAL nhi = (C0[i].y + min0_neg(mm)) + D0[i].y;
                t = rmin(nhi + th[i], nlo + tl[i]);
                P2 nc;
                nc.x = nlo;
                nc.y = nhi;
                bstore(nc, rs.lohi, head ? (lb + l) * (uint32_t)sizeof(P2) : OOB);
                bstore(mm, rs.mm, head ? E0[i] * (uint32_t)sizeof(REAL) : OOB);
                if (d.x_layer != nullptr && head) d.x_layer[lb + l] = (nhi - nlo) + mm;
            } else {
                t = rmin(th[i] + C0[i].y, tl[i] + C0[i].x);  // backward_step, bdd_cuda_base.cu:646-667
                if (MODE == BWD_MARGINALS && head) {
                    d.mm0_out[lb + l] = lds[oMa + l];
                    d.mm1_out[lb + l] = lds[oMb + l];
                }
            }
            if (act) lds[oTn + j] = t;
            if (j == rt0) lb_stag += (double)t;
            bstore(t, rs.T, act ? (nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        if (NEED_M) {
            const uint32_t nl1 = nl_of(q - 1);
            for (uint32_t l = tid; l < nl1; l += T) { lds[oMa_n + l] = INF; lds[oMb_n + l] = INF; }
        }
        __syncthreads();
        tc ^= 1;
        cur ^= 1;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            W0[i] = W1[i]; W1[i] = W2[i]; W2[i] = W3[i];
            if (NEED_F) { F0[i] = F1[i]; F1[i] = F2[i]; F2[i] = F3[i]; }
            E0[i] = E1[i]; E1[i] = E2[i];
            C0[i] = C1[i];
            if (MODE == BWD_SOLVE) D0[i] = D1[i];
        }
        rt0 = rt1; rt1 = rt2;
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

template <typename REAL, int MODE, int NPT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(BDDMMA_WIDE2_WAVES(REAL, MODE, NPT)))) k_bwd_wide2(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    bwd_wide2_body<REAL, MODE, NPT>(d, pk, omega, ww, blockIdx.x);
}

// Instances with narrow AND wide packs: one launch for both.  The first n_wide workgroups sweep one wide pack each (they are the
// long pole, so they start first), the rest are the narrow launch unchanged; the workgroup size is the narrow one (64 * WPB threads,
// NPT = nodes of a wide hop per thread).  Sequential launches add their times (61 us = 36 + 20 + boundary on the knapsack
// benchmark), a second stream costs more in event fork / join than it returns; inside one grid the two kinds of workgroups simply
// share the CUs.
// (the narrow part stays first generation: instances with wide packs are general linear rows, whose BDDs share no structure templates, and
// per-lane records that are not shared cost four times the node words' bytes — 40 000 knapsack rows, 10 M nodes: sweeps 73 / 73 us with
// node words, 91 / 110 us with records, profiles/r04_widebench.txt)
// Register cap of the mixed kernels (round 5; tools/exp_r05_q.sh, same-box A/B, it/s float): uncapped they hold 101-106 VGPRs = 4 waves per
// SIMD; 96 (5 waves, no spills with 64-slot packs, 2 with 128-slot ones) and 80 (6 waves, 2-5 spills, 64-slot packs only) give 40 000
// knapsack rows 6 580 -> 6 860 / 6 890, 30 000 + 100 000 covering rows 6 910 -> 7 210 / 7 590, 4 000 rows 17 650 -> 20 170 / 20 170;
// 72 registers spill 15, 64 spill 70.  Double loses with every cap (40 000 rows 4 360 -> 4 110 / 2 890) and stays uncapped, as do
// the kernels with four nodes of a wide hop per thread (149 VGPRs).
#ifndef BDDMMA_MIXED_WAVES
#define BDDMMA_MIXED_WAVES(REAL, R, NPT) ((NPT) == 4 ? 3 : sizeof(REAL) == 4 ? ((R) == 1 ? 6 : 5) : 4)
#endif
template <typename REAL, int R, int WPB, int NPT>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_MIXED_WAVES(REAL, R, NPT)))) k_fwd_mixed(DevPtrs<REAL> d, PackDev pkn, PackDev pkw, REAL omega, uint32_t ww)
{
    const uint32_t nw8 = (pkw.n_packs + 7u) & ~7u;  // a multiple of 8, so that the narrow workgroups keep their XCD-aware block -> pack map
    if (blockIdx.x < nw8) fwd_wide2_body<REAL, FWD_SOLVE, NPT>(d, pkw, omega, ww, blockIdx.x);
    else fwd_narrow_body<REAL, R, FWD_SOLVE, WPB>(d, pkn, omega, blockIdx.x - nw8);
}
template <typename REAL, int R, int WPB, int NPT>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_MIXED_WAVES(REAL, R, NPT)))) k_bwd_mixed(DevPtrs<REAL> d, PackDev pkn, PackDev pkw, REAL omega, uint32_t ww)
{
    const uint32_t nw8 = (pkw.n_packs + 7u) & ~7u;
    if (blockIdx.x < nw8) bwd_wide2_body<REAL, BWD_SOLVE, NPT>(d, pkw, omega, ww, blockIdx.x);
    else bwd_narrow_body<REAL, R, BWD_SOLVE, WPB>(d, pkn, omega, blockIdx.x - nw8);
}

}  // namespace bddmma
