// kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4) of the parallel-MMA hot path.
//
// One kernel launch sweeps ALL BDDs for a whole pass (the reference launches 3 kernels per hop,
// bdd_cuda_parallel_mma.cu:207-257,301-346).  A *pack* of BDDs is walked hop by hop by one
// wavefront (narrow packs, 64 threads, barrier-free) or one workgroup (wide packs):
//   - node words / potentials / layer costs are hop-major SoA inside the pack, so every global
//     access of a wave is a contiguous stream (coalesced);
//   - the frontier (cost-from-root of the current and next hop, cost-from-terminal of the next
//     hop) lives in LDS; children are addressed by their local index inside the next hop;
//   - the per-layer min-marginal is a segmented wavefront reduction: __ballot of the layer-head
//     flags gives the segment boundaries, __shfl_down halving steps do the min, __shfl broadcasts;
//   - no MFMA: this is an HBM-bound gather/scan (2 flops per 4-8 bytes).
//
// Arithmetic order follows the reference exactly (SURVEY.md §8 a'):
//   m0 = (F[u] + lo) + T[lo(u)],  m1 = (F[u] + hi) + T[hi(u)]          bdd_cuda_parallel_mma.cu:83-84
//   mm = finite(m0) && finite(m1) ? omega * (m1 - m0) : 0              :36-39
//   lo' = (lo + min(mm,0)) + delta[2v],  hi' = (hi + min(-mm,0)) + delta[2v+1]   :191-197, :286-287
//   F[child] = min(F[child], F[u] + cost')                             :194-198
//   T[u] = min(hi' + T[hi(u)], lo' + T[lo(u)])                         :292
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "layout.hpp"

namespace bddmma {

enum : int { FWD_PLAIN = 0, FWD_SOLVE = 1, FWD_SOLUTION = 2 };
enum : int { BWD_PLAIN = 0, BWD_SOLVE = 1, BWD_MARGINALS = 2 };

template <typename REAL>
struct DevPtrs {
    const uint32_t* nwords;  // narrow node words, indexed by slot
    const uint64_t* wwords;  // wide node words, indexed by slot - wide_slot_base
    uint32_t wide_slot_base;
    REAL* F;                 // cost from root, per slot
    REAL* T;                 // cost from terminal, per slot
    REAL* lo;                // per layer
    REAL* hi;
    REAL* mm;                // deferred min-marginal difference, per layer
    const int32_t* var;      // per layer
    const REAL* delta_in;    // 2V, values to add (already normalised)
    REAL* delta_out;         // N_XCD x 2V accumulators, one per XCD (nullptr: skip accumulation)
    uint32_t delta_stride;   // 2V
    double* lb_partial;      // per pack (narrow packs first, then wide)
    REAL* mm0_out;           // BWD_MARGINALS outputs, per layer
    REAL* mm1_out;
    char* sol_out;           // FWD_SOLUTION output, per layer
};

struct PackDev {
    const uint32_t* pack_hop_ptr;
    const uint32_t* hop_node_off;
    const uint32_t* hop_layer_off;
    const uint8_t* pack_steps;
    uint32_t n_packs;
    uint32_t lb_base;  // index of this set's first pack in lb_partial
};

template <typename REAL> __device__ __forceinline__ REAL inf_v();
template <> __device__ __forceinline__ float inf_v<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_v<double>() { return __builtin_huge_val(); }

__device__ __forceinline__ float rmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double rmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ bool rfinite(float a) { return __builtin_isfinite(a); }
__device__ __forceinline__ bool rfinite(double a) { return __builtin_isfinite(a); }

template <typename REAL>
__device__ __forceinline__ void lds_min(REAL* p, REAL v)
{
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_min_f32 / ds_min_f64
}

// XCD-aware block -> pack map: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md),
// so giving XCD x the contiguous pack range [x*per, (x+1)*per) keeps neighbouring packs — which
// share variables in structured problems — behind the same 4 MiB L2.
__device__ __forceinline__ uint32_t block_to_pack(uint32_t bid, uint32_t n_packs)
{
    const uint32_t per = (n_packs + 7u) >> 3;
    return (bid & 7u) * per + (bid >> 3);
}

// Segmented min of (a, b) over runs of lanes that belong to one layer, result broadcast to every
// lane of the run.  `heads` = __ballot(lane is the first node of its layer).
template <typename REAL>
__device__ __forceinline__ void seg_min2(REAL& a, REAL& b, int lane, unsigned long long heads, int steps)
{
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long above = heads & ~le;
    const int seg_end = above ? (__ffsll((long long)above) - 1) : 64;
    for (int s = 0; s < steps; ++s) {
        const int off = 1 << s;
        const REAL a2 = __shfl_down(a, off);
        const REAL b2 = __shfl_down(b, off);
        if (lane + off < seg_end) {
            a = rmin(a, a2);
            b = rmin(b, b2);
        }
    }
    const int seg_start = 63 - __clzll((long long)(heads & le));
    a = __shfl(a, seg_start);
    b = __shfl(b, seg_start);
}

// Per-XCD accumulators.  The 8 XCD L2s are kept coherent by probes, so float atomics from all XCDs
// on one 2V-array ping-pong its lines between L2s (measured: 5 M atomicAdd cost 180 us per pass,
// 2.4x the rest of the sweep).  Each workgroup therefore adds into the accumulator of the XCD it is
// actually running on (HW_REG_XCC_ID), whose lines only that XCD's L2 ever owns; k_finish_delta
// sums the N_XCD slices.  Correctness does not depend on the placement: any slice index is valid.
constexpr int N_XCD = 8;
__device__ __forceinline__ uint32_t xcc_id()
{
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (N_XCD - 1);  // hwreg(HW_REG_XCC_ID, 0, 4)
}

template <typename REAL>
__device__ __forceinline__ void accumulate_delta(REAL* delta_out, int v, REAL mm)
{
    // compute_delta_atomic, bdd_cuda_parallel_mma.cu:358-376
    if (mm > 0) atomicAdd(&delta_out[2 * v + 1], mm);
    else if (mm < 0) atomicAdd(&delta_out[2 * v], -mm);
}

// =============================================================================================
// narrow packs: one wavefront per pack, R groups of 64 slots per hop, no barriers
// =============================================================================================
template <typename REAL, int R, int MODE>
__global__ void __launch_bounds__(64) k_fwd_narrow(DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    constexpr int W = 64 * R;
    __shared__ REAL sF[2][W];
    __shared__ REAL sT[W];
    __shared__ unsigned char sAct[2][MODE == FWD_SOLUTION ? W : 1];
    const int lane = threadIdx.x;
    const uint32_t p = block_to_pack(blockIdx.x, pk.n_packs);
    if (p >= pk.n_packs) return;
    if (d.delta_out) d.delta_out += (size_t)xcc_id() * d.delta_stride;
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    const int steps = pk.pack_steps[p];
    const REAL INF = inf_v<REAL>();
    uint32_t nb = pk.hop_node_off[q0], ne = pk.hop_node_off[q0 + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        sF[0][j] = (j < ne - nb) ? REAL(0) : INF;  // every slot of hop 0 is a root (flush_costs_from_root)
        if (MODE == FWD_SOLUTION) sAct[0][j] = (j < ne - nb) ? 1 : 0;
    }
    int cur = 0;
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t n = ne - nb;
        const bool last = (q + 1 == q1);
        const uint32_t ne2 = last ? ne : pk.hop_node_off[q + 2];
        const uint32_t n2 = ne2 - ne;
        const uint32_t lbase = pk.hop_layer_off[q];
        uint32_t w[R];
        REAL f[R], lc[R], hc[R], d0[R], d1[R];
        int v[R];
        uint32_t lg[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            if (MODE != FWD_PLAIN && j < n2) sT[j] = d.T[ne + j];
            sF[cur ^ 1][j] = INF;
            if (MODE == FWD_SOLUTION) sAct[cur ^ 1][j] = 0;
            w[r] = (j < n) ? d.nwords[nb + j] : NW_PAD_WORD;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            const bool act = !(w[r] & NW_PAD);
            lg[r] = lbase + ((w[r] >> (2 * NW_CHILD_BITS)) & 1023u);
            lc[r] = act ? d.lo[lg[r]] : REAL(0);
            hc[r] = act ? d.hi[lg[r]] : REAL(0);
            v[r] = (MODE == FWD_SOLVE && act) ? d.var[lg[r]] : 0;
            f[r] = sF[cur][j];
        }
        if (MODE == FWD_SOLVE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool act = !(w[r] & NW_PAD);
                d0[r] = act ? d.delta_in[2 * v[r]] : REAL(0);
                d1[r] = act ? d.delta_in[2 * v[r] + 1] : REAL(0);
            }
        }
        __syncthreads();  // one wave: compiles to a wait on the LDS stores above
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            const bool act = !(w[r] & NW_PAD);
            const uint32_t lo_i = w[r] & NW_CHILD_MASK, hi_i = (w[r] >> NW_CHILD_BITS) & NW_CHILD_MASK;
            REAL nlo = lc[r], nhi = hc[r];
            if (MODE == FWD_SOLVE || MODE == FWD_SOLUTION) {
                const REAL tl = lo_i == NW_BOT ? INF : (lo_i == NW_TOP ? REAL(0) : sT[lo_i]);
                const REAL th = hi_i == NW_BOT ? INF : (hi_i == NW_TOP ? REAL(0) : sT[hi_i]);
                if (MODE == FWD_SOLVE) {
                    REAL m0 = act ? (f[r] + lc[r]) + tl : INF;
                    REAL m1 = act ? (f[r] + hc[r]) + th : INF;
                    const unsigned long long heads = __ballot((w[r] & NW_HEAD) != 0);
                    seg_min2(m0, m1, lane, heads, steps);
                    const REAL mm = (rfinite(m0) && rfinite(m1)) ? omega * (m1 - m0) : REAL(0);
                    nlo = (lc[r] + rmin(mm, REAL(0))) + d0[r];
                    nhi = (hc[r] + rmin(-mm, REAL(0))) + d1[r];
                    if (act && (w[r] & NW_HEAD)) {
                        d.lo[lg[r]] = nlo;
                        d.hi[lg[r]] = nhi;
                        d.mm[lg[r]] = mm;
                        if (d.delta_out) accumulate_delta(d.delta_out, v[r], mm);
                    }
                } else {
                    // compute_bdd_sol_func, bdd_cuda_base.cu:1103-1137 (with the `< 0` fix of SURVEY.md §8)
                    if (act && sAct[cur][j]) {
                        const REAL hi_path = f[r] + (th + hc[r]);  // backward_step_with_path_costs, :633-640
                        const REAL lo_path = f[r] + (tl + lc[r]);
                        const bool take_lo = (hi_path - lo_path) > 0;
                        d.sol_out[lg[r]] = take_lo ? 0 : 1;
                        const uint32_t c = take_lo ? lo_i : hi_i;
                        if (c < NW_TOP) sAct[cur ^ 1][c] = 1;
                    }
                }
            }
            if (act) {
                if (lo_i < NW_TOP) lds_min(&sF[cur ^ 1][lo_i], f[r] + nlo);
                if (hi_i < NW_TOP) lds_min(&sF[cur ^ 1][hi_i], f[r] + nhi);
                d.F[nb + j] = f[r];
            }
        }
        __syncthreads();
        cur ^= 1;
        nb = ne;
        ne = ne2;
    }
}

template <typename REAL, int R, int MODE>
__global__ void __launch_bounds__(64) k_bwd_narrow(DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    constexpr int W = 64 * R;
    __shared__ REAL sT[2][W];
    const int lane = threadIdx.x;
    const uint32_t p = block_to_pack(blockIdx.x, pk.n_packs);
    if (p >= pk.n_packs) return;
    if (d.delta_out) d.delta_out += (size_t)xcc_id() * d.delta_stride;
    const uint32_t q0 = pk.pack_hop_ptr[p], q1 = pk.pack_hop_ptr[p + 1];
    const int steps = pk.pack_steps[p];
    const REAL INF = inf_v<REAL>();
    int cur = 0;
    for (uint32_t q = q1; q-- > q0;) {
        const uint32_t nb = pk.hop_node_off[q], ne = pk.hop_node_off[q + 1];
        const uint32_t n = ne - nb;
        const uint32_t lbase = pk.hop_layer_off[q];
        uint32_t w[R];
        REAL f[R], lc[R], hc[R], d0[R], d1[R];
        int v[R];
        uint32_t lg[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            w[r] = (j < n) ? d.nwords[nb + j] : NW_PAD_WORD;
            f[r] = (MODE != BWD_PLAIN && j < n) ? d.F[nb + j] : REAL(0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool act = !(w[r] & NW_PAD);
            lg[r] = lbase + ((w[r] >> (2 * NW_CHILD_BITS)) & 1023u);
            lc[r] = act ? d.lo[lg[r]] : REAL(0);
            hc[r] = act ? d.hi[lg[r]] : REAL(0);
            v[r] = (MODE == BWD_SOLVE && act) ? d.var[lg[r]] : 0;
        }
        if (MODE == BWD_SOLVE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool act = !(w[r] & NW_PAD);
                d0[r] = act ? d.delta_in[2 * v[r]] : REAL(0);
                d1[r] = act ? d.delta_in[2 * v[r] + 1] : REAL(0);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            const bool act = !(w[r] & NW_PAD);
            const uint32_t lo_i = w[r] & NW_CHILD_MASK, hi_i = (w[r] >> NW_CHILD_BITS) & NW_CHILD_MASK;
            const REAL tl = lo_i == NW_BOT ? INF : (lo_i == NW_TOP ? REAL(0) : sT[cur][lo_i]);
            const REAL th = hi_i == NW_BOT ? INF : (hi_i == NW_TOP ? REAL(0) : sT[cur][hi_i]);
            REAL t;
            if (MODE == BWD_SOLVE) {
                REAL m0 = act ? (f[r] + lc[r]) + tl : INF;
                REAL m1 = act ? (f[r] + hc[r]) + th : INF;
                const unsigned long long heads = __ballot((w[r] & NW_HEAD) != 0);
                seg_min2(m0, m1, lane, heads, steps);
                const REAL mm = (rfinite(m0) && rfinite(m1)) ? omega * (m1 - m0) : REAL(0);
                const REAL nlo = (lc[r] + rmin(mm, REAL(0))) + d0[r];
                const REAL nhi = (hc[r] + rmin(-mm, REAL(0))) + d1[r];
                t = rmin(nhi + th, nlo + tl);
                if (act && (w[r] & NW_HEAD)) {
                    d.lo[lg[r]] = nlo;
                    d.hi[lg[r]] = nhi;
                    d.mm[lg[r]] = mm;
                    if (d.delta_out) accumulate_delta(d.delta_out, v[r], mm);
                }
            } else {
                const REAL ch = th + hc[r], cl = tl + lc[r];  // backward_step, bdd_cuda_base.cu:646-667
                t = rmin(ch, cl);
                if (MODE == BWD_MARGINALS) {
                    REAL lp = act ? f[r] + cl : INF;  // backward_step_with_path_costs, :633-641
                    REAL hp = act ? f[r] + ch : INF;
                    const unsigned long long heads = __ballot((w[r] & NW_HEAD) != 0);
                    seg_min2(lp, hp, lane, heads, steps);
                    if (act && (w[r] & NW_HEAD)) {
                        d.mm0_out[lg[r]] = lp;
                        d.mm1_out[lg[r]] = hp;
                    }
                }
            }
            if (act) {
                sT[cur ^ 1][j] = t;
                d.T[nb + j] = t;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251)
    const uint32_t n0 = pk.hop_node_off[q0 + 1] - pk.hop_node_off[q0];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        if (j < n0) s += (double)sT[cur][j];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = s;
}

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

template <typename REAL, int MODE>
__global__ void __launch_bounds__(WIDE_THREADS) k_fwd_wide(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WideLds<REAL> s = carve_lds<REAL>(smem, ww);
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x;
    if (p >= pk.n_packs) return;
    if (d.delta_out) d.delta_out += (size_t)xcc_id() * d.delta_stride;
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
            s.lc[l] = d.lo[lbase + l];
            s.hc[l] = d.hi[lbase + l];
        }
        __syncthreads();
        if (MODE == FWD_SOLVE) {
            for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
                const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
                const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
                const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
                const REAL f = Fc[j];
                const REAL tl = lo_i == WW_BOT ? INF : (lo_i == WW_TOP ? REAL(0) : s.t[lo_i]);
                const REAL th = hi_i == WW_BOT ? INF : (hi_i == WW_TOP ? REAL(0) : s.t[hi_i]);
                lds_min(&s.m0[l], (f + s.lc[l]) + tl);
                lds_min(&s.m1[l], (f + s.hc[l]) + th);
            }
            __syncthreads();
        }
        for (uint32_t j = tid; j < n; j += WIDE_THREADS) {
            const uint64_t w = d.wwords[nb + j - d.wide_slot_base];
            const uint32_t lo_i = (uint32_t)(w & WW_CHILD_MASK), hi_i = (uint32_t)((w >> WW_CHILD_BITS) & WW_CHILD_MASK);
            const uint32_t l = (uint32_t)((w >> (2 * WW_CHILD_BITS)) & WW_CHILD_MASK);
            const REAL f = Fc[j];
            REAL nlo = s.lc[l], nhi = s.hc[l];
            if (MODE == FWD_SOLVE) {
                const REAL m0 = s.m0[l], m1 = s.m1[l];
                const int v = d.var[lbase + l];
                const REAL mm = (rfinite(m0) && rfinite(m1)) ? omega * (m1 - m0) : REAL(0);
                nlo = (nlo + rmin(mm, REAL(0))) + d.delta_in[2 * v];
                nhi = (nhi + rmin(-mm, REAL(0))) + d.delta_in[2 * v + 1];
                if (w & WW_HEAD) {
                    d.lo[lbase + l] = nlo;
                    d.hi[lbase + l] = nhi;
                    d.mm[lbase + l] = mm;
                    if (d.delta_out) accumulate_delta(d.delta_out, v, mm);
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
            if (lo_i < WW_TOP) lds_min(&Fn[lo_i], f + nlo);
            if (hi_i < WW_TOP) lds_min(&Fn[hi_i], f + nhi);
            d.F[nb + j] = f;
        }
        __syncthreads();
        REAL* tmp = Fc; Fc = Fn; Fn = tmp;
        unsigned char* ta = Ac; Ac = An; An = ta;
        nb = ne;
        ne = ne2;
    }
}

template <typename REAL, int MODE>
__global__ void __launch_bounds__(WIDE_THREADS) k_bwd_wide(DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t ww)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WideLds<REAL> s = carve_lds<REAL>(smem, ww);
    __shared__ double red[WIDE_THREADS / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x;
    if (p >= pk.n_packs) return;
    if (d.delta_out) d.delta_out += (size_t)xcc_id() * d.delta_stride;
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
            s.lc[l] = d.lo[lbase + l];
            s.hc[l] = d.hi[lbase + l];
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
                    lds_min(&s.m0[l], (f + s.lc[l]) + tl);
                    lds_min(&s.m1[l], (f + s.hc[l]) + th);
                } else {
                    lds_min(&s.m0[l], f + (tl + s.lc[l]));
                    lds_min(&s.m1[l], f + (th + s.hc[l]));
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
                const REAL m0 = s.m0[l], m1 = s.m1[l];
                const int v = d.var[lbase + l];
                const REAL mm = (rfinite(m0) && rfinite(m1)) ? omega * (m1 - m0) : REAL(0);
                const REAL nlo = (s.lc[l] + rmin(mm, REAL(0))) + d.delta_in[2 * v];
                const REAL nhi = (s.hc[l] + rmin(-mm, REAL(0))) + d.delta_in[2 * v + 1];
                t = rmin(nhi + th, nlo + tl);
                if (w & WW_HEAD) {
                    d.lo[lbase + l] = nlo;
                    d.hi[lbase + l] = nhi;
                    d.mm[lbase + l] = mm;
                    if (d.delta_out) accumulate_delta(d.delta_out, v, mm);
                }
            } else {
                t = rmin(th + s.hc[l], tl + s.lc[l]);
                if (MODE == BWD_MARGINALS && (w & WW_HEAD)) {
                    d.mm0_out[lbase + l] = s.m0[l];
                    d.mm1_out[lbase + l] = s.m1[l];
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
// small elementwise / per-variable kernels
// =============================================================================================

// normalize_delta (bdd_cuda_parallel_mma.cu:410-430) fused with the zero-fill of compute_delta (:384)
// for the next pass: in[i] = out[i] / nr_bdds(i/2); out[i] = 0.
template <typename REAL>
__global__ void k_finish_delta(REAL* __restrict__ delta_in, REAL* __restrict__ delta_out,
                               const int32_t* __restrict__ nbdds, uint32_t n2, int n_slices)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    REAL s = 0;
    for (int x = 0; x < n_slices; ++x) {
        s += delta_out[(size_t)x * n2 + i];
        delta_out[(size_t)x * n2 + i] = REAL(0);
    }
    const int nb = nbdds[i >> 1];
    delta_in[i] = nb > 0 ? s / REAL(nb) : REAL(0);
}

// sum of the per-XCD slices without normalisation (explicit forward_mm / backward_mm API)
template <typename REAL>
__global__ void k_sum_slices(REAL* __restrict__ out, const REAL* __restrict__ slices, uint32_t n2, int n_slices)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    REAL s = 0;
    for (int x = 0; x < n_slices; ++x) s += slices[(size_t)x * n2 + i];
    out[i] = s;
}

template <typename REAL>
__global__ void k_normalize_delta(REAL* __restrict__ delta, const int32_t* __restrict__ nbdds, uint32_t n2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int nb = nbdds[i >> 1];
    if (nb > 0) delta[i] /= REAL(nb);
}

// Deterministic alternative to the atomics of compute_delta: per-variable gather over the
// (variable,bdd)-sorted layer list (the reduce_by_key variant commented out at :395-407).
template <typename REAL>
__global__ void k_delta_gather(const REAL* __restrict__ mm, const uint32_t* __restrict__ var_ptr,
                               const uint32_t* __restrict__ var_layers, REAL* __restrict__ delta_out, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    REAL lo = 0, hi = 0;
    for (uint32_t k = var_ptr[v]; k < var_ptr[v + 1]; ++k) {
        const REAL m = mm[var_layers[k]];
        if (m > 0) hi += m;
        else if (m < 0) lo += -m;
    }
    delta_out[2 * v] = lo;
    delta_out[2 * v + 1] = hi;
}

// set_vars_costs_func (bdd_cuda_base.cu:457-474).  The quotient is formed in double and rounded
// once to REAL, as the reference CPU solver does (bdd_parallel_mma_base.cpp:640,651).
template <typename REAL, typename TIN>
__global__ void k_update_costs(REAL* __restrict__ cost, const int32_t* __restrict__ var, const int32_t* __restrict__ nbdds,
                               const TIN* __restrict__ c, uint64_t n_c, uint32_t n_layers)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_layers) return;
    const int v = var[l];
    if ((uint64_t)v >= n_c) {
        cost[l] = REAL(0);  // :465-469
        return;
    }
    cost[l] += REAL((double)c[v] / (double)nbdds[v]);
}

template <typename REAL>
__global__ void k_set_cost(REAL* __restrict__ hi, const uint32_t* __restrict__ var_layers, uint32_t k0, uint32_t k1, REAL c)
{
    const uint32_t k = k0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k < k1) hi[var_layers[k]] += c;
}

// Deterministic fixed-shape reduction of the per-pack partial lower bounds.
static __global__ void k_lb_reduce(const double* __restrict__ part, uint32_t n, double* __restrict__ out)
{
    __shared__ double red[16];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += red[i];
        *out = t;
    }
}

template <typename REAL>
__global__ void k_lb_per_bdd(const REAL* __restrict__ T, const uint32_t* __restrict__ root_slot, REAL* __restrict__ out, uint32_t nb)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) out[b] = T[root_slot[b]];
}

// compute_net_costs_func (bdd_cuda_parallel_mma.cu:432-446)
template <typename REAL>
__global__ void k_net_costs(const REAL* __restrict__ lo, const REAL* __restrict__ hi, const REAL* __restrict__ mm,
                            REAL* __restrict__ out, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) out[l] = hi[l] - lo[l] + mm[l];
}

// distribute_deffered_mm_diff_func (bdd_cuda_base.cu:1396-1414) + the zero-fill of :1427
template <typename REAL>
__global__ void k_distribute_delta(REAL* __restrict__ lo, REAL* __restrict__ hi, REAL* __restrict__ mm, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const REAL m = mm[l];
    if (m > 0) hi[l] += m;
    else lo[l] -= m;
    mm[l] = REAL(0);
}

// add_scaled_product_func (bdd_cuda_parallel_mma.h:54-60)
template <typename REAL>
__global__ void k_gradient_step(REAL* __restrict__ hi, const REAL* __restrict__ g, REAL step, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) hi[l] = hi[l] + step * g[l];
}

// make_dual_feasible (bdd_cuda_base.cu:1261-1303): g[l] -= (sum over layers of var) / nr_bdds(var)
template <typename REAL>
__global__ void k_make_dual_feasible(REAL* __restrict__ g, const uint32_t* __restrict__ var_ptr,
                                     const uint32_t* __restrict__ var_layers, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    const uint32_t k0 = var_ptr[v], k1 = var_ptr[v + 1];
    if (k1 == k0) return;
    REAL s = 0;
    for (uint32_t k = k0; k < k1; ++k) s += g[var_layers[k]];
    const REAL q = s / REAL(k1 - k0);
    for (uint32_t k = k0; k < k1; ++k) g[var_layers[k]] -= q;
}

// compute_primal_objective_vec (bdd_cuda_base.cu:1352-1362)
template <typename REAL>
__global__ void k_primal_objective(const REAL* __restrict__ lo, const REAL* __restrict__ hi, const uint32_t* __restrict__ var_ptr,
                                   const uint32_t* __restrict__ var_layers, REAL* __restrict__ out, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    REAL s = 0;
    for (uint32_t k = var_ptr[v]; k < var_ptr[v + 1]; ++k) s += hi[var_layers[k]] - lo[var_layers[k]];
    out[v] = s;
}

template <typename T>
__global__ void k_gather(const T* __restrict__ in, const uint32_t* __restrict__ idx, T* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

static __global__ void k_gather_var(const int32_t* __restrict__ in, const uint32_t* __restrict__ idx, int32_t* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, T v, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// L-BFGS vector helpers (lbfgs_impl.h two-loop recursion; thrust::inner_product / transform there)
template <typename TA, typename TB>
__global__ void k_dot(const TA* __restrict__ a, const TB* __restrict__ b, double* __restrict__ partial, uint32_t n)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += (double)a[i] * (double)b[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}

// y[i] += alpha * x[i]   (x may be char-typed: the subgradient history, lbfgs.h:60)
template <typename REAL, typename TX>
__global__ void k_axpy(REAL* __restrict__ y, const TX* __restrict__ x, REAL alpha, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * REAL(x[i]);
}

template <typename REAL>
__global__ void k_scale(REAL* __restrict__ y, REAL alpha, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] *= alpha;
}

// out[i] = a[i] - b[i]
template <typename TO, typename TA>
__global__ void k_diff(TO* __restrict__ out, const TA* __restrict__ a, const TA* __restrict__ b, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = TO(a[i] - b[i]);
}

}  // namespace bddmma
