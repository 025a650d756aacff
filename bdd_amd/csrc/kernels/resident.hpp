// kernels/resident.hpp — narrow packs, resident sweeps (k_*_res: node words in LDS; k_*_res2: per-lane records).
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// narrow packs, resident sweeps: the whole pack is copied to LDS in one round trip, then swept out of LDS
// =============================================================================================
// The streaming kernels above walk a pack with a software pipeline that is two hops deep: enough when 5 waves per SIMD hide the
// rest of the memory latency (the 10 M-node benchmark), but a small or medium instance has ~1 wave per SIMD, and then every hop
// waits for memory it asked for two hops ago, after a start-up chain of 7-8 dependent round trips (pack tables, hop offsets, words,
// layer costs, staging tables, delta pairs).  Packs are short — tens of hops — so here a wave fetches EVERYTHING its pack needs at
// once: the pack's node words, opposite-direction potentials and arc costs are contiguous in memory, so they arrive as a few 1 KiB
// direct-to-LDS copies (global_load_lds_dwordx4, no staging registers) issued back to back from one 32-byte header; the hop loop
// then runs out of LDS with no loads at all.  Three dependent round trips per sweep (header; bulk copies + staging tables; delta
// pairs) instead of ~8 + one per two hops.  Same arithmetic, same order, same results as k_fwd_narrow / k_bwd_narrow.
struct ResDev {
    const uint32_t* pack_hdr;  // layout.hpp: struct Resident
    const uint32_t* quad_hdr;
    uint32_t ns;               // node slots reserved per wave in LDS (multiple of 256: whole 1 KiB pieces)
    uint32_t nl;               // layers reserved per wave in LDS (multiple of 128)
};
// What a resident sweep needs for its FIRST loads comes as leading plain kernel arguments: with -mllvm -amdgpu-kernarg-preload-count the
// command processor hands the first 16 dwords of plain (non-struct) arguments over in SGPRs at wave launch, so the header loads do not wait
// for the kernarg segment's own round trip (measured on the exchange, whose arguments are all plain: 4.4 -> 4.15 us at 1.05 M nodes).
// The stop word of the device-resident run_solver is among them: its load is issued at once and tested when the headers have arrived (no
// side effect happens before), instead of a dependent round trip in front of everything else.
#define RES_LEADING_ARGS const uint32_t* __restrict__ res_pack_hdr, const uint32_t* __restrict__ res_quad_hdr, uint32_t res_ns, uint32_t res_nl, \
                         uint32_t res_n_packs, uint32_t res_xcd_chunk, const uint32_t* res_stop, uint32_t res_run_iter
typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef __attribute__((address_space(1))) const void* glb_vptr_t;

// One wave copies `bytes` (rounded up to whole 1 KiB pieces) from global memory to LDS: lane l of piece k moves the 16 bytes at
// src + 1024 k + 16 l to dst + 1024 k + 16 l.  dst is wave-uniform and 16-byte aligned; the source only needs 4-byte alignment.
// Reads up to 1008 bytes past the range: device allocations are padded by 1 KiB (SolverT::dalloc).
__device__ __forceinline__ void wave_copy_to_lds(const void* src, void* dst, uint32_t bytes, int lane)
{
    const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + lane * 16;
    unsigned char* l = reinterpret_cast<unsigned char*>(dst);
    for (uint32_t o = 0; o < bytes; o += 1024)
        __builtin_amdgcn_global_load_lds((glb_vptr_t)(g + o), (lds_vptr_t)(l + o), 16, 0, 0);
}
__host__ __device__ inline uint32_t res_wave_bytes(uint32_t real_size, uint32_t ns, uint32_t nl)
{
    return ns * 4u + (ns + 4u) * real_size + nl * 2u * real_size;  // words | potentials + 2 sink entries (+2 pad) | {lo, hi}
}

template <typename REAL, int R, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_fwd_res(RES_LEADING_ARGS, DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    const ResDev rd{res_pack_hdr, res_quad_hdr, res_ns, res_nl};
    constexpr int W = 64 * R;
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ REAL sF_[WPB][2][W + 2];  // frontier: cost from root of the current / next hop; [W], [W + 1]: dummy push targets of sink children
    __shared__ uint32_t sOffN_[WPB][64], sOffL_[WPB][64];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    // per-wave scratch of the segmented minimum (seg_min2): behind the rest of the dynamic LDS, only reserved when a pack has layers
    // wider than two nodes (pk.seg_off; never dereferenced otherwise)
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    auto& sF = sF_[wave];
    uint32_t* sOffN = sOffN_[wave];
    uint32_t* sOffL = sOffL_[wave];
    const uint32_t n_quads = (res_n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(blockIdx.x, n_quads, res_xcd_chunk);
    if (quad >= n_quads) return;
    const uint32_t stop_word = res_stop != nullptr ? *res_stop : RUN_NOT_STOPPED;  // tested below, with the headers
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < res_n_packs;
    BDDMMA_STAMP(p, 0);
    // dynamic LDS: [staged {delta_lo, delta_hi} / mm: WPB * stage_cap pairs][per wave: words | T of every slot | {lo, hi} of every layer]
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    P2* sDw = sD + (size_t)wave * pk.stage_cap;
    const uint32_t wave_off = WPB * pk.stage_cap * (uint32_t)sizeof(P2) + (uint32_t)wave * res_wave_bytes(sizeof(REAL), rd.ns, rd.nl);
    uint32_t* sW = reinterpret_cast<uint32_t*>(dyn_lds + wave_off);
    REAL* sTa = reinterpret_cast<REAL*>(dyn_lds + wave_off + rd.ns * 4u);
    P2* sC = reinterpret_cast<P2*>(dyn_lds + wave_off + rd.ns * 4u + (rd.ns + 4u) * (uint32_t)sizeof(REAL));
    // ---- round trip 1: the headers
    const uint32_t* hp = rd.pack_hdr + 8 * (size_t)(has_pack ? p : 0);
    const uint32_t slot0 = hp[0], layer0 = hp[2], q0 = hp[4], woff = hp[6];
    const uint32_t nslots = has_pack ? hp[1] : 0, nlayers = has_pack ? hp[3] : 0, nh = has_pack ? (hp[5] & 0xFFFFu) : 0;
    const int steps = (int)(hp[5] >> 16);
    const uint32_t c0 = rd.quad_hdr[4 * (size_t)quad], cnt = rd.quad_hdr[4 * (size_t)quad + 1];
    if (stop_word <= res_run_iter) return;  // run_solver has stopped (uniform for the grid): nothing has been written yet
    const REAL INF = inf_v<REAL>();
    const NarrowRs<REAL> rs(d);
    BDDMMA_STAMP(p, 1);
    // ---- round trip 2: the whole pack -> LDS, hop offsets, staging tables; round trip 3 (inside stage_load): the delta pairs
    wave_copy_to_lds(d.nwords + woff, sW, nslots * 4u, lane);
    wave_copy_to_lds(d.T + slot0, sTa, nslots * (uint32_t)sizeof(REAL), lane);
    wave_copy_to_lds(d.lohi + 2 * (size_t)layer0, sC, nlayers * (uint32_t)sizeof(P2), lane);
    {
        const uint32_t q = q0 + min((uint32_t)lane, nh);
        const uint32_t on = pk.hop_node_off[q], ol = pk.hop_layer_off[q];
        sOffN[lane] = on - slot0;
        sOffL[lane] = ol - layer0;
    }
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    stage_load<REAL, WPB>(sD, ent, esl, rs, c0, cnt, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the direct-to-LDS copies have landed
    if (lane < 2) sTa[rd.ns + lane] = lane == 0 ? REAL(0) : INF;  // sink entries: cost to terminal 0 (top) / +inf (bot)
    wave_sync();
    uint32_t nb = 0, ne = __builtin_amdgcn_readfirstlane(sOffN[1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        sF[0][j] = (j < ne) ? REAL(0) : INF;  // every slot of hop 0 is a root (flush_costs_from_root)
    }
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    int cur = 0;
    for (uint32_t h = 0; h < nh; ++h) {
        const uint32_t ne2 = __builtin_amdgcn_readfirstlane(sOffN[min(h + 2, nh)]);
        const uint32_t n = ne - nb;
        uint32_t lb = __builtin_amdgcn_readfirstlane(sOffL[h]);
        REAL f[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            sF[cur ^ 1][j] = INF;
            f[r] = sF[cur][j];
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            constexpr uint32_t PADW = nw_pad_word(W);
            const uint32_t w = j < n ? sW[nb + j] : PADW;
            const bool act = !(w & NW_PAD);
            const uint32_t lo_i = w & NW_CHILD_MASK, hi_i = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
            // layer index inside the pack = layers of the hops and lane groups before + the word's index inside its lane group
            const uint32_t ll = act ? lb + nw_lidx(w) : 0u;
            lb += (uint32_t)__popcll(__ballot(nw_head(w)));
            const P2 c = sC[ll];
            const REAL tl = sTa[lo_i < (uint32_t)W ? ne + lo_i : rd.ns + (lo_i - W)];
            const REAL th = sTa[hi_i < (uint32_t)W ? ne + hi_i : rd.ns + (hi_i - W)];
            const P2 dd = sDw[ll];
            REAL m0 = act ? (f[r] + c.x) + tl : INF;
            REAL m1 = act ? (f[r] + c.y) + th : INF;
            seg_min2(m0, m1, lane, nw_pos(w), nw_len(w), steps, sM);
            const REAL mm = mm_diff(m0, m1, omega);
            const REAL nlo = (c.x + min0(mm)) + dd.x;
            const REAL nhi = (c.y + min0_neg(mm)) + dd.y;
            const bool head = nw_head(w);
            P2 nc;
            nc.x = nlo;
            nc.y = nhi;
            bstore(nc, rs.lohi, head ? (layer0 + ll) * (uint32_t)sizeof(P2) : OOB);
            if (head) sDw[ll].x = mm;  // every lane of the layer has read its pair above (same wave, in order)
            const bool plo = lo_i < (uint32_t)W, phi = hi_i < (uint32_t)W;  // sink children and padding lanes: no-op on the own slot (see k_fwd_narrow)
            lds_min(&sF[cur ^ 1][plo ? lo_i : j], plo ? f[r] + nlo : INF);
            lds_min(&sF[cur ^ 1][phi ? hi_i : j], phi ? f[r] + nhi : INF);
            bstore(f[r], rs.F, act ? (slot0 + nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        wave_sync();
        cur ^= 1;
        nb = ne;
        ne = ne2;
    }
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);  // min-marginal differences -> entry array
    BDDMMA_STAMP(p, 4);
}

template <typename REAL, int R, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_bwd_res(RES_LEADING_ARGS, DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    const ResDev rd{res_pack_hdr, res_quad_hdr, res_ns, res_nl};
    constexpr int W = 64 * R;
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ REAL sT_[WPB][2][W + 2];  // cost to terminal of the hop above / of this hop; [W] = 0 (top sink), [W + 1] = +inf (bot sink)
    __shared__ uint32_t sOffN_[WPB][64], sOffL_[WPB][64];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    // per-wave scratch of the segmented minimum (seg_min2): behind the rest of the dynamic LDS, only reserved when a pack has layers
    // wider than two nodes (pk.seg_off; never dereferenced otherwise)
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    auto& sT = sT_[wave];
    uint32_t* sOffN = sOffN_[wave];
    uint32_t* sOffL = sOffL_[wave];
    const uint32_t n_quads = (res_n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(blockIdx.x, n_quads, res_xcd_chunk);
    if (quad >= n_quads) return;
    const uint32_t stop_word = res_stop != nullptr ? *res_stop : RUN_NOT_STOPPED;  // tested below, with the headers
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < res_n_packs;
    BDDMMA_STAMP(p, 0);
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    P2* sDw = sD + (size_t)wave * pk.stage_cap;
    const uint32_t wave_off = WPB * pk.stage_cap * (uint32_t)sizeof(P2) + (uint32_t)wave * res_wave_bytes(sizeof(REAL), rd.ns, rd.nl);
    uint32_t* sW = reinterpret_cast<uint32_t*>(dyn_lds + wave_off);
    REAL* sFa = reinterpret_cast<REAL*>(dyn_lds + wave_off + rd.ns * 4u);  // cost from root of every slot (forward sweep)
    P2* sC = reinterpret_cast<P2*>(dyn_lds + wave_off + rd.ns * 4u + (rd.ns + 4u) * (uint32_t)sizeof(REAL));
    const uint32_t* hp = rd.pack_hdr + 8 * (size_t)(has_pack ? p : 0);
    const uint32_t slot0 = hp[0], layer0 = hp[2], q0 = hp[4], woff = hp[6];
    const uint32_t nslots = has_pack ? hp[1] : 0, nlayers = has_pack ? hp[3] : 0, nh = has_pack ? (hp[5] & 0xFFFFu) : 0;
    const int steps = (int)(hp[5] >> 16);
    const uint32_t c0 = rd.quad_hdr[4 * (size_t)quad], cnt = rd.quad_hdr[4 * (size_t)quad + 1];
    if (stop_word <= res_run_iter) return;  // run_solver has stopped (uniform for the grid): nothing has been written yet
    const REAL INF = inf_v<REAL>();
    const NarrowRs<REAL> rs(d);
    BDDMMA_STAMP(p, 1);
    wave_copy_to_lds(d.nwords + woff, sW, nslots * 4u, lane);
    wave_copy_to_lds(d.F + slot0, sFa, nslots * (uint32_t)sizeof(REAL), lane);
    wave_copy_to_lds(d.lohi + 2 * (size_t)layer0, sC, nlayers * (uint32_t)sizeof(P2), lane);
    {
        const uint32_t q = q0 + min((uint32_t)lane, nh);
        const uint32_t on = pk.hop_node_off[q], ol = pk.hop_layer_off[q];
        sOffN[lane] = on - slot0;
        sOffL[lane] = ol - layer0;
    }
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    stage_load<REAL, WPB>(sD, ent, esl, rs, c0, cnt, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < 4) sT[lane >> 1][W + (lane & 1)] = (lane & 1) ? INF : REAL(0);
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    int cur = 0;
    for (uint32_t h = nh; h-- > 0;) {
        const uint32_t nb = __builtin_amdgcn_readfirstlane(sOffN[h]), ne = __builtin_amdgcn_readfirstlane(sOffN[h + 1]);
        const uint32_t n = ne - nb;
        uint32_t lb = __builtin_amdgcn_readfirstlane(sOffL[h]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            constexpr uint32_t PADW = nw_pad_word(W);
            const uint32_t w = j < n ? sW[nb + j] : PADW;
            const bool act = !(w & NW_PAD);
            const uint32_t lo_i = w & NW_CHILD_MASK, hi_i = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
            const uint32_t ll = act ? lb + nw_lidx(w) : 0u;
            lb += (uint32_t)__popcll(__ballot(nw_head(w)));
            const P2 c = sC[ll];
            const REAL fa = sFa[act ? nb + j : 0];
            const REAL tl = sT[cur][lo_i];  // sinks: [W] = 0, [W + 1] = +inf
            const REAL th = sT[cur][hi_i];
            const P2 dd = sDw[ll];
            REAL m0 = act ? (fa + c.x) + tl : INF;
            REAL m1 = act ? (fa + c.y) + th : INF;
            seg_min2(m0, m1, lane, nw_pos(w), nw_len(w), steps, sM);
            const REAL mm = mm_diff(m0, m1, omega);
            const REAL nlo = (c.x + min0(mm)) + dd.x;
            const REAL nhi = (c.y + min0_neg(mm)) + dd.y;
            const REAL t = rmin(nhi + th, nlo + tl);
            const bool head = nw_head(w);
            P2 nc;
            nc.x = nlo;
            nc.y = nhi;
            bstore(nc, rs.lohi, head ? (layer0 + ll) * (uint32_t)sizeof(P2) : OOB);
            if (head) sDw[ll] = P2{mm, nhi - nlo};  // .y: hi' - lo' for x_layer
            if (act) sT[cur ^ 1][j] = t;
            bstore(t, rs.T, act ? (slot0 + nb + j) * (uint32_t)sizeof(REAL) : OOB);
        }
        wave_sync();
        cur ^= 1;
    }
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
    BDDMMA_STAMP(p, 4);
    if (!has_pack) return;
    if (d.x_layer != nullptr)
        for (uint32_t j = lane; j < nlayers; j += 64) d.x_layer[layer0 + j] = sDw[j].y + sDw[j].x;  // (hi' - lo') + mm
    // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251)
    const uint32_t n0 = __builtin_amdgcn_readfirstlane(sOffN[1]);
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        if (j < n0) sum += (double)sT[cur][j];
    }
    for (int off2 = 32; off2 > 0; off2 >>= 1) sum += __shfl_down(sum, off2);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = sum;
}

// =============================================================================================
// narrow packs, resident sweeps, second generation: ready-made LDS addresses per lane and hop
// =============================================================================================
// k_fwd_res / k_bwd_res removed the loads from the hop loop; what is left there is ~165 instructions per hop of 64 slots of which ~25 are
// floating point — unpacking the node word, hop-local index -> LDS address, sink selects, per-hop offsets through LDS -> SGPR
// (profiles/r04_hop_isa.txt), on a wave that is alone on its SIMD most of the time, so every instruction's latency is the hop's.
// Here a lane's hop is one 16-byte record (layout.hpp: Res2Records) of ready-made 16-bit byte offsets into the wave's LDS region:
// children's costs-from-terminal, push targets, the layer's cost / staging pair, the node's own slot, the store offset of the new arc
// costs (an out-of-range offset for lanes that are not their layer's head).  Hops are dense (64 records per hop), so there are no
// per-hop offsets at all; sink children and padding lanes need no select (constant entries / a private dummy entry per lane); all of a
// pack's potentials sit in LDS by slot (no double-buffered frontier to reset); a two-node layer's minimum is one DPP swap of
// neighbouring lanes (layers of two nodes start at even lanes, layout.cpp: PackBuilder::place).  Records are prefetched four hops
// ahead (shared by all packs of a structure template: L2 hits).  Same arithmetic, same order, same results as k_fwd_res / k_bwd_res.
// Packs of 64 slots whose layers have <= 2 nodes (SolverT::use_res2); everything else runs the first generation.
using u4v = decltype(__builtin_amdgcn_raw_buffer_load_b128(*static_cast<const rsrc_t*>(nullptr), 0, 0, 0));
template <typename T>
__device__ __forceinline__ T lds_ld(const unsigned char* lds, uint32_t off) { return *reinterpret_cast<const T*>(lds + off); }
template <typename T>
__device__ __forceinline__ void lds_st(unsigned char* lds, uint32_t off, T v) { *reinterpret_cast<T*>(lds + off) = v; }

// minimum over the two lanes of an aligned pair where `two` holds; every lane of the wave executes it (a DPP source lane that EXEC
// masks out would count as invalid, see seg_pair_min)
__device__ __forceinline__ void pair_min_aligned(float& a, float& b, bool two)
{
    float ta, tb;
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(ta), "=&v"(tb)
                 : "v"(a), "v"(b));
    a = two ? ta : a;
    b = two ? tb : b;
}
__device__ __forceinline__ void pair_min_aligned(double& a, double& b, bool two)
{
    const double a2 = dpp_row<0xB1>(a), b2 = dpp_row<0xB1>(b);  // quad_perm [1, 0, 3, 2]
    a = two ? rmin(a, a2) : a;
    b = two ? rmin(b, b2) : b;
}
// mm = omega * (m1 - m0), or 0 unless both minima are finite (bdd_cuda_parallel_mma.cu:36-39): the minima are never -inf or NaN (sums of
// finite costs and +inf), so "both finite" is "their difference is finite" — one class test instead of two
template <typename REAL>
__device__ __forceinline__ REAL mm_diff1(REAL m0, REAL m1, REAL omega)
{
    const REAL dm = m1 - m0;
    return rfinite(dm) ? omega * dm : REAL(0);
}

#define RES2_ARGS RES_LEADING_ARGS, const uint32_t* __restrict__ res2_rec, const uint32_t* __restrict__ res2_rec_off, uint32_t res2_n_words

template <typename REAL, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_fwd_res2(RES2_ARGS, DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    constexpr uint32_t S = sizeof(REAL);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    const uint32_t n_quads = (res_n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(blockIdx.x, n_quads, res_xcd_chunk);
    if (quad >= n_quads) return;
    const uint32_t stop_word = res_stop != nullptr ? *res_stop : RUN_NOT_STOPPED;
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < res_n_packs;
    BDDMMA_STAMP(p, 0);
    // dynamic LDS: [staged {delta_lo, delta_hi} / mm: WPB * stage_cap pairs][per wave: T | F | {lo, hi}] (layout.hpp: res2_*_off)
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);
    const uint32_t wb = WPB * pk.stage_cap * (uint32_t)sizeof(P2) + (uint32_t)wave * res2_wave_bytes(S, res_ns, res_nl);
    const uint32_t wbF = wb + res2_f_off(S, res_ns), wbC = wb + res2_c_off(S, res_ns);
    // ---- round trip 1: the headers
    const uint32_t* hp = res_pack_hdr + 8 * (size_t)(has_pack ? p : 0);
    const uint32_t slot0 = hp[0], layer0 = hp[2];
    const uint32_t nslots = has_pack ? hp[1] : 0, nlayers = has_pack ? hp[3] : 0, nh = has_pack ? (hp[5] & 0xFFFFu) : 0;
    const uint32_t rbase = res2_rec_off[has_pack ? p : 0];
    const uint32_t c0 = res_quad_hdr[4 * (size_t)quad], cnt = res_quad_hdr[4 * (size_t)quad + 1];
    if (stop_word <= res_run_iter) return;  // run_solver has stopped (uniform for the grid): nothing has been written yet
    const REAL INF = inf_v<REAL>();
    const NarrowRs<REAL> rs(d);
    BDDMMA_STAMP(p, 1);
    // ---- round trip 2: costs-from-terminal and arc costs of the pack -> LDS, the first records, the staging tables; 3: the delta pairs
    wave_copy_to_lds(d.T + slot0, dyn_lds + wb, nslots * S, lane);
    wave_copy_to_lds(d.lohi + 2 * (size_t)layer0, dyn_lds + wbC, nlayers * (uint32_t)sizeof(P2), lane);
    const rsrc_t rr = make_rsrc(res2_rec, res2_n_words);
    auto ldrec = [&](uint32_t h) { return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + h * 64u) * 16u, 0); };
    u4v r0 = ldrec(0), r1 = ldrec(1), r2 = ldrec(2), r3 = ldrec(3), r4 = ldrec(4), r5 = ldrec(5), r6 = ldrec(6), r7 = ldrec(7);
    if (has_pack)
        for (uint32_t o = (uint32_t)lane; o < res_ns + 64u; o += 64u) lds_st<REAL>(dyn_lds, wbF + o * S, INF);  // costs-from-root and the dummy entries
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    stage_load<REAL, WPB>(sD, ent, esl, rs, c0, cnt, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the direct-to-LDS copies have landed
    if (has_pack) {
        if (lane < 2) lds_st<REAL>(dyn_lds, wb + (res_ns + (uint32_t)lane) * S, lane == 0 ? REAL(0) : INF);  // sinks: cost to terminal 0 (top) / +inf (bot)
        if (r0[3] != RES2_PAD) lds_st<REAL>(dyn_lds, wbF + (r0[2] >> 16), REAL(0));  // every node of hop 0 is a root (flush_costs_from_root)
    }
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    // the pack's slices of the arrays the hop loop stores into: offsets past their ends are dropped
    const rsrc_t rF = make_rsrc(d.F + slot0, nslots), rC = make_rsrc(d.lohi + 2 * (size_t)layer0, 2ull * nlayers);
    auto hop = [&](const u4v& r) {
        const bool real = r[3] != RES2_PAD;
        const bool two = (r[3] & 0x10000u) != 0;
        const uint32_t ll = r[2] & 0xFFFFu, fs = r[2] >> 16;
        const REAL f = lds_ld<REAL>(dyn_lds, wbF + fs);
        const REAL tl = lds_ld<REAL>(dyn_lds, wb + (r[0] & 0xFFFFu)), th = lds_ld<REAL>(dyn_lds, wb + (r[0] >> 16));
        const P2 c = lds_ld<P2>(dyn_lds, wbC + ll);
        const P2 dd = lds_ld<P2>(dyn_lds, db + ll);
        REAL m0 = (f + c.x) + tl, m1 = (f + c.y) + th;  // padding lanes: +inf
        pair_min_aligned(m0, m1, two);
        const REAL mm = mm_diff1(m0, m1, omega);
        P2 nc;
        nc.x = (c.x + min0(mm)) + dd.x;
        nc.y = (c.y + min0_neg(mm)) + dd.y;
        bstore(nc, rC, r[3] & 0xFFFFu);                        // heads only: RES2_NO_STORE lies past the pack's layers
        if (real) lds_st<REAL>(dyn_lds, db + ll, mm);          // every lane of a layer holds the same value
        lds_min(reinterpret_cast<REAL*>(dyn_lds + wb + (r[1] & 0xFFFFu)), f + nc.x);  // sinks / padding: the lane's own dummy entry
        lds_min(reinterpret_cast<REAL*>(dyn_lds + wb + (r[1] >> 16)), f + nc.y);
        bstore(f, rF, fs);                                      // padding lanes: past the pack's slots
        wave_sync();
    };
    // Sixteen hops in a straight line with side exits: inside it the compiler counts the outstanding record loads and stores (s_waitcnt
    // vmcnt(N)); at a loop header it drains them all, which would expose the L2 latency of the newest prefetch on every trip of a 4-hop loop.
    // (records: a ring of eight, i.e. requested eight hops = several L2 round trips ahead; packs of <= 8 hops have them all before the loop)
#define RES2_HOP(RK, HK)            \
    hop(RK);                        \
    RK = ldrec(h + (HK) + 8);       \
    if (h + (HK) + 1 >= nh) break;
    for (uint32_t h = 0; h < nh; h += 16) {
        RES2_HOP(r0, 0) RES2_HOP(r1, 1) RES2_HOP(r2, 2) RES2_HOP(r3, 3) RES2_HOP(r4, 4) RES2_HOP(r5, 5) RES2_HOP(r6, 6) RES2_HOP(r7, 7)
        RES2_HOP(r0, 8) RES2_HOP(r1, 9) RES2_HOP(r2, 10) RES2_HOP(r3, 11) RES2_HOP(r4, 12) RES2_HOP(r5, 13) RES2_HOP(r6, 14) RES2_HOP(r7, 15)
    }
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);  // min-marginal differences -> entry array
    BDDMMA_STAMP(p, 4);
}

template <typename REAL, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_bwd_res2(RES2_ARGS, DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    constexpr uint32_t S = sizeof(REAL);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    const uint32_t n_quads = (res_n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(blockIdx.x, n_quads, res_xcd_chunk);
    if (quad >= n_quads) return;
    const uint32_t stop_word = res_stop != nullptr ? *res_stop : RUN_NOT_STOPPED;
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < res_n_packs;
    BDDMMA_STAMP(p, 0);
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    P2* sDw = sD + (size_t)wave * pk.stage_cap;
    const uint32_t db = (uint32_t)wave * pk.stage_cap * (uint32_t)sizeof(P2);
    const uint32_t wb = WPB * pk.stage_cap * (uint32_t)sizeof(P2) + (uint32_t)wave * res2_wave_bytes(S, res_ns, res_nl);
    const uint32_t wbF = wb + res2_f_off(S, res_ns), wbC = wb + res2_c_off(S, res_ns);
    const uint32_t* hp = res_pack_hdr + 8 * (size_t)(has_pack ? p : 0);
    const uint32_t slot0 = hp[0], layer0 = hp[2];
    const uint32_t nslots = has_pack ? hp[1] : 0, nlayers = has_pack ? hp[3] : 0, nh = has_pack ? (hp[5] & 0xFFFFu) : 0;
    const uint32_t rbase = res2_rec_off[has_pack ? p : 0];
    const uint32_t c0 = res_quad_hdr[4 * (size_t)quad], cnt = res_quad_hdr[4 * (size_t)quad + 1];
    if (stop_word <= res_run_iter) return;
    const REAL INF = inf_v<REAL>();
    const NarrowRs<REAL> rs(d);
    BDDMMA_STAMP(p, 1);
    wave_copy_to_lds(d.F + slot0, dyn_lds + wbF, nslots * S, lane);  // costs from root of every slot (forward sweep)
    wave_copy_to_lds(d.lohi + 2 * (size_t)layer0, dyn_lds + wbC, nlayers * (uint32_t)sizeof(P2), lane);
    const rsrc_t rr = make_rsrc(res2_rec, res2_n_words);
    // k-th hop processed = hop nh - 1 - k of the pack; past the first hop: any record (never used)
    auto ldrec = [&](uint32_t k) { return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + (k < nh ? nh - 1u - k : 0u) * 64u) * 16u, 0); };
    u4v r0 = ldrec(0), r1 = ldrec(1), r2 = ldrec(2), r3 = ldrec(3), r4 = ldrec(4), r5 = ldrec(5), r6 = ldrec(6), r7 = ldrec(7);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    stage_load<REAL, WPB>(sD, ent, esl, rs, c0, cnt, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (has_pack && lane < 2) lds_st<REAL>(dyn_lds, wb + (res_ns + (uint32_t)lane) * S, lane == 0 ? REAL(0) : INF);
    if (WPB > 1) __syncthreads(); else wave_sync();
    BDDMMA_STAMP(p, 2);
    const rsrc_t rT = make_rsrc(d.T + slot0, nslots), rC = make_rsrc(d.lohi + 2 * (size_t)layer0, 2ull * nlayers);
    auto hop = [&](const u4v& r) {
        const bool real = r[3] != RES2_PAD;
        const bool two = (r[3] & 0x10000u) != 0;
        const uint32_t ll = r[2] & 0xFFFFu, fs = r[2] >> 16;
        const REAL f = lds_ld<REAL>(dyn_lds, wbF + fs);  // padding lanes: whatever the dummy entry holds; their results go nowhere
        const REAL tl = lds_ld<REAL>(dyn_lds, wb + (r[0] & 0xFFFFu)), th = lds_ld<REAL>(dyn_lds, wb + (r[0] >> 16));
        const P2 c = lds_ld<P2>(dyn_lds, wbC + ll);
        const P2 dd = lds_ld<P2>(dyn_lds, db + ll);
        REAL m0 = (f + c.x) + tl, m1 = (f + c.y) + th;
        pair_min_aligned(m0, m1, two);
        const REAL mm = mm_diff1(m0, m1, omega);
        P2 nc;
        nc.x = (c.x + min0(mm)) + dd.x;
        nc.y = (c.y + min0_neg(mm)) + dd.y;
        const REAL t = rmin(nc.y + th, nc.x + tl);
        bstore(nc, rC, r[3] & 0xFFFFu);
        if (real) {
            lds_st<P2>(dyn_lds, db + ll, P2{mm, nc.y - nc.x});  // .y: hi' - lo' for x_layer
            lds_st<REAL>(dyn_lds, wb + fs, t);
        }
        bstore(t, rT, fs);
        wave_sync();
    };
    for (uint32_t h = 0; h < nh; h += 16) {  // see k_fwd_res2; h counts the hops processed, from the pack's last hop upwards
        RES2_HOP(r0, 0) RES2_HOP(r1, 1) RES2_HOP(r2, 2) RES2_HOP(r3, 3) RES2_HOP(r4, 4) RES2_HOP(r5, 5) RES2_HOP(r6, 6) RES2_HOP(r7, 7)
        RES2_HOP(r0, 8) RES2_HOP(r1, 9) RES2_HOP(r2, 10) RES2_HOP(r3, 11) RES2_HOP(r4, 12) RES2_HOP(r5, 13) RES2_HOP(r6, 14) RES2_HOP(r7, 15)
    }
#undef RES2_HOP
    const u4v rroot = __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, rbase * 16u, 0);  // the pack's first hop again, for the bound
    BDDMMA_STAMP(p, 3);
    if (WPB > 1) __syncthreads(); else wave_sync();
    stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
    BDDMMA_STAMP(p, 4);
    if (!has_pack) return;
    if (d.x_layer != nullptr)
        for (uint32_t j = lane; j < nlayers; j += 64) d.x_layer[layer0 + j] = sDw[j].y + sDw[j].x;  // (hi' - lo') + mm
    // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251); every node of the first hop is a root
    double lb = rroot[3] != RES2_PAD ? (double)lds_ld<REAL>(dyn_lds, wb + (rroot[2] >> 16)) : 0.0;
    for (int off2 = 32; off2 > 0; off2 >>= 1) lb += __shfl_down(lb, off2);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = lb;
}

}  // namespace bddmma
