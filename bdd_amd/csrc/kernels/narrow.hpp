// kernels/narrow.hpp — narrow packs, streaming sweeps, first generation (k_fwd_narrow / k_bwd_narrow): node words unpacked per hop; every mode.
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// narrow packs: one wavefront per pack, R groups of 64 slots per hop, no barriers
// =============================================================================================
// Window of per-hop offsets kept in LDS.  Reading pk.hop_node_off[q] inside the hop loop compiles to a
// *vector* global load followed by s_waitcnt vmcnt(0): it serialises two extra memory round trips per
// hop and drains every prefetch in flight.  Instead 64 consecutive offsets are fetched with one
// coalesced load and read back with (broadcast) LDS reads, which are counted by lgkmcnt only.
// Ordering point for LDS traffic of ONE wave.  The LDS unit executes a wave's DS instructions in order,
// so a wave that only consumes what it wrote itself needs no hardware barrier — only the compiler must not
// move LDS accesses across this point.  (In a one-wave workgroup __syncthreads() compiles to the same;
// with several waves per workgroup it would be a real s_barrier per hop.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t HOP_WIN = 64;
#ifndef BDDMMA_HOP_UNROLL
#define BDDMMA_HOP_UNROLL 2
#endif
constexpr int HOP_UNROLL = BDDMMA_HOP_UNROLL;  // hops per trip of the narrow kernels' hop loops (see k_fwd_narrow)
struct HopWindow {
    uint32_t* node;   // LDS [HOP_WIN]
    uint32_t* layer;  // LDS [HOP_WIN]
    uint32_t* root;   // LDS [HOP_WIN]: PackDev::hop_root of the record (NO_ROOT past the pack's last hop)
    uint32_t base;    // record index of window slot 0
    uint32_t q1;      // one past the pack's last hop record (offsets clamp there)
    // Offsets are kept RELATIVE to the pack's first slot / layer (round 4): the sweeps address F, T and {lo, hi} through pointers
    // rebased to the pack (64-bit, once per pack), so that the 32-bit byte offsets of the buffer instructions stay small whatever the
    // arrays' size — arrays beyond 4 GiB (>= 512 M slots in double) no longer overflow them.
    uint32_t n0, l0;  // the pack's first slot and first layer
    __device__ __forceinline__ void fill(const PackDev& pk, uint32_t new_base, int lane)
    {
        base = new_base;
        const uint32_t q = min(new_base + (uint32_t)lane, q1);
        node[lane] = pk.hop_node_off[q] - n0;
        layer[lane] = pk.hop_layer_off[q] - l0;
        root[lane] = new_base + (uint32_t)lane < q1 ? (uint32_t)pk.hop_root[q] : (uint32_t)NO_ROOT;
        wave_sync();
    }
    // root slot of hop q (below the pack's first hop), NO_ROOT if no BDD starts there
    __device__ __forceinline__ uint32_t root_of(uint32_t q) const
    {
        return __builtin_amdgcn_readfirstlane(root[min(q, q1) - base]);
    }
    __device__ __forceinline__ uint32_t node_off(uint32_t q) const
    {
        return __builtin_amdgcn_readfirstlane(node[min(q, q1) - base]);
    }
    __device__ __forceinline__ uint32_t layer_off(uint32_t q) const
    {
        return __builtin_amdgcn_readfirstlane(layer[min(q, q1) - base]);
    }
};

// Per-hop register sets of the software pipeline.  A wave's hop is a chain of dependent memory round
// trips (node words -> layer costs -> LDS), and with <= 32 waves per CU the sweep was latency-bound
// (SQ_WAIT_ANY 80 % of wave cycles, 3.9 TB/s).  All addresses of later hops are plain streams, so the
// words of hop q+2 and the layer data / potentials of hop q+1 are requested while hop q is computed.
template <typename REAL, int R>
struct HopLayer {
    typename Pair<REAL>::type c[R];  // {lo, hi}
    uint32_t lg[R];                  // global layer index of the lane's node
};

__device__ __forceinline__ uint32_t nw_pos(uint32_t w) { return (w >> NW_POS_SHIFT) & NW_FIELD6; }
__device__ __forceinline__ uint32_t nw_lidx(uint32_t w) { return (w >> NW_LIDX_SHIFT) & NW_FIELD6; }  // layer index inside the lane group
__device__ __forceinline__ uint32_t nw_len(uint32_t w) { return (w & NW_TWO) ? 2u : 0u; }  // only "is it a two-node layer" is stored
__device__ __forceinline__ bool nw_head(uint32_t w) { return (w & (NW_PAD | (NW_FIELD6 << NW_POS_SHIFT))) == 0; }

__device__ __forceinline__ void hop_load(float2& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rh, voff, soff, BDDMMA_LD_AUX));
}
__device__ __forceinline__ void hop_load(double2& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rh, voff, soff, 0));
}
template <typename REAL, int R>
__device__ __forceinline__ void load_layer(HopLayer<REAL, R>& L, const uint32_t (&w)[R], uint32_t lbase, const NarrowRs<REAL>& rs)
{
    using P2 = typename Pair<REAL>::type;
    uint32_t base = lbase;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t li = nw_lidx(w[r]);
        L.lg[r] = base + li;
        // the group's first layer goes into the scalar offset; padding lanes (index 0) read that layer's pair and ignore it
        hop_load(L.c[r], rs.lohi, li * (uint32_t)sizeof(P2), base * (uint32_t)sizeof(P2));
        if (r + 1 < R) base += (uint32_t)__popcll(__ballot(nw_head(w[r])));  // layers of this lane group
    }
}

// One hop's slice [nb, nb + n) of a slot-indexed array.  A lane addresses it with its constant byte offset j * sizeof(T); the slice's
// start goes into the scalar offset of the buffer instruction and the descriptor ends where the slice ends, so the lanes past the
// hop's last slot drop out by themselves: no per-lane address arithmetic or select in the hop (3 VALU per access before).  (gfx950
// range-checks voffset + soffset against num_records — measured: with num_records = the slice's length every lane was dropped.)
template <typename T>
__device__ __forceinline__ rsrc_t hop_rsrc(const T* base, uint32_t nb, uint32_t n)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (nb + n) * (uint32_t)sizeof(T), 0x00020000);
}
template <int R>
__device__ __forceinline__ void load_words(uint32_t (&w)[R], const uint32_t* words, uint32_t nb, uint32_t n, int lane)
{
    constexpr uint32_t PADW = nw_pad_word(64 * R);
    const rsrc_t rh = hop_rsrc(words, nb, n);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = lane + 64 * r;
        const uint32_t x = __builtin_amdgcn_raw_buffer_load_b32(rh, j * 4u, nb * 4u, BDDMMA_LD_AUX);
        w[r] = (j < n) ? x : PADW;
    }
}
// Cache policy of the potentials' loads (F / T: written by one sweep, read once by the next) — a build knob (round 5, tools/exp_r05_af.sh,
// profiles/r05_placement.txt #7): loaded non-temporally (2) they leave more of the arrays that ARE read again in the Infinity Cache once the working
// set is beyond its reach — double at 10.5 M nodes 4 410-4 460 -> 4 490-4 560 it/s, with the staging tables' loads (BDDMMA_LD_TAB_AUX) 4 590-4 610;
// float at 21 M nodes 3 570-3 840 -> 3 750-3 990 — and cost 7-12 % within its reach (float at 10.5 M nodes: 8 790 -> 8 200 / 7 700).  A RUN-TIME
// switch (a uniform branch around the loads, as the non-temporal stores have) was built and lost all of it (double 4 370-4 455 against 4 420-4 556
// without, the branches cost the hop loop its schedule): the switch has to be a template parameter of the sweeps — next round.
#ifndef BDDMMA_LD_POT_AUX
#define BDDMMA_LD_POT_AUX BDDMMA_LD_AUX
#endif
__device__ __forceinline__ void hop_load(float& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, voff, soff, BDDMMA_LD_POT_AUX));
}
__device__ __forceinline__ void hop_load(double& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rh, voff, soff, BDDMMA_LD_POT_AUX));
}
// Cache policy of the potentials' stores (the F / T streams of a streaming sweep).  Double-precision instances whose arrays exceed the
// Infinity Cache several times store them non-temporally (aux bit 1 = nt; PackDev::nt_potentials, chosen by the solver from the
// instance's footprint): the 152 MB a sweep writes there no longer displace the arc costs and exchange arrays before the next launch
// reads them — 10.5 M nodes 3 930 -> 4 095 it/s, row size 32: 3 310 -> 3 485 (A/B on one box).  Instances that fit the cache lose with
// it (4.2 M nodes: 10 170 -> 9 070), and so does float at every size (10.5 M: 8 290 -> 7 090 it/s — the 4-byte hop slices end in partial
// lines, which the cached path merges with the next hop's store), hence the run-time switch and double only.
#ifndef BDDMMA_ST_FT_AUX_F32
#define BDDMMA_ST_FT_AUX_F32 BDDMMA_ST_AUX
#endif
__device__ __forceinline__ void hop_store(float v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rh, voff, soff, BDDMMA_ST_FT_AUX_F32);
}
template <int AUX = BDDMMA_ST_AUX>
__device__ __forceinline__ void hop_store(double v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(rh, 0, 0, 0)), v), rh, voff, soff, AUX);
}
// values of the hop's slots; lanes past the last slot read 0
template <typename REAL, int R>
__device__ __forceinline__ void load_vals(REAL (&v)[R], const REAL* src, uint32_t nb, uint32_t n, int lane)
{
    const rsrc_t rh = hop_rsrc(src, nb, n);
#pragma unroll
    for (int r = 0; r < R; ++r) hop_load(v[r], rh, (lane + 64 * r) * (uint32_t)sizeof(REAL), nb * (uint32_t)sizeof(REAL));
}
// ... with the cache policy as a template parameter (NT: non-temporal; the third-generation sweeps' instantiation for footprints beyond the Infinity
// Cache's reach, see BDDMMA_LD_POT_AUX above)
__device__ __forceinline__ void hop_load_nt(float& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, voff, soff, 2));
}
__device__ __forceinline__ void hop_load_nt(double& v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rh, voff, soff, 2));
}
template <typename REAL, int R, bool NT>
__device__ __forceinline__ void load_vals_p(REAL (&v)[R], const REAL* src, uint32_t nb, uint32_t n, int lane)
{
    if constexpr (!NT) {
        load_vals<REAL, R>(v, src, nb, n, lane);
    } else {
        const rsrc_t rh = hop_rsrc(src, nb, n);
#pragma unroll
        for (int r = 0; r < R; ++r) hop_load_nt(v[r], rh, (lane + 64 * r) * (uint32_t)sizeof(REAL), nb * (uint32_t)sizeof(REAL));
    }
}
// ... and the store of one value per slot of the hop (padding slots inside the hop included: nothing reads them)
template <int R>
__device__ __forceinline__ void store_vals(const float (&v)[R], float* dst, uint32_t nb, uint32_t n, int lane, uint32_t /*nt*/)
{
    const rsrc_t rh = hop_rsrc(dst, nb, n);
#pragma unroll
    for (int r = 0; r < R; ++r) hop_store(v[r], rh, (lane + 64 * r) * 4u, nb * 4u);
}
template <int R>
__device__ __forceinline__ void store_vals(const double (&v)[R], double* dst, uint32_t nb, uint32_t n, int lane, uint32_t nt)
{
    const rsrc_t rh = hop_rsrc(dst, nb, n);
    if (nt) {  // uniform (a kernel argument)
#pragma unroll
        for (int r = 0; r < R; ++r) hop_store<2>(v[r], rh, (lane + 64 * r) * 8u, nb * 8u);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) hop_store<>(v[r], rh, (lane + 64 * r) * 8u, nb * 8u);
    }
}

#ifndef BDDMMA_LOOKAHEAD
#define BDDMMA_LOOKAHEAD 1
#endif
// SEG = false: no pack of the launch has a layer wider than two nodes — the segmented minimum is the DPP pair, its LDS variant and the
// per-lane-group branch on the pack's step count are compiled out.
template <typename REAL, int R, int MODE, int WPB, int LA = BDDMMA_LOOKAHEAD, bool SEG = true, bool NT = false>
__device__ __forceinline__ void fwd_narrow_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t block_id)
{
    constexpr int W = 64 * R;
    constexpr bool NEED_T = (MODE != FWD_PLAIN);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);  // staged {delta_lo, delta_hi} of the workgroup's stage groups; .x is overwritten by mm
    // per wave; +2: constant sink entries at index TOP = W (cost-from-terminal 0) and BOT = W + 1 (+inf);
    // for sF they are dummy push targets, so sink children need no branch
    __shared__ REAL sF_[WPB][2][W + 2];
    __shared__ REAL sT_[WPB][2][W + 2];  // costs-from-terminal of the next hop, written one hop ahead (double buffer)
    __shared__ unsigned char sAct_[WPB][2][MODE == FWD_SOLUTION ? W + 2 : 1];
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    // per-wave scratch of the segmented minimum (seg_min2): behind the rest of the dynamic LDS, only reserved when a pack has layers
    // wider than two nodes (pk.seg_off; never dereferenced otherwise)
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    auto& sF = sF_[wave];
    auto& sT = sT_[wave];
    auto& sAct = sAct_[wave];
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)  // uniform for the workgroup
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;  // the last quad may be partial: such a wave only helps staging
    const bool hdr = pk.hdr_pack != nullptr;  // uniform: resident headers (PackDev::hdr_pack)
    const uint32_t* const hp = hdr ? pk.hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t c0_h = (hdr && MODE == FWD_SOLVE) ? pk.hdr_quad[4 * (size_t)quad] : 0, cnt_h = (hdr && MODE == FWD_SOLVE) ? pk.hdr_quad[4 * (size_t)quad + 1] : 0;
    const int steps = SEG ? (has_pack ? pk.pack_steps[p] : 0) : 1;
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);  // the pack's first slot / layer: everything below is relative to them (HopWindow)
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr && MODE == FWD_SOLVE) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);  // on their way while the pipeline is set up
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    (void)lohi_p;
    // hop_node_off / hop_layer_off have one entry past the last hop of the last pack, so index q1 is
    // always readable; offsets beyond q1 are clamped (those hops have no nodes for this pack)
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    auto off = [&](uint32_t q) { return hw.node_off(q); };
    // word address of slot s of this pack = s + wd (the pack's words live in a sequence shared by all packs of its structure)
    const uint32_t wd = !has_pack ? 0 : (hdr ? hp[6] : pk.pack_word_off[p]);  // (slot offsets are relative to the pack's first slot)
    // Software pipeline with a look-ahead of D hops: at the start of hop q the wave holds the node words of hops q .. q+2D-1, the
    // costs-from-terminal of hops q+2 .. q+D+1 (those of hop q+1 are already in LDS) and the layer data of hops q .. q+D-1; during hop q
    // it requests the words of hop q+2D, T of hop q+D+2 and — from the words of hop q+D, which were requested D hops ago — the layer
    // data of hop q+D.  Every request has D hop times to arrive.  o[i] = first slot of hop q+i (uniform); the newest offset and the
    // layer offset of the next hop are read from the LDS window one hop before they are used, in the hop's single batch of LDS reads.
    constexpr int D = LA;
    uint32_t o[2 * D + 3];
    uint32_t lcur = 0;  // first layer of hop q+D
    uint32_t wr[2 * D + 1][R];
    REAL tr[D + 1][R];
    HopLayer<REAL, R> Lr[D + 1];
#pragma unroll
    for (int i = 0; i < 2 * D + 3; ++i) o[i] = 0;
    if (has_pack) {
        hw.fill(pk, q0, lane);
#pragma unroll
        for (int i = 0; i < 2 * D + 3; ++i) o[i] = off(q0 + i);
        lcur = hw.layer_off(q0 + D);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = lane + 64 * r;
            sF[0][j] = (j < o[1] - o[0]) ? REAL(0) : INF;  // every slot of hop 0 is a root (flush_costs_from_root)
            if (MODE == FWD_SOLUTION) sAct[0][j] = (j < o[1] - o[0]) ? 1 : 0;
        }
        if (lane < 4) sT[lane >> 1][W + (lane & 1)] = (lane & 1) ? INF : REAL(0);
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) load_words<R>(wr[i], d.nwords, o[i] + wd, o[i + 1] - o[i], lane);   // none past the last hop
        if (NEED_T) {
            REAL t1[R];
            load_vals_p<REAL, R, NT>(t1, Tp, o[1], o[2] - o[1], lane);  // T of hop q0+1: straight into LDS
#pragma unroll
            for (int i = 0; i < D; ++i) load_vals_p<REAL, R, NT>(tr[i], Tp, o[i + 2], o[i + 3] - o[i + 2], lane);  // T of hop q0+2+i
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                if (j < o[2] - o[1]) sT[0][j] = t1[r];
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) load_layer<REAL, R>(Lr[i], wr[i], hw.layer_off(q0 + i), rs);
        wave_sync();
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) wr[i][r] = nw_pad_word(64 * R);
    }
    int cur = 0;
    uint32_t q = q0;
    uint32_t rt = NO_ROOT;  // root slot of hop q when a BDD starts there (staggered packs); the first hop's roots are set up above
    const uint32_t g0 = (MODE == FWD_SOLVE && has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = (MODE == FWD_SOLVE && has_pack) ? (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0) : 0;
    const uint32_t r0 = (MODE == FWD_SOLVE && !hdr) ? pk.quad_round_ptr[quad] : 0;
    const uint32_t n_rounds = (MODE == FWD_SOLVE && !hdr) ? pk.quad_round_ptr[quad + 1] - r0 : 1;
    P2* sDw = sD + (size_t)wave * pk.stage_cap;  // this wave's slots of the staging area
    for (uint32_t k = 0; k < n_rounds; ++k) {
        uint32_t gl0 = 0, cnt = 0, qe = q1;
        if (MODE == FWD_SOLVE) {
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
        }
        // One hop of the pack.  The loop below executes HOP_UNROLL of them per trip: the compiler drains all outstanding
        // memory operations at the loop header (s_waitcnt vmcnt(0), which also waits for the stores of the hop just
        // finished), so the chain "wait, LDS round trips, stores" is paid once per trip; inside a trip the waits are
        // counted and the pipeline-register rotation is renamed away.  Latency-bound cases gain most: sweeps of
        // 100-variable rows -10 % (solve) / -22 % (plain), the 1 M-node benchmark -6 %; the saturated 10.5 M one +-1 %.
        // One hop of the pack.  A wave's hop used to be a chain of ~9 dependent LDS round trips (offsets, frontier, T set-up -> gather,
        // staged pairs, per 64-lane group in turn), ~130 cycles each: with few waves per SIMD that chain, not HBM, set the hop time.
        // Now everything a hop reads from LDS — the frontier, the children's costs-from-terminal (written one hop ahead), the staged
        // pairs, the next hop's offsets — is one batch of reads for all R groups, followed by the arithmetic, followed by the writes.
        auto hop = [&]() {
            if (q + 2 * D + 3 >= hw.base + HOP_WIN && hw.base + HOP_WIN <= q1) hw.fill(pk, q, lane);
            const uint32_t nb = o[0];
            const uint32_t n3 = o[3] - o[2];  // slots of hop q+2
            // ---- global prefetch
            load_words<R>(wr[2 * D], d.nwords, o[2 * D] + wd, o[2 * D + 1] - o[2 * D], lane);
            if (NEED_T) load_vals_p<REAL, R, NT>(tr[D], Tp, o[D + 2], o[D + 3] - o[D + 2], lane);
            load_layer<REAL, R>(Lr[D], wr[D], lcur, rs);  // all padding past the last hop: no loads
            uint32_t (&wa)[R] = wr[0];
            HopLayer<REAL, R>& La = Lr[0];
            // ---- the hop's LDS reads, one batch
            REAL f[R], tl[R], th[R];
            P2 dd[R];
            bool on_path[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                const uint32_t w = wa[r];
                const bool act = !(w & NW_PAD);
                const uint32_t lo_i = w & NW_CHILD_MASK, hi_i = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
                f[r] = sF[cur][j];
                if (j == rt) f[r] = REAL(0);  // a BDD that starts at this hop: its root has no parents (flush_costs_from_root)
                if (NEED_T) {
                    tl[r] = sT[cur][lo_i];  // sinks: [W] = 0, [W+1] = +inf
                    th[r] = sT[cur][hi_i];
                }
                if (MODE == FWD_SOLVE) dd[r] = sDw[act ? La.lg[r] - gl0 : 0];  // staging index: position of the layer inside its group
                if (MODE == FWD_SOLUTION) on_path[r] = act && (sAct[cur][j] || j == rt);
            }
            const uint32_t o_new = off(q + 2 * D + 3);
            const uint32_t l_next = hw.layer_off(q + D + 1);
            const uint32_t rt_next = hw.root_of(q + 1);
            // ---- set-up of the next hop's buffers (nothing above depends on it)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                if (NEED_T && j < n3) sT[cur ^ 1][j] = tr[0][r];  // T of hop q+2, gathered by hop q+1
                sF[cur ^ 1][j] = INF;
                if (MODE == FWD_SOLUTION) sAct[cur ^ 1][j] = 0;
            }
            wave_sync();
            // ---- arithmetic
            REAL nlo[R], nhi[R], mmv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t w = wa[r];
                const bool act = !(w & NW_PAD);
                const REAL lc = La.c[r].x, hc = La.c[r].y;
                nlo[r] = lc;
                nhi[r] = hc;
                if (MODE == FWD_SOLVE) {
                    REAL m0 = act ? (f[r] + lc) + tl[r] : INF;
                    REAL m1 = act ? (f[r] + hc) + th[r] : INF;
                    seg_min2(m0, m1, lane, nw_pos(w), nw_len(w), steps, sM);
                    const REAL mm = mm_diff(m0, m1, omega);
                    mmv[r] = mm;
                    nlo[r] = (lc + min0(mm)) + dd[r].x;
                    nhi[r] = (hc + min0_neg(mm)) + dd[r].y;
                }
            }
            // ---- writes: staged min-marginal differences, pushes into the next frontier, global stores
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                const uint32_t w = wa[r];
                const uint32_t lo_i = w & NW_CHILD_MASK, hi_i = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
                if (MODE == FWD_SOLVE) {
                    const bool head = nw_head(w);
                    P2 nc;
                    nc.x = nlo[r];
                    nc.y = nhi[r];
                    // the offset goes through an opaque register: otherwise the compiler folds the select into the `if (head)` below and
                    // emits the store twice, in two out-of-line blocks (four taken branches per lane group and hop)
                    uint32_t soff = head ? La.lg[r] * (uint32_t)sizeof(P2) : OOB;
                    asm volatile("" : "+v"(soff));
                    bstore(nc, rs.lohi, soff);
                    if (head) sDw[La.lg[r] - gl0].x = mmv[r];  // every lane of the layer has read its pair above (same wave, in order)
                }
                if (MODE == FWD_SOLUTION) {
                    // compute_bdd_sol_func, bdd_cuda_base.cu:1103-1137 (with the `< 0` fix of SURVEY.md §8)
                    if (on_path[r]) {
                        const REAL hi_path = f[r] + (th[r] + La.c[r].y);  // backward_step_with_path_costs, :633-640
                        const REAL lo_path = f[r] + (tl[r] + La.c[r].x);
                        const bool take_lo = (hi_path - lo_path) > 0;
                        d.sol_out[l0 + La.lg[r]] = take_lo ? 0 : 1;
                        sAct[cur ^ 1][take_lo ? lo_i : hi_i] = 1;  // sink entries are dummies
                    }
                }
                // Pushes into the sinks (and from padding lanes, whose children are BOT) have no reader: they are masked out.  As plain
                // pushes into two dummy entries they were the slowest instructions of the sweep — same-address LDS atomics serialise
                // at 20-100 cycles per lane (measured with half-empty packs: +0.56 us per hop for 64 such lane-ops), and in a pack of
                // equal rows every lane's last hop pushes into a sink.
                // (Branch-free: a masked lane "pushes" +inf into its own slot j — a no-op on a distinct address.  As `if (child < W)` the
                // compiler moved every push out of line, two taken branches each.)
                const bool plo = lo_i < (uint32_t)W, phi = hi_i < (uint32_t)W;
                lds_min(&sF[cur ^ 1][plo ? lo_i : j], plo ? f[r] + nlo[r] : INF);
                lds_min(&sF[cur ^ 1][phi ? hi_i : j], phi ? f[r] + nhi[r] : INF);
            }
            // (the argmin-path sweep leaves the stored costs-from-root alone: nothing reads them after it, and it is 38 MB of the ~120 MB the
            // sweep moves at 10.5 M nodes)
            if (MODE != FWD_SOLUTION) store_vals<R>(f, Fp, nb, o[1] - o[0], lane, pk.nt_potentials);
            wave_sync();
            cur ^= 1;
            // ---- rotate the pipeline registers
#pragma unroll
            for (int i = 0; i < 2 * D + 2; ++i) o[i] = o[i + 1];
            o[2 * D + 2] = o_new;
            lcur = l_next;
            rt = rt_next;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) wr[i][r] = wr[i + 1][r];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                Lr[i] = Lr[i + 1];
                if (NEED_T) {
#pragma unroll
                    for (int r = 0; r < R; ++r) tr[i][r] = tr[i + 1][r];
                }
            }
            ++q;
        };
        while (q + HOP_UNROLL <= qe) {
#pragma unroll
            for (int u = 0; u < HOP_UNROLL; ++u) hop();
        }
        while (q < qe) hop();
        if (MODE == FWD_SOLVE) {
            if (WPB > 1) __syncthreads(); else wave_sync();
            stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);  // min-marginal differences of the round -> entry array
            if (WPB > 1) __syncthreads();                        // the next round overwrites the staging area
        }
    }
}

// register budget of the float solve sweeps of general linear rows without wide packs, as the mixed kernels' (wide.hpp): 80 VGPRs with 64-slot
// packs; 96 with 128-slot packs swept one per workgroup (staggered packs; 109-122 before: 100 000 rows of 11 variables in 128-slot packs
// 6 080 -> 7 280 it/s, tools/exp_r05_u.sh) — workgroups of four or eight packs (the large set-cover instances) keep their registers
#ifndef BDDMMA_N1_WAVES
#define BDDMMA_N1_WAVES(REAL, R, MODE, WPB) ((MODE) == 1 && sizeof(REAL) == 4 ? ((R) == 1 ? 6 : (R) == 2 && (WPB) == 1 ? 5 : 1) : 1)
#endif
template <typename REAL, int R, int MODE, int WPB, bool SEG = true, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N1_WAVES(REAL, R, MODE, WPB)))) k_fwd_narrow(DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    fwd_narrow_body<REAL, R, MODE, WPB, BDDMMA_LOOKAHEAD, SEG, NT>(d, pk, omega, blockIdx.x);
}

template <typename REAL, int R, int MODE, int WPB, int LA = BDDMMA_LOOKAHEAD, bool SEG = true, bool NT = false>
__device__ __forceinline__ void bwd_narrow_body(const DevPtrs<REAL>& d, const PackDev& pk, REAL omega, uint32_t block_id)
{
    constexpr int W = 64 * R;
    constexpr bool NEED_F = (MODE != BWD_PLAIN);
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    P2* sD = reinterpret_cast<P2*>(dyn_lds);
    __shared__ REAL sT_[WPB][2][W + 2];  // per wave; +2: sink entries TOP = W (0) and BOT = W + 1 (+inf)
    __shared__ uint32_t sOffN_[WPB][HOP_WIN], sOffL_[WPB][HOP_WIN], sOffR_[WPB][HOP_WIN];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    // per-wave scratch of the segmented minimum (seg_min2): behind the rest of the dynamic LDS, only reserved when a pack has layers
    // wider than two nodes (pk.seg_off; never dereferenced otherwise)
    REAL* sM = reinterpret_cast<REAL*>(dyn_lds + pk.seg_off) + wave * 128;
    auto& sT = sT_[wave];
    const uint32_t n_quads = (pk.n_packs + WPB - 1) / WPB;
    const uint32_t quad = block_to_pack(block_id, n_quads, pk.xcd_chunk);
    BDDMMA_EXIT_IF(quad >= n_quads, d)
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    const bool hdr = pk.hdr_pack != nullptr;  // uniform: resident headers (PackDev::hdr_pack)
    const uint32_t* const hp = hdr ? pk.hdr_pack + 8 * (size_t)(has_pack ? p : 0) : nullptr;
    const uint32_t q0 = !has_pack ? 0 : (hdr ? hp[4] : pk.pack_hop_ptr[p]);
    const uint32_t q1 = !has_pack ? 0 : (hdr ? q0 + (hp[5] & 0xFFFFu) : pk.pack_hop_ptr[p + 1]);
    const uint32_t c0_h = (hdr && MODE == BWD_SOLVE) ? pk.hdr_quad[4 * (size_t)quad] : 0, cnt_h = (hdr && MODE == BWD_SOLVE) ? pk.hdr_quad[4 * (size_t)quad + 1] : 0;
    const int steps = SEG ? (has_pack ? pk.pack_steps[p] : 0) : 1;
    const REAL INF = inf_v<REAL>();
    const uint32_t slot_first = !has_pack ? 0 : (hdr ? hp[0] : pk.hop_node_off[q0]), l0 = !has_pack ? 0 : (hdr ? hp[2] : pk.hop_layer_off[q0]);  // the pack's first slot / layer: everything below is relative to them (HopWindow)
    NarrowRs<REAL> rs(d);
    rs.rebase_layers(d, l0);
    uint32_t ent[STAGE_ITERS], esl[STAGE_ITERS];
    if (hdr && MODE == BWD_SOLVE) stage_load_tables<REAL, WPB, (NT ? 2 : BDDMMA_LD_TAB_AUX)>(ent, esl, rs, c0_h, cnt_h, tid);  // on their way while the pipeline is set up
    REAL* const Tp = d.T + slot_first;
    REAL* const Fp = d.F + slot_first;
    const REAL* const lohi_p = d.lohi + 2 * (size_t)l0;
    (void)lohi_p;
    const rsrc_t rxl = make_rsrc(d.x_layer != nullptr ? d.x_layer + l0 : d.x_layer, d.x_layer != nullptr ? d.n_layers - l0 : 0u);  // DevPtrs::x_layer, from the pack's first layer on
    (void)rxl;
    HopWindow hw{sOffN_[wave], sOffL_[wave], sOffR_[wave], q0, q1, slot_first, l0};
    double lb_stag = 0.0;  // costs-to-terminal of the roots below the pack's first hop (staggered packs), for the lower bound
    // node range of hop q; hops below q0 (pipeline run-off) are empty
    auto nb_of = [&](uint32_t q) { return hw.node_off(q); };
    const uint32_t wd = !has_pack ? 0 : (hdr ? hp[6] : pk.pack_word_off[p]);  // (slot offsets are relative to the pack's first slot)  // see k_fwd_narrow
    // Software pipeline with a look-ahead of D hops, mirrored from k_fwd_narrow: before hop q is processed (q counts down) the wave
    // holds the node words of hops q .. q-2D+1, the costs-from-root of hops q .. q-D and the layer data of hops q .. q-D+1; during the hop
    // it requests the words of hop q-2D, F of hop q-D-1 and the layer data of hop q-D.  o[i] = first slot of hop q+1-i (hops below q0
    // are empty: their offset is the one of q0); the offsets the next hop needs are read one hop ahead, in the hop's LDS batch.
    constexpr int D = LA;
    uint32_t o[2 * D + 2];
    uint32_t lcur = 0;  // first layer of hop q-D
    uint32_t wr[2 * D + 1][R];
    REAL fr[D + 2][R];
    HopLayer<REAL, R> Lr[D + 1];
#pragma unroll
    for (int i = 0; i < 2 * D + 2; ++i) o[i] = 0;
    uint32_t q = q1;
    if (has_pack) {
        if (lane < 4) sT[lane >> 1][W + (lane & 1)] = (lane & 1) ? INF : REAL(0);
        hw.fill(pk, q1 + 1 > q0 + HOP_WIN ? q1 + 1 - HOP_WIN : q0, lane);  // window ends at record q1
        // state of the first hop, q = q1-1: o[i] = nb_of(q1 - i)
#pragma unroll
        for (int i = 0; i < 2 * D + 2; ++i) o[i] = nb_of(q1 >= q0 + i ? q1 - i : q0);
        lcur = hw.layer_off(q1 >= q0 + D + 1 ? q1 - 1 - D : q0);
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) load_words<R>(wr[i], d.nwords, o[i + 1] + wd, o[i] - o[i + 1], lane);  // hop q1-1-i (none below q0)
        if (NEED_F) {
#pragma unroll
            for (int i = 0; i < D + 1; ++i) load_vals_p<REAL, R, NT>(fr[i], Fp, o[i + 1], o[i] - o[i + 1], lane);   // F of hop q1-1-i
        }
#pragma unroll
        for (int i = 0; i < D; ++i) load_layer<REAL, R>(Lr[i], wr[i], hw.layer_off(q1 >= q0 + i + 1 ? q1 - 1 - i : q0), rs);
    } else {
#pragma unroll
        for (int i = 0; i < 2 * D; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) wr[i][r] = nw_pad_word(64 * R);
    }
    int cur = 0;
    const uint32_t g0 = (MODE == BWD_SOLVE && has_pack && !hdr) ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = (MODE == BWD_SOLVE && has_pack) ? (hdr ? 1u : pk.pack_group_ptr[p + 1] - g0) : 0;
    const uint32_t r0 = (MODE == BWD_SOLVE && !hdr) ? pk.quad_round_ptr[quad] : 0;
    const uint32_t n_rounds = (MODE == BWD_SOLVE && !hdr) ? pk.quad_round_ptr[quad + 1] - r0 : 1;
    P2* sDw = sD + (size_t)wave * pk.stage_cap;
    for (uint32_t k = n_rounds; k-- > 0;) {  // same rounds as the forward sweep, in reverse
        uint32_t gl0 = 0, cnt = 0, qs = q0;
        if (MODE == BWD_SOLVE) {
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
        }
        auto hop = [&]() {  // see k_fwd_narrow: one batch of LDS reads, the arithmetic, the writes
            --q;
            // o[i] = nb_of(q + 1 - i), i <= 2D+1; the next hop adds nb_of(q - 1 - 2D) and the first layer of hop q - 1 - D
            if (q < hw.base + 2 * D + 1 && hw.base > q0) hw.fill(pk, q + 1 > q0 + HOP_WIN ? q + 1 - HOP_WIN : q0, lane);
            const uint32_t nb = o[1];
            // ---- prefetch: words of hop q-2D, F of hop q-D-1, layer data of hop q-D
            load_words<R>(wr[2 * D], d.nwords, o[2 * D + 1] + wd, o[2 * D] - o[2 * D + 1], lane);
            if (NEED_F) load_vals_p<REAL, R, NT>(fr[D + 1], Fp, o[D + 2], o[D + 1] - o[D + 2], lane);
            load_layer<REAL, R>(Lr[D], wr[D], lcur, rs);  // all padding below the first hop: no loads
            uint32_t (&wa)[R] = wr[0];
            REAL (&fa)[R] = fr[0];
            HopLayer<REAL, R>& La = Lr[0];
            // ---- LDS reads
            REAL tl[R], th[R];
            P2 dd[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t w = wa[r];
                const bool act = !(w & NW_PAD);
                const uint32_t lo_i = w & NW_CHILD_MASK, hi_i = (w >> NW_CHILD_BITS) & NW_CHILD_MASK;
                tl[r] = sT[cur][lo_i];  // sinks: [W] = 0, [W+1] = +inf
                th[r] = sT[cur][hi_i];
                if (MODE == BWD_SOLVE) dd[r] = sDw[act ? La.lg[r] - gl0 : 0];
            }
            const uint32_t o_new = (q >= q0 + 2 * D + 1) ? nb_of(q - 1 - 2 * D) : o[2 * D + 1];
            const uint32_t l_next = hw.layer_off(q >= q0 + D + 1 ? q - 1 - D : q0);
            const uint32_t rt = q > q0 ? hw.root_of(q) : (uint32_t)NO_ROOT;  // the first hop's roots are summed behind the loop
            // ---- arithmetic
            REAL t[R], nlo[R], nhi[R], mmv[R], lp[R], hp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t w = wa[r];
                const bool act = !(w & NW_PAD);
                const REAL lc = La.c[r].x, hc = La.c[r].y;
                if (MODE == BWD_SOLVE) {
                    REAL m0 = act ? (fa[r] + lc) + tl[r] : INF;
                    REAL m1 = act ? (fa[r] + hc) + th[r] : INF;
                    seg_min2(m0, m1, lane, nw_pos(w), nw_len(w), steps, sM);
                    const REAL mm = mm_diff(m0, m1, omega);
                    mmv[r] = mm;
                    nlo[r] = (lc + min0(mm)) + dd[r].x;
                    nhi[r] = (hc + min0_neg(mm)) + dd[r].y;
                    t[r] = rmin(nhi[r] + th[r], nlo[r] + tl[r]);
                } else {
                    const REAL ch = th[r] + hc, cl = tl[r] + lc;  // backward_step, bdd_cuda_base.cu:646-667
                    t[r] = rmin(ch, cl);
                    if (MODE == BWD_MARGINALS) {
                        lp[r] = act ? fa[r] + cl : INF;  // backward_step_with_path_costs, :633-641
                        hp[r] = act ? fa[r] + ch : INF;
                        seg_min2(lp[r], hp[r], lane, nw_pos(w), nw_len(w), steps, sM);
                    }
                }
            }
            // ---- writes
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t j = lane + 64 * r;
                const uint32_t w = wa[r];
                const bool act = !(w & NW_PAD);
                if (MODE == BWD_SOLVE) {
                    const bool head = nw_head(w);
                    P2 nc;
                    nc.x = nlo[r];
                    nc.y = nhi[r];
                    uint32_t soff = head ? La.lg[r] * (uint32_t)sizeof(P2) : OOB;  // see k_fwd_narrow
                    asm volatile("" : "+v"(soff));
                    bstore(nc, rs.lohi, soff);
                    if (head) sDw[La.lg[r] - gl0].x = mmv[r];
                    if (d.x_layer != nullptr) {  // uniform: net_solver_costs in layer order for an L-BFGS wrapper, straight from the hop (as a pass over the staging
                        // area behind the round it cost 13 us of a 41 us sweep at 10.5 M nodes, tools/xlayer_cost.py)
                        uint32_t xoff = head ? La.lg[r] * (uint32_t)sizeof(REAL) : OOB;
                        asm volatile("" : "+v"(xoff));
                        bstore((nhi[r] - nlo[r]) + mmv[r], rxl, xoff);
                    }
                }
                if (MODE == BWD_MARGINALS) {
                    if (nw_head(w)) {
                        d.mm0_out[l0 + La.lg[r]] = lp[r];
                        d.mm1_out[l0 + La.lg[r]] = hp[r];
                    }
                }
                if (act) sT[cur ^ 1][j] = t[r];
                if (j == rt) lb_stag += (double)t[r];
            }
            store_vals<R>(t, Tp, nb, o[0] - o[1], lane, pk.nt_potentials);
            wave_sync();
            cur ^= 1;
#pragma unroll
            for (int i = 0; i < 2 * D + 1; ++i) o[i] = o[i + 1];
            o[2 * D + 1] = o_new;
            lcur = l_next;
#pragma unroll
            for (int i = 0; i < 2 * D; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) wr[i][r] = wr[i + 1][r];
#pragma unroll
            for (int i = 0; i < D; ++i) Lr[i] = Lr[i + 1];
            if (NEED_F) {
#pragma unroll
                for (int i = 0; i < D + 1; ++i)
#pragma unroll
                    for (int r = 0; r < R; ++r) fr[i][r] = fr[i + 1][r];
            }
        };
        while (q >= qs + HOP_UNROLL) {
#pragma unroll
            for (int u = 0; u < HOP_UNROLL; ++u) hop();
        }
        while (q > qs) hop();
        if (MODE == BWD_SOLVE) {
            if (WPB > 1) __syncthreads(); else wave_sync();
            stage_flush<REAL, WPB>(sD, ent, esl, rs, cnt, tid);
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
        if (j < n0) s += (double)sT[cur][j];
    }
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2);
    if (lane == 0) d.lb_partial[pk.lb_base + p] = s;
}

template <typename REAL, int R, int MODE, int WPB, bool SEG = true, bool NT = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(BDDMMA_N1_WAVES(REAL, R, MODE, WPB)))) k_bwd_narrow(DevPtrs<REAL> d, PackDev pk, REAL omega)
{
    bwd_narrow_body<REAL, R, MODE, WPB, BDDMMA_LOOKAHEAD, SEG, NT>(d, pk, omega, blockIdx.x);
}

}  // namespace bddmma
