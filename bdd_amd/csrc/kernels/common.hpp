// kernels/common.hpp — shared device-side types and helpers of the sweep kernels: DevPtrs / PackDev, run_solver gate, per-layer segmented minimum, buffer-descriptor memory ops, cooperative staging.
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

enum : int { FWD_PLAIN = 0, FWD_SOLVE = 1, FWD_SOLUTION = 2 };
enum : int { BWD_PLAIN = 0, BWD_SOLVE = 1, BWD_MARGINALS = 2 };

template <typename REAL>
struct DevPtrs {
    const uint32_t* nwords;  // narrow node words: distinct pack sequences, pack p's at PackDev::pack_word_off[p]
    uint32_t n_nwords;
    const uint64_t* wwords;  // wide node words, indexed by slot - wide_slot_base
    uint32_t wide_slot_base;
    REAL* F;                 // cost from root, per slot
    REAL* T;                 // cost from terminal, per slot
    REAL* lohi;              // per layer: {lo, hi} arc costs, interleaved
    // variable <-> layer exchange arrays in binned entry order (layout.hpp, struct Exchange)
    const REAL* delta_lay;   // 2 REAL per entry: {delta_lo, delta_hi} of the entry's variable (normalised)
    REAL* mm_binned;         // 1 REAL per entry: deferred min-marginal difference of the entry's layer
    const uint32_t* lpos;    // per layer: entry index
    const uint32_t* cs_entry;  // cooperative staging: staged item -> entry
    const uint16_t* cs_slot;   // cooperative staging: staged item -> LDS slot
    uint32_t n_slots;        // element counts (buffer descriptors of the narrow kernels)
    uint32_t n_layers;
    uint32_t n_narrow_layers;
    double* lb_partial;      // per pack (narrow packs first, then wide)
    REAL* x_layer;           // BWD_SOLVE: net_solver_costs x = (hi' - lo') + mm (bdd_cuda_parallel_mma.cu:432-463) in layer order, formed by the sweep itself
                             // from the new arc costs and the deferred difference (nullptr: not wanted; SolverT::lbfgs_views)
    REAL* mm0_out;           // BWD_MARGINALS outputs, per layer
    REAL* mm1_out;
    char* sol_out;           // FWD_SOLUTION output, per layer
    // Device-resident run_solver (run_ctl_step): when the termination test of run_solver_util.h:56-73 has fired on the device, the
    // launches of the iterations the host had already queued return at once.  nullptr outside run_solver (one scalar compare of a
    // kernel argument); otherwise one scalar load per launch.  *stop = number of iterations after which the loop ended (UINT32_MAX
    // while it runs); run_iter = index of the iteration this launch belongs to.  A launch is skipped when *stop <= run_iter, so the
    // launch that latches the word (it belongs to iteration *stop - 1) can never skip part of its own grid — with a plain flag the
    // workgroups dispatched after workgroup 0 had latched it returned without doing their share of the exchange.
    const uint32_t* stop;
    uint32_t run_iter;
    uint32_t big;            // an entry- or slot-indexed array reaches 4 GiB: the staging transfers use 64-bit addresses (stage_load / stage_flush)
};
struct RunGate {  // the same pair for the kernels that do not take a DevPtrs
    const uint32_t* stop = nullptr;
    uint32_t iter = 0;
};
constexpr uint32_t RUN_NOT_STOPPED = 0xFFFFFFFFu;

__device__ __forceinline__ bool run_stopped(const RunGate& g) { return g.stop != nullptr && *g.stop <= g.iter; }
// The sweep kernels test the word together with their first uniform exit: the pointer is a kernel argument, so outside run_solver
// (nullptr) the test is one more scalar compare on values the kernel loads anyway — no extra dependent round trip at its start.
#define BDDMMA_EXIT_IF(done_cond, dev)                                     \
    {                                                                      \
        const bool done_ = (done_cond);                                    \
        if (done_ | ((dev).stop != nullptr)) {                             \
            if (done_ || *(dev).stop <= (dev).run_iter) return;            \
        }                                                                  \
    }

struct PackDev {
    const uint32_t* pack_hop_ptr;
    const uint32_t* hop_node_off;
    const uint32_t* hop_layer_off;
    const uint8_t* pack_steps;
    const uint16_t* hop_root;        // narrow packs: per (pack, hop) record the local slot of a BDD root below the pack's first hop, or NO_ROOT (layout.hpp)
    const uint32_t* pack_word_off;   // narrow packs: first word of the pack's (shared) word sequence
    const uint32_t* pack_group_ptr;  // narrow packs: stage groups
    const uint32_t* grp_layer_off;
    const uint32_t* grp_hop_end;
    const uint32_t* quad_round_ptr;  // cooperative staging rounds of each quad of packs
    const uint32_t* cs_ptr;          // first staged item of each (quad, round)
    uint32_t stage_cap;
    uint32_t seg_off;  // byte offset of the seg_min2 scratch (128 REALs per wave) inside the dynamic LDS
    uint32_t n_packs;
    uint32_t lb_base;  // index of this set's first pack in lb_partial
    uint32_t nt_potentials;  // streaming narrow sweeps, double: store F / T non-temporally (see hop_store)
    uint32_t xcd_chunk;      // block_to_pack: workgroups per chunk of the XCD-interleaved map (0: contiguous eighths)
    // narrow packs, streaming sweeps: the resident headers (layout.hpp: struct Resident — 8 words per pack, 4 per quad) where they hold for the
    // whole set (one stage group per pack, one round per quad, no staggered packs), else null: a wave then has its pack's hop / slot / layer /
    // word ranges and its quad's range of the staging tables after ONE round trip instead of two dependent ones each
    const uint32_t* hdr_pack;
    const uint32_t* hdr_quad;
};

// -DBDDMMA_STAMPS (tools/build_variant.sh): per-wave s_memrealtime stamps at the phase boundaries of the small-instance kernels, for
// the latency budget of profiles/r03_1m_latency.txt.  Stamp i of slot s is taken after everything issued before it has arrived
// (s_waitcnt 0), so the differences are the phases' durations on that wave.  Not compiled into the shipped library.
#ifdef BDDMMA_STAMPS
__device__ unsigned long long* g_bddmma_stamps = nullptr;
#define BDDMMA_STAMP(slot, idx)                                                                              \
    do {                                                                                                     \
        if (g_bddmma_stamps != nullptr) {                                                                    \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); /* 100 MHz */                     \
            if ((threadIdx.x & 63) == 0) g_bddmma_stamps[(size_t)(slot) * 8 + (idx)] = t_;                   \
        }                                                                                                    \
    } while (0)
#else
#define BDDMMA_STAMP(slot, idx) do { } while (0)
#endif

template <typename REAL> struct Pair;
template <> struct Pair<float> { using type = float2; };
template <> struct Pair<double> { using type = double2; };

template <typename REAL> __device__ __forceinline__ REAL inf_v();
template <> __device__ __forceinline__ float inf_v<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_v<double>() { return __builtin_huge_val(); }

__device__ __forceinline__ float rmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double rmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ bool rfinite(float a) { return __builtin_isfinite(a); }
__device__ __forceinline__ bool rfinite(double a) { return __builtin_isfinite(a); }

// mm = omega * (m1 - m0), or 0 unless both minima are finite (bdd_cuda_parallel_mma.cu:36-39).  Branch-free: with `&&` the compiler
// built two nested exec regions with a skip branch around one subtraction.
template <typename REAL>
__device__ __forceinline__ REAL mm_diff(REAL m0, REAL m1, REAL omega)
{
    const bool fin = (int)rfinite(m0) & (int)rfinite(m1);
    const REAL t = omega * (m1 - m0);
    return fin ? t : REAL(0);
}
// min(x, 0) and min(-x, 0) of a min-marginal difference (never NaN).  One instruction; __builtin_fminf on a value that went through a
// select costs a v_max x, x canonicalisation first.
__device__ __forceinline__ float min0(float x)
{
    float r;
    asm("v_min_f32_e64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float min0_neg(float x)
{
    float r;
    asm("v_min_f32_e64 %0, -%1, 0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ double min0(double x) { return rmin(x, 0.0); }
__device__ __forceinline__ double min0_neg(double x) { return rmin(-x, 0.0); }

template <typename REAL>
__device__ __forceinline__ void lds_min(REAL* p, REAL v)
{
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_min_f32 / ds_min_f64
}

// (Measured and dropped: the forward frontier as order-preserving integers with ds_min_u32 / ds_min_u64 instead of ds_min_f32 / f64 —
// same speed on every benchmark, so the float LDS minimum is not what makes the forward pushes slower than the backward gathers.)
// frontier minimum of the workgroup-per-pack kernels: LDS (ds_min) or, for huge packs whose frontier does not fit
// in LDS, global scratch memory (L2 atomic; a CAS loop where the hardware has no float minimum)
template <bool GLOBAL, typename REAL>
__device__ __forceinline__ void frontier_min(REAL* p, REAL v)
{
    if (GLOBAL) __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// values other waves of the workgroup produced with frontier_min: huge packs read them from L2 (the atomics do not
// update this CU's vector L1)
template <bool GLOBAL, typename REAL>
__device__ __forceinline__ REAL frontier_load(const REAL* p)
{
    if (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// XCD-aware block -> pack map: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md).  XCD x gets chunks of `chunk`
// consecutive workgroups' packs — chunk x, x + 8, x + 16, ... — so that neighbouring packs, which share variables in structured
// problems, sit behind the same 4 MiB L2, while every XCD sees every part of the pack sequence: instances that mix constraint
// families (cheap 10-hop covering packs, expensive 30-hop knapsack packs; packs are ordered by family) had all their expensive packs
// on one or two XCDs when each XCD owned one contiguous eighth (chunk = 0: that map, kept for A/B runs, variant_flags bit 7).
// The grid is a multiple of 8 * chunk workgroups (the launcher rounds up; surplus workgroups exit at once).
__device__ __forceinline__ uint32_t block_to_pack(uint32_t bid, uint32_t n_packs, uint32_t chunk)
{
    if (chunk == 0) {
        const uint32_t per = (n_packs + 7u) >> 3;
        return (bid & 7u) * per + (bid >> 3);
    }
    const uint32_t x = bid & 7u, i = bid >> 3;
    return ((i / chunk) * 8u + x) * chunk + i % chunk;
}

// ---- per-layer min across the lanes of a layer ---------------------------------------------------
// A layer occupies `len` consecutive lanes starting `pos` lanes below the current one (fields of the
// node word).  Result: min over the layer, in every lane of the layer.
__device__ __forceinline__ float dpp_from_next(float v)  // lane i <- lane i+1 (v_mov_b32_dpp wave_shl:1)
{
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_prev(float v)  // lane i <- lane i-1 (wave_shr:1)
{
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ double dpp_from_next(double v)
{
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = (int)x, hi = (int)(x >> 32);
    const unsigned int l2 = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);
    const unsigned int h2 = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)h2 << 32) | l2);
}
#ifndef BDDMMA_SEG_FOLD_F32
#define BDDMMA_SEG_FOLD_F32 2
#endif
template <typename REAL>
constexpr int SEG_FOLD_STEPS = sizeof(REAL) == 4 ? BDDMMA_SEG_FOLD_F32 : 1;  // see seg_min2
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)  // DPP move with control CTRL (row_shl:n = 0x100 + n); lanes without a source keep their value
{
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v)
{
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = (int)x, hi = (int)(x >> 32);
    const unsigned int l2 = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const unsigned int h2 = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)h2 << 32) | l2);
}
__device__ __forceinline__ double dpp_from_prev(double v)
{
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = (int)x, hi = (int)(x >> 32);
    const unsigned int l2 = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);
    const unsigned int h2 = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)h2 << 32) | l2);
}

// Wider layers: every lane folds its two values into the layer's LDS slots with ds_min (slot = lane of the layer's head, unique inside
// the 64-lane group) and reads the result back — 6 LDS instructions and 3 dependent LDS round trips whatever the width.  The first
// version did ceil(log2(width)) __shfl_down halving steps + a __shfl broadcast per value: 14 ds_bpermute in 7 dependent round trips and
// ~75 VALU for 64-wide layers, which made packs of knapsack-like BDDs instruction- and latency-bound (185 VALU per wave and hop).
// sM: 128 REALs of LDS owned by this wave ([0, 64) for a, [64, 128) for b); a wave's LDS operations execute in order, so only the
// compiler needs the fences.
__device__ __forceinline__ void seg_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// layers of 1 or 2 nodes (simplex / covering / cardinality-1 rows): the head takes the minimum with its right neighbour, the second
// node copies the head's result — DPP moves, no LDS crossbar traffic.  A 2-node layer never straddles the 64-lane group.
__device__ __forceinline__ void seg_pair_min(double& a, double& b, uint32_t pos, uint32_t len)
{
    const double a2 = dpp_from_next(a), b2 = dpp_from_next(b);
    if (pos == 0 && len == 2) {
        a = rmin(a, a2);
        b = rmin(b, b2);
    }
    const double a1 = dpp_from_prev(a), b1 = dpp_from_prev(b);
    if (pos == 1) {
        a = a1;
        b = b1;
    }
}
// float: the minimum with the DPP-shifted operand is one instruction (v_min_f32_dpp).  Through the builtins the compiler emits
// v_mov_b32_dpp, two v_max x, x canonicalisations and v_min per value (it cannot see that a moved float is canonical): 16 VALU per lane
// group and hop instead of 8.  s_nop 1: a DPP operand written by the preceding VALU instruction needs two wait states, and the hazard
// recogniser does not look into inline assembly.  The DPP ops run with all lanes enabled (a source lane masked out by EXEC would
// count as invalid); the selects apply the layer structure.
__device__ __forceinline__ void seg_pair_min(float& a, float& b, uint32_t pos, uint32_t len)
{
    float ta, tb;
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(ta), "=&v"(tb)
                 : "v"(a), "v"(b));
    const bool head2 = pos == 0 && len == 2;
    a = head2 ? ta : a;
    b = head2 ? tb : b;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(ta), "=&v"(tb)
                 : "v"(a), "v"(b));
    a = pos == 1 ? ta : a;
    b = pos == 1 ? tb : b;
}
// One fold step: a, b <- min with the values SH lanes up the row where `same` holds.  float: v_min_f32_dpp takes the shifted operand
// directly (through the builtins the compiler emits v_mov, v_mov_dpp, two v_max canonicalisations and v_min per value: 14 instead of 5
// VALU per step); s_nop 1: a DPP operand written by the preceding VALU instruction needs two wait states (see seg_pair_min).  Lanes
// without a source lane keep an undefined destination, which `same` (false there) never selects.
template <int SH>
__device__ __forceinline__ void seg_fold_step(float& a, float& b, bool same)
{
    float ta, tb;
#define BDDMMA_FOLD_ASM(N)                                                        \
    asm("s_nop 1\n\t"                                                             \
        "v_min_f32_dpp %0, %2, %2 row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
        "v_min_f32_dpp %1, %3, %3 row_shl:" #N " row_mask:0xf bank_mask:0xf"       \
        : "=&v"(ta), "=&v"(tb)                                                     \
        : "v"(a), "v"(b))
    static_assert(SH == 1 || SH == 2 || SH == 4, "row_shl:1 / 2 / 4");
    if (SH == 1) BDDMMA_FOLD_ASM(1);
    else if (SH == 2) BDDMMA_FOLD_ASM(2);
    else BDDMMA_FOLD_ASM(4);
#undef BDDMMA_FOLD_ASM
    a = same ? ta : a;
    b = same ? tb : b;
}
template <int SH>
__device__ __forceinline__ void seg_fold_step(double& a, double& b, bool same)
{
    const double an = dpp_row<0x100 + SH>(a), bn = dpp_row<0x100 + SH>(b);
    a = same ? rmin(a, an) : a;
    b = same ? rmin(b, bn) : b;
}
// The same fold for the wide packs, whose lanes know their layer's index in the hop (`key`: equal for the consecutive lanes of a layer;
// inactive lanes pass a key no layer has) instead of a position: returns true in the lanes that must issue the LDS atomics.
template <typename REAL>
__device__ __forceinline__ bool seg_fold_by_key(REAL& a, REAL& b, uint32_t key, int lane)
{
    constexpr int K = SEG_FOLD_STEPS<REAL>;
    constexpr uint32_t G = 1u << K;
#define BDDMMA_SEG_STEP(SH)                                                                                                \
    {                                                                                                                      \
        const uint32_t kn = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x100 + SH, 0xF, 0xF, false);      \
        seg_fold_step<SH>(a, b, kn == key);   /* past the row: kn = ~key */                                                \
    }
    BDDMMA_SEG_STEP(1)
    if (K >= 2) BDDMMA_SEG_STEP(2)
    if (K >= 3) BDDMMA_SEG_STEP(4)
#undef BDDMMA_SEG_STEP
    const uint32_t kp = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x111, 0xF, 0xF, false);  // row_shr:1: lane i <- lane i - 1
    return ((uint32_t)lane & (G - 1u)) == 0u || kp != key;
}
template <typename REAL>
__device__ __forceinline__ void seg_min2(REAL& a, REAL& b, int lane, uint32_t pos, uint32_t len, int steps, REAL* sM)
{
    if (steps <= 1) {
        seg_pair_min(a, b, pos, len);
        return;
    }
#ifdef BDDMMA_EXP_NOSEG  // timing experiment only (wrong results): what the LDS segmented minimum costs
    seg_pair_min(a, b, pos, len);
    return;
#endif
    const REAL INF = inf_v<REAL>();
    const uint32_t head = (uint32_t)lane - pos;
    sM[lane] = INF;
    sM[64 + lane] = INF;
    seg_fence();
    // Before LDS: the lanes of a layer fold their values with K DPP steps (row_shl 1, 2: lane i takes lane i + 2^j of its 16-lane row if
    // that lane belongs to the same layer, i.e. its position is pos + 2^j), so lane i holds the minimum over the next 2^K lanes of its
    // layer and row, and only every 2^K-th lane of a layer plus the first lane of each row issue the atomics.  All lanes of a layer hit ONE
    // address, which LDS serialises: at 10 M knapsack nodes the solve sweeps were LDS-bound (57 % busy, half of it these conflicts).
    // Measured there (it/s float / double): K = 0: 3 730 / 2 940, 1: 4 140 / 3 230, 2: 4 220 / 3 170, 3: 4 050 / 3 080, 4: 3 830 / 2 950
    // (measured with the builtin form of the step, ~14 VALU in float; the sweeps are VALU-bound next) -> K = 2 in float, 1 in double.
    {
        constexpr int K = SEG_FOLD_STEPS<REAL>;
        constexpr uint32_t G = 1u << K;
#define BDDMMA_SEG_STEP(SH)                                                                                               \
    {                                                                                                                     \
        const uint32_t pn = (uint32_t)__builtin_amdgcn_update_dpp((int)pos, (int)pos, 0x100 + SH, 0xF, 0xF, false);    \
        seg_fold_step<SH>(a, b, pn == pos + SH);   /* past the row: pn = pos */                                           \
    }
        BDDMMA_SEG_STEP(1)
        if (K >= 2) BDDMMA_SEG_STEP(2)
        if (K >= 3) BDDMMA_SEG_STEP(4)
#undef BDDMMA_SEG_STEP
        if ((pos & (G - 1u)) == 0u || ((uint32_t)lane & 15u) == 0u) {
            __hip_atomic_fetch_min(&sM[head], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&sM[64 + head], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    seg_fence();
    a = sM[head];
    b = sM[64 + head];
    seg_fence();  // the next group's reset must not overtake these reads
}

// ---- buffer-descriptor memory ops ---------------------------------------------------------------
// Every per-lane predicate of the narrow kernels is folded into the byte offset of a raw buffer op: an
// offset past the descriptor's size makes the hardware drop the lane (loads return 0, stores are
// discarded).  With `if (active) x = p[i]` hipcc emits an exec-masked branch per access; the waitcnt
// pass then cannot count the outstanding loads and falls back to s_waitcnt vmcnt(0), which drains
// every prefetch in flight (seen in the ISA of the first pipelined version).  Branch-free buffer ops
// keep the instruction stream straight-line, so the waits become counted vmcnt(N).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t OOB = 0xFFFFFFFFu;
#ifndef BDDMMA_LD_AUX
#define BDDMMA_LD_AUX 0
#endif
#ifndef BDDMMA_ST_AUX
#define BDDMMA_ST_AUX 0
#endif
#ifndef BDDMMA_LD_TAB_AUX   // the staging tables' loads (stage_load_tables): read once per sweep; see BDDMMA_LD_POT_AUX (kernels/narrow.hpp)
#define BDDMMA_LD_TAB_AUX 0
#endif

template <typename T>
__device__ __forceinline__ rsrc_t make_rsrc(const T* p, uint64_t n_elems)
{
    const uint64_t bytes = n_elems * sizeof(T);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, (uint32_t)(bytes > 0xFFFFFFFEull ? 0xFFFFFFFEull : bytes), 0x00020000);
}
__device__ __forceinline__ uint32_t bload_u32(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, BDDMMA_LD_AUX); }
__device__ __forceinline__ uint32_t bload_u16(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0); }
__device__ __forceinline__ void bload(float& v, rsrc_t r, uint32_t off) { v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, BDDMMA_LD_AUX)); }
__device__ __forceinline__ void bload(double& v, rsrc_t r, uint32_t off) { v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, BDDMMA_LD_AUX)); }
__device__ __forceinline__ void bload(float2& v, rsrc_t r, uint32_t off) { v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, BDDMMA_LD_AUX)); }
__device__ __forceinline__ void bload(double2& v, rsrc_t r, uint32_t off) { v = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ void bstore(float v, rsrc_t r, uint32_t off) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, BDDMMA_ST_AUX); }
__device__ __forceinline__ void bstore(double v, rsrc_t r, uint32_t off)
{
    using u2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, off, 0, BDDMMA_ST_AUX);
}
__device__ __forceinline__ void bstore(float2 v, rsrc_t r, uint32_t off)
{
    using u2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, off, 0, BDDMMA_ST_AUX);
}
__device__ __forceinline__ void bstore(double2 v, rsrc_t r, uint32_t off)
{
    using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, 0);
}

template <typename REAL>
struct NarrowRs {
    rsrc_t words, T, F, lohi, cse, css, dlay, mm;
    const uint32_t* cse_p;  // the staging tables as plain pointers (stage_load rebases them to its round)
    const uint16_t* css_p;
    const REAL* dlay_p;  // the entry arrays as plain pointers: instances whose arrays reach 4 GiB address them with 64 bits (DevPtrs::big)
    REAL* mm_p;
    bool big;
    __device__ __forceinline__ explicit NarrowRs(const DevPtrs<REAL>& d)
    {
        cse_p = d.cs_entry;
        css_p = d.cs_slot;
        dlay_p = d.delta_lay;
        mm_p = d.mm_binned;
        big = d.big != 0;
        words = make_rsrc(d.nwords, d.n_nwords);
        T = make_rsrc(d.T, d.n_slots);
        F = make_rsrc(d.F, d.n_slots);
        lohi = make_rsrc(d.lohi, 2ull * d.n_layers);
        cse = make_rsrc(d.cs_entry, d.n_narrow_layers);
        css = make_rsrc(d.cs_slot, d.n_narrow_layers);
        dlay = make_rsrc(d.delta_lay, 2ull * d.n_layers);
        mm = make_rsrc(d.mm_binned, d.n_layers);
    }
    // {lo, hi} from the pack's first layer on (see HopWindow: layer indices in the sweeps are relative to it)
    __device__ __forceinline__ void rebase_layers(const DevPtrs<REAL>& d, uint32_t l0) { lohi = make_rsrc(d.lohi + 2 * (size_t)l0, 2ull * (d.n_layers - l0)); }
};

// Cooperative stage transfer between the entry arrays and LDS: the WPB waves of a workgroup sweep WPB
// consecutive packs; in every round they load the delta pairs of their packs' stage groups together.
// Staged items are sorted by entry index, so consecutive threads touch consecutive entries — runs of
// (bin, quad) instead of (bin, pack) length — and scatter them to the owning wave's LDS slots.  The
// (entry, slot) pairs stay in registers for the write-back of the min-marginal differences.
constexpr int STAGE_ITERS = 10;  // stage_cap <= 64 * STAGE_ITERS

// (two halves, so that a kernel that knows its round's item range early — resident headers — can have the tables on their way while it sets
// up its pipeline: stage_load_tables issues the table loads, stage_load_pairs the dependent pair loads and the scatter into LDS)
template <typename REAL, int WPB, int AUX = BDDMMA_LD_TAB_AUX>
__device__ __forceinline__ void stage_load_tables(uint32_t (&e)[STAGE_ITERS], uint32_t (&sl)[STAGE_ITERS], const NarrowRs<REAL>& rs, uint32_t c0, uint32_t cnt,
                                                  uint32_t tid)
{
    // (the round's range of the staging tables, rebased: item offsets stay small whatever the tables' size)
    const rsrc_t rce = make_rsrc(rs.cse_p + c0, cnt), rcs = make_rsrc(rs.css_p + c0, cnt);
#pragma unroll
    for (int u = 0; u < STAGE_ITERS; ++u) {
        const uint32_t i = 64 * WPB * u + tid;
        e[u] = __builtin_amdgcn_raw_buffer_load_b32(rce, i * 4u, 0, AUX);   // past the round: dropped
        sl[u] = __builtin_amdgcn_raw_buffer_load_b16(rcs, i * 2u, 0, AUX);
    }
}
template <typename REAL, int WPB>
__device__ __forceinline__ void stage_load_pairs(typename Pair<REAL>::type* sD, const uint32_t (&e)[STAGE_ITERS], const uint32_t (&sl)[STAGE_ITERS],
                                                 const NarrowRs<REAL>& rs, uint32_t cnt, uint32_t tid)
{
    using P2 = typename Pair<REAL>::type;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        P2 v[STAGE_ITERS / 2];
        if (rs.big) {  // uniform: 64-bit addresses (entry * 8 or 16 bytes does not fit the 32-bit buffer offset); unused slots read entry 0
#pragma unroll
            for (int u = 0; u < STAGE_ITERS / 2; ++u) {
                const int k = half * (STAGE_ITERS / 2) + u;
                const uint32_t i = 64 * WPB * k + tid;
                v[u] = reinterpret_cast<const P2*>(rs.dlay_p)[i < cnt ? e[k] : 0u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < STAGE_ITERS / 2; ++u) {
                const int k = half * (STAGE_ITERS / 2) + u;
                const uint32_t i = 64 * WPB * k + tid;
                bload(v[u], rs.dlay, i < cnt ? e[k] * (uint32_t)sizeof(P2) : OOB);
            }
        }
#pragma unroll
        for (int u = 0; u < STAGE_ITERS / 2; ++u) {
            const int k = half * (STAGE_ITERS / 2) + u;
            const uint32_t i = 64 * WPB * k + tid;
            if (i < cnt) sD[sl[k]] = v[u];
        }
    }
}
template <typename REAL, int WPB, int AUX = BDDMMA_LD_TAB_AUX>
__device__ __forceinline__ void stage_load(typename Pair<REAL>::type* sD, uint32_t (&e)[STAGE_ITERS], uint32_t (&sl)[STAGE_ITERS],
                                           const NarrowRs<REAL>& rs, uint32_t c0, uint32_t cnt, uint32_t tid)
{
    stage_load_tables<REAL, WPB, AUX>(e, sl, rs, c0, cnt, tid);
    stage_load_pairs<REAL, WPB>(sD, e, sl, rs, cnt, tid);
}

template <typename REAL, int WPB>
__device__ __forceinline__ void stage_flush(const typename Pair<REAL>::type* sD, const uint32_t (&e)[STAGE_ITERS],
                                            const uint32_t (&sl)[STAGE_ITERS], const NarrowRs<REAL>& rs, uint32_t cnt, uint32_t tid)
{
    if (rs.big) {
#pragma unroll
        for (int u = 0; u < STAGE_ITERS; ++u) {
            const uint32_t i = 64 * WPB * u + tid;
            if (i < cnt) rs.mm_p[e[u]] = sD[sl[u]].x;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < STAGE_ITERS; ++u) {
        const uint32_t i = 64 * WPB * u + tid;
        const REAL m = sD[i < cnt ? sl[u] : 0].x;
        bstore(m, rs.mm, i < cnt ? e[u] * (uint32_t)sizeof(REAL) : OOB);
    }
}

}  // namespace bddmma
