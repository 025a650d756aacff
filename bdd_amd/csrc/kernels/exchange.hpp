// kernels/exchange.hpp — device-resident run_solver control, the variable <-> layer exchange (k_exchange_reduce, k_exchange_seg, k_exchange_byvar, k_delta_gather ...).
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// small elementwise / per-variable kernels
// =============================================================================================

// Device-resident run_solver (include/run_solver_util.h:40-73).  The reference's loop is iteration(); lower_bound(); three tests on the
// bound — a host round trip per iteration that leaves the GPU idle between the reduce kernel and the next forward sweep (119 -> 135 us
// per iteration at 10.5 M nodes, 33 -> 47 us at 1 M).  Here workgroup 0 of the exchange launch that ends an iteration also reduces the per-pack bounds the
// backward sweep has just written and runs the tests, in the reference's order and in the same double arithmetic, and latches `stop`:
// the launches of the iterations the host queued ahead see it and return (DevPtrs::stop), so the solver state is exactly the one
// after the iteration that met the criterion.  No extra launch, no synchronisation; the host only watches `RunHost` (pinned) for
// the bounds to print and for the end.  The wall-clock limit stays a host test.
constexpr uint32_t RUN_RING = 256;
struct RunCtl {  // device memory
    double lb_initial, lb_first, lb_post, tolerance, slope;
    double time_limit;   // seconds since the start of run_solver (run_solver_util.h:45-55); tested on the device's constant 100 MHz clock
    uint64_t t0;         // s_memrealtime at the start of the run (minus what the host had already spent), set by k_run_begin
    uint64_t iter;
    uint32_t stop, reason;
};
constexpr double RUN_TICKS_PER_SECOND = 1e8;  // s_memrealtime
static __global__ void k_run_begin(RunCtl* ctl, uint64_t host_ticks_so_far)
{
    ctl->t0 = __builtin_amdgcn_s_memrealtime() - host_ticks_so_far;
}
struct RunHost {  // pinned host memory, written by the device
    uint64_t state;       // (iterations whose bound has been published) | (stop reason << 56): one word, so the host never sees half an update
    double lb[RUN_RING];  // bound after iteration i at [i % RUN_RING]
};
struct RunStep {  // argument of the launch that ends an iteration (ctl == nullptr: nothing to do)
    const double* part;  // per-pack lower bounds
    uint32_t n;
    RunCtl* ctl;
    RunHost* host;
};
// The tests of run_solver_util.h:45-73 on the bound `t` of the iteration that has just ended, by ONE thread; `c` = the control block as it was
// before the iteration.  Publishes the bound, advances the iteration count, latches `stop`.  Returns the stop reason (0: go on).
// `publish` = false (k_iterate_small between the iterations of one launch): the bound goes to the ring but the host's view of the iteration
// count waits for a later call — the one that stops the run or ends the launch publishes everything (one system-scope fence for all).
__device__ __forceinline__ uint32_t run_ctl_finish(const RunStep& r, const RunCtl& c, double t, bool publish = true)
{
    RunCtl* ctl = r.ctl;
    const uint64_t it = c.iter;
    const double lb_prev = c.lb_post, lb_post = t;
    const double lb_first = it == 0 ? lb_post : c.lb_first, lb_initial = c.lb_initial;
    if (it == 0) ctl->lb_first = lb_post;
    ctl->lb_post = lb_post;
    ctl->iter = it + 1;
    uint32_t reason = 0;
    // the wall-clock limit first, as the reference tests it (:45-55) — on the device, so that no iteration queued behind the one that crossed
    // the limit runs (the host-side test let up to window - 1 more iterations execute)
    const double time_spent = (double)(__builtin_amdgcn_s_memrealtime() - c.t0) / RUN_TICKS_PER_SECOND;
    if (time_spent > c.time_limit) reason = 1;
    else if (__builtin_fabs(lb_prev - lb_post) < __builtin_fabs(c.tolerance * lb_prev)) reason = 2;           // run_solver_util.h:56-61
    else if (__builtin_fabs(lb_prev - lb_post) < c.slope * __builtin_fabs(lb_initial - lb_first)) reason = 3;  // :62-67
    else if (lb_post == __builtin_huge_val()) reason = 4;                                                       // :68-73
    if (reason) { ctl->reason = reason; ctl->stop = (uint32_t)(it + 1 < (uint64_t)RUN_NOT_STOPPED ? it + 1 : (uint64_t)RUN_NOT_STOPPED - 1); }  // launches of iterations >= it + 1 are skipped
    volatile RunHost* h = r.host;
    h->lb[it % RUN_RING] = lb_post;
    if (publish || reason) {
        __threadfence_system();
        h->state = (it + 1) | ((uint64_t)reason << 56);
    }
    return reason;
}
// Executed by every thread of ONE workgroup of 256, 512 or 1024 threads.  The sum has the shape and order of k_lb_reduce (1024
// threads: 16 waves of strided partial sums, an in-wave tree, the 16 results added in order) whatever the workgroup size, so the
// published bound equals lower_bound() bit for bit.
__device__ __forceinline__ void run_ctl_step(const RunStep& r)
{
    __shared__ double run_red[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // the control block is read while the partial sums are on their way (one round trip instead of two)
    RunCtl c{};
    if (threadIdx.x == 0) c = *r.ctl;
    for (uint32_t vw = wave; vw < 16; vw += nw) {
        double acc = 0.0;
        uint32_t i = vw * 64 + lane;
        for (; i + 7 * 1024u < r.n; i += 8 * 1024u) {  // eight loads in flight, additions in the plain loop's order (see k_lb_reduce)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = r.part[i + u * 1024u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < r.n; i += 1024) acc += r.part[i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) run_red[vw] = acc;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t = 0.0;
    for (uint32_t i = 0; i < 16; ++i) t += run_red[i];
    (void)run_ctl_finish(r, c, t);
}

template <typename REAL>
__device__ __forceinline__ void lds_add(REAL* p, REAL v)
{
#ifdef BDDMMA_EXP_INT_ATOMIC  // timing experiment only (wrong results).  NOT the rate of integer atomics, as rounds 4-5 first read it: the garbage sums make
    // every cost NaN, the sweeps then defer 0 everywhere and the exchange skips ALL its atomics — it times the exchange without accumulation (tools/exp_r05_j.sh)
    using U = typename std::conditional<sizeof(REAL) == 8, unsigned long long, unsigned int>::type;
    __hip_atomic_fetch_add(reinterpret_cast<U*>(p), __builtin_bit_cast(U, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f32 / ds_add_f64
#endif
}

// Exchange kernel: one workgroup per bin of variables; the bin's 2*vars_per_bin accumulators live in LDS.
//   EX_ITER : compute_delta (bdd_cuda_parallel_mma.cu:358-393) + normalize_delta (:410-430) + broadcast of the
//             normalised pairs to the bin's entries (what the next sweep adds, :191-197); the per-variable
//             result is also stored in delta_var (the solver's delta_lo_hi_).
//   EX_RAW  : compute_delta only — un-normalised sums into delta_var (explicit forward_mm / backward_mm API).
// The accumulators are ACC-typed: LDS f32 atomics (ds_add_f32) run at half the rate of ds_add_f64 on
// gfx950 (measured: 33 us vs 16 us for the same 5 M entries), so float solvers accumulate in double
// and round once per variable.
enum : int { EX_ITER = 0, EX_RAW = 1 };
constexpr int EX_THREADS = 1024;
constexpr int EX_UNROLL = 12;  // entries per thread and chunk: a bin of <= 24576 entries is one chunk — every load of the bin in flight at once,
                               // and the local variable indices stay in registers for the broadcast (no second round trip)
constexpr int EX_NPT = 19;     // 2 * vars_per_bin <= EX_NPT * EX_THREADS
// Small instances (few hundred bins of 1024 threads would leave most CUs idle and pay 16-wave barriers for a handful of entries per
// thread): the same kernel with 256-thread workgroups over bins of <= 1024 variables.
constexpr int EXS_THREADS = 256;
constexpr int EXS_UNROLL = 12;
constexpr int EXS_NPT = 8;
constexpr uint32_t EXS_MAX_VARS_PER_BIN = EXS_THREADS * EXS_NPT / 2;
// Bins of <= 2048 variables: 512-thread workgroups.  The 1024-thread kernel holds ~100 VGPRs per lane, i.e. ONE workgroup per CU, so
// with more bins than CUs its workgroups run in rounds, each paying the whole latency chain (bin range -> loads -> accumulate ->
// normalise -> broadcast); two 512-thread workgroups per CU overlap one bin's broadcast with the other's loads.
constexpr int EXM_THREADS = 512;
constexpr int EXM_UNROLL = 12;
constexpr int EXM_NPT = 8;
constexpr uint32_t EXM_MAX_VARS_PER_BIN = EXM_THREADS * EXM_NPT / 2;

// (History: round 2 shelved the scalar-offset form of this kernel because ~1 % of the differential fuzz runs came out with 1e-7 errors when
// several processes shared the GPU; round 3 bisected it to a hardware write-data hazard of 16-byte buffer stores with an SGPR soffset that
// the compiler does not guard — hop_store(double2) below, profiles/r03_exchange_variant_rootcause.txt — and made it the only form.)
// pair stores with the chunk's first entry in the scalar offset
#ifndef BDDMMA_EX_ST_AUX   // cache policy of the broadcast pairs' stores: a build knob (tools/exp_r06_ex_nt.sh)
#define BDDMMA_EX_ST_AUX BDDMMA_ST_AUX
#endif
__device__ __forceinline__ void hop_store(float2 v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    using u2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(rh, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rh, voff, soff, BDDMMA_EX_ST_AUX);
}
// 16-byte store with an SGPR soffset: on gfx950 a VMEM store of more than 64 bits needs one wait state before a VALU instruction
// overwrites its data registers — also when soffset is an SGPR, which the ISA manuals exempt and the compiler therefore does not pad
// (LLVM GCNHazardRecognizer::createsVALUHazard).  Without the s_nop 1.1 % of such pairs store the overwritten first dword
// (tools/store_hazard.hip, profiles/r03_exchange_variant_rootcause.txt): this was the round-2 exchange rewrite's "rare 1e-7 error".
// -DBDDMMA_REPRODUCE_STORE_HAZARD builds without it (tools/build_variant.sh), for the record only; tests/test_isa_lint.py checks the
// built library for unpadded pairs.
__device__ __forceinline__ void hop_store(double2 v, rsrc_t rh, uint32_t voff, uint32_t soff)
{
    using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(rh, 0, 0, 0));
    const u4 data = __builtin_bit_cast(u4, v);
    __builtin_amdgcn_raw_buffer_store_b128(data, rh, voff, soff, BDDMMA_EX_ST_AUX);
#ifndef BDDMMA_REPRODUCE_STORE_HAZARD
    // the data registers are an input of the nop: they stay live up to it, so no VALU write of them can be scheduled between the store and
    // the wait state (the Makefile runs tools/isa_lint.py on every build)
    asm volatile("s_nop 0" ::"v"(data) : "memory");
#endif
}
// Entry addressing: a lane's offset is tid * size with the chunk's first entry in the scalar offset, the descriptors end at the bin's last
// entry (lanes past it drop out by themselves); one predicated LDS atomic per entry (slot 2 v + [mm > 0], value |mm|).
template <typename REAL, typename ACC, int MODE, int EX_THREADS, int EX_UNROLL, int NPT>
__device__ __forceinline__ bool exchange_reduce_body(const REAL* __restrict__ mm_binned, const uint32_t* __restrict__ bin_ptr,
                                                     const uint16_t* __restrict__ bvar, const int32_t* __restrict__ nbdds,
                                                     REAL* __restrict__ delta_var, REAL* __restrict__ delta_lay,
                                                     uint32_t vars_per_bin, uint32_t n_vars,
                                                     uint32_t stop_word = RUN_NOT_STOPPED, uint32_t run_iter = 0)  // false: run_solver has stopped
{
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    ACC* tile = reinterpret_cast<ACC*>(dyn_lds);
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t v0 = b * vars_per_bin;
    const uint32_t nv = min(vars_per_bin, n_vars - v0);
    // the bin's ranges of the entry arrays, rebased (64-bit, once): entry offsets below are relative to the bin's first entry, so the arrays
    // may exceed the 4 GiB that a 32-bit buffer offset reaches
    const uint32_t e0_abs = bin_ptr[b], e1_abs = bin_ptr[b + 1];
    mm_binned += e0_abs;
    bvar += e0_abs;
    if (delta_lay) delta_lay += 2 * (size_t)e0_abs;
    const uint32_t e1 = e1_abs - e0_abs;  // entries of the bin
    BDDMMA_STAMP(0x100000u + blockIdx.x * (EX_THREADS / 64) + (tid >> 6), 0);
    const rsrc_t rmm = make_rsrc(mm_binned, e1), rev = make_rsrc(bvar, e1);
    const rsrc_t rnb = make_rsrc(nbdds, n_vars);
    const uint32_t vo_m = tid * (uint32_t)sizeof(REAL), vo_v = tid * 2u, vo_p = tid * (uint32_t)sizeof(P2);
    constexpr uint32_t CH = EX_THREADS * EX_UNROLL;
    const bool one_chunk = e1 <= CH;
    auto load_chunk = [&](REAL (&mm_)[EX_UNROLL], uint32_t (&lv_)[EX_UNROLL], uint32_t start) {  // `start` is uniform
#pragma unroll
        for (int u = 0; u < EX_UNROLL; ++u) {
            const uint32_t es = start + u * EX_THREADS;
#ifdef BDDMMA_EXP_EX_LD_AUX   // cache-policy experiments (tools/exp_r06_ex_nt.sh)
            if constexpr (sizeof(REAL) == 4) mm_[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmm, vo_m, es * 4u, BDDMMA_EXP_EX_LD_AUX));
            else mm_[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rmm, vo_m, es * 8u, BDDMMA_EXP_EX_LD_AUX));
            lv_[u] = __builtin_amdgcn_raw_buffer_load_b16(rev, vo_v, es * 2u, BDDMMA_EXP_EX_LD_AUX);
#else
            hop_load(mm_[u], rmm, vo_m, es * (uint32_t)sizeof(REAL));  // past the bin: 0 -> no contribution
            lv_[u] = __builtin_amdgcn_raw_buffer_load_b16(rev, vo_v, es * 2u, 0);
#endif
        }
    };
    // first chunk: every load of the workgroup is issued before anything is consumed
    REAL m[EX_UNROLL];
    uint32_t lv[EX_UNROLL];
    load_chunk(m, lv, 0);
    BDDMMA_STAMP(0x100000u + blockIdx.x * (EX_THREADS / 64) + (tid >> 6), 1);  // (waits for the first chunk here: the kernel itself does not)
    if (stop_word <= run_iter) return false;  // uniform for the grid; nothing has been written yet
    // number of BDDs of the variables this thread normalises (needed only after the accumulation)
    int nb[NPT];
    if (MODE == EX_ITER) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const uint32_t i = tid + k * EX_THREADS;
            nb[k] = (int)bload_u32(rnb, i < 2 * nv ? (v0 + (i >> 1)) * 4u : OOB);
        }
    }
    for (uint32_t i = tid; i < 2 * nv; i += EX_THREADS) tile[i] = ACC(0);
    __syncthreads();
    auto accumulate = [&](const REAL (&mm_)[EX_UNROLL], const uint32_t (&lv_)[EX_UNROLL]) {
#pragma unroll
        for (int u = 0; u < EX_UNROLL; ++u) {
            const REAL mv = mm_[u];
            const uint32_t slot = 2 * lv_[u] + (mv > 0 ? 1u : 0u);
            if (mv != 0) lds_add(&tile[slot], ACC(mv > 0 ? mv : -mv));
        }
    };
    {  // bins larger than one chunk: the loads of chunk c + 1 are in flight while chunk c is accumulated
        REAL mc[EX_UNROLL];
        uint32_t lc[EX_UNROLL];
        uint32_t cs = CH;  // start of the next chunk (uniform)
        bool have = cs < e1;
        if (have) load_chunk(mc, lc, cs);
        accumulate(m, lv);
        while (have) {
            REAL mn[EX_UNROLL];
            uint32_t ln[EX_UNROLL];
            const uint32_t ns = cs + CH;
            const bool more = ns < e1;
            if (more) load_chunk(mn, ln, ns);
            accumulate(mc, lc);
            if (!more) break;
#pragma unroll
            for (int u = 0; u < EX_UNROLL; ++u) { mc[u] = mn[u]; lc[u] = ln[u]; }
            cs = ns;
        }
    }
    __syncthreads();
    BDDMMA_STAMP(0x100000u + blockIdx.x * (EX_THREADS / 64) + (tid >> 6), 2);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const uint32_t i = tid + k * EX_THREADS;
        if (i < 2 * nv) {
            REAL x = REAL(tile[i]);
            if (MODE == EX_ITER) {
                x = nb[k] > 0 ? x / REAL(nb[k]) : REAL(0);
                tile[i] = ACC(x);
            }
            // EX_ITER leaves delta_var alone when the solver passes nullptr: the hot loop only needs the broadcast pairs,
            // and the per-variable copy (8 MB at V = 1 M) is rebuilt on demand by k_delta_var_from_lay
            if (MODE != EX_ITER || delta_var) delta_var[2 * (size_t)v0 + i] = x;
        }
    }
    if (MODE != EX_ITER) return true;
    __syncthreads();
    BDDMMA_STAMP(0x100000u + blockIdx.x * (EX_THREADS / 64) + (tid >> 6), 3);
    const rsrc_t rdl = make_rsrc(delta_lay, 2ull * e1);
#pragma unroll
    for (int u = 0; u < EX_UNROLL; ++u) {  // first chunk: the local variable indices are still in registers
        P2 pr;
        pr.x = REAL(tile[2 * lv[u]]);
        pr.y = REAL(tile[2 * lv[u] + 1]);
        hop_store(pr, rdl, vo_p, (u * EX_THREADS) * (uint32_t)sizeof(P2));
    }
    BDDMMA_STAMP(0x100000u + blockIdx.x * (EX_THREADS / 64) + (tid >> 6), 4);
    if (one_chunk) return true;
    for (uint32_t base = CH; base < e1; base += CH) {  // uniform
        uint32_t lv2[EX_UNROLL];
#pragma unroll
        for (int u = 0; u < EX_UNROLL; ++u) lv2[u] = __builtin_amdgcn_raw_buffer_load_b16(rev, vo_v, (base + u * EX_THREADS) * 2u, 0);
#pragma unroll
        for (int u = 0; u < EX_UNROLL; ++u) {
            P2 pr;
            pr.x = REAL(tile[2 * lv2[u]]);
            pr.y = REAL(tile[2 * lv2[u] + 1]);
            hop_store(pr, rdl, vo_p, (base + u * EX_THREADS) * (uint32_t)sizeof(P2));
        }
    }
    return true;
}

// The launch: `stop` (device-resident run_solver, DevPtrs::stop) makes it return at once when the termination test has fired; `run`
// (only on the launch that ends an iteration) makes workgroup 0 reduce the lower bound and run the tests after its bin is done —
// behind the body, where no register of the exchange is live any more (the 1024-thread double instantiation sits at its 128-VGPR limit).
// RUN = false is the kernel every other caller launches: `stop` and `run` are not looked at, the code is the body alone.
template <typename REAL, typename ACC, int MODE, int EX_THREADS = bddmma::EX_THREADS, int EX_UNROLL = bddmma::EX_UNROLL, int NPT = bddmma::EX_NPT,
          bool RUN = false>
__global__ void __launch_bounds__(EX_THREADS) k_exchange_reduce(const REAL* __restrict__ mm_binned, const uint32_t* __restrict__ bin_ptr,
                                                                  const uint16_t* __restrict__ bvar, const uint32_t* stop, uint32_t run_iter,
                                                                  uint32_t vars_per_bin, uint32_t n_vars, uint32_t n_entries,
                                                                  const int32_t* __restrict__ nbdds, REAL* __restrict__ delta_var,
                                                                  REAL* __restrict__ delta_lay, RunStep run = RunStep{})
{
    // argument order: what the first loads need comes first (the first 16 dwords of plain arguments are preloaded into SGPRs, see
    // RES_LEADING_ARGS); the stop word's load is issued at once and tested inside the body when the first chunk's loads are in flight
    const uint32_t stop_word = (RUN && stop != nullptr) ? *stop : RUN_NOT_STOPPED;
    // The launch that ends an iteration of run_solver has one workgroup more than bins: it adds up the per-pack bounds the backward sweep
    // has left and runs the termination tests while the others work on their bins (as the tail of workgroup 0, behind its bin, the
    // reduction's dependent round trips were the end of the launch: run_solver at 1.05 M nodes 36.3 -> 35.6 us per iteration with the stop
    // word's load overlapped, -> 33.3 us with the extra workgroup, the plain loop being 32.0; 10.5 M nodes 125.1 -> 122.3 us).
    if (RUN && run.ctl != nullptr && blockIdx.x == gridDim.x - 1) {  // uniform
        if (stop_word <= run_iter) return;
        run_ctl_step(run);
        return;
    }
    if (!exchange_reduce_body<REAL, ACC, MODE, EX_THREADS, EX_UNROLL, NPT>(mm_binned, bin_ptr, bvar, nbdds, delta_var, delta_lay, vars_per_bin, n_vars,
                                                                            stop_word, RUN ? run_iter : 0u))
        return;
}

// The binned exchange as a fixed schedule instead of LDS atomics (layout.hpp: struct SegExchange has the idea and the tables; round 5).
// One workgroup per bin, as k_exchange_reduce:
//   1. every load the workgroup needs is issued at once and depends on nothing but the bin's 16-byte header: the bin's deferred differences
//      (16-byte coalesced loads -> LDS, entry order), the thread's run (entry offsets of its positions, <= SEG_MAX_RUN u16 in <= 4 registers
//      quads) and its {end mask, first slot};
//   2. a thread walks its run: plain LDS reads, REAL sums in (variable, bdd) order — the order and the arithmetic of k_delta_gather, so the
//      result is bit-reproducible and equal to the `deterministic` path's —, at the last entry of a variable the pair and the entry count go
//      to the variable's slot;
//   3. the pairs are normalised (one division per value, slots spread over the threads), and every thread writes its slot numbers to its
//      entries' places (u16, over the differences, which nobody reads any more);
//   4. the broadcast streams entry -> slot -> pair -> delta_lay with the chunk's first entry in the scalar offset (hop_store).
// Three barriers, no atomics, no dependent global load behind the header.
// Measured (10.5 M nodes, float, rocprofv3 in sequence): 19.5 us per launch against the LDS-atomic kernel's 17.9 — the phases (loads 5.8 us at
// the chip's full rate, sums 4.0, normalise + slots 2.3, broadcast 3 + 4 of drain) do not overlap any more than the atomic kernel's do
// (profiles/r05_exchange.txt) —, so this is the `deterministic` exchange (it replaces k_delta_gather + k_exchange_bcast, two launches of
// gathers) and the LDS atomics stay the default.
constexpr int SEG_MAXL = 12;  // 16-byte loads of differences per thread: a bin holds <= SEG_MAXL * T * 16 / sizeof(REAL) entries
// G: 16-byte groups of run positions per thread (the largest bin's; the tables pad every run to it) — a template parameter so that every
// register array below is indexed by constants (with run-time group counts and early exits the arrays went to scratch memory)
template <typename REAL, int T, int G, bool RUN = false>
__global__ void __launch_bounds__(T) k_exchange_seg(const REAL* __restrict__ mm_binned, const uint4* __restrict__ seg_bin, const uint32_t* stop, uint32_t run_iter,
                                                      const uint4* __restrict__ seg_perm, const uint2* __restrict__ seg_thr, uint32_t tile_off, uint32_t cnt_off,
                                                      REAL* __restrict__ delta_lay, RunStep run = RunStep{})
{
    const uint32_t stop_word = (RUN && stop != nullptr) ? *stop : RUN_NOT_STOPPED;
    if (RUN && run.ctl != nullptr && blockIdx.x == gridDim.x - 1) {  // the extra workgroup of the launch that ends a run_solver iteration (see k_exchange_reduce)
        if (stop_word <= run_iter) return;
        run_ctl_step(run);
        return;
    }
    using P2 = typename Pair<REAL>::type;
    constexpr uint32_t VEC = 16 / sizeof(REAL);
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    REAL* mm_lds = reinterpret_cast<REAL*>(dyn_lds);           // [entries rounded up to VEC, + 1], then the entries' slot numbers
    uint16_t* slot_lds = reinterpret_cast<uint16_t*>(dyn_lds);
    P2* tile = reinterpret_cast<P2*>(dyn_lds + tile_off);       // [slots] {sum of -mm over mm < 0, sum of mm over mm > 0}
    uint16_t* cnt = reinterpret_cast<uint16_t*>(dyn_lds + cnt_off);  // [slots] entries of the slot's variable = its number of BDDs
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint4 hdr = seg_bin[b];  // first group of the bin in seg_perm, groups per thread | slots << 8, first entry, entries
    const uint32_t slots = hdr.y >> 8, E = hdr.w;
    mm_binned += hdr.z;
    delta_lay += 2 * (size_t)hdr.z;
    BDDMMA_STAMP(0x100000u + blockIdx.x * (T / 64) + (tid >> 6), 0);
    const rsrc_t rmm = make_rsrc(mm_binned, E);
    using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(rmm, 0, 0, 0));
    u4 buf[SEG_MAXL];
#pragma unroll
    for (int j = 0; j < SEG_MAXL; ++j)
        if ((uint32_t)j * T * VEC < E) buf[j] = __builtin_amdgcn_raw_buffer_load_b128(rmm, ((uint32_t)j * T + tid) * 16u, 0, 0);
    uint4 pw[G];
    const uint4* pg = seg_perm + hdr.x + tid;
#pragma unroll
    for (int g = 0; g < G; ++g) pw[g] = pg[(size_t)g * T];
    const uint2 th = seg_thr[(size_t)b * T + tid];
    if (stop_word <= run_iter) return;  // uniform for the grid; nothing has been written yet
#pragma unroll
    for (int j = 0; j < SEG_MAXL; ++j) {
        const uint32_t i = ((uint32_t)j * T + tid) * VEC;
        if ((uint32_t)j * T * VEC < E && i < E) *reinterpret_cast<u4*>(mm_lds + i) = buf[j];
    }
    const uint32_t Z = (E + VEC - 1) / VEC * VEC;  // the place the positions past a run's end point at: never written above
    if (tid == 0) mm_lds[Z] = REAL(0);
    __syncthreads();
    BDDMMA_STAMP(0x100000u + blockIdx.x * (T / 64) + (tid >> 6), 1);
    const uint32_t ends = th.x;
    // position k of the run -> entry offset
    uint32_t pl[8 * G];
#pragma unroll
    for (int k = 0; k < 8 * G; ++k) {
        const uint4 q = pw[k / 8];
        const uint32_t w = (k % 8) / 2 == 0 ? q.x : (k % 8) / 2 == 1 ? q.y : (k % 8) / 2 == 2 ? q.z : q.w;
        pl[k] = (k & 1) ? w >> 16 : w & 0xFFFFu;
    }
    uint32_t sl[8 * G];  // slot of the variable of position k (the slots are numbered along the runs)
    {
        // all the run's differences first (independent LDS reads in flight together), then the sums in order.  Branch-free: the sums of
        // compute_delta (bdd_cuda_parallel_mma.cu:358-393: hi += m if m > 0, lo += -m if m < 0) as hi += max(m, 0), lo += max(-m, 0) — adding
        // +0 changes nothing —, in k_delta_gather's order; at the last entry of a variable ((ends >> k) & 1) the pair and the count go to LDS
        // under the lane mask and the sums restart.
        REAL mv[8 * G];
#pragma unroll
        for (int k = 0; k < 8 * G; ++k) mv[k] = mm_lds[pl[k]];
        uint32_t slot = th.y, first = 0;
        REAL lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8 * G; ++k) {
            const REAL m = mv[k];
            hi += m > REAL(0) ? m : REAL(0);
            lo += m < REAL(0) ? -m : REAL(0);
            sl[k] = slot;
            const bool end = (ends >> k) & 1u;
            if (end) {
                P2 pr;
                pr.x = lo;
                pr.y = hi;
                tile[slot] = pr;
                cnt[slot] = (uint16_t)(k + 1 - first);
            }
            slot += end ? 1u : 0u;
            first = end ? (uint32_t)(k + 1) : first;
            lo = end ? REAL(0) : lo;
            hi = end ? REAL(0) : hi;
        }
    }
    __syncthreads();
    BDDMMA_STAMP(0x100000u + blockIdx.x * (T / 64) + (tid >> 6), 2);
    for (uint32_t i = tid; i < slots; i += T) {  // normalize_delta, :410-430
        P2 pr = tile[i];
        const REAL c = REAL(cnt[i]);
        pr.x /= c;
        pr.y /= c;
        tile[i] = pr;
    }
#pragma unroll
    for (int k = 0; k < 8 * G; ++k) slot_lds[pl[k]] = (uint16_t)sl[k];  // past the run's end: place Z, which no entry reads
    __syncthreads();
    BDDMMA_STAMP(0x100000u + blockIdx.x * (T / 64) + (tid >> 6), 3);
    const rsrc_t rdl = make_rsrc(delta_lay, 2ull * E);
    const uint32_t vo_p = tid * (uint32_t)sizeof(P2);
    for (uint32_t base = 0; base < E; base += 4 * T) {
        uint32_t sl[4];
        P2 pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = base + u * T + tid;
            sl[u] = e < E ? slot_lds[e] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = tile[sl[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * T < E) hop_store(pr[u], rdl, vo_p, (base + u * T) * (uint32_t)sizeof(P2));  // lanes past the bin's end: out of the descriptor's range
    }
    BDDMMA_STAMP(0x100000u + blockIdx.x * (T / 64) + (tid >> 6), 4);
}

// Exchange for entry arrays ordered by (variable, bdd) (layout.hpp: Exchange::entry_by_var): the entries of variable v are
// var_ptr[v] .. var_ptr[v + 1], so compute_delta (bdd_cuda_parallel_mma.cu:358-393), normalize_delta (:410-430) and the broadcast of
// the pair to the variable's layers are one thread per variable over a contiguous run — neighbouring threads read and write
// neighbouring addresses, there are no LDS accumulators and no barriers, and the sum has the fixed order of the reduce_by_key variant
// the reference keeps commented out (:395-407).  One dependent round trip (var_ptr) before the values instead of the binned kernel's
// chain of loads, LDS atomics and three workgroup barriers: 3.5 us instead of 9.7 us at 1 M nodes.
template <typename REAL>
__global__ void __launch_bounds__(256) k_exchange_byvar(const REAL* __restrict__ mm, const uint32_t* __restrict__ var_ptr,
                                                          REAL* __restrict__ delta_lay, uint32_t n_vars, uint32_t n_entries,
                                                          RunGate gate = RunGate{}, RunStep run = RunStep{})
{
    if (run_stopped(gate)) return;
    if (run.ctl != nullptr && blockIdx.x == 0) run_ctl_step(run);
    using P2 = typename Pair<REAL>::type;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const rsrc_t rvp = make_rsrc(var_ptr, (uint64_t)n_vars + 1), rmm = make_rsrc(mm, n_entries), rdl = make_rsrc(delta_lay, 2ull * n_entries);
    const uint32_t k0 = bload_u32(rvp, v < n_vars ? v * 4u : OOB), k1 = bload_u32(rvp, v < n_vars ? (v + 1) * 4u : OOB);
    const uint32_t n = k1 - k0;  // 0 for threads past the last variable and for variables in no BDD
    constexpr int J = 8;         // values requested together; variables in more BDDs continue one by one
    REAL m[J];
#pragma unroll
    for (int j = 0; j < J; ++j) bload(m[j], rmm, (uint32_t)j < n ? (k0 + j) * (uint32_t)sizeof(REAL) : OOB);  // dropped: 0
    REAL lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (m[j] > 0) hi += m[j];
        else if (m[j] < 0) lo += -m[j];
    }
    for (uint32_t j = J; j < n; ++j) {
        REAL x;
        bload(x, rmm, (k0 + j) * (uint32_t)sizeof(REAL));
        if (x > 0) hi += x;
        else if (x < 0) lo += -x;
    }
    if (n == 0) return;
    P2 pr;
    pr.x = lo / REAL(n);
    pr.y = hi / REAL(n);
#pragma unroll
    for (int j = 0; j < J; ++j) bstore(pr, rdl, (uint32_t)j < n ? (k0 + j) * (uint32_t)sizeof(P2) : OOB);
    for (uint32_t j = J; j < n; ++j) bstore(pr, rdl, (k0 + j) * (uint32_t)sizeof(P2));
}

// Exchange, step B: broadcast the per-variable pairs to every entry (what the next sweep adds to the
// arc costs, bdd_cuda_parallel_mma.cu:191-197).  Entries of one bin are contiguous, so the pairs of
// vars_per_bin consecutive variables are re-read from L1/L2 while the writes stream out coalesced.
// Four entries per thread: one 16-byte index load, four independent pair gathers, 16-byte stores.
template <typename REAL>
__global__ void __launch_bounds__(256) k_exchange_bcast(const REAL* __restrict__ delta_var, const uint32_t* __restrict__ evar,
                                                          REAL* __restrict__ delta_lay, uint32_t n_entries, uint32_t n_vars,
                                                          RunGate gate = RunGate{}, RunStep run = RunStep{})
{
    if (run_stopped(gate)) return;
    // the deterministic exchange is two launches (k_delta_gather, this one): the tests latch `stop` in the LAST launch of the iteration
    if (run.ctl != nullptr && blockIdx.x == 0) run_ctl_step(run);
    using P2 = typename Pair<REAL>::type;
    const uint32_t e = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= n_entries) return;
    const rsrc_t rdv = make_rsrc(delta_var, 2ull * n_vars);
    uint32_t v[4];
    if (e + 4 <= n_entries) {
        const uint4 vv = *reinterpret_cast<const uint4*>(evar + e);
        v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; v[3] = vv.w;
    } else {
        for (int u = 0; u < 4; ++u) v[u] = e + u < n_entries ? evar[e + u] : 0xFFFFFFFFu;
    }
    P2 pr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bload(pr[u], rdv, v[u] != 0xFFFFFFFFu ? v[u] * (uint32_t)sizeof(P2) : OOB);
    P2* out = reinterpret_cast<P2*>(delta_lay) + e;
    if (e + 4 <= n_entries) {
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u] = pr[u];
    } else {
        for (int u = 0; u < 4; ++u)
            if (e + u < n_entries) out[u] = pr[u];
    }
}

template <typename REAL>
__global__ void k_normalize_delta(REAL* __restrict__ delta, const int32_t* __restrict__ nbdds, uint32_t n2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int nb = nbdds[i >> 1];
    if (nb > 0) delta[i] /= REAL(nb);
}

// Deterministic alternative to the LDS atomics of k_exchange: per-variable gather over the
// (variable,bdd)-sorted entry list (the reduce_by_key variant commented out at bdd_cuda_parallel_mma.cu:395-407).
template <typename REAL, bool NORMALIZE>
__global__ void k_delta_gather(const REAL* __restrict__ mm_binned, const uint32_t* __restrict__ var_ptr,
                               const uint32_t* __restrict__ vpos, REAL* __restrict__ delta_var, uint32_t n_vars,
                               RunGate gate = RunGate{})
{
    if (run_stopped(gate)) return;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    REAL lo = 0, hi = 0;
    const uint32_t k0 = var_ptr[v], k1 = var_ptr[v + 1];
    for (uint32_t k = k0; k < k1; ++k) {
        const REAL m = mm_binned[vpos[k]];
        if (m > 0) hi += m;
        else if (m < 0) lo += -m;
    }
    if (NORMALIZE && k1 > k0) {
        lo /= REAL(k1 - k0);
        hi /= REAL(k1 - k0);
    }
    delta_var[2 * (size_t)v] = lo;
    delta_var[2 * (size_t)v + 1] = hi;
}

// delta_var[v] = the pair broadcast to the entries of v (any of them; 0 for a variable in no BDD)
template <typename REAL>
__global__ void k_delta_var_from_lay(const REAL* __restrict__ delta_lay, const uint32_t* __restrict__ var_ptr,
                                     const uint32_t* __restrict__ vpos, REAL* __restrict__ delta_var, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    const uint32_t k0 = var_ptr[v], k1 = var_ptr[v + 1];
    const size_t e = k1 > k0 ? vpos[k0] : 0;
    delta_var[2 * (size_t)v] = k1 > k0 ? delta_lay[2 * e] : REAL(0);
    delta_var[2 * (size_t)v + 1] = k1 > k0 ? delta_lay[2 * e + 1] : REAL(0);
}

// binned entry order <-> internal layer order (rare elementwise ops, checkpointing)
template <typename REAL>
__global__ void k_entries_to_layers(const REAL* __restrict__ binned, const uint32_t* __restrict__ lpos, REAL* __restrict__ out, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) out[l] = binned[lpos[l]];
}
template <typename REAL>
__global__ void k_layers_to_entries(const REAL* __restrict__ in, const uint32_t* __restrict__ lpos, REAL* __restrict__ binned, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) binned[lpos[l]] = in[l];
}

}  // namespace bddmma
