// kernels/small.hpp — instances that fit ONE workgroup: whole MMA iterations inside one launch (k_iterate_small).
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// =============================================================================================
// tiny instances: n iterations per launch, the whole solver state resident in one workgroup's LDS
// =============================================================================================
// An instance of a handful of packs (the 8 x 8 assignment problem of BASELINE.json configs[0] is ONE pack of 16 BDDs) is four dependent
// launches per iteration of which each is a few hundred nanoseconds of work behind ~4 us of launch, header, table and pair round trips:
// 18.5 us per iteration (profiles/r03_1m_latency.txt).  Between the workgroups of a grid a pass boundary costs what a launch costs
// (tools/gridsync.hip, round 5) — but when ALL packs fit one workgroup the boundary is a __syncthreads.  So here one workgroup copies the
// packs' costs-from-terminal, arc costs and staged delta pairs into LDS once, runs
//     forward solve sweep | exchange | backward solve sweep | exchange + bound + run_solver's tests
// n times out of LDS with barriers in between, and writes the state back once.  The sweeps are the hop loops of k_fwd_res2 / k_bwd_res2
// on the same per-lane records (same operations in the same order: bit-equal potentials, costs and deferred differences); the exchange is
// one thread per variable over the variable's entries in (variable, bdd) order, accumulated in the exchange tile's type (double) and
// rounded once, as k_exchange_reduce does — bit-equal to its LDS atomics whenever the double sum is exact (float instances) or has at
// most two terms (any variable in <= 2 BDDs), and within rounding of a different summation order otherwise.  run_solver's tests run in
// the kernel after every iteration (run_ctl_finish: the reference's order, the published bound, the latched stop): the launch ends with
// the iteration that met a criterion, so state and iteration count are the sequential loop's (run_solver_util.h:40-73).
// Preconditions (SolverT::small_ok): the second-generation resident records exist (packs of 64 slots, layers of <= 2 nodes, one stage
// group per pack, one staging round per quad), nothing but narrow packs, at most SMALL_MAX_PACKS of them, no L-BFGS wrapper attached
// (x_layer), not the deterministic / by-variable exchanges, everything within the CU's LDS.
constexpr uint32_t SMALL_MAX_PACKS = 16;
struct SmallDev {
    const uint32_t* pack_hdr;   // layout.hpp: struct Resident — 8 words per pack
    const uint32_t* quad_hdr;   // 4 words per quad of the staging tables: first item, items
    const uint32_t* rec;        // Res2Records
    const uint32_t* rec_off;
    uint32_t rec_words;
    uint32_t ns, nl;            // slot / layer capacity of a pack's LDS region (res2_wave_bytes)
    uint32_t n_packs, n_quads, wpb;  // wpb: packs per quad (the staging tables' LDS slots are quad-relative)
    const uint32_t* var_ptr;    // [n_vars + 1]: the variable's entries in var_lds / var_ent, (variable, bdd) order
    const uint32_t* var_lds;    // byte offset of the entry's staging pair inside the staging area
    const uint32_t* var_ent;    // the entry's index in mm_binned / delta_lay
    uint32_t n_vars, n_entries;
    uint32_t rec_cap;           // bytes of a pack's records kept in LDS (RL instantiation): 1 KiB per hop of the longest pack + one spare
    uint32_t off_regions, off_vp, off_vl, off_mm, off_misc, off_rec;  // byte offsets inside the dynamic LDS: pack regions | var_ptr copy | var_lds copy |
                                                                      // the last backward sweep's differences by (variable, bdd) | bounds + flag | records
};
// (rec_cap = 0: the records stay in global memory)
// A pack's staging area and its {lo, hi} array carry ONE pair more than the nl the records address: the dummy pair at offset nl * 2 S that the
// records rewritten in LDS (RL instantiation) point padding lanes and non-head lanes at, so that the hop's stores need no exec masks.
__host__ __device__ inline uint32_t small_stage_stride(uint32_t nl) { return nl + 1u; }   // pairs per pack in the staging area
__host__ __device__ inline uint32_t small_region_bytes(uint32_t real_size, uint32_t ns, uint32_t nl) { return res2_wave_bytes(real_size, ns, nl) + 2u * real_size; }
inline uint32_t small_lds_bytes(uint32_t real_size, uint32_t n_packs, uint32_t n_vars, uint32_t n_entries, SmallDev& sm)
{
    const uint32_t ns = sm.ns, nl = sm.nl;
    uint32_t *off_regions = &sm.off_regions, *off_vp = &sm.off_vp, *off_vl = &sm.off_vl, *off_misc = &sm.off_misc;
    uint32_t o = n_packs * small_stage_stride(nl) * 2u * real_size;   // staging area: a {lo, hi} pair per layer slot, nl (+ the dummy) per pack (the sweeps' own staging reserves stage_cap)
    o = (o + 15u) & ~15u;
    *off_regions = o;
    o += n_packs * small_region_bytes(real_size, ns, nl);
    o = (o + 15u) & ~15u;
    *off_vp = o;
    o += (n_vars + 1u) * 4u;
    o = (o + 15u) & ~15u;
    *off_vl = o;
    o += n_entries * 4u;
    o = (o + 15u) & ~15u;
    sm.off_mm = o;
    o += n_entries * real_size;
    o = (o + 15u) & ~15u;
    *off_misc = o;
    o += (SMALL_MAX_PACKS + 2u) * 8u;
    o = (o + 15u) & ~15u;
    sm.off_rec = o;
    o += n_packs * sm.rec_cap;
    return o;
}

// NW waves, pack p on wave p (n_packs <= NW); RL: the packs' records live in LDS too (1 KiB per hop), else they stream from L2 eight hops ahead
template <typename REAL, int NW, bool RL>
__global__ void __launch_bounds__(64 * NW) k_iterate_small(SmallDev sm, DevPtrs<REAL> d, PackDev pk, REAL omega, uint32_t n_iters, RunStep run)
{
    constexpr uint32_t S = sizeof(REAL);
    constexpr uint32_t NT = 64 * NW;
    using P2 = typename Pair<REAL>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    // run_solver: launches queued behind the iteration that met a criterion do nothing (DevPtrs::stop / run_iter = this launch's first iteration)
    if (d.stop != nullptr && *d.stop <= d.run_iter) return;
    const REAL INF = inf_v<REAL>();
    const uint32_t region = small_region_bytes(S, sm.ns, sm.nl), dstride = small_stage_stride(sm.nl);
    const uint32_t ll_dummy = sm.nl * 2u * S;   // the dummy pair of a pack's {lo, hi} array and of its staging area
    uint32_t* const vp = reinterpret_cast<uint32_t*>(dyn_lds + sm.off_vp);
    uint32_t* const vl = reinterpret_cast<uint32_t*>(dyn_lds + sm.off_vl);
    double* const lbp = reinterpret_cast<double*>(dyn_lds + sm.off_misc);          // per-pack bounds of the iteration
    uint32_t* const flag = reinterpret_cast<uint32_t*>(dyn_lds + sm.off_misc + SMALL_MAX_PACKS * 8u);  // [0]: stop reason of the iteration
    const rsrc_t rr = make_rsrc(sm.rec, sm.rec_words);

    REAL* const mmE = reinterpret_cast<REAL*>(dyn_lds + sm.off_mm);
    // ---- this wave's pack: everything the sweeps need from the headers, once
    const uint32_t p = (uint32_t)wave;
    const bool has_pack = p < sm.n_packs;
    const uint32_t* hp = sm.pack_hdr + 8 * (size_t)(has_pack ? p : 0);
    const uint32_t slot0 = hp[0], layer0 = hp[2];
    const uint32_t nslots = has_pack ? hp[1] : 0, nlayers = has_pack ? hp[3] : 0, nh = has_pack ? (hp[5] & 0xFFFFu) : 0;
    const uint32_t rbase = sm.rec_off[has_pack ? p : 0];
    const uint32_t db = p * dstride * (uint32_t)sizeof(P2);
    const uint32_t wb = sm.off_regions + p * region, wbF = wb + res2_f_off(S, sm.ns), wbC = wb + res2_c_off(S, sm.ns);
    const uint32_t rl = sm.off_rec + p * sm.rec_cap, rlane = rl + (uint32_t)lane * 16u;
    const uint32_t nhm1 = nh ? nh - 1u : 0u;
    // ---- prologue: the state -> LDS
    if (has_pack) {
        wave_copy_to_lds(d.T + slot0, dyn_lds + wb, nslots * S, lane);
        wave_copy_to_lds(d.lohi + 2 * (size_t)layer0, dyn_lds + wbC, nlayers * (uint32_t)sizeof(P2), lane);
        if (RL) {
            // hop h's 64 records: 1 KiB at rl + 1024 h — REWRITTEN on the way so that the hop loop needs no exec masks: padding lanes read and
            // write the dummy pair and the spare entry ns + 2 of the costs-from-terminal (nothing reads either); both lanes of a two-node
            // layer store the layer's new arc costs (the same pair to the same address — in LDS a duplicate costs nothing, the head-only
            // rule of the streaming sweeps saves global bandwidth).  A padding lane is then the lane whose pair offset is the dummy's;
            // .w keeps only the two-node flag, as the sign.
            for (uint32_t h = 0; h < nh; ++h) {
                u4v r = __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + h * 64u) * 16u, 0);
                if (r[3] == RES2_PAD) {
                    r[2] = ll_dummy | (((sm.ns + 2u) * S) << 16);
                    r[3] = 0u;
                } else {
                    r[3] = (r[3] & 0x10000u) ? 0x80000000u : 0u;   // the two-node flag as the sign: one compare in the hop
                }
                lds_st<u4v>(dyn_lds, rl + h * 1024u + (uint32_t)lane * 16u, r);
            }
        }
    }
    for (uint32_t i = tid; i <= sm.n_vars; i += NT) vp[i] = sm.var_ptr[i];
    for (uint32_t i = tid; i < sm.n_entries; i += NT) vl[i] = sm.var_lds[i];
    {   // the staged delta pairs of every quad: entry -> its layer's slot of the staging area (quad-relative slots, as the sweeps' own staging)
        P2* const sD = reinterpret_cast<P2*>(dyn_lds);
        for (uint32_t q = 0; q < sm.n_quads; ++q) {
            const uint32_t c0 = sm.quad_hdr[4 * (size_t)q], cnt = sm.quad_hdr[4 * (size_t)q + 1];
            for (uint32_t i = tid; i < cnt; i += NT) {
                const uint32_t e = d.cs_entry[c0 + i], sl = d.cs_slot[c0 + i];
                sD[(size_t)(q * sm.wpb + sl / pk.stage_cap) * dstride + sl % pk.stage_cap] = reinterpret_cast<const P2*>(d.delta_lay)[e];   // quad-relative slot -> (pack, layer)
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the direct-to-LDS copies have landed
    if (has_pack && lane < 2) lds_st<REAL>(dyn_lds, wb + (sm.ns + (uint32_t)lane) * S, lane == 0 ? REAL(0) : INF);  // sinks: cost to terminal 0 (top) / +inf (bot)
    RunCtl c{};
    if (run.ctl != nullptr && tid == 0) c = *run.ctl;
    __syncthreads();

    // The exchange (compute_delta + normalize_delta + the broadcast to the variable's layers, bdd_cuda_parallel_mma.cu:358-430,191-197): a
    // thread per variable; the layer slots hold the sweeps' min-marginal differences in .x and receive the variable's normalised pair.
    // `save_mm` (the exchange behind a backward sweep): the differences are kept by (variable, bdd) for the epilogue — they are the
    // solver's deferred differences (get_solver_costs / net_solver_costs) if this turns out to be the launch's last iteration.
    // A thread's first variable (v = tid) with at most four entries is served from registers: its entries' addresses are read once, before the
    // iterations, so an exchange is one LDS round trip (the differences), the sums, the pair, the stores.  Other variables: the generic loop,
    // eight entries per batch of independent loads.  Same order of the additions in both: (variable, bdd).
    // sum / (number of BDDs), rounded once like k_exchange_reduce's division: a count that is a power of two (every variable of an assignment
    // problem sits in two BDDs) divides exactly by an exponent shift — one instruction instead of the ~10 of a correctly rounded division
    auto mean_pair = [&](double lo, double hi, uint32_t n) -> P2 {
        P2 pr;
        if ((n & (n - 1u)) == 0u) {
            const int sh = -(int)__builtin_ctz(n);
            if constexpr (sizeof(REAL) == 4) { pr.x = __builtin_ldexpf(REAL(lo), sh); pr.y = __builtin_ldexpf(REAL(hi), sh); }
            else { pr.x = __builtin_ldexp(REAL(lo), sh); pr.y = __builtin_ldexp(REAL(hi), sh); }
        } else {
            const REAL nb = REAL(n);
            pr.x = REAL(lo) / nb;
            pr.y = REAL(hi) / nb;
        }
        return pr;
    };
    const uint32_t v_own = tid;
    uint32_t own_k0 = 0, own_n = 0, own_a[4] = {0u, 0u, 0u, 0u};
    if (v_own < sm.n_vars) {
        own_k0 = vp[v_own];
        own_n = vp[v_own + 1] - own_k0;
#pragma unroll
        for (int j = 0; j < 4; ++j) own_a[j] = vl[own_k0 + ((uint32_t)j < own_n ? j : 0)];   // past the last entry: the first again (see the exchange)
    }
    auto exchange = [&](bool save_mm) {
        auto generic = [&](uint32_t v) {
            const uint32_t k0 = vp[v], k1 = vp[v + 1];
            if (k1 == k0) return;
            double lo = 0.0, hi = 0.0;
            for (uint32_t kb = k0; kb < k1; kb += 8) {
                uint32_t a[8];
                REAL m[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = kb + u < k1 ? vl[kb + u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u) m[u] = lds_ld<REAL>(dyn_lds, a[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (kb + u < k1) {
                        if (m[u] > 0) hi += (double)m[u];
                        else if (m[u] < 0) lo += (double)(-m[u]);
                        if (save_mm) mmE[kb + u] = m[u];
                    }
                }
            }
            const P2 pr = mean_pair(lo, hi, k1 - k0);
            for (uint32_t kb = k0; kb < k1; kb += 8) {
                uint32_t a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = kb + u < k1 ? vl[kb + u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (kb + u < k1) lds_st<P2>(dyn_lds, a[u], pr);
            }
        };
        if (own_n != 0 && own_n <= 4) {
            REAL m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = lds_ld<REAL>(dyn_lds, own_a[j]);
            // without branches: an entry past the variable's last reads its first entry's slot again and counts as 0 — adding +0.0 changes no
            // sum, a NaN difference adds nothing either way (max(NaN, 0) = 0 as `NaN > 0` is false) — and stores the pair there once more
            double lo = 0.0, hi = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const REAL x = (uint32_t)j < own_n ? m[j] : REAL(0);
                if constexpr (sizeof(REAL) == 4) { hi += (double)__builtin_fmaxf(x, 0.0f); lo += (double)__builtin_fmaxf(-x, 0.0f); }
                else { hi += __builtin_fmax(x, 0.0); lo += __builtin_fmax(-x, 0.0); }
                if (save_mm && (uint32_t)j < own_n) mmE[own_k0 + j] = m[j];
            }
            const P2 pr = mean_pair(lo, hi, own_n);
#pragma unroll
            for (int j = 0; j < 4; ++j) lds_st<P2>(dyn_lds, own_a[j], pr);
        } else if (own_n > 4) {
            generic(v_own);
        }
        for (uint32_t v = tid + NT; v < sm.n_vars; v += NT) generic(v);
    };

    uint32_t it = 0;
    for (; it < n_iters; ++it) {
        // ---------------------------------------------------------------- forward solve sweep (k_fwd_res2's hop, state in LDS)
#ifdef BDDMMA_EXP_SMALL_SKIP   // timing experiments only (wrong results): bit 0 forward sweep, 1 first exchange, 2 backward sweep, 3 second exchange, 4 bound
        if (has_pack && !(BDDMMA_EXP_SMALL_SKIP & 1)) {
#else
        if (has_pack) {
#endif
            for (uint32_t o = (uint32_t)lane; o < sm.ns + 64u; o += 64u) lds_st<REAL>(dyn_lds, wbF + o * S, INF);  // costs-from-root and the dummy entries
            wave_sync();
            auto hop = [&](const u4v& r) {
                const uint32_t ll = r[2] & 0xFFFFu, fs = r[2] >> 16;
                const REAL f = lds_ld<REAL>(dyn_lds, wbF + fs);
                const REAL tl = lds_ld<REAL>(dyn_lds, wb + (r[0] & 0xFFFFu)), th = lds_ld<REAL>(dyn_lds, wb + (r[0] >> 16));
                const P2 cc = lds_ld<P2>(dyn_lds, wbC + ll);
                const P2 dd = lds_ld<P2>(dyn_lds, db + ll);
                REAL m0 = (f + cc.x) + tl, m1 = (f + cc.y) + th;  // padding lanes: +inf
                pair_min_aligned(m0, m1, RL ? (int32_t)r[3] < 0 : (r[3] & 0x10000u) != 0);
                const REAL mm = mm_diff1(m0, m1, omega);
                P2 nc;
                nc.x = (cc.x + min0(mm)) + dd.x;
                nc.y = (cc.y + min0_neg(mm)) + dd.y;
                if constexpr (RL) {   // rewritten records: every lane stores (both lanes of a layer the same pair; padding lanes the dummy pair)
                    lds_st<P2>(dyn_lds, wbC + ll, nc);
                    lds_st<REAL>(dyn_lds, db + ll, mm);
                } else {
                    const bool real = r[3] != RES2_PAD;
                    if ((r[3] & 0xFFFFu) != RES2_NO_STORE && real) lds_st<P2>(dyn_lds, wbC + ll, nc);  // the layer's head: new arc costs (every lane of the layer has read the old ones)
                    if (real) lds_st<REAL>(dyn_lds, db + ll, mm);                                       // every lane of a layer holds the same value
                }
                lds_min(reinterpret_cast<REAL*>(dyn_lds + wb + (r[1] & 0xFFFFu)), f + nc.x);        // sinks / padding: the lane's own dummy entry
                lds_min(reinterpret_cast<REAL*>(dyn_lds + wb + (r[1] >> 16)), f + nc.y);
                wave_sync();
            };
            if constexpr (RL) {
                // records in LDS: one hop ahead is all the distance an LDS read needs; the record behind the pack's last hop is the spare KiB of
                // its record area (never used)
                uint32_t ra = rlane;
                u4v ra_rec = lds_ld<u4v>(dyn_lds, ra), rb_rec;
                if ((ra_rec[2] & 0xFFFFu) != ll_dummy) lds_st<REAL>(dyn_lds, wbF + (ra_rec[2] >> 16), REAL(0));  // every node of hop 0 is a root (flush_costs_from_root)
                wave_sync();
                for (uint32_t h = 0; h < nh; h += 2) {
                    rb_rec = lds_ld<u4v>(dyn_lds, ra + 1024u);
                    hop(ra_rec);
                    if (h + 1 >= nh) break;
                    ra_rec = lds_ld<u4v>(dyn_lds, ra + 2048u);
                    hop(rb_rec);
                    ra += 2048u;
                }
            } else {
                auto ldrec = [&](uint32_t h) -> u4v { return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + h * 64u) * 16u, 0); };
                u4v r0 = ldrec(0), r1 = ldrec(1), r2 = ldrec(2), r3 = ldrec(3), r4 = ldrec(4), r5 = ldrec(5), r6 = ldrec(6), r7 = ldrec(7);
                if (r0[3] != RES2_PAD) lds_st<REAL>(dyn_lds, wbF + (r0[2] >> 16), REAL(0));  // every node of hop 0 is a root (flush_costs_from_root)
                wave_sync();
#define SMALL_HOP(RK, HK)           \
    hop(RK);                        \
    RK = ldrec(h + (HK) + 8);       \
    if (h + (HK) + 1 >= nh) break;
                for (uint32_t h = 0; h < nh; h += 16) {
                    SMALL_HOP(r0, 0) SMALL_HOP(r1, 1) SMALL_HOP(r2, 2) SMALL_HOP(r3, 3) SMALL_HOP(r4, 4) SMALL_HOP(r5, 5) SMALL_HOP(r6, 6) SMALL_HOP(r7, 7)
                    SMALL_HOP(r0, 8) SMALL_HOP(r1, 9) SMALL_HOP(r2, 10) SMALL_HOP(r3, 11) SMALL_HOP(r4, 12) SMALL_HOP(r5, 13) SMALL_HOP(r6, 14) SMALL_HOP(r7, 15)
                }
            }
        }
        __syncthreads();
#ifdef BDDMMA_EXP_SMALL_SKIP
        if (!(BDDMMA_EXP_SMALL_SKIP & 2))
#endif
        exchange(false);
        __syncthreads();
        // ---------------------------------------------------------------- backward solve sweep (k_bwd_res2's hop)
#ifdef BDDMMA_EXP_SMALL_SKIP
        if (has_pack && !(BDDMMA_EXP_SMALL_SKIP & 4)) {
#else
        if (has_pack) {
#endif
            auto hop = [&](const u4v& r) {
                const uint32_t ll = r[2] & 0xFFFFu, fs = r[2] >> 16;
                const REAL f = lds_ld<REAL>(dyn_lds, wbF + fs);  // padding lanes: whatever the dummy entry holds; their results go nowhere
                const REAL tl = lds_ld<REAL>(dyn_lds, wb + (r[0] & 0xFFFFu)), th = lds_ld<REAL>(dyn_lds, wb + (r[0] >> 16));
                const P2 cc = lds_ld<P2>(dyn_lds, wbC + ll);
                const P2 dd = lds_ld<P2>(dyn_lds, db + ll);
                REAL m0 = (f + cc.x) + tl, m1 = (f + cc.y) + th;
                pair_min_aligned(m0, m1, RL ? (int32_t)r[3] < 0 : (r[3] & 0x10000u) != 0);
                const REAL mm = mm_diff1(m0, m1, omega);
                P2 nc;
                nc.x = (cc.x + min0(mm)) + dd.x;
                nc.y = (cc.y + min0_neg(mm)) + dd.y;
                const REAL t = rmin(nc.y + th, nc.x + tl);
                if constexpr (RL) {   // rewritten records: no exec masks (padding lanes: the dummy pair, the spare entry ns + 2)
                    lds_st<P2>(dyn_lds, wbC + ll, nc);
                    lds_st<REAL>(dyn_lds, db + ll, mm);
                    lds_st<REAL>(dyn_lds, wb + fs, t);
                } else {
                    const bool real = r[3] != RES2_PAD;
                    if ((r[3] & 0xFFFFu) != RES2_NO_STORE && real) lds_st<P2>(dyn_lds, wbC + ll, nc);
                    if (real) {
                        lds_st<REAL>(dyn_lds, db + ll, mm);
                        lds_st<REAL>(dyn_lds, wb + fs, t);
                    }
                }
                wave_sync();
            };
            if constexpr (RL) {
                // from the pack's last hop upwards; the record read "before" hop 0 lies in front of the record area (any LDS contents, never used)
                uint32_t ra = rlane + nhm1 * 1024u;
                u4v ra_rec = lds_ld<u4v>(dyn_lds, ra), rb_rec;
                for (uint32_t k = 0; k < nh; k += 2) {
                    rb_rec = lds_ld<u4v>(dyn_lds, ra - 1024u);
                    hop(ra_rec);
                    if (k + 1 >= nh) break;
                    ra_rec = lds_ld<u4v>(dyn_lds, ra - 2048u);
                    hop(rb_rec);
                    ra -= 2048u;
                }
            } else {
                // k-th hop processed = hop nh - 1 - k of the pack; past the first hop: any record (never used)
                auto ldrec = [&](uint32_t k) -> u4v {
                    const uint32_t h = k < nhm1 ? nhm1 - k : 0u;
                    return __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, (rbase + h * 64u) * 16u, 0);
                };
                u4v r0 = ldrec(0), r1 = ldrec(1), r2 = ldrec(2), r3 = ldrec(3), r4 = ldrec(4), r5 = ldrec(5), r6 = ldrec(6), r7 = ldrec(7);
                for (uint32_t h = 0; h < nh; h += 16) {  // h counts the hops processed, from the pack's last hop upwards
                    SMALL_HOP(r0, 0) SMALL_HOP(r1, 1) SMALL_HOP(r2, 2) SMALL_HOP(r3, 3) SMALL_HOP(r4, 4) SMALL_HOP(r5, 5) SMALL_HOP(r6, 6) SMALL_HOP(r7, 7)
                    SMALL_HOP(r0, 8) SMALL_HOP(r1, 9) SMALL_HOP(r2, 10) SMALL_HOP(r3, 11) SMALL_HOP(r4, 12) SMALL_HOP(r5, 13) SMALL_HOP(r6, 14) SMALL_HOP(r7, 15)
                }
            }
#undef SMALL_HOP
            // lower bound contribution of this pack: sum of root costs-from-terminal (bdd_cuda_base.cu:1243-1251); every node of the first hop is a root
            if (run.ctl != nullptr || it + 1 == n_iters) {   // (uniform) the bound: run_solver's tests, or the state the launch leaves
            const u4v rroot = RL ? lds_ld<u4v>(dyn_lds, rl + (uint32_t)lane * 16u) : __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)lane * 16u, rbase * 16u, 0);
            double lb = (RL ? (rroot[2] & 0xFFFFu) != ll_dummy : rroot[3] != RES2_PAD) ? (double)lds_ld<REAL>(dyn_lds, wb + (rroot[2] >> 16)) : 0.0;
            for (int off2 = 32; off2 > 0; off2 >>= 1) lb += __shfl_down(lb, off2);
            if (lane == 0) lbp[p] = lb;
            }
        }
        __syncthreads();
#ifdef BDDMMA_EXP_SMALL_SKIP
        if (!(BDDMMA_EXP_SMALL_SKIP & 8))
#endif
        exchange(true);
        if (run.ctl != nullptr && wave == 0) {
            // the bound of the iteration in the shape of k_lb_reduce / run_ctl_step (<= 64 partial sums: lane i holds pack i's, an in-wave tree,
            // the other fifteen partial results are 0) and run_solver's tests
            double t = (uint32_t)lane < sm.n_packs ? lbp[lane] : 0.0;
            for (int off2 = 32; off2 > 0; off2 >>= 1) t += __shfl_down(t, off2);
            if (lane == 0) {
                // (the bounds are published to the host — a system-scope fence, ~2 us of PCIe round trip — with the launch's last iteration or
                // the one that stops the run; in between they are only written to the ring)
                const uint32_t reason = run_ctl_finish(run, c, t, it + 1 == n_iters);
                if (c.iter == 0) c.lb_first = t;
                c.lb_post = t;
                c.iter += 1;
                flag[0] = reason;
            }
        }
        __syncthreads();
        if (run.ctl != nullptr && flag[0] != 0) { ++it; break; }  // uniform: the iteration that met a criterion is complete
    }
    (void)it;
    // ---- epilogue: the state -> global memory, as the last iteration's four launches would have left it
    if (has_pack) {
        for (uint32_t j = (uint32_t)lane; j < nslots; j += 64) {
            d.T[slot0 + j] = lds_ld<REAL>(dyn_lds, wb + j * S);
            d.F[slot0 + j] = lds_ld<REAL>(dyn_lds, wbF + j * S);
        }
        for (uint32_t j = (uint32_t)lane; j < nlayers; j += 64)
            reinterpret_cast<P2*>(d.lohi)[layer0 + j] = lds_ld<P2>(dyn_lds, wbC + j * (uint32_t)sizeof(P2));
        if (lane == 0) d.lb_partial[pk.lb_base + p] = lbp[p];
    }
    for (uint32_t k = tid; k < sm.n_entries; k += NT) d.mm_binned[sm.var_ent[k]] = mmE[k];   // the deferred differences of the last backward sweep
    {
        const P2* const sD = reinterpret_cast<const P2*>(dyn_lds);
        for (uint32_t q = 0; q < sm.n_quads; ++q) {
            const uint32_t c0 = sm.quad_hdr[4 * (size_t)q], cnt = sm.quad_hdr[4 * (size_t)q + 1];
            for (uint32_t i = tid; i < cnt; i += NT) {
                const uint32_t e = d.cs_entry[c0 + i], sl = d.cs_slot[c0 + i];
                const_cast<P2*>(reinterpret_cast<const P2*>(d.delta_lay))[e] = sD[(size_t)(q * sm.wpb + sl / pk.stage_cap) * small_stage_stride(sm.nl) + sl % pk.stage_cap];
            }
        }
    }
}

}  // namespace bddmma
