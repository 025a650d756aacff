// kernels/elementwise.hpp — cost updates, bounds, dual operations, the staged projection of the L-BFGS direction, rounding, vector helpers.
// Part of kernels.hpp (include that, not this file: the parts build on each other in its order).
#pragma once

namespace bddmma {

// set_vars_costs_func (bdd_cuda_base.cu:457-474).  Quotient and sum are formed in double and rounded
// once to REAL, as the reference CPU solver does (bdd_parallel_mma_base.cpp:640,651,674-677).
// Two steps.  As one kernel every layer gathered its variable's BDD count and both cost entries itself: three random reads per layer
// over 4-8 MB arrays, 935 MB of sector traffic for 5 M layers (126 us).  Now k_cost_quotients reads the caller's
// vectors once, coalesced, and leaves {c_lo / n, c_hi / n} per variable as one 16-byte record; the per-layer pass makes ONE gather.
// Flags per record: bit 0 / 1 = the side is SET to 0 (variable past the end of a shorter vector, :465-469).
struct CostQuot {
    double lo, hi;
};
template <typename TIN>
__global__ void k_cost_quotients(CostQuot* __restrict__ q, uint8_t* __restrict__ flags, const int32_t* __restrict__ nbdds, const TIN* __restrict__ c_lo, uint64_t n_lo,
                                 const TIN* __restrict__ c_hi, uint64_t n_hi, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    const double nb = (double)nbdds[v];
    CostQuot r;
    r.lo = (n_lo && v < n_lo) ? (double)c_lo[v] / nb : 0.0;
    r.hi = (n_hi && v < n_hi) ? (double)c_hi[v] / nb : 0.0;
    q[v] = r;
    flags[v] = (uint8_t)(((n_lo && v >= n_lo) ? 1 : 0) | ((n_hi && v >= n_hi) ? 2 : 0));
}
template <typename REAL>
__global__ void k_update_costs(REAL* __restrict__ lohi, const int32_t* __restrict__ var, const CostQuot* __restrict__ q, const uint8_t* __restrict__ flags,
                               uint32_t do_lo, uint32_t do_hi, uint32_t n_layers)
{
    using P2 = typename Pair<REAL>::type;
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_layers) return;
    const int v = var[l];
    const CostQuot d = q[v];
    const uint32_t f = flags[v];
    P2 c = reinterpret_cast<P2*>(lohi)[l];
    if (do_lo) c.x = (f & 1u) ? REAL(0) : REAL((double)c.x + d.lo);
    if (do_hi) c.y = (f & 2u) ? REAL(0) : REAL((double)c.y + d.hi);
    reinterpret_cast<P2*>(lohi)[l] = c;
}

template <typename REAL>
__global__ void k_set_cost(REAL* __restrict__ hi, const uint32_t* __restrict__ var_layers, uint32_t k0, uint32_t k1, REAL c)
{
    const uint32_t k = k0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k < k1) hi[2 * (size_t)var_layers[k]] += c;
}

// Deterministic fixed-shape reduction of the per-pack partial lower bounds.
// `seq_out` (pinned host memory, may be null): receives `seq` after the bound has been written — the host polls it instead of waiting for
// the stream (an interrupt-driven wait was measured to leave the GPU idle for 26 us per bound read in the L-BFGS loop, tools/gaps.sh).
static __global__ void k_lb_reduce(const double* __restrict__ part, uint32_t n, double* __restrict__ out, uint64_t* seq_out = nullptr, uint64_t seq = 0)
{
    __shared__ double red[16];
    double acc = 0.0;
    // a thread's partial sums eight at a time: their loads are in flight together, the additions keep the order of the plain loop (as a
    // plain loop every element was a dependent round trip: 4.7 us per launch for the 7 813 packs of the 10.5 M-node instance)
    uint32_t i = threadIdx.x;
    for (; i + 7 * blockDim.x < n; i += 8 * blockDim.x) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n; i += blockDim.x) acc += part[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += red[i];
        *out = t;
        if (seq_out != nullptr) {
            __threadfence_system();
            *reinterpret_cast<volatile uint64_t*>(seq_out) = seq;
        }
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
__global__ void k_net_costs(const REAL* __restrict__ lo, const REAL* __restrict__ hi, const REAL* __restrict__ mm_binned,
                            const uint32_t* __restrict__ lpos, REAL* __restrict__ out, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) out[l] = hi[2 * (size_t)l] - lo[2 * (size_t)l] + mm_binned[lpos[l]];
}

// distribute_deffered_mm_diff_func (bdd_cuda_base.cu:1396-1414) + the zero-fill of :1427
template <typename REAL>
__global__ void k_distribute_delta(REAL* __restrict__ lo, REAL* __restrict__ hi, REAL* __restrict__ mm_binned,
                                   const uint32_t* __restrict__ lpos, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const uint32_t e = lpos[l];
    const REAL m = mm_binned[e];
    if (m > 0) hi[2 * (size_t)l] += m;
    else lo[2 * (size_t)l] -= m;
    mm_binned[e] = REAL(0);
}

// add_scaled_product_func (bdd_cuda_parallel_mma.h:54-60)
template <typename REAL>
__global__ void k_gradient_step(REAL* __restrict__ hi, const REAL* __restrict__ g, REAL step, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n) hi[2 * (size_t)l] = hi[2 * (size_t)l] + step * g[l];
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
    // the layers of a variable are scattered over the whole vector: issue the first MAXR gathers together
    // instead of one dependent round trip per layer (same summation order as the plain loop)
    constexpr int MAXR = 8;
    uint32_t idx[MAXR];
    REAL val[MAXR];
#pragma unroll
    for (int u = 0; u < MAXR; ++u) idx[u] = k0 + u < k1 ? var_layers[k0 + u] : 0u;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) val[u] = k0 + u < k1 ? g[idx[u]] : REAL(0);
    REAL s = 0;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) s += val[u];
    for (uint32_t k = k0 + MAXR; k < k1; ++k) s += g[var_layers[k]];
    const REAL q = s / REAL(k1 - k0);
#pragma unroll
    for (int u = 0; u < MAXR; ++u)
        if (k0 + u < k1) g[idx[u]] = val[u] - q;
    for (uint32_t k = k0 + MAXR; k < k1; ++k) g[var_layers[k]] -= q;
}

// The two halves of make_dual_feasible for a vector that is only ever applied as a cost update (the L-BFGS direction): the
// per-variable means are gathered once (no scattered write-back of the projected vector) ...
template <typename REAL>
__global__ void k_projection_means(const REAL* __restrict__ g, const uint32_t* __restrict__ var_ptr, const uint32_t* __restrict__ var_layers,
                                   REAL* __restrict__ q, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    const uint32_t k0 = var_ptr[v], k1 = var_ptr[v + 1];
    if (k1 == k0) { q[v] = REAL(0); return; }
    constexpr int MAXR = 8;
    uint32_t idx[MAXR];
    REAL val[MAXR];
#pragma unroll
    for (int u = 0; u < MAXR; ++u) idx[u] = k0 + u < k1 ? var_layers[k0 + u] : 0u;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) val[u] = k0 + u < k1 ? g[idx[u]] : REAL(0);
    REAL s = 0;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) s += val[u];
    for (uint32_t k = k0 + MAXR; k < k1; ++k) s += g[var_layers[k]];
    q[v] = s / REAL(k1 - k0);   // same summation order and quotient as k_make_dual_feasible
}
// ... and subtracted where the step is applied: hi += step * (g[l] - q[var(l)]), the value k_make_dual_feasible + k_gradient_step produce
template <typename REAL>
__global__ void k_gradient_step_projected(REAL* __restrict__ hi, const REAL* __restrict__ g, const REAL* __restrict__ q,
                                          const uint32_t* __restrict__ layer_var, REAL step, uint32_t n)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const REAL gp = g[l] - q[layer_var[l]];
    hi[2 * (size_t)l] = hi[2 * (size_t)l] + step * gp;
}

// ---- make_dual_feasible of a device vector through the staging tables (the L-BFGS direction at large sizes) ------------------------
// As gathers (k_projection_means + k_gradient_step_projected) the layer <-> variable coupling costs a random access per layer in
// each direction: 63 us for the means and 44 us per trial step at 5 M layers (profiles/r03_lbfgs_time_to_bound.txt), ~30 us of each
// being the gather.  The sweeps do the same coupling through the (entry, slot) staging tables in runs of consecutive entries, and the
// exchange reduces per variable in LDS.  The same three steps for any layer-ordered vector:
//   k_stage_transpose<.., 0> : layers -> entries   (a quad's waves copy their stage group to LDS, the items stream it out by entry)
//   k_project_entries        : per bin: sums per variable in LDS (double accumulators), x_e -= sum / nr_bdds(var), in place
//   k_stage_transpose<.., 1> : entries -> layers   (the reverse), optionally applying the first gradient step on the way
// after which every trial step is the plain streaming k_gradient_step.  Layers of wide / huge packs have no staging tables: they go
// through lpos (k_layers_to_entries / k_entries_to_layers on their range).
template <typename REAL, int WPB, int TO_LAYERS>
__global__ void __launch_bounds__(64 * WPB) k_stage_transpose(const REAL* __restrict__ in, REAL* __restrict__ out, PackDev pk,
                                                               const uint32_t* __restrict__ cs_entry, const uint16_t* __restrict__ cs_slot,
                                                               uint32_t n_narrow_layers, uint32_t n_layers, REAL* __restrict__ lohi, REAL step)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    REAL* sD = reinterpret_cast<REAL*>(dyn_lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t quad = blockIdx.x;
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    const uint32_t g0 = has_pack ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = has_pack ? pk.pack_group_ptr[p + 1] - g0 : 0;
    const uint32_t r0 = pk.quad_round_ptr[quad], n_rounds = pk.quad_round_ptr[quad + 1] - r0;
    const rsrc_t rse = make_rsrc(cs_entry, n_narrow_layers), rss = make_rsrc(cs_slot, n_narrow_layers);
    const rsrc_t rin = make_rsrc(in, n_layers), rout = make_rsrc(out, n_layers), rlh = make_rsrc(lohi, 2ull * n_layers);
    REAL* sDw = sD + (size_t)wave * pk.stage_cap;
    for (uint32_t k = 0; k < n_rounds; ++k) {
        const uint32_t c0 = pk.cs_ptr[r0 + k], cnt = pk.cs_ptr[r0 + k + 1] - c0;
        uint32_t gl0 = 0, gn = 0;
        if (k < ng) {
            gl0 = pk.grp_layer_off[g0 + k];
            gn = pk.grp_layer_off[g0 + k + 1] - gl0;
        }
        uint32_t e[STAGE_ITERS], sl[STAGE_ITERS];
#pragma unroll
        for (int u = 0; u < STAGE_ITERS; ++u) {
            const uint32_t i = 64 * WPB * u + tid;
            e[u] = bload_u32(rse, i < cnt ? (c0 + i) * 4u : OOB);
            sl[u] = bload_u16(rss, i < cnt ? (c0 + i) * 2u : OOB);
        }
        if (!TO_LAYERS) {
            REAL x[STAGE_ITERS];
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = lane + 64 * u;
                bload(x[u], rin, i < gn ? (gl0 + i) * (uint32_t)sizeof(REAL) : OOB);
            }
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = lane + 64 * u;
                if (i < gn) sDw[i] = x[u];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = 64 * WPB * u + tid;
                const REAL v = sD[i < cnt ? sl[u] : 0];
                bstore(v, rout, i < cnt ? e[u] * (uint32_t)sizeof(REAL) : OOB);
            }
            __syncthreads();  // the next round overwrites the staging area
        } else {
            REAL v[STAGE_ITERS];
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = 64 * WPB * u + tid;
                bload(v[u], rin, i < cnt ? e[u] * (uint32_t)sizeof(REAL) : OOB);
            }
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = 64 * WPB * u + tid;
                if (i < cnt) sD[sl[u]] = v[u];
            }
            __syncthreads();
            // the group's layers are contiguous: values out, and hi += step * x on whole {lo, hi} pairs (lo rewritten unchanged: full-width
            // stores instead of every other word), all loads of the group in flight together
            using P2 = typename Pair<REAL>::type;
            P2 c[STAGE_ITERS];
            if (lohi != nullptr) {
#pragma unroll
                for (int u = 0; u < STAGE_ITERS; ++u) {
                    const uint32_t i = lane + 64 * u;
                    bload(c[u], rlh, i < gn ? (gl0 + i) * (uint32_t)sizeof(P2) : OOB);
                }
            }
#pragma unroll
            for (int u = 0; u < STAGE_ITERS; ++u) {
                const uint32_t i = lane + 64 * u;
                const REAL x = sDw[i < gn ? i : 0];
                bstore(x, rout, i < gn ? (gl0 + i) * (uint32_t)sizeof(REAL) : OOB);
                if (lohi != nullptr) {
                    c[u].y = c[u].y + step * x;   // k_gradient_step
                    bstore(c[u], rlh, i < gn ? (gl0 + i) * (uint32_t)sizeof(P2) : OOB);
                }
            }
            __syncthreads();
        }
    }
}

// k_stage_transpose<..., 0> whose input is not an array but a linear combination of stored vectors evaluated on the fly (layout.hpp:
// LinComb — the L-BFGS direction q = g + sum cy y + sum cs s of lbfgs.hip, same operations in the same order as its k_lb_direction):
// the direction goes straight from the history into the staging area and leaves in entry order, instead of being written in layer
// order by one pass and read back by the next (2 x 20 / 40 MB and a launch at 5 M layers).  A lane forms four consecutive layers per trip —
// 16-byte loads of the REAL vectors, 4-byte loads of the char vectors, as the wrapper's own passes — starting at the multiple of 4 at or
// below the group's first layer; what falls outside the group is computed and dropped.
typedef uint32_t lc_u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t lc_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lc_ld4(float (&v)[4], const float* p)
{
    const lc_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const lc_u32x4*>(p));
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};  // (__builtin_bit_cast straight from a vector element was seen to take element 0 for every one)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(float, w[i]);
}
__device__ __forceinline__ void lc_ld4(double (&v)[4], const double* p)
{
    const lc_u64x2 a = __builtin_nontemporal_load(reinterpret_cast<const lc_u64x2*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const lc_u64x2*>(p) + 1);
    const uint64_t w[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(double, w[i]);
}
template <typename REAL, int WPB, int NS>  // NS = 0: lc.ns vectors, known at run time only
__global__ void __launch_bounds__(64 * WPB) k_stage_lincomb(LinComb lc, REAL* __restrict__ out, PackDev pk, const uint32_t* __restrict__ cs_entry,
                                                             const uint16_t* __restrict__ cs_slot, uint32_t n_narrow_layers, uint32_t n_layers)
{
    constexpr int NK = NS > 0 ? NS : LINCOMB_MAX;
    const int ns = NS > 0 ? NS : lc.ns;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    REAL* sD = reinterpret_cast<REAL*>(dyn_lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t quad = blockIdx.x;
    const uint32_t p = quad * WPB + wave;
    const bool has_pack = p < pk.n_packs;
    const uint32_t g0 = has_pack ? pk.pack_group_ptr[p] : 0;
    const uint32_t ng = has_pack ? pk.pack_group_ptr[p + 1] - g0 : 0;
    const uint32_t r0 = pk.quad_round_ptr[quad], n_rounds = pk.quad_round_ptr[quad + 1] - r0;
    const rsrc_t rse = make_rsrc(cs_entry, n_narrow_layers), rss = make_rsrc(cs_slot, n_narrow_layers);
    const rsrc_t rout = make_rsrc(out, n_layers);
    REAL* sDw = sD + (size_t)wave * pk.stage_cap;
    const REAL* sk[NK];
    const char* yk[NK];
    double cy[NK], cs[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const uint32_t ps = k < ns ? lc.order[k] : 0u;
        sk[k] = reinterpret_cast<const REAL*>(lc.S) + (size_t)ps * lc.slot;
        yk[k] = lc.Y + (size_t)ps * lc.slot;
        cy[k] = k < ns ? lc.cy[ps] : 0.0;
        cs[k] = k < ns ? lc.cs[ps] : 0.0;
    }
    for (uint32_t k = 0; k < n_rounds; ++k) {
        const uint32_t c0 = pk.cs_ptr[r0 + k], cnt = pk.cs_ptr[r0 + k + 1] - c0;
        uint32_t gl0 = 0, gn = 0;
        if (k < ng) {
            gl0 = pk.grp_layer_off[g0 + k];
            gn = pk.grp_layer_off[g0 + k + 1] - gl0;
        }
        uint32_t e[STAGE_ITERS], sl[STAGE_ITERS];
#pragma unroll
        for (int u = 0; u < STAGE_ITERS; ++u) {
            const uint32_t i = 64 * WPB * u + tid;
            e[u] = bload_u32(rse, i < cnt ? (c0 + i) * 4u : OOB);
            sl[u] = bload_u16(rss, i < cnt ? (c0 + i) * 2u : OOB);
        }
        const uint32_t a0 = gl0 & ~3u;
        const uint32_t nch = gn ? (gl0 + gn - a0 + 3u) / 4u : 0u;
        for (uint32_t c = lane; c < nch; c += 64) {
            const size_t j = (size_t)a0 + 4 * (size_t)c;
            const uint32_t g4 = *reinterpret_cast<const uint32_t*>(lc.g + j);
            uint32_t y4[NK];
            REAL s4[NK][4];
#pragma unroll
            for (int h = 0; h < NK; ++h)
                if (h < ns) {
                    y4[h] = *reinterpret_cast<const uint32_t*>(yk[h] + j);
                    lc_ld4(s4[h], sk[h] + j);
                }
#pragma unroll
            for (int el = 0; el < 4; ++el) {
                double q = (double)(char)(signed char)((g4 >> (8 * el)) & 0xFFu);
#pragma unroll
                for (int h = 0; h < NK; ++h)
                    if (h < ns) q += cy[h] * (double)(char)(signed char)((y4[h] >> (8 * el)) & 0xFFu);
#pragma unroll
                for (int h = 0; h < NK; ++h)
                    if (h < ns) q += cs[h] * (double)s4[h][el];
                const uint32_t li = (uint32_t)(j + el) - gl0;  // (wraps below the group's first layer)
                if (li < gn) sDw[li] = REAL(q);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < STAGE_ITERS; ++u) {
            const uint32_t i = 64 * WPB * u + tid;
            const REAL v = sD[i < cnt ? sl[u] : 0];
            bstore(v, rout, i < cnt ? e[u] * (uint32_t)sizeof(REAL) : OOB);
        }
        __syncthreads();  // the next round overwrites the staging area
    }
}

// One workgroup per bin of variables (the exchange's bins and its u16 local variable indices): x_e -= (sum over the entries of the
// variable) / nr_bdds(variable).  The sum is accumulated in double by LDS atomics and rounded to REAL once, then divided in REAL as
// k_make_dual_feasible does.
template <typename REAL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_project_entries(REAL* __restrict__ x, const uint32_t* __restrict__ bin_ptr, const uint16_t* __restrict__ bvar,
                                                               const int32_t* __restrict__ nbdds, uint32_t vars_per_bin, uint32_t n_vars, uint32_t n_entries)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    double* tile = reinterpret_cast<double*>(dyn_lds);
    constexpr int U = 8;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t v0 = b * vars_per_bin;
    const uint32_t nv = min(vars_per_bin, n_vars - v0);
    const uint32_t e0 = bin_ptr[b], e1 = bin_ptr[b + 1];
    const rsrc_t rx = make_rsrc(x, n_entries), rv = make_rsrc(bvar, n_entries);
    for (uint32_t i = tid; i < nv; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    // Bins of up to KEEP chunks (3 x 8 entries per thread: every bin of the automatic layout) keep their entries in registers between the
    // accumulation and the subtraction — one read of the vector instead of two (round 6: 84 -> 53 MB per launch at 5 M entries).  Same
    // additions into the same accumulators (LDS atomics on doubles, as before), same subtraction: the same results.
    constexpr int KEEP = 3;
    if (e1 - e0 <= (uint32_t)(KEEP * THREADS * U)) {   // uniform
        REAL m[KEEP][U];
        uint32_t lv[KEEP][U];
#pragma unroll
        for (int c = 0; c < KEEP; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = e0 + (uint32_t)c * THREADS * U + tid + u * THREADS;
                bload(m[c][u], rx, e < e1 ? e * (uint32_t)sizeof(REAL) : OOB);
                lv[c][u] = bload_u16(rv, e < e1 ? e * 2u : OOB);
            }
        }
#pragma unroll
        for (int c = 0; c < KEEP; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = e0 + (uint32_t)c * THREADS * U + tid + u * THREADS;
                if (e < e1) lds_add(&tile[lv[c][u]], (double)m[c][u]);
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < nv; i += THREADS) {
            const int nb = nbdds[v0 + i];
            const REAL s = REAL(tile[i]);
            tile[i] = nb > 0 ? (double)(s / REAL(nb)) : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KEEP; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = e0 + (uint32_t)c * THREADS * U + tid + u * THREADS;
                bstore(m[c][u] - REAL(tile[lv[c][u]]), rx, e < e1 ? e * (uint32_t)sizeof(REAL) : OOB);
            }
        }
        return;
    }
    for (uint32_t base = e0; base < e1; base += THREADS * U) {
        REAL m[U];
        uint32_t lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = base + tid + u * THREADS;
            bload(m[u], rx, e < e1 ? e * (uint32_t)sizeof(REAL) : OOB);
            lv[u] = bload_u16(rv, e < e1 ? e * 2u : OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = base + tid + u * THREADS;
            if (e < e1) lds_add(&tile[lv[u]], (double)m[u]);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nv; i += THREADS) {
        const int nb = nbdds[v0 + i];
        const REAL s = REAL(tile[i]);
        tile[i] = nb > 0 ? (double)(s / REAL(nb)) : 0.0;
    }
    __syncthreads();
    for (uint32_t base = e0; base < e1; base += THREADS * U) {
        REAL m[U];
        uint32_t lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = base + tid + u * THREADS;
            bload(m[u], rx, e < e1 ? e * (uint32_t)sizeof(REAL) : OOB);
            lv[u] = bload_u16(rv, e < e1 ? e * 2u : OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = base + tid + u * THREADS;
            bstore(m[u] - REAL(tile[lv[u]]), rx, e < e1 ? e * (uint32_t)sizeof(REAL) : OOB);
        }
    }
}

// compute_primal_objective_vec (bdd_cuda_base.cu:1352-1362)
template <typename REAL>
__global__ void k_primal_objective(const REAL* __restrict__ lo, const REAL* __restrict__ hi, const uint32_t* __restrict__ var_ptr,
                                   const uint32_t* __restrict__ var_layers, REAL* __restrict__ out, uint32_t n_vars)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    REAL s = 0;
    for (uint32_t k = var_ptr[v]; k < var_ptr[v + 1]; ++k) s += hi[2 * (size_t)var_layers[k]] - lo[2 * (size_t)var_layers[k]];
    out[v] = s;
}

// ---- primal rounding by cost perturbation (incremental_mm_agreement_rounding_cuda.cu) -----------------
// One thread per variable over its (variable,bdd)-sorted layers: sign agreement of the min-marginal
// differences (mm_diff_direction_func :29-41, fill_mm_type_func :43-65), their sums (compute_mm_sums
// :110-134) and the cost perturbation of mm_types_transform (:136-205, only_perturb_inconsistent = false).
// counts[0..3] = #one, #zero, #equal, #inconsistent.  The reference draws its random numbers from
// thrust::default_random_engine discarded by thread id (:177-181); here a counter-based hash of
// (variable, round, seed) — statistically equivalent, not bit-identical.
__device__ __forceinline__ float hash_uniform(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

template <typename REAL>
__global__ void k_round_perturb(const REAL* __restrict__ mm0, const REAL* __restrict__ mm1, const uint32_t* __restrict__ var_ptr,
                                const uint32_t* __restrict__ var_layers, REAL* __restrict__ cost_delta_0, REAL* __restrict__ cost_delta_1,
                                char* __restrict__ sol, uint32_t* __restrict__ counts, uint32_t n_vars, double delta, uint32_t round,
                                uint32_t seed)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vars) return;
    int dmin = 2, dmax = -2;
    REAL s0 = 0, s1 = 0;
    for (uint32_t k = var_ptr[v]; k < var_ptr[v + 1]; ++k) {
        const REAL a = mm0[var_layers[k]], b = mm1[var_layers[k]];
        // mm_diff_direction_func (:29-41): `mm_0 + 1e-6 <= mm_1` with a double literal, i.e. compared in double
        const int dir = ((double)a + 1e-6 <= (double)b) ? -1 : (((double)b + 1e-6 <= (double)a) ? 1 : 0);
        dmin = min(dmin, dir);
        dmax = max(dmax, dir);
        s0 += a;
        s1 += b;
    }
    int type;  // 0 one, 1 zero, 2 equal, 3 inconsistent
    if (dmin == 2) type = 1;            // variable in no BDD: any value is consistent, take 0
    else if (dmin > 0) type = 0;
    else if (dmax < 0) type = 1;
    else if (dmin == 0 && dmax == 0) type = 2;
    else type = 3;
    atomicAdd(&counts[type], 1u);
    sol[v] = type == 0 ? 1 : 0;
    REAL c0 = 0, c1 = 0;
    if (type == 0) c0 = REAL(delta);
    else if (type == 1) c1 = REAL(delta);
    else {
        const float r = (2.0f * hash_uniform(v, round, seed) - 1.0f) * (float)delta;  // U(-delta, delta)
        const REAL mag = REAL(fabsf(r) * delta);
        if (type == 2) {
            if (r < 0.0f) c0 = mag; else c1 = mag;
        } else {
            if (s0 < s1) c1 = mag; else c0 = mag;
        }
    }
    cost_delta_0[v] = c0;
    cost_delta_1[v] = c1;
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

// dst[i * ds] = src[i * ss]   (interleaved {lo,hi} array <-> the API's separate cost vectors)
template <typename T>
__global__ void k_strided_copy(T* __restrict__ dst, uint32_t ds, const T* __restrict__ src, uint32_t ss, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(size_t)i * ds] = src[(size_t)i * ss];
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, T v, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// STREAM triad a = b + s*c (COPY: a = b): the measured HBM ceilings bench.py quotes next to the sweeps.  One 16-byte vector per
// thread, no loop: a one-shot grid keeps every CU's memory queue full for the whole launch and reaches the 6.3-6.7 TB/s of
// MI355X_MICROARCH.md, where the persistent grid-stride version of rounds 1-2 (4 vectors per thread and trip, non-temporal) stayed at
// 5.1-5.3 TB/s — below what the solver's own sweeps sustain, i.e. not a ceiling (VERDICT r2 #4b).  profiles/r01_stream_ceiling.txt has
// the variants.
typedef float stream_v4 __attribute__((ext_vector_type(4)));
template <bool COPY>
static __global__ void __launch_bounds__(256) k_stream(stream_v4* __restrict__ a, const stream_v4* __restrict__ b,
                                                       const stream_v4* __restrict__ c, float s, uint64_t n4)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    stream_v4 x = b[i];
    if (!COPY) x += s * c[i];
    a[i] = x;
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
