// lbfgs.hip — L-BFGS outer loop around the HIP parallel-MMA solver.
//
// Mirrors LPMP::lbfgs<bdd_cuda_parallel_mma<REAL>, device_vector<REAL>, REAL, device_vector<char>, true>
// (reference: include/bdd_solver/lbfgs.h:35-111, src/bdd_solver/lbfgs_impl.h:45-419).
// The reference's device branches are compiled out (`#ifdef CUDACC`, lbfgs_impl.h:59...303) and its
// alpha history is pushed uninitialised (:251-263), so bit-parity with it is undefined; this file
// implements the maths those lines intend: the standard two-loop recursion with
//   s_k = x_k - x_{k-1},  y_k = g_{k-1} - g_k,  rho_inv_k = <s_k, y_k>  (kept only if > 1e-8),
//   initial H diagonal rho_inv_last / (1e-8 + |y_last|^2) folded into the first beta,
// the step-size search of :159-224 and the mma/lbfgs switch of :409-417.
// Vectors stay on the device; dot products are two-stage deterministic reductions.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bdd_mma.h"
#include "kernels.hpp"
#include "solver.hpp"

using namespace bddmma;

struct bddmma_lbfgs {
    bddmma_solver* s = nullptr;
    bddmma_lbfgs_params p{};
    std::string err;
    virtual ~bddmma_lbfgs() {}
    virtual int iteration() = 0;
    virtual void flush() = 0;
    virtual void get_state(bddmma_lbfgs_state* out) const = 0;
};

namespace {

#define LHIP(expr)                                                                 \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return BDDMMA_ERR_DEVICE;                                              \
        }                                                                          \
    } while (0)

// Scalars of the two-loop recursion stay on the device (no host round trip per dot product):
//   sc[i] = alpha_i (i < 32), sc[SC_DOT] = last dot product, sc[SC_YNORM] = |y_last|^2, sc[SC_COEF] = coefficient of an unfused axpy
enum : int { SC_DOT = 32, SC_YNORM = 33, SC_COEF = 34, SC_COUNT = 40 };
// What the last block of a dot product does with the result r (the scalar kernels of the recursion, fused):
//   FIN_STORE : sc[slot] = r
//   FIN_ALPHA : alpha_i = r / rho_inv_i ; sc[i] = alpha_i ; sc[SC_COEF] = -alpha_i            (next axpy: d -= alpha_i y_i)
//   FIN_BETA  : rho = 1 / rho_inv_i (times the initial H diagonal rho_inv_last / (1e-8 + |y_last|^2) for the first);
//               sc[SC_COEF] = alpha_i - rho r                                                   (next axpy: d += (alpha_i - beta) s_i)
enum : int { FIN_STORE = 0, FIN_ALPHA = 1, FIN_BETA = 2 };
constexpr int LBFGS_UNROLL = 8;  // elements of a thread's strided range whose loads are issued together (k_axpy_dot)
struct DotFin {
    int op, slot, i, first;
    double rho_inv, rho_inv_last;
};

// Second stage of a dot product (k_dot leaves one partial sum per block): one block adds the partials in a fixed
// order — deterministic — and applies `fin`.  (A single-launch variant with a ticket counter was slower: 1024
// same-address atomics, and __threadfence() writes back the XCD's L2: 23-33 us instead of 12 + 5 us.)
static __global__ void __launch_bounds__(1024) k_reduce_fin(const double* __restrict__ partial, uint32_t n_partial, double* __restrict__ sc, DotFin fin)
{
    __shared__ double red[16];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n_partial; i += blockDim.x) acc += partial[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double r = 0.0;
    for (uint32_t i = 0; i < blockDim.x / 64; ++i) r += red[i];
    if (fin.op == FIN_STORE) {
        sc[fin.slot] = r;
    } else if (fin.op == FIN_ALPHA) {
        const double al = r / fin.rho_inv;
        sc[fin.i] = al;
        sc[SC_COEF] = -al;
    } else {
        double rho = 1.0 / fin.rho_inv;
        if (fin.first) rho *= fin.rho_inv_last / (1e-8 + sc[SC_YNORM]);
        sc[SC_COEF] = sc[fin.i] - rho * r;
    }
}
// Fused passes of the L-BFGS vector algebra.  The two-loop recursion is a chain of 2m (dot product -> coefficient -> axpy) steps over
// vectors of nr_layers entries; as separate launches every step reads the direction twice.  Here the axpy of step k and the dot
// product of step k + 1 are one pass: d += coef * x, then partial sums of <a, d> (same per-block partition and summation order as
// k_dot, so the results are bit-identical to the unfused version).
// The coefficient of the axpy is the finalised previous dot product (DotFin, as k_reduce_fin): every block adds the previous
// kernel's partial sums itself, in the same fixed order, instead of a 1-block launch in between.  Partial sums alternate between
// two buffers (a block may still be reading the previous ones while another already writes its own).
template <typename REAL, typename TX, typename TA, bool DOT>
static __global__ void __launch_bounds__(256) k_axpy_dot(REAL* __restrict__ d, const TX* __restrict__ x, const TA* __restrict__ a, const double* __restrict__ prev_partial,
                                                         uint32_t n_partial, double* __restrict__ sc, DotFin fin, double* __restrict__ partial, uint32_t n)
{
    __shared__ double red[4];
    __shared__ double s_coef;
    double acc = 0.0;
    {   // n_partial <= 1024: the (up to) four partial sums of this thread in one batch of loads, added in the order of the plain loop
        double pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = threadIdx.x + 256u * u;
            pp[u] = i < n_partial ? prev_partial[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (threadIdx.x + 256u * u < n_partial) acc += pp[u];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double r = (red[0] + red[1]) + (red[2] + red[3]);
        if (fin.op == FIN_ALPHA) {
            const double al = r / fin.rho_inv;
            if (blockIdx.x == 0) sc[fin.i] = al;
            s_coef = -al;
        } else {
            double rho = 1.0 / fin.rho_inv;
            if (fin.first) rho *= fin.rho_inv_last / (1e-8 + sc[SC_YNORM]);
            s_coef = sc[fin.i] - rho * r;
        }
    }
    __syncthreads();
    const REAL c = REAL(s_coef);
    acc = 0.0;
    // A thread owns ~20 strided elements (1024 blocks over 5 M entries).  As a plain loop every element was a dependent round trip
    // (load, fma, store: 22-29 us per pass); LBFGS_UNROLL elements per trip have their loads in flight together.  Same elements per
    // thread, same order of the additions: the partial sums are bit-identical.
    const uint32_t stride = gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (uint64_t)(LBFGS_UNROLL - 1) * stride < n; i += (uint64_t)LBFGS_UNROLL * stride) {
        REAL dv[LBFGS_UNROLL];
        TX xv[LBFGS_UNROLL];
        TA av[LBFGS_UNROLL];
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            dv[u] = d[i + (uint64_t)u * stride];
            xv[u] = x[i + (uint64_t)u * stride];
            if (DOT) av[u] = a[i + (uint64_t)u * stride];
        }
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            const REAL v = dv[u] + c * REAL(xv[u]);
            d[i + (uint64_t)u * stride] = v;
            if (DOT) acc += (double)av[u] * (double)v;
        }
    }
    for (; i < n; i += stride) {
        const REAL v = d[i] + c * REAL(x[i]);
        d[i] = v;
        if (DOT) acc += (double)a[i] * (double)v;
    }
    if (!DOT) return;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// d = g (the subgradient, char-valued) and partial sums of <a, d>: the start of the recursion
template <typename REAL, typename TA>
static __global__ void k_init_dot(REAL* __restrict__ d, const char* __restrict__ g, const TA* __restrict__ a, double* __restrict__ partial, uint32_t n)
{
    __shared__ double red[4];
    double acc = 0.0;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (uint64_t)(LBFGS_UNROLL - 1) * stride < n; i += (uint64_t)LBFGS_UNROLL * stride) {  // see k_axpy_dot
        char gv[LBFGS_UNROLL];
        TA av[LBFGS_UNROLL];
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            gv[u] = g[i + (uint64_t)u * stride];
            av[u] = a[i + (uint64_t)u * stride];
        }
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            const REAL v = REAL(0) + REAL(1) * REAL(gv[u]);
            d[i + (uint64_t)u * stride] = v;
            acc += (double)av[u] * (double)v;
        }
    }
    for (; i < n; i += stride) {
        const REAL v = REAL(0) + REAL(1) * REAL(g[i]);  // as fill(0) followed by axpy(1, g)
        d[i] = v;
        acc += (double)a[i] * (double)v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}
// store_iterate in one pass (lbfgs_impl.h:78-134): s = x - x_prev, y = g_prev - g, partial sums of <s, y>, and the new "previous" state
template <typename REAL>
static __global__ void k_store_iterate(const REAL* __restrict__ cur_x, REAL* __restrict__ prev_x, const char* __restrict__ cur_g, char* __restrict__ prev_g,
                                       REAL* __restrict__ s_out, char* __restrict__ y_out, double* __restrict__ partial, uint32_t n)
{
    __shared__ double red[4];
    double acc = 0.0;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (uint64_t)(LBFGS_UNROLL - 1) * stride < n; i += (uint64_t)LBFGS_UNROLL * stride) {  // see k_axpy_dot
        REAL xc[LBFGS_UNROLL], xp[LBFGS_UNROLL];
        char gc[LBFGS_UNROLL], gp[LBFGS_UNROLL];
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            const uint64_t j = i + (uint64_t)u * stride;
            xc[u] = cur_x[j];
            gc[u] = cur_g[j];
            xp[u] = prev_x[j];
            gp[u] = prev_g[j];
        }
#pragma unroll
        for (int u = 0; u < LBFGS_UNROLL; ++u) {
            const uint64_t j = i + (uint64_t)u * stride;
            const REAL sv = REAL(xc[u] - xp[u]);
            const char yv = (char)(gp[u] - gc[u]);
            s_out[j] = sv;
            y_out[j] = yv;
            prev_x[j] = xc[u];
            prev_g[j] = gc[u];
            acc += (double)sv * (double)yv;
        }
    }
    for (; i < n; i += stride) {
        const REAL x = cur_x[i];
        const char g = cur_g[i];
        const REAL sv = REAL(x - prev_x[i]);
        const char yv = (char)(prev_g[i] - g);
        s_out[i] = sv;
        y_out[i] = yv;
        prev_x[i] = x;
        prev_g[i] = g;
        acc += (double)sv * (double)yv;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}


// =================================================================================================================================
// Gram-matrix ("vector-free") form of compute_update_direction (round 3; Chen et al., "Large-scale L-BFGS using MapReduce", 2014)
// =================================================================================================================================
// The direction of lbfgs_impl.h:226-316 is always  q = g + sum_p cy[p] y_p + sum_p cs[p] s_p , so the two loops can run on the
// COEFFICIENTS once the dot products among {s_p, y_p, g} are known:  <s_i, q> = Sg[i] + sum cy[r] SY[i][r] + sum cs[r] SS[i][r]  etc.
// The two-loop recursion as vector passes is 2 m dependent (dot -> coefficient -> axpy) launches over nr_layers-sized vectors (13
// launches, ~325 us of a ~580 us iteration at 10.5 M nodes); here an iteration makes TWO passes over the history:
//   k_lb_store_gram : store_iterate (x = hi - lo + deferred mm formed in place — the mm in layer order come from the backward sweep itself,
//                     SolverT::lbfgs_views —, s = x - x_prev, y = g_prev - g, prev <- cur) fused
//                     with every dot product the new pair and the new g add to the Gram matrices (6 per kept pair + 5);
//   k_lb_finalize   : one workgroup: sums the partial dots in a fixed order (deterministic), applies the curvature filter
//                     rho_inv > 1e-8 (:112) and the ring update ON THE DEVICE (no host round trip for rho_inv), runs the two loops
//                     on the coefficients;
//   k_lb_direction  : q = g + sum cy y + sum cs s in one pass (accumulated in double, rounded to REAL once).
// The Gram entries between kept pairs are computed once, when a pair is stored, and stay valid while both are kept.
// Physical slots: LB_P = LB_MAXS + 1 (one spare that receives the new pair); `order` lists the kept slots, oldest first.
constexpr int LB_MAXS = 8;           // history sizes of the fused path (larger ones take the two-loop path below)
constexpr int LB_P = LB_MAXS + 1;
constexpr int LB_BLOCKS = 768;       // workgroups of the store pass = partial sums per dot product (3 per CU: all resident at once, one epilogue each)
struct LbDev {                       // device memory
    double SS[LB_P][LB_P], SY[LB_P][LB_P], YY[LB_P][LB_P];  // SY[a][b] = <s_a, y_b>; SY[a][a] = rho_inv of pair a
    double Sg[LB_P], Yg[LB_P];                                // products with the current subgradient
    double cy[LB_P], cs[LB_P];                                // coefficients of the direction, by physical slot
    double rho_inv_new;
    uint32_t order[LB_P];
    uint32_t count, free_slot, prev_stored, accepted, have_dir, pad_;
};
struct LbHost {                      // pinned host memory: what the host reads (only while the history fills up, and for diagnostics)
    double rho_inv_new;
    uint32_t count, prev_stored, accepted, have_dir;
    uint64_t seq;
};
__host__ __device__ constexpr int lb_ndots(int ns) { return 6 * ns + 5; }

// Four consecutive elements per thread and trip: 16-byte loads of the REAL vectors, 4-byte loads of the char vectors (one byte per
// lane and instruction made the char streams — y of every kept pair, g, g_prev — cost as many load instructions as the REAL ones for a
// quarter of the bytes).  `slot` = elements between two slots of S / Y (a multiple of 64, so every slot is 16-byte aligned).
typedef uint32_t lb_u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t lb_u64x2 __attribute__((ext_vector_type(2)));
// The vectors are streams (each element is read once or twice per iteration, ~280 MB per store pass at 10.5 M layers): non-temporal
// accesses, so that they do not evict the sweeps' arrays from L2 / Infinity Cache — 10.5 M nodes float 2 160 -> 2 250 it/s, double and
// instances that fit the caches unchanged (A/B on one box).
template <typename T>
__device__ __forceinline__ void ld4(T (&v)[4], const T* p)
{
    if (sizeof(T) == 4) {
        const lb_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const lb_u32x4*>(p));
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(T, (typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type)w[i]);
    } else {
        const lb_u64x2 a = __builtin_nontemporal_load(reinterpret_cast<const lb_u64x2*>(p)), b2 = __builtin_nontemporal_load(reinterpret_cast<const lb_u64x2*>(p) + 1);
        const uint64_t w[4] = {a.x, a.y, b2.x, b2.y};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(T, (typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type)w[i]);
    }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const T (&v)[4])
{
    if (sizeof(T) == 4) {
        lb_u32x4 x;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(uint32_t, (typename std::conditional<sizeof(T) == 4, T, float>::type)v[i]);
        __builtin_nontemporal_store(x, reinterpret_cast<lb_u32x4*>(p));
    } else {
        lb_u64x2 a, b2;
        a.x = __builtin_bit_cast(uint64_t, (typename std::conditional<sizeof(T) == 8, T, double>::type)v[0]);
        a.y = __builtin_bit_cast(uint64_t, (typename std::conditional<sizeof(T) == 8, T, double>::type)v[1]);
        b2.x = __builtin_bit_cast(uint64_t, (typename std::conditional<sizeof(T) == 8, T, double>::type)v[2]);
        b2.y = __builtin_bit_cast(uint64_t, (typename std::conditional<sizeof(T) == 8, T, double>::type)v[3]);
        __builtin_nontemporal_store(a, reinterpret_cast<lb_u64x2*>(p));
        __builtin_nontemporal_store(b2, reinterpret_cast<lb_u64x2*>(p) + 1);
    }
}
__device__ __forceinline__ void ldc4(char (&v)[4], const char* p)
{
    const uint32_t x = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (char)(signed char)((x >> (8 * i)) & 0xFFu);
}
__device__ __forceinline__ void stc4(char* p, const char (&v)[4])
{
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) x |= (uint32_t)(unsigned char)v[i] << (8 * i);
    *reinterpret_cast<uint32_t*>(p) = x;
}

template <typename REAL, int NS>
static __global__ void __launch_bounds__(256) k_lb_store_gram(const REAL* __restrict__ x_layer,
                                                              REAL* __restrict__ prev_x, char* __restrict__ cur_g, char* __restrict__ prev_g,
                                                              REAL* __restrict__ S, char* __restrict__ Y, size_t slot, const LbDev* __restrict__ st,
                                                              double* __restrict__ partial, uint32_t n)
{
    constexpr int ND = lb_ndots(NS);
    constexpr int NK = NS > 0 ? NS : 1;
    __shared__ double red[ND][4];
    const bool have_prev = st->prev_stored != 0;   // uniform
    const uint32_t w = st->free_slot;
    REAL* const sw = S + (size_t)w * slot;
    char* const yw = Y + (size_t)w * slot;
    const REAL* sk[NK];
    const char* yk[NK];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const uint32_t p = st->order[k];
        sk[k] = S + (size_t)p * slot;
        yk[k] = Y + (size_t)p * slot;
    }
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;
    // one element: x = hi - lo + mm (net_solver_costs: formed by the backward solve sweep, SolverBase::lbfgs_views), s = x - x_prev,
    // y = g_prev - g (lbfgs_impl.h:81,100), and the products
    auto element = [&](const REAL x, const char g, const REAL xp, const char gp, const REAL (&sv_)[NK], const char (&yv_)[NK],
                       REAL& s_out, char& y_out) {
        const REAL sv = REAL(x - xp);
        const char yv = (char)(gp - g);
        s_out = sv; y_out = yv;
        const double ds = have_prev ? (double)sv : 0.0, dy = have_prev ? (double)yv : 0.0, dg = (double)g;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const double a = (double)sv_[k], b = (double)yv_[k];
            acc[6 * k + 0] += ds * a;
            acc[6 * k + 1] += ds * b;
            acc[6 * k + 2] += a * dy;
            acc[6 * k + 3] += dy * b;
            acc[6 * k + 4] += a * dg;
            acc[6 * k + 5] += b * dg;
        }
        acc[6 * NS + 0] += ds * ds;
        acc[6 * NS + 1] += ds * dy;
        acc[6 * NS + 2] += dy * dy;
        acc[6 * NS + 3] += ds * dg;
        acc[6 * NS + 4] += dy * dg;
    };
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += stride) {
        const uint64_t j = 4 * c;
        REAL x4[4], xp[4] = {REAL(0), REAL(0), REAL(0), REAL(0)}, sv_[NK][4];
        char g[4], gp[4] = {0, 0, 0, 0}, yv_[NK][4];
        ld4(x4, x_layer + j);
        ldc4(g, cur_g + j);
        if (have_prev) {
            ld4(xp, prev_x + j);
            ldc4(gp, prev_g + j);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            ld4(sv_[k], sk[k] + j);
            ldc4(yv_[k], yk[k] + j);
        }
        REAL s4[4];
        char y4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            REAL se[NK];
            char ye[NK];
#pragma unroll
            for (int k = 0; k < NS; ++k) { se[k] = sv_[k][e]; ye[k] = yv_[k][e]; }
            element(x4[e], g[e], xp[e], gp[e], se, ye, s4[e], y4[e]);
        }
        if (have_prev) {
            st4(sw + j, s4);
            stc4(yw + j, y4);
        }
        st4(prev_x + j, x4);
        stc4(prev_g + j, g);
        *reinterpret_cast<uint32_t*>(cur_g + j) = 0u;  // the next argmin-path sweep only writes the layers on a path (bdds_solution_async, prezeroed)
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {  // the last n % 4 elements
        const uint64_t i = 4 * n4 + threadIdx.x;
        REAL se[NK];
        char ye[NK];
#pragma unroll
        for (int k = 0; k < NS; ++k) { se[k] = sk[k][i]; ye[k] = yk[k][i]; }
        REAL sv;
        char yv;
        const REAL x = x_layer[i];
        const char g = cur_g[i];
        element(x, g, have_prev ? prev_x[i] : REAL(0), have_prev ? prev_g[i] : (char)0, se, ye, sv, yv);
        if (have_prev) { sw[i] = sv; yw[i] = yv; }
        prev_x[i] = x;
        prev_g[i] = g;
        cur_g[i] = 0;
    }
    // per-block partial sums, layout [dot][block]: in-wave trees for all dots, ONE barrier, then thread d adds the four waves' values
    // (a barrier pair per dot cost ~4 us per workgroup for the 35 dots of a history of five)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        double a = acc[d];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if ((threadIdx.x & 63) == 0) red[d][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x < ND) partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// One workgroup of 1024: fixed-order sums of the partial dots, curvature filter + ring update, two loops on the coefficients.
// m = history size; want_dir: the host will run the L-BFGS branch if the history is full after this store.
// The state is copied to LDS first and written back at the end: thread 0's recursion is ~400 dependent accesses, which took 30 us
// against global memory (one L2 round trip each) and takes ~2 us against LDS.
static __global__ void __launch_bounds__(1024) k_lb_finalize(const double* __restrict__ partial, uint32_t nb, int ns, int m, int want_dir, LbDev* __restrict__ st,
                                                             LbHost* __restrict__ host)
{
    __shared__ double dots[lb_ndots(LB_MAXS)];
    __shared__ LbDev T;
    static_assert(sizeof(LbDev) % 8 == 0, "LbDev is copied as 8-byte words");
    constexpr uint32_t WORDS = sizeof(LbDev) / 8;
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(st);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&T);
        for (uint32_t i = threadIdx.x; i < WORDS; i += blockDim.x) dst[i] = src[i];
    }
    const int nd = lb_ndots(ns);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // nb <= LB_BLOCKS = 12 x 64 partial sums per dot product, up to four dot products per wave (d = wave, wave + 16, ...): ALL of a lane's
        // partial sums in one batch of loads (one memory round trip instead of one per dot product), added in the order of the plain loop
        constexpr int DPW = (lb_ndots(LB_MAXS) + 15) / 16;
        double pp[DPW][LB_BLOCKS / 64];
#pragma unroll
        for (int r = 0; r < DPW; ++r) {
            const int d = (int)wave + 16 * r;
#pragma unroll
            for (int u = 0; u < LB_BLOCKS / 64; ++u) {
                const uint32_t b = lane + 64u * u;
                pp[r][u] = (d < nd && b < nb) ? partial[(size_t)d * nb + b] : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < DPW; ++r) {
            const int d = (int)wave + 16 * r;
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < LB_BLOCKS / 64; ++u)
                if (lane + 64u * u < nb) a += pp[r][u];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if (lane == 0 && d < nd) dots[d] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t w = T.free_slot;
        uint32_t count = T.count;
        for (int k = 0; k < ns; ++k) {
            const uint32_t p = T.order[k];
            T.Sg[p] = dots[6 * k + 4];
            T.Yg[p] = dots[6 * k + 5];
        }
        uint32_t accepted = 0;
        if (T.prev_stored) {
            const double rho_inv = dots[6 * ns + 1];
            T.rho_inv_new = rho_inv;
            if (rho_inv > 1e-8) {  // lbfgs_impl.h:112
                for (int k = 0; k < ns; ++k) {
                    const uint32_t p = T.order[k];
                    T.SS[w][p] = T.SS[p][w] = dots[6 * k + 0];
                    T.SY[w][p] = dots[6 * k + 1];
                    T.SY[p][w] = dots[6 * k + 2];
                    T.YY[w][p] = T.YY[p][w] = dots[6 * k + 3];
                }
                T.SS[w][w] = dots[6 * ns + 0];
                T.SY[w][w] = rho_inv;
                T.YY[w][w] = dots[6 * ns + 2];
                T.Sg[w] = dots[6 * ns + 3];
                T.Yg[w] = dots[6 * ns + 4];
                uint32_t next_free;
                if ((int)count == m) {  // the oldest pair leaves
                    next_free = T.order[0];
                    for (int k = 0; k + 1 < m; ++k) T.order[k] = T.order[k + 1];
                    T.order[m - 1] = w;
                } else {
                    T.order[count++] = w;
                    uint32_t used = 0;
                    for (uint32_t k = 0; k < count; ++k) used |= 1u << T.order[k];
                    next_free = 0;
                    while (used & (1u << next_free)) ++next_free;
                }
                T.free_slot = next_free;
                T.count = count;
                accepted = 1;
            } else {
                T.prev_stored = 0;  // :118-121: the next call only records the state
            }
        } else {
            T.prev_stored = 1;
        }
        T.accepted = accepted;
        T.have_dir = (want_dir && (int)count == m) ? 1u : 0u;
    }
    __syncthreads();
    // compute_update_direction (:226-316) on the coefficients, by wavefront 0: lane r holds the coefficients of the pair with logical
    // index r (physical slot order[r]); every step's dot product is one LDS read per lane and a butterfly sum, so the 2 m dependent
    // steps cost ~0.2 us each instead of ~2 us as a scalar loop over LDS.  The sum runs over the lanes in a fixed order: deterministic.
    if (wave == 0 && T.have_dir) {
        const int r = (int)lane;
        const bool act = r < m;
        const uint32_t pr = act ? T.order[r] : 0u;
        double cy = 0.0, cs = 0.0, alpha_mine = 0.0;
        auto wave_sum = [&](double v) {
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            return v;
        };
        for (int i = m - 1; i >= 0; --i) {
            const uint32_t p = T.order[i];
            const double part = act ? cy * T.SY[p][pr] + cs * T.SS[p][pr] : 0.0;
            const double dot = T.Sg[p] + wave_sum(part);
            const double al = dot / T.SY[p][p];
            if (r == i) { alpha_mine = al; cy -= al; }
        }
        const uint32_t pl = T.order[m - 1];
        const double h_diag = T.SY[pl][pl] / (1e-8 + T.YY[pl][pl]);  // :291
        for (int i = 0; i < m; ++i) {
            const uint32_t p = T.order[i];
            double rho = 1.0 / T.SY[p][p];
            if (i == 0) rho *= h_diag;
            const double part = act ? cy * T.YY[p][pr] + cs * T.SY[pr][p] : 0.0;
            const double dot = T.Yg[p] + wave_sum(part);
            if (r == i) cs += alpha_mine - rho * dot;
        }
        if (lane < LB_P) { T.cy[lane] = 0.0; T.cs[lane] = 0.0; }
        __builtin_amdgcn_wave_barrier();
        if (act) { T.cy[pr] = cy; T.cs[pr] = cs; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && host != nullptr) {  // (null once the history is full: the host no longer reads the count)
        volatile LbHost* h = host;
        h->rho_inv_new = T.rho_inv_new;
        h->count = T.count;
        h->prev_stored = T.prev_stored;
        h->accepted = T.accepted;
        h->have_dir = T.have_dir;
        __threadfence_system();
        h->seq = ++T.pad_;
    }
    __syncthreads();
    {
        uint64_t* dst = reinterpret_cast<uint64_t*>(st);
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&T);
        for (uint32_t i = threadIdx.x; i < WORDS; i += blockDim.x) dst[i] = src[i];
    }
}

// q = g + sum_k cy[order[k]] y_k + sum_k cs[order[k]] s_k  (oldest first; accumulated in double, rounded to REAL once)
template <typename REAL, int NS>
static __global__ void __launch_bounds__(256) k_lb_direction(REAL* __restrict__ dir, const char* __restrict__ cur_g, const REAL* __restrict__ S, const char* __restrict__ Y,
                                                             size_t slot, const LbDev* __restrict__ st, uint32_t n)
{
    const REAL* sk[NS];
    const char* yk[NS];
    double cy[NS], cs[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const uint32_t p = st->order[k];
        sk[k] = S + (size_t)p * slot;
        yk[k] = Y + (size_t)p * slot;
        cy[k] = st->cy[p];
        cs[k] = st->cs[p];
    }
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += stride) {  // four consecutive elements, see k_lb_store_gram
        const uint64_t j = 4 * c;
        char g[4], yv[NS][4];
        REAL sv[NS][4];
        ldc4(g, cur_g + j);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            ldc4(yv[k], yk[k] + j);
            ld4(sv[k], sk[k] + j);
        }
        REAL out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double q = (double)g[e];
#pragma unroll
            for (int k = 0; k < NS; ++k) q += cy[k] * (double)yv[k][e];
#pragma unroll
            for (int k = 0; k < NS; ++k) q += cs[k] * (double)sv[k][e];
            out[e] = REAL(q);
        }
        st4(dir + j, out);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const uint64_t i = 4 * n4 + threadIdx.x;
        double q = (double)cur_g[i];
#pragma unroll
        for (int k = 0; k < NS; ++k) q += cy[k] * (double)yk[k][i];
#pragma unroll
        for (int k = 0; k < NS; ++k) q += cs[k] * (double)sk[k][i];
        dir[i] = REAL(q);
    }
}
static __global__ void k_lb_reset(LbDev* st)
{
    st->count = 0;
    st->free_slot = 0;
    st->prev_stored = 0;
    st->accepted = 0;
    st->have_dir = 0;
}

template <typename REAL>
struct Lbfgs final : bddmma_lbfgs {
    struct Hist {
        REAL* s = nullptr;
        char* y = nullptr;
        double rho_inv = 0;
    };
    std::deque<Hist> history;
    std::vector<Hist> free_slots;  // history_size + 1 preallocated (s, y) pairs: no hipMalloc / hipFree per iteration
    std::vector<void*> allocs;
    REAL *prev_x = nullptr, *cur_x = nullptr, *dir = nullptr;
    char *prev_g = nullptr, *cur_g = nullptr;
    double *d_partial = nullptr, *d_scalar = nullptr;
    std::deque<double> lb_history;
    // The bound after an iteration is only ENQUEUED (SolverBase::lower_bound_enqueue, slot 0); its value is fetched together with the
    // first trial step's bound of the next iteration (slot 1), so the steady L-BFGS iteration has one host wait instead of two.  While
    // it is pending, lb_history's last entry is a placeholder.
    bool lb_pending = false;
    uint64_t lb_pending_epoch = 0;  // SolverBase::cost_epoch when it was enqueued: costs changed behind the wrapper's back make it history only
    double step_size = 0;
    int unsuccessful = 0;
    bool prev_stored = false;
    // diagnostics (bddmma_lbfgs_get_state): what the last iteration() did
    int last_kind = 0, last_trials = 0;
    double last_applied_step = 0;
    uint64_t mma_iterations = 0, lbfgs_iterations = 0;
    uint32_t n = 0;
    hipStream_t st = nullptr;
    // Gram-matrix path (history sizes <= LB_MAXS; BDDMMA_LBFGS_TWO_LOOP=1 keeps the two-loop passes for A/B runs)
    bool gram = false;
    REAL* S = nullptr;       // LB_P slots of n entries
    char* Y = nullptr;
    LbDev* d_lb = nullptr;
    LbHost *h_lb = nullptr, *d_lb_host = nullptr;  // pinned + its device address
    double* d_gpartial = nullptr;
    size_t slot = 0;         // elements between two slots of S / Y (n rounded up to 64: 16-byte aligned slots for the vector loads)
    int h_count = 0;         // pairs kept: read back while the history fills up, constant (= history_size) afterwards
    bool fused_dir = false;  // this iteration's direction is formed inside the solver's projection (projection_means_lincomb)

    int device = 0;  // cached: the wrapped solver may already be gone when the wrapper is destroyed
    ~Lbfgs() override
    {
        (void)hipSetDevice(device);
        for (void* q : allocs) (void)hipFree(q);
        if (h_lb) (void)hipHostFree(h_lb);
    }
    template <typename T>
    int alloc(T** q, size_t cnt)
    {
        LHIP(hipMalloc((void**)q, (cnt ? cnt : 1) * sizeof(T)));
        allocs.push_back(*q);
        return 0;
    }
    int init()
    {
        SolverBase* b = s->impl;
        device = b->device;
        LHIP(hipSetDevice(device));
        n = (uint32_t)b->n_layers;
        // the vector passes and the staged projection (k_stage_transpose, k_project_entries) carry 32-bit byte offsets into per-layer / per-entry arrays
        if (2ull * b->n_layers * (b->precision == BDDMMA_F64 ? 8 : 4) >= 0xFFFF0000ull) {
            err = "the L-BFGS wrapper supports instances whose per-layer arrays stay below 4 GiB";
            return BDDMMA_ERR_UNSUPPORTED;
        }
        st = (hipStream_t)b->stream_handle();
        step_size = p.init_step_size;
        int rc;
        if (p.history_size >= SC_DOT) { err = "history size must be < 32"; return BDDMMA_ERR_INVALID_ARGUMENT; }
#ifdef BDDMMA_EXPERIMENTAL  // make EXPERIMENTAL=1: the two-loop passes (the form histories > LB_MAXS take) for short histories too
        const char* two_loop = std::getenv("BDDMMA_LBFGS_TWO_LOOP");
        gram = p.history_size <= LB_MAXS && !(two_loop && two_loop[0] == '1');
#else
        gram = p.history_size <= LB_MAXS;
#endif
        // (the char vectors are read four at a time up to the next multiple of 4 past the last layer: k_stage_lincomb)
        if ((rc = alloc(&prev_x, n)) || (rc = alloc(&dir, n)) || (rc = alloc(&prev_g, (size_t)n + 64)) || (rc = alloc(&cur_g, (size_t)n + 64))) return rc;
        if (gram) {
            slot = ((size_t)n + 63) & ~(size_t)63;
            if ((rc = alloc(&S, (size_t)(p.history_size + 1) * slot)) || (rc = alloc(&Y, (size_t)(p.history_size + 1) * slot)) ||
                (rc = alloc(&d_gpartial, (size_t)lb_ndots(LB_MAXS) * LB_BLOCKS)) || (rc = alloc(&d_lb, 1)))
                return rc;
            LHIP(hipHostMalloc((void**)&h_lb, sizeof(LbHost), hipHostMallocMapped | hipHostMallocCoherent));
            LHIP(hipHostGetDevicePointer((void**)&d_lb_host, h_lb, 0));
            std::memset((void*)h_lb, 0, sizeof(LbHost));
            LHIP(hipMemsetAsync(d_lb, 0, sizeof(LbDev), st));
            LHIP(hipMemsetAsync(cur_g, 0, n, st));
            return 0;
        }
        if ((rc = alloc(&cur_x, n)) || (rc = alloc(&d_partial, 2048)) || (rc = alloc(&d_scalar, SC_COUNT))) return rc;
        for (int i = 0; i < p.history_size + 1; ++i) {
            Hist h;
            if ((rc = alloc(&h.s, n)) || (rc = alloc(&h.y, n))) return rc;
            free_slots.push_back(h);
        }
        return 0;
    }
    void get_state(bddmma_lbfgs_state* out) const override
    {
        out->step_size = step_size;
        out->history_entries = gram ? h_count : (int32_t)history.size();
        out->num_unsuccessful_updates = unsuccessful;
        out->last_kind = last_kind;
        out->last_trials = last_trials;
        out->last_applied_step = last_applied_step;
        out->mma_iterations = mma_iterations;
        out->lbfgs_iterations = lbfgs_iterations;
    }
    void flush() override  // flush_lbfgs_states, lbfgs_impl.h:318-326
    {
        unsuccessful = 0;
        if (gram) {
            (void)hipSetDevice(device);
            hipLaunchKernelGGL(k_lb_reset, dim3(1), dim3(1), 0, st, d_lb);
            h_count = 0;
            return;
        }
        for (auto& h : history) free_slots.push_back(h);
        history.clear();
        prev_stored = false;
    }
    // store_iterate (lbfgs_impl.h:45-135) + the Gram updates, on the device; the host reads the number of kept pairs back only while
    // the history fills up (afterwards it is history_size whatever the curvature filter decides: a rejected pair replaces nothing)
    int store_iterate_gram(int want_dir)
    {
        SolverBase* b = s->impl;
        SolverBase::LbfgsViews v;
        int rcv = b->lbfgs_views(&v);
        if (rcv) { err = b->err; return rcv; }
        const uint32_t nb = std::min<uint32_t>(LB_BLOCKS, std::max<uint32_t>(1, (n / 4 + 255) / 256));
#define LB_STORE(NS_)                                                                                                                      \
    hipLaunchKernelGGL((k_lb_store_gram<REAL, NS_>), dim3(nb), dim3(256), 0, st, (const REAL*)v.x_layer, prev_x, cur_g, prev_g, S, Y, slot,     \
                       (const LbDev*)d_lb, d_gpartial, n)
        switch (h_count) {
            case 0: LB_STORE(0); break;
            case 1: LB_STORE(1); break;
            case 2: LB_STORE(2); break;
            case 3: LB_STORE(3); break;
            case 4: LB_STORE(4); break;
            case 5: LB_STORE(5); break;
            case 6: LB_STORE(6); break;
            case 7: LB_STORE(7); break;
            default: LB_STORE(8); break;
        }
#undef LB_STORE
        hipLaunchKernelGGL(k_lb_finalize, dim3(1), dim3(1024), 0, st, (const double*)d_gpartial, nb, h_count, p.history_size, want_dir, d_lb,
                           h_count < p.history_size ? d_lb_host : (LbHost*)nullptr);
        LHIP(hipGetLastError());
        if (h_count < p.history_size) {
            LHIP(hipStreamSynchronize(st));
            h_count = (int)((volatile LbHost*)h_lb)->count;
        }
        return 0;
    }
    int direction_gram()
    {
        const dim3 g(std::min<uint32_t>(4096, std::max<uint32_t>(1, (n / 4 + 255) / 256))), b(256);
#define LB_DIR(NS_) hipLaunchKernelGGL((k_lb_direction<REAL, NS_>), g, b, 0, st, dir, (const char*)prev_g, (const REAL*)S, (const char*)Y, slot, (const LbDev*)d_lb, n)
        switch (p.history_size) {
            case 2: LB_DIR(2); break;
            case 3: LB_DIR(3); break;
            case 4: LB_DIR(4); break;
            case 5: LB_DIR(5); break;
            case 6: LB_DIR(6); break;
            case 7: LB_DIR(7); break;
            default: LB_DIR(8); break;
        }
#undef LB_DIR
        LHIP(hipGetLastError());
        return 0;
    }
    template <typename TA, typename TB>
    int dot(const TA* a, const TB* b, double* out)
    {
        dot_dev(a, b, DotFin{FIN_STORE, SC_DOT, 0, 0, 0.0, 0.0});
        LHIP(hipMemcpyAsync(out, d_scalar + SC_DOT, sizeof(double), hipMemcpyDeviceToHost, st));
        LHIP(hipStreamSynchronize(st));
        return 0;
    }
    // dot product consumed on the device (see DotFin); ordered on the stream, no synchronisation
    uint32_t dot_blocks() const { return std::min<uint32_t>(1024, (n + 255) / 256 ? (n + 255) / 256 : 1); }
    template <typename TA, typename TB>
    void dot_dev(const TA* a, const TB* b, const DotFin& fin)
    {
        const uint32_t blocks = dot_blocks();
        hipLaunchKernelGGL((k_dot<TA, TB>), dim3(blocks), dim3(256), 0, st, a, b, d_partial, n);
        hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(1024), 0, st, d_partial, blocks, d_scalar, fin);
    }
    dim3 grid() const { return dim3((n + 255) / 256 ? (n + 255) / 256 : 1); }

    int resolve_pending()
    {
        if (!lb_pending) return 0;
        double lb;
        int rc = s->impl->lower_bound_fetch(0, &lb);
        if (rc) { err = s->impl->err; return rc; }
        lb_history.back() = lb;
        lb_pending = false;
        return 0;
    }
    int lower_bound(double* lb)
    {
        int rc = s->impl->lower_bound(lb);
        if (rc) err = s->impl->err;
        return rc;
    }

    // store_iterate, lbfgs_impl.h:45-135
    int store_iterate()
    {
        SolverBase* b = s->impl;
        int rc = b->net_solver_costs(cur_x, 1);
        if (rc) { err = b->err; return rc; }
        if (!prev_stored) {
            LHIP(hipMemcpyAsync(prev_x, cur_x, n * sizeof(REAL), hipMemcpyDeviceToDevice, st));
            LHIP(hipMemcpyAsync(prev_g, cur_g, n, hipMemcpyDeviceToDevice, st));
            prev_stored = true;
            return 0;
        }
        Hist h = free_slots.back();
        free_slots.pop_back();
        // x_k - x_{k-1}, g_{k-1} - g_k, <s, y> and prev <- cur in one pass
        const uint32_t blocks = dot_blocks();
        hipLaunchKernelGGL((k_store_iterate<REAL>), dim3(blocks), dim3(256), 0, st, cur_x, prev_x, cur_g, prev_g, h.s, h.y, d_partial, n);
        hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(1024), 0, st, d_partial, blocks, d_scalar, DotFin{FIN_STORE, SC_DOT, 0, 0, 0.0, 0.0});
        LHIP(hipMemcpyAsync(&h.rho_inv, d_scalar + SC_DOT, sizeof(double), hipMemcpyDeviceToHost, st));
        LHIP(hipStreamSynchronize(st));
        if (h.rho_inv > 1e-8) {
            history.push_back(h);
            if ((int)history.size() > p.history_size) {
                free_slots.push_back(history.front());
                history.pop_front();
            }
        } else {
            free_slots.push_back(h);
            prev_stored = false;
        }
        return 0;
    }

    bool update_possible() const { return (int)history.size() >= p.history_size && unsuccessful <= 5; }  // :334-340

    // compute_update_direction, :226-316 — everything queued on the stream, the scalars never leave the device
    int compute_direction()
    {
        const int m = (int)history.size();
        const uint32_t nb = dot_blocks();
        const dim3 g(nb), b(256);
        double* part[2] = {d_partial, d_partial + 1024};
        int cur = 0;
        // |y_last|^2 does not depend on the direction: first
        dot_dev(history.back().y, history.back().y, DotFin{FIN_STORE, SC_YNORM, 0, 0, 0.0, 0.0});
        // direction = grad_f; <s_{m-1}, d>
        hipLaunchKernelGGL((k_init_dot<REAL, REAL>), g, b, 0, st, dir, cur_g, history[m - 1].s, part[cur], n);
        // first loop: d -= alpha_i y_i with alpha_i from the pending dot product, fused with the next one
        // (<s_{i-1}, d>, or <y_0, d> when the second loop starts)
        for (int i = m - 1; i >= 0; --i, cur ^= 1) {
            const DotFin f{FIN_ALPHA, 0, i, 0, history[i].rho_inv, 0.0};
            if (i > 0)
                hipLaunchKernelGGL((k_axpy_dot<REAL, char, REAL, true>), g, b, 0, st, dir, history[i].y, history[i - 1].s, part[cur], nb, d_scalar, f, part[cur ^ 1], n);
            else
                hipLaunchKernelGGL((k_axpy_dot<REAL, char, char, true>), g, b, 0, st, dir, history[0].y, history[0].y, part[cur], nb, d_scalar, f, part[cur ^ 1], n);
        }
        // second loop: d += (alpha_i - beta_i) s_i fused with <y_{i+1}, d>
        for (int i = 0; i < m; ++i, cur ^= 1) {
            const DotFin f{FIN_BETA, 0, i, i == 0 ? 1 : 0, history[i].rho_inv, history.back().rho_inv};
            if (i + 1 < m)
                hipLaunchKernelGGL((k_axpy_dot<REAL, REAL, char, true>), g, b, 0, st, dir, history[i].s, history[i + 1].y, part[cur], nb, d_scalar, f, part[cur ^ 1], n);
            else
                hipLaunchKernelGGL((k_axpy_dot<REAL, REAL, char, false>), g, b, 0, st, dir, history[i].s, (const char*)nullptr, part[cur], nb, d_scalar, f, part[cur ^ 1], n);
        }
        LHIP(hipGetLastError());
        return 0;
    }

    // search_step_size_and_apply, :159-224
    int search_step_size_and_apply()
    {
        SolverBase* b = s->impl;
        double lb_pre = 0.0;
        int rc;
        // the bound before the step is the one enqueued at the end of the previous iteration (nothing has touched the costs since);
        // it is fetched with the first trial's bound
        const bool pre_pending = lb_pending;
        if (!pre_pending && (rc = lower_bound(&lb_pre))) return rc;
        const int m = p.history_size;
        auto rel_change = [&](double* out) -> int {
            double lb;
            int r;
            if (lb_pending) {
                if ((r = b->lower_bound_enqueue(1))) { err = b->err; return r; }
                if ((r = resolve_pending())) return r;
                lb_pre = lb_history.back();
                if ((r = b->lower_bound_fetch(1, &lb))) { err = b->err; return r; }
            } else if ((r = lower_bound(&lb))) return r;
            const double cur_inc = lb - lb_pre;
            const double past_inc = *(lb_history.rbegin() + m - 2) - *(lb_history.rbegin() + m - 1);
            *out = cur_inc / (1e-9 + past_inc);
            return 0;
        };
        double prev_step = 0.0;
        auto apply = [&](double new_step) -> int {
            const double net = new_step - prev_step;
            if (net != 0.0) {
                int r = b->gradient_step_projected(dir, net);
                if (r) { err = b->err; return r; }
                ++last_trials;
            }
            prev_step = new_step;
            last_applied_step = new_step;
            return 0;
        };
        last_trials = 0;
        size_t num_updates = 0;
        double cur = 0.0, best_step = 0.0, best_impr = 0.0;
        do {
            if ((rc = apply(step_size))) return rc;
            if ((rc = rel_change(&cur))) return rc;
            if (best_impr < cur) { best_impr = cur; best_step = step_size; }
            if (cur <= 0.0) step_size *= p.step_size_decrease_factor;
            else if (cur < p.req_rel_lb_increase) step_size *= p.step_size_increase_factor;
            if (num_updates > 5) {
                if (best_impr > p.req_rel_lb_increase / 10.0) {
                    if ((rc = apply(best_step))) return rc;
                } else {
                    if ((rc = apply(0.0))) return rc;
                    unsuccessful += 1;
                }
                return 0;
            }
            num_updates++;
        } while (cur < p.req_rel_lb_increase);
        if (num_updates == 1 && unsuccessful == 0) step_size *= p.step_size_increase_factor;
        unsuccessful = 0;
        return 0;
    }

    // iteration(), :137-157
    int iteration() override
    {
        SolverBase* b = s->impl;
        LHIP(hipSetDevice(b->device));
        int rc;
        double lb;
        if (lb_history.empty()) {
            if ((rc = lower_bound(&lb))) return rc;
            lb_history.push_back(lb);
        }
        bool lbfgs_step;
        // The pending bound doubles as "the bound before the step" of this iteration's search — true only while nobody else has changed the
        // costs since (set_cost, update_costs, gradient_step ... on the solver itself; the reference's search calls lower_bound() afresh,
        // lbfgs_impl.h:165): otherwise it is fetched now, as the history entry it also is, and the search computes its own.
        if (lb_pending && b->cost_epoch != lb_pending_epoch && (rc = resolve_pending())) return rc;
        if (gram) {
            // (cur_g is all 0 here: zeroed at init and by every store pass after its last read; the direction reads the copy in prev_g)
            if ((rc = b->bdds_solution_async(cur_g, 1))) { err = b->err; return rc; }
            // the branch is taken when the history is full AFTER this store: it may complete it
            const bool may = unsuccessful <= 5 && (int)lb_history.size() >= p.history_size && h_count + 1 >= p.history_size;
            if ((rc = store_iterate_gram(may ? 1 : 0))) return rc;
            lbfgs_step = may && h_count >= p.history_size;
            // the direction: its own pass (k_lb_direction), or — where the solver's projection takes a linear combination — formed inside
            // the projection's first pass straight from the history (SolverBase::projection_means_lincomb)
            fused_dir = lbfgs_step && b->projection_fuses_lincomb();
            if (lbfgs_step && !fused_dir && (rc = direction_gram())) return rc;
        } else {
            if ((rc = b->bdds_solution(0, cur_g, 1))) { err = b->err; return rc; }
            if ((rc = store_iterate())) return rc;
            lbfgs_step = update_possible() && (int)lb_history.size() >= p.history_size;  // choose_solver, :409-417
            if (lbfgs_step && (rc = compute_direction())) return rc;
        }
        if (lbfgs_step) {
            // make_dual_feasible(direction), applied inside the steps
            if (gram && fused_dir) {
                // (prev_g holds the current subgradient: the store pass has copied it there)
                const LinComb lc{prev_g, S, Y, (uint64_t)slot, &d_lb->cy[0], &d_lb->cs[0], &d_lb->order[0], p.history_size};
                if ((rc = b->projection_means_lincomb(lc, dir))) { err = b->err; return rc; }
            } else if ((rc = b->projection_means(dir))) { err = b->err; return rc; }
            if ((rc = search_step_size_and_apply())) return rc;
            last_kind = 1;
            ++lbfgs_iterations;
        } else {
            last_kind = 0;
            last_trials = 0;
            last_applied_step = 0.0;
            ++mma_iterations;
        }
        if ((rc = b->iteration(0.5))) { err = b->err; return rc; }
        if ((rc = resolve_pending())) return rc;   // (only after an iteration without an L-BFGS step: the search fetches it otherwise)
        if (gram) {
            if ((rc = b->lower_bound_enqueue(0))) { err = b->err; return rc; }
            lb_history.push_back(0.0);
            lb_pending = true;
            lb_pending_epoch = b->cost_epoch;
        } else {
            if ((rc = lower_bound(&lb))) return rc;
            lb_history.push_back(lb);
        }
        if (lb_history.size() > (size_t)std::max(64, p.history_size + 2)) lb_history.pop_front();
        return 0;
    }
};

thread_local std::string g_lbfgs_error;

}  // namespace

extern "C" {

int bddmma_lbfgs_create(bddmma_lbfgs** out, bddmma_solver* s, const bddmma_lbfgs_params* p)
{
    if (!out || !s || !s->impl) return BDDMMA_ERR_INVALID_ARGUMENT;
    bddmma_lbfgs_params q;
    q.history_size = 5;                 // lbfgs.h:29-33
    q.init_step_size = 1e-6;
    q.req_rel_lb_increase = 1e-6;
    q.step_size_decrease_factor = 0.8;
    q.step_size_increase_factor = 1.1;
    if (p) {
        if (p->history_size > 0) q.history_size = p->history_size;
        if (p->init_step_size > 0) q.init_step_size = p->init_step_size;
        if (p->req_rel_lb_increase > 0) q.req_rel_lb_increase = p->req_rel_lb_increase;
        if (p->step_size_decrease_factor > 0) q.step_size_decrease_factor = p->step_size_decrease_factor;
        if (p->step_size_increase_factor > 0) q.step_size_increase_factor = p->step_size_increase_factor;
    }
    // lbfgs_impl.h:27-31
    if (q.history_size < 2 || q.step_size_decrease_factor >= 1.0 || q.step_size_increase_factor <= 1.0) {
        s->impl->err = "invalid L-BFGS parameters";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    bddmma_lbfgs* l = nullptr;
    int rc;
    if (s->impl->precision == BDDMMA_F32) {
        auto* t = new Lbfgs<float>();
        t->s = s; t->p = q;
        rc = t->init();
        l = t;
    } else {
        auto* t = new Lbfgs<double>();
        t->s = s; t->p = q;
        rc = t->init();
        l = t;
    }
    if (rc) {
        s->impl->err = l->err;
        delete l;
        return rc;
    }
    *out = l;
    return BDDMMA_OK;
}

void bddmma_lbfgs_destroy(bddmma_lbfgs* l) { delete l; }

int bddmma_lbfgs_iteration(bddmma_lbfgs* l)
{
    if (!l) return BDDMMA_ERR_INVALID_ARGUMENT;
    int rc = l->iteration();
    if (rc) l->s->impl->err = l->err;
    return rc;
}

int bddmma_lbfgs_flush(bddmma_lbfgs* l)
{
    if (!l) return BDDMMA_ERR_INVALID_ARGUMENT;
    l->flush();
    return BDDMMA_OK;
}

int bddmma_lbfgs_get_state(const bddmma_lbfgs* l, bddmma_lbfgs_state* out)
{
    if (!l || !out) return BDDMMA_ERR_INVALID_ARGUMENT;
    l->get_state(out);
    return BDDMMA_OK;
}

int bddmma_lbfgs_update_costs(bddmma_lbfgs* l, const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi,
                              int elem_precision, int on_device)
{
    if (!l) return BDDMMA_ERR_INVALID_ARGUMENT;
    l->flush();  // lbfgs_impl.h:343-364
    return l->s->impl->update_costs(lo, n_lo, hi, n_hi, elem_precision, on_device);
}

}  // extern "C"
