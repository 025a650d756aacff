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
#include <deque>
#include <string>
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
    double step_size = 0;
    int unsuccessful = 0;
    bool prev_stored = false;
    // diagnostics (bddmma_lbfgs_get_state): what the last iteration() did
    int last_kind = 0, last_trials = 0;
    double last_applied_step = 0;
    uint64_t mma_iterations = 0, lbfgs_iterations = 0;
    uint32_t n = 0;
    hipStream_t st = nullptr;

    int device = 0;  // cached: the wrapped solver may already be gone when the wrapper is destroyed
    ~Lbfgs() override
    {
        (void)hipSetDevice(device);
        for (void* q : allocs) (void)hipFree(q);
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
        st = (hipStream_t)b->stream_handle();
        step_size = p.init_step_size;
        int rc;
        if ((rc = alloc(&prev_x, n)) || (rc = alloc(&cur_x, n)) || (rc = alloc(&dir, n)) || (rc = alloc(&prev_g, n)) ||
            (rc = alloc(&cur_g, n)) || (rc = alloc(&d_partial, 2048)) || (rc = alloc(&d_scalar, SC_COUNT)))
            return rc;
        if (p.history_size >= SC_DOT) { err = "history size must be < 32"; return BDDMMA_ERR_INVALID_ARGUMENT; }
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
        out->history_entries = (int32_t)history.size();
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
        for (auto& h : history) free_slots.push_back(h);
        history.clear();
        prev_stored = false;
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
        double lb_pre;
        int rc;
        if ((rc = lower_bound(&lb_pre))) return rc;
        const int m = p.history_size;
        auto rel_change = [&](double* out) -> int {
            double lb;
            int r = lower_bound(&lb);
            if (r) return r;
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
        if ((rc = b->bdds_solution(0, cur_g, 1))) { err = b->err; return rc; }
        if ((rc = store_iterate())) return rc;
        if (update_possible() && (int)lb_history.size() >= p.history_size) {  // choose_solver, :409-417
            if ((rc = compute_direction())) return rc;
            if ((rc = b->projection_means(dir))) { err = b->err; return rc; }  // make_dual_feasible(direction), applied inside the steps
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
        if ((rc = lower_bound(&lb))) return rc;
        lb_history.push_back(lb);
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
