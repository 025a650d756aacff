// solver_base.hip — the precision-free part of the solver object: SolverBase's own members (profiling, timing), device queries and
// create_solver, which picks the instantiation of SolverT<REAL> (solver_impl.hpp) built by solver_f32.hip / solver_f64.hip.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bdd_mma.h"
#include "layout.hpp"
#include "solver.hpp"

namespace bddmma {

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                                   \
            return BDDMMA_ERR_DEVICE;                                                                  \
        }                                                                                              \
    } while (0)

SolverBase* make_solver_f32();  // solver_f32.hip
SolverBase* make_solver_f64();  // solver_f64.hip

// ---------------------------------------------------------------------------------------------
int SolverBase::synchronize()
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    return BDDMMA_OK;
}

void SolverBase::prof_begin(int kclass)
{
    if (!profiling || !prof_active) return;
    if (ev_used == ev_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { profiling = false; return; }
        ev_pool.push_back({a, b});
        ev_class.push_back(kclass);
    }
    ev_class[ev_used] = kclass;
    (void)hipEventRecord(ev_pool[ev_used].first, stream);
}
void SolverBase::prof_end(int)
{
    if (!profiling || !prof_active) return;
    (void)hipEventRecord(ev_pool[ev_used].second, stream);
    ++ev_used;
}
int SolverBase::set_profiling(int on)
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    profiling = on != 0;
    prof_stride = on > 0 ? (uint32_t)on : 1;
    prof_iter = 0;
    prof_active = profiling;
    ev_used = 0;
    return BDDMMA_OK;
}
int SolverBase::get_profile(bddmma_profile* out)
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    std::memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
        out->launches[ev_class[i]]++;
        out->total_ms[ev_class[i]] += ms;
    }
    return BDDMMA_OK;
}
int SolverBase::time_iterations(double omega, uint64_t n, double* ms)
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipEventRecord(ev_t0, stream));
    for (uint64_t i = 0; i < n; ++i) {
        int rc = iteration(omega);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(ev_t1, stream));
    HIPCHK(hipEventSynchronize(ev_t1));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, ev_t0, ev_t1));
    *ms = f;
    return BDDMMA_OK;
}

int device_count()
{
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

int query_chip(int device, ChipInfo* out, std::string& err)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return BDDMMA_OK;  // create_solver reports it; defaults meanwhile
    hipDeviceProp_t prop;
    const hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return BDDMMA_ERR_DEVICE; }
    if (prop.multiProcessorCount > 0) out->n_cus = (uint32_t)prop.multiProcessorCount;
    if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) out->lds_bytes = (uint32_t)prop.maxSharedMemoryPerMultiProcessor;
    if (prop.multiProcessorCount > 0 && prop.maxThreadsPerMultiProcessor > 0)
        out->max_resident_threads = (uint64_t)prop.multiProcessorCount * (uint64_t)prop.maxThreadsPerMultiProcessor;
    return BDDMMA_OK;
}

int create_solver(SolverBase** out, int precision, int device, const HostLayout& L, const bddmma_options* opts, std::string& err)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        err = std::string("no HIP device available (") + (e == hipSuccess ? "device count 0" : hipGetErrorString(e)) +
              "); this library has no CPU fallback";
        return BDDMMA_ERR_DEVICE;
    }
    if (device < 0 || device >= count) {
        err = "device index out of range";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    std::unique_ptr<SolverBase> s;
    int rc;
    if (precision == BDDMMA_F32 || precision == BDDMMA_F64) {
        s.reset(precision == BDDMMA_F32 ? make_solver_f32() : make_solver_f64());
        s->precision = precision;
        s->device = device;
        rc = s->init_from_layout(L, opts);
    } else {
        err = "precision must be BDDMMA_F32 or BDDMMA_F64";
        return BDDMMA_ERR_INVALID_ARGUMENT;
    }
    if (rc) {
        err = s->err;
        return rc;
    }
    *out = s.release();
    return BDDMMA_OK;
}

}  // namespace bddmma
