// atomics_rate.hip — can the sweeps of a SMALL instance (configs[1]: 1563 packs, 500 k layers, 100 k variables) do the variable <-> layer
// exchange themselves, without the two exchange launches?  They would push one floating-point atomic per layer into a per-variable
// accumulator pair (instead of the staged flush) and pull the pair of the previous pass per layer (instead of the staged load).
// This measures what those two access patterns cost on gfx950 with the shape of configs[1]:
//   mode 0  baseline: the loads of the index table + a coalesced store per layer (what the flush does today)
//   mode 1  one f64 atomic add per layer to acc[2 * var + side] (agent scope)
//   mode 2  the same with workgroup scope (does the instruction / the rate differ?)
//   mode 3  one f32 atomic add per layer
//   mode 4  gather of the 16-byte pair acc[var] per layer (the pull side)
//   mode 5  1 + 4 together (what a fused sweep would add)
//   hipcc --offload-arch=gfx950 -O3 tools/atomics_rate.hip -o build/atomics_rate && ./build/atomics_rate [packs] [layers per pack] [variables]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k_touch(const unsigned* __restrict__ lvar, const float* __restrict__ mm, double* acc, float* accf,
                                                float* __restrict__ out, unsigned layers_per_pack)
{
    const unsigned p = blockIdx.x, lane = threadIdx.x;
    const unsigned base = p * layers_per_pack;
    float sink = 0.f;
    for (unsigned j = lane; j < layers_per_pack; j += 64) {
        const unsigned v = lvar[base + j];
        const float m = mm[base + j];
        if (MODE == 0) out[base + j] = m + (float)v;
        if (MODE == 1 || MODE == 5) __hip_atomic_fetch_add(&acc[2 * v + (m > 0 ? 1 : 0)], (double)(m > 0 ? m : -m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_add(&acc[2 * v + (m > 0 ? 1 : 0)], (double)(m > 0 ? m : -m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) __hip_atomic_fetch_add(&accf[2 * v + (m > 0 ? 1 : 0)], m > 0 ? m : -m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 4 || MODE == 5) {
            const double2 pr = *reinterpret_cast<const double2*>(&acc[2 * (size_t)v + 2 * 131072 * 0]);
            sink += (float)(pr.x + pr.y);
        }
    }
    if (MODE >= 4) out[base + lane] = sink;
}

template <int MODE>
static float run(const unsigned* lvar, const float* mm, double* acc, float* accf, float* out, unsigned packs, unsigned lpp, int reps)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) k_touch<MODE><<<packs, 64>>>(lvar, mm, acc, accf, out, lpp);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_touch<MODE><<<packs, 64>>>(lvar, mm, acc, accf, out, lpp);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

int main(int argc, char** argv)
{
    const unsigned packs = argc > 1 ? std::atoi(argv[1]) : 1563, lpp = argc > 2 ? std::atoi(argv[2]) : 320, nv = argc > 3 ? std::atoi(argv[3]) : 100000;
    const size_t n = (size_t)packs * lpp;
    std::mt19937 g(1);
    std::vector<unsigned> lv(n);
    std::vector<float> m(n);
    for (size_t i = 0; i < n; ++i) { lv[i] = g() % nv; m[i] = (float)((int)(g() % 2001) - 1000) / 64.f; }
    unsigned* d_lv; float *d_m, *d_out, *d_accf; double* d_acc;
    CHK(hipMalloc(&d_lv, n * 4)); CHK(hipMalloc(&d_m, n * 4)); CHK(hipMalloc(&d_out, n * 4 + 256));
    CHK(hipMalloc(&d_acc, (size_t)nv * 16)); CHK(hipMalloc(&d_accf, (size_t)nv * 8));
    CHK(hipMemcpy(d_lv, lv.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_m, m.data(), n * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(d_acc, 0, (size_t)nv * 16)); CHK(hipMemset(d_accf, 0, (size_t)nv * 8));
    std::printf("packs %u x %u layers = %zu layers, %u variables (accumulators %.1f MB)\n", packs, lpp, n, nv, nv * 16 / 1e6);
    std::printf("  0 baseline (loads + coalesced store)   %7.2f us\n", run<0>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    std::printf("  1 f64 atomic per layer, agent scope     %7.2f us\n", run<1>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    std::printf("  2 f64 atomic per layer, workgroup scope %7.2f us\n", run<2>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    std::printf("  3 f32 atomic per layer, agent scope     %7.2f us\n", run<3>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    std::printf("  4 16-byte gather per layer              %7.2f us\n", run<4>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    std::printf("  5 atomic + gather                       %7.2f us\n", run<5>(d_lv, d_m, d_acc, d_accf, d_out, packs, lpp, 200));
    // the atomics must have summed exactly (doubles of small dyadic rationals): check one variable against the host
    std::vector<double> acc(2 * (size_t)nv);
    CHK(hipMemset(d_acc, 0, (size_t)nv * 16));
    k_touch<1><<<packs, 64>>>(d_lv, d_m, d_acc, d_accf, d_out, lpp);
    CHK(hipMemcpy(acc.data(), d_acc, (size_t)nv * 16, hipMemcpyDeviceToHost));
    std::vector<double> ref(2 * (size_t)nv, 0.0);
    for (size_t i = 0; i < n; ++i) ref[2 * lv[i] + (m[i] > 0 ? 1 : 0)] += m[i] > 0 ? m[i] : -m[i];
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += acc[i] != ref[i];
    std::printf("  agent-scope sums wrong: %zu of %zu\n", bad, ref.size());
    return 0;
}
