// chunk_stream.hip — does the ORDER in which long-lived waves walk their data decide what HBM delivers?
// Round 5 found every kernel of a large instance (nothing cached) at 4.2-4.6 TB/s while the one-shot STREAM probe gets 6.0-6.2 on the
// same box (and a grid-stride STREAM 5.2, round 3).  The sweeps' access pattern, without any of their arithmetic: one wave per "pack",
// H hops, per hop it reads 512 B from each of two arrays and writes 512 B to each of two others (look-ahead of LA hops).
//   layout 0  pack-major (what layout.cpp builds): a pack's hops are contiguous, so the waves resident at one time touch ~4096 windows per array
//   layout 1  hop-major inside groups of G packs: the hop-h chunks of a group are contiguous, so waves that run in step touch one window
//   layout 2  one-shot: a workgroup per (pack, hop) chunk, no loop (the STREAM probe's shape with this chunk size)
//   hipcc --offload-arch=gfx950 -O3 tools/chunk_stream.hip -o build/chunk_stream && ./build/chunk_stream [packs] [hops] [group] [dynamic LDS bytes per workgroup]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int LAYOUT, int LA>
__global__ void __launch_bounds__(256) k_walk(const float2* __restrict__ a, const float2* __restrict__ b, float2* __restrict__ c, float2* __restrict__ d,
                                               unsigned n_packs, unsigned H, unsigned G)
{
    const unsigned lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_packs) return;
    // chunk index (units of 64 float2 = 512 B) of hop h of pack p
    auto chunk = [&](unsigned h) -> size_t {
        if (LAYOUT == 0) return (size_t)p * H + h;
        const unsigned g = p / G, i = p % G, gs = (g + 1) * G <= n_packs ? G : n_packs - g * G;
        return (size_t)g * G * H + (size_t)h * gs + i;
    };
    float2 ra[LA + 1], rb[LA + 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        ra[i] = a[chunk(i < (int)H ? i : H - 1) * 64 + lane];
        rb[i] = b[chunk(i < (int)H ? i : H - 1) * 64 + lane];
    }
    float carry = 0.f;
    for (unsigned h = 0; h < H; ++h) {
        const unsigned hn = h + LA < H ? h + LA : H - 1;
        ra[LA] = a[chunk(hn) * 64 + lane];
        rb[LA] = b[chunk(hn) * 64 + lane];
        float2 x = ra[0], y = rb[0];
        carry = fminf(carry + x.x, y.y);   // a dependence from hop to hop, as the frontier is
        x.x += carry;
        y.y += carry;
        c[chunk(h) * 64 + lane] = x;
        d[chunk(h) * 64 + lane] = y;
#pragma unroll
        for (int i = 0; i < LA; ++i) { ra[i] = ra[i + 1]; rb[i] = rb[i + 1]; }
    }
}

__global__ void __launch_bounds__(64) k_oneshot(const float2* __restrict__ a, const float2* __restrict__ b, float2* __restrict__ c, float2* __restrict__ d)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    float2 x = a[i], y = b[i];
    x.x += y.y;
    y.y += x.x;
    c[i] = x;
    d[i] = y;
}

template <typename F>
static double time_us(F&& launch, int reps)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3;
}

int main(int argc, char** argv)
{
    const unsigned packs = argc > 1 ? std::atoi(argv[1]) : 125000, H = argc > 2 ? std::atoi(argv[2]) : 10, G = argc > 3 ? std::atoi(argv[3]) : 4096;
    const size_t chunks = (size_t)packs * H, bytes = chunks * 512;
    float2 *a, *b, *c, *d;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&c, bytes)); CHK(hipMalloc(&d, bytes));
    CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
    const double gb = 4.0 * bytes / 1e9;
    std::printf("%u packs x %u hops x 512 B x 4 arrays = %.2f GB per launch; groups of %u packs\n", packs, H, gb, G);
    const dim3 grid((packs + 3) / 4), block(256);
    const unsigned lds = argc > 4 ? std::atoi(argv[4]) : 0;   // dynamic LDS per workgroup of 4 waves: caps the waves a CU holds (40960 -> 16, 32768 -> 20, 0 -> 32)
    for (auto k : {(const void*)&k_walk<0, 1>, (const void*)&k_walk<0, 2>, (const void*)&k_walk<0, 4>, (const void*)&k_walk<1, 1>, (const void*)&k_walk<1, 2>, (const void*)&k_walk<1, 4>})
        CHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    std::printf("dynamic LDS per workgroup: %u B\n", lds);
#define RUN(L_, LA_, name)                                                                                       \
    {                                                                                                             \
        const double us = time_us([&] { hipLaunchKernelGGL((k_walk<L_, LA_>), grid, block, lds, 0, a, b, c, d, packs, H, G); }, 10); \
        std::printf("  %-44s %8.1f us  %6.2f TB/s\n", name, us, gb / us * 1e3);                        \
    }
    RUN(0, 1, "pack-major, look-ahead 1")
    RUN(0, 2, "pack-major, look-ahead 2")
    RUN(0, 4, "pack-major, look-ahead 4")
    RUN(1, 1, "hop-major in groups, look-ahead 1")
    RUN(1, 2, "hop-major in groups, look-ahead 2")
    RUN(1, 4, "hop-major in groups, look-ahead 4")
    {
        const double us = time_us([&] { hipLaunchKernelGGL(k_oneshot, dim3((unsigned)chunks), dim3(64), 0, 0, a, b, c, d); }, 10);
        std::printf("  %-44s %8.1f us  %6.2f TB/s\n", "one-shot (a workgroup per chunk)", us, gb / us * 1e3);
    }
    return 0;
}
