// exploration: HBM copy bandwidth vs block size / per-thread contiguity / array size (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
// each thread copies V consecutive float4 per trip (V*16 contiguous bytes), grid-stride
template <int V, bool NT, int BS>
__global__ void __launch_bounds__(BS) kc(v4f* __restrict__ a, const v4f* __restrict__ b, uint64_t n4)
{
    const uint64_t stride = (uint64_t)gridDim.x * BS * V;
    for (uint64_t i = ((uint64_t)blockIdx.x * BS + threadIdx.x) * V; i + V <= n4; i += stride) {
        v4f x[V];
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = NT ? __builtin_nontemporal_load(&b[i + k]) : b[i + k];
#pragma unroll
        for (int k = 0; k < V; ++k) { if (NT) __builtin_nontemporal_store(x[k], &a[i + k]); else a[i + k] = x[k]; }
    }
}
// block-contiguous: a block owns a contiguous chunk, threads interleave inside it (U trips in flight)
template <int U, bool NT, int BS>
__global__ void __launch_bounds__(BS) kb(v4f* __restrict__ a, const v4f* __restrict__ b, uint64_t n4)
{
    const uint64_t per = n4 / gridDim.x;
    const uint64_t base = (uint64_t)blockIdx.x * per;
    for (uint64_t i = threadIdx.x; i + (U - 1) * BS < per; i += (uint64_t)U * BS) {
        v4f x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) x[k] = NT ? __builtin_nontemporal_load(&b[base + i + k * BS]) : b[base + i + k * BS];
#pragma unroll
        for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(x[k], &a[base + i + k * BS]); else a[base + i + k * BS] = x[k]; }
    }
}
template <typename F>
void timeit(const char* name, uint64_t bytes, F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %5lluMiB : %7.1f GB/s\n", name, (unsigned long long)(bytes >> 20), 2.0 * bytes * 20 / (ms * 1e-3) / 1e9);
}
int main()
{
    const uint64_t maxb = 4ull << 30;
    v4f *a, *b; hipMalloc(&a, maxb); hipMalloc(&b, maxb);
    hipMemset(a, 0, maxb); hipMemset(b, 0, maxb);
    for (uint64_t bytes : {1ull << 30, 4ull << 30}) {
        const uint64_t n4 = bytes / 16;
        timeit("hipMemcpyAsync D2D", bytes, [&] { hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0); });
        timeit("kc V=1 BS=256 grid=n/256 (one-shot)", bytes, [&] { kc<1, false, 256><<<(unsigned)(n4 / 256), 256>>>(a, b, n4); });
        timeit("kc V=1 BS=1024 one-shot", bytes, [&] { kc<1, false, 1024><<<(unsigned)(n4 / 1024), 1024>>>(a, b, n4); });
        timeit("kc V=2 BS=256 one-shot", bytes, [&] { kc<2, false, 256><<<(unsigned)(n4 / 512), 256>>>(a, b, n4); });
        timeit("kc V=4 BS=256 one-shot", bytes, [&] { kc<4, false, 256><<<(unsigned)(n4 / 1024), 256>>>(a, b, n4); });
        timeit("kc V=1 BS=256 grid=2048", bytes, [&] { kc<1, false, 256><<<2048, 256>>>(a, b, n4); });
        timeit("kc V=1 BS=512 grid=1024", bytes, [&] { kc<1, false, 512><<<1024, 512>>>(a, b, n4); });
        timeit("kc V=1 BS=1024 grid=512", bytes, [&] { kc<1, false, 1024><<<512, 1024>>>(a, b, n4); });
        timeit("kc V=1 NT BS=256 grid=n/256", bytes, [&] { kc<1, true, 256><<<(unsigned)(n4 / 256), 256>>>(a, b, n4); });
        timeit("kc V=1 NT BS=1024 grid=512", bytes, [&] { kc<1, true, 1024><<<512, 1024>>>(a, b, n4); });
        timeit("kb U=4 BS=256 grid=2048", bytes, [&] { kb<4, false, 256><<<2048, 256>>>(a, b, n4); });
        timeit("kb U=8 BS=256 grid=2048", bytes, [&] { kb<8, false, 256><<<2048, 256>>>(a, b, n4); });
        timeit("kb U=4 NT BS=256 grid=2048", bytes, [&] { kb<4, true, 256><<<2048, 256>>>(a, b, n4); });
        timeit("kb U=4 BS=256 grid=8192", bytes, [&] { kb<4, false, 256><<<8192, 256>>>(a, b, n4); });
        timeit("kb U=4 NT BS=256 grid=8192", bytes, [&] { kb<4, true, 256><<<8192, 256>>>(a, b, n4); });
        timeit("kb U=4 BS=1024 grid=1024", bytes, [&] { kb<4, false, 1024><<<1024, 1024>>>(a, b, n4); });
        timeit("kb U=2 NT BS=1024 grid=2048", bytes, [&] { kb<2, true, 1024><<<2048, 1024>>>(a, b, n4); });
    }
    return 0;
}
