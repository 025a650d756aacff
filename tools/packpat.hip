// exploration: does the placement of per-pack, per-hop chunks matter for HBM bandwidth?  Each wave copies H chunks of
// 512 B (A -> B) one after the other, like a pack's hops.  Layout 0: [pack][hop][128 floats].  Layout G: packs are
// interleaved in groups of G: [pack / G][hop][pack % G][128 floats] (waves of a group touch one contiguous region per hop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int DEPTH>
__global__ void __launch_bounds__(64) k(const float2* __restrict__ a, float2* __restrict__ b, uint32_t n_packs, int H, int G)
{
    const uint32_t p = blockIdx.x;
    if (p >= n_packs) return;
    const int lane = threadIdx.x;
    auto off = [&](int h) -> size_t {
        if (G <= 1) return ((size_t)p * H + h) * 64 + lane;
        return (((size_t)(p / G) * H + h) * G + (p % G)) * 64 + lane;
    };
    float2 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = a[off(d < H ? d : 0)];
    for (int h = 0; h < H; h += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const float2 x = v[d];
            if (h + d + DEPTH < H) v[d] = a[off(h + d + DEPTH)];
            if (h + d < H) b[off(h + d)] = make_float2(x.x + 1.f, x.y);
        }
    }
}
int main()
{
    const uint32_t P = 78130;  // 10x the benchmark's pack count: 400 MB per array
    const int H = 10;
    const size_t bytes = (size_t)(P + 512) * H * 512;
    float2 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {1, 2, 4, 8, 16, 64, 256}) {
        for (int depth : {1, 2, 5}) {
            auto run = [&]() {
                if (depth == 1) k<1><<<P, 64>>>(a, b, P, H, G);
                else if (depth == 2) k<2><<<P, 64>>>(a, b, P, H, G);
                else k<5><<<P, 64>>>(a, b, P, H, G);
            };
            run();
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("G=%3d depth=%d : %7.1f GB/s\n", G, depth, 2.0 * P * H * 512 * 10 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
