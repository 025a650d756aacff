#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using rsrc_t = __amdgpu_buffer_rsrc_t;
// WG b owns pairs [b*S, (b+1)*S); it stores 12 sub-blocks of 256 lanes starting at its first pair with a descriptor that ends at its last pair
__global__ void k(double2* p, unsigned S, int clamp)
{
    const unsigned b = blockIdx.x, e0 = b * S, e1 = e0 + S;
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, e1 * 16u, 0x00020000);
    const unsigned voff = threadIdx.x * 16u;
    using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0));
    const double2 v = make_double2((double)b, (double)b);
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        unsigned e = e0 + u * 256u;
        if (clamp) e = e < e1 ? e : e1;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, e * 16u, 0);
    }
}
int main(int argc, char** argv)
{
    const int clamp = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned S = 100, B = 4096, N = S * B + 4096;
    double2* d;
    hipMalloc(&d, N * 16);
    std::vector<double2> h(N);
    long bad = 0;
    for (int it = 0; it < 300; ++it) {
        hipMemset(d, 0xFF, N * 16);
        hipLaunchKernelGGL(k, dim3(B), dim3(256), 0, 0, d, S, clamp);
        hipMemcpy(h.data(), d, N * 16, hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < S * B; ++i)
            if (h[i].x != (double)(i / S) || h[i].y != (double)(i / S)) { if (bad < 5) printf("it %d pair %u: %g %g (owner %u)\n", it, i, h[i].x, h[i].y, i / S); ++bad; }
    }
    printf("clamp %d: %ld bad pairs\n", clamp, bad);
    return 0;
}
