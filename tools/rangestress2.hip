// Do range-dropped buffer stores (voffset < num_records, voffset + soffset >= num_records: every lane dropped) disturb lines that a
// workgroup on ANOTHER XCD writes at the same time?  Even blocks write `it` over their slice; odd blocks only issue dropped stores
// that would land in the neighbour's slice.  The slices held it - 1 from the previous launch.
#include <hip/hip_runtime.h>
#include <cstdio>
using rsrc_t = __amdgpu_buffer_rsrc_t;
__global__ void k(double2* p, unsigned S, double val, int mode)
{
    const unsigned b = blockIdx.x, owner = b & ~1u, e0 = owner * S;
    rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0, 0);
    using u4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(r0, 0, 0, 0));
    const double2 v = make_double2(val, val);
    if ((b & 1u) == 0) {
        rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (e0 + S) * 16u, 0x00020000);
        for (unsigned s = 0; s < S; s += 256) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, threadIdx.x * 16u, (e0 + s) * 16u, 0);
    } else if (mode == 1) {
        rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, e0 * 16u, 0x00020000);  // ends BEFORE the slice
        const double2 junk = make_double2(-777.0, -777.0);
        for (int rep = 0; rep < 4; ++rep)
            for (unsigned s = 0; s < S; s += 256) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, junk), r, threadIdx.x * 16u, (e0 + s) * 16u, 0);
    }
}
__global__ void check(const double2* p, unsigned S, unsigned n, double val, unsigned* bad)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ((i / S) & 1u) == 0 && (p[i].x != val || p[i].y != val)) atomicAdd(bad, 1u);
}
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const unsigned S = 256, B = 4096, N = S * B;
    double2* d; unsigned* bad;
    hipMalloc(&d, N * 16); hipMalloc(&bad, 4);
    hipMemset(bad, 0, 4);
    hipMemset(d, 0, N * 16);
    for (int it = 1; it <= 6000; ++it) {
        hipLaunchKernelGGL(k, dim3(B), dim3(256), 0, 0, d, S, (double)it, mode);
        hipLaunchKernelGGL(check, dim3(N / 256), dim3(256), 0, 0, d, S, N, (double)it, bad);
    }
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d: bad pairs %u (of %.3g slice-launches)\n", mode, h, 6000.0 * B / 2);
    return 0;
}
