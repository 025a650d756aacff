// multi_stream.hip — what does HBM deliver when one kernel reads K arrays at once (the L-BFGS store pass reads 14 and writes 4, and runs at 4.6 TB/s
// where a three-array STREAM kernel gets 6.2)?  K read streams of N floats each (16-byte loads, non-temporal or not), one write stream; grid-stride
// blocks of 256 threads as k_lb_store_gram.  Layout 0: K separate arrays.  Layout 1: tiles of T elements, the K arrays' tiles adjacent
// ([tile][k][T]): every array still contiguous inside a tile (full-line accesses), but the K streams of a block touch one 16 K-element window.
//   hipcc --offload-arch=gfx950 -O3 tools/multi_stream.hip -o build/multi_stream && ./build/multi_stream [N = 5250000] [blocks = 768]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, bool NT, int LAYOUT>
__global__ void __launch_bounds__(256) k_read(const float* __restrict__ base, size_t stride, float* __restrict__ out, size_t n4, unsigned tile4)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += step) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f4* p;
            if (LAYOUT == 0) p = reinterpret_cast<const f4*>(base + (size_t)k * stride) + c;
            else p = reinterpret_cast<const f4*>(base) + ((c / tile4) * K + k) * (size_t)tile4 + (c % tile4);
            const f4 v = NT ? __builtin_nontemporal_load(p) : *p;
            acc += v;
        }
        if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(out) + c);
        else reinterpret_cast<f4*>(out)[c] = acc;
    }
}

template <int K, bool NT, int LAYOUT>
static void run(const float* base, size_t stride, float* out, size_t n, unsigned blocks, unsigned tile)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read<K, NT, LAYOUT>), dim3(blocks), dim3(256), 0, 0, base, stride, out, n / 4, tile / 4);
    CHK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<K, NT, LAYOUT>), dim3(blocks), dim3(256), 0, 0, base, stride, out, n / 4, tile / 4);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)(K + 1) * n * 4;
    std::printf("K=%2d %s layout %d (tile %u): %7.1f us  %5.2f TB/s\n", K, NT ? "nt" : "  ", LAYOUT, tile, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv)
{
    const size_t n = (argc > 1 ? std::atoll(argv[1]) : 5250000) / 16384 * 16384;
    const unsigned blocks = argc > 2 ? std::atoi(argv[2]) : 768;
    constexpr int KMAX = 16;
    float *base, *out;
    CHK(hipMalloc(&base, (size_t)KMAX * n * 4 + 4096)); CHK(hipMalloc(&out, n * 4 + 4096));
    CHK(hipMemset(base, 0, (size_t)KMAX * n * 4)); CHK(hipMemset(out, 0, n * 4));
    std::printf("n = %zu floats per array (%.1f MB), %u blocks of 256\n", n, n * 4 / 1e6, blocks);
    run<2, false, 0>(base, n, out, n, blocks, 16384);
    run<2, true, 0>(base, n, out, n, blocks, 16384);
    run<6, true, 0>(base, n, out, n, blocks, 16384);
    run<10, true, 0>(base, n, out, n, blocks, 16384);
    run<14, false, 0>(base, n, out, n, blocks, 16384);
    run<14, true, 0>(base, n, out, n, blocks, 16384);
    run<14, true, 1>(base, n, out, n, blocks, 16384);
    run<14, true, 1>(base, n, out, n, blocks, 4096);
    run<14, true, 1>(base, n, out, n, blocks, 1024);
    run<14, true, 0>(base, n, out, n, 4096, 16384);
    run<14, true, 1>(base, n, out, n, 4096, 4096);
    return 0;
}
