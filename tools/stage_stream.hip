// stage_stream.hip — the sweeps' memory traffic without their arithmetic, WITH the cooperative staging: does the staging (a quad's 2 560 entries
// read through a sorted index table in runs of a few entries, written back the same way) pull the achievable rate from the walker's
// 5.5-6.1 TB/s (tools/chunk_stream.hip) down to the 4.1-4.6 TB/s every kernel of a large instance shows (profiles/r05_hbm_only.txt)?
// A workgroup = 4 waves = 4 packs (a quad).  mode bits: 1 = the hop streams (per wave and hop 512 B from each of two arrays, 512 B to each of
// two others, look-ahead 1), 2 = staging loads (index table 4 B + 2 B per item, then the 8-byte pairs through it, barrier before the hops),
// 4 = staging write-back (4 B per item through the same indices, behind a barrier after the hops).
// Entries are laid out as the solver does: (bin, quad, layer) with a random bin per layer, so a quad's sorted items form runs of
// layers_per_quad / bins entries.
//   hipcc --offload-arch=gfx950 -O3 tools/stage_stream.hip -o build/stage_stream && ./build/stage_stream [packs] [hops] [bins]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr int ITEMS = 10;  // items per thread: 256 threads x 10 = 2 560 layers per quad (4 packs x 10 hops x 64 layers)

template <int MODE, bool XCD_MAP>
__global__ void __launch_bounds__(256) k_quad(const float2* __restrict__ a, const float2* __restrict__ b, float2* __restrict__ c, float2* __restrict__ d,
                                               const unsigned* __restrict__ cse, const unsigned short* __restrict__ css, const float2* __restrict__ pairs,
                                               float* __restrict__ mm, unsigned n_packs, unsigned H)
{
    __shared__ float2 sD[256 * ITEMS];
    // workgroup b runs on XCD b % 8; XCD_MAP: every XCD gets a contiguous eighth of the quads (neighbouring quads' runs share L2 lines), as the solver does
    const unsigned n_quads = gridDim.x, per = (n_quads + 7) / 8;
    const unsigned tid = threadIdx.x, lane = tid & 63, quad = XCD_MAP ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x, p = quad * 4 + (tid >> 6);
    if (quad >= n_quads) return;
    unsigned e[ITEMS];
    unsigned short sl[ITEMS];
    if (MODE & 6) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            e[u] = cse[(size_t)quad * 256 * ITEMS + u * 256 + tid];
            sl[u] = css[(size_t)quad * 256 * ITEMS + u * 256 + tid];
        }
    }
    if (MODE & 2) {
        float2 v[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) v[u] = pairs[e[u]];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) sD[sl[u]] = v[u];
        __syncthreads();
    }
    float carry = (MODE & 2) ? sD[tid].x : 0.f;
    if ((MODE & 1) && p < n_packs) {
        const size_t base = (size_t)p * H;
        float2 ra = a[base * 64 + lane], rb = b[base * 64 + lane];
        for (unsigned h = 0; h < H; ++h) {
            const unsigned hn = h + 1 < H ? h + 1 : H - 1;
            const float2 na = a[(base + hn) * 64 + lane], nb = b[(base + hn) * 64 + lane];
            float2 x = ra, y = rb;
            carry = fminf(carry + x.x, y.y);
            x.x += carry;
            y.y += carry;
            c[(base + h) * 64 + lane] = x;
            d[(base + h) * 64 + lane] = y;
            if (MODE & 2) sD[(tid >> 6) * 64 * ITEMS + h * 64 + lane].x = carry;
            ra = na;
            rb = nb;
        }
    }
    if (MODE & 4) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) mm[e[u]] = (MODE & 2) ? sD[sl[u]].x : carry;
    }
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
    const unsigned packs = argc > 1 ? std::atoi(argv[1]) : 31248, H = 10, bins = argc > 3 ? std::atoi(argv[3]) : 977;
    const unsigned quads = packs / 4;
    const size_t layers = (size_t)quads * 256 * ITEMS, chunks = (size_t)packs * H;
    // entry order (bin, quad, layer): rank of every layer; per quad the items sorted by entry and the LDS slot (= the layer's index in the quad)
    std::mt19937_64 g(7);
    std::vector<unsigned> bin(layers);
    for (auto& x : bin) x = (unsigned)(g() % bins);
    std::vector<unsigned> order(layers);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return bin[x] < bin[y]; });  // stable: (quad, layer) order inside a bin
    std::vector<unsigned> rank(layers);
    for (size_t i = 0; i < layers; ++i) rank[order[i]] = (unsigned)i;
    std::vector<unsigned> cse(layers);
    std::vector<unsigned short> css(layers);
    double runs = 0;
    for (unsigned q = 0; q < quads; ++q) {
        const size_t o = (size_t)q * 256 * ITEMS;
        std::vector<unsigned> idx(256 * ITEMS);
        std::iota(idx.begin(), idx.end(), 0u);
        std::sort(idx.begin(), idx.end(), [&](unsigned x, unsigned y) { return rank[o + x] < rank[o + y]; });
        for (unsigned i = 0; i < 256 * ITEMS; ++i) {
            cse[o + i] = rank[o + idx[i]];
            css[o + i] = (unsigned short)idx[i];
            if (i == 0 || cse[o + i] != cse[o + i - 1] + 1) runs += 1;
        }
    }
    std::printf("%u packs (%u quads) x %u hops, %zu layers in %u bins: staged runs of %.2f entries on average\n", packs, quads, H, layers, bins, layers / runs);
    float2 *a, *b, *c, *d, *pairs;
    float* mm;
    unsigned* d_cse;
    unsigned short* d_css;
    CHK(hipMalloc(&a, chunks * 512)); CHK(hipMalloc(&b, chunks * 512)); CHK(hipMalloc(&c, chunks * 512)); CHK(hipMalloc(&d, chunks * 512));
    CHK(hipMalloc(&pairs, layers * 8)); CHK(hipMalloc(&mm, layers * 4)); CHK(hipMalloc(&d_cse, layers * 4)); CHK(hipMalloc(&d_css, layers * 2));
    CHK(hipMemset(a, 0, chunks * 512)); CHK(hipMemset(b, 0, chunks * 512)); CHK(hipMemset(pairs, 0, layers * 8));
    CHK(hipMemcpy(d_cse, cse.data(), layers * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_css, css.data(), layers * 2, hipMemcpyHostToDevice));
    const double gb_stream = 4.0 * chunks * 512 / 1e9, gb_ld = layers * (4 + 2 + 8) / 1e9, gb_st = layers * 4 / 1e9;
#define RUN(M_, X_, name, gb)                                                                                                                     \
    {                                                                                                                                          \
        const double us = time_us([&] { hipLaunchKernelGGL((k_quad<M_, X_>), dim3(quads), dim3(256), 0, 0, a, b, c, d, d_cse, d_css, pairs, mm, packs, H); }, 10); \
        std::printf("  %-58s %8.1f us  %6.2f GB  %5.2f TB/s\n", name, us, (double)(gb), (gb) / us * 1e3);                                     \
    }
    for (int x = 0; x < 2; ++x) {
        std::printf(x ? " XCD-aware map (an eighth of the quads per XCD):\n" : " workgroup b -> quad b (neighbouring quads on different XCDs):\n");
#define RUNX(M_, name, gb) if (x) RUN(M_, true, name, gb) else RUN(M_, false, name, gb)
        RUNX(1, "hop streams only", gb_stream)
        RUNX(2, "staging loads only (tables -> pairs -> LDS)", gb_ld)
        RUNX(6, "staging loads + write-back only", gb_ld + gb_st)
        RUNX(3, "streams + staging loads", gb_stream + gb_ld)
        RUNX(7, "streams + staging loads + write-back (a sweep's traffic)", gb_stream + gb_ld + gb_st)
    }
    return 0;
}
