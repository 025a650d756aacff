// gridsync.hip — what does a cross-workgroup hand-over cost on gfx950 when NOTHING but the exchanged words is made coherent?
// (round 5; the fenced grid barrier of round 1, tools/gridbar.hip in the history, cost 17-90 us because the release fence writes the
// XCD's whole L2 back.)  Data words are stored / loaded with sc1 (write-through to, and read from, the point where the eight XCDs' L2s
// meet), arrival counters are relaxed atomics behind `s_waitcnt vmcnt(0)`, waiters poll with s_sleep.  No buffer_wbl2, no buffer_inv.
//   mode 0  flat: one agent-scope counter, everybody arrives on it and polls it
//   mode 1  two levels: a counter per XCD in that XCD's L2 (atomic without sc1, polled with sc0 loads), the last arriver of an XCD
//           arrives on the agent-scope counter, polls it and releases its XCD's flag
//   mode 2  fenced: plain words, __threadfence() either side (what round 1 measured)
// Every round each workgroup publishes W words per thread, arrives, waits for all, and checks words of 3 other workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync.hip -o build/gridsync && ./build/gridsync [workgroups] [threads] [rounds] [words per thread] [data: 0 none, 1 dword atomics, 2 16-byte buffer ops]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000); }
constexpr int AUX_SC1 = 1 << 4, AUX_SC0 = 1;

struct Ctl {
    unsigned global_count;   // agent scope
    unsigned pad0[31];
    unsigned xcd_count[8][32];   // one line per XCD: [x][0] arrivals, [x][16] release flag (round number)
};

template <int MODE, int DATA>
__global__ void k_rounds(unsigned* data, Ctl* ctl, unsigned* errors, unsigned rounds, unsigned words, unsigned long long* spins, unsigned* xcd_sizes)
{
    const unsigned G = gridDim.x, T = blockDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID
    const unsigned n_local = xcd_sizes[xcc];
    unsigned bad = 0;
    unsigned long long my_spins = 0;
    const unsigned half = G * T * words;
    for (unsigned r = 0; r < rounds; ++r) {
        if (__hip_atomic_load(errors + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // somebody timed out: give up together
        unsigned* d = data + (r & 1u) * half;
        const rsrc_t rs = make_rsrc(d, half * 4);
        if (DATA == 1) {
            for (unsigned w = 0; w < words; ++w) {
                const unsigned i = (wg * T + tid) * words + w;
                if (MODE != 2) __hip_atomic_store(&d[i], r * 0x10001u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else d[i] = r * 0x10001u + i;
            }
        } else if (DATA == 2) {
            for (unsigned w = 0; w < words; w += 4) {
                const unsigned i = (wg * T * words) + w * T + tid * 4;
                u4 v;
                v.x = r * 0x10001u + i; v.y = v.x + 1; v.z = v.x + 2; v.w = v.x + 3;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, i * 4, 0, MODE != 2 ? AUX_SC1 : 0);
            }
        }
        if (MODE != 2) __builtin_amdgcn_s_waitcnt(0);  // the write-through stores have been acknowledged
        else __threadfence();
        __syncthreads();
        if (tid == 0) {
            unsigned polls = 0;
            if (MODE == 1) {
                const unsigned mine = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (mine == (r + 1) * n_local - 1) {   // last of this XCD
                    __hip_atomic_fetch_add(&ctl->global_count, n_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(&ctl->global_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (r + 1) * G) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++polls > 100000u || errors[1]) { atomicAdd(errors + 1, 1u); break; }
                    }
                    __hip_atomic_store(&ctl->xcd_count[xcc][16], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    while (__hip_atomic_load(&ctl->xcd_count[xcc][16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < r + 1) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++polls > 100000u || errors[1]) { atomicAdd(errors + 1, 1u); break; }
                    }
                }
            } else {
                __hip_atomic_fetch_add(&ctl->global_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&ctl->global_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (r + 1) * G) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > 100000u || errors[1]) { atomicAdd(errors + 1, 1u); break; }  // never hang the box
                }
            }
            my_spins += polls;
        }
        __syncthreads();
        if (MODE == 2) __threadfence();
        if (DATA) {
            for (unsigned k = 1; k <= 3; ++k) {
                const unsigned other = (wg + k * 37u) % G;
                if (DATA == 1) {
                    for (unsigned w = 0; w < words; ++w) {
                        const unsigned i = (other * T + tid) * words + w;
                        const unsigned v = MODE != 2 ? __hip_atomic_load(&d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(&d[i]);
                        bad += v != r * 0x10001u + i;
                    }
                } else {
                    for (unsigned w = 0; w < words; w += 4) {
                        const unsigned i = (other * T * words) + w * T + tid * 4;
                        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 4, 0, MODE != 2 ? AUX_SC1 : 0);
                        const unsigned e = r * 0x10001u + i;
                        bad += (v.x != e) + (v.y != e + 1) + (v.z != e + 2) + (v.w != e + 3);
                    }
                }
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) atomicAdd(spins, my_spins);
}

__global__ void k_census(unsigned* xcd_sizes, unsigned* mismatch)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    if (threadIdx.x == 0) {
        atomicAdd(&xcd_sizes[xcc], 1u);
        if (xcc != blockIdx.x % 8) atomicAdd(mismatch, 1u);
    }
}

int main(int argc, char** argv)
{
    const unsigned G = argc > 1 ? std::atoi(argv[1]) : 512, T = argc > 2 ? std::atoi(argv[2]) : 256, R = argc > 3 ? std::atoi(argv[3]) : 2000,
                   W = argc > 4 ? std::atoi(argv[4]) : 4;
    const int DATA = argc > 5 ? std::atoi(argv[5]) : 2;
    unsigned *data, *errors, *xcd_sizes;
    Ctl* ctl;
    unsigned long long* spins;
    CHK(hipMalloc(&data, 2ull * G * T * W * 4));
    CHK(hipMalloc(&ctl, sizeof(Ctl)));
    CHK(hipMalloc(&errors, 8));
    CHK(hipMalloc(&spins, 8));
    CHK(hipMalloc(&xcd_sizes, 64));
    CHK(hipMemset(xcd_sizes, 0, 64));
    // which XCD does a workgroup of this grid shape land on?  (the same grid is launched again below; the census is what mode 1 relies on)
    void* cargs[] = {&xcd_sizes, (void*)nullptr};
    unsigned* mism = xcd_sizes + 8;
    cargs[1] = &mism;
    CHK(hipLaunchCooperativeKernel((const void*)k_census, dim3(G), dim3(T), cargs, 0, nullptr));
    unsigned h[9];
    CHK(hipMemcpy(h, xcd_sizes, 36, hipMemcpyDeviceToHost));
    std::printf("census G=%u: workgroups per XCC id %u %u %u %u %u %u %u %u; blockIdx %% 8 != XCC id for %u workgroups\n", G, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    std::fflush(stdout);
    for (int mode = 0; mode < 3; ++mode) {
        CHK(hipMemset(data, 0xff, 2ull * G * T * W * 4));
        CHK(hipMemset(ctl, 0, sizeof(Ctl)));
        CHK(hipMemset(errors, 0, 8));
        CHK(hipMemset(spins, 0, 8));
        hipEvent_t a, b;
        CHK(hipEventCreate(&a));
        CHK(hipEventCreate(&b));
        void* args[] = {&data, &ctl, &errors, (void*)&R, (void*)&W, &spins, &xcd_sizes};
        const void* fn[3][3] = {{(const void*)k_rounds<0, 0>, (const void*)k_rounds<0, 1>, (const void*)k_rounds<0, 2>},
                                {(const void*)k_rounds<1, 0>, (const void*)k_rounds<1, 1>, (const void*)k_rounds<1, 2>},
                                {(const void*)k_rounds<2, 0>, (const void*)k_rounds<2, 1>, (const void*)k_rounds<2, 2>}};
        CHK(hipEventRecord(a));
        CHK(hipLaunchCooperativeKernel(fn[mode][DATA], dim3(G), dim3(T), args, 0, nullptr));  // fails instead of deadlocking when the grid is not co-resident
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a, b));
        unsigned e[2];
        unsigned long long sp;
        CHK(hipMemcpy(e, errors, 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&sp, spins, 8, hipMemcpyDeviceToHost));
        const char* names[3] = {"flat agent-scope counter", "per-XCD L2 counters + one agent", "plain words + __threadfence()"};
        std::printf("%-34s G=%u T=%u data=%d words/thread=%u (%.2f MB per round): %.2f us per round, %u wrong words, %u timeouts, %.1f polls per round and workgroup\n",
                    names[mode], G, T, DATA, W, DATA ? G * T * W * 4 / 1e6 : 0.0, ms * 1e3 / R, e[0], e[1], (double)sp / R / G);
        std::fflush(stdout);
    }
    return 0;
}
