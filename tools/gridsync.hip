// gridsync.hip — what does a cross-workgroup hand-over cost on gfx950 when NOTHING but the exchanged words is made coherent?
// (round 5; the fenced grid barrier of round 1, tools/gridbar.hip in the history, cost 17-90 us because the release fence writes the
// XCD's whole L2 back.)  Here: data words are stored / loaded as relaxed agent-scope atomics (global_store / global_load with sc1:
// write-through to, and read from, the point where the eight XCDs' L2s meet), the arrival counter is a relaxed agent-scope
// fetch_add behind `s_waitcnt vmcnt(0)`, the waiters poll it with sc1 loads and s_sleep.  No buffer_wbl2, no buffer_inv.
// Every round each workgroup publishes W words, arrives, waits for all, and checks words of K other workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync.hip -o gridsync && ./gridsync [workgroups] [threads] [rounds] [words per thread]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int MODE>  // 0: relaxed agent-scope words + counter; 1: plain stores + __threadfence() + plain loads behind an acquire fence
__global__ void k_rounds(unsigned* data, unsigned* counter, unsigned* errors, unsigned rounds, unsigned words, unsigned long long* spins)
{
    const unsigned G = gridDim.x, T = blockDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned bad = 0;
    unsigned long long my_spins = 0;
    for (unsigned r = 0; r < rounds; ++r) {
        for (unsigned w = 0; w < words; ++w) {
            const unsigned i = (wg * T + tid) * words + w;
            if (MODE == 0) __hip_atomic_store(&data[i], r * 0x10001u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else data[i] = r * 0x10001u + i;
        }
        if (MODE == 0) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0) expcnt(0): the write-through stores have been acknowledged
        else __threadfence();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (r + 1) * G;
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > 4000000u) { atomicAdd(errors + 1, 1u); break; }  // never hang the box
            }
            my_spins += polls;
        }
        __syncthreads();
        if (MODE == 1) __threadfence();
        for (unsigned k = 1; k <= 3; ++k) {
            const unsigned other = (wg + k * 37u) % G;
            for (unsigned w = 0; w < words; ++w) {
                const unsigned i = (other * T + tid) * words + w;
                const unsigned v = MODE == 0 ? __hip_atomic_load(&data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(&data[i]);
                bad += v != r * 0x10001u + i;
            }
        }
        // everybody must have read before the next round overwrites: second arrival on the same counter would need 2 G per round; instead the
        // data of round r + 1 carries r + 1, and a reader that sees it early reports an error — so rounds alternate between two halves
        data += (r & 1u) ? -(long)(G * T * words) : (long)(G * T * words);
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) atomicAdd(spins, my_spins);
}

int main(int argc, char** argv)
{
    const unsigned G = argc > 1 ? std::atoi(argv[1]) : 512, T = argc > 2 ? std::atoi(argv[2]) : 256, R = argc > 3 ? std::atoi(argv[3]) : 2000,
                   W = argc > 4 ? std::atoi(argv[4]) : 4;
    unsigned *data, *counter, *errors;
    unsigned long long* spins;
    CHK(hipMalloc(&data, 2ull * G * T * W * 4));
    CHK(hipMalloc(&counter, 256));
    CHK(hipMalloc(&errors, 8));
    CHK(hipMalloc(&spins, 8));
    for (int mode = 0; mode < 2; ++mode) {
        CHK(hipMemset(data, 0xff, 2ull * G * T * W * 4));
        CHK(hipMemset(counter, 0, 256));
        CHK(hipMemset(errors, 0, 8));
        CHK(hipMemset(spins, 0, 8));
        hipEvent_t a, b;
        CHK(hipEventCreate(&a));
        CHK(hipEventCreate(&b));
        void* args[] = {&data, &counter, &errors, (void*)&R, (void*)&W, &spins};
        CHK(hipEventRecord(a));
        // cooperative launch: fails instead of deadlocking when the grid is not co-resident
        if (mode == 0) CHK(hipLaunchCooperativeKernel((const void*)k_rounds<0>, dim3(G), dim3(T), args, 0, nullptr));
        else CHK(hipLaunchCooperativeKernel((const void*)k_rounds<1>, dim3(G), dim3(T), args, 0, nullptr));
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a, b));
        unsigned e[2];
        unsigned long long sp;
        CHK(hipMemcpy(e, errors, 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&sp, spins, 8, hipMemcpyDeviceToHost));
        std::printf("%-34s G=%u T=%u words/thread=%u (%.2f MB per round): %.2f us per round, %u wrong words, %u timeouts, %.1f polls per round and workgroup\n",
                    mode == 0 ? "relaxed agent-scope words (sc1)" : "plain words + __threadfence()", G, T, W, G * T * W * 4 / 1e6, ms * 1e3 / R, e[0], e[1],
                    (double)sp / R / G);
    }
    return 0;
}
