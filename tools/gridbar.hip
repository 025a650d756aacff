// exploration: cost of a grid-wide barrier (atomic counter + agent-scope fences) vs a kernel boundary
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& phase)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release this block's writes to the device
        phase += nblocks;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}
__global__ void k_bar(unsigned* counter, float* data, int n_bar)
{
    unsigned phase = 0;
    for (int i = 0; i < n_bar; ++i) {
        data[blockIdx.x * blockDim.x + threadIdx.x] += 1.f;
        grid_barrier(counter, gridDim.x, phase);
    }
}
__global__ void k_small(float* data) { data[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
int main()
{
    unsigned* c; float* d; hipMalloc(&c, 4); hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) for (int threads : {64, 256}) {
        const int n_bar = 1000;
        hipMemset(c, 0, 4);
        void* args[] = {&c, &d, (void*)&n_bar};
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((void*)k_bar, dim3(blocks), dim3(threads), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("blocks=%4d threads=%3d: %s  %.2f us per grid barrier\n", blocks, threads, hipGetErrorString(e), ms * 1e3 / n_bar);
    }
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) k_small<<<256, 256>>>(d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dependent tiny kernels: %.2f us per launch\n", ms);
    return 0;
}
