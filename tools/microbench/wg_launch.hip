// Microbenchmark: how fast does the hardware start big workgroups?  Every block records the wall clock at its
// first instruction, then idles `busy_us` microseconds.  Prints the spread of the start times per configuration.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int THREADS>
__global__ void __launch_bounds__(THREADS) wg_kernel(unsigned long long *start, int busy_ticks, int touch)
{
    __shared__ unsigned lds[LDS_BYTES / 4];
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    if (touch)
        for (int x = threadIdx.x; x < LDS_BYTES / 4; x += THREADS) lds[x] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        while (wall_clock64() - t0 < (unsigned long long)busy_ticks) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    if (touch && lds[(threadIdx.x * 7) % (LDS_BYTES / 4)] == 0xFFFFFFFFu) start[0] = 0;
}

template <int LDS_BYTES, int THREADS> void run(int blocks, int busy_us, int touch)
{
    unsigned long long *d;
    (void)hipMalloc(&d, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<unsigned long long> h(blocks);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((wg_kernel<LDS_BYTES, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, busy_us * 100, touch);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    auto at = [&](double f) { return (h[(size_t)(f * (blocks - 1))] - h[0]) / 100.0; };
    printf("LDS %6d B  threads %4d  blocks %5d  busy %3d us  touch %d : kernel %8.1f us   start spread p50 %7.1f  p90 %7.1f  max %7.1f us\n",
           LDS_BYTES, THREADS, blocks, busy_us, touch, ms * 1e3, at(0.5), at(0.9), at(1.0));
    (void)hipFree(d);
}

int main()
{
    run<1024, 1024>(104, 50, 0);
    run<57 * 1024, 1024>(104, 50, 0);
    run<155 * 1024, 1024>(104, 50, 0);
    run<155 * 1024, 1024>(104, 50, 1);
    run<155 * 1024, 256>(104, 50, 0);
    run<57 * 1024, 1024>(921, 30, 0);
    run<57 * 1024, 1024>(921, 30, 1);
    run<57 * 1024, 256>(921, 30, 0);
    run<20 * 1024, 256>(11257, 15, 0);
    run<5 * 1024, 64>(50219, 10, 0);
    run<1024, 64>(50219, 10, 0);
    return 0;
}
