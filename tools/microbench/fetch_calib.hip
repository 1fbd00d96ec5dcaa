// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of this library (MI355X_MICROARCH.md, HBM section: the
// counter reads exactly half the bytes of a wide coalesced streaming read; "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel touches a known number of DISTINCT 64-byte lines of a table far
// larger than the Infinity Cache (4 GiB), each line once, so the bytes that must come from HBM are known:
//   stream16   16 B per lane, coalesced            (the conversion's streams, c_vals stores' counterpart)
//   stream4     4 B per lane, coalesced            (index streams: pairs_a / pairs_b, offsets)
//   gather8     one 8-byte word per lane from a random line          (operand value gathers of step 3)
//   gather4     one 4-byte word per lane from a random line          (record-word gathers)
//   gather64x16 sixteen lanes read the sixteen words of one random 64-byte record (tile_rec gathers of a whole tile)
//   gather32x2  a lane reads 32 consecutive bytes (two 16-byte loads) of a random line  (tile masks in step 2)
// Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- ./fetch_calib
// and compare FETCH_SIZE (KiB) * 1024 with the "expect" column this program prints (tools/fetch_calib_report.py does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
// a bijection on [0, 2^bits): every line index is produced exactly once
__device__ __forceinline__ uint64_t perm(uint64_t i, int bits)
{
    const uint64_t m = (1ull << bits) - 1;
    i = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & m;      // odd multiplier: a bijection mod 2^bits
    i ^= i >> (bits / 2);
    i = (i * 0xD6E8FEB86659FD93ull) & m;
    i ^= i >> (bits / 2);
    return i & m;
}
__global__ void stream16(const uint4 *__restrict__ t, size_t n16, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream4(const unsigned *__restrict__ t, size_t n4, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= t[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather8(const uint64_t *__restrict__ t, int line_bits, size_t ngather, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (; i < ngather; i += (size_t)gridDim.x * blockDim.x) acc ^= t[perm(i, line_bits) * 8 + (i & 7)];
    if (acc == 0x12345678u) *sink = (unsigned)acc;
}
__global__ void gather4(const unsigned *__restrict__ t, int line_bits, size_t ngather, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < ngather; i += (size_t)gridDim.x * blockDim.x) acc ^= t[perm(i, line_bits) * 16 + (i & 15)];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather64x16(const unsigned *__restrict__ t, int line_bits, size_t nrec, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; (i >> 4) < nrec; i += (size_t)gridDim.x * blockDim.x) acc ^= t[perm(i >> 4, line_bits) * 16 + (i & 15)];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather32x2(const uint4 *__restrict__ t, int line_bits, size_t ngather, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < ngather; i += (size_t)gridDim.x * blockDim.x) {
        const size_t l = perm(i, line_bits) * 4 + 2 * (i & 1);
        const uint4 a = t[l], b = t[l + 1];
        acc ^= a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const int line_bits = 26;                                   // 2^26 lines of 64 B = 4 GiB
    const size_t bytes = (size_t)64 << line_bits;
    void *t = nullptr;
    unsigned *sink = nullptr;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(t, 1, bytes);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 16, block = 256;
    const size_t ng = (size_t)1 << 24;                          // 16 Mi gathers -> 16 Mi distinct lines = 1 GiB of lines
    printf("kernel,expect_bytes,what\n");
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4 *)t, bytes / 4 / 16, sink);   // the first GiB
    printf("stream16,%zu,1 GiB read once 16 B per lane\n", bytes / 4);
    hipLaunchKernelGGL(stream4, grid, block, 0, 0, (const unsigned *)t + bytes / 8, bytes / 4 / 4, sink);   // another GiB
    printf("stream4,%zu,1 GiB read once 4 B per lane\n", bytes / 4);
    hipLaunchKernelGGL(gather8, grid, block, 0, 0, (const uint64_t *)t, line_bits, ng, sink);
    printf("gather8,%zu,%zu distinct 64-byte lines, 8 B used of each\n", ng * 64, ng);
    hipLaunchKernelGGL(gather4, grid, block, 0, 0, (const unsigned *)t, line_bits, ng, sink);
    printf("gather4,%zu,%zu distinct 64-byte lines, 4 B used of each\n", ng * 64, ng);
    hipLaunchKernelGGL(gather64x16, grid, block, 0, 0, (const unsigned *)t, line_bits, ng, sink);
    printf("gather64x16,%zu,%zu distinct 64-byte records read whole by 16 lanes\n", ng * 64, ng);
    hipLaunchKernelGGL(gather32x2, grid, block, 0, 0, (const uint4 *)t, line_bits, ng, sink);
    printf("gather32x2,%zu,%zu distinct 64-byte lines, 32 B used of each\n", ng * 64, ng);
    (void)hipDeviceSynchronize();
    (void)hipFree(t);
    (void)hipFree(sink);
    return 0;
}
