// Development probe: register layout of v_mfma_f64_16x16x4_f64 and whether it equals the ascending-k chain of fused
// multiply-adds bit for bit.
// hipcc --offload-arch=gfx950 -O2 tools/dev/mfma_f64_order.hip -o tools/dev/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, const double *C, double *Draw, int clayout)
{
    // A: 16 x 16 (row-major, i x k), B: 16 x 16 (k x j), C: 16 x 16.  Four MFMA steps of K = 4.
    const int l = threadIdx.x;
    double4_t acc;
    for (int v = 0; v < 4; ++v) acc[v] = clayout == 0 ? C[(4 * (l / 16) + v) * 16 + (l % 16)] : C[((l / 16) + 4 * v) * 16 + (l % 16)];
    for (int q = 0; q < 4; ++q) {
        const double a = A[(l % 16) * 16 + 4 * q + l / 16];
        const double b = B[(4 * q + l / 16) * 16 + (l % 16)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) Draw[l * 4 + v] = acc[v];
}
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ void probe32(const float *A, const float *B, float *Draw)
{
    const int l = threadIdx.x;
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 4; ++q) {
        const float a = A[(l % 16) * 16 + 4 * q + l / 16];
        const float b = B[(4 * q + l / 16) * 16 + (l % 16)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) Draw[l * 4 + v] = acc[v];
}
static void run32()
{
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    float *dA, *dB, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    for (int layout = 0; layout < 2; ++layout) {
        long bad = 0, far = 0, total = 0;
        for (int trial = 0; trial < 400; ++trial) {
            std::vector<float> A(256), B(256), D(256);
            const bool sparse = trial & 1;
            for (auto &x : A) x = (sparse && U(rng) < -0.2f) ? 0.0f : U(rng);
            for (auto &x : B) x = (sparse && U(rng) < -0.2f) ? 0.0f : U(rng);
            hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l)
                for (int v = 0; v < 4; ++v) {
                    const int i = layout == 0 ? 4 * (l / 16) + v : (l / 16) + 4 * v, j = l % 16;
                    float asc = 0.f, mag = 0.f;
                    for (int k = 0; k < 16; ++k) asc = __builtin_fmaf(A[i * 16 + k], B[k * 16 + j], asc);
                    for (int k = 0; k < 16; ++k) mag += fabsf(A[i * 16 + k] * B[k * 16 + j]);
                    uint32_t ua, ug;
                    memcpy(&ua, &asc, 4); memcpy(&ug, &D[l * 4 + v], 4);
                    bad += ua != ug; ++total; far += fabsf(D[l * 4 + v] - asc) > 1e-4f * mag;
                }
        }
        printf("f32 16x16x4, layout %d: entries %ld, differ from the ascending-k fmaf chain %ld, off by more than 1e-4 of the magnitude %ld\n", layout, total, bad, far);
    }
}
int main()
{
    run32();
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    for (int layout = 0; layout < 2; ++layout) {
        long bad_asc = 0, bad_skip = 0, total = 0, far = 0;
        for (int trial = 0; trial < 400; ++trial) {
            std::vector<double> A(256), B(256), C(256), D(256);
            const bool sparse = trial & 1;     // half the trials: 40 % zeros in A and B, C = 0 (the tile case)
            for (auto &x : A) x = (sparse && U(rng) < -0.2) ? 0.0 : U(rng) * (trial % 7 == 0 ? 1e-160 : 1.0);
            for (auto &x : B) x = (sparse && U(rng) < -0.2) ? 0.0 : U(rng) * (trial % 7 == 0 ? 1e-160 : 1.0);
            for (auto &x : C) x = sparse ? 0.0 : U(rng);
            hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, layout);
            hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l)
                for (int v = 0; v < 4; ++v) {
                    const int i = layout == 0 ? 4 * (l / 16) + v : (l / 16) + 4 * v, j = l % 16;
                    double asc = C[i * 16 + j], skip = C[i * 16 + j], mag = fabs(C[i * 16 + j]);
                    for (int k = 0; k < 16; ++k) asc = __builtin_fma(A[i * 16 + k], B[k * 16 + j], asc);
                    for (int k = 0; k < 16; ++k) mag += fabs(A[i * 16 + k] * B[k * 16 + j]);
                    for (int k = 0; k < 16; ++k)
                        if (A[i * 16 + k] != 0.0 && B[k * 16 + j] != 0.0) skip = __builtin_fma(A[i * 16 + k], B[k * 16 + j], skip);
                    const double got = D[l * 4 + v];
                    uint64_t ua, us, ug;
                    memcpy(&ua, &asc, 8); memcpy(&us, &skip, 8); memcpy(&ug, &got, 8);
                    bad_asc += ua != ug; bad_skip += us != ug; ++total;
                    far += fabs(got - asc) > 1e-12 * mag;
                }
        }
        printf("layout %d (row = %s): entries %ld, differ from the ascending-k fma chain %ld, from the chain that skips zero operands %ld, off by more than 1e-12 of the products' magnitude %ld\n",
               layout, layout == 0 ? "4*(lane/16)+v" : "lane/16+4*v", total, bad_asc, bad_skip, far);
    }
    return 0;
}
