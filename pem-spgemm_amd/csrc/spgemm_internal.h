// spgemm_internal.h -- what the translation units of the three-step hot path share (step1.hip, step2.hip, step3.hip,
// export.hip, spgemm.hip): constants that tie a producer kernel to its consumer, the repeat-pass size check, the fused
// multiply-add of step 3, and the host-side step drivers.
#pragma once
#include "pem_internal.h"
#include <algorithm>
#include <chrono>

namespace pem {

constexpr int S3_CHUNK = 256;   // C entries per chunk of step 3's work index (= entries one wave takes, or half of them)
constexpr int S2_GROUP = 256;   // C tiles per group of step 2's entry counts

// repeat pass on an unchanged plan: the sizes the host assumed (from the previous pass) against what this pass computed
// counters of a pass of step 1 (plan->bin_count, ints): populations of the five row bins and of the oversized rows, the
// oversized rows' live products (total, largest row), and the allocator of the live list
enum { BC_BIN0 = 0, BC_XL_ROWS = 5, BC_XL_TOTAL = 6, BC_XL_MAX = 7, BC_SEGS = 8 /* segments of the big rows */, BC_FAULT = 9, BC_BUMP = 10 /* 64-bit: ints 10-11 */, BC_INTS = 16 };

struct WarmCheck {
    int on;
    long long P, Pall, TC, nnz, nxl;
    int c[5];
    int nsegs;    int *host_flags;     // where set: the pass's status flags are left in host memory by the checking thread (no copy node after the pass)
};
__device__ __forceinline__ void warm_check(const WarmCheck &w, const long long *__restrict__ d_scalars, const int *__restrict__ bin_count,
                                           int *__restrict__ flags)
{
    if (d_scalars[0] != w.P || d_scalars[1] != w.TC || d_scalars[2] != w.nnz || d_scalars[3] != w.Pall || bin_count[0] != w.c[0] ||
        bin_count[1] != w.c[1] || bin_count[2] != w.c[2] || bin_count[3] != w.c[3] || bin_count[4] != w.c[4] || bin_count[BC_XL_TOTAL] != w.nxl || bin_count[BC_SEGS] != w.nsegs)
        flags[FLAG_CAPACITY] = 1;
    if (w.host_flags) {
        // the caller guarantees that nothing after this thread sets a flag in this pass (s2_offsets_kernel + step 3: none do)
        for (int i = 0; i < NUM_FLAGS; ++i) w.host_flags[i] = flags[i];
    }
}

// one fused multiply-add per product in the operands' own precision (the oracle's chain; the reference computes in
// double, spgemm.cu:728 -- fp32 is SURVEY 8(f)-3)
__device__ __forceinline__ double pem_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float pem_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// host side (each in the unit of its step)
pem_status step_elapsed(pem_ctx *ctx, int e0, int e1, double *dst);
void retire_graph(pem_ctx *ctx, pem_cplan *plan);
pem_status step1_impl(pem_ctx *ctx, pem_cplan *p, bool allow_warm);
pem_status step2_impl(pem_ctx *ctx, pem_cplan *p);
pem_status step3_impl(pem_ctx *ctx, pem_cplan *p);
pem_status ensure_compact(pem_ctx *ctx, const pem_cplan *cp);       // _C_tileColIdx / pair offsets in the reference layout
pem_status ensure_c_rowidx(pem_ctx *ctx, const pem_cplan *p);       // _C_tileRowIdx
pem_status ensure_c_rowptr(pem_ctx *ctx, const pem_cplan *p);       // Ctiles_rowPtr
pem_status ensure_c_rowcolidx(pem_ctx *ctx, const pem_cplan *cp);   // Ctiles_rowColIdx

}  // namespace pem
