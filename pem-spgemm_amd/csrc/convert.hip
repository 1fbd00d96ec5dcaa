// convert.hip -- rows a2-a8: COO/CSR -> 16x16 tiled CSR on the device, plus contexts.
//
// The reference sorts the tile keys, stable-sorts the (I,J,val) zip into CSR, then lets one
// 256-thread block per tile binary-search the CSR (spgemm.cu:832-1062).  Here ONE radix sort
// on the key (tileRow | tileCol | r | c) puts every nonzero directly at its final slot of the
// tiled layout; tile boundaries, masks, intra-tile row pointers and the transposed masks
// then fall out of the sorted keys with wave64 ballots / 16-lane scans.  The arrays produced
// are bit-identical to the reference's (layouts in include/pem_spgemm.h).
#include "pem_internal.h"
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

using namespace pem;

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// a2 decide_which_tile (spgemm.cu:112-135), extended with the intra-tile (r,c) byte so the
// sort order IS the tiled storage order.  transpose swaps the roles of I and J (:788-792).
__global__ void conv_make_keys_kernel(const int *__restrict__ I, const int *__restrict__ J, size_t nnz, int rows, int cols,
                                      int transpose, int bits_tc, uint64_t *__restrict__ keys, uint32_t *__restrict__ perm,
                                      int *__restrict__ flags)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int r = transpose ? J[e] : I[e];
    int c = transpose ? I[e] : J[e];
    if (r < 0 || r >= rows || c < 0 || c >= cols) {
        flags[FLAG_RANGE] = 1;
        r = 0;
        c = 0;
    }
    keys[e] = ((uint64_t)(unsigned)(r >> 4) << (bits_tc + 8)) | ((uint64_t)(unsigned)(c >> 4) << 8) |
              (uint64_t)(((r & 15) << 4) | (c & 15));
    perm[e] = (uint32_t)e;
}

// CSR input: expand row pointers into the same keys (thread per row).
__global__ void conv_csr_keys_kernel(const int *__restrict__ rowptr, const int *__restrict__ colidx, int rows, int cols,
                                     int bits_tc, uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, int *__restrict__ flags)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
        int c = colidx[e];
        if (c < 0 || c >= cols) {
            flags[FLAG_RANGE] = 1;
            c = 0;
        }
        keys[e] = ((uint64_t)(unsigned)(r >> 4) << (bits_tc + 8)) | ((uint64_t)(unsigned)(c >> 4) << 8) |
                  (uint64_t)(((r & 15) << 4) | (c & 15));
        perm[e] = (uint32_t)e;
    }
}

// a3 (spgemm.cu:866-892 sort/unique/reduce_by_key): tile heads of the sorted key stream.
__global__ void conv_heads_kernel(const uint64_t *__restrict__ keys, size_t nnz, int *__restrict__ head, int *__restrict__ flags)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    uint64_t k = keys[e];
    int h = 1;
    if (e > 0) {
        uint64_t p = keys[e - 1];
        h = (k >> 8) != (p >> 8);
        if (k == p) flags[FLAG_DUP] = 1;
    }
    head[e] = h;
}

// a5 payload (spgemm.cu:195, 218-222): values + (r<<4|c) bytes in tile order; tile list +
// perTileNnz offsets (spgemm.cu:873-877).  headx = exclusive scan of the head flags.
template <typename VT>
__global__ void conv_fill_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm, const int *__restrict__ headx,
                                 size_t nnz, const VT *__restrict__ V, int bits_tc, VT *__restrict__ vals,
                                 uint8_t *__restrict__ rowcolidx, long long *__restrict__ tile_keys, int *__restrict__ tile_nnz_ptr)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    uint64_t k = keys[e];
    vals[e] = V[perm[e]];
    rowcolidx[e] = (uint8_t)(k & 0xFF);
    int t = headx[e];
    bool is_head = headx[e + 1] != t;
    if (is_head) {
        uint64_t tc = (k >> 8) & ((1ull << bits_tc) - 1ull);
        uint64_t tr = k >> (8 + bits_tc);
        tile_keys[t] = (long long)((tr << 32) | tc);
        tile_nnz_ptr[t] = (int)e;
    }
    if (e == nnz - 1) tile_nnz_ptr[headx[nnz]] = (int)nnz;   // perTileNnz[T] = nnz
}

// a5 masks + intra-tile row pointers (spgemm.cu:196-209) and a6 transposed masks
// (spgemm.cu:228-258).  16 lanes per tile, lane = tile row; one wave64 covers 4 tiles, so a
// ballot returns the four 16-bit transposed rows at once.
__global__ void __launch_bounds__(256) conv_tile_meta_kernel(const uint8_t *__restrict__ rowcolidx, const int *__restrict__ tile_nnz_ptr,
                                                             long long ntiles, uint16_t *__restrict__ masks, uint8_t *__restrict__ rowptr,
                                                             uint16_t *__restrict__ masks_t, uint32_t *__restrict__ rec,
                                                             uint32_t *__restrict__ occ, uint32_t *__restrict__ rec_t)
{
    long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int r = threadIdx.x & 15;
    const int grp = (threadIdx.x & 63) >> 4;
    const bool live = t < ntiles;
    // The sixteen lanes of a tile read its (row, col) bytes TOGETHER, sixteen per trip, and OR each entry's column bit into its
    // row's word in LDS (word = lane of the row; only this wave touches it).  Every lane walking all of the tile's bytes on its own
    // was a chain of dependent loads per tile, and five vector-ALU instructions per (lane, entry).
    __shared__ unsigned s_rowmask[256];
    s_rowmask[threadIdx.x] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live) {
        const int e0 = tile_nnz_ptr[t], e1 = tile_nnz_ptr[t + 1];
        for (int e = e0 + r; e < e1; e += 16) {
            const unsigned rc = rowcolidx[e];
            atomicOr(&s_rowmask[(threadIdx.x & ~15u) + (rc >> 4)], 1u << (rc & 15u));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned mask = s_rowmask[threadIdx.x];
    // exclusive scan of the row populations across the 16-lane group
    int cnt = __popc(mask), inc = cnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int v = __shfl_up(inc, d, 16);
        if (r >= d) inc += v;
    }
    // transposed masks: the 16 x 16 bit matrix held one row per lane is transposed by four butterfly stages (swap the off-diagonal
    // s x s blocks with lane r ^ s, s = 8, 4, 2, 1; ds_swizzle: no LDS memory, no address arithmetic) -- ~25 vector-ALU
    // instructions where sixteen ballots with per-lane extraction took ~190 of the kernel's 305 (it was ALU-bound: 96 us)
    unsigned bt = mask;
    {
        unsigned y;
        y = (unsigned)__builtin_amdgcn_ds_swizzle((int)bt, (8 << 10) | 0x1F);
        bt = (r & 8) ? ((bt & 0xFF00u) | ((y >> 8) & 0x00FFu)) : ((bt & 0x00FFu) | ((y & 0x00FFu) << 8));
        y = (unsigned)__builtin_amdgcn_ds_swizzle((int)bt, (4 << 10) | 0x1F);
        bt = (r & 4) ? ((bt & 0xF0F0u) | ((y >> 4) & 0x0F0Fu)) : ((bt & 0x0F0Fu) | ((y & 0x0F0Fu) << 4));
        y = (unsigned)__builtin_amdgcn_ds_swizzle((int)bt, (2 << 10) | 0x1F);
        bt = (r & 2) ? ((bt & 0xCCCCu) | ((y >> 2) & 0x3333u)) : ((bt & 0x3333u) | ((y & 0x3333u) << 2));
        y = (unsigned)__builtin_amdgcn_ds_swizzle((int)bt, (1 << 10) | 0x1F);
        bt = (r & 1) ? ((bt & 0xAAAAu) | ((y >> 1) & 0x5555u)) : ((bt & 0x5555u) | ((y & 0x5555u) << 1));
    }
    // occupancy word: which columns hold an entry (OR of the row masks) | which rows are non-empty << 16
    unsigned co = mask;
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) co |= (unsigned)__shfl_xor((int)co, d, 16);
    const unsigned long long nzrows = __ballot(mask != 0);
    const unsigned ro = (unsigned)(nzrows >> (16 * grp)) & 0xFFFFu;
    if (live && r == 0) occ[t] = co | (ro << 16);
    // transposed record of column c = r: rows holding it | number of entries in the columns before it
    int ccnt = __popc(bt), cinc = ccnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int v = __shfl_up(cinc, d, 16);
        if (r >= d) cinc += v;
    }
    if (live) rec_t[16 * t + r] = bt | ((unsigned)(cinc - ccnt) << 16);
    if (live) {
        masks[16 * t + r] = (uint16_t)mask;
        rowptr[16 * t + r] = (uint8_t)(inc - cnt);
        masks_t[16 * t + r] = (uint16_t)bt;
        rec[16 * t + r] = mask | ((unsigned)(inc - cnt) << 16);
    }
}

// column-major copy of every tile's values: entry (rr, c) of tile t goes to slot (entries in columns < c) + (rows < rr
// holding column c).  headx = exclusive scan of the tile-head flags (tile of entry e = headx[e] - !is_head).
template <typename VT>
__global__ void conv_vals_t_kernel(const uint8_t *__restrict__ rowcolidx, const VT *__restrict__ vals, const int *__restrict__ headx,
                                   size_t nnz, const int *__restrict__ tile_nnz_ptr, const uint32_t *__restrict__ rec_t,
                                   VT *__restrict__ vals_t)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int hx = headx[e];
    const int t = (headx[e + 1] != hx) ? hx : hx - 1;
    const unsigned rc = rowcolidx[e];
    const unsigned rr = rc >> 4, c = rc & 15u;
    const unsigned w = rec_t[16 * (size_t)t + c];
    vals_t[tile_nnz_ptr[t] + (int)(w >> 16) + __popc(w & 0xFFFFu & ((1u << rr) - 1u))] = vals[e];
}

// a7 (spgemm.cu:986-1031): tile-level CSR from the sorted tile list -- boundary fill, no
// reduce_by_key / scatter / scan needed.  Also the packed (tile column, occupancy) record step 1 expands products from.
__global__ void conv_tile_csr_kernel(const long long *__restrict__ tile_keys, long long ntiles, int tile_rows,
                                     int *__restrict__ tile_rowptr, int *__restrict__ tile_colidx, const uint32_t *__restrict__ occ,
                                     int2 *__restrict__ colocc)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    long long k = tile_keys[t];
    int tr = (int)(k >> 32);
    tile_colidx[t] = (int)(k & 0xFFFFFFFFll);
    colocc[t] = make_int2((int)(k & 0xFFFFFFFFll), (int)occ[t]);
    int prev = t > 0 ? (int)(tile_keys[t - 1] >> 32) : -1;
    for (int row = prev + 1; row <= tr; ++row) tile_rowptr[row] = (int)t;
    if (t == ntiles - 1)
        for (int row = tr + 1; row <= tile_rows; ++row) tile_rowptr[row] = (int)ntiles;
}

// a7 (spgemm.cu:1033-1040): column-major re-sort keys, payload = CSR tile id
__global__ void conv_csc_keys_kernel(const long long *__restrict__ tile_keys, long long ntiles, int bits_tr,
                                     uint64_t *__restrict__ keys, uint32_t *__restrict__ perm)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    long long k = tile_keys[t];
    keys[t] = ((uint64_t)(k & 0xFFFFFFFFll) << bits_tr) | (uint64_t)(k >> 32);
    perm[t] = (uint32_t)t;
}

// a7 (spgemm.cu:1042-1061): tile-level CSC + _B_tileOffsets
__global__ void conv_tile_csc_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm, long long ntiles,
                                     int bits_tr, int tile_cols, int *__restrict__ tile_colptr, int *__restrict__ tile_rowidx,
                                     int *__restrict__ tile_offsets)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    uint64_t k = keys[t];
    int tc = (int)(k >> bits_tr);
    tile_rowidx[t] = (int)(k & ((1ull << bits_tr) - 1ull));
    tile_offsets[t] = (int)perm[t];
    int prev = t > 0 ? (int)(keys[t - 1] >> bits_tr) : -1;
    for (int col = prev + 1; col <= tc; ++col) tile_colptr[col] = (int)t;
    if (t == ntiles - 1)
        for (int col = tc + 1; col <= tile_cols; ++col) tile_colptr[col] = (int)ntiles;
}

// a8 flop count (spgemm.cu:1068-1079) without the host loop: flop = sum_k colnnz_A(k) * rownnz_B(k).
// 16 lanes per tile row of B (lane = matrix row), resp. per tile column of A (lane = matrix column).
__global__ void __launch_bounds__(256) flop_rownnz_kernel(const int *__restrict__ tile_rowptr, const uint16_t *__restrict__ masks,
                                                          int tile_rows, int *__restrict__ rownnz)
{
    int tr = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int r = threadIdx.x & 15;
    if (tr >= tile_rows) return;
    int cnt = 0;
    for (int t = tile_rowptr[tr]; t < tile_rowptr[tr + 1]; ++t) cnt += __popc((unsigned)masks[16 * (size_t)t + r]);
    rownnz[16 * (size_t)tr + r] = cnt;
}

__global__ void __launch_bounds__(256) flop_colnnz_kernel(const int *__restrict__ tile_colptr, const int *__restrict__ tile_offsets,
                                                          const uint16_t *__restrict__ masks_t, int tile_cols, int *__restrict__ colnnz)
{
    int tc = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int c = threadIdx.x & 15;
    if (tc >= tile_cols) return;
    int cnt = 0;
    for (int p = tile_colptr[tc]; p < tile_colptr[tc + 1]; ++p)
        cnt += __popc((unsigned)masks_t[16 * (size_t)tile_offsets[p] + c]);
    colnnz[16 * (size_t)tc + c] = cnt;
}

__global__ void __launch_bounds__(256) flop_dot_kernel(const int *__restrict__ colnnz_a, const int *__restrict__ rownnz_b, size_t n,
                                                       unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long wsum[4];
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        s += (unsigned long long)colnnz_a[i] * (unsigned long long)rownnz_b[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);   // integer sum: order-independent
}

// ------------------------------------------------------------------------------------------
// contexts
// ------------------------------------------------------------------------------------------
extern "C" pem_status pem_ctx_create_on_stream(int device, void *stream, pem_ctx **out)
{
    if (!out) return PEM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return PEM_E_NODEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d devices)", device, ndev);
        return PEM_E_INVALID;
    }
    PEM_HIP(hipSetDevice(device));
    pem_ctx *ctx = new pem_ctx();
    ctx->device = device;
    ctx->arena = std::make_shared<pem::Arena>(device);
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        PEM_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess) ctx->cu_count = cus;
        else (void)hipGetLastError();
    }
    PEM_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_scalars), sizeof(int64_t) * 64, hipHostMallocDefault));
    {
        static_assert(NUM_FLAGS <= 16, "the flag mirror takes the last eight 64-bit slots of h_scalars");
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->h_scalars, 0) == hipSuccess && dp) {
            ctx->h_flags = reinterpret_cast<volatile int *>(ctx->h_scalars + 56);
            ctx->h_flags_dev = reinterpret_cast<int *>(reinterpret_cast<int64_t *>(dp) + 56);
        } else {
            (void)hipGetLastError();
        }
    }
    PEM_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_scalars), sizeof(int64_t) * 64));
    PEM_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_flags), sizeof(int) * NUM_FLAGS));
    PEM_HIP(hipMemsetAsync(ctx->d_scalars, 0, sizeof(int64_t) * 64, ctx->stream));
    PEM_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(int) * NUM_FLAGS, ctx->stream));
    for (auto &ev : ctx->ev) PEM_HIP(hipEventCreate(&ev));
    for (auto &a : ctx->aux) PEM_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    const char *graph_env = getenv("PEM_GRAPH");   // default for pem_set_graph_replay (tools, CLI)
    ctx->graph_replay = graph_env && !strcmp(graph_env, "1");
    const char *gd_env = getenv("PEM_DEBUG_GRAPH_DESTROY");   // diagnostic, tools/graph_churn.py: see retire_graph
    ctx->dbg_destroy_graphs = gd_env && !strcmp(gd_env, "1");
    PEM_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (auto &e : ctx->ev_join) PEM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx;
    return PEM_OK;
}

extern "C" pem_status pem_ctx_create(int device, pem_ctx **out) { return pem_ctx_create_on_stream(device, nullptr, out); }

extern "C" pem_status pem_ctx_destroy(pem_ctx *ctx)
{
    if (!ctx) return PEM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto ge : ctx->retired_graphs) (void)hipGraphExecDestroy(ge);
    ctx->retired_graphs.clear();
    for (auto &s : ctx->pending) {
        (void)hipEventDestroy(s.e0);
        (void)hipEventDestroy(s.e1);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->ev) (void)hipEventDestroy(e);
    for (auto a : ctx->aux)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (auto e : ctx->ev_join)
        if (e) (void)hipEventDestroy(e);
    (void)hipHostFree(ctx->h_scalars);
    (void)hipFree(ctx->d_scalars);
    (void)hipFree(ctx->d_flags);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PEM_OK;
}

extern "C" pem_status pem_ctx_synchronize(pem_ctx *ctx)
{
    if (!ctx) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    // also where an asynchronous call (pem_c_export_csr_device) reports that one of its device scans gave up
    int hf[NUM_FLAGS];
    PEM_TRY(read_flags(ctx, hf));   // (synchronises the stream)
    return check_internal(hf);
}

extern "C" pem_status pem_ctx_reserve(pem_ctx *ctx, int64_t bytes)
{
    if (!ctx || bytes < 0) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    return bytes ? ctx->arena->reserve((size_t)bytes) : PEM_OK;
}

extern "C" pem_status pem_ctx_trim(pem_ctx *ctx)
{
    if (!ctx) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->arena->trim();
    return PEM_OK;
}

extern "C" pem_status pem_ctx_memory_stats(pem_ctx *ctx, pem_memory_stats *out)
{
    if (!ctx || !out) return PEM_E_INVALID;
    const pem::Arena::Stats s = ctx->arena->stats();
    out->slab_bytes = (int64_t)s.slab_bytes;
    out->in_use_bytes = (int64_t)s.in_use_bytes;
    out->peak_in_use_bytes = (int64_t)s.peak_in_use_bytes;
    out->largest_free_bytes = (int64_t)s.largest_free_bytes;
    out->driver_allocs = s.driver_allocs;
    out->block_allocs = s.block_allocs;
    return PEM_OK;
}

extern "C" pem_status pem_get_timings(pem_ctx *ctx, pem_timings *t)
{
    if (!ctx || !t) return PEM_E_INVALID;
    *t = ctx->timings;
    return PEM_OK;
}

extern "C" pem_status pem_set_kernel_profiling(pem_ctx *ctx, int enabled)
{
    if (!ctx) return PEM_E_INVALID;
    PEM_TRY(resolve_kernel_spans(ctx));
    ctx->profiling = enabled != 0;
    return PEM_OK;
}

extern "C" pem_status pem_reset_kernel_stats(pem_ctx *ctx)
{
    if (!ctx) return PEM_E_INVALID;
    PEM_TRY(resolve_kernel_spans(ctx));
    ctx->stats.clear();
    return PEM_OK;
}

extern "C" pem_status pem_kernel_stats_count(pem_ctx *ctx, int *n)
{
    if (!ctx || !n) return PEM_E_INVALID;
    PEM_TRY(resolve_kernel_spans(ctx));
    *n = (int)ctx->stats.size();
    return PEM_OK;
}

extern "C" pem_status pem_kernel_stats_get(pem_ctx *ctx, int idx, char *name, int name_cap, int64_t *calls, double *total_ms)
{
    if (!ctx || idx < 0 || idx >= (int)ctx->stats.size()) return PEM_E_INVALID;
    const KernelStat &s = ctx->stats[idx];
    if (name && name_cap > 0) {
        strncpy(name, s.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (calls) *calls = s.calls;
    if (total_ms) *total_ms = s.total_ms;
    return PEM_OK;
}

// ------------------------------------------------------------------------------------------
// conversion driver
// ------------------------------------------------------------------------------------------
// sizing phase "tiles": every per-tile array of a tiling (and the small sort of its tile CSC) in one driver allocation
static pem_status reserve_tile_arrays(pem_ctx *ctx, pem_tiled *T)
{
    const size_t nnz = (size_t)T->nnz, nt = (size_t)T->ntiles;
    return arena_phase(ctx->arena,
                       {{&T->masks, 32 * (nt + 1)}, {&T->masks_t, 32 * (nt + 1)}, {&T->rowptr, 16 * (nt + 1)}, {&T->tile_rec, 64 * (nt + 1)},
                        {&T->tile_occ, 4 * (nt + 4)}, {&T->tile_colocc, 8 * (nt + 4)}, {&T->tile_rec_t, 64 * (nt + 1)},
                        {&T->vals_t, (size_t)T->value_bytes * (nnz + 1)}, {&T->tile_rowptr, 4 * ((size_t)T->tile_rows + 4)},
                        {&T->tile_colidx, 4 * (nt + 4)}, {&T->tile_colptr, 4 * ((size_t)T->tile_cols + 4)}, {&T->tile_rowidx, 4 * (nt + 4)},
                        {&T->tile_offsets, 4 * (nt + 4)}, {&ctx->tmp[8], 8 * nt}, {&ctx->tmp[9], 8 * nt}, {&ctx->tmp[10], 4 * nt},
                        {&ctx->tmp[11], 4 * nt}, {&T->tile_keys, 8 * (nt + 1)}, {&T->tile_nnz_ptr, 4 * (nt + 4)},
                        {&T->vals, (size_t)T->value_bytes * (nnz + 1)}, {&T->rowcolidx, nnz + 16}, {&ctx->tmp[0], 4 * (nnz + 4)}});
}

// Everything that follows from the sorted tile payload (tile_keys, tile_nnz_ptr, rowcolidx, vals): masks, intra-tile
// row pointers, transposed masks (a5/a6), the step-3 records, the column-major values and the tile-level CSR/CSC
// indices (a7).  Shared by the conversion and by the cache loader.  headx = exclusive scan of the tile-head flags;
// ctx->ev[6] was recorded by the caller where its payload kernels start.
static pem_status derive_tiled(pem_ctx *ctx, pem_tiled *T, const int *headx, int bits_tr, int bits_tc)
{
    const size_t nnz = (size_t)T->nnz, nt = (size_t)T->ntiles;
    const int64_t ntiles = T->ntiles;
    hipStream_t st = ctx->stream;
    PEM_TRY(T->masks.reserve(sizeof(uint16_t) * 16 * (nt + 1)));
    PEM_TRY(T->masks_t.reserve(sizeof(uint16_t) * 16 * (nt + 1)));
    PEM_TRY(T->rowptr.reserve(16 * (nt + 1)));
    PEM_TRY(T->tile_rec.reserve(sizeof(uint32_t) * 16 * (nt + 1)));
    PEM_TRY(T->tile_occ.reserve(sizeof(uint32_t) * (nt + 4)));
    PEM_TRY(T->tile_colocc.reserve(sizeof(int2) * (nt + 4)));
    PEM_TRY(T->tile_rec_t.reserve(sizeof(uint32_t) * 16 * (nt + 1)));
    PEM_TRY(T->vals_t.reserve((size_t)T->value_bytes * (nnz + 1)));
    PEM_TRY(T->tile_rowptr.reserve(sizeof(int) * ((size_t)T->tile_rows + 4)));
    PEM_TRY(T->tile_colidx.reserve(sizeof(int) * (nt + 4)));
    PEM_TRY(T->tile_colptr.reserve(sizeof(int) * ((size_t)T->tile_cols + 4)));
    PEM_TRY(T->tile_rowidx.reserve(sizeof(int) * (nt + 4)));
    PEM_TRY(T->tile_offsets.reserve(sizeof(int) * (nt + 4)));
    PEM_HIP(hipMemsetAsync(T->tile_rowptr.p, 0, sizeof(int) * ((size_t)T->tile_rows + 1), st));
    PEM_HIP(hipMemsetAsync(T->tile_colptr.p, 0, sizeof(int) * ((size_t)T->tile_cols + 1), st));
    if (nnz) {
        PEM_LAUNCH(ctx, conv_tile_meta_kernel, grid_for(nt * 16, 256), 256, T->rowcolidx.as<uint8_t>(), T->tile_nnz_ptr.as<int>(),
                   (long long)ntiles, T->masks.as<uint16_t>(), T->rowptr.as<uint8_t>(), T->masks_t.as<uint16_t>(), T->tile_rec.as<uint32_t>(),
                   T->tile_occ.as<uint32_t>(), T->tile_rec_t.as<uint32_t>());
        if (T->value_bytes == 4)
            PEM_LAUNCH(ctx, conv_vals_t_kernel<float>, grid_for(nnz, 256), 256, T->rowcolidx.as<uint8_t>(), T->vals.as<float>(), headx, nnz,
                       T->tile_nnz_ptr.as<int>(), T->tile_rec_t.as<uint32_t>(), T->vals_t.as<float>());
        else
            PEM_LAUNCH(ctx, conv_vals_t_kernel<double>, grid_for(nnz, 256), 256, T->rowcolidx.as<uint8_t>(), T->vals.as<double>(), headx, nnz,
                       T->tile_nnz_ptr.as<int>(), T->tile_rec_t.as<uint32_t>(), T->vals_t.as<double>());
    }
    PEM_HIP(hipEventRecord(ctx->ev[7], st));
    if (nt) {
        PEM_LAUNCH(ctx, conv_tile_csr_kernel, grid_for(nt, 256), 256, T->tile_keys.as<long long>(), (long long)ntiles, T->tile_rows,
                   T->tile_rowptr.as<int>(), T->tile_colidx.as<int>(), T->tile_occ.as<uint32_t>(), T->tile_colocc.as<int2>());
        // column-major order of the tiles: second (small) radix sort, payload = CSR tile id
        DevBuf &k0 = ctx->tmp[8], &k1 = ctx->tmp[9], &v0 = ctx->tmp[10], &v1 = ctx->tmp[11];   // context-owned, grow-only
        PEM_TRY(k0.reserve(sizeof(uint64_t) * nt));
        PEM_TRY(k1.reserve(sizeof(uint64_t) * nt));
        PEM_TRY(v0.reserve(sizeof(uint32_t) * nt));
        PEM_TRY(v1.reserve(sizeof(uint32_t) * nt));
        PEM_LAUNCH(ctx, conv_csc_keys_kernel, grid_for(nt, 256), 256, T->tile_keys.as<long long>(), (long long)ntiles, bits_tr,
                   k0.as<uint64_t>(), v0.as<uint32_t>());
        uint64_t *ck = nullptr;
        uint32_t *cp = nullptr;
        // (the tiles arrive sorted by (tile row, tile column) and the passes are stable: sorting on the tile-column digits alone
        // leaves the tile rows ascending inside every column -- two passes where the whole key took four)
        PEM_TRY(radix_sort_u64_u32(ctx, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), nt, bits_tr + bits_tc,
                                   &ck, &cp, bits_tr));
        PEM_LAUNCH(ctx, conv_tile_csc_kernel, grid_for(nt, 256), 256, ck, cp, (long long)ntiles, bits_tr, T->tile_cols,
                   T->tile_colptr.as<int>(), T->tile_rowidx.as<int>(), T->tile_offsets.as<int>());
    }
    T->h_tile_rowptr.assign((size_t)T->tile_rows + 1, 0);
    PEM_HIP(hipMemcpyAsync(T->h_tile_rowptr.data(), T->tile_rowptr.p, sizeof(int) * ((size_t)T->tile_rows + 1), hipMemcpyDeviceToHost, st));
    PEM_HIP(hipStreamSynchronize(st));
    PEM_TRY(launch_status(ctx));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess) T->conv_tile_kernel_ms = ms;
    return PEM_OK;
}

// keys/perm already filled in k0/v0 (nnz entries); V = device values in input order.
static pem_status build_tiled(pem_ctx *ctx, pem_tiled *T, DevBuf &k0, DevBuf &k1, DevBuf &v0, DevBuf &v1, const void *dV,
                              int bits_tr, int bits_tc)
{
    const size_t nnz = (size_t)T->nnz;
    hipStream_t st = ctx->stream;
    uint64_t *keys = k0.as<uint64_t>();
    uint32_t *perm = v0.as<uint32_t>();
    PEM_TRY(radix_sort_u64_u32(ctx, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), nnz,
                               8 + bits_tc + bits_tr, &keys, &perm));
    // tile heads -> exclusive scan -> T
    DevBuf &head = ctx->tmp[0];
    PEM_TRY(head.reserve(sizeof(int) * (nnz + 4)));
    if (nnz) PEM_LAUNCH(ctx, conv_heads_kernel, grid_for(nnz, 256), 256, keys, nnz, head.as<int>(), ctx->d_flags);
    PEM_TRY(exclusive_scan_i32(ctx, head.as<int>(), head.as<int>(), nnz, ctx->d_scalars));
    int64_t ntiles = 0;
    PEM_TRY(read_scalars(ctx, ctx->d_scalars, 1, &ntiles));
    int hf[NUM_FLAGS];
    PEM_TRY(read_flags(ctx, hf));
    PEM_TRY(check_internal(hf));   // (the scan gave up: ntiles and every offset behind it would be wrong)
    if (hf[FLAG_RANGE]) {
        set_error("index out of range for a %d x %d matrix", T->rows, T->cols);
        return PEM_E_INVALID;
    }
    if (hf[FLAG_DUP]) {
        set_error("duplicate (row, col) entries in the input");
        return PEM_E_DUPLICATE;
    }
    T->ntiles = ntiles;
    const size_t nt = (size_t)ntiles;
    PEM_TRY(reserve_tile_arrays(ctx, T));
    PEM_TRY(T->tile_keys.reserve(sizeof(long long) * (nt + 1)));
    PEM_TRY(T->tile_nnz_ptr.reserve(sizeof(int) * (nt + 4)));
    PEM_TRY(T->vals.reserve((size_t)T->value_bytes * (nnz + 1)));
    PEM_TRY(T->rowcolidx.reserve(nnz + 16));
    PEM_HIP(hipMemsetAsync(T->tile_nnz_ptr.p, 0, sizeof(int) * (nt + 1), st));
    PEM_HIP(hipEventRecord(ctx->ev[6], st));
    if (nnz && T->value_bytes == 4)
        PEM_LAUNCH(ctx, conv_fill_kernel<float>, grid_for(nnz, 256), 256, keys, perm, head.as<int>(), nnz, static_cast<const float *>(dV), bits_tc,
                   T->vals.as<float>(), T->rowcolidx.as<uint8_t>(), T->tile_keys.as<long long>(), T->tile_nnz_ptr.as<int>());
    else if (nnz)
        PEM_LAUNCH(ctx, conv_fill_kernel<double>, grid_for(nnz, 256), 256, keys, perm, head.as<int>(), nnz, static_cast<const double *>(dV), bits_tc,
                   T->vals.as<double>(), T->rowcolidx.as<uint8_t>(), T->tile_keys.as<long long>(), T->tile_nnz_ptr.as<int>());
    return derive_tiled(ctx, T, head.as<int>(), bits_tr, bits_tc);
}

static pem_status tiled_from_device(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int *dI, const int *dJ, const int *d_rowptr,
                                    const void *dV, int value_bytes, int transpose, pem_tiled **out)
{
    if (!ctx || !out || rows <= 0 || cols <= 0 || nnz < 0) {
        set_error("pem_tiled_from_*: bad arguments (rows=%d cols=%d nnz=%lld)", rows, cols, (long long)nnz);
        return PEM_E_INVALID;
    }
    if (nnz > 0x7FFFFFFFll) {
        set_error("nnz=%lld exceeds the reference's int32 index range", (long long)nnz);
        return PEM_E_OVERFLOW;
    }
    *out = nullptr;
    PEM_ENTER(ctx);
    auto t0 = std::chrono::high_resolution_clock::now();
    pem_tiled *T = new pem_tiled();
    T->value_bytes = value_bytes;
    T->rows = transpose ? cols : rows;
    T->cols = transpose ? rows : cols;
    T->nnz = nnz;
    T->tile_rows = (T->rows + 15) / 16;
    T->tile_cols = (T->cols + 15) / 16;
    const int bits_tr = bits_for((uint64_t)T->tile_rows), bits_tc = bits_for((uint64_t)T->tile_cols);
    pem_status s = PEM_OK;
    {
        DevBuf k0, k1, v0, v1;
        const size_t n = (size_t)nnz;
        s = zero_flags(ctx);
        // sizing phase "entries": sort buffers, head flags and everything nnz-sized of the tiling in one driver allocation
        if (s == PEM_OK)
            s = arena_phase(ctx->arena, {{&k0, sizeof(uint64_t) * (n + 1)}, {&k1, sizeof(uint64_t) * (n + 1)}, {&v0, sizeof(uint32_t) * (n + 1)},
                                         {&v1, sizeof(uint32_t) * (n + 1)}, {&ctx->tmp[0], sizeof(int) * (n + 4)},
                                         {&T->vals, (size_t)value_bytes * (n + 1)}, {&T->vals_t, (size_t)value_bytes * (n + 1)},
                                         {&T->rowcolidx, n + 16}, {&ctx->sort_hist, sizeof(int) * (256 * (n / 4096 + 1) + 4)}});
        if (s == PEM_OK) s = k0.reserve(sizeof(uint64_t) * (n + 1));
        if (s == PEM_OK) s = k1.reserve(sizeof(uint64_t) * (n + 1));
        if (s == PEM_OK) s = v0.reserve(sizeof(uint32_t) * (n + 1));
        if (s == PEM_OK) s = v1.reserve(sizeof(uint32_t) * (n + 1));
        if (s == PEM_OK && n) {
            if (d_rowptr)
                PEM_LAUNCH(ctx, conv_csr_keys_kernel, grid_for((size_t)rows, 256), 256, d_rowptr, dJ, rows, cols, bits_tc, k0.as<uint64_t>(),
                           v0.as<uint32_t>(), ctx->d_flags);
            else
                PEM_LAUNCH(ctx, conv_make_keys_kernel, grid_for(n, 256), 256, dI, dJ, n, T->rows, T->cols, transpose, bits_tc,
                           k0.as<uint64_t>(), v0.as<uint32_t>(), ctx->d_flags);
        }
        if (s == PEM_OK) s = build_tiled(ctx, T, k0, k1, v0, v1, dV, bits_tr, bits_tc);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (s != PEM_OK) {
        delete T;
        return s;
    }
    T->conv_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    *out = T;
    return PEM_OK;
}

static pem_status from_coo_host(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *I, const int32_t *J, const void *V,
                                int value_bytes, int transpose, pem_tiled **out)
{
    if (!ctx || nnz < 0 || (nnz > 0 && (!I || !J || !V))) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    DevBuf dI, dJ, dV;
    const size_t n = (size_t)nnz;
    PEM_TRY(dI.reserve(sizeof(int) * (n + 1)));
    PEM_TRY(dJ.reserve(sizeof(int) * (n + 1)));
    PEM_TRY(dV.reserve((size_t)value_bytes * (n + 1)));
    if (n) {   // H2D of the COO triplets (spgemm.cu:832-838)
        PEM_HIP(hipMemcpyAsync(dI.p, I, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        PEM_HIP(hipMemcpyAsync(dJ.p, J, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        PEM_HIP(hipMemcpyAsync(dV.p, V, (size_t)value_bytes * n, hipMemcpyHostToDevice, ctx->stream));
    }
    pem_status s = tiled_from_device(ctx, rows, cols, nnz, dI.as<int>(), dJ.as<int>(), nullptr, dV.p, value_bytes, transpose, out);
    (void)hipStreamSynchronize(ctx->stream);
    return s;
}

static pem_status from_csr_host(pem_ctx *ctx, int rows, int cols, const int32_t *rowptr, const int32_t *colidx, const void *V, int value_bytes,
                                pem_tiled **out)
{
    if (!ctx || !rowptr || rows <= 0) return PEM_E_INVALID;
    for (int r = 0; r < rows; ++r)
        if (rowptr[r + 1] < rowptr[r] || rowptr[0] != 0) {
            set_error("pem_tiled_from_csr: rowptr is not a non-decreasing sequence starting at 0");
            return PEM_E_INVALID;
        }
    const int64_t nnz = rowptr[rows];
    if (nnz > 0 && (!colidx || !V)) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    DevBuf dR, dJ, dV;
    const size_t n = (size_t)nnz;
    PEM_TRY(dR.reserve(sizeof(int) * ((size_t)rows + 1)));
    PEM_TRY(dJ.reserve(sizeof(int) * (n + 1)));
    PEM_TRY(dV.reserve((size_t)value_bytes * (n + 1)));
    PEM_HIP(hipMemcpyAsync(dR.p, rowptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        PEM_HIP(hipMemcpyAsync(dJ.p, colidx, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        PEM_HIP(hipMemcpyAsync(dV.p, V, (size_t)value_bytes * n, hipMemcpyHostToDevice, ctx->stream));
    }
    pem_status s = tiled_from_device(ctx, rows, cols, nnz, nullptr, dJ.as<int>(), dR.as<int>(), dV.p, value_bytes, 0, out);
    (void)hipStreamSynchronize(ctx->stream);
    return s;
}

extern "C" pem_status pem_tiled_from_coo_device(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *dI, const int32_t *dJ,
                                                const double *dV, int transpose, pem_tiled **out)
{
    if (nnz > 0 && (!dI || !dJ || !dV)) return PEM_E_INVALID;
    return tiled_from_device(ctx, rows, cols, nnz, dI, dJ, nullptr, dV, 8, transpose, out);
}

extern "C" pem_status pem_tiled_from_coo_device_f32(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *dI, const int32_t *dJ,
                                                    const float *dV, int transpose, pem_tiled **out)
{
    if (nnz > 0 && (!dI || !dJ || !dV)) return PEM_E_INVALID;
    return tiled_from_device(ctx, rows, cols, nnz, dI, dJ, nullptr, dV, 4, transpose, out);
}

extern "C" pem_status pem_tiled_from_coo(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *I, const int32_t *J,
                                         const double *V, int transpose, pem_tiled **out)
{
    return from_coo_host(ctx, rows, cols, nnz, I, J, V, 8, transpose, out);
}

extern "C" pem_status pem_tiled_from_coo_f32(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *I, const int32_t *J,
                                             const float *V, int transpose, pem_tiled **out)
{
    return from_coo_host(ctx, rows, cols, nnz, I, J, V, 4, transpose, out);
}

extern "C" pem_status pem_tiled_from_csr(pem_ctx *ctx, int rows, int cols, const int32_t *rowptr, const int32_t *colidx, const double *V,
                                         pem_tiled **out)
{
    return from_csr_host(ctx, rows, cols, rowptr, colidx, V, 8, out);
}

extern "C" pem_status pem_tiled_from_csr_f32(pem_ctx *ctx, int rows, int cols, const int32_t *rowptr, const int32_t *colidx, const float *V,
                                             pem_tiled **out)
{
    return from_csr_host(ctx, rows, cols, rowptr, colidx, V, 4, out);
}

// ------------------------------------------------------------------------------------------
// on-disk cache of the tiled format (include/pem_spgemm.h documents the contract)
// ------------------------------------------------------------------------------------------
namespace {
constexpr char CACHE_MAGIC[8] = {'P', 'E', 'M', 'T', 'I', 'L', 'E', '1'};
struct CacheHeader {             // 128 bytes, little-endian, followed by the four arrays, each padded to 64 bytes
    char magic[8];
    uint32_t version;            // 1
    uint32_t tile_size;          // 16
    uint32_t value_bytes;        // 8 (fp64)
    uint32_t header_bytes;       // 128
    int32_t rows, cols;
    int64_t nnz, ntiles;
    pem_cache_key key;
    uint64_t payload_bytes;      // everything after the header
    uint64_t payload_hash;       // hash64 of the payload
    uint64_t header_hash;        // hash64 of the header with this field zero
    uint8_t pad[128 - 96];
};
static_assert(sizeof(CacheHeader) == 128, "cache header layout");

inline size_t pad64(size_t n) { return (n + 63) & ~size_t(63); }

// 64-bit multiply-rotate hash over 8-byte words (the tail is zero-padded), four interleaved lanes so the multiplies
// pipeline (word i feeds lane i & 3); a corruption check, not a MAC.  Restated in tests/cachefmt.py.
inline uint64_t hash_step(uint64_t h, uint64_t w)
{
    h = (h ^ w) * 0xD6E8FEB86659FD93ull;
    return (h << 29) | (h >> 35);
}
uint64_t hash64(const void *data, size_t bytes, uint64_t seed)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h[4];
    for (uint64_t k = 0; k < 4; ++k) h[k] = seed ^ (bytes * 0x9E3779B97F4A7C15ull) ^ (k * 0xA0761D6478BD642Full);
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        h[0] = hash_step(h[0], w[0]);
        h[1] = hash_step(h[1], w[1]);
        h[2] = hash_step(h[2], w[2]);
        h[3] = hash_step(h[3], w[3]);
    }
    for (int k = 0; i < bytes; i += 8, ++k) {
        uint64_t w = 0;
        memcpy(&w, p + i, bytes - i < 8 ? bytes - i : 8);
        h[k] = hash_step(h[k], w);
    }
    uint64_t r = h[0];
    for (int k = 1; k < 4; ++k) r = hash_step(r, h[k]);
    r ^= r >> 32;
    r *= 0xD6E8FEB86659FD93ull;
    r ^= r >> 29;
    return r;
}

struct CacheLayout {
    size_t off_keys, off_ptr, off_rc, off_vals, total;
};
CacheLayout cache_layout(int64_t nnz, int64_t ntiles, int value_bytes)
{
    CacheLayout L;
    L.off_keys = 0;
    L.off_ptr = L.off_keys + pad64(sizeof(long long) * (size_t)ntiles);
    L.off_rc = L.off_ptr + pad64(sizeof(int) * ((size_t)ntiles + 1));
    L.off_vals = L.off_rc + pad64((size_t)nnz);
    L.total = L.off_vals + pad64((size_t)value_bytes * (size_t)nnz);
    return L;
}
}   // namespace

// One thread per tile: is the uploaded payload a valid tiled matrix?  Any violation raises FLAG_RANGE.  Offsets are
// range-checked before they are used as indices, so a hostile file cannot steer a load out of bounds.
__global__ void cache_check_kernel(const long long *__restrict__ tile_keys, const int *__restrict__ tile_nnz_ptr,
                                   const uint8_t *__restrict__ rowcolidx, long long ntiles, long long nnz, int rows, int cols,
                                   int tile_rows, int tile_cols, int *__restrict__ flags)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    bool ok = true;
    const long long k = tile_keys[t];
    const long long tr = k >> 32, tc = k & 0xFFFFFFFFll;
    ok = ok && k >= 0 && tr < tile_rows && tc < tile_cols;
    if (t > 0) ok = ok && tile_keys[t - 1] < k;                    // sorted, distinct (spgemm.cu:869-871)
    const long long e0 = tile_nnz_ptr[t], e1 = tile_nnz_ptr[t + 1];
    if (t == 0) ok = ok && e0 == 0;
    if (t == ntiles - 1) ok = ok && e1 == nnz;
    ok = ok && e0 >= 0 && e1 <= nnz && e1 > e0 && e1 - e0 <= 256;   // a listed tile holds 1..256 entries
    if (ok) {
        int prev = -1;
        for (long long e = e0; e < e1; ++e) {
            const int rc = rowcolidx[e];
            ok = ok && rc > prev;                                   // row-major, no duplicates (spgemm.cu:195-222)
            prev = rc;
            ok = ok && tr * 16 + (rc >> 4) < rows && tc * 16 + (rc & 15) < cols;
        }
    }
    if (!ok) flags[FLAG_RANGE] = 1;
}

__global__ void cache_heads_kernel(const int *__restrict__ tile_nnz_ptr, long long ntiles, int *__restrict__ head)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) head[tile_nnz_ptr[t]] = 1;
}

extern "C" pem_status pem_tiled_save(pem_ctx *ctx, const pem_tiled *T, const char *path, const pem_cache_key *key)
{
    if (!ctx || !T || !path || !*path) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    const CacheLayout L = cache_layout(T->nnz, T->ntiles, T->value_bytes);
    std::vector<unsigned char> buf(sizeof(CacheHeader) + L.total, 0);
    unsigned char *payload = buf.data() + sizeof(CacheHeader);
    const size_t nt = (size_t)T->ntiles, nnz = (size_t)T->nnz;
    hipStream_t st = ctx->stream;
    if (nt) {
        PEM_HIP(hipMemcpyAsync(payload + L.off_keys, T->tile_keys.p, sizeof(long long) * nt, hipMemcpyDeviceToHost, st));
        PEM_HIP(hipMemcpyAsync(payload + L.off_ptr, T->tile_nnz_ptr.p, sizeof(int) * (nt + 1), hipMemcpyDeviceToHost, st));
    }
    if (nnz) {
        PEM_HIP(hipMemcpyAsync(payload + L.off_rc, T->rowcolidx.p, nnz, hipMemcpyDeviceToHost, st));
        PEM_HIP(hipMemcpyAsync(payload + L.off_vals, T->vals.p, (size_t)T->value_bytes * nnz, hipMemcpyDeviceToHost, st));
    }
    PEM_HIP(hipStreamSynchronize(st));
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, CACHE_MAGIC, 8);
    h.version = 1;
    h.tile_size = 16;
    h.value_bytes = (uint32_t)T->value_bytes;
    h.header_bytes = sizeof(CacheHeader);
    h.rows = T->rows;
    h.cols = T->cols;
    h.nnz = T->nnz;
    h.ntiles = T->ntiles;
    if (key) h.key = *key;
    h.payload_bytes = L.total;
    h.payload_hash = hash64(payload, L.total, 0x70656D74696C6531ull);
    h.header_hash = 0;
    h.header_hash = hash64(&h, sizeof h, 0x6865616465723031ull);
    memcpy(buf.data(), &h, sizeof h);
    // write next to the target, then rename: a reader never sees a half-written cache
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) {
        set_error("pem_tiled_save: cannot create %s: %s", tmp.c_str(), strerror(errno));
        return PEM_E_IO;
    }
    const bool wrote = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    const bool closed = fclose(f) == 0;
    if (!wrote || !closed || rename(tmp.c_str(), path) != 0) {
        set_error("pem_tiled_save: writing %s failed: %s", path, strerror(errno));
        (void)remove(tmp.c_str());
        return PEM_E_IO;
    }
    return PEM_OK;
}

extern "C" pem_status pem_tiled_load(pem_ctx *ctx, const char *path, const pem_cache_key *expect, pem_tiled **out)
{
    if (!ctx || !path || !out) return PEM_E_INVALID;
    *out = nullptr;
    auto t0 = std::chrono::high_resolution_clock::now();
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("pem_tiled_load: cannot open %s: %s", path, strerror(errno));
        return PEM_E_IO;
    }
    CacheHeader h;
    auto fail = [&](const char *why) {
        set_error("pem_tiled_load: %s: %s", path, why);
        if (f) fclose(f);
        f = nullptr;
        return PEM_E_IO;
    };
    if (fread(&h, 1, sizeof h, f) != sizeof h) return fail("shorter than a cache header");
    if (memcmp(h.magic, CACHE_MAGIC, 8) != 0) return fail("not a tiled-format cache file");
    {
        CacheHeader z = h;
        z.header_hash = 0;
        if (hash64(&z, sizeof z, 0x6865616465723031ull) != h.header_hash) return fail("header checksum mismatch");
    }
    if (h.version != 1 || h.tile_size != 16 || (h.value_bytes != 8 && h.value_bytes != 4) || h.header_bytes != sizeof(CacheHeader))
        return fail("unsupported cache version, tile size or value type");
    if (h.rows <= 0 || h.cols <= 0 || h.nnz < 0 || h.nnz > 0x7FFFFFFFll || h.ntiles < 0 || h.ntiles > h.nnz ||
        (h.nnz > 0 && h.ntiles == 0))
        return fail("impossible dimensions in the header");
    const CacheLayout L = cache_layout(h.nnz, h.ntiles, (int)h.value_bytes);
    if (h.payload_bytes != L.total) return fail("payload size does not match the dimensions");
    if (expect && (expect->source_size != h.key.source_size || expect->source_mtime_ns != h.key.source_mtime_ns ||
                   expect->transpose != h.key.transpose)) {
        set_error("pem_tiled_load: %s was made from a different source (size %llu mtime %lld transpose %u)", path,
                  (unsigned long long)h.key.source_size, (long long)h.key.source_mtime_ns, h.key.transpose);
        fclose(f);
        return PEM_E_STALE;
    }
    const bool trace = getenv("PEM_TRACE") != nullptr;
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[pem_tiled_load] %-18s %8.2f ms\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
    };
    lap("header");
    std::vector<unsigned char> payload(L.total + 8);
    lap("buffer");
    if (fread(payload.data(), 1, L.total, f) != L.total) return fail("truncated payload");
    lap("read");
    if (fgetc(f) != EOF) return fail("trailing bytes after the payload");
    fclose(f);
    f = nullptr;
    if (hash64(payload.data(), L.total, 0x70656D74696C6531ull) != h.payload_hash) return fail("payload checksum mismatch");
    lap("checksum");

    PEM_ENTER(ctx);
    hipStream_t st = ctx->stream;
    pem_tiled *T = new pem_tiled();
    T->value_bytes = (int)h.value_bytes;
    T->rows = h.rows;
    T->cols = h.cols;
    T->nnz = h.nnz;
    T->ntiles = h.ntiles;
    T->tile_rows = (T->rows + 15) / 16;
    T->tile_cols = (T->cols + 15) / 16;
    const int bits_tr = bits_for((uint64_t)T->tile_rows), bits_tc = bits_for((uint64_t)T->tile_cols);
    const size_t nt = (size_t)h.ntiles, nnz = (size_t)h.nnz;
    auto body = [&]() -> pem_status {
        PEM_TRY(zero_flags(ctx));
        PEM_TRY(reserve_tile_arrays(ctx, T));
        PEM_TRY(T->tile_keys.reserve(sizeof(long long) * (nt + 1)));
        PEM_TRY(T->tile_nnz_ptr.reserve(sizeof(int) * (nt + 4)));
        PEM_TRY(T->vals.reserve((size_t)T->value_bytes * (nnz + 1)));
        PEM_TRY(T->rowcolidx.reserve(nnz + 16));
        DevBuf &head = ctx->tmp[0];
        PEM_TRY(head.reserve(sizeof(int) * (nnz + 4)));
        PEM_HIP(hipMemsetAsync(T->tile_nnz_ptr.p, 0, sizeof(int) * (nt + 1), st));
        if (nt) {
            PEM_HIP(hipMemcpyAsync(T->tile_keys.p, payload.data() + L.off_keys, sizeof(long long) * nt, hipMemcpyHostToDevice, st));
            PEM_HIP(hipMemcpyAsync(T->tile_nnz_ptr.p, payload.data() + L.off_ptr, sizeof(int) * (nt + 1), hipMemcpyHostToDevice, st));
        }
        if (nnz) {
            PEM_HIP(hipMemcpyAsync(T->rowcolidx.p, payload.data() + L.off_rc, nnz, hipMemcpyHostToDevice, st));
            PEM_HIP(hipMemcpyAsync(T->vals.p, payload.data() + L.off_vals, (size_t)T->value_bytes * nnz, hipMemcpyHostToDevice, st));
        }
        PEM_HIP(hipEventRecord(ctx->ev[6], st));
        lap("alloc + upload");
        if (nt)
            PEM_LAUNCH(ctx, cache_check_kernel, grid_for(nt, 256), 256, T->tile_keys.as<long long>(), T->tile_nnz_ptr.as<int>(),
                       T->rowcolidx.as<uint8_t>(), (long long)nt, (long long)nnz, T->rows, T->cols, T->tile_rows, T->tile_cols, ctx->d_flags);
        int hf[NUM_FLAGS];
        PEM_TRY(read_flags(ctx, hf));   // before anything indexes through the file's offsets
        if (hf[FLAG_RANGE]) {
            set_error("pem_tiled_load: %s: the payload is not a valid tiled matrix (checksum intact: written by a faulty producer)", path);
            return PEM_E_IO;
        }
        PEM_HIP(hipMemsetAsync(head.p, 0, sizeof(int) * (nnz + 1), st));
        if (nt) PEM_LAUNCH(ctx, cache_heads_kernel, grid_for(nt, 256), 256, T->tile_nnz_ptr.as<int>(), (long long)nt, head.as<int>());
        PEM_TRY(exclusive_scan_i32(ctx, head.as<int>(), head.as<int>(), nnz, nullptr));
        PEM_TRY(read_flags(ctx, hf));
        PEM_TRY(check_internal(hf));
        lap("check");
        return derive_tiled(ctx, T, head.as<int>(), bits_tr, bits_tc);
    };
    pem_status s = body();
    (void)hipStreamSynchronize(st);
    lap("derive");
    if (s != PEM_OK) {
        delete T;
        return s;
    }
    T->conv_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    *out = T;
    return PEM_OK;
}

extern "C" pem_status pem_tiled_destroy(pem_ctx *ctx, pem_tiled *t)
{
    // the tiling's blocks go back to the arena and may be handed out again at once: whatever still reads them must be done
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    } else {
        (void)hipDeviceSynchronize();   // (no context given: the caller's current device)
    }
    delete t;
    return PEM_OK;
}

extern "C" pem_status pem_tiled_get_info(const pem_tiled *t, pem_tiled_info *info)
{
    if (!t || !info) return PEM_E_INVALID;
    info->rows = t->rows;
    info->cols = t->cols;
    info->nnz = t->nnz;
    info->tile_rows = t->tile_rows;
    info->tile_cols = t->tile_cols;
    info->ntiles = t->ntiles;
    info->conv_ms = t->conv_ms;
    info->conv_tile_kernel_ms = t->conv_tile_kernel_ms;
    info->value_bytes = t->value_bytes;
    info->reserved = 0;
    return PEM_OK;
}

extern "C" pem_status pem_tiled_get_array(pem_ctx *ctx, const pem_tiled *t, pem_tiled_array which, void *host_dst, int64_t bytes)
{
    if (!ctx || !t || (!host_dst && bytes > 0)) return PEM_E_INVALID;
    const size_t T = (size_t)t->ntiles, nnz = (size_t)t->nnz;
    const void *src = nullptr;
    size_t want = 0;
    switch (which) {
    case PEM_T_TILE_KEYS: src = t->tile_keys.p; want = 8 * T; break;
    case PEM_T_TILE_NNZ_PTR: src = t->tile_nnz_ptr.p; want = 4 * (T + 1); break;
    case PEM_T_MASKS: src = t->masks.p; want = 32 * T; break;
    case PEM_T_ROWPTR: src = t->rowptr.p; want = 16 * T; break;
    case PEM_T_ROWCOLIDX: src = t->rowcolidx.p; want = nnz; break;
    case PEM_T_VALS: src = t->vals.p; want = (size_t)t->value_bytes * nnz; break;   // native type: double or float
    case PEM_T_MASKS_T: src = t->masks_t.p; want = 32 * T; break;
    case PEM_T_TILE_ROWPTR: src = t->tile_rowptr.p; want = 4 * ((size_t)t->tile_rows + 1); break;
    case PEM_T_TILE_COLIDX: src = t->tile_colidx.p; want = 4 * T; break;
    case PEM_T_TILE_COLPTR: src = t->tile_colptr.p; want = 4 * ((size_t)t->tile_cols + 1); break;
    case PEM_T_TILE_ROWIDX: src = t->tile_rowidx.p; want = 4 * T; break;
    case PEM_T_TILE_OFFSETS: src = t->tile_offsets.p; want = 4 * T; break;
    default: set_error("unknown pem_tiled_array %d", (int)which); return PEM_E_INVALID;
    }
    if ((size_t)bytes != want) {
        set_error("pem_tiled_get_array(%d): caller passed %lld bytes, array has %zu", (int)which, (long long)bytes, want);
        return PEM_E_INVALID;
    }
    if (want == 0) return PEM_OK;
    PEM_ENTER(ctx);
    PEM_HIP(hipMemcpyAsync(host_dst, src, want, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return PEM_OK;
}

extern "C" pem_status pem_flop_count(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, uint64_t *flop)
{
    if (!ctx || !A || !B || !flop) return PEM_E_INVALID;
    if (A->cols != B->rows) {
        set_error("pem_flop_count: A is %d x %d, B is %d x %d", A->rows, A->cols, B->rows, B->cols);
        return PEM_E_INVALID;
    }
    PEM_ENTER(ctx);
    const size_t n = 16 * (size_t)A->tile_cols;   // == 16 * B->tile_rows
    DevBuf &ca = ctx->tmp[0], &rb = ctx->tmp[1];
    PEM_TRY(ca.reserve(sizeof(int) * n + 16));
    PEM_TRY(rb.reserve(sizeof(int) * n + 16));
    PEM_HIP(hipMemsetAsync(ctx->d_scalars, 0, sizeof(int64_t), ctx->stream));
    PEM_LAUNCH(ctx, flop_colnnz_kernel, grid_for(n, 256), 256, A->tile_colptr.as<int>(), A->tile_offsets.as<int>(), A->masks_t.as<uint16_t>(),
               A->tile_cols, ca.as<int>());
    PEM_LAUNCH(ctx, flop_rownnz_kernel, grid_for(n, 256), 256, B->tile_rowptr.as<int>(), B->masks.as<uint16_t>(), B->tile_rows, rb.as<int>());
    PEM_LAUNCH(ctx, flop_dot_kernel, 256, 256, ca.as<int>(), rb.as<int>(), n, reinterpret_cast<unsigned long long *>(ctx->d_scalars));
    int64_t v = 0;
    PEM_TRY(read_scalars(ctx, ctx->d_scalars, 1, &v));
    *flop = (uint64_t)v;
    return PEM_OK;
}
