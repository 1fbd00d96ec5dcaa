// export.hip -- row a14: tiled C -> CSR / COO, and the tile-row weights of the row-block split.
#include "spgemm_internal.h"

using namespace pem;

// ------------------------------------------------------------------------------------------
// a14 export: tiled C -> CSR without the reference's 16-byte-record stable_sort
// (spgemm.cu:1516-1519): C tiles are already sorted by (tile row, tile col) and entries are
// row-major inside a tile, so row R = 16 i + r is the concatenation over the tiles of tile row
// i of that tile's row-r entries.  16 lanes per tile row, lane = r.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ex_rowcount_kernel(const int *__restrict__ c_tile_rowptr, const uint16_t *__restrict__ c_mask16,
                                                          int mt, int nrows, int *__restrict__ rowcnt)
{
    int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int r = threadIdx.x & 15;
    if (i >= mt) return;
    int cnt = 0;
    for (int t = c_tile_rowptr[i]; t < c_tile_rowptr[i + 1]; ++t) cnt += __popc((unsigned)c_mask16[16 * (size_t)t + (r ^ 1)]);
    int row = 16 * i + r;
    if (row < nrows) rowcnt[row] = cnt;
}

template <typename VT>
__global__ void __launch_bounds__(256) ex_fill_kernel(const int *__restrict__ c_tile_rowptr, const int *__restrict__ c_tile_colidx,
                                                      const uint16_t *__restrict__ c_mask16, const int *__restrict__ c_tile_nnz_ptr,
                                                      const uint8_t *__restrict__ c_rowptr, const VT *__restrict__ c_vals, int mt,
                                                      int nrows, const int *__restrict__ rowptr, int *__restrict__ colidx,
                                                      VT *__restrict__ vals)
{
    int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int r = threadIdx.x & 15;
    if (i >= mt) return;
    int row = 16 * i + r;
    if (row >= nrows) return;
    int dst = rowptr[row];
    for (int t = c_tile_rowptr[i]; t < c_tile_rowptr[i + 1]; ++t) {
        unsigned m = c_mask16[16 * (size_t)t + (r ^ 1)];
        if (!m) continue;
        int src = c_tile_nnz_ptr[t] + c_rowptr[16 * (size_t)t + r];
        int cbase = c_tile_colidx[t] << 4;
        while (m) {
            int c = __builtin_ctz(m);
            m &= m - 1;
            colidx[dst] = cbase + c;
            vals[dst] = c_vals[src];
            ++dst;
            ++src;
        }
    }
}


// ------------------------------------------------------------------------------------------
// a14 export, balanced form (default).  CSR order inside a tile row is (row r, tile col, c), the tiled
// order is (tile col, r, c): a stable 16-bucket partition per tile row.  And a tile row's CSR segment starts
// where its first tile's entries start (tiles are sorted by tile row), so the CSR row pointer needs no scan
// over the rows: rowptr[16 i + r] = Ctiles_nnz_ptr[first tile of row i] + (entries of rows < r in tile row i).
// Tile rows are cut into chunks of 64 consecutive tiles, one wave per chunk, one tile per lane:
//   ex_chunkcount + scan   chunks per tile row -> first chunk of every tile row
//   ex_chunkrow    the tile row of every chunk (one table instead of a search in every wave of the two kernels below)
//   ex_chunkhist   per chunk, the entry count of each of the 16 rows        (reads the 32-byte C masks)
//   ex_chunkscan   one wave per tile row: exclusive scan of its chunks' counts (four chunks x sixteen rows per trip)
//                  -> chunk bases; the rows' totals -> the CSR row pointer
//   ex_chunkfill   one C ENTRY per lane: the wave's 64 tiles put their row prefixes, mask words and per-row exclusive
//                  tile prefixes into LDS; an entry finds its tile by a shuffle search over the tiles' offsets, its
//                  (row, rank in the row, column) off the mask (as step 3's DECODE does), and goes to
//                  rowptr[row] + chunk base + tiles before + rank.  Values are read in tiled order -- coalesced --
//                  and land inside the tile row's own CSR segment (a few tens of KB: the L2 merges the lines).
// The first form of ex_chunkfill gave every lane one TILE and walked its entries serially: ~100 vector-memory
// instructions per 64 tiles against ~16 here; it took 1.17 of the export's 1.52 ms on webbase-1M (now 0.51 of 0.83).
// ------------------------------------------------------------------------------------------
__global__ void ex_chunkcount_kernel(const int *__restrict__ c_tile_rowptr, int mt, int *__restrict__ chunkcnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < mt) chunkcnt[i] = (c_tile_rowptr[i + 1] - c_tile_rowptr[i] + 63) >> 6;
}

__global__ void __launch_bounds__(256) ex_chunkrow_kernel(const int *__restrict__ chunkptr, int mt, int *__restrict__ chunk_row)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < mt; i += nwaves)
        for (int ch = chunkptr[i] + lane; ch < chunkptr[i + 1]; ch += 64) chunk_row[ch] = i;
}

// per-row entry counts of one C tile packed as 16-bit fields: p[j] holds rows 4j..4j+3 (a chunk sums to <= 1024 per row)
__device__ __forceinline__ void ex_pack_counts(const uint4 M0, const uint4 M1, unsigned long long (&p)[4])
{
    const unsigned w[8] = {M0.x, M0.y, M0.z, M0.w, M1.x, M1.y, M1.z, M1.w};   // word q = (row 2q) << 16 | row 2q+1
#pragma unroll
    for (int j = 0; j < 4; ++j)
        p[j] = (unsigned long long)__popc(w[2 * j] >> 16) | ((unsigned long long)__popc(w[2 * j] & 0xFFFFu) << 16) |
               ((unsigned long long)__popc(w[2 * j + 1] >> 16) << 32) | ((unsigned long long)__popc(w[2 * j + 1] & 0xFFFFu) << 48);
}

__global__ void __launch_bounds__(256) ex_chunkhist_kernel(const int *__restrict__ chunkptr, const int *__restrict__ chunk_row, int mt,
                                                           const int *__restrict__ c_tile_rowptr, const uint32_t *__restrict__ c_mask,
                                                           int *__restrict__ chunkhist)
{
    const int lane = threadIdx.x & 63;
    const int ch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (ch >= chunkptr[mt]) return;        // (the grid covers the host's bound on the number of chunks)
    const int i = chunk_row[ch];
    const int t0 = c_tile_rowptr[i] + ((ch - chunkptr[i]) << 6);
    const int ntl = c_tile_rowptr[i + 1] - t0 < 64 ? c_tile_rowptr[i + 1] - t0 : 64;
    unsigned long long pk[4] = {0, 0, 0, 0};
    if (lane < ntl) {
        const long long t = t0 + lane;
        ex_pack_counts(*reinterpret_cast<const uint4 *>(c_mask + 8 * t), *reinterpret_cast<const uint4 *>(c_mask + 8 * t + 4), pk);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) pk[j] += __shfl_xor(pk[j], d, 64);
    if (lane < 16) {
        const unsigned long long sel = (lane >> 2) == 0 ? pk[0] : (lane >> 2) == 1 ? pk[1] : (lane >> 2) == 2 ? pk[2] : pk[3];
        chunkhist[16 * (size_t)ch + lane] = (int)((sel >> (16 * (lane & 3))) & 0xFFFFull);
    }
}

// one wave per tile row, lane = (chunk of the trip c4, row r): counts -> exclusive bases inside the tile row (in place), and
// the CSR row pointer of the tile row's sixteen rows
__global__ void __launch_bounds__(256) ex_chunkscan_kernel(const int *__restrict__ chunkptr, int mt, int nrows, int *chunkhist,
                                                           const int *__restrict__ c_tile_rowptr, const int *__restrict__ c_tile_nnz_ptr,
                                                           int *__restrict__ rowptr)
{
    const int lane = threadIdx.x & 63, c4 = lane >> 4, r = lane & 15;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= mt) return;
    const int ch_begin = chunkptr[i], ch_end = chunkptr[i + 1];
    int carry = 0;
    for (int ch0 = ch_begin; ch0 < ch_end; ch0 += 4) {
        const int ch = ch0 + c4;
        const int h = ch < ch_end ? chunkhist[16 * (size_t)ch + r] : 0;
        int inc = h;
        int o = __shfl_up(inc, 16, 64);
        if (c4 >= 1) inc += o;
        o = __shfl_up(inc, 32, 64);
        if (c4 >= 2) inc += o;
        if (ch < ch_end) chunkhist[16 * (size_t)ch + r] = carry + inc - h;
        carry += __shfl(inc, 48 + r, 64);
    }
    // carry = entries of row r in this tile row; the tile row's CSR segment starts where its first tile's entries start
    int pre = carry;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(pre, d, 16);
        if (r >= d) pre += o;
    }
    const int seg = c_tile_nnz_ptr[c_tile_rowptr[i]];
    if (c4 == 0 && 16 * i + r < nrows) rowptr[16 * i + r] = seg + pre - carry;
    if (i == mt - 1 && lane == 15) rowptr[nrows] = seg + pre;       // closing entry = C_nnz of the slice
}

template <typename VT>
__global__ void __launch_bounds__(256) ex_chunkfill_kernel(const int *__restrict__ chunkptr, const int *__restrict__ chunk_row, int mt,
                                                           const int *__restrict__ c_tile_rowptr, const int *__restrict__ c_tile_colidx,
                                                           const uint32_t *__restrict__ c_mask, const int *__restrict__ c_tile_nnz_ptr,
                                                           const VT *__restrict__ c_vals, const int *__restrict__ chunkbase,
                                                           const int *__restrict__ rowptr, int *__restrict__ colidx, VT *__restrict__ vals)
{
    __shared__ __attribute__((aligned(16))) uint4 s_rp[4 * 64];            // [wave][tile] prefix counts of the tile's rows, a byte each
    __shared__ __attribute__((aligned(16))) uint4 s_ex[4 * 64 * 2];        // [wave][tile] entries of row r in the chunk's earlier tiles, 16 bits each
    __shared__ unsigned s_mw[4 * 8 * 64];                                  // [wave][word q][tile]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (ch >= chunkptr[mt]) return;
    uint4 *const my_rp = s_rp + wv * 64;
    uint4 *const my_ex = s_ex + wv * 128;
    unsigned *const my_mw = s_mw + wv * 512;
    const int i = chunk_row[ch];
    const int t0 = c_tile_rowptr[i] + ((ch - chunkptr[i]) << 6);
    const int ntl = c_tile_rowptr[i + 1] - t0 < 64 ? c_tile_rowptr[i + 1] - t0 : 64;
    const bool live = lane < ntl;
    const long long t = t0 + (live ? lane : 0);
    uint4 M0 = make_uint4(0, 0, 0, 0), M1 = M0;
    if (live) {
        M0 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t);
        M1 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t + 4);
    }
    const int my_off = live ? c_tile_nnz_ptr[t] : 0x7FFFFFFF;
    const int e_end = c_tile_nnz_ptr[t0 + ntl];
    const int cbase = live ? (c_tile_colidx[t] << 4) : 0;
    // lanes 0..15: where row r of this chunk starts in the CSR arrays
    int rb = 0;
    if (lane < 16) rb = rowptr[16 * i + lane] + chunkbase[16 * (size_t)ch + lane];   // (rowptr has 16 * mt + 1 slots; rows past nrows hold no entry)
    unsigned long long pk[4], ex[4];
    ex_pack_counts(M0, M1, pk);
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // exclusive scan over the lanes of the packed per-row counts
        unsigned long long v = pk[j];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long u = __shfl_up(v, d, 64);
            if (lane >= d) v += u;
        }
        ex[j] = v - pk[j];
    }
    const unsigned w[8] = {M0.x, M0.y, M0.z, M0.w, M1.x, M1.y, M1.z, M1.w};
    unsigned rp[4] = {0, 0, 0, 0};
    {
        int run = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            rp[q >> 1] |= (unsigned)run << (16 * (q & 1));
            run += __popc(w[q] >> 16);
            rp[q >> 1] |= (unsigned)run << (16 * (q & 1) + 8);
            run += __popc(w[q] & 0xFFFFu);
        }
    }
    my_rp[lane] = make_uint4(rp[0], rp[1], rp[2], rp[3]);
    my_ex[2 * lane] = make_uint4((unsigned)ex[0], (unsigned)(ex[0] >> 32), (unsigned)ex[1], (unsigned)(ex[1] >> 32));
    my_ex[2 * lane + 1] = make_uint4((unsigned)ex[2], (unsigned)(ex[2] >> 32), (unsigned)ex[3], (unsigned)(ex[3] >> 32));
#pragma unroll
    for (int q = 0; q < 8; ++q) my_mw[q * 64 + lane] = w[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned short *const ex16 = reinterpret_cast<const unsigned short *>(my_ex);
    const int e_begin = __shfl(my_off, 0, 64);
    for (int ebase = e_begin; ebase < e_end; ebase += 64) {   // wave-uniform trip count
        const int e = ebase + lane;
        const bool valid = e < e_end;
        int ti = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const int probe = __shfl(my_off, ti + step, 64);
            if (probe <= e) ti += step;
        }
        const int toff = __shfl(my_off, ti, 64), cb = __shfl(cbase, ti, 64);
        const unsigned n = valid ? (unsigned)(e - toff) : 0u;
        const uint4 rp4 = my_rp[ti];
        const bool h8 = (rp4.z & 0xFFu) <= n;
        const unsigned d0 = h8 ? rp4.z : rp4.x, d1 = h8 ? rp4.w : rp4.y;
        const bool h4 = (d1 & 0xFFu) <= n;
        const unsigned d = h4 ? d1 : d0;
        const bool h2 = ((d >> 16) & 0xFFu) <= n;
        const unsigned hh = h2 ? d >> 16 : d & 0xFFFFu;
        const bool h1 = (hh >> 8) <= n;
        const int r = (h8 ? 8 : 0) + (h4 ? 4 : 0) + (h2 ? 2 : 0) + (h1 ? 1 : 0);
        const unsigned k0 = n - (h1 ? hh >> 8 : hh & 0xFFu);            // rank inside the tile's row r
        const int base_r = __shfl(rb, r, 64);
        if (!valid) continue;
        const unsigned word = my_mw[(r >> 1) * 64 + ti];
        unsigned m = (r & 1) ? word & 0xFFFFu : word >> 16, k = k0;
        const unsigned t8 = __popc(m & 0xFFu);
        const bool g8 = k >= t8;
        k -= g8 ? t8 : 0u;
        m = g8 ? m >> 8 : m;
        const unsigned t4 = __popc(m & 0xFu);
        const bool g4 = k >= t4;
        k -= g4 ? t4 : 0u;
        m = g4 ? m >> 4 : m;
        const unsigned t2 = __popc(m & 3u);
        const bool g2 = k >= t2;
        k -= g2 ? t2 : 0u;
        m = g2 ? m >> 2 : m;
        const bool g1 = k >= (m & 1u);
        const int c = (g8 ? 8 : 0) + (g4 ? 4 : 0) + (g2 ? 2 : 0) + (g1 ? 1 : 0);
        const int dst = base_r + (int)ex16[ti * 16 + r] + (int)k0;
        colidx[dst] = cb + c;
        vals[dst] = c_vals[e];
    }
}

// per tile row of A: tile-level intermediate products (work estimate for the row-block split)
__global__ void split_rowprod_kernel(const int *__restrict__ a_tile_rowptr, const int *__restrict__ a_tile_colidx,
                                     const int *__restrict__ b_tile_rowptr, int mt, long long *__restrict__ rowprod)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mt) return;
    long long s = 0;
    for (int a = a_tile_rowptr[i]; a < a_tile_rowptr[i + 1]; ++a) {
        int k = a_tile_colidx[a];
        s += b_tile_rowptr[k + 1] - b_tile_rowptr[k];
    }
    rowprod[i] = s;
}

// the plan's value type is A's (checked equal to B's at plan creation)
static bool export_type_ok(const pem_cplan *p, int value_bytes, const char *fn)
{
    if (p->A->value_bytes == value_bytes) return true;
    set_error("%s: the plan holds %s values; use the %s export entry points", fn, p->A->value_bytes == 4 ? "fp32" : "fp64",
              p->A->value_bytes == 4 ? "_f32" : "fp64");
    return false;
}

template <typename VT>
static pem_status export_csr_device_impl(pem_ctx *ctx, const pem_cplan *p, int32_t *d_rowptr, int32_t *d_colidx, VT *d_vals)
{
    if (!ctx || !p || !d_rowptr) return PEM_E_INVALID;
    if (p->state < 3) {
        set_error("pem_c_export_csr: step 3 has not run");
        return PEM_E_STATE;
    }
    if (!export_type_ok(p, (int)sizeof(VT), "pem_c_export_csr")) return PEM_E_INVALID;
    if (p->nnz_c > 0 && (!d_colidx || !d_vals)) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const int mt = p->tr_hi - p->tr_lo;
    const int r0 = p->tr_lo * 16, r1 = p->tr_hi * 16 < p->A->rows ? p->tr_hi * 16 : p->A->rows;
    const int nrows = r1 - r0;
    PEM_HIP(hipEventRecord(ctx->ev[6], st));
    PEM_HIP(hipMemsetAsync(d_rowptr, 0, sizeof(int) * ((size_t)nrows + 1), st));
    if (mt > 0 && nrows > 0 && p->opt_export_rows) {   // 16 lanes per tile row, serial over its tiles (A/B baseline)
        PEM_TRY(ensure_c_rowptr(ctx, p));
        PEM_LAUNCH(ctx, ex_rowcount_kernel, grid_for((size_t)mt * 16, 256), 256, p->c_tile_rowptr.as<int>(), p->c_mask.as<uint16_t>(), mt, nrows,
                   d_rowptr);
        PEM_TRY(exclusive_scan_i32(ctx, d_rowptr, d_rowptr, (size_t)nrows, nullptr));
        if (p->nnz_c > 0)
            PEM_LAUNCH(ctx, ex_fill_kernel<VT>, grid_for((size_t)mt * 16, 256), 256, p->c_tile_rowptr.as<int>(), p->c_tile_colidx.as<int>(),
                       p->c_mask.as<uint16_t>(), p->c_tile_nnz_ptr.as<int>(), p->c_rowptr.as<uint8_t>(), p->c_vals.as<VT>(), mt, nrows,
                       d_rowptr, d_colidx, d_vals);
    } else if (mt > 0 && nrows > 0) {
        const size_t maxchunks = (size_t)p->ntiles_c / 64 + (size_t)mt + 1;   // every tile row adds at most one partial chunk
        DevBuf &chunkptr = ctx->tmp[4], &chunkhist = ctx->tmp[5], &chunkrow = ctx->tmp[6], &rp16 = ctx->tmp[7];
        PEM_TRY(arena_phase(ctx->arena, {{&chunkptr, sizeof(int) * ((size_t)mt + 4)}, {&chunkhist, sizeof(int) * 16 * (maxchunks + 1)},
                                         {&chunkrow, sizeof(int) * (maxchunks + 4)}, {&rp16, sizeof(int) * (16 * (size_t)mt + 4)}}));
        PEM_TRY(chunkptr.reserve(sizeof(int) * ((size_t)mt + 4)));
        PEM_TRY(chunkhist.reserve(sizeof(int) * 16 * (maxchunks + 1)));
        PEM_TRY(chunkrow.reserve(sizeof(int) * (maxchunks + 4)));
        PEM_LAUNCH(ctx, ex_chunkcount_kernel, grid_for((size_t)mt, 256), 256, p->c_tile_rowptr.as<int>(), mt, chunkptr.as<int>());
        PEM_TRY(exclusive_scan_i32(ctx, chunkptr.as<int>(), chunkptr.as<int>(), (size_t)mt, nullptr));
        // (the number of chunks is only known on the device: the grids cover the bound, waves past the end leave at once)
        if (p->ntiles_c > 0 && p->nnz_c > 0) {
            // the row pointer is written in slots of sixteen per tile row; the caller's array ends at nrows + 1, which the last
            // tile row may fall short of filling -- so a slice whose row count is no multiple of 16 goes through a padded copy
            int *rp = d_rowptr;
            const bool padded = nrows != 16 * mt;
            if (padded) {
                PEM_TRY(rp16.reserve(sizeof(int) * (16 * (size_t)mt + 4)));
                rp = rp16.as<int>();
            }
            PEM_LAUNCH(ctx, ex_chunkrow_kernel, grid_for((size_t)mt * 64, 256), 256, chunkptr.as<int>(), mt, chunkrow.as<int>());
            PEM_LAUNCH(ctx, ex_chunkhist_kernel, grid_for(maxchunks * 64, 256), 256, chunkptr.as<int>(), chunkrow.as<int>(), mt,
                       p->c_tile_rowptr.as<int>(), p->c_mask.as<uint32_t>(), chunkhist.as<int>());
            PEM_LAUNCH(ctx, ex_chunkscan_kernel, grid_for((size_t)mt * 64, 256), 256, chunkptr.as<int>(), mt, padded ? 16 * mt : nrows, chunkhist.as<int>(),
                       p->c_tile_rowptr.as<int>(), p->c_tile_nnz_ptr.as<int>(), rp);
            PEM_LAUNCH(ctx, ex_chunkfill_kernel<VT>, grid_for(maxchunks * 64, 256), 256, chunkptr.as<int>(), chunkrow.as<int>(), mt,
                       p->c_tile_rowptr.as<int>(), p->c_tile_colidx.as<int>(), p->c_mask.as<uint32_t>(), p->c_tile_nnz_ptr.as<int>(),
                       p->c_vals.as<VT>(), chunkhist.as<int>(), rp, d_colidx, d_vals);
            if (padded) PEM_HIP(hipMemcpyAsync(d_rowptr, rp, sizeof(int) * ((size_t)nrows + 1), hipMemcpyDeviceToDevice, st));
        }
    }
    PEM_HIP(hipEventRecord(ctx->ev[7], st));
    return PEM_OK;
}

template <typename VT>
static pem_status export_csr_impl(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rowptr, int32_t *colidx, VT *vals)
{
    if (!ctx || !p) return PEM_E_INVALID;
    if (p->state < 3) {
        set_error("pem_c_export_csr: step 3 has not run");
        return PEM_E_STATE;
    }
    if (nnz) *nnz = p->nnz_c;
    if (!rowptr) return PEM_OK;   // size query
    if (!export_type_ok(p, (int)sizeof(VT), "pem_c_export_csr")) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    const int r0 = p->tr_lo * 16, r1 = p->tr_hi * 16 < p->A->rows ? p->tr_hi * 16 : p->A->rows;
    const size_t nrows = (size_t)(r1 - r0), nz = (size_t)p->nnz_c;
    DevBuf dR, dC, dV;
    PEM_TRY(dR.reserve(sizeof(int) * (nrows + 4)));
    PEM_TRY(dC.reserve(sizeof(int) * (nz + 4)));
    PEM_TRY(dV.reserve(sizeof(VT) * (nz + 1)));
    PEM_TRY(export_csr_device_impl<VT>(ctx, p, dR.as<int>(), dC.as<int>(), dV.as<VT>()));
    PEM_HIP(hipMemcpyAsync(rowptr, dR.p, sizeof(int) * (nrows + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (nz) {
        if (!colidx || !vals) return PEM_E_INVALID;
        PEM_HIP(hipMemcpyAsync(colidx, dC.p, sizeof(int) * nz, hipMemcpyDeviceToHost, ctx->stream));
        PEM_HIP(hipMemcpyAsync(vals, dV.p, sizeof(VT) * nz, hipMemcpyDeviceToHost, ctx->stream));
    }
    int hf[NUM_FLAGS];
    PEM_TRY(read_flags(ctx, hf));    // (synchronises) -- the export's scans gave up: the row pointer just copied is not valid
    PEM_TRY(check_internal(hf));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess) ctx->timings.export_ms = ms;
    return PEM_OK;
}

template <typename VT>
static pem_status export_coo_impl(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rows, int32_t *cols, VT *vals)
{
    if (!ctx || !p) return PEM_E_INVALID;
    if (p->state < 3) {
        set_error("pem_c_export_coo: step 3 has not run");
        return PEM_E_STATE;
    }
    if (nnz) *nnz = p->nnz_c;
    if (!rows) return PEM_OK;
    const int r0 = p->tr_lo * 16, r1 = p->tr_hi * 16 < p->A->rows ? p->tr_hi * 16 : p->A->rows;
    std::vector<int> rp((size_t)(r1 - r0) + 1);
    PEM_TRY(export_csr_impl<VT>(ctx, p, nullptr, rp.data(), cols, vals));
    for (int r = r0; r < r1; ++r)   // sorted (row, col) order = CSR order (spgemm.cu:1516-1519)
        for (int e = rp[(size_t)(r - r0)]; e < rp[(size_t)(r - r0) + 1]; ++e) rows[e] = r;
    return PEM_OK;
}

extern "C" pem_status pem_c_export_csr_device(pem_ctx *ctx, const pem_cplan *p, int32_t *d_rowptr, int32_t *d_colidx, double *d_vals)
{
    return export_csr_device_impl<double>(ctx, p, d_rowptr, d_colidx, d_vals);
}
extern "C" pem_status pem_c_export_csr_device_f32(pem_ctx *ctx, const pem_cplan *p, int32_t *d_rowptr, int32_t *d_colidx, float *d_vals)
{
    return export_csr_device_impl<float>(ctx, p, d_rowptr, d_colidx, d_vals);
}
extern "C" pem_status pem_c_export_csr(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rowptr, int32_t *colidx, double *vals)
{
    return export_csr_impl<double>(ctx, p, nnz, rowptr, colidx, vals);
}
extern "C" pem_status pem_c_export_csr_f32(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rowptr, int32_t *colidx, float *vals)
{
    return export_csr_impl<float>(ctx, p, nnz, rowptr, colidx, vals);
}
extern "C" pem_status pem_c_export_coo(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rows, int32_t *cols, double *vals)
{
    return export_coo_impl<double>(ctx, p, nnz, rows, cols, vals);
}
extern "C" pem_status pem_c_export_coo_f32(pem_ctx *ctx, const pem_cplan *p, int64_t *nnz, int32_t *rows, int32_t *cols, float *vals)
{
    return export_coo_impl<float>(ctx, p, nnz, rows, cols, vals);
}

// weight of every tile row of A in C = A*B: its tile-level products + its tiles + 1 (so empty-product rows still spread out)
static pem_status tile_row_weights(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, std::vector<double> &w)
{
    if (A->cols != B->rows) {
        set_error("tile-row weights: inner dimensions differ");
        return PEM_E_INVALID;
    }
    const int mt = A->tile_rows;
    DevBuf &rp = ctx->tmp[3];
    PEM_TRY(rp.reserve(sizeof(long long) * ((size_t)mt + 1)));
    PEM_LAUNCH(ctx, split_rowprod_kernel, grid_for((size_t)mt, 256), 256, A->tile_rowptr.as<int>(), A->tile_colidx.as<int>(),
               B->tile_rowptr.as<int>(), mt, rp.as<long long>());
    std::vector<long long> h((size_t)mt);
    PEM_HIP(hipMemcpyAsync(h.data(), rp.p, sizeof(long long) * (size_t)mt, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    w.resize((size_t)mt);
    for (int i = 0; i < mt; ++i) w[(size_t)i] = (double)h[(size_t)i] + (double)(A->h_tile_rowptr[(size_t)i + 1] - A->h_tile_rowptr[(size_t)i]) + 1.0;
    return PEM_OK;
}

extern "C" pem_status pem_tile_row_weights(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, double *weights)
{
    if (!ctx || !A || !B || !weights) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    std::vector<double> w;
    PEM_TRY(tile_row_weights(ctx, A, B, w));
    for (size_t i = 0; i < w.size(); ++i) weights[i] = w[i];
    return PEM_OK;
}

extern "C" pem_status pem_split_tile_rows(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, int nparts, int32_t *bounds)
{
    if (!ctx || !A || !B || !bounds || nparts < 1) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    const int mt = A->tile_rows;
    std::vector<double> w;
    PEM_TRY(tile_row_weights(ctx, A, B, w));
    std::vector<double> pre((size_t)mt + 1, 0.0);
    for (int i = 0; i < mt; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + w[(size_t)i];
    bounds[0] = 0;
    int row = 0;
    for (int g = 1; g < nparts; ++g) {
        double target = pre[(size_t)mt] * (double)g / (double)nparts;
        while (row < mt && pre[(size_t)row + 1] <= target) ++row;
        bounds[g] = row;
    }
    bounds[nparts] = mt;
    return PEM_OK;
}
