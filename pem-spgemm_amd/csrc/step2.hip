// step2.hip -- rows a10 (offsets), a11, a12: C tile masks, per-tile entry counts and offsets.
#include "spgemm_internal.h"

using namespace pem;

// ------------------------------------------------------------------------------------------
// step 2
// ------------------------------------------------------------------------------------------
// a10 pairs_a / pairs_b (spgemm.cu:423-432): gather the expanded ids through the sort permutation
__global__ void s2_pairs_kernel(const uint32_t *__restrict__ perm, const int *__restrict__ prod_a, const int *__restrict__ prod_b, size_t n,
                                int *__restrict__ pairs_a, int *__restrict__ pairs_b)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t q = perm[p];
    pairs_a[p] = prod_a[q];
    pairs_b[p] = prod_b[q];
}

// a11 (spgemm.cu:499-550).  16 lanes per C tile, lane = tile row r.  For every pair the C row
// is OR_{kk in Amask[r]} Bmask[kk] -- work proportional to the A tile's nnz, not 16x16 ANDs.
// Stored in the reference's packing: uint32 word q = (row 2q)<<16 | row 2q+1, i.e. the
// uint16 at index r^1.
__global__ void __launch_bounds__(256) s2_cmask_kernel(const int *__restrict__ pairs_offset, const int *__restrict__ pairs_a,
                                                       const int *__restrict__ pairs_b, long long ntc,
                                                       const uint16_t *__restrict__ a_masks, const uint16_t *__restrict__ b_masks,
                                                       uint16_t *__restrict__ c_mask16, int *__restrict__ c_tile_nnz)
{
    long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int r = threadIdx.x & 15;
    const bool live = t < ntc;
    unsigned cm = 0;
    if (live) {
        int p0 = pairs_offset[t], p1 = pairs_offset[t + 1];
        for (int p = p0; p < p1; ++p) {
            int a = pairs_a[p], b = pairs_b[p];
            unsigned am = a_masks[16 * (size_t)a + r];
            const uint16_t *bm = b_masks + 16 * (size_t)b;
            while (am) {
                int kk = __builtin_ctz(am);
                am &= am - 1;
                cm |= bm[kk];
            }
        }
    }
    int cnt = __popc(cm);
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 16);
    if (live) {
        c_mask16[16 * t + (r ^ 1)] = (uint16_t)cm;
        if (r == 0) c_tile_nnz[t] = cnt;
    }
}

// a12 (spgemm.cu:552-591): intra-tile row pointers + packed (r<<4|c) bytes
__global__ void __launch_bounds__(256) s2_crowcol_kernel(const uint16_t *__restrict__ c_mask16, const int *__restrict__ c_tile_nnz_ptr,
                                                         long long ntc, uint8_t *__restrict__ c_rowptr, uint8_t *__restrict__ c_rowcolidx)
{
    long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int r = threadIdx.x & 15;
    const bool live = t < ntc;
    unsigned cm = live ? c_mask16[16 * t + (r ^ 1)] : 0u;
    int cnt = __popc(cm), inc = cnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int v = __shfl_up(inc, d, 16);
        if (r >= d) inc += v;
    }
    if (!live) return;
    int off = inc - cnt;
    c_rowptr[16 * t + r] = (uint8_t)off;
    uint8_t *dst = c_rowcolidx + c_tile_nnz_ptr[t] + off;
    while (cm) {
        int c = __builtin_ctz(cm);
        cm &= cm - 1;
        *dst++ = (uint8_t)((r << 4) | c);
    }
}

// ------------------------------------------------------------------------------------------
// step 2/3, wide mappings (default).  The 16-lanes-per-tile kernels above issue one vector
// memory instruction per 4 tiles with most lanes idle (C tiles hold ~3 entries, ~1 pair) and
// are bound by memory-instruction issue, not bytes.  These forms give every lane a whole
// unit of work: one C tile per lane for the masks (two 16-byte loads per operand tile, the
// 16x16 boolean product in registers), one C entry per lane for the numeric step.
// ------------------------------------------------------------------------------------------
// The boolean product of one tile pair, shared by the fused step-2 kernel: B's 16 row masks are parked in LDS
// ([dword q][lane]; a lane only ever reads what it wrote itself -- same wave, program order -- so no barrier is
// needed) and C row r |= OR_{kk in A row r} B row kk iterates over A's nonzeros only.  Two rows share a dword
// (w[q] = row 2q | row 2q+1 << 16, the natural uint16 layout).
struct S2Masks {
    uint4 A0, A1, B0, B1;   // the 16 row masks of the A tile and of the B tile, two rows per dword
};
__device__ __forceinline__ S2Masks s2_load_masks(const uint16_t *__restrict__ a_masks, const uint16_t *__restrict__ b_masks, const int a, const int b)
{
    S2Masks m;
    m.A0 = *reinterpret_cast<const uint4 *>(a_masks + 16 * (size_t)a);
    m.A1 = *reinterpret_cast<const uint4 *>(a_masks + 16 * (size_t)a + 8);
    m.B0 = *reinterpret_cast<const uint4 *>(b_masks + 16 * (size_t)b);
    m.B1 = *reinterpret_cast<const uint4 *>(b_masks + 16 * (size_t)b + 8);
    return m;
}
__device__ __forceinline__ void s2_pair_mask(const S2Masks &m, unsigned (*bl)[256], const int tid, unsigned (&cw)[8])
{
    bl[0][tid] = m.B0.x; bl[1][tid] = m.B0.y; bl[2][tid] = m.B0.z; bl[3][tid] = m.B0.w;
    bl[4][tid] = m.B1.x; bl[5][tid] = m.B1.y; bl[6][tid] = m.B1.z; bl[7][tid] = m.B1.w;
    const unsigned aw[8] = {m.A0.x, m.A0.y, m.A0.z, m.A0.w, m.A1.x, m.A1.y, m.A1.z, m.A1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned am = aw[q];            // bits 0-15: row 2q, bits 16-31: row 2q+1
        unsigned acc = 0;
        while (am) {                    // one iteration per nonzero of A in these two rows
            const int bit = __builtin_ctz(am);
            am &= am - 1;
            const int kk = bit & 15;
            const unsigned bwd = bl[kk >> 1][tid];            // rows kk&~1 (low half) and kk|1 (high half)
            const unsigned brow = (kk & 1) ? (bwd >> 16) : (bwd & 0xFFFFu);
            acc |= brow << (bit & 16);
        }
        cw[q] |= acc;
    }
}

// The same product from A's TRANSPOSED masks (column k of the A tile = the rows that hold it): C[r] |= B[k] for every row r in
// column k -- iterated over the inner indices k that are occupied on BOTH sides (A's occupied columns & B's occupied rows: one to
// three for most pairs, it is what step 1's pruning tested), each spread over the eight row-pair words by bit-field ops with no
// branch.  The row-major form above walks every nonzero of the A tile, at the pace of the wave's densest row pair; this one does
// popcount(live) trips of ~40 vector-ALU instructions (s2_tiles_kernel was bound by them: 786 per 64 pairs).
__device__ __forceinline__ void s2_pair_mask_t(const uint4 T0, const uint4 T1, const uint4 B0, const uint4 B1, unsigned live, unsigned (*bl)[256],
                                               const int tid, unsigned (&cw)[8])
{
    bl[0][tid] = B0.x; bl[1][tid] = B0.y; bl[2][tid] = B0.z; bl[3][tid] = B0.w;
    bl[4][tid] = B1.x; bl[5][tid] = B1.y; bl[6][tid] = B1.z; bl[7][tid] = B1.w;
    bl[8][tid] = T0.x; bl[9][tid] = T0.y; bl[10][tid] = T0.z; bl[11][tid] = T0.w;
    bl[12][tid] = T1.x; bl[13][tid] = T1.y; bl[14][tid] = T1.z; bl[15][tid] = T1.w;
    while (live) {
        const int k = __builtin_ctz(live);
        live &= live - 1;
        const unsigned bwd = bl[k >> 1][tid], twd = bl[8 + (k >> 1)][tid];
        const unsigned brow = (k & 1) ? (bwd >> 16) : (bwd & 0xFFFFu);     // B's row k
        const int cm = (int)((k & 1) ? (twd >> 16) : (twd & 0xFFFFu));      // the rows of the A tile that hold column k
        const unsigned bb = brow | (brow << 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(cm, 2 * q, 1), hi = (unsigned)__builtin_amdgcn_sbfe(cm, 2 * q + 1, 1);   // 0 or ~0
            cw[q] |= bb & ((lo & 0xFFFFu) | (hi & 0xFFFF0000u));
        }
    }
}

// (r<<4|c) bytes of one C tile from its masks in the natural layout (a12, spgemm.cu:582-587), row-major
__device__ __forceinline__ void s2_emit_rowcol(const unsigned (&cw)[8], uint8_t *__restrict__ dst)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned m = cw[q] & 0xFFFFu;         // row 2q
        while (m) {
            const int c = __builtin_ctz(m);
            m &= m - 1;
            *dst++ = (uint8_t)(((2 * q) << 4) | c);
        }
        m = cw[q] >> 16;                      // row 2q+1
        while (m) {
            const int c = __builtin_ctz(m);
            m &= m - 1;
            *dst++ = (uint8_t)(((2 * q + 1) << 4) | c);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Step 2 (a10 offsets + a11 + a12, spgemm.cu:483-484, 499-550, 552-591) in two kernels over step 1's pair stream.
// Step 1 leaves the live pairs in their final order (pairs_a / pairs_b: sorted by C tile, ascending k inside a tile) and,
// per pair, pair_col[p] = tile column of its C tile, sign bit set on the FIRST pair of every C tile; and per 256 pairs the
// number of first pairs among them (scanned: blk_base).  So a C tile's dense index is
//     t = blk_base[p >> 8] + (first pairs before p inside its 256)
// with no scan over pairs and no dependence between workgroups.
//
// s2_tiles_kernel, ONE PAIR PER LANE (round 4; rounds 2-3: one C tile per lane, each lane walking its tile's pair list --
// 1 325 vector instructions per wave where a lane's useful share was ~110, because a wave ran at the pace of its longest
// list, and one pair's gathers in flight per lane).  One wave per 256 pairs, four trips of 64:
//   mask    every lane forms the 16x16 boolean product of ITS pair (two 16-byte loads per operand tile); the products of a
//           C tile's pairs -- consecutive lanes -- are OR-ed toward the tile's first pair: a few shift-and-OR steps on the
//           vector ALU's data-parallel paths where the tiles hold few pairs (1.25 on average on webbase-1M: one or two
//           steps), a six-step doubling by shuffles where they hold many (band matrices).  A tile that runs over the end of
//           a trip is carried in registers; one that runs over the end of the wave's 256 pairs is finished by the wave that
//           holds its first pair (it reads on), and the next wave skips the pairs in front of its first first-pair.
//   out     _C_tileColIdx[t], pair offsets[t], Ctiles_mask[8t..] straight into the reference's dense layout, the 2-byte
//           entry count, and the entry count of every 256 tiles (at most two integer atomics per wave).
// (one small scan of the 256-tile group counts in between)
// s2_entries_kernel / s2_offsets_kernel: perTileNnz offsets from the group base, (the (r<<4|c) bytes,) step 3's chunk index.
//
// Measured and dropped: carrying (tiles, entries) through a decoupled look-back inside ONE kernel.  Flat window
// of 64 blocks: 1.27 ms (2000 blocks in flight = 30 round trips of ~2 us agent-scope loads behind the nearest
// prefix); with a ticket for the block order 1.40 ms (81 k atomics on one address, 11 ns each); two-level
// (groups of 64 blocks): 1.20 ms -- in-order completion puts every resident block behind the slowest lane of the
// oldest one, and a lane with a 40-pair tile takes 80 us; tile counts only, published at block start: 0.94 ms
// with every block polling from its first cycle, 0.78 ms with the look-back moved behind the mask loop; without
// any look-back the same kernel takes 0.36 ms.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned s2_wave_shl1(unsigned v)   // lane i <- lane i + 1 (lane 63 <- 0), on the DPP path
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

template <bool TRANSPOSED>
__global__ void __launch_bounds__(256, 8) s2_tiles_kernel(const int *__restrict__ pair_col, const int *__restrict__ pairs_a, const int *__restrict__ pairs_b,
                                                       long long npairs, const int *__restrict__ blk_base,
                                                       const uint16_t *__restrict__ a_masks_t, const uint16_t *__restrict__ b_masks,
                                                       const uint32_t *__restrict__ a_occ, const uint32_t *__restrict__ b_occ, long long ntc,
                                                       int *__restrict__ c_colidx, int *__restrict__ pairs_offset, uint32_t *__restrict__ c_mask,
                                                       int *__restrict__ group_nnz, uint16_t *__restrict__ c_cnt)
{
    __shared__ unsigned bl[TRANSPOSED ? 16 : 8][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long w = (long long)blockIdx.x * 4 + (tid >> 6);          // the wave's 256 pairs
    const long long p_lo = w * 256;
    if (p_lo >= npairs) return;
    if (w == 0 && lane == 0) pairs_offset[ntc] = (int)npairs;            // closing pair offset
    const unsigned long long lt = (1ull << lane) - 1ull;
    long long t_next = blk_base[w];                                      // dense index of the next C tile that starts in these pairs
    const long long g0 = t_next / S2_GROUP;                              // the wave's tiles are consecutive: they span at most two groups
    int gs0 = 0, gs1 = 0;                                                // entries this lane stored, by group
    // the C tile left open at the end of a trip (wave-uniform)
    bool open = false;
    unsigned ccw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long c_t = 0;
    int c_j = 0, c_p0 = 0;
    auto store_tile = [&](const long long t, const int j, const int p0, const unsigned (&cw)[8]) {
        int nnz_t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) nnz_t += __popc(cw[q]);
        if (t / S2_GROUP == g0) gs0 += nnz_t; else gs1 += nnz_t;
        c_colidx[t] = j;
        pairs_offset[t] = p0;
        if (c_cnt) c_cnt[t] = (uint16_t)nnz_t;      // the entry offsets then come from 2 bytes per tile, not from its 32-byte mask
        // reference packing: word q = (row 2q) << 16 | row 2q+1  (spgemm.cu:533-543)
        *reinterpret_cast<uint4 *>(c_mask + 8 * t) = make_uint4((cw[0] << 16) | (cw[0] >> 16), (cw[1] << 16) | (cw[1] >> 16),
                                                                (cw[2] << 16) | (cw[2] >> 16), (cw[3] << 16) | (cw[3] >> 16));
        *reinterpret_cast<uint4 *>(c_mask + 8 * t + 4) = make_uint4((cw[4] << 16) | (cw[4] >> 16), (cw[5] << 16) | (cw[5] >> 16),
                                                                    (cw[6] << 16) | (cw[6] >> 16), (cw[7] << 16) | (cw[7] >> 16));
    };
    // one trip: 64 consecutive pairs from pb; col / ida / idb = the lane's pair_col word and pair ids; over = past the wave's own
    // 256 pairs, where only the open tile is still ours
    auto trip = [&](const long long pb, const int col, const int ida, const int idb, const bool over) {
        const long long p = pb + lane;
        const bool valid = p < npairs;
        const bool head = col < 0;
        const unsigned long long H = __ballot(head);
        const int f = H ? __builtin_ctzll(H) : 64;                       // the trip's first first-pair
        // my pair is this wave's if a tile of ours covers it: the open tile covers the lanes in front of f, tiles that start in
        // the trip cover the rest -- unless the wave is past its own 256 pairs
        const bool mine = valid && (lane < f ? open : !over);
        unsigned cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // natural layout: cw[q] = row 2q | row 2q+1 << 16
        if (mine) {
            if constexpr (TRANSPOSED) {   // (a_masks_t = A's transposed masks)
                const uint4 T0 = *reinterpret_cast<const uint4 *>(a_masks_t + 16 * (size_t)ida), T1 = *reinterpret_cast<const uint4 *>(a_masks_t + 16 * (size_t)ida + 8);
                const uint4 B0 = *reinterpret_cast<const uint4 *>(b_masks + 16 * (size_t)idb), B1 = *reinterpret_cast<const uint4 *>(b_masks + 16 * (size_t)idb + 8);
                const unsigned live = (a_occ[ida] & 0xFFFFu) & (b_occ[idb] >> 16);
                s2_pair_mask_t(T0, T1, B0, B1, live, bl, tid, cw);
            } else {                      // (a_masks_t = A's row masks)
                const S2Masks m = s2_load_masks(a_masks_t, b_masks, ida, idb);
                s2_pair_mask(m, bl, tid, cw);
            }
        }
        // OR toward the first pair of every tile: lane i absorbs lane i + 1 while lane i + 1 is a further pair of the same tile
        const unsigned long long N = __ballot(mine && !head);            // absorbed lanes
        const bool absorb = lane < 63 && ((N >> (lane + 1)) & 1ull);
        int longest = 0;
        for (unsigned long long m = N; m; m = N & (m << 1)) ++longest;   // longest run of absorbed lanes
        if (longest <= 8) {
            for (int it = 0; it < longest; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned nx = s2_wave_shl1(cw[q]);
                    cw[q] |= absorb ? nx : 0u;
                }
            }
        } else {
            // lanes behind me in my tile (within the trip): the run of absorbed lanes that follows
            const unsigned long long after = lane < 63 ? (~N >> (lane + 1)) | (1ull << (63 - lane)) : 1ull;
            const int reach = __builtin_ctzll(after);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned nx = (unsigned)__shfl_down((int)cw[q], d, 64);
                    cw[q] |= reach >= d ? nx : 0u;
                }
            }
        }
        if (open) {
            if (f > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ccw[q] |= (unsigned)__builtin_amdgcn_readlane((int)cw[q], 0);
            }
            if (f < 64) {                                                // a first pair follows: the open tile is complete
                if (lane == 0) store_tile(c_t, c_j, c_p0, ccw);
                open = false;
#pragma unroll
                for (int q = 0; q < 8; ++q) ccw[q] = 0;
            }
        }
        if (over) return;
        // tiles that start in this trip: all but the last are complete (the next first pair follows them inside the trip); the
        // last one stays open -- whether a further pair of it follows is only known in the next trip
        const unsigned long long Hv = H & __ballot(valid);
        if (Hv) {
            const int hl = 63 - __builtin_clzll(Hv);
            const long long t = t_next + __popcll(Hv & lt);
            if (head && valid && lane != hl) store_tile(t, col & 0x7FFFFFFF, (int)p, cw);
            c_t = t_next + __popcll(Hv) - 1;
            c_j = __builtin_amdgcn_readlane(col, hl) & 0x7FFFFFFF;
            c_p0 = (int)(pb + hl);
#pragma unroll
            for (int q = 0; q < 8; ++q) ccw[q] = (unsigned)__builtin_amdgcn_readlane((int)cw[q], hl);
            open = true;
            t_next += __popcll(Hv);
        }
    };
    // The wave's own four trips: their pair words and ids are loaded up front, all in flight together -- a trip then waits for ONE
    // round trip (its mask gathers) instead of three in a row (pair word -> ids -> masks): the kernel is a chain of dependent
    // round trips per wave, not bandwidth (67 us for 0.25 GB).
    int pc[4], pia[4], pib[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = p_lo + 64 * j + lane;
        const bool valid = p < npairs;
        pc[j] = valid ? pair_col[p] : (int)0x80000000;                   // (past the end of the pairs: a first pair, which closes the open tile)
        pia[j] = valid ? pairs_a[p] : 0;
        pib[j] = valid ? pairs_b[p] : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (p_lo + 64 * j < npairs) trip(p_lo + 64 * j, pc[j], pia[j], pib[j], false);
    for (long long pb = p_lo + 256; pb < npairs && open; pb += 64) {     // the open tile runs on past the wave's pairs
        const long long p = pb + lane;
        const bool valid = p < npairs;
        trip(pb, valid ? pair_col[p] : (int)0x80000000, valid ? pairs_a[p] : 0, valid ? pairs_b[p] : 0, true);
    }
    if (open && lane == 0) store_tile(c_t, c_j, c_p0, ccw);
    // entry counts per group of S2_GROUP tiles
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        gs0 += __shfl_xor(gs0, d, 64);
        gs1 += __shfl_xor(gs1, d, 64);
    }
    if (lane == 0) {
        if (gs0) atomicAdd(&group_nnz[g0], gs0);
        if (gs1) atomicAdd(&group_nnz[g0 + 1], gs1);
    }
}

// a11's offsets + a12 (spgemm.cu:546, 1288, 552-591), one C tile per lane, S2_GROUP tiles per block: entry offsets =
// the group's base (scanned group counts) + a block scan of the masks' popcounts; the (r<<4|c) bytes; and, for step 3,
// the tile every S3_CHUNK-entry chunk of C starts in.
__global__ void __launch_bounds__(S2_GROUP) s2_entries_kernel(const uint32_t *__restrict__ c_mask, long long ntc, const int *__restrict__ group_base,
                                                              long long cap_nnz, int *__restrict__ c_tile_nnz_ptr, uint8_t *__restrict__ c_rowcolidx,
                                                              int *__restrict__ chunk_tile, int *__restrict__ flags, WarmCheck wc,
                                                              const long long *__restrict__ d_scalars, const int *__restrict__ bin_count)
{
    __shared__ int w_nnz[S2_GROUP / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every size of the pass is final by now (the entry total came out of the group scan just before this launch): the
    // check of a repeat pass rides along instead of taking a launch of its own at the end
    if (wc.on && blockIdx.x == 0 && tid == 0) warm_check(wc, d_scalars, bin_count, flags);
    const long long t = (long long)blockIdx.x * S2_GROUP + tid;
    unsigned cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < ntc) {
        const uint4 M0 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t);
        const uint4 M1 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t + 4);
        const unsigned w[8] = {M0.x, M0.y, M0.z, M0.w, M1.x, M1.y, M1.z, M1.w};   // word q = (row 2q) << 16 | row 2q+1
#pragma unroll
        for (int q = 0; q < 8; ++q) cw[q] = (w[q] << 16) | (w[q] >> 16);          // natural layout
    }
    int nnz_t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) nnz_t += __popc(cw[q]);
    int inc = nnz_t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) w_nnz[wave] = inc;
    __syncthreads();
    long long off = (long long)group_base[blockIdx.x] + inc - nnz_t;
#pragma unroll
    for (int w = 0; w < S2_GROUP / 64; ++w)
        if (w < wave) off += w_nnz[w];
    if (t > ntc) return;
    if (t == ntc) {                       // closing offset = C_nnz
        c_tile_nnz_ptr[ntc] = (int)off;
        return;
    }
    c_tile_nnz_ptr[t] = (int)off;
    if (off + nnz_t > cap_nnz) {          // cannot happen: the host sized the buffers from the same counts
        flags[FLAG_CAPACITY] = 1;
        return;
    }
    // step 3 deals C entries in chunks of S3_CHUNK: note the tile every chunk starts in (saves its waves a search)
    for (long long ch = (off + S3_CHUNK - 1) / S3_CHUNK; ch * S3_CHUNK < off + nnz_t; ++ch) chunk_tile[ch] = (int)t;
    s2_emit_rowcol(cw, c_rowcolidx + off);
}

// The same offsets without the entries: where step 3 reads an entry's (row, column) off the tile's mask (DECODE below),
// nothing on the pass needs the (r<<4|c) bytes, and the offsets come from the 2-byte entry counts s2_tiles_kernel left --
// 39 MB in, 78 MB out on webbase-1M, where s2_entries_kernel re-reads 618 MB of masks to emit 69 MB of bytes (0.18 ms
// against 0.03).  The bytes (Ctiles_rowColIdx, spgemm.cu:582-587) are then materialised on demand like Ctiles_rowPtr
// (ensure_c_rowcolidx).
__global__ void __launch_bounds__(256) s2_offsets_kernel(const uint16_t *__restrict__ c_cnt, long long ntc, const int *__restrict__ group_base,
                                                         int *__restrict__ c_tile_nnz_ptr, int *__restrict__ chunk_tile, int *__restrict__ flags,
                                                         WarmCheck wc, const long long *__restrict__ d_scalars, const int *__restrict__ bin_count)
{
    // one WAVE per group of S2_GROUP = 256 tiles, four consecutive tiles per lane (one 8-byte load, one 16-byte store): the
    // group's base comes from the scanned group counts, so no wave waits for another (one tile per lane and a block scan
    // took 73 us)
    static_assert(S2_GROUP == 256, "four tiles per lane of one wave");
    const int lane = threadIdx.x & 63;
    const long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wc.on && g == 0 && lane == 0) warm_check(wc, d_scalars, bin_count, flags);
    const long long t = g * S2_GROUP + 4 * lane;
    if (t > ntc) return;                                    // (only lanes above a live one leave: the scan below reads downwards)
    int n[4] = {0, 0, 0, 0};
    if (t + 4 <= ntc) {
        const uint2 q = *reinterpret_cast<const uint2 *>(c_cnt + t);
        n[0] = (int)(q.x & 0xFFFFu);
        n[1] = (int)(q.x >> 16);
        n[2] = (int)(q.y & 0xFFFFu);
        n[3] = (int)(q.y >> 16);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) n[k] = t + k < ntc ? (int)c_cnt[t + k] : 0;
    }
    const int tsum = n[0] + n[1] + n[2] + n[3];
    int inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    int o4[4];
    o4[0] = group_base[g] + inc - tsum;
    o4[1] = o4[0] + n[0];
    o4[2] = o4[1] + n[1];
    o4[3] = o4[2] + n[2];
    if (t + 4 <= ntc) {
        *reinterpret_cast<int4 *>(c_tile_nnz_ptr + t) = make_int4(o4[0], o4[1], o4[2], o4[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t + k <= ntc) c_tile_nnz_ptr[t + k] = o4[k];             // (t + k == ntc: the closing offset = C_nnz)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (t + k >= ntc) break;
        const long long off = o4[k];
        for (long long ch = (off + S3_CHUNK - 1) / S3_CHUNK; ch * S3_CHUNK < off + n[k]; ++ch) chunk_tile[ch] = (int)(t + k);
    }
}

// Ctiles_rowPtr (spgemm.cu:579-580) from the stored masks, one C tile per lane.  Nothing on the default path reads
// it (step 3 and the export work from the masks), so it is materialised on demand: 16 bytes per C tile that the
// mask kernel no longer writes on every pass (0.3 GB on webbase-1M).
__global__ void __launch_bounds__(256) s2_crowptr_kernel(const uint32_t *__restrict__ c_mask, long long ntc, uint8_t *__restrict__ c_rowptr)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntc) return;
    const uint4 M0 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t);
    const uint4 M1 = *reinterpret_cast<const uint4 *>(c_mask + 8 * t + 4);
    const unsigned w[8] = {M0.x, M0.y, M0.z, M0.w, M1.x, M1.y, M1.z, M1.w};   // word q = (row 2q) << 16 | row 2q+1
    unsigned rp[4] = {0, 0, 0, 0};
    int run = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        rp[q >> 1] |= (unsigned)run << (16 * (q & 1));
        run += __popc(w[q] >> 16);
        rp[q >> 1] |= (unsigned)run << (16 * (q & 1) + 8);
        run += __popc(w[q] & 0xFFFFu);
    }
    *reinterpret_cast<uint4 *>(c_rowptr + 16 * t) = make_uint4(rp[0], rp[1], rp[2], rp[3]);
}

pem_status pem::step2_impl(pem_ctx *ctx, pem_cplan *p)
{
    if (p->state < 1) {
        set_error("pem_spgemm_step2 called before step 1");
        return PEM_E_STATE;
    }
    const pem_tiled *A = p->A, *B = p->B;
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)p->npairs, ntc = (size_t)p->ntiles_c;
    if (!ctx->chain_events) PEM_HIP(hipEventRecord(ctx->ev[2], st));   // inside pem_spgemm the previous step's end event is the start
    // sizing phase "C tiles"
    PEM_TRY(arena_phase(ctx->arena, {{&p->pairs_a, sizeof(int) * (n + 4)}, {&p->pairs_b, sizeof(int) * (n + 4)},
                                     {&p->c_mask, sizeof(uint32_t) * 8 * (ntc + 1)}, {&p->c_tile_nnz_ptr, sizeof(int) * (ntc + 4)},
                                     {&p->c_tile_colidx, sizeof(int) * (ntc + 4)}, {&p->pairs_offset, sizeof(int) * (ntc + 4)},
                                     {&p->group_nnz, sizeof(int) * ((ntc + S2_GROUP - 1) / S2_GROUP + 4)}}));
    PEM_TRY(p->pairs_a.reserve(sizeof(int) * (n + 4)));
    PEM_TRY(p->pairs_b.reserve(sizeof(int) * (n + 4)));
    PEM_TRY(p->c_mask.reserve(sizeof(uint32_t) * 8 * (ntc + 1)));
    PEM_TRY(p->c_tile_nnz_ptr.reserve(sizeof(int) * (ntc + 4)));
    p->c_rowptr_valid = false;
    // the fused kernel reads the row-local scratch of the default step 1; the global-sort step 1 (PEM_OPT_STEP1_GLOBAL_SORT)
    // and PEM_OPT_WIDE = 0 take the 16-lanes-per-tile baseline kernels over the dense layout
    const bool fused = p->pairs_ready && p->opt_wide;
    p->wide = fused;
    p->verify_folded = false;
    p->flags_mirrored = false;
    p->s3_decode = false;
    p->c_rowcolidx_valid = false;
    int64_t nnzc = 0;
    if (fused) {
        PEM_TRY(p->c_tile_colidx.reserve(sizeof(int) * (ntc + 4)));
        PEM_TRY(p->pairs_offset.reserve(sizeof(int) * (ntc + 4)));
        if (n > 0) {
            // entry counts of every S2_GROUP tiles, accumulated by s2_tiles_kernel
            const size_t ngroups = (ntc + S2_GROUP - 1) / S2_GROUP;
            PEM_TRY(p->group_nnz.reserve(sizeof(int) * (ngroups + 4)));
            // (step 1's reset clears the counters of a repeat pass; the note holds for ONE step 2 -- a second step 2 on the
            // same step-1 result, through the step-wise API, must not add onto the scanned counts of the first)
            if (!p->group_nnz_cleared) PEM_HIP(hipMemsetAsync(p->group_nnz.p, 0, sizeof(int) * (ngroups + 4), st));
            p->group_nnz_cleared = false;
            int *group_nnz = p->group_nnz.as<int>();
            // Shallow plans (fewer than two pairs per C tile) read an entry's (row, column) off the mask in step 3, so the pass
            // needs no (r<<4|c) bytes: offsets come from 2-byte entry counts and Ctiles_rowColIdx is materialised on demand.
            // Deep plans keep the bytes: their many-pair kernel walks a tile's entries 64 at a time and the bytes are a small
            // part of their traffic.
            const bool deep = p->npairs >= 2 * p->ntiles_c;
            const bool decode = p->opt_decode && !deep;
            p->s3_decode = decode;
            if (decode) PEM_TRY(p->c_tile_cnt.reserve(sizeof(uint16_t) * (ntc + 8)));
            // the tile product from A's transposed masks pays on cage15-class plans (1/8 share: step 2 2.29 -> 1.66 ms) and costs on
            // the sparse ones (r2 stand-in 0.39 -> 0.46 ms, scircuit 0.077 -> 0.093: two more gathers per pair): by default on the
            // repeat passes of plans with two or more pairs per C tile, whose sizes the pass before left
            const bool transposed = p->opt_s2_transposed == 1 || (p->opt_s2_transposed == 2 && p->warm_pass && p->w_P >= 2 * p->w_TC);
            if (transposed)
                PEM_LAUNCH_NAMED(ctx, "s2_tiles_kernel<transposed>", s2_tiles_kernel<true>, (unsigned)((n + 1023) / 1024), 256, p->pair_col.as<int>(),
                                 p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)n, p->blk_heads.as<int>(), A->masks_t.as<uint16_t>(),
                                 B->masks.as<uint16_t>(), A->tile_occ.as<uint32_t>(), B->tile_occ.as<uint32_t>(), (long long)ntc,
                                 p->c_tile_colidx.as<int>(), p->pairs_offset.as<int>(), p->c_mask.as<uint32_t>(), group_nnz,
                                 decode ? p->c_tile_cnt.as<uint16_t>() : (uint16_t *)nullptr);
            else
                PEM_LAUNCH_NAMED(ctx, "s2_tiles_kernel", s2_tiles_kernel<false>, (unsigned)((n + 1023) / 1024), 256, p->pair_col.as<int>(),
                                 p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)n, p->blk_heads.as<int>(), A->masks.as<uint16_t>(),
                                 B->masks.as<uint16_t>(), A->tile_occ.as<uint32_t>(), B->tile_occ.as<uint32_t>(), (long long)ntc,
                                 p->c_tile_colidx.as<int>(), p->pairs_offset.as<int>(), p->c_mask.as<uint32_t>(), group_nnz,
                                 decode ? p->c_tile_cnt.as<uint16_t>() : (uint16_t *)nullptr);
            PEM_TRY(exclusive_scan_i32(ctx, group_nnz, group_nnz, ngroups, ctx->d_scalars + 2));
            if (p->warm_pass) {
                nnzc = p->w_nnz;
            } else {
                int64_t sc[1];
                PEM_TRY(read_scalars(ctx, ctx->d_scalars + 2, 1, sc));
                int hf[NUM_FLAGS];
                PEM_TRY(read_flags(ctx, hf));
                PEM_TRY(check_internal(hf));
                if (hf[FLAG_OVERFLOW] || sc[0] > 0x7FFFFFFFll) {
                    set_error("step 2: C has more than 2^31-1 nonzeros, beyond the int32 range of the reference's offsets");
                    return PEM_E_OVERFLOW;
                }
                nnzc = sc[0];
            }
            // sizing phase "C entries"
            PEM_TRY(arena_phase(ctx->arena, {{&p->c_rowcolidx, decode ? (size_t)0 : (size_t)nnzc + 16},
                                             {&p->s3_chunk_tile, sizeof(int) * ((size_t)nnzc / S3_CHUNK + 4)},
                                             {&p->c_vals, (size_t)A->value_bytes * ((size_t)nnzc + 1)}}));
            PEM_TRY(p->s3_chunk_tile.reserve(sizeof(int) * ((size_t)nnzc / S3_CHUNK + 4)));
            WarmCheck wc = {};
            if (p->warm_pass) wc = WarmCheck{1, p->w_P, p->w_Pall, p->w_TC, p->w_nnz, p->w_nxl, {p->w_counts[0], p->w_counts[1], p->w_counts[2], p->w_counts[3], p->w_counts[4]}, p->w_nsegs, nullptr};
            p->verify_folded = p->warm_pass;
            // s2_offsets_kernel's checking thread is the last writer of a flag in the pass: it leaves all of them in host memory
            if (p->warm_pass && decode && ctx->h_flags_dev) {
                wc.host_flags = ctx->h_flags_dev;
                p->flags_mirrored = true;
            }
            if (decode) {
                PEM_LAUNCH(ctx, s2_offsets_kernel, grid_for(((ntc + S2_GROUP) / S2_GROUP) * 64, 256), 256, p->c_tile_cnt.as<uint16_t>(), (long long)ntc,
                           group_nnz, p->c_tile_nnz_ptr.as<int>(), p->s3_chunk_tile.as<int>(), ctx->d_flags, wc,
                           reinterpret_cast<const long long *>(ctx->d_scalars), p->bin_count.as<int>());
                p->c_rowcolidx_valid = false;
            } else {
                PEM_TRY(p->c_rowcolidx.reserve((size_t)nnzc + 16));
                PEM_LAUNCH(ctx, s2_entries_kernel, (unsigned)((ntc + S2_GROUP) / S2_GROUP), S2_GROUP, p->c_mask.as<uint32_t>(), (long long)ntc, group_nnz,
                           (long long)nnzc, p->c_tile_nnz_ptr.as<int>(), p->c_rowcolidx.as<uint8_t>(), p->s3_chunk_tile.as<int>(), ctx->d_flags, wc,
                           reinterpret_cast<const long long *>(ctx->d_scalars), p->bin_count.as<int>());
                p->c_rowcolidx_valid = true;
            }
            p->compact_valid = true;
        } else {
            PEM_TRY(exclusive_scan_i32(ctx, p->c_tile_nnz_ptr.as<int>(), p->c_tile_nnz_ptr.as<int>(), 0, ctx->d_scalars + 2));
            p->compact_valid = true;   // pairs_offset[0] = 0 was set by step 1's reset
            p->c_rowcolidx_valid = true;
        }
    } else {
        PEM_TRY(ensure_compact(ctx, p));
        if (n > 0 && !p->pairs_ready)
            PEM_LAUNCH(ctx, s2_pairs_kernel, grid_for(n, 256), 256, p->sorted_perm, p->prod_a.as<int>(), p->prod_b.as<int>(), n, p->pairs_a.as<int>(),
                       p->pairs_b.as<int>());
        if (ntc > 0)
            PEM_LAUNCH(ctx, s2_cmask_kernel, grid_for(ntc * 16, 256), 256, p->pairs_offset.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(),
                       (long long)ntc, A->masks.as<uint16_t>(), B->masks.as<uint16_t>(), p->c_mask.as<uint16_t>(), p->c_tile_nnz_ptr.as<int>());
        PEM_TRY(exclusive_scan_i32(ctx, p->c_tile_nnz_ptr.as<int>(), p->c_tile_nnz_ptr.as<int>(), ntc, ctx->d_scalars + 2));
        if (p->warm_pass) {
            nnzc = p->w_nnz;
        } else {
            PEM_TRY(read_scalars(ctx, ctx->d_scalars + 2, 1, &nnzc));
        }
    }
    if (!p->warm_pass) {
        if (!fused) {
            int hf[NUM_FLAGS];
            PEM_TRY(read_flags(ctx, hf));
            if (hf[FLAG_OVERFLOW] || nnzc > 0x7FFFFFFFll) {
                set_error("step 2: C has more than 2^31-1 nonzeros, beyond the int32 range of the reference's offsets");
                return PEM_E_OVERFLOW;
            }
        }
        p->w_nnz = nnzc;
    }
    p->nnz_c = nnzc;
    if (!(fused && p->s3_decode)) {
        PEM_TRY(p->c_rowcolidx.reserve((size_t)nnzc + 16));
        p->c_rowcolidx_valid = fused;       // (the baseline kernels below fill it)
    }
    PEM_TRY(p->s3_chunk_tile.reserve(sizeof(int) * ((size_t)nnzc / S3_CHUNK + 4)));
    PEM_TRY(p->c_vals.reserve((size_t)A->value_bytes * ((size_t)nnzc + 1)));
    if (ntc > 0 && !fused) {   // the 16-lanes-per-tile baseline writes Ctiles_rowPtr as it goes, like the reference (spgemm.cu:579-580)
        PEM_TRY(p->c_rowptr.reserve(16 * (ntc + 1)));
        PEM_LAUNCH(ctx, s2_crowcol_kernel, grid_for(ntc * 16, 256), 256, p->c_mask.as<uint16_t>(), p->c_tile_nnz_ptr.as<int>(), (long long)ntc,
                   p->c_rowptr.as<uint8_t>(), p->c_rowcolidx.as<uint8_t>());
        p->c_rowptr_valid = true;
        p->c_rowcolidx_valid = true;
    }
    if (!ctx->capturing) PEM_HIP(hipEventRecord(ctx->ev[3], st));
    p->state = 2;
    return PEM_OK;
}

// Ctiles_rowPtr on demand (see s2_crowptr_kernel)
pem_status pem::ensure_c_rowptr(pem_ctx *ctx, const pem_cplan *p)
{
    if (p->c_rowptr_valid || p->state < 2) return PEM_OK;
    const size_t ntc = (size_t)p->ntiles_c;
    PEM_ENTER(ctx);
    PEM_TRY(p->c_rowptr.reserve(16 * (ntc + 1)));
    if (ntc > 0) PEM_LAUNCH(ctx, s2_crowptr_kernel, grid_for(ntc, 256), 256, p->c_mask.as<uint32_t>(), (long long)ntc, p->c_rowptr.as<uint8_t>());
    p->c_rowptr_valid = true;
    return PEM_OK;
}

// Ctiles_rowColIdx on demand (plans whose step 3 reads the masks): the entry kernel of the other plans, run once -- it
// recomputes the same offsets from the same scanned group counts and emits the bytes
pem_status pem::ensure_c_rowcolidx(pem_ctx *ctx, const pem_cplan *cp)
{
    pem_cplan *p = const_cast<pem_cplan *>(cp);
    if (p->c_rowcolidx_valid || p->state < 2) return PEM_OK;
    const size_t ntc = (size_t)p->ntiles_c;
    PEM_ENTER(ctx);
    PEM_TRY(p->c_rowcolidx.reserve((size_t)p->nnz_c + 16));
    if (ntc > 0) {
        WarmCheck wc = {};
        PEM_LAUNCH(ctx, s2_entries_kernel, (unsigned)((ntc + S2_GROUP) / S2_GROUP), S2_GROUP, p->c_mask.as<uint32_t>(), (long long)ntc,
                   p->group_nnz.as<int>(), (long long)p->nnz_c, p->c_tile_nnz_ptr.as<int>(), p->c_rowcolidx.as<uint8_t>(), p->s3_chunk_tile.as<int>(),
                   ctx->d_flags, wc, reinterpret_cast<const long long *>(ctx->d_scalars), p->bin_count.as<int>());
    }
    p->c_rowcolidx_valid = true;
    return PEM_OK;
}
