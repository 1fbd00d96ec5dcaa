// step3.hip -- row a13: the numeric step.
#include "spgemm_internal.h"

using namespace pem;

// ------------------------------------------------------------------------------------------
// step 3 (spgemm.cu:593-661).  16 lanes per C tile, one C entry per lane (strided by 16);
// pairs ascending in k-tile, bits of Amask[r] & BT[c] ascending, one fma per product, the
// accumulator lives in a register and is stored once (no global RMW, no dependence on
// zero-filled memory -- SURVEY 2.3 #1).
// ------------------------------------------------------------------------------------------
template <typename VT>
__global__ void __launch_bounds__(256) s3_accumulate_kernel(
    const int *__restrict__ pairs_offset, const int *__restrict__ pairs_a, const int *__restrict__ pairs_b, long long ntc,
    const int *__restrict__ c_tile_nnz_ptr, const uint8_t *__restrict__ c_rowcolidx, VT *__restrict__ c_vals,
    const int *__restrict__ a_nnz_ptr, const VT *__restrict__ a_vals, const uint16_t *__restrict__ a_masks,
    const uint8_t *__restrict__ a_rowptr, const int *__restrict__ b_nnz_ptr, const VT *__restrict__ b_vals,
    const uint16_t *__restrict__ b_masks, const uint8_t *__restrict__ b_rowptr, const uint16_t *__restrict__ b_masks_t)
{
    long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (t >= ntc) return;
    const int off = c_tile_nnz_ptr[t], nnz_t = c_tile_nnz_ptr[t + 1] - off;
    const int p0 = pairs_offset[t], p1 = pairs_offset[t + 1];
    for (int n = l; n < nnz_t; n += 16) {
        const unsigned rc = c_rowcolidx[off + n];
        const int r = rc >> 4, c = rc & 15;
        const unsigned clt = (1u << c) - 1u;
        VT acc = VT(0);
        for (int p = p0; p < p1; ++p) {
            const int a = pairs_a[p], b = pairs_b[p];
            const unsigned am = a_masks[16 * (size_t)a + r];
            unsigned m = am & b_masks_t[16 * (size_t)b + c];
            if (!m) continue;
            const VT *av = a_vals + a_nnz_ptr[a] + a_rowptr[16 * (size_t)a + r];
            const VT *bvbase = b_vals + b_nnz_ptr[b];
            while (m) {
                const int kk = __builtin_ctz(m);
                m &= m - 1;
                const int ao = __popc(am & ((1u << kk) - 1u));
                const int bo = __popc((unsigned)b_masks[16 * (size_t)b + kk] & clt);
                acc = pem_fma(av[ao], bvbase[b_rowptr[16 * (size_t)b + kk] + bo], acc);
            }
        }
        c_vals[off + n] = acc;
    }
}


// a13 (spgemm.cu:593-661): one C entry per lane.  A wave owns 64 consecutive C tiles; their
// value offsets and pair ranges sit one per lane in registers, so the entry -> tile lookup is a
// 6-step shuffle search and costs no memory traffic.  Per (entry, pair): one gather of the A
// row record (mask | rowptr<<16), one of B's transposed mask; per product one B row record and
// the two operand values.  Pairs ascend in k-tile, bits ascend, one fma per product: the same
// chain as the oracle.
constexpr int S3_EPW = S3_CHUNK;   // C entries per wave
constexpr int S3_BAND_MIN = 8;       // C tiles with at least this many pairs go to s3_band_kernel (deep plans)
constexpr int S3_BAND_CH = 16;       // pairs whose records one wave stages in LDS at a time (multiple of 4, at most 64)
constexpr int S3_BAND_RS = S3_BAND_CH + 4;   // row stride of the staged records (words): 16-byte aligned, rows on different banks
constexpr int S3_BAND_H = 1;         // meeting pairs a lane sums per trip of the gather loop
// 32-bit addressing (IDX32): base pointer in scalar registers + a 32-bit byte offset -- one shift per gather where 64-bit
// indexing takes a sign extension and a 64-bit shift-add (49 + 20 of the kernel's 341 static vector-ALU instructions; the step
// is bound by instruction issue on dense-tile inputs).  Valid only while every array is smaller than 4 GiB: the host checks.
template <bool IDX32, typename T> __device__ __forceinline__ T s3_ld(const T *__restrict__ base, const long long idx)
{
    if constexpr (IDX32)
        return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)((unsigned)idx * (unsigned)sizeof(T)));
    else
        return base[idx];
}

// DECODE: the entries' (row, column) are read off the C tile's mask instead of Ctiles_rowColIdx.  The 64 tiles a wave holds
// one per lane put their mask words and their intra-tile row pointers (the sixteen bytes of Ctiles_rowPtr, spgemm.cu:579-580)
// into a wave-private patch of LDS; entry n of a tile then finds its row by a 4-step search over those bytes (they never
// decrease) and its column as the k-th set bit of the row's mask: ~35 VALU and two LDS reads per entry in place of a global
// byte load, and step 2 no longer has to write (or re-read its masks for) the bytes at all.
// MARK (pruned plans only: every C tile has an entry, so tile offsets strictly increase): the entry -> tile lookup of a trip
// without the six-step shuffle search -- the tiles that start inside the trip's 64 entries mark their first entry in a
// 64-word LDS strip, one ballot turns the strip into a bit mask, and an entry's tile is (tiles started before the trip) +
// (marks at or below its lane) - 1.
template <typename VT, bool DEEP, bool BAND = false, bool DECODE = false, bool IDX32 = false, bool MARK = false>
#ifndef PEM_S3_DEEP_WAVES
#define PEM_S3_DEEP_WAVES 1   // (8 = at most 64 VGPRs for the deep variants too: 6 of them spill and a cage15 share's step 3 is 6.83 against 6.74 ms)
#endif
__global__ void __launch_bounds__(256, DEEP ? PEM_S3_DEEP_WAVES : 8) s3_accumulate_wide_kernel(
    const int *__restrict__ pairs_offset, const int *__restrict__ pairs_a, const int *__restrict__ pairs_b, long long ntc,
    const int *__restrict__ c_tile_nnz_ptr, long long nnz_c, const uint8_t *__restrict__ c_rowcolidx, VT *__restrict__ c_vals,
    const int *__restrict__ a_nnz_ptr, const VT *__restrict__ a_vals, const uint32_t *__restrict__ a_rec,
    const int *__restrict__ b_nnz_ptr, const VT *__restrict__ b_vals_t, const uint32_t *__restrict__ b_rec_t,
    const int *__restrict__ chunk_tile, const uint32_t *__restrict__ c_mask, const int epw, const int xcd)
{
    __shared__ __attribute__((aligned(16))) uint4 s_rp[DECODE ? 4 * 64 : 1];        // [wave][tile]: prefix counts of the tile's 16 rows, one byte each
    __shared__ unsigned s_mw[DECODE ? 4 * 8 * 64 : 1];                                // [wave][word q][tile]: (row 2q) << 16 | row 2q+1
    static_assert(!(DEEP && (IDX32 || MARK)), "the shallow variant's options");
    __shared__ int s_head[MARK ? 4 * 64 : 1];                                         // [wave][entry of the trip]: a tile starts here
    int *const my_head = s_head + (MARK ? (threadIdx.x >> 6) * 64 : 0);
    uint4 *const my_rp = s_rp + (DECODE ? (threadIdx.x >> 6) * 64 : 0);
    unsigned *const my_mw = s_mw + (DECODE ? (threadIdx.x >> 6) * 8 * 64 : 0);
    // Work is dealt by ENTRIES, S3_EPW per wave, so hub rows (tiles with many entries and pairs) cannot pile
    // up in one wave.  The wave starts at the tile its first entry lies in (noted by step 2d; a 64-ary search over
    // the tile offsets -- three dependent gathers per wave -- before that), then walks the tiles 64 at a time:
    // their value offsets and pair ranges sit one per lane in registers, and the entry -> tile lookup is a 6-step
    // shuffle search with no memory traffic.
    const int lane = threadIdx.x & 63;
    // xcd != 0 (the grid is then a multiple of eight blocks): workgroups go to the eight XCDs round-robin, so XCD x takes the
    // x-th contiguous eighth of C -- consecutive entry ranges, which share their A and B tiles, then meet in ONE L2
    const unsigned vblock = xcd ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const long long wave = ((long long)vblock * blockDim.x + threadIdx.x) >> 6;
    // epw = entries per wave, a multiple of S3_CHUNK: 256 where C tiles are sparse (a wave's entries then span ~64 tiles, one
    // load of the per-tile registers); more where they are dense (25 entries per tile on the round-3 webbase-1M stand-in: the
    // 64 tiles a wave loads cover 1 600 entries, and at 256 entries per wave six waves would each load them)
    const long long eb = wave * epw;
    if (eb >= nnz_c) return;
    const int e_lo = (int)eb, e_hi = (int)(eb + epw < nnz_c ? eb + epw : nnz_c);
    const long long lo = chunk_tile[wave * (epw / S3_CHUNK)];   // the tile entry e_lo lies in (noted by step 2)
    for (long long t0 = lo; t0 < ntc; t0 += 64) {
        const long long tl = t0 + lane < ntc ? t0 + lane : ntc - 1;
        const int my_off = (t0 + lane < ntc) ? s3_ld<IDX32>(c_tile_nnz_ptr, tl) : 0x7FFFFFFF;   // value offset of tile t0+lane
        const int my_p0 = s3_ld<IDX32>(pairs_offset, tl), my_p1 = s3_ld<IDX32>(pairs_offset, tl + 1);
        // the tile's FIRST pair and its operands' value offsets, one gather set per tile: 92 % of webbase-1M's C tiles have
        // one pair, so most entries get their whole pair record by shuffle instead of four loads of their own (the step is
        // bound by the number of vector-memory instructions, section 4 of DESIGN.md)
        const int my_a0 = s3_ld<IDX32>(pairs_a, my_p0), my_b0 = s3_ld<IDX32>(pairs_b, my_p0);
        const int my_av0 = s3_ld<IDX32>(a_nnz_ptr, my_a0), my_bv0 = s3_ld<IDX32>(b_nnz_ptr, my_b0);
        // ... and the second pair of the tiles that have one (7 %): their entries' second trip then costs 4 instructions, not 8
        const bool two = my_p1 - my_p0 >= 2;
        const int my_a1 = two ? s3_ld<IDX32>(pairs_a, (long long)my_p0 + 1) : 0, my_b1 = two ? s3_ld<IDX32>(pairs_b, (long long)my_p0 + 1) : 0;
        const int my_av1 = two ? s3_ld<IDX32>(a_nnz_ptr, my_a1) : 0, my_bv1 = two ? s3_ld<IDX32>(b_nnz_ptr, my_b1) : 0;
        const long long tend = t0 + 64 < ntc ? t0 + 64 : ntc;
        const int chunk_end = c_tile_nnz_ptr[tend];
        const int first = __shfl(my_off, 0, 64);
        if (first >= e_hi) break;
        if constexpr (DECODE) {
            const uint4 M0 = *reinterpret_cast<const uint4 *>(c_mask + 8 * tl), M1 = *reinterpret_cast<const uint4 *>(c_mask + 8 * tl + 4);
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
            // (a full tile's last prefix is 240: everything fits a byte)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the previous 64 tiles' entries have been read
            __builtin_amdgcn_wave_barrier();
            my_rp[lane] = make_uint4(rp[0], rp[1], rp[2], rp[3]);
#pragma unroll
            for (int q = 0; q < 8; ++q) my_mw[q * 64 + lane] = w[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const int e_begin = first > e_lo ? first : e_lo, e_end = chunk_end < e_hi ? chunk_end : e_hi;
    for (int ebase = e_begin; ebase < e_end; ebase += 64) {   // wave-uniform trip count: every lane stays live for the shuffles
        const int e = ebase + lane;
        const bool valid = e < e_end;
        const int nvalid = e_end - ebase < 64 ? e_end - ebase : 64;   // (wave-uniform)
        // tile of entry e: largest lane index ti with off[ti] <= e (offsets are non-decreasing)
        int ti = 0;
        if constexpr (MARK) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the previous trip's marks have been read
            __builtin_amdgcn_wave_barrier();
            my_head[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned st = (unsigned)(my_off - ebase);             // (a tile that started before the trip, or a lane past the last tile: out of range)
            if (st < 64u) my_head[st] = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long marks = __ballot(my_head[lane] != 0);
            const int before = __popcll(__ballot(my_off < ebase));      // tiles that started before the trip (>= 1 unless one starts at its first entry)
            ti = before - 1 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(marks >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)marks, 0u)) +
                 (int)((marks >> lane) & 1ull);
        } else {
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                int probe = __shfl(my_off, ti + step, 64);
                if (probe <= e) ti += step;
            }
        }
        // The shallow kernel keeps every lane to the end of the trip (its third and later pairs are fetched by the wave together, below):
        // a lane past the last entry takes no pairs and stores nothing.
        const int p0 = __shfl(my_p0, ti, 64), p1_ = __shfl(my_p1, ti, 64);
        const int p1 = (DEEP || valid) ? p1_ : p0;
        const int a0 = __shfl(my_a0, ti, 64), b0 = __shfl(my_b0, ti, 64), av0 = __shfl(my_av0, ti, 64), bv0 = __shfl(my_bv0, ti, 64);
        // (every shuffle sits in front of the `continue`: a lane that has left cannot be read from)
        const int a1 = __shfl(my_a1, ti, 64), b1 = __shfl(my_b1, ti, 64), av1 = __shfl(my_av1, ti, 64), bv1 = __shfl(my_bv1, ti, 64);
        const int toff = DECODE ? __shfl(my_off, ti, 64) : 0;
        if constexpr (DEEP) {
            if (!valid) continue;
            if (BAND && p1 - p0 >= S3_BAND_MIN) continue;   // many-pair tiles: s3_band_kernel's
        }
        int r, c;
        if constexpr (DECODE) {
            const unsigned n = (unsigned)(e - toff);                     // entry n of its tile, row-major
            const uint4 rp = my_rp[ti];
            // largest row r with prefix[r] <= n (the prefixes never decrease, so rows without entries are stepped over)
            const bool h8 = (rp.z & 0xFFu) <= n;
            const unsigned d0 = h8 ? rp.z : rp.x, d1 = h8 ? rp.w : rp.y;
            const bool h4 = (d1 & 0xFFu) <= n;
            const unsigned d = h4 ? d1 : d0;
            const bool h2 = ((d >> 16) & 0xFFu) <= n;
            const unsigned hh = h2 ? d >> 16 : d & 0xFFFFu;
            const bool h1 = (hh >> 8) <= n;
            r = (h8 ? 8 : 0) + (h4 ? 4 : 0) + (h2 ? 2 : 0) + (h1 ? 1 : 0);
            unsigned k = n - (h1 ? hh >> 8 : hh & 0xFFu);               // ... and the k-th entry of that row
            const unsigned word = my_mw[(r >> 1) * 64 + ti];
            unsigned m = (r & 1) ? word & 0xFFFFu : word >> 16;
            unsigned t8 = __popc(m & 0xFFu);
            const bool g8 = k >= t8;
            k -= g8 ? t8 : 0u;
            m = g8 ? m >> 8 : m;
            unsigned t4 = __popc(m & 0xFu);
            const bool g4 = k >= t4;
            k -= g4 ? t4 : 0u;
            m = g4 ? m >> 4 : m;
            unsigned t2 = __popc(m & 3u);
            const bool g2 = k >= t2;
            k -= g2 ? t2 : 0u;
            m = g2 ? m >> 2 : m;
            const bool g1 = k >= (m & 1u);
            c = (g8 ? 8 : 0) + (g4 ? 4 : 0) + (g2 ? 2 : 0) + (g1 ? 1 : 0);
        } else {
            const unsigned rc = valid ? c_rowcolidx[e] : 0u;
            r = rc >> 4;
            c = rc & 15;
        }
        VT acc = VT(0);
        int p = p0;
        if (!DEEP) {   // first pair: everything but the two records and the values is already here
            // The step is bound by the LATENCY of its dependent gathers (records -> values), eight waves per SIMD being all the
            // hardware holds: halving the resident waves took it from 0.51 to 0.93 ms.  So the second pair's
            // records go out beside the first pair's, and the first product of either pair is gathered before anything is summed:
            // two round trips for an entry with two pairs instead of four.  The sums still run pair by pair, bit by bit.
            const bool has1 = p0 < p1, has2 = p0 + 1 < p1;
            // the third and later pairs of the trip's tiles (64 % of webbase-1M's entries lie in tiles with three to eight pairs): the
            // pairs of consecutive tiles are consecutive, so the wave fetches the trip's whole pair range ONCE, one pair per lane --
            // ids, then value offsets -- and a lane reads the record of ITS q-th pair by shuffle.  Before: every lane gathered ids,
            // offsets and records per pair (six wave-wide vector-memory instructions and four dependent round trips per pair; the
            // step is bound by the texture addresser, 16 cycles per such instruction -- DESIGN.md section 4); now two per pair.
            const int P0 = __builtin_amdgcn_readfirstlane(p0);
            const int Pn = __builtin_amdgcn_readlane(p1, nvalid - 1) - P0;       // (the trip's lanes are a prefix: nvalid of them)
            const bool staged = __ballot(p1 - p0 > 2) != 0ull && Pn <= 64;
            int sa = 0, sb = 0, sav = 0, sbv = 0;
            if (staged && lane < Pn) {
                sa = s3_ld<IDX32>(pairs_a, (long long)P0 + lane);
                sb = s3_ld<IDX32>(pairs_b, (long long)P0 + lane);
            }
            unsigned aw = 0, bw = 0;
            if (has1) {
                aw = s3_ld<IDX32>(a_rec, 16ll * a0 + r);
                bw = s3_ld<IDX32>(b_rec_t, 16ll * b0 + c);
            }
            unsigned aw1 = 0, bw1 = 0;
            if (has2) {
                aw1 = s3_ld<IDX32>(a_rec, 16ll * a1 + r);
                bw1 = s3_ld<IDX32>(b_rec_t, 16ll * b1 + c);
            }
            if (staged && lane < Pn) {
                sav = s3_ld<IDX32>(a_nnz_ptr, sa);
                sbv = s3_ld<IDX32>(b_nnz_ptr, sb);
            }
            const unsigned am = aw & 0xFFFFu, bm = bw & 0xFFFFu, am1 = aw1 & 0xFFFFu, bm1 = bw1 & 0xFFFFu;
            unsigned m = am & bm, m1 = am1 & bm1;
            const int ao = av0 + (int)(aw >> 16), bo = bv0 + (int)(bw >> 16), ao1 = av1 + (int)(aw1 >> 16), bo1 = bv1 + (int)(bw1 >> 16);
            VT x0 = VT(0), y0 = VT(0), x1 = VT(0), y1 = VT(0);
            if (m) {
                const unsigned below = (1u << __builtin_ctz(m)) - 1u;
                x0 = s3_ld<IDX32>(a_vals, (long long)ao + __popc(am & below));
                y0 = s3_ld<IDX32>(b_vals_t, (long long)bo + __popc(bm & below));
            }
            if (m1) {
                const unsigned below = (1u << __builtin_ctz(m1)) - 1u;
                x1 = s3_ld<IDX32>(a_vals, (long long)ao1 + __popc(am1 & below));
                y1 = s3_ld<IDX32>(b_vals_t, (long long)bo1 + __popc(bm1 & below));
            }
            if (m) {
                acc = pem_fma(x0, y0, acc);
                m &= m - 1;
                while (m) {
                    const int kk = __builtin_ctz(m);
                    m &= m - 1;
                    const unsigned below = (1u << kk) - 1u;
                    acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)ao + __popc(am & below)), s3_ld<IDX32>(b_vals_t, (long long)bo + __popc(bm & below)), acc);
                }
            }
            if (m1) {
                acc = pem_fma(x1, y1, acc);
                m1 &= m1 - 1;
                while (m1) {
                    const int kk = __builtin_ctz(m1);
                    m1 &= m1 - 1;
                    const unsigned below = (1u << kk) - 1u;
                    acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)ao1 + __popc(am1 & below)), s3_ld<IDX32>(b_vals_t, (long long)bo1 + __popc(bm1 & below)), acc);
                }
            }
            p += has2 ? 2 : has1 ? 1 : 0;
            if (staged) {
                for (int q = 2;; ++q) {
                    const bool on = p0 + q < p1;
                    if (__ballot(on) == 0ull) break;                            // wave-uniform
                    const int src = on ? p0 + q - P0 : 0;
                    const int a = __shfl(sa, src, 64), b = __shfl(sb, src, 64), av = __shfl(sav, src, 64), bv = __shfl(sbv, src, 64);
                    if (on) {
                        const unsigned awq = s3_ld<IDX32>(a_rec, 16ll * a + r), bwq = s3_ld<IDX32>(b_rec_t, 16ll * b + c);
                        const unsigned amq = awq & 0xFFFFu, bmq = bwq & 0xFFFFu;
                        unsigned mq = amq & bmq;
                        const int aoq = av + (int)(awq >> 16), boq = bv + (int)(bwq >> 16);
                        while (mq) {
                            const int kk = __builtin_ctz(mq);
                            mq &= mq - 1;
                            const unsigned below = (1u << kk) - 1u;
                            acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)aoq + __popc(amq & below)), s3_ld<IDX32>(b_vals_t, (long long)boq + __popc(bmq & below)), acc);
                        }
                    }
                }
                p = p1;
            }
        }
        // DEEP: plans averaging two or more pairs per C tile (3.1 on cage15-class inputs, 30+ where a band multiplies itself).
        // Not for everyone: webbase-1M's tiles hold 1.08 pairs and the extra code costs it 10 % (3 % when guarded by a wave vote,
        // which in turn loses cage15's gain).
        if constexpr (DEEP) {
            // Four pairs per trip, their eight record gathers in flight together; and one trip ahead, the NEXT four pairs' ids and
            // value offsets: the step is bound by the latency of its dependent gathers (ids -> records / value offsets -> values;
            // one vector-memory instruction per ~17 cycles and CU on a cage15 slice), and this takes two of the four round trips
            // off a trip's chain (cage15 slice: 14.6 -> 13.0 ms).  Going further -- records a trip ahead too, all loads
            // unconditional so that the in-order memory counter can leave the younger ones in flight -- costs registers
            // (88-134 VGPRs, 3-5 waves per SIMD) and loses: 13.9-15.7 ms.  Pairs are still added in ascending order.
            int a4[4], b4[4], av4[4], bv4[4];
            if (p + 4 <= p1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a4[k] = pairs_a[p + k];
                    b4[k] = pairs_b[p + k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    av4[k] = a_nnz_ptr[a4[k]];
                    bv4[k] = b_nnz_ptr[b4[k]];
                }
            }
            for (; p + 4 <= p1; p += 4) {
                unsigned aw4[4], bw4[4];
                int na4[4], nb4[4], nav4[4], nbv4[4];
                const bool more = p + 8 <= p1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    na4[k] = more ? pairs_a[p + 4 + k] : 0;
                    nb4[k] = more ? pairs_b[p + 4 + k] : 0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    aw4[k] = a_rec[16 * (size_t)a4[k] + r];
                    bw4[k] = b_rec_t[16 * (size_t)b4[k] + c];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nav4[k] = more ? a_nnz_ptr[na4[k]] : 0;
                    nbv4[k] = more ? b_nnz_ptr[nb4[k]] : 0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned am = aw4[k] & 0xFFFFu, bm = bw4[k] & 0xFFFFu;
                    unsigned m = am & bm;
                    if (!m) continue;
                    const VT *av = a_vals + av4[k] + (aw4[k] >> 16);
                    const VT *bv = b_vals_t + bv4[k] + (bw4[k] >> 16);
                    while (m) {
                        const int kk = __builtin_ctz(m);
                        m &= m - 1;
                        const unsigned below = (1u << kk) - 1u;
                        acc = pem_fma(av[__popc(am & below)], bv[__popc(bm & below)], acc);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a4[k] = na4[k];
                    b4[k] = nb4[k];
                    av4[k] = nav4[k];
                    bv4[k] = nbv4[k];
                }
            }
        }
        for (; p < p1; ++p) {
            const int a = s3_ld<IDX32>(pairs_a, p), b = s3_ld<IDX32>(pairs_b, p);
            const unsigned aw = s3_ld<IDX32>(a_rec, 16ll * a + r);
            const unsigned am = aw & 0xFFFFu;
            // B is read by column here: its transposed record (rows holding column c | entries in the columns before)
            // and its column-major values give every operand with one record gather + one value gather per product
            const unsigned bw = s3_ld<IDX32>(b_rec_t, 16ll * b + c);
            const unsigned bm = bw & 0xFFFFu;
            unsigned m = am & bm;
            if (!m) continue;
            const int ao = s3_ld<IDX32>(a_nnz_ptr, a) + (int)(aw >> 16), bo = s3_ld<IDX32>(b_nnz_ptr, b) + (int)(bw >> 16);
            while (m) {
                const int kk = __builtin_ctz(m);
                m &= m - 1;
                const unsigned below = (1u << kk) - 1u;
                acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)ao + __popc(am & below)), s3_ld<IDX32>(b_vals_t, (long long)bo + __popc(bm & below)), acc);
            }
        }
        // (streamed past the L2 -- nontemporal: C's values are never read here, and the records and values the gathers come back
        // to stay resident: 0.51 -> 0.47 ms on the webbase-1M stand-in)
        if (!valid) continue;
        if constexpr (IDX32)
            __builtin_nontemporal_store(acc, reinterpret_cast<VT *>(reinterpret_cast<char *>(c_vals) + (size_t)((unsigned)e * (unsigned)sizeof(VT))));
        else
            __builtin_nontemporal_store(acc, c_vals + e);
    }
        if (chunk_end >= e_hi) break;
    }
}

// ------------------------------------------------------------------------------------------
// Step 3 for C tiles with many pairs (deep plans: where a band multiplies itself a C tile holds ~40 entries and ~35 pairs).
// In the entry-per-lane kernel every lane of such a tile walks the SAME pair list and gathers the same two 64-byte records
// per pair; that kernel is bound by vector-memory issue.  Here ONE WAVE takes one tile at a time: the pairs' ids and value
// offsets are loaded once (one lane per pair), their records go to LDS transposed -- recA[row][pair], recB[col][pair] --
// with half a load instruction per pair, and every lane (= one C entry) scans its row of A words against its column of B
// words FOUR pairs per 16-byte LDS read.  Where the masks meet, the product waits in a per-lane queue (value offsets + the
// two masks); when a queue fills, all lanes gather their operands together.  Products are queued and summed in ascending
// pair order, so the fma chain -- and every bit of C -- equals the entry-per-lane kernel's.
// Grid: one wave per 64 consecutive C tiles; the wave finds the many-pair tiles among them by ballot.
// ------------------------------------------------------------------------------------------
template <typename VT>
__global__ void __launch_bounds__(256) s3_band_kernel(const int *__restrict__ pairs_offset, const int *__restrict__ pairs_a,
                                                      const int *__restrict__ pairs_b, long long ntc, const int *__restrict__ c_tile_nnz_ptr,
                                                      const uint8_t *__restrict__ c_rowcolidx, VT *__restrict__ c_vals,
                                                      const int *__restrict__ a_nnz_ptr, const VT *__restrict__ a_vals,
                                                      const unsigned *__restrict__ a_rec, const int *__restrict__ b_nnz_ptr,
                                                      const VT *__restrict__ b_vals_t, const unsigned *__restrict__ b_rec_t)
{
    __shared__ __attribute__((aligned(16))) unsigned s_rec[4][2 * 16 * S3_BAND_RS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Workgroups go to the eight XCDs round-robin; XCD x takes the x-th contiguous eighth of the C tiles (the grid is a multiple of
    // eight blocks), so that each L2 holds the A and B records of ITS stretch of the band instead of all eight holding the same
    // (too large) one: L2 hit rate 31 % -> see DESIGN.md
    const unsigned vblock = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long long t = ((long long)vblock * 4 + wv) * 64 + lane;
    int my_off = 0, my_off1 = 0, my_p0 = 0, my_p1 = 0;
    if (t < ntc) {
        my_off = c_tile_nnz_ptr[t];
        my_off1 = c_tile_nnz_ptr[t + 1];
        my_p0 = pairs_offset[t];
        my_p1 = pairs_offset[t + 1];
    }
    unsigned long long big = __ballot(my_p1 - my_p0 >= S3_BAND_MIN);
    unsigned *recA = s_rec[wv], *recB = recA + 16 * S3_BAND_RS;
    const int w = lane & 31, half = lane >> 5;
    while (big) {                                                   // wave-uniform: one trip per many-pair tile
        const int L = __builtin_ctzll(big);
        big &= big - 1;
        const int e0 = __builtin_amdgcn_readlane(my_off, L), n = __builtin_amdgcn_readlane(my_off1, L) - e0;
        const int pb = __builtin_amdgcn_readlane(my_p0, L), np = __builtin_amdgcn_readlane(my_p1, L) - pb;
        for (int sub = 0; sub < n; sub += 64) {                     // a tile holds up to 256 entries: 64 per trip
            const bool mine = sub + lane < n;
            const int e = e0 + sub + lane;
            unsigned src = 0;
            if (mine) src = c_rowcolidx[e];
            const unsigned *rowA = recA + (src >> 4) * S3_BAND_RS, *colB = recB + (src & 15) * S3_BAND_RS;
            VT acc = VT(0);
            for (int pcs = 0; pcs < np; pcs += S3_BAND_CH) {        // S3_BAND_CH pairs per stage: lane k holds pair pcs + k
                const int M = np - pcs < S3_BAND_CH ? np - pcs : S3_BAND_CH, M4 = (M + 3) & ~3;
                int ia = 0, ib = 0, oa = 0, ob = 0;
                if (lane < M) {
                    ia = pairs_a[pb + pcs + lane];
                    ib = pairs_b[pb + pcs + lane];
                    oa = a_nnz_ptr[ia];
                    ob = b_nnz_ptr[ib];
                }
                // the stage's records: word w of pair k is A's row word (w < 16) or B's column word; lanes 0-31 take the even pairs,
                // 32-63 the odd ones.  All the loads first, then the LDS writes: one round trip per stage.
                unsigned v[S3_BAND_CH / 2];
#pragma unroll
                for (int i = 0; i < S3_BAND_CH / 2; ++i) {
                    if (2 * i >= M4) break;                         // (wave-uniform)
                    const int k = 2 * i + half;
                    const int ka = __shfl(ia, k, 64), kb = __shfl(ib, k, 64);
                    v[i] = 0;                                       // (pairs M .. M4-1 pad the last group of four with empty masks)
                    if (k < M) v[i] = w < 16 ? a_rec[16 * (size_t)ka + w] : b_rec_t[16 * (size_t)kb + (w - 16)];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous stage's reads are done before the records change
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < S3_BAND_CH / 2; ++i) {
                    if (2 * i >= M4) break;
                    recA[w * S3_BAND_RS + 2 * i + half] = v[i];     // (w >= 16 lands in recB: the arrays are adjacent)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // scan: which of the stage's pairs meet in this entry -- LDS and VALU only
                unsigned long long hits = 0;
                for (int k4 = 0; k4 < M4; k4 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4 *>(rowA + k4), b4 = *reinterpret_cast<const uint4 *>(colB + k4);
                    const unsigned nib = ((a4.x & b4.x & 0xFFFFu) ? 1u : 0u) | ((a4.y & b4.y & 0xFFFFu) ? 2u : 0u) |
                                         ((a4.z & b4.z & 0xFFFFu) ? 4u : 0u) | ((a4.w & b4.w & 0xFFFFu) ? 8u : 0u);
                    hits |= (unsigned long long)nib << k4;
                }
                if (!mine) hits = 0;
                // sum: S3_BAND_H meeting pairs per lane and trip, lowest pair first; the operand gathers of all lanes share instructions
                while (__ballot(hits != 0)) {
                    unsigned am[S3_BAND_H], bm[S3_BAND_H];
                    const VT *av[S3_BAND_H], *bv[S3_BAND_H];
                    VT va[S3_BAND_H], vb[S3_BAND_H];
#pragma unroll
                    for (int h = 0; h < S3_BAND_H; ++h) {
                        const bool on = hits != 0;
                        const int k = on ? __builtin_ctzll(hits) : 0;
                        hits &= hits - 1;                           // (0 stays 0)
                        const unsigned aw = rowA[k], bw = colB[k];
                        const int kav = __shfl(oa, k, 64), kbv = __shfl(ob, k, 64);
                        am[h] = on ? aw & 0xFFFFu : 0u;
                        bm[h] = on ? bw & 0xFFFFu : 0u;
                        av[h] = a_vals + kav + (aw >> 16);
                        bv[h] = b_vals_t + kbv + (bw >> 16);
                        const unsigned mm = am[h] & bm[h];
                        const unsigned below = mm ? (1u << __builtin_ctz(mm)) - 1u : 0u;
                        va[h] = mm ? av[h][__popc(am[h] & below)] : a_vals[0];
                        vb[h] = mm ? bv[h][__popc(bm[h] & below)] : b_vals_t[0];
                    }
#pragma unroll
                    for (int h = 0; h < S3_BAND_H; ++h) {
                        unsigned mm = am[h] & bm[h];
                        if (mm) {
                            acc = pem_fma(va[h], vb[h], acc);
                            mm &= mm - 1;
                            while (mm) {
                                const int kk = __builtin_ctz(mm);
                                mm &= mm - 1;
                                const unsigned below = (1u << kk) - 1u;
                                acc = pem_fma(av[h][__popc(am[h] & below)], bv[h][__popc(bm[h] & below)], acc);
                            }
                        }
                    }
                }
            }
            if (mine) __builtin_nontemporal_store(acc, c_vals + e);   // (never re-read here: keep the L2 for the operand records and values)
        }
    }
}

pem_status pem::step3_impl(pem_ctx *ctx, pem_cplan *p)
{
    if (p->state < 2) {
        set_error("pem_spgemm_step3 called before step 2");
        return PEM_E_STATE;
    }
    const pem_tiled *A = p->A, *B = p->B;
    hipStream_t st = ctx->stream;
    const size_t ntc = (size_t)p->ntiles_c;
    if (!ctx->chain_events) PEM_HIP(hipEventRecord(ctx->ev[4], st));
    const bool wide = p->wide;   // step 2's choice: the entry-per-lane kernel needs the chunk index the fused path wrote
    const bool f32 = A->value_bytes == 4;
#define PEM_S3_WIDE(VT, DEEP, NAME)                                                                                                            \
    PEM_LAUNCH_NAMED(ctx, NAME, (s3_accumulate_wide_kernel<VT, DEEP>), (grid_for(((size_t)p->nnz_c + s3_epw - 1) / s3_epw * 64, 256) + 7u) & ~7u, 256,     \
                     p->pairs_offset.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)ntc, p->c_tile_nnz_ptr.as<int>(),       \
                     (long long)p->nnz_c, p->c_rowcolidx.as<uint8_t>(), p->c_vals.as<VT>(), A->tile_nnz_ptr.as<int>(), A->vals.as<VT>(),       \
                     A->tile_rec.as<uint32_t>(), B->tile_nnz_ptr.as<int>(), B->vals_t.as<VT>(), B->tile_rec_t.as<uint32_t>(),                  \
                     p->s3_chunk_tile.as<int>(), p->c_mask.as<uint32_t>(), (int)s3_epw, s3_xcd)
#define PEM_S3_DECODE(VT, I32, MK, NAME)                                                                                                       \
    PEM_LAUNCH_NAMED(ctx, NAME, (s3_accumulate_wide_kernel<VT, false, false, true, I32, MK>), (grid_for(((size_t)p->nnz_c + s3_epw - 1) / s3_epw * 64, 256) + 7u) & ~7u, 256, \
                     p->pairs_offset.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)ntc, p->c_tile_nnz_ptr.as<int>(),       \
                     (long long)p->nnz_c, (const uint8_t *)nullptr, p->c_vals.as<VT>(), A->tile_nnz_ptr.as<int>(), A->vals.as<VT>(),           \
                     A->tile_rec.as<uint32_t>(), B->tile_nnz_ptr.as<int>(), B->vals_t.as<VT>(), B->tile_rec_t.as<uint32_t>(),                  \
                     p->s3_chunk_tile.as<int>(), p->c_mask.as<uint32_t>(), (int)s3_epw, s3_xcd)
#define PEM_S3_WIDE3(VT, NAME)                                                                                                                 \
    PEM_LAUNCH_NAMED(ctx, NAME, (s3_accumulate_wide_kernel<VT, true, true>), (grid_for(((size_t)p->nnz_c + s3_epw - 1) / s3_epw * 64, 256) + 7u) & ~7u, 256, \
                     p->pairs_offset.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)ntc, p->c_tile_nnz_ptr.as<int>(),       \
                     (long long)p->nnz_c, p->c_rowcolidx.as<uint8_t>(), p->c_vals.as<VT>(), A->tile_nnz_ptr.as<int>(), A->vals.as<VT>(),       \
                     A->tile_rec.as<uint32_t>(), B->tile_nnz_ptr.as<int>(), B->vals_t.as<VT>(), B->tile_rec_t.as<uint32_t>(),                  \
                     p->s3_chunk_tile.as<int>(), p->c_mask.as<uint32_t>(), (int)s3_epw, s3_xcd)
#define PEM_S3_LAUNCH(VT)                                                                                                                      \
    do {                                                                                                                                       \
        if (wide && deep && p->opt_band) {                                                                                                     \
            PEM_S3_WIDE3(VT, "s3_accumulate_wide_kernel<" #VT ",deep,band>");                                                                  \
            PEM_LAUNCH_NAMED(ctx, "s3_band_kernel<" #VT ">", (s3_band_kernel<VT>), (grid_for(ntc, 256) + 7u) & ~7u, 256, p->pairs_offset.as<int>(),         \
                             p->pairs_a.as<int>(), p->pairs_b.as<int>(), (long long)ntc, p->c_tile_nnz_ptr.as<int>(),                          \
                             p->c_rowcolidx.as<uint8_t>(), p->c_vals.as<VT>(), A->tile_nnz_ptr.as<int>(), A->vals.as<VT>(),                    \
                             A->tile_rec.as<uint32_t>(), B->tile_nnz_ptr.as<int>(), B->vals_t.as<VT>(), B->tile_rec_t.as<uint32_t>());         \
        } else if (wide && deep)                                                                                                               \
            PEM_S3_WIDE(VT, true, "s3_accumulate_wide_kernel<" #VT ",deep>");                                                                  \
        else if (wide && p->s3_decode && idx32 && mark)                                                                                        \
            PEM_S3_DECODE(VT, true, true, "s3_accumulate_wide_kernel<" #VT ",decode,idx32,mark>");                                             \
        else if (wide && p->s3_decode && idx32)                                                                                                \
            PEM_S3_DECODE(VT, true, false, "s3_accumulate_wide_kernel<" #VT ",decode,idx32>");                                                 \
        else if (wide && p->s3_decode && mark)                                                                                                 \
            PEM_S3_DECODE(VT, false, true, "s3_accumulate_wide_kernel<" #VT ",decode,mark>");                                                  \
        else if (wide && p->s3_decode)                                                                                                         \
            PEM_S3_DECODE(VT, false, false, "s3_accumulate_wide_kernel<" #VT ",decode>");                                                      \
        else if (wide)                                                                                                                         \
            PEM_S3_WIDE(VT, false, "s3_accumulate_wide_kernel<" #VT ">");                                                                      \
        else                                                                                                                                   \
            PEM_LAUNCH(ctx, s3_accumulate_kernel<VT>, grid_for(ntc * 16, 256), 256, p->pairs_offset.as<int>(), p->pairs_a.as<int>(),           \
                       p->pairs_b.as<int>(), (long long)ntc, p->c_tile_nnz_ptr.as<int>(), p->c_rowcolidx.as<uint8_t>(), p->c_vals.as<VT>(),    \
                       A->tile_nnz_ptr.as<int>(), A->vals.as<VT>(), A->masks.as<uint16_t>(), A->rowptr.as<uint8_t>(),                          \
                       B->tile_nnz_ptr.as<int>(), B->vals.as<VT>(), B->masks.as<uint16_t>(), B->rowptr.as<uint8_t>(), B->masks_t.as<uint16_t>()); \
    } while (0)
    const bool deep = p->npairs >= 2 * p->ntiles_c;   // two or more pairs per C tile on average (see the kernel)
    // entries per wave: 256, or 512 where the C tiles hold 8+ entries on average (see the kernel; 562 / 534 / 536 / 574 us at 256 / 512 /
    // 1024 / 2048 on the round-3 webbase-1M stand-in); PEM_OPT_S3_EPW forces 256 * value
    // ... and only where that still leaves several rounds of waves (8 waves x 4 SIMDs per CU): a 1/8 row block of webbase-1M is
    // 1.5 rounds at 512 entries per wave, and runs 10 % faster as three rounds of 256 (0.098 -> 0.088 ms)
    const size_t s3_slots = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 32;
    const bool s3_many = (size_t)p->nnz_c >= 4 * s3_slots * (2 * (size_t)S3_EPW);
    const size_t s3_epw = (size_t)S3_EPW * (size_t)(p->opt_epw > 0 ? p->opt_epw : (ntc > 0 && (size_t)p->nnz_c >= 8 * ntc && !deep && s3_many) ? 2 : 1);
    // 32-bit byte offsets on scalar bases where every array the shallow kernel touches is smaller than 4 GiB; the marked
    // entry -> tile lookup where no C tile is empty (pruned lists)
    const size_t gib4 = (size_t)1 << 32, vb = (size_t)A->value_bytes;
    const bool idx32 = !p->opt_idx64 && 32 * (ntc + 1) < gib4 && 4 * ((size_t)p->npairs + 4) < gib4 && 64 * ((size_t)A->ntiles + 1) < gib4 &&
                       64 * ((size_t)B->ntiles + 1) < gib4 && vb * ((size_t)A->nnz + 1) < gib4 && vb * ((size_t)B->nnz + 1) < gib4 &&
                       vb * ((size_t)p->nnz_c + 1) < gib4;
    const bool mark = p->opt_mark && p->opt_prune;
    const int s3_xcd = p->opt_s3_xcd;
    if (ntc > 0 && f32)
        PEM_S3_LAUNCH(float);
    else if (ntc > 0)
        PEM_S3_LAUNCH(double);
#undef PEM_S3_LAUNCH
#undef PEM_S3_WIDE
#undef PEM_S3_WIDE3
#undef PEM_S3_DECODE
    if (!ctx->capturing) PEM_HIP(hipEventRecord(ctx->ev[5], st));
    p->state = 3;
    return PEM_OK;
}
