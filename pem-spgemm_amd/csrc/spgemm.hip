// spgemm.hip -- rows a9-a14: the three-step tiled SpGEMM hot path and the C export.
//
// Reference: step 1 = SPA bitmask kernels run twice (count + emit) or the NSPARSE binned
// hash path (spgemm.cu:1141-1218); step 2a/2b = warp-per-C-tile list intersection by binary
// search, again run twice (spgemm.cu:387-497); step 2c/2d masks + intra-tile CSR
// (spgemm.cu:499-591); step 3 numeric with a global read-modify-write per product
// (spgemm.cu:593-661).
//
// Here: every (A tile (i,k), B tile (k,j)) with matching k is a *product* of tile row i keyed by j, so grouping
// a row's products by j yields the C tile list (step 1) and the pair lists in ascending k (step 2a/2b) in one
// pass -- no SPA, no hash tables, no binary searches, nothing computed twice.  Products whose tiles cannot
// meet (A tile's occupied columns miss B tile's occupied rows) are dropped up front (exact; PEM_PRUNE=0 keeps
// the reference's lists).  Default path: per-row LDS/register bitonic sorts, binned by row size; the global
// expand + radix-sort path ("esc") serves oversized rows and PEM_STEP1=esc.  Step 2c is the boolean row
// product (C row r = OR of B rows kk over kk in A row r), one C tile per lane; step 3 keeps one C entry per
// lane in a register, accumulates in ascending k with one fma per product, and stores once.  All outputs keep
// the reference layouts (include/pem_spgemm.h).  Kernel variants (A/B baselines kept for tests) are plan options
// (pem_cplan_set_option); the environment only sets a new plan's defaults:
//   PEM_STEP1=esc  PEM_WIDE=0  PEM_PRUNE=0  PEM_NO_WARM=1  PEM_EXPORT=rows
#include "pem_internal.h"
#include <algorithm>
#include <chrono>

using namespace pem;

// ------------------------------------------------------------------------------------------
// step 1
// ------------------------------------------------------------------------------------------
// per A tile (i,k): number of tiles in B's tile row k (= tile-level intermediate products;
// the quantity of spgemm_nsparse_kernel.h:135-151 per A tile instead of per row)
// A product (A tile (i,k), B tile (k,j)) can only contribute if some column occupied in the A tile is a row
// occupied in the B tile.  The reference's tile-level symbolic product keeps every product and so
// materialises pairs -- and whole C tiles -- that stay empty (83 % of the pairs of the scircuit stand-in).
// With prune != 0 those dead products are dropped here, before anything is sorted or stored: the final C
// is unchanged, only the intermediate C tile / pair lists lose their empty members.  prune == 0 reproduces
// the reference's lists exactly.  16 lanes per A tile: aprod = all products, lprod = live products.
__global__ void __launch_bounds__(256) s1_aprod_kernel(const int *__restrict__ a_tile_colidx, const uint32_t *__restrict__ a_occ, int a_lo,
                                                       int nA, const int *__restrict__ b_tile_rowptr, const uint32_t *__restrict__ b_occ,
                                                       int prune, int *__restrict__ aprod, int *__restrict__ lprod,
                                                       const long long *__restrict__ a_tile_keys, int tr_lo, int *__restrict__ row_n,
                                                       int *__restrict__ row_l)
{
    constexpr int G = 8;        // lanes per A tile (B tile rows average ~34 tiles; 16 lanes: 88 us, 8: 59 us, 4: 58 us)
    const int arel = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int l = threadIdx.x & (G - 1);
    const bool in = arel < nA;
    constexpr int LONG = 64 * G;   // a B tile row this long is walked by the whole wave, not by the tile's G lanes
    int len = 0, cnt = 0, b0 = 0;
    unsigned acol = 0;
    if (in) {
        const int k = a_tile_colidx[a_lo + arel];
        b0 = b_tile_rowptr[k];
        len = b_tile_rowptr[k + 1] - b0;
        if (prune) {
            acol = a_occ[a_lo + arel] & 0xFFFFu;
            if (len < LONG) {
#pragma unroll 4
                for (int q = l; q < len; q += G) cnt += (acol & (b_occ[b0 + q] >> 16)) != 0;
            }
        }
    }
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, G);
    if (prune) {
        // hub rows of B (4 700 tiles on webbase-1M): left to 8 lanes, one such A tile kept its wave busy for 590 trips and
        // the kernel waited for it (80 us, 60 of them this tail); the wave takes them together, 64 tiles per trip
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(in && l == 0 && len >= LONG);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int hb0 = __shfl(b0, src, 64), hlen = __shfl(len, src, 64);
            const unsigned hcol = (unsigned)__shfl((int)acol, src, 64);
            int c = 0;
#pragma unroll 4
            for (int q = lane; q < hlen; q += 64) c += (hcol & (b_occ[hb0 + q] >> 16)) != 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
            if ((lane & ~(G - 1)) == src) cnt = c;   // every lane of the tile's group holds its count
        }
    }
    if (!prune) cnt = len;
    if (in && l == 0) {
        aprod[arel] = len;
        lprod[arel] = cnt;
    }
    // Per tile-row totals (row-local step 1: the rows' product counts are all the scan that is left -- the offsets of
    // the A tiles inside a row are rebuilt in LDS by the row's own workgroup).  The wave's eight A tiles are
    // consecutive, so tiles of one row sit next to each other: the first of each run adds the run's sums, one atomic
    // pair per run (a hub row of 4 700 A tiles: 590 adds on its two counters instead of 4 700).
    if (row_n == nullptr) return;
    const int row = in ? (int)(a_tile_keys[a_lo + arel] >> 32) - tr_lo : -1 - (int)(threadIdx.x / G);   // distinct dummies never merge
    const int lane = threadIdx.x & 63;
    // suffix sums over the run, by doubling: tiles are sorted by row, so "the tile d further on has my row" implies the
    // ones in between have it too, and its partial sum only ever covers tiles of that same row
    int sum_n = len, sum_l = cnt;
#pragma unroll
    for (int d = 1; d < 64 / G; d <<= 1) {
        const int src = lane + d * G;
        const int orow = __shfl(row, src & 63, 64), on = __shfl(sum_n, src & 63, 64), ol = __shfl(sum_l, src & 63, 64);
        if (src < 64 && orow == row) {
            sum_n += on;
            sum_l += ol;
        }
    }
    const int prow = __shfl(row, (lane - G) & 63, 64);
    const bool head = in && l == 0 && (lane < G || prow != row);
    if (head) {
        atomicAdd(&row_n[row], sum_n);
        atomicAdd(&row_l[row], sum_l);
    }
}

// global expand (16 lanes per A tile walk B's tile row k): live products only, compacted by ballot;
// product x gets key (i - tr_lo, j).  xl_base == nullptr: every row (PEM_STEP1=esc), positions = global
// live offsets; else only the oversized rows (xl_base[i] >= 0), positions relative to the row's slot.
__global__ void __launch_bounds__(256) s1_xl_expand_kernel(const long long *__restrict__ a_tile_keys, const int *__restrict__ a_tile_rowptr,
                                                           const uint32_t *__restrict__ a_occ, int a_lo, int nA, int tr_lo,
                                                           const int *__restrict__ lprod_off, const int *__restrict__ xl_base,
                                                           const int *__restrict__ b_tile_rowptr, const int *__restrict__ b_tile_colidx,
                                                           const uint32_t *__restrict__ b_occ, int prune, int bits_tc,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, int *__restrict__ prod_a,
                                                           int *__restrict__ prod_b, int local_keys, const int *__restrict__ xl_rows)
{
    // xl_rows != nullptr: a two-dimensional grid over the oversized rows only -- blockIdx.y picks the row, blockIdx.x sixteen of
    // its A tiles (the grid covers the plan's longest tile row; blocks past a row's end leave at once) -- instead of one pass
    // over every A tile of the slice, of which all but the few oversized rows' exit after two loads
    int arel;
    bool in;
    if (xl_rows) {
        const int xi = xl_rows[blockIdx.y];
        const int r0 = a_tile_rowptr[tr_lo + xi] - a_lo, r1 = a_tile_rowptr[tr_lo + xi + 1] - a_lo;
        arel = r0 + (int)blockIdx.x * 16 + (int)(threadIdx.x >> 4);
        in = arel < r1;
    } else {
        arel = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
        in = arel < nA;
    }
    const int l = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
    int a = 0, i = 0, k = 0, x0 = -1, b0 = 0, len = 0;
    unsigned acol = 0xFFFFu;
    if (in) {
        a = a_lo + arel;
        const long long ak = a_tile_keys[a];
        i = (int)(ak >> 32) - tr_lo;
        k = (int)(ak & 0xFFFFFFFFll);
        if (xl_base) {          // lprod_off: live offsets relative to the row (s1_xl_rel_kernel), only valid in oversized rows
            const int base = xl_base[i];
            if (base >= 0) x0 = base + lprod_off[arel];
        } else {                // global live offsets (PEM_STEP1=esc)
            x0 = lprod_off[arel];
        }
        if (x0 >= 0) {
            b0 = b_tile_rowptr[k];
            len = b_tile_rowptr[k + 1] - b0;
            if (prune) acol = a_occ[a] & 0xFFFFu;
        }
    }
    // the four 16-lane groups of a wave walk different B rows: iterate to the longest, compact per group.  A B tile row of
    // 256+ tiles (a directory page of webbase-1M: 4 700) is left out of that walk -- sixteen lanes took 294 dependent trips
    // over it and the whole grid waited (121 us for 1.6 M products) -- and walked by the whole wave afterwards.
    constexpr int XL_LONG = 256;
    const bool is_long = len >= XL_LONG;
    const int glen = is_long ? 0 : len;
    int maxlen = glen;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(maxlen, d, 64);
        maxlen = o > maxlen ? o : maxlen;
    }
    const uint64_t hi = (uint64_t)(unsigned)i << bits_tc;
    auto emit_product = [&](const int x, const int row_hi_src, const int aa, const int bb) {
        const uint64_t hh = (uint64_t)(unsigned)row_hi_src << bits_tc;
        const unsigned col = (unsigned)b_tile_colidx[bb];
        keys[x] = local_keys ? ((uint64_t)col << 32) | (uint64_t)(unsigned)x : hh | (uint64_t)col;
        perm[x] = (uint32_t)x;
        prod_a[x] = aa;
        prod_b[x] = bb;
    };
    (void)hi;
    int run = 0;
    for (int q0 = 0; q0 < maxlen; q0 += 16) {
        const int q = q0 + l;
        const bool live = q < glen && (!prune || (acol & (b_occ[b0 + q] >> 16)) != 0);
        const unsigned m16 = (unsigned)(__ballot(live) >> (16 * grp)) & 0xFFFFu;
        // (keys: local_keys -> (tile column, position) for the per-row sort of s1_xl_rowsort_kernel -- a row's products already
        // sit in the row's own stretch of the buffers, in product order; else (row, tile column) for the global sort)
        if (live) emit_product(x0 + run + __popc(m16 & ((1u << l) - 1u)), i, a, b0 + q);
        run += __popc(m16);
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(l == 0 && is_long && x0 >= 0);
    while (todo) {                                              // wave-uniform
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int ha = __shfl(a, src, 64), hb0 = __shfl(b0, src, 64), hlen = __shfl(len, src, 64), hx0 = __shfl(x0, src, 64), hrow = __shfl(i, src, 64);
        const unsigned hcol = (unsigned)__shfl((int)acol, src, 64);
        int hrun = 0;
        // four trips' occupancy words (and then their tile columns) are requested together: one trip at a time the walk was a
        // chain of 74 dependent round trips for a directory row
        for (int q0 = 0; q0 < hlen; q0 += 256) {
            unsigned occ[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 64 * u + lane;
                occ[u] = (prune && q < hlen) ? b_occ[hb0 + q] : 0xFFFF0000u;
            }
            bool live[4];
            unsigned col[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 64 * u + lane;
                live[u] = q < hlen && (hcol & (occ[u] >> 16)) != 0;
                col[u] = live[u] ? (unsigned)b_tile_colidx[hb0 + q] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 64 * u + lane;
                const unsigned long long bal = __ballot(live[u]);
                if (live[u]) {
                    const int x = hx0 + hrun + __popcll(bal & lt);
                    keys[x] = local_keys ? ((uint64_t)col[u] << 32) | (uint64_t)(unsigned)x : ((uint64_t)(unsigned)hrow << bits_tc) | (uint64_t)col[u];
                    perm[x] = (uint32_t)x;
                    prod_a[x] = ha;
                    prod_b[x] = hb0 + q;
                }
                hrun += __popcll(bal);
            }
        }
    }
}

__global__ void s1_heads_kernel(const uint64_t *__restrict__ keys, size_t n, int *__restrict__ head)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    head[p] = (p == 0) || (keys[p] != keys[p - 1]);
}

// C tile list (spgemm.cu:374-381 output contract: ascending tile column inside a tile row)
// + pair offsets (spgemm.cu:483-484 + :1242): both read off the sorted product stream.
__global__ void s1_emit_ctiles_kernel(const uint64_t *__restrict__ keys, const int *__restrict__ headx, size_t n, int tr_lo, int bits_tc,
                                      int *__restrict__ c_rowidx, int *__restrict__ c_colidx, int *__restrict__ pairs_offset)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int t = headx[p];
    if (headx[p + 1] != t) {
        uint64_t k = keys[p];
        c_rowidx[t] = (int)(k >> bits_tc) + tr_lo;
        c_colidx[t] = (int)(k & ((1ull << bits_tc) - 1ull));
        pairs_offset[t] = (int)p;
    }
    if (p == n - 1) pairs_offset[headx[n]] = (int)n;
}

// _C_rowPtr (spgemm.cu:1166-1168) by boundary fill over the sorted C tile rows
__global__ void s1_c_rowptr_kernel(const int *__restrict__ c_rowidx, long long ntc, int tr_lo, int mt, int *__restrict__ c_rowptr)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntc) return;
    int tr = c_rowidx[t] - tr_lo;
    int prev = t > 0 ? c_rowidx[t - 1] - tr_lo : -1;
    for (int row = prev + 1; row <= tr; ++row) c_rowptr[row] = (int)t;
    if (t == ntc - 1)
        for (int row = tr + 1; row <= mt; ++row) c_rowptr[row] = (int)ntc;
}


// ------------------------------------------------------------------------------------------
// step 1, row-local form (default).  The products of one tile row of A only ever meet
// products of the same row, so the grouping by C tile is a per-row sort on the tile column:
// one workgroup expands the row's live products into LDS as (tile col, product index) keys,
// sorts them there and streams the sorted pair list out once -- no global sort passes.  Rows
// are binned by their LIVE product count: <=512 one wave and <=2048 four waves (bitonic network
// in registers), <=8192 and <=32768 sixteen waves (keys kept in product order + a stable LDS
// radix sort on the column bits); larger rows take the global expand/radix-sort path above.
// C tile columns and per-tile pair offsets go to row-local scratch (a row has at most as many
// C tiles as products) and are compacted into the reference layout once the per-row tile
// counts have been scanned.
// ------------------------------------------------------------------------------------------
constexpr int S1_CAP0 = 512, S1_CAP1 = 2048, S1_CAP2 = 8192, S1_CAP3 = 32768;
constexpr int S1_NCAP0 = 8 * S1_CAP0, S1_NCAP1 = 8 * S1_CAP1;   // ... and products before pruning, for the two small bins
constexpr int S1_RCAP0 = 256, S1_RCAP1 = 1024, S1_RCAP2 = 2048, S1_RCAP3 = 1024;   // A tiles per row a bin's LDS table holds
constexpr int S1_COARSE = 512;   // 64-product blocks indexed per row (covers the 32768 products a 15-bit index field allows)

__global__ void __launch_bounds__(256) s1_reset_kernel(int *__restrict__ flags, int *__restrict__ bin_count, long long *__restrict__ scalars,
                                                       int *__restrict__ pairs_offset, int *__restrict__ row_tc, int mt, int *__restrict__ row_n,
                                                       int *__restrict__ row_l, int *__restrict__ group_nnz, int ngroups)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NUM_FLAGS) flags[i] = 0;
    if (i < 8) bin_count[i] = 0;
    if (i < 4) scalars[i] = 0;
    if (i == 0) {
        pairs_offset[0] = 0;
        row_tc[mt] = 0;
    }
    if (i <= mt) {              // per-row product totals, accumulated by s1_aprod_kernel
        row_n[i] = 0;
        row_l[i] = 0;
    }
    if (i < ngroups) group_nnz[i] = 0;   // entry counts per S2_GROUP tiles, accumulated by s2_tiles_kernel (repeat passes: size known)
}

__global__ void __launch_bounds__(256) s1_rowclass_kernel(const int *__restrict__ a_tile_rowptr, int tr_lo, int mt,
                                                          const int *__restrict__ row_n, const int *__restrict__ row_lbase, int cap3,
                                                          int qcap, int xlcap, int rcap2, int qcap2, int ncap0, int ncap1, int *__restrict__ row_list,
                                                          int *__restrict__ bin_count, int *__restrict__ xl_base, int *__restrict__ row_tc,
                                                          long long *__restrict__ scalars)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int n = 0, nl = 0, R = 0;
    if (i < mt) {
        R = a_tile_rowptr[tr_lo + i + 1] - a_tile_rowptr[tr_lo + i];
        n = row_n[i];
        nl = row_lbase[i + 1] - row_lbase[i];
        xl_base[i] = -1;
        row_tc[i] = 0;
    }
    // bins by LIVE products (what gets sorted); the key's index field must still hold every product of the row, and the
    // bin's LDS table every A tile of the row (a row with more A tiles moves up, or to the global path)
    int bin = nl == 0 ? -1 : (n > qcap || nl > xlcap) ? 4 : nl <= S1_CAP0 ? 0 : nl <= S1_CAP1 ? 1 : nl <= S1_CAP2 ? 2 : nl <= cap3 ? 3 : 4;
    if (bin == 0 && (R > S1_RCAP0 || n > ncap0)) bin = 1;
    if (bin == 1 && (R > S1_RCAP1 || n > ncap1)) bin = 2;
    if (bin == 2 && (R > rcap2 || n > qcap2)) bin = 4;
    if (bin == 3 && R > S1_RCAP3) bin = 4;
    // slots by ballot + prefix popcount inside a wave, one LDS atomic per wave and bin inside the block, ONE global
    // atomic per block and bin (order inside a bin is irrelevant): 4 k wave-level atomics on four counters serialised
    // for ~20 us of a 33 us kernel
    __shared__ int blk_cnt[5], blk_base[5];
    __shared__ long long blk_all;
    if (threadIdx.x < 5) blk_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) blk_all = 0;
    __syncthreads();
    int wbase = 0, rank = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const unsigned long long m = __ballot(bin == b);
        if (m == 0) continue;
        const int leader = __builtin_ctzll(m);
        int base = 0;
        if (lane == leader) base = atomicAdd(&blk_cnt[b], __popcll(m));
        base = __shfl(base, leader, 64);
        if (bin == b) {
            wbase = base;
            rank = __popcll(m & lt);
        }
    }
    {   // every tile-level product of the slice (the reference's P), 64-bit: one atomic per block
        long long wn = n;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) wn += __shfl_xor(wn, d, 64);
        if (lane == 0 && wn) atomicAdd(reinterpret_cast<unsigned long long *>(&blk_all), (unsigned long long)wn);
    }
    __syncthreads();
    if (threadIdx.x < 5) blk_base[threadIdx.x] = blk_cnt[threadIdx.x] ? atomicAdd(&bin_count[threadIdx.x], blk_cnt[threadIdx.x]) : 0;
    if (threadIdx.x == 0 && blk_all) atomicAdd(reinterpret_cast<unsigned long long *>(&scalars[3]), (unsigned long long)blk_all);
    __syncthreads();
    if (bin >= 0) row_list[(size_t)bin * mt + blk_base[bin] + wbase + rank] = i;
    if (bin == 4) {                                             // oversized rows are few
        xl_base[i] = atomicAdd(&bin_count[5], nl);
        atomicMax(&bin_count[6], nl);                           // the largest of them decides between the per-row and the global sort
    }
}

// oversized rows: live-product offsets of the row's A tiles relative to the row, one 1024-thread block per row (a directory
// row of webbase-1M has 4 700 A tiles: five trips; with 256 threads it took nineteen, 16 us on a chain that is the critical
// path of a rank's share of an 8-way split)
__global__ void __launch_bounds__(1024) s1_xl_rel_kernel(const int *__restrict__ xl_rows, int nrows_xl, const int *__restrict__ a_tile_rowptr, int tr_lo,
                                                         int a_lo, const int *__restrict__ lcnt, int *__restrict__ lrel)
{
    constexpr int WAVES = 16;
    __shared__ int wsum[WAVES];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int li = blockIdx.x; li < nrows_xl; li += gridDim.x) {
        const int i = xl_rows[li];
        const int a0 = a_tile_rowptr[tr_lo + i] - a_lo, a1 = a_tile_rowptr[tr_lo + i + 1] - a_lo;
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int x0 = a0; x0 < a1; x0 += 1024) {
            const int x = x0 + threadIdx.x;
            const int c = x < a1 ? lcnt[x] : 0;
            int inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int woff = carry_s, tot = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                if (w < wave) woff += wsum[w];
                tot += wsum[w];
            }
            if (x < a1) lrel[x] = woff + inc - c;
            __syncthreads();
            if (threadIdx.x == 0) carry_s += tot;
            __syncthreads();
        }
    }
}

// largest a in [lo, hi) with off[a] <= x
__device__ __forceinline__ int s1_find_a(const int *__restrict__ off, int lo, int hi, int x)
{
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// Bitonic sort of THREADS*EPT keys held EPT per thread in the blocked layout (element e = tid*EPT + m, see
// s1_bitonic_regs): the smallest strides are register-local, the next six wave shuffles, the rest through LDS.
// Ends with the keys in `lds`.
template <typename KeyT> __device__ __forceinline__ KeyT s1_shfl_xor(KeyT v, int mask);
template <> __device__ __forceinline__ uint32_t s1_shfl_xor<uint32_t>(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask, 64); }
template <> __device__ __forceinline__ uint64_t s1_shfl_xor<uint64_t>(uint64_t v, int mask)
{
    return (uint64_t)__shfl_xor((unsigned long long)v, mask, 64);
}

template <typename KeyT, int THREADS, int EPT, int LOGT>
__device__ __forceinline__ void s1_bitonic_regs(KeyT (&v)[EPT], KeyT *lds, const int tid, const bool reverse = false)
{
    // Blocked layout: element e = tid*EPT + m.  The log2(EPT) SMALLEST strides -- which every merge level runs --
    // are then exchanges between registers of one thread, the next six are wave shuffles (lane ^ jj/EPT) and only
    // strides >= 64*EPT cross waves through LDS (3 of the 66 stages at 2048 keys; the strided layout e = m*T + tid
    // made the LARGEST strides register-local, which only the last levels have, and needed 9 LDS + 39 shuffle stages).
    constexpr int LOGE = EPT == 1 ? 0 : EPT == 2 ? 1 : EPT == 4 ? 2 : EPT == 8 ? 3 : EPT == 16 ? 4 : 5;
    constexpr int LOGNP = LOGT + LOGE;
#pragma unroll
    for (int lk = 1; lk <= LOGNP; ++lk) {
        const int kk = 1 << lk;
#pragma unroll
        for (int lj = lk - 1; lj >= 0; --lj) {
            const int jj = 1 << lj;
            if (lj < LOGE) {             // partner in another register of this thread
#pragma unroll
                for (int m = 0; m < EPT; ++m) {
                    if ((m & jj) == 0) {
                        const int m2 = m | jj;
                        const bool up = (((tid << LOGE) | m) & kk) == 0;
                        const KeyT x = v[m], y = v[m2];
                        const bool sw = (x > y) == up;
                        v[m] = sw ? y : x;
                        v[m2] = sw ? x : y;
                    }
                }
            } else if (lj < LOGE + 6) {  // partner in another lane of this wave
                const int lm = jj >> LOGE;
                const bool lower = (tid & lm) == 0;
                // all EPT exchanges are issued before the first result is used: written as one loop the compiler
                // emitted ds_bpermute / s_waitcnt lgkmcnt(0) pairs, i.e. one full LDS latency per key and stage
                KeyT pv[EPT];
#pragma unroll
                for (int m = 0; m < EPT; ++m) pv[m] = s1_shfl_xor<KeyT>(v[m], lm);
#pragma unroll
                for (int m = 0; m < EPT; ++m) {   // compare + select (the lane predicate folds into the mask on the scalar unit)
                    const bool up = (((tid << LOGE) | m) & kk) == 0;
                    v[m] = ((v[m] < pv[m]) == (lower == up)) ? v[m] : pv[m];
                }
            } else {                     // partner in another wave: through LDS ([m][tid] image: conflict-free)
                const int tm = jj >> LOGE;
                const bool lower = (tid & tm) == 0;
#pragma unroll
                for (int m = 0; m < EPT; ++m) lds[m * THREADS + tid] = v[m];
                __syncthreads();
#pragma unroll
                for (int m = 0; m < EPT; ++m) {
                    const KeyT pv = lds[m * THREADS + (tid ^ tm)];
                    const bool up = (((tid << LOGE) | m) & kk) == 0;
                    v[m] = ((v[m] < pv) == (lower == up)) ? v[m] : pv;
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int m = 0; m < EPT; ++m) lds[reverse ? (THREADS * EPT - 1 - ((tid << LOGE) | m)) : ((tid << LOGE) | m)] = v[m];
    __syncthreads();
}

template <typename KeyT, int CAP, int QB, int THREADS, int RCAP, bool RANK = false>
struct S1Row {
    KeyT *keys;
    uint16_t *qmap;                // RANK: product index of the key at every live position (the key carries the position)
    const int *roff, *rbs;
    const unsigned *rco;           // occupied columns of every A tile of the row (pruning)
    const int *cstart;             // A tile holding product 64*c, for every 64th product (rows of up to 32768 products)
    bool coarse;
    // (the one-wave bin has few A tiles per row: its search is short)
    static constexpr bool COARSE_OK = THREADS >= 256;
    static constexpr bool ORDERED = THREADS == 1024;   // live keys compacted in product order (see expand_compact)
    int R, a0, n, a_lo, prune;
    const int2 *b_colocc;          // per B tile: (tile column, occupancy word) -- one 8-byte gather gives the key and the pruning test
    struct Product {
        int a, b;          // operand tile ids
        unsigned acol;     // occupied columns of the A tile
    };
    // by value: address-taken locals would put the kernel on a scratch (private memory) segment
    __device__ __forceinline__ Product tile_b(int q, bool want_acol) const
    {
        Product r;
        int ar;
        r.acol = 0xFFFFu;
        if (COARSE_OK && coarse) {   // a short walk from the tile of the 64-product block instead of a log2(R)-step search
            ar = cstart[q >> 6];
            while (roff[ar + 1] <= q) ++ar;
        } else {
            ar = s1_find_a(roff, 0, R, q);
        }
        r.b = rbs[ar] + (q - roff[ar]);
        if (want_acol) r.acol = rco[ar];
        r.a = a_lo + a0 + ar;
        return r;
    }
    // key of product q: (tile col, q); a product whose tiles cannot meet gets the padding key and sorts to the end
    __device__ __forceinline__ KeyT product_key(int q) const
    {
        if (q >= n) return ~KeyT(0);
        const Product pr = tile_b(q, prune != 0);
        const int2 co = b_colocc[pr.b];
        if (prune && !(pr.acol & ((unsigned)co.y >> 16))) return ~KeyT(0);
        if constexpr (RANK) return KeyT(co.x) << QB;        // the low bits take the key's live position (expand_compact)
        return (KeyT(co.x) << QB) | KeyT(q);
    }
    // expand all n products of the row, keep the live ones: their keys are packed into keys[0..nlive) in
    // arbitrary order (ballot + one LDS atomic per wave and chunk) -- the sort that follows fixes the order,
    // and only live keys get sorted.  Returns nlive; keys[nlive..npad_to) are set to the padding key.
    __device__ __forceinline__ int expand_compact(const int tid, int *s_cnt, int npad_to_mult, int *ordcnt) const
    {
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long lt = (1ull << lane) - 1ull;
        // four chunks per trip: their table searches and B-side gathers are independent and overlap; the
        // compaction follows once the keys are in registers
        constexpr int U = 4;
        if constexpr (ORDERED) {
            // 16-wave bins keep the live keys in PRODUCT ORDER (chunk, wave, lane ascending), so that a stable sort on
            // the tile column alone finishes the job: every (chunk, wave) posts its live count, a barrier, and each
            // wave adds up the counts in front of it (at most 64 LDS reads)
            constexpr int WAVES = THREADS / 64;
            int total = 0;
            for (int q0 = 0; q0 < n; q0 += U * THREADS) {
                KeyT key[U];
                unsigned long long bal[U];
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = product_key(q0 + u * THREADS + tid);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bal[u] = __ballot(key[u] != ~KeyT(0));
                    if (lane == 0) ordcnt[u * WAVES + wave] = __popcll(bal[u]);
                }
                __syncthreads();
                int c = lane < U * WAVES ? ordcnt[lane] : 0;          // U * WAVES = 64 counts, one per lane
                int inc = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += o;
                }
                const int trip_total = __shfl(inc, 63, 64);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int base = total + __shfl(inc - c, u * WAVES + wave, 64);
                    if (key[u] != ~KeyT(0)) {
                        const int pos = base + __popcll(bal[u] & lt);
                        if constexpr (RANK) {
                            // RANK keys: (tile column, live position).  Product order = position order, so the stable sort on
                            // the column bits still yields ascending k inside a C tile; the product index -- which needs up to
                            // 16 bits and would push a 19-bit tile column past 32 -- waits in a 2-byte side table
                            keys[pos] = key[u] | KeyT(pos);
                            qmap[pos] = (uint16_t)(q0 + u * THREADS + tid);
                        } else {
                            keys[pos] = key[u];
                        }
                    }
                }
                total += trip_total;
                __syncthreads();                                      // the counts are re-posted by the next trip
            }
            if (tid == 0) *s_cnt = total;
        } else {
            // ballot + one LDS atomic per wave and chunk: arbitrary order, the full-key sort that follows fixes it
            for (int q0 = 0; q0 < n; q0 += U * THREADS) {
                KeyT key[U];
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = product_key(q0 + u * THREADS + tid);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = key[u] != ~KeyT(0);
                    const unsigned long long bal = __ballot(live);
                    if (bal) {
                        int base = 0;
                        const int leader = __builtin_ctzll(bal);
                        if (lane == leader) base = atomicAdd(s_cnt, __popcll(bal));
                        base = __shfl(base, leader, 64);
                        if (live) keys[base + __popcll(bal & lt)] = key[u];
                    }
                }
            }
        }
        __syncthreads();
        const int nlive = *s_cnt;
        int upto = npad_to_mult;               // THREADS * 2^e >= nlive: what the register sort will load
        while (upto < nlive) upto <<= 1;
        if (upto > CAP) upto = CAP;
        for (int x = nlive + tid; x < upto; x += THREADS) keys[x] = ~KeyT(0);
        __syncthreads();
        return nlive;
    }
    // lanes of the wave holding the same 8-bit digit as this one (among the valid lanes)
    static __device__ __forceinline__ unsigned long long match_digit(const bool valid, const unsigned d)
    {
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        return m;
    }
    template <int EPT> __device__ __forceinline__ void sort_radix(const int tid, const int n, unsigned *hist, int *wsum, const int first_bit,
                                                                  const int key_bits) const
    {
        static_assert(THREADS == 1024, "sized for 16 waves: 4096 counters, four per thread in the scan");
        constexpr int WAVES = THREADS / 64;
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int rpw = (n + THREADS - 1) / THREADS;   // rounds per wave, <= EPT
        const int e0 = wave * rpw * 64 + lane;
        unsigned *myhist = hist + wave * 256;
        for (int shift = first_bit; shift < key_bits; shift += 8) {
            for (int x = tid; x < WAVES * 256; x += THREADS) hist[x] = 0;
            __syncthreads();
            // digit counts (keys read straight from LDS: they are only held in registers for the scatter below, which
            // keeps 32 key registers from living across the scan).  One LDS atomic per key, except where the whole
            // round holds one digit (the product-index bits of neighbouring products, already grouped columns) -- 64
            // atomics on one counter serialise, so there the first lane adds the round's population instead
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const bool valid = r < rpw && e0 + r * 64 < n;
                const unsigned long long vm = __ballot(valid);
                if (vm != 0) {                             // wave-uniform
                    const unsigned d = valid ? (unsigned)(keys[e0 + r * 64] >> shift) & 255u : 0u;
                    const unsigned d0 = (unsigned)__shfl((int)d, __builtin_ctzll(vm), 64);
                    if (__ballot(valid && d != d0) == 0) {
                        if (lane == 0) myhist[d0] += (unsigned)__popcll(vm);
                    } else if (valid) {
                        atomicAdd(&myhist[d], 1u);
                    }
                }
            }
            __syncthreads();
            {   // exclusive scan over (digit, wave): thread t owns digit t>>2, waves 4(t&3) .. 4(t&3)+3
                const int d = tid >> 2, w0 = (tid & 3) * 4;
                unsigned v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = hist[(w0 + j) * 256 + d];
                const int tsum = (int)(v[0] + v[1] + v[2] + v[3]);
                int inc = tsum;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const int o = __shfl_up(inc, dd, 64);
                    if (lane >= dd) inc += o;
                }
                if (lane == 63) wsum[wave] = inc;
                __syncthreads();
                int ex = inc - tsum;
#pragma unroll
                for (int w = 0; w < WAVES; ++w)
                    if (w < wave) ex += wsum[w];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hist[(w0 + j) * 256 + d] = (unsigned)ex;
                    ex += (int)v[j];
                }
            }
            KeyT k[EPT];
#pragma unroll
            for (int r = 0; r < EPT; ++r) k[r] = (r < rpw && e0 + r * 64 < n) ? keys[e0 + r * 64] : KeyT(0);
            __syncthreads();   // counters scanned, and every key is in a register before the first one is overwritten
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                if (r < rpw) {                             // wave-uniform
                    const bool valid = e0 + r * 64 < n;
                    const unsigned d = (unsigned)(k[r] >> shift) & 255u;
                    const unsigned long long m = match_digit(valid, d);
                    if (valid) {
                        const unsigned base = myhist[d];
                        const int rank = __popcll(m & lt);
                        keys[base + rank] = k[r];
                        if (rank == 0) myhist[d] = base + (unsigned)__popcll(m);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // rounds are serial through myhist anyway: keep their ballots from piling up in registers
            }
            __syncthreads();
        }
    }
    template <int EPT, int LOGT> __device__ __forceinline__ void sort_regs(const int tid) const
    {
        KeyT v[EPT];
#pragma unroll
        for (int m = 0; m < EPT; ++m) v[m] = keys[m * THREADS + tid];
        __syncthreads();   // everyone has its keys in registers before the sort's LDS stages overwrite them
        s1_bitonic_regs<KeyT, THREADS, EPT, LOGT>(v, keys, tid);
    }
};

#ifdef PEM_S1_DEBUG
// diagnostic build only (make EXTRA=-DPEM_S1_DEBUG): phase clocks of the row-sort bins, spread over 1024 slots per
// bin so the bookkeeping atomics do not serialise; [bin][slot][stage, expand, sort, emit, rows, max row, -, -]
__device__ unsigned long long g_s1dbg[4][1024][8];
__device__ unsigned long long g_s1blk[4][1024][4];   // first 1024 blocks of every bin: start, end, HW_ID, XCC_ID
extern "C" void pem_debug_s1_blocks(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s1blk), sizeof(unsigned long long) * 4 * 1024 * 4); }
#define S1_DBG_MARK(k)                                                   \
    do {                                                                 \
        __syncthreads();                                                 \
        if (tid == 0) {                                                  \
            unsigned long long now = wall_clock64();                     \
            atomicAdd(&g_s1dbg[DBG_BIN][blockIdx.x & 1023][k], now - dbg_t); \
            dbg_t = now;                                                 \
        }                                                                \
    } while (0)
extern "C" void pem_debug_s1(unsigned long long *out32, int reset)
{
    static unsigned long long h[4][1024][8];
    if (reset) {
        memset(h, 0, sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s1dbg), h, sizeof(h));
    } else {
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s1dbg), sizeof(h));
        for (int b = 0; b < 4; ++b)
            for (int k = 0; k < 8; ++k) {
                unsigned long long acc = 0;
                for (int sl = 0; sl < 1024; ++sl) acc = k == 5 ? (h[b][sl][k] > acc ? h[b][sl][k] : acc) : acc + h[b][sl][k];
                out32[b * 8 + k] = acc;
            }
    }
}
#else
#define S1_DBG_MARK(k)
#endif

template <typename KeyT, int CAP, int QB, int THREADS, int RCAP, bool RANK = false>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? (CAP > 8192 || sizeof(KeyT) == 8 ? 4 : 8) : 1) s1_rowsort_kernel(const int *__restrict__ row_list, int nrows_bin, const int *__restrict__ a_tile_rowptr,
                                                             int tr_lo, int a_lo, const int *__restrict__ a_tile_colidx,
                                                             const int *__restrict__ acnt, const int *__restrict__ row_n, const int *__restrict__ row_lbase,
                                                             const int *__restrict__ b_tile_rowptr, const int2 *__restrict__ b_colocc,
                                                             const uint32_t *__restrict__ a_occ, int prune,
                                                             int *__restrict__ pairs_a, int *__restrict__ pairs_b,
                                                             int *__restrict__ scratch_col, int *__restrict__ scratch_off,
                                                             int2 *__restrict__ block_info, int *__restrict__ row_tc, int key_bits)
{
    constexpr int LOGT = THREADS == 64 ? 6 : THREADS == 256 ? 8 : 10;
    constexpr int EMAX = CAP / THREADS;
    static_assert(EMAX == 8 || EMAX == 32, "a bin sorts up to 8 (or, for the largest, 32) keys per thread");
    __shared__ KeyT keys[CAP];
    __shared__ int roff[RCAP + 1];     // product offset of every A tile of the row, relative to the row
    __shared__ int rbs[RCAP];          // first B tile id of that A tile's B tile row
    __shared__ unsigned rco[RCAP];     // occupied columns of that A tile
    __shared__ int wsum[THREADS / 64];
    __shared__ int s_cnt;
    static_assert(!RANK || (THREADS == 1024 && CAP <= (1 << QB) && sizeof(KeyT) == 4), "rank keys: ordered compaction, position fits the low bits");
    __shared__ uint16_t qmap[RANK ? CAP : 1];
    constexpr bool COARSE = S1Row<KeyT, CAP, QB, THREADS, RCAP, RANK>::COARSE_OK;
    __shared__ int cstart[COARSE ? S1_COARSE : 1];
    __shared__ unsigned radix_hist[THREADS == 1024 ? (THREADS / 64) * 256 : 1];   // digit counters of the radix sort (16-wave bins)
    __shared__ int ordcnt[64];                                                   // live counts per (chunk, wave) of the ordered compaction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef PEM_S1_DEBUG
    constexpr int DBG_BIN = CAP == 512 ? 0 : CAP == 2048 ? 1 : CAP == 8192 ? 2 : 3;
    unsigned long long dbg_t = 0;
#endif
    for (int li = blockIdx.x; li < nrows_bin; li += gridDim.x) {
        const int i = row_list[li];
#ifdef PEM_S1_DEBUG
        if (tid == 0) dbg_t = wall_clock64();      // 100 MHz
        const unsigned long long dbg_row0 = dbg_t;
#endif
        if (tid == 0) s_cnt = 0;
        S1Row<KeyT, CAP, QB, THREADS, RCAP, RANK> row;
        row.keys = keys;
        row.qmap = qmap;
        row.roff = roff;
        row.rbs = rbs;
        row.rco = rco;
        row.prune = prune;
        row.a_lo = a_lo;
        row.b_colocc = b_colocc;
        row.a0 = a_tile_rowptr[tr_lo + i] - a_lo;
        row.R = a_tile_rowptr[tr_lo + i + 1] - a_lo - row.a0;   // <= RCAP: the row classification saw to that
        row.n = row_n[i];
        {
            // the row's A-tile table: product counts -> offsets relative to the row (exclusive scan, THREADS entries per
            // trip -- one trip for all but hub rows), first B tile, occupied columns
            int carry = 0;
            for (int x0 = 0; x0 < row.R; x0 += THREADS) {
                const int x = x0 + tid;
                int c = 0;
                if (x < row.R) {
                    const int a = a_lo + row.a0 + x;
                    c = acnt[row.a0 + x];
                    rbs[x] = b_tile_rowptr[a_tile_colidx[a]];
                    rco[x] = a_occ[a] & 0xFFFFu;
                }
                int inc = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += o;
                }
                int ex = carry + inc - c;
                if (THREADS > 64 && row.R - x0 > 64) {   // (block-uniform) this trip's entries spill over the first wave
                    if (lane == 63) wsum[wave] = inc;
                    __syncthreads();
                    int tot = 0;
#pragma unroll
                    for (int w = 0; w < THREADS / 64; ++w) {
                        if (w < wave) ex += wsum[w];
                        tot += wsum[w];
                    }
                    carry += tot;
                    __syncthreads();                      // wsum is re-posted by the next trip
                } else {
                    carry += __shfl(inc, 63, 64);         // only wave 0 holds entries; THREADS == 64: the wave's total
                }
                if (x < row.R) roff[x] = ex;
            }
            if (tid == 0) roff[row.R] = row.n;
        }
        row.cstart = cstart;
        row.coarse = COARSE && row.n <= 64 * S1_COARSE;
        __syncthreads();
        if (row.coarse) {
            for (int x = tid; x < row.R; x += THREADS) {      // every 64-product block start inside this A tile's range
                const int lo = roff[x], hi = roff[x + 1];
                for (int c = (lo + 63) >> 6; (c << 6) < hi; ++c) cstart[c] = x;
            }
        }
        __syncthreads();
        S1_DBG_MARK(0);
        // expand the row's products into (tile col, product index) keys -- live ones only -- and sort them; equal
        // tile columns stay in product (= ascending k) order because the index is part of the key
        const int nl = row.expand_compact(tid, &s_cnt, THREADS, ordcnt);
        S1_DBG_MARK(1);
        if constexpr (THREADS == 1024) {
            // 16-wave bins (more than 2048 live keys): the keys sit in product order, so a STABLE radix sort on the tile
            // column bits alone (2 passes for up to 65536 tile columns) replaces a bitonic network over the whole key
            row.template sort_radix<EMAX>(tid, nl, radix_hist, wsum, QB, key_bits);
        } else if (nl <= THREADS)
            row.template sort_regs<1, LOGT>(tid);
        else if (nl <= THREADS * 2)
            row.template sort_regs<2, LOGT>(tid);
        else if (nl <= THREADS * 4)
            row.template sort_regs<4, LOGT>(tid);
        else
            row.template sort_regs<8, LOGT>(tid);             // a bin never holds more than CAP = THREADS * EMAX live keys
        // stream out the live products (the dead ones sorted behind them): sorted pairs, and per distinct tile
        // column (C tile) its column + first pair; output positions count live products only
        S1_DBG_MARK(2);
        const int lp0 = row_lbase[i], nlive = nl;
        int base = 0;
        for (int s0 = 0; s0 < nlive; s0 += THREADS) {
            const int s = s0 + tid;
            const bool valid = s < nlive;
            int j = 0, a = 0, b = 0;
            bool head = false;
            if (valid) {
                KeyT key = keys[s];
                int q = (int)(key & KeyT((1u << QB) - 1u));
                if constexpr (RANK) q = qmap[q];
                j = (int)(key >> QB);
                head = s == 0 || (int)(keys[s - 1] >> QB) != j;
                const auto pr = row.tile_b(q, false);
                a = pr.a;
                b = pr.b;
            }
            unsigned long long bal = __ballot(head);
            if (lane == 0) wsum[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                int c = wsum[w];
                if (w < wave) woff += c;
                tot += c;
            }
            if (valid) {
                pairs_a[lp0 + s] = a;
                pairs_b[lp0 + s] = b;
                if (head) {
                    int rank = base + woff + __popcll(bal & lt);
                    scratch_col[lp0 + rank] = j;
                    scratch_off[lp0 + rank] = lp0 + s;
                }
            }
            base += tot;
            __syncthreads();
        }
        // the row's slots behind its last tile start no tile: marked, and holding the end of the row's pairs (step 2 reads
        // a tile's pair range as [scratch_off[slot], scratch_off[slot + 1]))
        for (int x = base + tid; x < nlive; x += THREADS) {
            scratch_col[lp0 + x] = -1;
            scratch_off[lp0 + x] = lp0 + nlive;
        }
        // step 2 walks the slots in blocks of 256: note, for every block boundary inside this row's range, the row and
        // the boundary's position in the range (how many of the row's slots lie before it)
        for (long long b = ((long long)lp0 + 255) / 256 + tid; b * 256 < (long long)lp0 + nlive; b += THREADS)   // (64-bit: lp0 + nlive reaches 2^31 - 1)
            block_info[b] = make_int2(i, (int)(b * 256 - lp0));
        if (tid == 0) row_tc[i] = base;
        S1_DBG_MARK(3);
#ifdef PEM_S1_DEBUG
        if (tid == 0) {
            if (blockIdx.x < 1024) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                g_s1blk[DBG_BIN][blockIdx.x][0] = dbg_row0;
                g_s1blk[DBG_BIN][blockIdx.x][1] = dbg_t;
                g_s1blk[DBG_BIN][blockIdx.x][2] = hw;
                g_s1blk[DBG_BIN][blockIdx.x][3] = xcc;
            }
            atomicAdd(&g_s1dbg[DBG_BIN][blockIdx.x & 1023][4], 1ull);
            atomicMax(&g_s1dbg[DBG_BIN][blockIdx.x & 1023][5], dbg_t - dbg_row0);
        }
#endif
    }
}

// rows above the largest LDS bin: global expand (s1_xl_expand_kernel above) + radix sort + emit
__global__ void s1_xl_rowstart_kernel(const uint64_t *__restrict__ keys, size_t n, int bits_tc, int *__restrict__ xl_rowstart)
{
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    int i = (int)(keys[x] >> bits_tc);
    if (x == 0 || (int)(keys[x - 1] >> bits_tc) != i) xl_rowstart[i] = (int)x;
}

__global__ void s1_xl_emit_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm, const int *__restrict__ headx, size_t n,
                                  int bits_tc, const int *__restrict__ xl_rowstart, const int *__restrict__ row_lbase,
                                  const int *__restrict__ prod_a, const int *__restrict__ prod_b,
                                  int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ scratch_col,
                                  int *__restrict__ scratch_off, int2 *__restrict__ block_info, int *__restrict__ row_tc)
{
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    uint64_t key = keys[x];
    int i = (int)(key >> bits_tc), j = (int)(key & ((1ull << bits_tc) - 1ull));
    int rs = xl_rowstart[i];
    int s = (int)x - rs;
    int p0 = row_lbase[i], ni = row_lbase[i + 1] - p0;     // the row's live-product (= slot) range
    uint32_t o = perm[x];
    pairs_a[p0 + s] = prod_a[o];
    pairs_b[p0 + s] = prod_b[o];
    int hx = headx[x];
    const int ntiles_row = headx[rs + ni] - headx[rs];
    if (headx[x + 1] != hx) {
        int rank = hx - headx[rs];
        scratch_col[p0 + rank] = j;
        scratch_off[p0 + rank] = p0 + s;
    }
    if (s >= ntiles_row) {   // slots behind the row's last tile (see s1_rowsort_kernel)
        scratch_col[p0 + s] = -1;
        scratch_off[p0 + s] = p0 + ni;
    }
    if (((p0 + s) & 255) == 0) block_info[(p0 + s) >> 8] = make_int2(i, s);   // block boundary of step 2 (see s1_rowsort_kernel)
    if (s == 0) row_tc[i] = ntiles_row;
}

// Oversized rows, one workgroup per row.  The global path above sorts all oversized rows' products together: four radix
// passes over (row, tile column) keys, each a histogram launch, a scan and a scatter launch, then heads, a scan, row starts and
// the emit -- nineteen launches for what is, on webbase-1M, forty tile rows of ~40 k products (its directory pages: a row of
// 4 700 A tiles fits no LDS table): 0.3 ms of launch latency on the critical path of a 1.1 ms pass.  But s1_xl_expand_kernel
// has already put every such row's live products into the row's OWN stretch of the key buffer, in product order.  So each row
// is sorted where it lies by one 1024-thread workgroup: a stable LSD radix sort on the tile-column bits with the keys in
// global memory (L2-resident: a row is a few hundred KB) -- per-wave digit histograms in LDS, one scan of the 16 x 256
// counters, ballot-ranked scatter, as in the 16-wave LDS bins -- followed by the same emit as s1_rowsort_kernel.  One launch.
constexpr int S1_XLL_MAX = 1 << 18;     // rows with more live products than this keep the global path (one workgroup would take too long)
__global__ void __launch_bounds__(1024) s1_xl_rowsort_kernel(const int *__restrict__ xl_rows, int nrows_xl, const int *__restrict__ xl_base,
                                                             const int *__restrict__ row_lbase, uint64_t *k0, uint64_t *k1, int bits_tc,
                                                             const int *__restrict__ prod_a, const int *__restrict__ prod_b,
                                                             int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ scratch_col,
                                                             int *__restrict__ scratch_off, int2 *__restrict__ block_info, int *__restrict__ row_tc)
{
    constexpr int WAVES = 16;
    __shared__ unsigned hist[WAVES * 256];
    __shared__ int wsum[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned *myhist = hist + wave * 256;
    for (int li = blockIdx.x; li < nrows_xl; li += gridDim.x) {
        const int i = xl_rows[li];
        const int base = xl_base[i], lp0 = row_lbase[i], n = row_lbase[i + 1] - lp0;
        uint64_t *src = k0 + base, *dst = k1 + base;
        const int per = (((n + WAVES - 1) / WAVES) + 63) & ~63;      // every wave sorts one contiguous stretch: wave order = product order
        const int w0 = wave * per, w1 = w0 + per < n ? w0 + per : n;
        for (int shift = 32; shift < 32 + bits_tc; shift += 8) {
            for (int x = tid; x < WAVES * 256; x += 1024) hist[x] = 0;
            __syncthreads();
            for (int x = w0 + lane; x < w1; x += 64) atomicAdd(&myhist[(unsigned)(src[x] >> shift) & 255u], 1u);
            __syncthreads();
            {   // exclusive scan over (digit, wave): thread t owns digit t>>2, waves 4(t&3) .. 4(t&3)+3
                const int d = tid >> 2, wq = (tid & 3) * 4;
                unsigned v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = hist[(wq + j) * 256 + d];
                const int tsum = (int)(v[0] + v[1] + v[2] + v[3]);
                int inc = tsum;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const int o = __shfl_up(inc, dd, 64);
                    if (lane >= dd) inc += o;
                }
                if (lane == 63) wsum[wave] = inc;
                __syncthreads();
                int ex = inc - tsum;
#pragma unroll
                for (int w = 0; w < WAVES; ++w)
                    if (w < wave) ex += wsum[w];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hist[(wq + j) * 256 + d] = (unsigned)ex;
                    ex += (int)v[j];
                }
            }
            __syncthreads();
            for (int x0 = w0; x0 < w1; x0 += 64) {                   // (wave-uniform trip count)
                const int x = x0 + lane;
                const bool valid = x < w1;
                const uint64_t key = valid ? src[x] : 0ull;
                const unsigned d = (unsigned)(key >> shift) & 255u;
                unsigned long long m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                if (valid) {
                    const unsigned pos = myhist[d];
                    const int rank = __popcll(m & lt);
                    dst[pos + rank] = key;
                    if (rank == 0) myhist[d] = pos + (unsigned)__popcll(m);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __threadfence_block();
            __syncthreads();
            uint64_t *t = src;
            src = dst;
            dst = t;
        }
        // emit (as s1_rowsort_kernel): sorted pairs, and per distinct tile column its column + first pair into the row's slots
        int tiles = 0;
        for (int s0 = 0; s0 < n; s0 += 1024) {
            const int sidx = s0 + tid;
            const bool valid = sidx < n;
            int j = 0, a = 0, b = 0;
            bool head = false;
            if (valid) {
                const uint64_t key = src[sidx];
                j = (int)(key >> 32);
                const unsigned x = (unsigned)(key & 0xFFFFFFFFull);
                head = sidx == 0 || (int)(src[sidx - 1] >> 32) != j;
                a = prod_a[x];
                b = prod_b[x];
            }
            const unsigned long long bal = __ballot(head);
            if (lane == 0) wsum[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const int c = wsum[w];
                if (w < wave) woff += c;
                tot += c;
            }
            if (valid) {
                pairs_a[lp0 + sidx] = a;
                pairs_b[lp0 + sidx] = b;
                if (head) {
                    const int rank = tiles + woff + __popcll(bal & lt);
                    scratch_col[lp0 + rank] = j;
                    scratch_off[lp0 + rank] = lp0 + sidx;
                }
            }
            tiles += tot;
            __syncthreads();
        }
        for (int x = tiles + tid; x < n; x += 1024) {               // the row's slots behind its last tile (see s1_rowsort_kernel)
            scratch_col[lp0 + x] = -1;
            scratch_off[lp0 + x] = lp0 + n;
        }
        for (long long bb = ((long long)lp0 + 255) / 256 + tid; bb * 256 < (long long)lp0 + n; bb += 1024)
            block_info[bb] = make_int2(i, (int)(bb * 256 - lp0));
        if (tid == 0) row_tc[i] = tiles;
        __syncthreads();
    }
}

// row-local scratch -> reference layout (_C_tileColIdx, spgemm.cu:379; pair offsets :484)
// One block per tile row: the row knows where its tiles go (c_rowptr[i]) and where its scratch lives (its first
// pair), so the copy is two coalesced streams and needs no search.  (One WAVE per row was as fast on a whole matrix,
// where the kernel is bandwidth-bound, but left a 1/8 slice -- 8 k rows of ~300 tiles -- latency-bound: 34 us.)
__global__ void __launch_bounds__(256) s1_compact_kernel(const int *__restrict__ c_rowptr, int mt, long long ntc,
                                                         const int *__restrict__ row_lbase,
                                                         const int *__restrict__ scratch_col, const int *__restrict__ scratch_off, int npairs,
                                                         int *__restrict__ c_colidx, int *__restrict__ pairs_offset)
{
    for (int i = blockIdx.x; i < mt; i += gridDim.x) {
        const int t0 = c_rowptr[i], cnt = c_rowptr[i + 1] - t0;
        if (cnt == 0) continue;
        const int p0 = row_lbase[i];
        for (int r = threadIdx.x; r < cnt; r += blockDim.x) {
            c_colidx[t0 + r] = scratch_col[p0 + r];
            pairs_offset[t0 + r] = scratch_off[p0 + r];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pairs_offset[ntc] = npairs;
}

// _C_tileRowIdx (spgemm.cu:378) from _C_rowPtr, one wave per tile row.  Like Ctiles_rowPtr it has no reader on the
// default path (every consumer walks tile rows through _C_rowPtr) and is materialised on demand.
__global__ void __launch_bounds__(256) s1_crowidx_kernel(const int *__restrict__ c_rowptr, int mt, int tr_lo, int *__restrict__ c_rowidx)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < mt; i += nwaves)
        for (int t = c_rowptr[i] + lane; t < c_rowptr[i + 1]; t += 64) c_rowidx[t] = i + tr_lo;
}

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
// step 3 (spgemm.cu:593-661).  16 lanes per C tile, one C entry per lane (strided by 16);
// pairs ascending in k-tile, bits of Amask[r] & BT[c] ascending, one fma per product, the
// accumulator lives in a register and is stored once (no global RMW, no dependence on
// zero-filled memory -- SURVEY 2.3 #1).
// ------------------------------------------------------------------------------------------
// one fused multiply-add per product in the operands' own precision (the oracle's chain; the reference computes in
// double, spgemm.cu:728 -- fp32 is SURVEY 8(f)-3)
__device__ __forceinline__ double pem_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float pem_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

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

constexpr int S3_CHUNK = 256;   // = S3_EPW below: C entries one wave of step 3 takes

// ------------------------------------------------------------------------------------------
// Step 2 (a10 offsets + a11 + a12, spgemm.cu:483-484, 499-550, 552-591) in two kernels over the row-local
// step-1 scratch.  Step 1 leaves, for every live product slot s in [0, P): scratch_col[s] = tile column of the
// C tile whose pair list starts there (or -1: no tile starts in this slot), scratch_off[s] = first pair of
// that tile (a gap slot holds the end of its row's pairs, so the end of any tile's pairs is scratch_off[s+1]).
// A row's tiles sit at the front of the row's slot range in ascending column order, so the valid slots, read
// in slot order, ARE the C tile list in the reference's order.
//
// s2_tiles_kernel, one slot per lane, 256 slots per block:
//   index   the tile's index t = number of valid slots before it = (valid slots before the block) + ballot rank.
//           Step 1 notes for every block boundary the tile row it falls in and how far into the row's slots
//           (block_info); with _C_rowPtr that gives the first term in three scalar loads -- no scan over slots,
//           no dependence between blocks.
//   mask    boolean product over the tile's pairs (two 16-byte loads per operand tile)
//   out     _C_tileColIdx[t], pair offsets[t], Ctiles_mask[8t..] straight into the reference's dense layout --
//           this replaces s1_compact -- and the entry count of every 256 tiles (one integer atomic per wave and
//           group) for the entry offsets.
// (one small scan of the 256-tile group counts in between)
// s2_entries_kernel, one tile per lane: perTileNnz offsets from the group base + a block scan of the masks'
//   popcounts, the (r<<4|c) bytes, and step 3's chunk index -- replacing the 3-launch scan over all tiles.
//
// Measured and dropped: carrying (tiles, entries) through a decoupled look-back inside ONE kernel.  Flat window
// of 64 blocks: 1.27 ms (2000 blocks in flight = 30 round trips of ~2 us agent-scope loads behind the nearest
// prefix); with a ticket for the block order 1.40 ms (81 k atomics on one address, 11 ns each); two-level
// (groups of 64 blocks): 1.20 ms -- in-order completion puts every resident block behind the slowest lane of the
// oldest one, and a lane with a 40-pair tile takes 80 us; tile counts only, published at block start: 0.94 ms
// with every block polling from its first cycle, 0.78 ms with the look-back moved behind the mask loop; without
// any look-back the same kernel takes 0.36 ms.
// ------------------------------------------------------------------------------------------
constexpr int S2_GROUP = 256;             // tiles per block of s2_entries_kernel

__global__ void __launch_bounds__(256) s2_tiles_kernel(const int *__restrict__ scratch_col, const int *__restrict__ scratch_off, long long nslots,
                                                       const int2 *__restrict__ block_info, const int *__restrict__ c_rowptr,
                                                       const int *__restrict__ pairs_a, const int *__restrict__ pairs_b,
                                                       const uint16_t *__restrict__ a_masks, const uint16_t *__restrict__ b_masks, long long ntc,
                                                       int *__restrict__ c_colidx, int *__restrict__ pairs_offset, uint32_t *__restrict__ c_mask,
                                                       int *__restrict__ group_nnz, uint16_t *__restrict__ c_cnt)
{
    __shared__ unsigned bl[8][256];
    __shared__ int w_tiles[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x;
    // valid slots before this block: the tiles of all earlier tile rows + those of the boundary row that lie before it
    const int2 bi = block_info[blk];                         // (tile row of slot 256 blk, that slot's position in the row's range)
    const int row_t0 = c_rowptr[bi.x], row_tiles = c_rowptr[bi.x + 1] - row_t0;
    const long long t_blk = (long long)row_t0 + (bi.y < row_tiles ? bi.y : row_tiles);
    const long long s = (long long)blk * 256 + tid;
    int col = -1, p0 = 0, p1 = 0;
    if (s < nslots) {
        col = scratch_col[s];
        if (col >= 0) {
            p0 = scratch_off[s];
            p1 = s + 1 < nslots ? scratch_off[s + 1] : (int)nslots;
        }
    }
    const bool valid = col >= 0;
    const unsigned long long vb = __ballot(valid);
    if (lane == 0) w_tiles[wave] = __popcll(vb);
    if (blk == 0 && tid == 0) pairs_offset[ntc] = (int)nslots;   // closing pair offset
    __syncthreads();
    int tile_off = __popcll(vb & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) tile_off += w_tiles[w];
    const long long t = t_blk + tile_off;
    // the masks
    unsigned cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // natural layout: cw[q] = row 2q | row 2q+1 << 16
    // the pair list is walked with the next pair's masks and the one after's ids already in flight: a lane's chain per pair
    // is then one gather deep instead of two (tiles of a band times a band hold 30+ pairs)
    if (p0 < p1) {
        S2Masks cur = s2_load_masks(a_masks, b_masks, pairs_a[p0], pairs_b[p0]);
        int na = 0, nb = 0;
        if (p0 + 1 < p1) {
            na = pairs_a[p0 + 1];
            nb = pairs_b[p0 + 1];
        }
        for (int p = p0; p < p1; ++p) {
            S2Masks nxt = cur;
            int nna = 0, nnb = 0;
            if (p + 1 < p1) {
                nxt = s2_load_masks(a_masks, b_masks, na, nb);
                if (p + 2 < p1) {
                    nna = pairs_a[p + 2];
                    nnb = pairs_b[p + 2];
                }
            }
            s2_pair_mask(cur, bl, tid, cw);
            cur = nxt;
            na = nna;
            nb = nnb;
        }
    }
    int nnz_t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) nnz_t += __popc(cw[q]);
    const bool store = valid && t < ntc;          // (t < ntc always: both count the same valid slots)
    // entry counts per group of S2_GROUP tiles: a wave's tiles are consecutive, so they span at most two groups
    {
        const long long t_first = __shfl(t, vb ? __builtin_ctzll(vb) : 0, 64);
        const long long g0 = t_first / S2_GROUP;
        int c0 = (store && t / S2_GROUP == g0) ? nnz_t : 0, c1 = (store && t / S2_GROUP != g0) ? nnz_t : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            c0 += __shfl_xor(c0, d, 64);
            c1 += __shfl_xor(c1, d, 64);
        }
        if (lane == 0 && vb) {
            if (c0) atomicAdd(&group_nnz[g0], c0);
            if (c1) atomicAdd(&group_nnz[g0 + 1], c1);
        }
    }
    if (!store) return;
    c_colidx[t] = col;
    pairs_offset[t] = p0;
    if (c_cnt) c_cnt[t] = (uint16_t)nnz_t;      // the entry offsets then come from 2 bytes per tile, not from its 32-byte mask
    // reference packing: word q = (row 2q) << 16 | row 2q+1  (spgemm.cu:533-543)
    *reinterpret_cast<uint4 *>(c_mask + 8 * t) = make_uint4((cw[0] << 16) | (cw[0] >> 16), (cw[1] << 16) | (cw[1] >> 16),
                                                            (cw[2] << 16) | (cw[2] >> 16), (cw[3] << 16) | (cw[3] >> 16));
    *reinterpret_cast<uint4 *>(c_mask + 8 * t + 4) = make_uint4((cw[4] << 16) | (cw[4] >> 16), (cw[5] << 16) | (cw[5] >> 16),
                                                                (cw[6] << 16) | (cw[6] >> 16), (cw[7] << 16) | (cw[7] >> 16));
}

// a11's offsets + a12 (spgemm.cu:546, 1288, 552-591), one C tile per lane, S2_GROUP tiles per block: entry offsets =
// the group's base (scanned group counts) + a block scan of the masks' popcounts; the (r<<4|c) bytes; and, for step 3,
// the tile every S3_CHUNK-entry chunk of C starts in.
// repeat pass on an unchanged plan: the sizes the host assumed (from the previous pass) against what this pass computed
struct WarmCheck {
    int on;
    long long P, Pall, TC, nnz, nxl;
    int c0, c1, c2, c3;
    int *host_flags;     // where set: the pass's status flags are left in host memory by the checking thread (no copy node after the pass)
};
__device__ __forceinline__ void warm_check(const WarmCheck &w, const long long *__restrict__ d_scalars, const int *__restrict__ bin_count,
                                           int *__restrict__ flags)
{
    if (d_scalars[0] != w.P || d_scalars[1] != w.TC || d_scalars[2] != w.nnz || d_scalars[3] != w.Pall || bin_count[0] != w.c0 ||
        bin_count[1] != w.c1 || bin_count[2] != w.c2 || bin_count[3] != w.c3 || bin_count[5] != w.nxl)
        flags[FLAG_CAPACITY] = 1;
    if (w.host_flags) {
        // the caller guarantees that nothing after this thread sets a flag in this pass (s2_offsets_kernel + step 3: none do)
        for (int i = 0; i < NUM_FLAGS; ++i) w.host_flags[i] = flags[i];
    }
}

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
__global__ void __launch_bounds__(256) s3_accumulate_wide_kernel(
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
        const int p0 = __shfl(my_p0, ti, 64), p1 = __shfl(my_p1, ti, 64);
        const int a0 = __shfl(my_a0, ti, 64), b0 = __shfl(my_b0, ti, 64), av0 = __shfl(my_av0, ti, 64), bv0 = __shfl(my_bv0, ti, 64);
        // (every shuffle sits in front of the `continue`: a lane that has left cannot be read from)
        const int a1 = __shfl(my_a1, ti, 64), b1 = __shfl(my_b1, ti, 64), av1 = __shfl(my_av1, ti, 64), bv1 = __shfl(my_bv1, ti, 64);
        const int toff = DECODE ? __shfl(my_off, ti, 64) : 0;
        if (!valid) continue;
        if (BAND && p1 - p0 >= S3_BAND_MIN) continue;   // many-pair tiles: s3_band_kernel's
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
            const unsigned rc = c_rowcolidx[e];
            r = rc >> 4;
            c = rc & 15;
        }
        VT acc = VT(0);
        int p = p0;
        if (!DEEP) {   // first pair: everything but the two records and the values is already here
            const unsigned aw = s3_ld<IDX32>(a_rec, 16ll * a0 + r), bw = s3_ld<IDX32>(b_rec_t, 16ll * b0 + c);
            const unsigned am = aw & 0xFFFFu, bm = bw & 0xFFFFu;
            unsigned m = am & bm;
            const int ao = av0 + (int)(aw >> 16), bo = bv0 + (int)(bw >> 16);
            while (m) {
                const int kk = __builtin_ctz(m);
                m &= m - 1;
                const unsigned below = (1u << kk) - 1u;
                acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)ao + __popc(am & below)), s3_ld<IDX32>(b_vals_t, (long long)bo + __popc(bm & below)), acc);
            }
            ++p;
            {
                if (p < p1) {             // second pair
                    const unsigned aw1 = s3_ld<IDX32>(a_rec, 16ll * a1 + r), bw1 = s3_ld<IDX32>(b_rec_t, 16ll * b1 + c);
                    const unsigned am1 = aw1 & 0xFFFFu, bm1 = bw1 & 0xFFFFu;
                    unsigned m1 = am1 & bm1;
                    const int ao1 = av1 + (int)(aw1 >> 16), bo1 = bv1 + (int)(bw1 >> 16);
                    while (m1) {
                        const int kk = __builtin_ctz(m1);
                        m1 &= m1 - 1;
                        const unsigned below = (1u << kk) - 1u;
                        acc = pem_fma(s3_ld<IDX32>(a_vals, (long long)ao1 + __popc(am1 & below)), s3_ld<IDX32>(b_vals_t, (long long)bo1 + __popc(bm1 & below)), acc);
                    }
                    ++p;
                }
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
        if constexpr (IDX32)
            *reinterpret_cast<VT *>(reinterpret_cast<char *>(c_vals) + (size_t)((unsigned)e * (unsigned)sizeof(VT))) = acc;
        else
            c_vals[e] = acc;
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
            if (mine) c_vals[e] = acc;
        }
    }
}

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


__global__ void warm_verify_kernel(const long long *__restrict__ d_scalars, const int *__restrict__ bin_count, WarmCheck wc, int *__restrict__ flags)
{
    warm_check(wc, d_scalars, bin_count, flags);
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------
extern "C" pem_status pem_cplan_create(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, int32_t tr_lo, int32_t tr_hi, pem_cplan **out)
{
    if (!ctx || !A || !B || !out) return PEM_E_INVALID;
    *out = nullptr;
    if (A->cols != B->rows) {
        set_error("pem_cplan_create: A is %d x %d but B is %d x %d", A->rows, A->cols, B->rows, B->cols);
        return PEM_E_INVALID;
    }
    if (A->value_bytes != B->value_bytes) {
        set_error("pem_cplan_create: A holds %s values, B %s; both operands must share one value type", A->value_bytes == 4 ? "fp32" : "fp64",
                  B->value_bytes == 4 ? "fp32" : "fp64");
        return PEM_E_INVALID;
    }
    if (tr_hi < 0) tr_hi = A->tile_rows;
    if (tr_lo < 0 || tr_lo > tr_hi || tr_hi > A->tile_rows) {
        set_error("pem_cplan_create: tile-row range [%d, %d) outside [0, %d]", tr_lo, tr_hi, A->tile_rows);
        return PEM_E_INVALID;
    }
    pem_cplan *p = new pem_cplan();
    p->owner = ctx;
    p->A = A;
    p->B = B;
    p->tr_lo = tr_lo;
    p->tr_hi = tr_hi;
    p->a_lo = A->h_tile_rowptr[(size_t)tr_lo];
    p->a_hi = A->h_tile_rowptr[(size_t)tr_hi];
    for (int i = tr_lo; i < tr_hi; ++i)      // the longest tile row of the slice (grid of the oversized rows' expansion)
        p->max_row_tiles = std::max(p->max_row_tiles, A->h_tile_rowptr[(size_t)i + 1] - A->h_tile_rowptr[(size_t)i]);
    // Every switch is a property of the PLAN, latched here (pem_cplan_set_option changes it later): a C ABI whose behaviour
    // followed the process environment at call time is not a boundary a maintainer can bind.  The environment variables only
    // give the DEFAULTS a new plan starts from (test hooks: PEM_S1_FORCE_KEY64 / PEM_S1_XLCAP push small inputs through the
    // code a B with more than 2^17 tile columns / a tile row beyond the LDS bins selects).
    auto env_is = [](const char *name, const char *val) {
        const char *e = getenv(name);
        return e && !strcmp(e, val);
    };
    p->opt_prune = !env_is("PEM_PRUNE", "0");
    p->opt_key64 = env_is("PEM_S1_FORCE_KEY64", "1");
    {
        const char *e = getenv("PEM_S1_XLCAP");
        p->opt_xlcap = e ? atoi(e) : 0;
    }
    p->opt_band = !env_is("PEM_S3_BAND", "0");
    p->opt_step1_esc = env_is("PEM_STEP1", "esc");
    p->opt_wide = !env_is("PEM_WIDE", "0");
    p->opt_warm = !env_is("PEM_NO_WARM", "1");
    p->opt_export_rows = env_is("PEM_EXPORT", "rows");
    p->opt_s1_serial = getenv("PEM_S1_SERIAL") != nullptr;
    p->opt_decode = !env_is("PEM_S3_DECODE", "0");
    p->opt_xl_global = env_is("PEM_S1_XL_GLOBAL", "1");
    p->opt_s3_xcd = !env_is("PEM_S3_XCD", "0");
    p->opt_idx64 = env_is("PEM_S3_IDX64", "1");
    p->opt_mark = !env_is("PEM_S3_MARK", "0");
    {
        const char *e = getenv("PEM_S3_EPW");
        p->opt_epw = e ? atoi(e) : 0;
    }
    *out = p;
    return PEM_OK;
}

// A plan's instantiated graph is not destroyed when the plan lets go of it but when its CONTEXT goes: the HIP runtime of this image
// (ROCm 7.x) keeps state across the graph executables of a process that a later hipGraphLaunch trips over once earlier ones have
// been destroyed -- a segmentation fault in hip::Graph::UpdateStreams, seen after 8-12 create / replay / destroy rounds of plans
// with forked streams (tools/split_tune_emulate.py).  Executables are small; a context that outlives thousands of plans can call
// pem_ctx_trim, which does not touch them either -- they go with pem_ctx_destroy.
static void retire_graph(pem_ctx *ctx, pem_cplan *plan)
{
    if (!plan->graph_exec) return;
    if (ctx)
        ctx->retired_graphs.push_back(plan->graph_exec);
    else
        (void)hipGraphExecDestroy(plan->graph_exec);
    plan->graph_exec = nullptr;
}

extern "C" pem_status pem_cplan_destroy(pem_ctx *ctx, pem_cplan *plan)
{
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (plan) retire_graph(ctx, plan);
    delete plan;
    return PEM_OK;
}

extern "C" pem_status pem_cplan_get_info(const pem_cplan *p, pem_cplan_info *info)
{
    if (!p || !info) return PEM_E_INVALID;
    info->tile_row_begin = p->tr_lo;
    info->tile_row_end = p->tr_hi;
    info->row_begin = p->tr_lo * 16;
    info->row_end = p->tr_hi * 16 < p->A->rows ? p->tr_hi * 16 : p->A->rows;
    info->ntiles_c = p->ntiles_c;
    info->npairs = p->npairs;
    info->nnz_c = p->nnz_c;
    info->npairs_all = p->npairs_all;
    return PEM_OK;
}

static int *plan_option_slot(pem_cplan *p, pem_option which)
{
    switch (which) {
    case PEM_OPT_PRUNE: return &p->opt_prune;
    case PEM_OPT_STEP1_GLOBAL_SORT: return &p->opt_step1_esc;
    case PEM_OPT_WIDE: return &p->opt_wide;
    case PEM_OPT_WARM: return &p->opt_warm;
    case PEM_OPT_S3_BAND: return &p->opt_band;
    case PEM_OPT_S1_FORCE_KEY64: return &p->opt_key64;
    case PEM_OPT_S1_XLCAP: return &p->opt_xlcap;
    case PEM_OPT_EXPORT_ROWS: return &p->opt_export_rows;
    case PEM_OPT_S1_SERIAL: return &p->opt_s1_serial;
    case PEM_OPT_S3_DECODE: return &p->opt_decode;
    case PEM_OPT_S1_XL_GLOBAL: return &p->opt_xl_global;
    case PEM_OPT_S3_EPW: return &p->opt_epw;
    case PEM_OPT_S3_XCD: return &p->opt_s3_xcd;
    case PEM_OPT_S3_IDX64: return &p->opt_idx64;
    case PEM_OPT_S3_MARK: return &p->opt_mark;
    default: return nullptr;
    }
}

extern "C" pem_status pem_cplan_set_option(pem_cplan *plan, pem_option which, int64_t value)
{
    if (!plan) return PEM_E_INVALID;
    int *slot = plan_option_slot(plan, which);
    if (!slot || value < 0 || value > 0x7FFFFFFF) {
        set_error("pem_cplan_set_option: unknown option %d or value %lld out of range", (int)which, (long long)value);
        return PEM_E_INVALID;
    }
    const int v = (which == PEM_OPT_S1_XLCAP || which == PEM_OPT_S3_EPW) ? (int)value : (value != 0);
    if (*slot == v) return PEM_OK;
    *slot = v;
    // a repeat pass re-uses the sizes (and possibly the captured graph) of the previous one: whatever changes the kernels
    // that run or the sizes they produce starts the plan over
    plan->warm = false;
    if (plan->graph_exec) {
        retire_graph(plan->owner, plan);
    }
    plan->graph_failed = false;
    return PEM_OK;
}

extern "C" pem_status pem_cplan_get_option(const pem_cplan *plan, pem_option which, int64_t *value)
{
    if (!plan || !value) return PEM_E_INVALID;
    const int *slot = plan_option_slot(const_cast<pem_cplan *>(plan), which);
    if (!slot) {
        set_error("pem_cplan_get_option: unknown option %d", (int)which);
        return PEM_E_INVALID;
    }
    *value = *slot;
    return PEM_OK;
}

// a device-side primitive gave up (the chained scan's bounded wait): fail the call rather than hand back wrong arrays
static pem_status check_internal(const int *hf)
{
    if (!hf[FLAG_INTERNAL]) return PEM_OK;
    set_error("internal: a device scan ran out of its poll budget; the pass's results are not valid");
    return PEM_E_HIP;
}

static pem_status step_elapsed(pem_ctx *ctx, int e0, int e1, double *dst)
{
    float ms = 0.f;
    PEM_HIP(hipEventElapsedTime(&ms, ctx->ev[e0], ctx->ev[e1]));
    *dst = ms;
    return PEM_OK;
}

static pem_status step1_esc_impl(pem_ctx *ctx, pem_cplan *p)
{
    const pem_tiled *A = p->A, *B = p->B;
    hipStream_t st = ctx->stream;
    const int nA = p->a_hi - p->a_lo, mt = p->tr_hi - p->tr_lo;
    const int bits_tc = bits_for((uint64_t)B->tile_cols), bits_row = bits_for((uint64_t)(mt > 0 ? mt : 1));
    p->state = 0;
    p->pairs_ready = false;
    p->ntiles_c = p->npairs = p->nnz_c = 0;
    PEM_HIP(hipEventRecord(ctx->ev[0], st));
    PEM_TRY(p->c_tile_rowptr.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_HIP(hipMemsetAsync(p->c_tile_rowptr.p, 0, sizeof(int) * ((size_t)mt + 1), st));
    // product offsets per A tile: all products (expansion / sort capacity) and live products (output positions)
    const int prune = p->opt_prune;
    PEM_TRY(p->aprod_off.reserve(sizeof(int) * ((size_t)nA + 4)));
    PEM_TRY(p->lprod_off.reserve(sizeof(int) * ((size_t)nA + 4)));
    if (nA > 0)
        PEM_LAUNCH(ctx, s1_aprod_kernel, grid_for((size_t)nA * 8, 256), 256, A->tile_colidx.as<int>(), A->tile_occ.as<uint32_t>(), p->a_lo, nA,
                   B->tile_rowptr.as<int>(), B->tile_occ.as<uint32_t>(), prune, p->aprod_off.as<int>(), p->lprod_off.as<int>(),
                   (const long long *)nullptr, 0, (int *)nullptr, (int *)nullptr);
    PEM_TRY(exclusive_scan_i32_pair(ctx, p->aprod_off.as<int>(), p->lprod_off.as<int>(), (size_t)nA, ctx->d_scalars + 3, ctx->d_scalars));
    int64_t P = 0, Pall = 0;
    {
        int64_t two[4];
        PEM_TRY(read_scalars(ctx, ctx->d_scalars, 4, two));
        P = two[0];
        Pall = two[3];
    }
    p->npairs_all = Pall;
    if (P > 0x7FFFFFFFll || Pall > 0x7FFFFFFFll) {
        set_error("step 1: %lld tile pairs exceed the int32 range of the reference's pair arrays", (long long)P);
        return PEM_E_OVERFLOW;
    }
    p->npairs = P;
    const size_t n = (size_t)P;
    PEM_TRY(p->pairs_offset.reserve(sizeof(int) * 4));
    int64_t TC = 0;
    if (n > 0) {
        PEM_TRY(p->prod_a.reserve(sizeof(int) * n));
        PEM_TRY(p->prod_b.reserve(sizeof(int) * n));
        PEM_TRY(p->sk0.reserve(sizeof(uint64_t) * n));
        PEM_TRY(p->sk1.reserve(sizeof(uint64_t) * n));
        PEM_TRY(p->sv0.reserve(sizeof(uint32_t) * n));
        PEM_TRY(p->sv1.reserve(sizeof(uint32_t) * n));
        PEM_LAUNCH(ctx, s1_xl_expand_kernel, grid_for((size_t)nA * 16, 256), 256, A->tile_keys.as<long long>(), A->tile_rowptr.as<int>(),
                   A->tile_occ.as<uint32_t>(), p->a_lo, nA, p->tr_lo, p->lprod_off.as<int>(), (const int *)nullptr, B->tile_rowptr.as<int>(),
                   B->tile_colidx.as<int>(), B->tile_occ.as<uint32_t>(), prune, bits_tc, p->sk0.as<uint64_t>(), p->sv0.as<uint32_t>(),
                   p->prod_a.as<int>(), p->prod_b.as<int>(), 0, (const int *)nullptr);
        uint64_t *keys = nullptr;
        PEM_TRY(radix_sort_u64_u32(ctx, p->sk0.as<uint64_t>(), p->sk1.as<uint64_t>(), p->sv0.as<uint32_t>(), p->sv1.as<uint32_t>(), n,
                                   bits_tc + bits_row, &keys, &p->sorted_perm));
        DevBuf &head = ctx->tmp[2];
        PEM_TRY(head.reserve(sizeof(int) * (n + 4)));
        PEM_LAUNCH(ctx, s1_heads_kernel, grid_for(n, 256), 256, keys, n, head.as<int>());
        PEM_TRY(exclusive_scan_i32(ctx, head.as<int>(), head.as<int>(), n, ctx->d_scalars + 1));
        PEM_TRY(read_scalars(ctx, ctx->d_scalars + 1, 1, &TC));
        const size_t ntc = (size_t)TC;
        PEM_TRY(p->c_tile_rowidx.reserve(sizeof(int) * (ntc + 4)));
        PEM_TRY(p->c_tile_colidx.reserve(sizeof(int) * (ntc + 4)));
        PEM_TRY(p->pairs_offset.reserve(sizeof(int) * (ntc + 4)));
        PEM_LAUNCH(ctx, s1_emit_ctiles_kernel, grid_for(n, 256), 256, keys, head.as<int>(), n, p->tr_lo, bits_tc, p->c_tile_rowidx.as<int>(),
                   p->c_tile_colidx.as<int>(), p->pairs_offset.as<int>());
        p->c_rowidx_valid = true;
        p->compact_valid = true;
        PEM_LAUNCH(ctx, s1_c_rowptr_kernel, grid_for(ntc, 256), 256, p->c_tile_rowidx.as<int>(), (long long)TC, p->tr_lo, mt,
                   p->c_tile_rowptr.as<int>());
    } else {
        PEM_HIP(hipMemsetAsync(p->pairs_offset.p, 0, sizeof(int), st));
    }
    p->ntiles_c = TC;
    PEM_HIP(hipEventRecord(ctx->ev[1], st));
    p->state = 1;
    return PEM_OK;
}

template <typename KeyT>
static void launch_rowsorts(pem_ctx *ctx, pem_cplan *p, const int *counts, int mt, int cap3, int prune, bool rank2)
{
    const pem_tiled *A = p->A, *B = p->B;
    int *rl = p->row_list.as<int>();
    constexpr int QBITS = sizeof(KeyT) == 4 ? 15 : 24;   // product-index field of the sort key
    const int key_bits = QBITS + bits_for((uint64_t)B->tile_cols);
#define PEM_ROWSORT(BIN, CAP, QB, THREADS, RCAP, MAXGRID)                                                                                 \
    if (counts[BIN] > 0) {                                                                                                           \
        int grid = counts[BIN] < (MAXGRID) ? counts[BIN] : (MAXGRID);                                                                \
        PEM_LAUNCH_NAMED(ctx, "s1_rowsort_kernel<" #CAP ">", (s1_rowsort_kernel<KeyT, CAP, QB, THREADS, RCAP>), grid, THREADS,             \
                         rl + (size_t)(BIN) * mt, counts[BIN], A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, A->tile_colidx.as<int>(), \
                         p->aprod_off.as<int>(), p->row_n.as<int>(), p->row_lbase.as<int>(), B->tile_rowptr.as<int>(),              \
                         B->tile_colocc.as<int2>(), A->tile_occ.as<uint32_t>(), prune, p->pairs_a.as<int>(), p->pairs_b.as<int>(),     \
                         p->scratch_col.as<int>(), p->scratch_off.as<int>(), p->block_info.as<int2>(), p->c_tile_rowptr.as<int>(),     \
                         key_bits);                                                                                                    \
    }
    // The bins are independent and run concurrently: the largest non-empty one on the main stream, the others
    // forked onto auxiliary streams and joined before the row-count scan.  Order matters: a block of the 32768-key
    // bin needs a CU's whole LDS, so it can only start on an EMPTY CU.  On the main stream it is dispatched the
    // moment the row classification retires, a few microseconds before the forked streams get through their
    // event waits, and its ~100 blocks are placed before the smaller bins flood the CUs (behind them it was
    // starved until they drained, which made it the critical path of step 1).
    hipStream_t main_stream = ctx->stream;
    (void)hipEventRecord(ctx->ev_fork, main_stream);
    bool forked[3] = {false, false, false};
    int next_aux = -1;                     // -1: the main stream is still free
    const bool serial = p->opt_s1_serial != 0;   // diagnostic: every bin alone, one after the other
    auto bin_begin = [&]() {
        if (next_aux < 0 || serial) return;
        (void)hipStreamWaitEvent(ctx->aux[next_aux], ctx->ev_fork, 0);
        ctx->stream = ctx->aux[next_aux];
        forked[next_aux] = true;
    };
    auto bin_end = [&]() {
        if (serial) return;
        if (next_aux >= 0) (void)hipEventRecord(ctx->ev_join[next_aux], ctx->aux[next_aux]);
        ctx->stream = main_stream;
        ++next_aux;
    };
    if constexpr (sizeof(KeyT) == 4) {
        if (cap3 > S1_CAP2 && counts[3] > 0) {
            bin_begin();
            PEM_ROWSORT(3, 32768, QBITS, 1024, 1024, 1 << 20)
            bin_end();
        }
    }
    if (counts[2] > 0) {
        bin_begin();
        if (rank2) {
            // B with 2^17 .. 2^19 tile columns (cage15: 322 179): the 8192-key bin sorts 32-bit (tile column, live position)
            // keys -- 78 KB of LDS, two workgroups per CU -- instead of 64-bit (tile column, product index) keys at 106 KB
            const int grid = counts[2];
            PEM_LAUNCH_NAMED(ctx, "s1_rowsort_kernel<8192,rank>", (s1_rowsort_kernel<uint32_t, 8192, 13, 1024, 1024, true>), grid, 1024,
                             rl + (size_t)2 * mt, counts[2], A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, A->tile_colidx.as<int>(),
                             p->aprod_off.as<int>(), p->row_n.as<int>(), p->row_lbase.as<int>(), B->tile_rowptr.as<int>(),
                             B->tile_colocc.as<int2>(), A->tile_occ.as<uint32_t>(), prune, p->pairs_a.as<int>(), p->pairs_b.as<int>(),
                             p->scratch_col.as<int>(), p->scratch_off.as<int>(), p->block_info.as<int2>(), p->c_tile_rowptr.as<int>(),
                             13 + bits_for((uint64_t)B->tile_cols));
        } else {
            PEM_ROWSORT(2, 8192, QBITS, 1024, 2048, 1 << 20)
        }
        bin_end();
    }
    if (counts[1] > 0) {
        bin_begin();
        PEM_ROWSORT(1, 2048, QBITS, 256, 1024, 1 << 20)
        bin_end();
    }
    if (counts[0] > 0) {
        bin_begin();
        PEM_ROWSORT(0, 512, QBITS, 64, 256, 1 << 20)
        bin_end();
    }
#undef PEM_ROWSORT
    for (int k = 0; k < 3; ++k)
        if (forked[k]) (void)hipStreamWaitEvent(main_stream, ctx->ev_join[k], 0);
}

static pem_status step1_rows_impl(pem_ctx *ctx, pem_cplan *p)
{
    const pem_tiled *A = p->A, *B = p->B;
    hipStream_t st = ctx->stream;
    const int nA = p->a_hi - p->a_lo, mt = p->tr_hi - p->tr_lo;
    const int bits_tc = bits_for((uint64_t)B->tile_cols), bits_row = bits_for((uint64_t)(mt > 0 ? mt : 1));
    // the 32768-key LDS bin needs 32-bit keys (tile col + 15 index bits); wider B goes to the global path above 8192
    // 32-bit keys = tile col + 15 index bits; wider B uses 64-bit keys (24 index bits, no 32768-key LDS bin)
    // (a row of exactly 2^15 products in a B of exactly 2^17 tile columns could form the key 0xFFFFFFFF, which is the
    // padding key: the index field holds 2^15 - 1 products at most)
    const bool k32 = bits_tc + 15 <= 32 && !p->opt_key64;
    const int cap3 = k32 ? S1_CAP3 : S1_CAP2;
    const int qcap = k32 ? (1 << 15) - 1 : (1 << 24) - 1;
    const int xlcap = p->opt_xlcap > 0 ? p->opt_xlcap : 0x7FFFFFFF;   // test hook: rows with more live products take the global path
    // the two small bins also bound a row's products BEFORE pruning: they are all expanded, 64 (256) per trip, and a row of 300
    // live products among 20 000 kept its one wave busy for 60 us -- the whole kernel's time on a 1/8 row block of webbase-1M.
    // Eight times the live capacity: at twice, the 8-way shares gained most (0.283 -> 0.270 ms on average) but the band matrices,
    // whose rows all carry 3-4 dead products per live one, moved up a bin wholesale (cage15 step 1 20.4 -> 23.9 ms)
    const int ncap0 = S1_NCAP0, ncap1 = S1_NCAP1;
    // 2^17 < tile columns < 2^19: the 8192-key bin takes 32-bit (tile column, live position) keys (see launch_rowsorts); its
    // product index lives in a 16-bit side table and its A-tile table is the smaller one
    const bool rank2 = !k32 && !p->opt_key64 && B->tile_cols < (1 << 19);
    const int rcap2 = rank2 ? 1024 : S1_RCAP2, qcap2 = rank2 ? 65535 : qcap;
    p->state = 0;
    p->pairs_ready = false;
    p->c_rowidx_valid = false;
    p->compact_valid = false;
    p->ntiles_c = p->npairs = p->nnz_c = 0;
    if (!ctx->capturing) PEM_HIP(hipEventRecord(ctx->ev[0], st));
    PEM_TRY(p->c_tile_rowptr.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->row_list.reserve(sizeof(int) * (5 * (size_t)mt + 4)));
    PEM_TRY(p->bin_count.reserve(sizeof(int) * 8));
    PEM_TRY(p->xl_base.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->pairs_offset.reserve(sizeof(int) * 4));
    // one launch clears the status flags, the bin counters, the pass scalars (P live, T_C, C_nnz, P all), pairs_offset[0]
    // and the per-row product totals; the per-row tile counts in c_tile_rowptr are zeroed by the row classification below
    PEM_TRY(p->row_n.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->row_lbase.reserve(sizeof(int) * ((size_t)mt + 4)));
    // (a repeat pass knows T_C, so the reset also clears step 2's group counters and saves it a memset)
    int ngroups_reset = 0;
    p->group_nnz_cleared = false;
    if (p->warm_pass && p->w_TC > 0) {
        ngroups_reset = (int)((p->w_TC + S2_GROUP - 1) / S2_GROUP) + 4;
        PEM_TRY(p->group_nnz.reserve(sizeof(int) * (size_t)ngroups_reset));
        p->group_nnz_cleared = true;
    }
    const size_t reset_n = std::max((size_t)mt + 1, (size_t)ngroups_reset);
    PEM_LAUNCH(ctx, s1_reset_kernel, grid_for(reset_n, 256), 256, ctx->d_flags, p->bin_count.as<int>(),
               reinterpret_cast<long long *>(ctx->d_scalars), p->pairs_offset.as<int>(), p->c_tile_rowptr.as<int>(), mt, p->row_n.as<int>(),
               p->row_lbase.as<int>(), p->group_nnz.as<int>(), ngroups_reset);
    // products per A tile (all: expansion; live: what is sorted and stored) and their totals per tile row.  The only scan
    // left is the one over the ROWS' live totals (row r's pairs, and its C tile slots, start at row_lbase[r]); offsets inside
    // a row are rebuilt in LDS by the row's workgroup, and the grand total of all products is only ever a 64-bit scalar --
    // so a product whose tile-level products exceed 2^31 (cage15 on one GPU: 2.8 G) is fine as long as the LIVE pairs,
    // which the reference's int arrays index, do not.
    const int prune = p->opt_prune;
    PEM_TRY(p->aprod_off.reserve(sizeof(int) * ((size_t)nA + 4)));
    PEM_TRY(p->lprod_off.reserve(sizeof(int) * ((size_t)nA + 4)));
    if (nA > 0)
        PEM_LAUNCH(ctx, s1_aprod_kernel, grid_for((size_t)nA * 8, 256), 256, A->tile_colidx.as<int>(), A->tile_occ.as<uint32_t>(), p->a_lo, nA,
                   B->tile_rowptr.as<int>(), B->tile_occ.as<uint32_t>(), prune, p->aprod_off.as<int>(), p->lprod_off.as<int>(),
                   A->tile_keys.as<long long>(), p->tr_lo, p->row_n.as<int>(), p->row_lbase.as<int>());
    PEM_TRY(exclusive_scan_i32(ctx, p->row_lbase.as<int>(), p->row_lbase.as<int>(), (size_t)mt, ctx->d_scalars));
    // per-row tile counts are accumulated in c_tile_rowptr and scanned in place afterwards
    if (mt > 0)
        PEM_LAUNCH(ctx, s1_rowclass_kernel, grid_for((size_t)mt, 256), 256, A->tile_rowptr.as<int>(), p->tr_lo, mt, p->row_n.as<int>(),
                   p->row_lbase.as<int>(), cap3, qcap, xlcap, rcap2, qcap2, ncap0, ncap1, p->row_list.as<int>(), p->bin_count.as<int>(), p->xl_base.as<int>(),
                   p->c_tile_rowptr.as<int>(), reinterpret_cast<long long *>(ctx->d_scalars));
    // one read-back: P, the bin populations and the product total of the oversized rows
    int64_t P = 0, Pall = 0;
    int counts[4];
    size_t n_xl;
    int nrows_xl = 0, max_xl = 0;
    if (p->warm_pass) {
        max_xl = p->w_max_xl;
        P = p->w_P;
        Pall = p->w_Pall;
        for (int b = 0; b < 4; ++b) counts[b] = p->w_counts[b];
        n_xl = (size_t)p->w_nxl;
        nrows_xl = p->w_nrows_xl;
    } else {
        int *hb = reinterpret_cast<int *>(ctx->h_scalars + 32);
        PEM_HIP(hipMemcpyAsync(hb, p->bin_count.p, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
        int64_t sc[4];
        PEM_TRY(read_scalars(ctx, ctx->d_scalars, 4, sc));
        P = sc[0];
        Pall = sc[3];
        for (int b = 0; b < 4; ++b) counts[b] = p->w_counts[b] = hb[b];
        n_xl = (size_t)hb[5];
        nrows_xl = p->w_nrows_xl = hb[4];
        max_xl = p->w_max_xl = hb[6];
        p->w_nxl = (int64_t)n_xl;
        p->w_P = P;
        p->w_Pall = Pall;
    }
    p->npairs_all = Pall;
    if (P > 0x7FFFFFFFll) {
        set_error("step 1: %lld live tile pairs exceed the int32 range of the reference's pair arrays", (long long)P);
        return PEM_E_OVERFLOW;
    }
    p->npairs = P;
    const size_t n = (size_t)P;
    int64_t TC = 0;
    if (n > 0) {
        // sizing phase "pairs": everything P-sized comes out of one driver allocation (a repeat pass finds it all in place)
        PEM_TRY(arena_phase(ctx->arena, {{&p->pairs_a, sizeof(int) * (n + 4)}, {&p->pairs_b, sizeof(int) * (n + 4)},
                                         {&p->scratch_col, sizeof(int) * (n + 4)}, {&p->scratch_off, sizeof(int) * (n + 4)},
                                         {&p->block_info, sizeof(int2) * (n / 256 + 4)}}));
        PEM_TRY(p->pairs_a.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->pairs_b.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->scratch_col.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->scratch_off.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->block_info.reserve(sizeof(int2) * (n / 256 + 4)));
        // Oversized rows.  Up to S1_XLL_MAX live products each they are sorted where they lie, one workgroup per row: three
        // launches with no shared scratch, so the chain runs on a stream of its own BESIDE the row bins (it is a third of the
        // bins' time on webbase-1M; behind them it was a quarter of the whole pass).  Larger ones go through the global sort,
        // after the bins (it uses the context's scan and sort scratch).
        const bool xl_local = n_xl > 0 && !p->opt_xl_global && max_xl <= S1_XLL_MAX;
        if (n_xl > 0) {
            PEM_TRY(p->prod_a.reserve(sizeof(int) * n_xl));
            PEM_TRY(p->prod_b.reserve(sizeof(int) * n_xl));
            PEM_TRY(p->sk0.reserve(sizeof(uint64_t) * n_xl));
            PEM_TRY(p->sk1.reserve(sizeof(uint64_t) * n_xl));
            PEM_TRY(p->sv0.reserve(sizeof(uint32_t) * n_xl));
            PEM_TRY(p->sv1.reserve(sizeof(uint32_t) * n_xl));
            PEM_TRY(p->xl_rowstart.reserve(sizeof(int) * ((size_t)mt + 4)));
            PEM_TRY(p->xl_lrel.reserve(sizeof(int) * ((size_t)nA + 4)));
        }
        auto xl_expand = [&](int local) {
            PEM_LAUNCH(ctx, s1_xl_rel_kernel, (unsigned)(nrows_xl > 0 ? nrows_xl : 1), 1024, p->row_list.as<int>() + (size_t)4 * mt, nrows_xl,
                       A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, p->lprod_off.as<int>(), p->xl_lrel.as<int>());
            // (the grid: the oversized rows x sixteen A tiles per block up to the plan's longest tile row)
            const dim3 xgrid((unsigned)((p->max_row_tiles + 15) / 16 > 0 ? (p->max_row_tiles + 15) / 16 : 1), (unsigned)(nrows_xl > 0 ? nrows_xl : 1));
            PEM_LAUNCH(ctx, s1_xl_expand_kernel, xgrid, 256, A->tile_keys.as<long long>(), A->tile_rowptr.as<int>(),
                       A->tile_occ.as<uint32_t>(), p->a_lo, nA, p->tr_lo, p->xl_lrel.as<int>(), p->xl_base.as<int>(), B->tile_rowptr.as<int>(),
                       B->tile_colidx.as<int>(), B->tile_occ.as<uint32_t>(), prune, bits_tc, p->sk0.as<uint64_t>(), p->sv0.as<uint32_t>(),
                       p->prod_a.as<int>(), p->prod_b.as<int>(), local, p->row_list.as<int>() + (size_t)4 * mt);
        };
        if (xl_local) {
            hipStream_t main_stream = ctx->stream;
            (void)hipEventRecord(ctx->ev_fork, main_stream);
            (void)hipStreamWaitEvent(ctx->aux[3], ctx->ev_fork, 0);
            ctx->stream = ctx->aux[3];
            xl_expand(1);
            PEM_LAUNCH(ctx, s1_xl_rowsort_kernel, (unsigned)(nrows_xl > 0 ? nrows_xl : 1), 1024, p->row_list.as<int>() + (size_t)4 * mt, nrows_xl,
                       p->xl_base.as<int>(), p->row_lbase.as<int>(), p->sk0.as<uint64_t>(), p->sk1.as<uint64_t>(), bits_tc, p->prod_a.as<int>(),
                       p->prod_b.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->scratch_col.as<int>(), p->scratch_off.as<int>(),
                       p->block_info.as<int2>(), p->c_tile_rowptr.as<int>());
            (void)hipEventRecord(ctx->ev_join[3], ctx->aux[3]);
            ctx->stream = main_stream;
        }
        if (k32)
            launch_rowsorts<uint32_t>(ctx, p, counts, mt, cap3, prune, false);
        else
            launch_rowsorts<uint64_t>(ctx, p, counts, mt, cap3, prune, rank2);
        if (xl_local) (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join[3], 0);
        if (n_xl > 0 && !xl_local) {   // global expand + stable radix sort on (row, tile col)
            xl_expand(0);
            uint64_t *keys = nullptr;
            uint32_t *perm = nullptr;
            PEM_TRY(radix_sort_u64_u32(ctx, p->sk0.as<uint64_t>(), p->sk1.as<uint64_t>(), p->sv0.as<uint32_t>(), p->sv1.as<uint32_t>(), n_xl,
                                       bits_tc + bits_row, &keys, &perm));
            DevBuf &head = ctx->tmp[2];
            PEM_TRY(head.reserve(sizeof(int) * (n_xl + 4)));
            PEM_LAUNCH(ctx, s1_heads_kernel, grid_for(n_xl, 256), 256, keys, n_xl, head.as<int>());
            PEM_TRY(exclusive_scan_i32(ctx, head.as<int>(), head.as<int>(), n_xl, nullptr));
            PEM_LAUNCH(ctx, s1_xl_rowstart_kernel, grid_for(n_xl, 256), 256, keys, n_xl, bits_tc, p->xl_rowstart.as<int>());
            PEM_LAUNCH(ctx, s1_xl_emit_kernel, grid_for(n_xl, 256), 256, keys, perm, head.as<int>(), n_xl, bits_tc, p->xl_rowstart.as<int>(),
                       p->row_lbase.as<int>(), p->prod_a.as<int>(), p->prod_b.as<int>(),
                       p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->scratch_col.as<int>(), p->scratch_off.as<int>(),
                       p->block_info.as<int2>(), p->c_tile_rowptr.as<int>());
        }
        // _C_rowPtr = exclusive scan of the per-row tile counts (spgemm.cu:1168); total = T_C
        PEM_TRY(exclusive_scan_i32(ctx, p->c_tile_rowptr.as<int>(), p->c_tile_rowptr.as<int>(), (size_t)mt, ctx->d_scalars + 1));
        if (p->warm_pass) {
            TC = p->w_TC;
        } else {
            PEM_TRY(read_scalars(ctx, ctx->d_scalars + 1, 1, &TC));
            p->w_TC = TC;
        }
        // _C_tileColIdx and the pair offsets in the reference's dense layout are written by step 2's fused kernel
        // straight from the row-local scratch; a caller that stops after step 1 gets them from ensure_compact()
        p->pairs_ready = true;
    }
    p->ntiles_c = TC;
    if (!ctx->capturing) PEM_HIP(hipEventRecord(ctx->ev[1], st));
    p->state = 1;
    return PEM_OK;
}

// row-local scratch -> _C_tileColIdx / pair offsets (reference layout) without step 2: the step-wise API after step 1,
// and the 16-lanes-per-tile baseline kernels
static pem_status ensure_compact(pem_ctx *ctx, const pem_cplan *cp)
{
    pem_cplan *p = const_cast<pem_cplan *>(cp);
    if (p->compact_valid || !p->pairs_ready || p->state < 1) return PEM_OK;
    const int mt = p->tr_hi - p->tr_lo;
    const size_t ntc = (size_t)p->ntiles_c;
    PEM_ENTER(ctx);
    PEM_TRY(p->c_tile_colidx.reserve(sizeof(int) * (ntc + 4)));
    PEM_TRY(p->pairs_offset.reserve(sizeof(int) * (ntc + 4)));
    if (mt > 0)
        PEM_LAUNCH(ctx, s1_compact_kernel, (unsigned)mt, 256, p->c_tile_rowptr.as<int>(), mt, (long long)ntc, p->row_lbase.as<int>(),
                   p->scratch_col.as<int>(), p->scratch_off.as<int>(), (int)p->npairs,
                   p->c_tile_colidx.as<int>(), p->pairs_offset.as<int>());
    p->compact_valid = true;
    return PEM_OK;
}

static pem_status step1_impl(pem_ctx *ctx, pem_cplan *p, bool allow_warm)
{
    p->warm_pass = false;
    if (p->opt_step1_esc) {
        p->warm = false;
        return step1_esc_impl(ctx, p);
    }
    p->warm_pass = allow_warm && p->warm && p->opt_warm;
    return step1_rows_impl(ctx, p);
}

static pem_status step2_impl(pem_ctx *ctx, pem_cplan *p)
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
            const size_t nblk = (n + 255) / 256, ngroups = (ntc + S2_GROUP - 1) / S2_GROUP;
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
            PEM_LAUNCH(ctx, s2_tiles_kernel, (unsigned)nblk, 256, p->scratch_col.as<int>(), p->scratch_off.as<int>(), (long long)n,
                       p->block_info.as<int2>(), p->c_tile_rowptr.as<int>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), A->masks.as<uint16_t>(),
                       B->masks.as<uint16_t>(), (long long)ntc, p->c_tile_colidx.as<int>(), p->pairs_offset.as<int>(), p->c_mask.as<uint32_t>(),
                       group_nnz, decode ? p->c_tile_cnt.as<uint16_t>() : (uint16_t *)nullptr);
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
            if (p->warm_pass) wc = WarmCheck{1, p->w_P, p->w_Pall, p->w_TC, p->w_nnz, p->w_nxl, p->w_counts[0], p->w_counts[1], p->w_counts[2], p->w_counts[3], nullptr};
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

static pem_status step3_impl(pem_ctx *ctx, pem_cplan *p)
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

extern "C" pem_status pem_spgemm_step1(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    PEM_TRY(step1_impl(ctx, plan, false));   // step-wise calls always read the sizes back
    PEM_TRY(ensure_compact(ctx, plan));      // ... and leave step 1's outputs in the reference layout
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return step_elapsed(ctx, 0, 1, &ctx->timings.step1_ms);
}

extern "C" pem_status pem_spgemm_step2(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    plan->warm_pass = false;                 // step-wise calls read every size back, also after a warm pem_spgemm on this plan
    PEM_TRY(step2_impl(ctx, plan));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return step_elapsed(ctx, 2, 3, &ctx->timings.step2_ms);
}

extern "C" pem_status pem_spgemm_step3(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    plan->warm_pass = false;
    PEM_TRY(step3_impl(ctx, plan));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return step_elapsed(ctx, 4, 5, &ctx->timings.step3_ms);
}

// one iteration of the reference's timed loop (spgemm.cu:1136-1341): wall clock around
// step1+step2+step3 including every allocation and size read-back, ended by a device sync.
extern "C" pem_status pem_set_graph_replay(pem_ctx *ctx, int on)
{
    if (!ctx) return PEM_E_INVALID;
    ctx->graph_replay = on != 0;
    return PEM_OK;
}

extern "C" pem_status pem_spgemm(pem_ctx *ctx, pem_cplan *plan)
{
    if (!ctx || !plan) return PEM_E_INVALID;
    PEM_ENTER(ctx);
    auto t0 = std::chrono::high_resolution_clock::now();
    // back-to-back steps share their boundary events (each record is a barrier packet, ~5 us of pipeline bubble)
    struct Chain {
        pem_ctx *c;
        explicit Chain(pem_ctx *c_) : c(c_) { c->chain_events = true; }
        ~Chain() { c->chain_events = false; }
    } chain(ctx);
    auto launch_verify = [&]() {
        if (plan->verify_folded) return;   // step 2's last kernel has already compared the sizes
        const WarmCheck wc = {1, plan->w_P, plan->w_Pall, plan->w_TC, plan->w_nnz, plan->w_nxl, plan->w_counts[0], plan->w_counts[1],
                              plan->w_counts[2], plan->w_counts[3], nullptr};
        PEM_LAUNCH(ctx, warm_verify_kernel, 1, 1, reinterpret_cast<const long long *>(ctx->d_scalars), plan->bin_count.as<int>(), wc, ctx->d_flags);
    };
    // Graph replay (pem_set_graph_replay): a repeat pass has fixed grids, sizes and buffer addresses, so its ~28 launches,
    // the fork onto the auxiliary streams and the joins are captured once and replayed as one hipGraph -- the launch
    // gaps go (3 % of a 2.3 ms pass, 7 % of a 0.27 ms one).  The graph is re-captured whenever any device buffer
    // was (re)allocated since the capture.
    const bool use_graph = ctx->graph_replay && plan->warm && !ctx->profiling && !plan->opt_step1_esc && plan->opt_warm;
    bool graphed = false;
    if (use_graph && !plan->graph_failed) {
        if (plan->graph_exec && plan->graph_gen != pem::alloc_generation()) {
            retire_graph(ctx, plan);
        }
        if (!plan->graph_exec) {   // capture; any failure falls back to plain launches for the rest of the plan's life
            hipGraph_t graph = nullptr;
            pem_status cs = PEM_OK;
            if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                ctx->capturing = true;
                cs = step1_impl(ctx, plan, true);
                if (cs == PEM_OK) cs = step2_impl(ctx, plan);
                if (cs == PEM_OK) cs = step3_impl(ctx, plan);
                if (cs == PEM_OK) launch_verify();
                const hipError_t ee = hipStreamEndCapture(ctx->stream, &graph);
                ctx->capturing = false;
                if (cs == PEM_OK && ee == hipSuccess && graph &&
                    hipGraphInstantiate(&plan->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess)
                    plan->graph_gen = pem::alloc_generation();
                else
                    plan->graph_exec = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!plan->graph_exec) {
                (void)hipGetLastError();
                plan->graph_failed = true;
            }
        }
        if (plan->graph_exec) {
            PEM_HIP(hipGraphLaunch(plan->graph_exec, ctx->stream));
            int hf[NUM_FLAGS];
            if (plan->flags_mirrored) {   // the flags are already in host memory when the graph ends: no copy behind it
                PEM_HIP(hipStreamSynchronize(ctx->stream));
                for (int i = 0; i < NUM_FLAGS; ++i) hf[i] = ctx->h_flags[i];
            } else {
                PEM_TRY(read_flags(ctx, hf));
            }
            PEM_TRY(check_internal(hf));
            if (hf[FLAG_CAPACITY]) {   // sizes differ from the captured ones (cannot happen while A and B are immutable)
                retire_graph(ctx, plan);
                plan->warm = false;
            } else {
                graphed = true;
            }
        }
    }
    if (!graphed) {
        PEM_TRY(step1_impl(ctx, plan, true));
        PEM_TRY(step2_impl(ctx, plan));
        PEM_TRY(step3_impl(ctx, plan));
        if (plan->warm_pass) {
            // repeat pass: the host never waited for P / T_C / C_nnz; check on the device that they are what it assumed
            launch_verify();
            int hf[NUM_FLAGS];
            if (plan->flags_mirrored) {
                PEM_HIP(hipStreamSynchronize(ctx->stream));   // the pass's one synchronisation
                for (int i = 0; i < NUM_FLAGS; ++i) hf[i] = ctx->h_flags[i];
            } else {
                PEM_TRY(read_flags(ctx, hf));
            }
            PEM_TRY(check_internal(hf));
            if (hf[FLAG_CAPACITY]) {        // cannot happen while A and B are immutable; recover by a full pass
                plan->warm = false;
                PEM_TRY(step1_impl(ctx, plan, false));
                PEM_TRY(step2_impl(ctx, plan));
                PEM_TRY(step3_impl(ctx, plan));
            }
        }
    }
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    plan->warm = plan->pairs_ready && plan->state == 3;
    ctx->timings.spgemm_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    if (graphed) {   // a replayed graph carries no per-step events (recorded inside a capture they do not time the replay)
        ctx->timings.step1_ms = ctx->timings.step2_ms = ctx->timings.step3_ms = 0.0;
        return PEM_OK;
    }
    PEM_TRY(step_elapsed(ctx, 0, 1, &ctx->timings.step1_ms));
    PEM_TRY(step_elapsed(ctx, 1, 3, &ctx->timings.step2_ms));
    PEM_TRY(step_elapsed(ctx, 3, 5, &ctx->timings.step3_ms));
    return PEM_OK;
}

// _C_tileRowIdx on demand (see s1_crowidx_kernel)
static pem_status ensure_c_rowidx(pem_ctx *ctx, const pem_cplan *p)
{
    if (p->c_rowidx_valid || p->state < 1) return PEM_OK;
    const int mt = p->tr_hi - p->tr_lo;
    PEM_ENTER(ctx);
    PEM_TRY(p->c_tile_rowidx.reserve(sizeof(int) * ((size_t)p->ntiles_c + 4)));
    if (mt > 0 && p->ntiles_c > 0)
        PEM_LAUNCH(ctx, s1_crowidx_kernel, grid_for((size_t)mt * 64, 256), 256, p->c_tile_rowptr.as<int>(), mt, p->tr_lo, p->c_tile_rowidx.as<int>());
    p->c_rowidx_valid = true;
    return PEM_OK;
}

// Ctiles_rowPtr on demand (see s2_crowptr_kernel)
static pem_status ensure_c_rowptr(pem_ctx *ctx, const pem_cplan *p)
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
static pem_status ensure_c_rowcolidx(pem_ctx *ctx, const pem_cplan *cp)
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

extern "C" pem_status pem_cplan_get_array(pem_ctx *ctx, const pem_cplan *p, pem_cplan_array which, void *host_dst, int64_t bytes)
{
    if (!ctx || !p || (!host_dst && bytes > 0)) return PEM_E_INVALID;
    const size_t TC = (size_t)p->ntiles_c, P = (size_t)p->npairs, NZ = (size_t)p->nnz_c, mt = (size_t)(p->tr_hi - p->tr_lo);
    const void *src = nullptr;
    size_t want = 0;
    int need = 1;
    switch (which) {
    case PEM_C_TILE_ROWPTR: src = p->c_tile_rowptr.p; want = 4 * (mt + 1); break;
    case PEM_C_TILE_ROWIDX:
        PEM_TRY(ensure_c_rowidx(ctx, p));
        src = p->c_tile_rowidx.p;
        want = 4 * TC;
        break;
    case PEM_C_TILE_COLIDX:
        PEM_TRY(ensure_compact(ctx, p));
        src = p->c_tile_colidx.p;
        want = 4 * TC;
        break;
    case PEM_C_PAIRS_OFFSET:
        PEM_TRY(ensure_compact(ctx, p));
        src = p->pairs_offset.p;
        want = 4 * (TC + 1);
        break;
    case PEM_C_PAIRS_A: src = p->pairs_a.p; want = 4 * P; need = 2; break;
    case PEM_C_PAIRS_B: src = p->pairs_b.p; want = 4 * P; need = 2; break;
    case PEM_C_MASK: src = p->c_mask.p; want = 32 * TC; need = 2; break;
    case PEM_C_TILE_NNZ_PTR: src = p->c_tile_nnz_ptr.p; want = 4 * (TC + 1); need = 2; break;
    case PEM_C_ROWPTR:
        PEM_TRY(ensure_c_rowptr(ctx, p));
        src = p->c_rowptr.p;
        want = 16 * TC;
        need = 2;
        break;
    case PEM_C_ROWCOLIDX:
        PEM_TRY(ensure_c_rowcolidx(ctx, p));
        src = p->c_rowcolidx.p;
        want = NZ;
        need = 2;
        break;
    case PEM_C_VALS: src = p->c_vals.p; want = (size_t)p->A->value_bytes * NZ; need = 3; break;   // native type
    default: set_error("unknown pem_cplan_array %d", (int)which); return PEM_E_INVALID;
    }
    if (p->state < need) {
        set_error("pem_cplan_get_array(%d): step %d has not run", (int)which, need);
        return PEM_E_STATE;
    }
    if ((size_t)bytes != want) {
        set_error("pem_cplan_get_array(%d): caller passed %lld bytes, array has %zu", (int)which, (long long)bytes, want);
        return PEM_E_INVALID;
    }
    if (want == 0) return PEM_OK;
    PEM_ENTER(ctx);
    PEM_HIP(hipMemcpyAsync(host_dst, src, want, hipMemcpyDeviceToHost, ctx->stream));
    PEM_HIP(hipStreamSynchronize(ctx->stream));
    return PEM_OK;
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
    PEM_HIP(hipStreamSynchronize(ctx->stream));
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
