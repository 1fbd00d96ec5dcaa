// step1.hip -- row a9 + a10: the tile-level symbolic product.  Kernels and host driver of step 1 (see spgemm.hip for the
// overview of the three steps).
#include "spgemm_internal.h"

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
pem_status pem::ensure_compact(pem_ctx *ctx, const pem_cplan *cp)
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

pem_status pem::step1_impl(pem_ctx *ctx, pem_cplan *p, bool allow_warm)
{
    p->warm_pass = false;
    if (p->opt_step1_esc) {
        p->warm = false;
        return step1_esc_impl(ctx, p);
    }
    p->warm_pass = allow_warm && p->warm && p->opt_warm;
    return step1_rows_impl(ctx, p);
}

// _C_tileRowIdx on demand (see s1_crowidx_kernel)
pem_status pem::ensure_c_rowidx(pem_ctx *ctx, const pem_cplan *p)
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
