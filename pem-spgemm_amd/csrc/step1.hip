// step1.hip -- rows a9 + a10: the tile-level symbolic product.  Kernels and host driver of step 1 (see spgemm.hip for the
// overview of the three steps).
//
// Reference: SPA bitmask kernels run twice (count + emit) or the NSPARSE binned hash path (spgemm.cu:271-384, 1141-1218;
// NSPARSE/spgemm_nsparse_kernel.h), then every C tile re-derives its pairs by binary-search intersection, twice
// (spgemm.cu:387-497).  Here every (A tile (i,k), B tile (k,j)) IS a product of tile row i keyed by j, so grouping a row's
// products by j yields the C tile list and the pair lists, in ascending k, in one pass.
//
// Round 4 layout of the default path (three phases, the row sorts concurrent):
//   s1_expand_kernel     one 8-wave workgroup per 512 consecutive A tiles of the slice: every product is formed and tested
//                        ONCE (the A tile's occupied columns against the B tile's occupied rows); the live ones are written,
//                        in product order, to a stretch of the live list the workgroup takes from a bump counter (tile column |
//                        A tile, B tile) -- so a tile row's live products are a few contiguous pieces (one per A tile; one per
//                        chunk of 512 the row touches where the pieces lie end to end)
//   s1_rowclass_kernel   per tile row: live total from the pieces, size class
//   (scan of the totals: where the row's pairs go)
//   s1_tiny_kernel / s1_rowsort_kernel<...>   per row: load the live keys (coalesced), sort by tile column, stream out the
//                        sorted pairs and the C tile list of the row
// Rounds 1-3 counted the live products in one kernel and expanded + tested them AGAIN inside the row sorts (68 % of webbase-1M's
// products are dead, 83 % of scircuit's: the sort kernels spent a third to a half of their instructions on products they then
// dropped, and the bins had to bound a row's products BEFORE pruning).
#include "spgemm_internal.h"

using namespace pem;

// ------------------------------------------------------------------------------------------
// global expand + radix sort ("esc"): PEM_OPT_STEP1_GLOBAL_SORT, the A/B baseline of step 1
// ------------------------------------------------------------------------------------------
// per A tile (i,k): number of tiles in B's tile row k (= tile-level intermediate products; the quantity of
// spgemm_nsparse_kernel.h:135-151 per A tile instead of per row), and how many of them are live.
// A product (A tile (i,k), B tile (k,j)) can only contribute if some column occupied in the A tile is a row
// occupied in the B tile.  The reference's tile-level symbolic product keeps every product and so
// materialises pairs -- and whole C tiles -- that stay empty (83 % of the pairs of the scircuit stand-in).
// With prune != 0 those dead products are dropped here, before anything is sorted or stored: the final C
// is unchanged, only the intermediate C tile / pair lists lose their empty members.  prune == 0 reproduces
// the reference's lists exactly.  8 lanes per A tile: aprod = all products, lprod = live products.
__global__ void __launch_bounds__(256) s1_aprod_kernel(const int *__restrict__ a_tile_colidx, const uint32_t *__restrict__ a_occ, int a_lo,
                                                       int nA, const int *__restrict__ b_tile_rowptr, const uint32_t *__restrict__ b_occ,
                                                       int prune, int *__restrict__ aprod, int *__restrict__ lprod)
{
    constexpr int G = 8;
    const int arel = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int l = threadIdx.x & (G - 1);
    const bool in = arel < nA;
    int len = 0, cnt = 0;
    if (in) {
        const int k = a_tile_colidx[a_lo + arel];
        const int b0 = b_tile_rowptr[k];
        len = b_tile_rowptr[k + 1] - b0;
        if (prune) {
            const unsigned acol = a_occ[a_lo + arel] & 0xFFFFu;
#pragma unroll 4
            for (int q = l; q < len; q += G) cnt += (acol & (b_occ[b0 + q] >> 16)) != 0;
        }
    }
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, G);
    if (!prune) cnt = len;
    if (in && l == 0) {
        aprod[arel] = len;
        lprod[arel] = cnt;
    }
}

// global expand (16 lanes per A tile walk B's tile row k): live products only, compacted by ballot;
// product x gets key (i - tr_lo, j), positions = global live offsets
__global__ void __launch_bounds__(256) s1_esc_expand_kernel(const long long *__restrict__ a_tile_keys, const uint32_t *__restrict__ a_occ, int a_lo,
                                                            int nA, int tr_lo, const int *__restrict__ lprod_off,
                                                            const int *__restrict__ b_tile_rowptr, const int *__restrict__ b_tile_colidx,
                                                            const uint32_t *__restrict__ b_occ, int prune, int bits_tc,
                                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, int *__restrict__ prod_a,
                                                            int *__restrict__ prod_b)
{
    const int arel = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool in = arel < nA;
    const int l = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
    int a = 0, i = 0, x0 = 0, b0 = 0, len = 0;
    unsigned acol = 0xFFFFu;
    if (in) {
        a = a_lo + arel;
        const long long ak = a_tile_keys[a];
        i = (int)(ak >> 32) - tr_lo;
        const int k = (int)(ak & 0xFFFFFFFFll);
        x0 = lprod_off[arel];
        b0 = b_tile_rowptr[k];
        len = b_tile_rowptr[k + 1] - b0;
        if (prune) acol = a_occ[a] & 0xFFFFu;
    }
    // the four 16-lane groups of a wave walk different B rows: iterate to the longest, compact per group
    int maxlen = len;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(maxlen, d, 64);
        maxlen = o > maxlen ? o : maxlen;
    }
    const uint64_t hi = (uint64_t)(unsigned)i << bits_tc;
    int run = 0;
    for (int q0 = 0; q0 < maxlen; q0 += 16) {
        const int q = q0 + l;
        const bool live = q < len && (!prune || (acol & (b_occ[b0 + q] >> 16)) != 0);
        const unsigned m16 = (unsigned)(__ballot(live) >> (16 * grp)) & 0xFFFFu;
        if (live) {
            const int x = x0 + run + __popc(m16 & ((1u << l) - 1u));
            keys[x] = hi | (uint64_t)(unsigned)b_tile_colidx[b0 + q];
            perm[x] = (uint32_t)x;
            prod_a[x] = a;
            prod_b[x] = b0 + q;
        }
        run += __popc(m16);
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
// default path, phase 1: expansion.  One 512-thread workgroup per CHUNK of S1_CH = 512 consecutive A tiles of the slice (a
// chunk may span several tile rows, a tile row several chunks).  The chunk's products -- A tile by A tile, each A tile's B tile
// row in order -- form one sequence of N_b products; thread a tables where A tile a's products end in it (block scan of the B
// row lengths), its first B tile and its occupied columns.  The eight waves then walk the sequence together, iteration k (256
// products: four trips of 64) by wave k mod 8: the A tiles a trip covers are read off the table with a scalar cursor (no search:
// v_readlane of the table window the lanes hold), the four B-side gathers go out together, every product is tested ONCE, and the
// live ones are compacted by ballot.  Where an iteration's live products go depends on how many came before it: that count is
// handed from iteration to iteration through LDS (t_cum; the wave of iteration k waits for the word iteration k - 1 leaves
// behind -- all eight waves are resident, so the chain always advances), which keeps the chunk's live products contiguous and in
// PRODUCT ORDER however the waves interleave:
//     lj[x] = tile column j of the product,  lab[x] = (A tile, B tile)
// The chunk's stretch of the list is N_b slots (ALL its products -- known after the block scan, so the block takes its place
// with ONE atomic before it has tested anything); the live products fill its front.  Per A tile the kernel leaves where its
// live products start and how many they are (aseg; read off the trips' ballots afterwards), per chunk the same (chunk_seg): a
// tile row's live products are the concatenation of one piece per chunk it touches, each piece contiguous.
// A hub -- an A tile whose B tile row holds thousands of tiles, webbase-1M's directory pages -- is thereby spread over the
// eight waves (one wave per 64 A tiles walked a chunk with three of them for 105 us, of a kernel that should take 40).
// The first block also clears the pass's status words (flags, bin populations, scalars): nothing in this kernel reads or
// sets them, and every later kernel of the pass comes after it -- the pass needs no reset launch.
// ------------------------------------------------------------------------------------------
constexpr int S1_CH = 512;               // A tiles per chunk
constexpr int S1_XW = 8;                 // waves of the expansion's workgroup (= S1_CH / 64)
constexpr int S1_XU = 4;                 // trips (64 products each) per iteration
constexpr int S1_XTRIPS = 2048;          // trips per epoch: LDS holds one ballot and one running count per trip (131 072 products; longer chunks take several epochs)
constexpr unsigned S1_SENT = 0xFFFFFFFFu;
constexpr int S1_CAP0 = 64, S1_CAP1 = 512, S1_CAP2 = 2048, S1_CAP3 = 8192, S1_CAP4 = 32768;   // live products per row of a bin
constexpr int S1_QB0 = 6, S1_QB1 = 9, S1_QB2 = 11, S1_QB3 = 13, S1_QB4 = 15;                  // ... and the bits of a key's index field
constexpr int S1_NLIST = 6;              // row lists: the five bins + the oversized rows
constexpr int S1_PC1 = 64, S1_PC2 = 256, S1_PC3 = 1024;   // pieces (chunks) per row a bin's table holds (= its threads)
constexpr int S1_SEG_T = 1024;           // keys per column-range segment of a big row (aimed at; a segment's sort holds 2048)
constexpr int S1_XLL_MAX = 1 << 18;      // rows with more live products than this take the global sort (one workgroup -- or 256 segments -- per row would take too long)

__global__ void __launch_bounds__(256) s1_total_kernel(const int *__restrict__ a_tile_colidx, int a_lo, int nA, const int *__restrict__ b_tile_rowptr,
                                                       unsigned long long *__restrict__ total)
{
    // all tile-level products of the slice (the reference's P): the capacity of the live list, first pass of a plan only
    const int arel = blockIdx.x * blockDim.x + threadIdx.x;
    long long len = 0;
    if (arel < nA) {
        const int k = a_tile_colidx[a_lo + arel];
        len = b_tile_rowptr[k + 1] - b_tile_rowptr[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) len += __shfl_xor(len, d, 64);
    // one atomic per block (one per wave -- 10 k of them on the one word -- took 128 us of a first pass: 11 ns apiece)
    __shared__ long long wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (s) atomicAdd(total, (unsigned long long)s);
    }
}

#ifdef PEM_S1_DEBUG
__device__ unsigned long long g_k1dbg[32768][8];   // per chunk: start, after the allocation, end (100 MHz), products; summed over its iterations: walk, gather, chain wait, stores
extern "C" void pem_debug_k1(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k1dbg), sizeof(unsigned long long) * 32768 * 8); }
#endif

// inclusive prefix sum over the wave's 64 lanes on the vector ALU's data-parallel-primitive paths (row shifts inside the rows of
// sixteen, then the two row broadcasts): six instructions, no LDS
__device__ __forceinline__ int s1_wave_inclusive_scan_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}

__global__ void __launch_bounds__(64 * S1_XW, 8) s1_expand_kernel(const int *__restrict__ a_tile_colidx, const uint32_t *__restrict__ a_occ, int a_lo, int nA,
                                                               const int *__restrict__ b_tile_rowptr, const int2 *__restrict__ b_colocc, int prune,
                                                               int *__restrict__ bin_count, unsigned long long cap, int2 *__restrict__ aseg,
                                                               int2 *__restrict__ chunk_seg, long long *__restrict__ chunk_n, int *__restrict__ lj,
                                                               int2 *__restrict__ lab, int *__restrict__ flags, long long *__restrict__ scalars)
{
    static_assert(S1_CH == 64 * S1_XW, "one thread per A tile of the chunk");
    __shared__ unsigned t_end[S1_CH];                 // A tile a's products are [t_end[a - 1], t_end[a]) of the chunk's sequence
    __shared__ uint2 t_pay[S1_CH];                    // (d, occupied columns of A tile a): B tile of product q of A tile a = q + d
    __shared__ unsigned t_hist[S1_XW][64 * S1_XU];    // per wave: how many A tiles end at each product of the iteration in hand
    __shared__ unsigned long long t_bal[S1_XTRIPS];   // per trip of the epoch: its live products
    __shared__ unsigned t_cum[S1_XTRIPS + 1];         // ... and the chunk's live products before it (S1_SENT: not known yet)
    __shared__ unsigned s_wsum[S1_XW];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = (int)blockIdx.x;
    if (c == 0) {
        if (tid < NUM_FLAGS) flags[tid] = 0;
        if (tid < BC_FAULT) bin_count[tid] = 0;
        if (tid < 4) scalars[tid] = 0;
    }
#ifdef PEM_S1_DEBUG
    const unsigned long long dbg0 = wall_clock64();
#endif
    const int arel = c * S1_CH + tid;
    const bool in = arel < nA;
    int b0 = 0;
    unsigned len = 0, acol = 0xFFFFu;
    if (in) {
        const unsigned ao = a_occ[a_lo + arel];                  // (requested with the tile column, not behind it)
        const int k = a_tile_colidx[a_lo + arel];
        b0 = b_tile_rowptr[k];
        len = (unsigned)(b_tile_rowptr[k + 1] - b0);
        if (prune) acol = ao & 0xFFFFu;
    }
    unsigned inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned wpre = 0, N_b = 0;                                  // (host: the slice holds fewer than 2^32 products, so 32 bits do)
#pragma unroll
    for (int w = 0; w < S1_XW; ++w) {
        if (w < wave) wpre += s_wsum[w];
        N_b += s_wsum[w];
    }
    inc += wpre;
    t_end[tid] = inc;
    t_pay[tid] = make_uint2((unsigned)(b0 - (int)(inc - len)), acol);
    if (tid == 0) s_base = N_b ? atomicAdd(reinterpret_cast<unsigned long long *>(bin_count + BC_BUMP), (unsigned long long)N_b) : 0ull;
    __syncthreads();
    // (cannot fail while A and B are what the list was sized from; if it does the chunk contributes nothing, the pass's sizes
    // come out wrong and the row classification raises FLAG_CAPACITY)
    const unsigned long long base64 = s_base;
    const bool ok = base64 + (unsigned long long)N_b <= cap;
    if (!ok && tid == 0) bin_count[BC_FAULT] = 1;
    const unsigned base = (unsigned)base64;
#ifdef PEM_S1_DEBUG
    const unsigned long long dbg1 = wall_clock64();
#endif
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned st_a = tid ? t_end[tid - 1] : 0u, en_a = inc;   // my A tile's products
    unsigned my_pos = 0, my_cnt = 0;                               // ... and, filled in epoch by epoch, where its live ones start and how many they are
    unsigned carry = 0;                                            // live products of the earlier epochs
    const unsigned N_run = ok ? N_b : 0u;
    for (unsigned e0 = 0; e0 < N_run; e0 += 64u * S1_XTRIPS) {
        const unsigned left = N_run - e0;
        const int ntr = left >= 64u * S1_XTRIPS ? S1_XTRIPS : (int)((left + 63u) >> 6);
        const int nit = (ntr + S1_XU - 1) / S1_XU;
        for (int x = tid; x <= ntr; x += 64 * S1_XW) t_cum[x] = x == 0 ? carry : S1_SENT;
        __syncthreads();
        for (int k = wave; k < nit; k += S1_XW) {
            const unsigned qb0 = e0 + 256u * (unsigned)k;
#ifdef PEM_S1_DEBUG
            const unsigned long long dt0 = wall_clock64();
#endif
            // the A tile holding product qb0 = the number of A tiles whose products end at or before it: two ballots over the table
            // (every eighth entry, then the eight entries of the group) instead of a nine-step search
            int a0;
            {
                const int c8 = __popcll(__ballot(t_end[8 * lane + 7] <= qb0));
                const int g = c8 < 64 ? c8 : 63;
                a0 = 8 * g + __popcll(__ballot(lane < 8 && t_end[8 * g + (lane & 7)] <= qb0));
            }
            // The A tile of every product of the iteration, without a search per product: the tile of product q is
            // a0 + #{tiles a' >= a0 whose products end at or before q}.  The lanes, as A TILES a0 + lane, drop a count at the position
            // their products end (if inside the iteration's 256); the lanes, as PRODUCTS, read the counts back and prefix-sum them
            // (six DPP steps per trip).  A dozen LDS operations and ~60 vector instructions per iteration however many A tiles it
            // spans (a six-step shuffle search per product made the LDS crossbar the kernel's bound, a scalar walk over the tiles
            // its instruction issue: 18 instructions per A tile and trip).
            volatile unsigned *hist = t_hist[wave];
#pragma unroll
            for (int u = 0; u < S1_XU; ++u) hist[64 * u + lane] = 0;
            for (int aw = a0; aw < S1_CH; aw += 64) {            // (one window of 64 tiles, unless more than 64 end inside these 256 products: empty B rows)
                const unsigned ve = aw + lane < S1_CH ? t_end[aw + lane] : 0xFFFFFFFFu;
                const unsigned m = ve - qb0;                     // (tiles from a0 on end beyond qb0: m >= 1)
                const bool inside = m < 64u * S1_XU;
                if (inside) __hip_atomic_fetch_add(const_cast<unsigned *>(&hist[m]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (__popcll(__ballot(inside)) < 64) break;
            }
            int bb[S1_XU], aa[S1_XU];
            unsigned ac[S1_XU];
            int before_trip = a0;                                // tiles ending at or before the trip's first product (+ a0)
#pragma unroll
            for (int u = 0; u < S1_XU; ++u) {
                const int incl = s1_wave_inclusive_scan_dpp((int)hist[64 * u + lane]);
                const int mine = before_trip + incl;
                before_trip += __builtin_amdgcn_readlane(incl, 63);
                aa[u] = mine < S1_CH ? mine : S1_CH - 1;         // (products past the chunk's end: any tile)
                const uint2 pay = t_pay[aa[u]];
                bb[u] = (int)(qb0 + 64u * (unsigned)u + (unsigned)lane) + (int)pay.x;
                ac[u] = pay.y;
            }
#ifdef PEM_S1_DEBUG
            const unsigned long long dt1 = wall_clock64();
#endif
            int2 co[S1_XU];
#pragma unroll
            for (int u = 0; u < S1_XU; ++u) co[u] = b_colocc[qb0 + 64u * (unsigned)u + (unsigned)lane < N_run ? bb[u] : 0];   // (unconditional: the four gathers go out together)
            unsigned long long bal[S1_XU];
            int cnt[S1_XU];
#pragma unroll
            for (int u = 0; u < S1_XU; ++u) {
                const bool live = qb0 + 64u * (unsigned)u + (unsigned)lane < N_run && (!prune || (ac[u] & ((unsigned)co[u].y >> 16)) != 0);
                bal[u] = __ballot(live);
                cnt[u] = __popcll(bal[u]);
            }
#ifdef PEM_S1_DEBUG
            const unsigned long long dt2 = wall_clock64();
#endif
            // the chunk's live products before this iteration: left behind by the wave of iteration k - 1
            unsigned before = 0;
            if (lane == 0) {
                while ((before = __hip_atomic_load(&t_cum[S1_XU * k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == S1_SENT) __builtin_amdgcn_s_sleep(1);
                unsigned run = before;
#pragma unroll
                for (int u = 0; u < S1_XU; ++u) {
                    if (S1_XU * k + u < ntr) {
                        t_bal[S1_XU * k + u] = bal[u];
                        run += (unsigned)cnt[u];
                        __hip_atomic_store(&t_cum[S1_XU * k + u + 1], run, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            before = (unsigned)__builtin_amdgcn_readfirstlane((int)before);
#ifdef PEM_S1_DEBUG
            const unsigned long long dt3 = wall_clock64();
#endif
            unsigned x0 = base + before;
#pragma unroll
            for (int u = 0; u < S1_XU; ++u) {
                if ((bal[u] >> lane) & 1ull) {
                    const unsigned x = x0 + (unsigned)__popcll(bal[u] & lt);
                    lj[x] = co[u].x;
                    lab[x] = make_int2(a_lo + c * S1_CH + aa[u], bb[u]);
                }
                x0 += (unsigned)cnt[u];
            }
#ifdef PEM_S1_DEBUG
            if (lane == 0 && c < 32768) {
                const unsigned long long dt4 = wall_clock64();
                atomicAdd(&g_k1dbg[c][4], dt1 - dt0);
                atomicAdd(&g_k1dbg[c][5], dt2 - dt1);
                atomicAdd(&g_k1dbg[c][6], dt3 - dt2);
                atomicAdd(&g_k1dbg[c][7], dt4 - dt3);
            }
#endif
        }
        __syncthreads();
        // my A tile's live products in this epoch, read off the trips' ballots: lp(x) = live products of the chunk before product x
        const unsigned e1 = e0 + 64u * (unsigned)ntr;
        auto lp = [&](unsigned x) {
            const unsigned r = (x - e0) >> 6, o = (x - e0) & 63u;
            return t_cum[r] + (o ? (unsigned)__popcll(t_bal[r] & ((1ull << o) - 1ull)) : 0u);
        };
        const bool last_epoch = e1 >= N_run;
        if (st_a >= e0 && (st_a < e1 || (last_epoch && st_a == e1))) my_pos = lp(st_a < e1 ? st_a : e1);
        {
            const unsigned lo = st_a > e0 ? st_a : e0, hi = en_a < e1 ? en_a : e1;
            if (hi > lo) my_cnt += lp(hi) - lp(lo);
        }
        carry = t_cum[ntr];
        __syncthreads();                                         // the epoch's words are re-armed by the next one
    }
    if (st_a > N_run) my_pos = carry;                            // (only when the chunk was skipped: every position collapses onto its start)
    if (N_run == 0) my_pos = 0;
    if (in) aseg[arel] = make_int2((int)(base + my_pos), (int)my_cnt);
    if (tid == 0) {
        chunk_seg[c] = make_int2((int)base, (int)carry);
        chunk_n[c] = (long long)N_b;
#ifdef PEM_S1_DEBUG
        if (c < 32768) {
            g_k1dbg[c][0] = dbg0;
            g_k1dbg[c][1] = dbg1;
            g_k1dbg[c][2] = wall_clock64();
            g_k1dbg[c][3] = (unsigned long long)N_b;
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------
// phase 2: per tile row, the live total (from the pieces) and the size class.  Rows are binned by LIVE products -- what gets
// sorted -- and by the number of pieces the bin's table holds: <= 64 live in at most two pieces: one wave, one key per lane
// (s1_tiny_kernel); <= 512 one wave, <= 2048 four waves (bitonic network in registers); <= 8192 and <= 32768 sixteen waves
// (stable LDS radix sort on the column bits); above that one workgroup per row with the keys in global memory.
// Also: the 64-bit total of ALL products (the reference's P, summed from the chunks), the closing words of the pass's arrays,
// step 2's group counters, and the re-arming of the live list's allocator for the next pass.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) s1_rowclass_kernel(const int *__restrict__ a_tile_rowptr, int tr_lo, int a_lo, int mt, int nchunks,
                                                          const int2 *__restrict__ aseg, const int2 *__restrict__ chunk_seg,
                                                          const long long *__restrict__ chunk_n, int tiny_ok, int cap4, int xlcap,
                                                          int *__restrict__ row_l, int4 *__restrict__ row_desc, int *__restrict__ row_list,
                                                          int *__restrict__ bin_count, int *__restrict__ xl_base, int *__restrict__ row_tc,
                                                          long long *__restrict__ scalars, int *__restrict__ flags, int *__restrict__ pairs_offset,
                                                          int *__restrict__ group_nnz, int ngroups, int *__restrict__ blk_heads, int nblk,
                                                          int seg_on, int2 *__restrict__ seg_list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nl = 0, np = 0;
    if (i < mt) {
        const int ra0 = a_tile_rowptr[tr_lo + i] - a_lo, ra1 = a_tile_rowptr[tr_lo + i + 1] - a_lo;
        unsigned pos0 = 0, pos1 = 0;
        int cnt0 = 0;
        if (ra1 > ra0) {
            const int c0 = ra0 / S1_CH, c1 = (ra1 - 1) / S1_CH;
            np = c1 - c0 + 1;
            const int2 f0 = aseg[ra0];
            pos0 = (unsigned)f0.x;
            if (np == 1) {
                const int2 e = aseg[ra1 - 1];
                cnt0 = (int)((unsigned)e.x + (unsigned)e.y - pos0);
                nl = cnt0;
            } else {
                const int2 e0 = aseg[c0 * S1_CH + S1_CH - 1], f1 = aseg[c1 * S1_CH], e1 = aseg[ra1 - 1];
                cnt0 = (int)((unsigned)e0.x + (unsigned)e0.y - pos0);
                pos1 = (unsigned)f1.x;
                long long tot = (long long)cnt0 + (long long)((unsigned)e1.x + (unsigned)e1.y - pos1);
                for (int c = c0 + 1; c < c1; ++c) tot += chunk_seg[c].y;          // (independent loads: a directory row of webbase-1M spans 74 chunks)
                nl = tot > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)tot;                  // (beyond int32 the scan of the totals reports the overflow)
            }
        }
        row_l[i] = nl;
        row_desc[i] = make_int4((int)pos0, cnt0, (int)pos1, nl);
        xl_base[i] = -1;
        row_tc[i] = 0;
    }
    int bin = -1;
    if (nl > 0) {
        bin = 5;
        if (nl <= S1_CAP0 && np <= 2 && tiny_ok) bin = 0;
        else if (nl <= S1_CAP1 && np <= S1_PC1) bin = 1;
        else if (nl <= S1_CAP2 && np <= S1_PC2) bin = 2;
        else if (nl <= S1_CAP3 && np <= S1_PC3) bin = 3;
        else if (nl <= cap4 && np <= S1_PC3) bin = 4;
        // rows above the 8192-key bin: column-range segments, one workgroup each (list 4 holds the rows, seg_list the segments)
        if (seg_on && bin >= 4) bin = (nl <= S1_XLL_MAX && np <= S1_PC2) ? 4 : 5;
        if (nl > xlcap) bin = 5;                                 // test hook: rows above xlcap live products take the oversized-row path
        if (seg_on && bin == 4) {
            const int G = (nl + S1_SEG_T - 1) / S1_SEG_T;
            const int s0 = atomicAdd(&bin_count[BC_SEGS], G);
            for (int g = 0; g < G; ++g) seg_list[s0 + g] = make_int2(i, g | (G << 16));
        }
    }
    // slots by ballot + prefix popcount inside a wave, one LDS atomic per wave and bin inside the block, ONE global
    // atomic per block and bin (order inside a bin is irrelevant)
    __shared__ int blk_cnt[S1_NLIST], blk_base[S1_NLIST];
    __shared__ long long blk_all;
    if (threadIdx.x < S1_NLIST) blk_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) blk_all = 0;
    __syncthreads();
    int wbase = 0, rank = 0;
#pragma unroll
    for (int b = 0; b < S1_NLIST; ++b) {
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
        long long wn = i < nchunks ? chunk_n[i] : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) wn += __shfl_xor(wn, d, 64);
        if (lane == 0 && wn) atomicAdd(reinterpret_cast<unsigned long long *>(&blk_all), (unsigned long long)wn);
    }
    __syncthreads();
    if (threadIdx.x < S1_NLIST) blk_base[threadIdx.x] = blk_cnt[threadIdx.x] ? atomicAdd(&bin_count[threadIdx.x], blk_cnt[threadIdx.x]) : 0;
    if (threadIdx.x == 0 && blk_all) atomicAdd(reinterpret_cast<unsigned long long *>(&scalars[3]), (unsigned long long)blk_all);
    __syncthreads();
    if (bin >= 0) row_list[(size_t)bin * mt + blk_base[bin] + wbase + rank] = i;
    if (bin == 5) {                                              // oversized rows are few
        xl_base[i] = atomicAdd(&bin_count[BC_XL_TOTAL], nl);
        atomicMax(&bin_count[BC_XL_MAX], nl);
    }
    for (int g = i; g < ngroups; g += gridDim.x * blockDim.x) group_nnz[g] = 0;   // step 2's entry counts per S2_GROUP tiles (repeat passes: size known)
    for (int g = i; g < nblk; g += gridDim.x * blockDim.x) blk_heads[g] = 0;      // first pairs per 256 pairs, counted by the row sorts (repeat passes: size known)
    if (i == 0) {
        pairs_offset[0] = 0;
        row_tc[mt] = 0;
        if (bin_count[BC_FAULT]) flags[FLAG_CAPACITY] = 1;       // a chunk found no room in the live list (see s1_expand_kernel)
        bin_count[BC_FAULT] = 0;
        *reinterpret_cast<unsigned long long *>(bin_count + BC_BUMP) = 0ull;   // the next pass's expansion starts from an empty list
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
// ------------------------------------------------------------------------------------------
// phase 3: the row sorts.  A row's live products arrive as (tile column, position in the row's live list) keys; equal tile
// columns stay in list (= product = ascending k) order because the position is part of the key, or -- the sixteen-wave bins --
// because the keys sit in list order and the radix sort on the column bits is stable.  The sorted stream is written once:
// pairs_a / pairs_b (final) and, per pair, pair_col = tile column | head << 31 (head: the first pair of a C tile); the row's
// tile count; and the number of heads in every block of 256 pairs (blk_heads, scanned afterwards), which is what lets step 2
// index C tiles densely without compacting anything.
// ------------------------------------------------------------------------------------------
// first pairs (C tiles) per 256 pairs of the stream, for step 2's dense tile index: 64 consecutive pairs starting at pair p0,
// `bal` = which of them are first pairs; they lie in at most two blocks of 256
__device__ __forceinline__ void s1_note_heads(int *__restrict__ blk_heads, const long long p0, const unsigned long long bal)
{
    if (!bal) return;
    const long long b0 = p0 >> 8;
    const int k = (int)(((b0 + 1) << 8) - p0);                   // pairs of the 64 that lie in block b0 (>= 1)
    const unsigned long long m0 = k >= 64 ? ~0ull : (1ull << k) - 1ull;
    const int c0 = __popcll(bal & m0), c1 = __popcll(bal & ~m0);
    if (c0) atomicAdd(&blk_heads[b0], c0);
    if (c1) atomicAdd(&blk_heads[b0 + 1], c1);
}

// rows of at most 64 live products in at most two pieces: one wave, one key per lane, everything in registers
__global__ void __launch_bounds__(256) s1_tiny_kernel(const int *__restrict__ row_list, int nrows_bin, const int4 *__restrict__ row_desc,
                                                      const int *__restrict__ row_lbase, const int *__restrict__ lj, const int2 *__restrict__ lab,
                                                      int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col,
                                                      int *__restrict__ blk_heads, int *__restrict__ row_tc)
{
    const int lane = threadIdx.x & 63;
    const int li = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (li >= nrows_bin) return;                                 // (wave-uniform)
    const int i = __builtin_amdgcn_readfirstlane(row_list[li]);
    const int4 d = row_desc[i];                                  // (first piece, its length, second piece, live total)
    const int lp0 = row_lbase[i], nl = d.w;
    const bool valid = lane < nl;
    unsigned key = 0xFFFFFFFFu;
    int2 ab = make_int2(0, 0);
    if (valid) {
        const unsigned src = lane < d.y ? (unsigned)d.x + (unsigned)lane : (unsigned)d.z + (unsigned)(lane - d.y);
        key = ((unsigned)lj[src] << S1_QB0) | (unsigned)lane;
        ab = lab[src];
    }
    // bitonic network over the wave's 64 keys (the padding key sorts to the end)
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)key, jj, 64);
            const bool lower = (lane & jj) == 0, up = (lane & kk) == 0;
            key = ((key < o) == (lower == up)) ? key : o;
        }
    }
    const int j = (int)(key >> S1_QB0), from = (int)(key & 63u);
    const int a = __shfl(ab.x, from, 64), b = __shfl(ab.y, from, 64);
    const int jprev = __shfl_up(j, 1, 64);
    const bool head = valid && (lane == 0 || jprev != j);        // (sorted: the nl live keys are the first nl lanes)
    const unsigned long long bal = __ballot(head);
    if (valid) {
        pairs_a[lp0 + lane] = a;
        pairs_b[lp0 + lane] = b;
        pair_col[lp0 + lane] = j | (head ? (int)0x80000000 : 0);
    }
    if (lane == 0) {
        s1_note_heads(blk_heads, lp0, bal);
        row_tc[i] = __popcll(bal);
    }
}

template <typename KeyT, int CAP, int QB, int THREADS>
struct S1Row {
    KeyT *keys;
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
            // round holds one digit (already grouped columns) -- 64 atomics on one counter serialise, so there the first
            // lane adds the round's population instead
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
// bin so the bookkeeping atomics do not serialise; [bin][slot][pieces, load, sort, emit, rows, max row, -, -]
__device__ unsigned long long g_s1dbg[4][1024][8];
__device__ unsigned long long g_s1blk[4][1024][4];   // [bin][block < 1024][start, end, HW_ID, XCC_ID] of the block's first row
extern "C" void pem_debug_s1_blocks(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s1blk), sizeof(g_s1blk)); }
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

// bins 1-4: one wave / 256 / 1024 / 1024 threads per tile row.  The row's pieces (one per chunk of A tiles it touches: where
// the piece lies in the live list, where it goes in the row's list) are tabled in LDS -- one per thread, one trip -- and the
// keys are loaded piece by piece, coalesced; the emit finds a sorted key's (A tile, B tile) through the same table.
// one tile row of a bin: the row's pieces -> keys in LDS -> sort -> emit (the LDS arrays are the calling kernel's)
template <typename KeyT, int CAP, int QB, int THREADS>
__device__ __forceinline__ void s1_rowsort_row(KeyT *const keys, unsigned *const psrc, int *const pdst, int *const wsum, unsigned *const radix_hist,
                                               const int i, const int li, const int *__restrict__ a_tile_rowptr, const int tr_lo, const int a_lo,
                                               const int2 *__restrict__ aseg, const int *__restrict__ row_lbase, const int *__restrict__ lj,
                                               const int2 *__restrict__ lab, int *__restrict__ pairs_a, int *__restrict__ pairs_b,
                                               int *__restrict__ pair_col, int *__restrict__ blk_heads, int *__restrict__ row_tc, const int key_bits)
{
    constexpr int LOGT = THREADS == 64 ? 6 : THREADS == 256 ? 8 : 10;
    constexpr int EMAX = CAP / THREADS, WAVES = THREADS / 64;
    static_assert(EMAX == 8 || EMAX == 32, "a bin sorts up to 8 (or, for the largest, 32) keys per thread");
    static_assert(CAP <= (1 << QB), "the key's index field holds every position of the row's list");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    (void)li;
    (void)LOGT;
#ifdef PEM_S1_DEBUG
    constexpr int DBG_BIN = CAP == 512 ? 0 : CAP == 2048 ? 1 : CAP == 8192 ? 2 : 3;
    unsigned long long dbg_t = 0;
#endif
#ifdef PEM_S1_DEBUG
    if (tid == 0) dbg_t = wall_clock64();      // 100 MHz
    const unsigned long long dbg_row0 = dbg_t;
#endif
    const int ra0 = a_tile_rowptr[tr_lo + i] - a_lo, ra1 = a_tile_rowptr[tr_lo + i + 1] - a_lo;
    const int lp0 = row_lbase[i], nl = row_lbase[i + 1] - lp0;
    const int c0 = ra0 / S1_CH, np = (ra1 - 1) / S1_CH - c0 + 1;       // <= THREADS: the row classification saw to that
    {
        unsigned src = 0;
        int cnt = 0;
        if (tid < np) {
            const int c = c0 + tid;
            const int first = ra0 > c * S1_CH ? ra0 : c * S1_CH, last = (ra1 < c * S1_CH + S1_CH ? ra1 : c * S1_CH + S1_CH) - 1;
            const int2 s = aseg[first], e = aseg[last];
            src = (unsigned)s.x;
            cnt = (int)((unsigned)e.x + (unsigned)e.y - src);
        }
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        int ex = inc - cnt;
        if (THREADS > 64 && np > 64) {               // (block-uniform) the pieces spill over the first wave
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < WAVES; ++w)
                if (w < wave) ex += wsum[w];
        }
        if (tid < np) {
            psrc[tid] = src;
            pdst[tid] = ex;
        }
        if (tid == 0) pdst[np] = nl;
    }
    __syncthreads();
    S1_DBG_MARK(0);
    // the keys: (tile column, position in the row's list), list order.  Every thread takes positions of the list, finds their
    // piece (a short search over the table; most rows have one piece) and loads from the live list: all the row's loads are
    // independent (a wave per piece walked a 1 200-key piece of a directory row in nineteen dependent trips)
    for (int x0 = tid; x0 < nl; x0 += 4 * THREADS) {
        unsigned src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + u * THREADS < nl ? x0 + u * THREADS : nl - 1;
            const int p = np == 1 ? 0 : s1_find_a(pdst, 0, np, x);
            src[u] = psrc[p] + (unsigned)(x - pdst[p]);
        }
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) jj[u] = lj[src[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (x0 + u * THREADS < nl) keys[x0 + u * THREADS] = (KeyT((unsigned)jj[u]) << QB) | KeyT(x0 + u * THREADS);
    }
    if constexpr (THREADS < 1024) {                  // the register sorts load THREADS * 2^e >= nl keys: pad
        int upto = THREADS;
        while (upto < nl) upto <<= 1;
        for (int x = nl + tid; x < upto; x += THREADS) keys[x] = ~KeyT(0);
    }
    __syncthreads();
    S1_DBG_MARK(1);
    S1Row<KeyT, CAP, QB, THREADS> row;
    row.keys = keys;
    if constexpr (THREADS == 1024) {
        // 16-wave bins (more than 2048 live keys): the keys sit in list order, so a STABLE radix sort on the tile
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
    S1_DBG_MARK(2);
    // Emit: one sweep over the sorted keys in segments of 64 (one wave each, round robin), no barrier -- the gathers of the
    // pairs' (A tile, B tile) are in flight together.  A pair that opens a C tile (a new tile column) carries the mark in
    // pair_col's sign bit; the marks per 256 pairs of the stream are counted for step 2's dense tile index.
    const int nseg = (nl + 63) >> 6;
    int mytiles = 0;

    // (four segments of the wave per trip, their (A tile, B tile) gathers in flight together: one gather per trip of this
    // run-time loop was a chain of up to 32 memory round trips per wave in the 32768-key bin -- 15 of its 57 us)
    for (int g0 = wave; g0 < nseg; g0 += 4 * WAVES) {
        int jv[4];
        int2 abv[4];
        bool hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = 64 * (g0 + u * WAVES) + lane;
            jv[u] = 0;
            abv[u] = make_int2(0, 0);
            hv[u] = false;
            if (s < nl) {
                const KeyT key = keys[s];
                const int idx = (int)(key & KeyT((1u << QB) - 1u));
                jv[u] = (int)(key >> QB);
                hv[u] = s == 0 || (int)(keys[s - 1] >> QB) != jv[u];
                const int p = np == 1 ? 0 : s1_find_a(pdst, 0, np, idx);
                abv[u] = lab[psrc[p] + (unsigned)(idx - pdst[p])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * WAVES;
            if (g >= nseg) break;                                        // (wave-uniform)
            const int s = 64 * g + lane;
            const unsigned long long bal = __ballot(hv[u]);
            if (s < nl) {
                pairs_a[lp0 + s] = abv[u].x;
                pairs_b[lp0 + s] = abv[u].y;
                pair_col[lp0 + s] = jv[u] | (hv[u] ? (int)0x80000000 : 0);
            }
            if (lane == 0) {
                s1_note_heads(blk_heads, (long long)lp0 + 64 * g, bal);
                mytiles += __popcll(bal);
            }
        }
    }
    // the row's C tiles
    int base = mytiles;
    if constexpr (THREADS > 64) {
        if (lane == 0) wsum[wave] = mytiles;
        __syncthreads();
        base = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) base += wsum[w];
    }
    if (tid == 0) row_tc[i] = base;
    S1_DBG_MARK(3);
#ifdef PEM_S1_DEBUG
    if (tid == 0) {
        atomicAdd(&g_s1dbg[DBG_BIN][blockIdx.x & 1023][4], 1ull);
        atomicMax(&g_s1dbg[DBG_BIN][blockIdx.x & 1023][5], dbg_t - dbg_row0);
        if (li == (int)blockIdx.x && blockIdx.x < 1024) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_s1blk[DBG_BIN][blockIdx.x][0] = dbg_row0;
            g_s1blk[DBG_BIN][blockIdx.x][1] = dbg_t;
            g_s1blk[DBG_BIN][blockIdx.x][2] = hw;
            g_s1blk[DBG_BIN][blockIdx.x][3] = xcc;
        }
    }
#endif
    __syncthreads();                                  // the tables are rebuilt by the next row
}

template <typename KeyT, int CAP, int QB, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? (CAP > 8192 || sizeof(KeyT) == 8 ? 4 : 8) : (sizeof(KeyT) == 8 ? 1 : 8))
    s1_rowsort_kernel(const int *__restrict__ row_list, int nrows_bin, const int *__restrict__ a_tile_rowptr, int tr_lo, int a_lo,
                      const int2 *__restrict__ aseg, const int *__restrict__ row_lbase, const int *__restrict__ lj, const int2 *__restrict__ lab,
                      int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col, int *__restrict__ blk_heads,
                      int *__restrict__ row_tc, int key_bits)
{
    __shared__ KeyT keys[CAP];
    __shared__ unsigned psrc[THREADS];   // piece p of the row: where it lies in the live list ...
    __shared__ int pdst[THREADS + 1];    // ... and where it goes in the row's list (exclusive scan of the piece lengths)
    __shared__ int wsum[THREADS / 64];
    __shared__ unsigned radix_hist[THREADS == 1024 ? (THREADS / 64) * 256 : 1];   // digit counters of the radix sort (16-wave bins)
    for (int li = blockIdx.x; li < nrows_bin; li += gridDim.x)
        s1_rowsort_row<KeyT, CAP, QB, THREADS>(keys, psrc, pdst, wsum, radix_hist, row_list[li], li, a_tile_rowptr, tr_lo, a_lo, aseg, row_lbase, lj, lab,
                                               pairs_a, pairs_b, pair_col, blk_heads, row_tc, key_bits);
}

// The two sixteen-wave bins in ONE launch (32-bit keys; few rows in either): the first n4 workgroups take the rows of the
// 32768-key bin, the others those of the 8192-key bin.  As two kernels one of them sat on an auxiliary stream, whose work starts
// 15-20 us after the main stream's in a replayed graph (cross-queue dependency) -- and both are ~57 us of one workgroup's
// latency on the webbase-1M stand-in, so that delay was on the critical path of step 1.  The LDS is the large bin's (one
// workgroup per CU), hence only where the 8192-key bin holds no more rows than there are CUs.
__global__ void __launch_bounds__(1024, 4)
    s1_rowsort_big_kernel(const int *__restrict__ list4, int n4, const int *__restrict__ list3, int n3, const int *__restrict__ a_tile_rowptr, int tr_lo,
                          int a_lo, const int2 *__restrict__ aseg, const int *__restrict__ row_lbase, const int *__restrict__ lj,
                          const int2 *__restrict__ lab, int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col,
                          int *__restrict__ blk_heads, int *__restrict__ row_tc, int key_bits4, int key_bits3)
{
    __shared__ uint32_t keys[S1_CAP4];
    __shared__ unsigned psrc[1024];
    __shared__ int pdst[1024 + 1];
    __shared__ int wsum[16];
    __shared__ unsigned radix_hist[16 * 256];
    const int b = blockIdx.x;
    if (b < n4)
        s1_rowsort_row<uint32_t, S1_CAP4, S1_QB4, 1024>(keys, psrc, pdst, wsum, radix_hist, list4[b], b, a_tile_rowptr, tr_lo, a_lo, aseg, row_lbase, lj, lab,
                                                        pairs_a, pairs_b, pair_col, blk_heads, row_tc, key_bits4);
    else if (b - n4 < n3)
        s1_rowsort_row<uint32_t, S1_CAP3, S1_QB3, 1024>(keys, psrc, pdst, wsum, radix_hist, list3[b - n4], b - n4, a_tile_rowptr, tr_lo, a_lo, aseg, row_lbase,
                                                        lj, lab, pairs_a, pairs_b, pair_col, blk_heads, row_tc, key_bits3);
}

// Big rows in column-range segments.  A row of more than 8192 live products sorted by ONE workgroup is a serial chain of 50-300
// microseconds (a directory page of webbase-1M: 11 k keys, 55 us; the round-2 stand-in has a hundred rows of 8-18 k keys) on a chip
// that runs hundreds of workgroups -- so the row is cut into G = ceil(nl / 1024) ranges of tile columns and every range gets a
// workgroup of its own: it reads ALL the row's keys (coalesced, a few tens of KB from L2), keeps those whose tile column lies in its
// range, in list order, and counts the keys of smaller columns -- which is where its pairs start in the row's output, no other
// segment needed -- then sorts its keys like the 2048-key bin and emits.  The first pair of a segment opens a C tile (its column is
// new), so the marks and counts that step 2 indexes C tiles by need nothing from the other segments either; the row's tile count is
// summed by atomics.  (Rows of 2049-8192 keys keep their one workgroup: cut up as well, the thousand such rows of the round-2
// stand-in -- every segment reads its whole row -- took 560 us where the 8192-key bin takes 300.)
// Measured (round 4): webbase-1M, five rows above 8192 keys: the segments take 30 us where the 32768-key bin takes 57, the step
// 8 us less (the other bins then set its length); the round-2 stand-in, 104 such rows: step 1 0.61 ms against 0.50 -- 1 300
// workgroups each reading a 12 k-key row.  So the option (PEM_OPT_S1_SEGMENTS) is OFF by default.  Ranges are cut evenly over B's tile columns; one that holds more than 2048 keys is halved until it fits, a
// single column with more than 2048 keys (a C tile of that many pairs) is emitted as it stands -- one column needs no sort.
template <typename KeyT>
__global__ void __launch_bounds__(1024) s1_rowseg_kernel(const int2 *__restrict__ seg_list, int nsegs, const int *__restrict__ a_tile_rowptr, int tr_lo,
                                                        int a_lo, const int2 *__restrict__ aseg, const int *__restrict__ row_lbase,
                                                        const int *__restrict__ lj, const int2 *__restrict__ lab, int tile_cols,
                                                        int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col,
                                                        int *__restrict__ blk_heads, int *__restrict__ row_tc)
{
    // (sixteen waves: every segment reads its whole row, and with four waves a directory row's 12 k keys were two dozen dependent
    // round trips per wave -- 60 us for a segment whose sort takes four)
    constexpr int CAP = S1_CAP2, QB = S1_QB2, THREADS = 1024, WAVES = 16, LOGT = 10;
    __shared__ KeyT keys[CAP];
    __shared__ unsigned gidx[CAP];       // position in the row's list of the key at every position of the segment's list
    __shared__ unsigned psrc[S1_PC2];
    __shared__ int pdst[S1_PC2 + 1];
    __shared__ int wcnt[WAVES], wlow[WAVES], wsum[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int si = blockIdx.x; si < nsegs; si += gridDim.x) {
        const int2 sd = seg_list[si];
        const int i = sd.x, g = sd.y & 0xFFFF, G = sd.y >> 16;
        const int ra0 = a_tile_rowptr[tr_lo + i] - a_lo, ra1 = a_tile_rowptr[tr_lo + i + 1] - a_lo;
        const int lp0 = row_lbase[i], nl = row_lbase[i + 1] - lp0;
        const int c0 = ra0 / S1_CH, np = (ra1 - 1) / S1_CH - c0 + 1;       // <= THREADS: the row classification saw to that
        {
            unsigned src = 0;
            int cnt = 0;
            if (tid < np) {
                const int c = c0 + tid;
                const int first = ra0 > c * S1_CH ? ra0 : c * S1_CH, last = (ra1 < c * S1_CH + S1_CH ? ra1 : c * S1_CH + S1_CH) - 1;
                const int2 s = aseg[first], e = aseg[last];
                src = (unsigned)s.x;
                cnt = (int)((unsigned)e.x + (unsigned)e.y - src);
            }
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            int ex = inc - cnt;
            if (np > 64) {                               // (block-uniform)
                if (lane == 63) wsum[wave] = inc;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < WAVES; ++w)
                    if (w < wave) ex += wsum[w];
            }
            if (tid < np) {
                psrc[tid] = src;
                pdst[tid] = ex;
            }
            if (tid == 0) pdst[np] = nl;
        }
        __syncthreads();
        auto list_src = [&](const int x) {               // where position x of the row's list lies in the live list
            const int p = np == 1 ? 0 : s1_find_a(pdst, 0, np, x);
            return psrc[p] + (unsigned)(x - pdst[p]);
        };
        // every wave scans one contiguous sixteenth of the row's list
        const int per = (((nl + WAVES - 1) / WAVES) + 63) & ~63;
        const int x_lo = wave * per < nl ? wave * per : nl, x_hi = x_lo + per < nl ? x_lo + per : nl;
        int tiles = 0;
        const long long seg_lo = (long long)tile_cols * g / G, seg_hi = (long long)tile_cols * (g + 1) / G;
        long long r_lo = seg_lo, r_hi = seg_hi;
        while (r_lo < seg_hi) {                          // (block-uniform) ranges of the segment, normally the one
            // keys of smaller columns, keys inside the range: counted per wave over its quarter
            int low = 0, cnt = 0;
            for (int x0 = x_lo; x0 < x_hi; x0 += 256) {
                int cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cc[u] = lj[list_src(x0 + 64 * u + lane < x_hi ? x0 + 64 * u + lane : x_lo)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (x0 + 64 * u + lane < x_hi) {
                        low += cc[u] < r_lo;
                        cnt += cc[u] >= r_lo && cc[u] < r_hi;
                    }
                }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                low += __shfl_xor(low, d, 64);
                cnt += __shfl_xor(cnt, d, 64);
            }
            __syncthreads();                             // (the counters of the range before are read)
            if (lane == 0) {
                wcnt[wave] = cnt;
                wlow[wave] = low;
            }
            __syncthreads();
            int below = 0, total = 0, wbase = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                below += wlow[w];
                if (w < wave) wbase += wcnt[w];
                total += wcnt[w];
            }
            if (total > CAP && r_hi - r_lo > 1) {        // too many keys for one sort: the lower half of the range first
                r_hi = r_lo + (r_hi - r_lo) / 2;
                continue;
            }
            if (total > 0) {
                const bool one_col = total > CAP;        // a single tile column: its keys, in list order, ARE the sorted stream
                // the range's keys, in list order: positions from the per-wave counts, ballot ranks inside a wave
                int run = wbase;
                for (int x0 = x_lo; x0 < x_hi; x0 += 256) {       // four trips' loads in flight together
                    unsigned ss[4];
                    int cc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ss[u] = list_src(x0 + 64 * u + lane < x_hi ? x0 + 64 * u + lane : x_lo);
#pragma unroll
                    for (int u = 0; u < 4; ++u) cc[u] = lj[ss[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int x = x0 + 64 * u + lane;
                        const bool in = x < x_hi && cc[u] >= r_lo && cc[u] < r_hi;
                        const unsigned long long bal = __ballot(in);
                        if (in) {
                            const int pos = run + __popcll(bal & lt);
                            if (one_col) {
                                const int2 ab = lab[ss[u]];
                                const long long o = (long long)lp0 + below + pos;
                                pairs_a[o] = ab.x;
                                pairs_b[o] = ab.y;
                                pair_col[o] = cc[u] | (pos == 0 ? (int)0x80000000 : 0);
                            } else {
                                keys[pos] = (KeyT((unsigned)cc[u]) << QB) | KeyT(pos);
                                gidx[pos] = (unsigned)x;
                            }
                        }
                        run += __popcll(bal);
                    }
                }
                if (one_col) {
                    if (tid == 0) {
                        atomicAdd(&blk_heads[((long long)lp0 + below) >> 8], 1);
                        ++tiles;
                    }
                } else {
                    int upto = THREADS;
                    while (upto < total) upto <<= 1;
                    for (int x = total + tid; x < upto; x += THREADS) keys[x] = ~KeyT(0);
                    __syncthreads();
                    {
                        S1Row<KeyT, CAP, QB, THREADS> row;
                        row.keys = keys;
                        if (total <= THREADS)
                            row.template sort_regs<1, LOGT>(tid);
                        else
                            row.template sort_regs<2, LOGT>(tid);
                    }
                    const int nseg64 = (total + 63) >> 6;
                    for (int q = wave; q < nseg64; q += WAVES) {
                        const int s = 64 * q + lane;
                        const bool valid = s < total;
                        int j = 0;
                        int2 ab = make_int2(0, 0);
                        bool head = false;
                        if (valid) {
                            const KeyT key = keys[s];
                            j = (int)(key >> QB);
                            head = s == 0 || (int)(keys[s - 1] >> QB) != j;
                            ab = lab[list_src((int)gidx[(int)(key & KeyT((1u << QB) - 1u))])];
                        }
                        const unsigned long long bal = __ballot(head);
                        const long long o = (long long)lp0 + below + s;
                        if (valid) {
                            pairs_a[o] = ab.x;
                            pairs_b[o] = ab.y;
                            pair_col[o] = j | (head ? (int)0x80000000 : 0);
                        }
                        if (lane == 0) {
                            s1_note_heads(blk_heads, (long long)lp0 + below + 64 * q, bal);
                            tiles += __popcll(bal);
                        }
                    }
                }
            }
            r_lo = r_hi;
            r_hi = seg_hi;
        }
        if (lane == 0 && tiles) atomicAdd(&row_tc[i], tiles);
        __syncthreads();                                 // the tables are rebuilt by the next segment
    }
}

// Oversized rows (above the largest LDS bin, or more pieces than a bin's table holds): the row's pieces are copied into the
// row's own stretch of the 64-bit key buffer -- key (tile column, slot) for the per-row sort, (row, tile column) for the global
// one -- with the (A tile, B tile) of every product beside it; any number of pieces, 1024 per trip.
__global__ void __launch_bounds__(1024) s1_xl_gather_kernel(const int *__restrict__ xl_rows, int nrows_xl, const int *__restrict__ xl_base,
                                                            const int *__restrict__ a_tile_rowptr, int tr_lo, int a_lo,
                                                            const int2 *__restrict__ aseg, const int *__restrict__ lj, const int2 *__restrict__ lab,
                                                            int bits_tc, int local_keys, uint64_t *__restrict__ keys, uint32_t *__restrict__ perm,
                                                            int *__restrict__ prod_a, int *__restrict__ prod_b)
{
    constexpr int WAVES = 16;
    __shared__ unsigned psrc[1024];
    __shared__ int pdst[1025];
    __shared__ int wsum[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int li = blockIdx.x; li < nrows_xl; li += gridDim.x) {
        const int i = xl_rows[li];
        const int ra0 = a_tile_rowptr[tr_lo + i] - a_lo, ra1 = a_tile_rowptr[tr_lo + i + 1] - a_lo;
        const int c0 = ra0 / S1_CH, c1 = (ra1 - 1) / S1_CH;
        int run = xl_base[i];                             // (block-uniform) next slot of the row's stretch
        for (int pc0 = c0; pc0 <= c1; pc0 += 1024) {
            const int nb = c1 - pc0 + 1 < 1024 ? c1 - pc0 + 1 : 1024;
            unsigned src = 0;
            int cnt = 0;
            if (tid < nb) {
                const int c = pc0 + tid;
                const int first = ra0 > c * S1_CH ? ra0 : c * S1_CH, last = (ra1 < c * S1_CH + S1_CH ? ra1 : c * S1_CH + S1_CH) - 1;
                const int2 s = aseg[first], e = aseg[last];
                src = (unsigned)s.x;
                cnt = (int)((unsigned)e.x + (unsigned)e.y - src);
            }
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int ex = inc - cnt, tot = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                if (w < wave) ex += wsum[w];
                tot += wsum[w];
            }
            if (tid < nb) {
                psrc[tid] = src;
                pdst[tid] = run + ex;
            }
            if (tid == 0) pdst[nb] = run + tot;
            __syncthreads();
            for (int p = wave; p < nb; p += WAVES) {
                const unsigned s = psrc[p];
                const int d0 = pdst[p], n = pdst[p + 1] - d0;
                for (int t = lane; t < n; t += 64) {
                    const int x = d0 + t;
                    const unsigned col = (unsigned)lj[s + (unsigned)t];
                    const int2 ab = lab[s + (unsigned)t];
                    keys[x] = local_keys ? ((uint64_t)col << 32) | (uint64_t)(unsigned)x : ((uint64_t)(unsigned)i << bits_tc) | (uint64_t)col;
                    perm[x] = (uint32_t)x;
                    prod_a[x] = ab.x;
                    prod_b[x] = ab.y;
                }
            }
            run += tot;
            __syncthreads();                              // the tables are rebuilt by the next trip
        }
    }
}

// oversized rows, global form: s1_xl_gather_kernel + radix sort on (row, tile column) + emit
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
                                  int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col, int *__restrict__ blk_heads,
                                  int *__restrict__ row_tc)
{
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    uint64_t key = keys[x];
    int i = (int)(key >> bits_tc), j = (int)(key & ((1ull << bits_tc) - 1ull));
    int rs = xl_rowstart[i];
    int s = (int)x - rs;
    int p0 = row_lbase[i], ni = row_lbase[i + 1] - p0;     // the row's live-product range
    uint32_t o = perm[x];
    pairs_a[p0 + s] = prod_a[o];
    pairs_b[p0 + s] = prod_b[o];
    const bool head = headx[x + 1] != headx[x];            // (heads mark the FIRST key of every (row, tile column) run: see s1_heads_kernel)
    pair_col[p0 + s] = j | (head ? (int)0x80000000 : 0);
    if (head) atomicAdd(&blk_heads[(p0 + s) >> 8], 1);
    if (s == 0) row_tc[i] = headx[rs + ni] - headx[rs];
}

// Oversized rows, one workgroup per row.  The global form sorts all oversized rows' products together: four radix passes over
// (row, tile column) keys, each a histogram launch, a scan and a scatter launch, then heads, a scan, row starts and the emit --
// nineteen launches.  But s1_xl_gather_kernel has put every such row's live products into the row's OWN stretch of the key
// buffer, in list order.  So each row is sorted where it lies by one 1024-thread workgroup: a stable LSD radix sort on the
// tile-column bits with the keys in global memory (L2-resident: a row is a few hundred KB) -- per-wave digit histograms in LDS,
// one scan of the 16 x 256 counters, ballot-ranked scatter, as in the 16-wave LDS bins -- followed by the same emit as
// s1_rowsort_kernel.  One launch.
__global__ void __launch_bounds__(1024) s1_xl_rowsort_kernel(const int *__restrict__ xl_rows, int nrows_xl, const int *__restrict__ xl_base,
                                                             const int *__restrict__ row_lbase, uint64_t *k0, uint64_t *k1, int bits_tc,
                                                             const int *__restrict__ prod_a, const int *__restrict__ prod_b,
                                                             int *__restrict__ pairs_a, int *__restrict__ pairs_b, int *__restrict__ pair_col,
                                                             int *__restrict__ blk_heads, int *__restrict__ row_tc)
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
        // emit (as s1_rowsort_kernel): sorted pairs, the first pair of every C tile marked
        int mytiles = 0;
        for (int s0 = 64 * wave; s0 < n; s0 += 1024) {
            const int sidx = s0 + lane;
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
            if (valid) {
                pairs_a[lp0 + sidx] = a;
                pairs_b[lp0 + sidx] = b;
                pair_col[lp0 + sidx] = j | (head ? (int)0x80000000 : 0);
            }
            if (lane == 0) {
                s1_note_heads(blk_heads, (long long)lp0 + s0, bal);
                mytiles += __popcll(bal);
            }
        }
        if (lane == 0) wsum[wave] = mytiles;
        __syncthreads();
        if (tid == 0) {
            int tiles = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) tiles += wsum[w];
            row_tc[i] = tiles;
        }
        __syncthreads();
    }
}

// pair stream -> reference layout (_C_tileColIdx, spgemm.cu:379; pair offsets :484) without step 2: the step-wise API after
// step 1, and the 16-lanes-per-tile baseline kernels.  One workgroup per 256 pairs: a tile's dense index is the scanned count of
// first pairs in front of the block + those in front of it inside the block.
__global__ void __launch_bounds__(256) s1_compact_kernel(const int *__restrict__ pair_col, long long npairs, const int *__restrict__ blk_base,
                                                         long long ntc, int *__restrict__ c_colidx, int *__restrict__ pairs_offset)
{
    __shared__ int wcnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const int col = p < npairs ? pair_col[p] : 0;
    const bool head = col < 0;
    const unsigned long long bal = __ballot(head);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    long long t = (long long)blk_base[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) t += wcnt[w];
    if (head) {
        c_colidx[t] = col & 0x7FFFFFFF;
        pairs_offset[t] = (int)p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pairs_offset[ntc] = (int)npairs;
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
                   B->tile_rowptr.as<int>(), B->tile_occ.as<uint32_t>(), prune, p->aprod_off.as<int>(), p->lprod_off.as<int>());
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
        PEM_LAUNCH(ctx, s1_esc_expand_kernel, grid_for((size_t)nA * 16, 256), 256, A->tile_keys.as<long long>(), A->tile_occ.as<uint32_t>(), p->a_lo,
                   nA, p->tr_lo, p->lprod_off.as<int>(), B->tile_rowptr.as<int>(), B->tile_colidx.as<int>(), B->tile_occ.as<uint32_t>(), prune,
                   bits_tc, p->sk0.as<uint64_t>(), p->sv0.as<uint32_t>(), p->prod_a.as<int>(), p->prod_b.as<int>());
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
// The row bins and the oversized-row chain run concurrently on FOUR streams -- the main one and three auxiliary ones, forked
// behind the row classification and joined before the row-count scan.  Four, because the runtime feeds four hardware queues: a
// fifth stream shares a queue with one of the others, and in round 4's first cut the one-key-per-lane kernel (62 k waves, the
// bulk of the work) landed behind the 32768-key bin (five workgroups, 57 us) on the main stream's queue.  The plan, by what
// each kernel takes alone on webbase-1M:
//     main: 32768-key bin (57 us; or its rows' segments, PEM_OPT_S1_SEGMENTS)     aux 0: tiny rows (33), then the 512-key bin (21)
//     aux 1: 8192-key bin (38)                                                     aux 2: 2048-key bin (28), then the oversized rows
// The 32768-key bin goes FIRST and on the main stream: a workgroup of it needs a CU's whole LDS, so it can only start on a CU that
// holds no other LDS-using workgroup; dispatched the moment the row classification retires -- the forked streams get through their
// event waits ~14 us later -- its workgroups are placed before the other bins' reach the CUs (behind them the hundred such rows of
// the round-2 stand-in waited for CUs to drain: 340 us).
struct S1Lanes {
    pem_ctx *ctx;
    hipStream_t main_stream;
    bool serial, forked[3] = {false, false, false};
    S1Lanes(pem_ctx *c, bool serial_) : ctx(c), main_stream(c->stream), serial(serial_) { (void)hipEventRecord(c->ev_fork, c->stream); }
    void on(int lane)                      // 0: main stream; 1..3: auxiliary stream lane - 1
    {
        if (lane == 0 || serial) {
            ctx->stream = main_stream;
            return;
        }
        if (!forked[lane - 1]) {
            (void)hipStreamWaitEvent(ctx->aux[lane - 1], ctx->ev_fork, 0);
            forked[lane - 1] = true;
        }
        ctx->stream = ctx->aux[lane - 1];
    }
    void join()
    {
        ctx->stream = main_stream;
        for (int k = 0; k < 3; ++k)
            if (forked[k]) {
                (void)hipEventRecord(ctx->ev_join[k], ctx->aux[k]);
                (void)hipStreamWaitEvent(main_stream, ctx->ev_join[k], 0);
            }
    }
};

static void launch_rowsorts(pem_ctx *ctx, pem_cplan *p, S1Lanes &lanes, const int *counts, int mt, int bits_tc, bool force64, int nsegs)
{
    const pem_tiled *A = p->A;
    int *rl = p->row_list.as<int>();

#define PEM_ROWSORT_ARGS(BIN, QB)                                                                                                        \
    rl + (size_t)(BIN) * mt, counts[BIN], A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, p->aseg.as<int2>(), p->row_lbase.as<int>(),       \
        p->live_j.as<int>(), p->live_ab.as<int2>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(),                   \
        p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>(), (QB) + bits_tc
#define PEM_ROWSORT(BIN, CAP, QB, THREADS)                                                                                               \
    do {                                                                                                                                 \
        if (bits_tc + (QB) <= 32 && !force64)                                                                                            \
            PEM_LAUNCH_NAMED(ctx, "s1_rowsort_kernel<" #CAP ">", (s1_rowsort_kernel<uint32_t, CAP, QB, THREADS>), counts[BIN], THREADS,    \
                             PEM_ROWSORT_ARGS(BIN, QB));                                                                                 \
        else                                                                                                                             \
            PEM_LAUNCH_NAMED(ctx, "s1_rowsort_kernel<" #CAP ",key64>", (s1_rowsort_kernel<uint64_t, CAP, QB, THREADS>), counts[BIN],      \
                             THREADS, PEM_ROWSORT_ARGS(BIN, QB));                                                                        \
    } while (0)
    const int ncu = ctx->cu_count > 0 ? ctx->cu_count : 256;
    const bool merge_big = !p->opt_s1_segments && !p->opt_s1_serial && counts[4] > 0 && counts[3] > 0 && counts[3] + counts[4] <= ncu &&
                           bits_tc + S1_QB4 <= 32 && !force64;
    if (p->opt_s1_segments) {
        if (nsegs > 0) {                   // rows above the 8192-key bin, one workgroup per column-range segment
            lanes.on(0);
            if (bits_tc + S1_QB2 <= 32 && !force64)
                PEM_LAUNCH_NAMED(ctx, "s1_rowseg_kernel", (s1_rowseg_kernel<uint32_t>), nsegs, 1024, p->seg_list.as<int2>(), nsegs, A->tile_rowptr.as<int>(),
                                 p->tr_lo, p->a_lo, p->aseg.as<int2>(), p->row_lbase.as<int>(), p->live_j.as<int>(), p->live_ab.as<int2>(),
                                 p->B->tile_cols, p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(), p->blk_heads.as<int>(),
                                 p->c_tile_rowptr.as<int>());
            else
                PEM_LAUNCH_NAMED(ctx, "s1_rowseg_kernel<key64>", (s1_rowseg_kernel<uint64_t>), nsegs, 1024, p->seg_list.as<int2>(), nsegs,
                                 A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, p->aseg.as<int2>(), p->row_lbase.as<int>(), p->live_j.as<int>(),
                                 p->live_ab.as<int2>(), p->B->tile_cols, p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(),
                                 p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>());
        }
    } else if (merge_big) {                // both sixteen-wave bins, few rows each: one launch on the main stream (see the kernel)
        lanes.on(0);
        PEM_LAUNCH_NAMED(ctx, "s1_rowsort_big_kernel", s1_rowsort_big_kernel, counts[4] + counts[3], 1024, rl + (size_t)4 * mt, counts[4],
                         rl + (size_t)3 * mt, counts[3], A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, p->aseg.as<int2>(), p->row_lbase.as<int>(),
                         p->live_j.as<int>(), p->live_ab.as<int2>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(),
                         p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>(), S1_QB4 + bits_tc, S1_QB3 + bits_tc);
    } else if (counts[4] > 0) {            // (only populated where 32-bit keys hold a 15-bit index: see the row classification's cap4)
        lanes.on(0);
        PEM_LAUNCH_NAMED(ctx, "s1_rowsort_kernel<32768>", (s1_rowsort_kernel<uint32_t, S1_CAP4, S1_QB4, 1024>), counts[4], 1024,
                         PEM_ROWSORT_ARGS(4, S1_QB4));
    }
    if (counts[3] > 0 && !merge_big) {
        lanes.on(2);
        PEM_ROWSORT(3, 8192, S1_QB3, 1024);
    }
    if (counts[2] > 0) {
        lanes.on(3);
        PEM_ROWSORT(2, 2048, S1_QB2, 256);
    }
    if (counts[0] > 0) {
        lanes.on(1);
        PEM_LAUNCH(ctx, s1_tiny_kernel, grid_for((size_t)counts[0] * 64, 256), 256, rl, counts[0], p->row_desc.as<int4>(), p->row_lbase.as<int>(),
                   p->live_j.as<int>(), p->live_ab.as<int2>(), p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(),
                   p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>());
    }
    if (counts[1] > 0) {
        lanes.on(merge_big ? 2 : 1);       // (the 8192-key bin's stream is free then: the one-wave bins run beside each other)
        PEM_ROWSORT(1, 512, S1_QB1, 64);
    }
    lanes.on(0);
#undef PEM_ROWSORT
#undef PEM_ROWSORT_ARGS
}

static pem_status step1_rows_impl(pem_ctx *ctx, pem_cplan *p)
{
    const pem_tiled *A = p->A, *B = p->B;
    hipStream_t st = ctx->stream;
    const int nA = p->a_hi - p->a_lo, mt = p->tr_hi - p->tr_lo;
    const int nchunks = (nA + S1_CH - 1) / S1_CH;
    const int bits_tc = bits_for((uint64_t)B->tile_cols), bits_row = bits_for((uint64_t)(mt > 0 ? mt : 1));
    // 32-bit keys = tile column + the bin's index bits (6 .. 15); where they do not fit (B with more than 2^17 tile columns in the
    // 32768-key bin, 2^19 in the 8192-key one, ...) the bin sorts 64-bit keys, and the 32768-key bin -- 256 KB of them -- is left out
    const bool force64 = p->opt_key64 != 0;
    const int tiny_ok = bits_tc + S1_QB0 <= 32 && !force64;
    const int cap4 = (bits_tc + S1_QB4 <= 32 && !force64) ? S1_CAP4 : 0;
    const int xlcap = p->opt_xlcap > 0 ? p->opt_xlcap : 0x7FFFFFFF;   // test hook: rows with more live products take the oversized-row path
    p->state = 0;
    p->pairs_ready = false;
    p->c_rowidx_valid = false;
    p->compact_valid = false;
    p->ntiles_c = p->npairs = p->nnz_c = 0;
    if (!ctx->capturing) PEM_HIP(hipEventRecord(ctx->ev[0], st));
    PEM_TRY(p->c_tile_rowptr.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->row_list.reserve(sizeof(int) * (S1_NLIST * (size_t)mt + 4)));
    PEM_TRY(p->bin_count.reserve(sizeof(int) * BC_INTS));
    PEM_TRY(p->xl_base.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->pairs_offset.reserve(sizeof(int) * 4));
    PEM_TRY(p->row_lbase.reserve(sizeof(int) * ((size_t)mt + 4)));
    PEM_TRY(p->row_desc.reserve(sizeof(int4) * ((size_t)mt + 1)));
    PEM_TRY(p->aseg.reserve(sizeof(int2) * ((size_t)nA + 4)));
    PEM_TRY(p->chunk_seg.reserve(sizeof(int2) * ((size_t)nchunks + 1)));
    PEM_TRY(p->chunk_n.reserve(sizeof(long long) * ((size_t)nchunks + 1)));
    // The live list holds one slot per tile-level product of the slice (a wave of the expansion takes its chunk's slots before
    // it has tested anything, see s1_expand_kernel); how many that is is counted on the first pass of a plan.  That pass also
    // arms the list's allocator; every later one finds it re-armed by the row classification of the pass before.
    if (!p->warm_pass) {
        PEM_HIP(hipMemsetAsync(p->bin_count.as<int>() + BC_FAULT, 0, sizeof(int) * (BC_INTS - BC_FAULT), st));
        PEM_HIP(hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(int64_t), st));
        if (nA > 0)
            PEM_LAUNCH(ctx, s1_total_kernel, grid_for((size_t)nA, 256), 256, A->tile_colidx.as<int>(), p->a_lo, nA, B->tile_rowptr.as<int>(),
                       reinterpret_cast<unsigned long long *>(ctx->d_scalars + 8));
        int64_t ntotal = 0;
        PEM_TRY(read_scalars(ctx, ctx->d_scalars + 8, 1, &ntotal));
        if (ntotal >= 0xFFFFFFFFll - 64) {
            set_error("step 1: %lld tile-level products in this row block exceed the 32-bit positions of the live-product list; split the rows "
                      "(pem_split_tile_rows) or use PEM_OPT_STEP1_GLOBAL_SORT", (long long)ntotal);
            return PEM_E_OVERFLOW;
        }
        p->w_ntotal = ntotal;
    }
    const size_t ncap = (size_t)p->w_ntotal;
    PEM_TRY(arena_phase(ctx->arena, {{&p->live_j, sizeof(int) * (ncap + 64)}, {&p->live_ab, sizeof(int2) * (ncap + 64)}}));
    PEM_TRY(p->live_j.reserve(sizeof(int) * (ncap + 64)));
    PEM_TRY(p->live_ab.reserve(sizeof(int2) * (ncap + 64)));
    // (a repeat pass knows T_C, so the row classification also clears step 2's group counters and saves it a memset)
    int ngroups_reset = 0;
    p->group_nnz_cleared = false;
    if (p->warm_pass && p->w_TC > 0) {
        ngroups_reset = (int)((p->w_TC + S2_GROUP - 1) / S2_GROUP) + 4;
        PEM_TRY(p->group_nnz.reserve(sizeof(int) * (size_t)ngroups_reset));
        p->group_nnz_cleared = true;
    }
    const int nblk_reset = p->warm_pass ? (int)(p->w_P / 256 + 1) : 0;   // (the buffer is in place since the plan's first pass)
    // rows above the 8192-key bin are sorted in column-range segments (PEM_OPT_S1_SEGMENTS = 0: one workgroup per row, the
    // 32768-key bin); a row has at most nl / 1024 + 1 segments
    const int seg_on = p->opt_s1_segments != 0;
    PEM_TRY(p->seg_list.reserve(sizeof(int2) * (ncap / S1_SEG_T + (size_t)mt + 16)));
    const int prune = p->opt_prune;
    PEM_LAUNCH(ctx, s1_expand_kernel, (unsigned)(nchunks > 0 ? nchunks : 1), 64 * S1_XW, A->tile_colidx.as<int>(), A->tile_occ.as<uint32_t>(),
               p->a_lo, nA, B->tile_rowptr.as<int>(), B->tile_colocc.as<int2>(), prune, p->bin_count.as<int>(), (unsigned long long)ncap,
               p->aseg.as<int2>(), p->chunk_seg.as<int2>(), p->chunk_n.as<long long>(), p->live_j.as<int>(), p->live_ab.as<int2>(), ctx->d_flags,
               reinterpret_cast<long long *>(ctx->d_scalars));
    // per-row tile counts are accumulated in c_tile_rowptr and scanned in place afterwards; the rows' live totals go to row_lbase and
    // are scanned in place (row r's pairs, and its C tile slots, start at row_lbase[r])
    {
        const size_t span = std::max(std::max((size_t)mt, (size_t)nchunks), (size_t)1);
        PEM_LAUNCH(ctx, s1_rowclass_kernel, grid_for(span, 256), 256, A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, mt, nchunks, p->aseg.as<int2>(),
                   p->chunk_seg.as<int2>(), p->chunk_n.as<long long>(), tiny_ok, cap4, xlcap, p->row_lbase.as<int>(), p->row_desc.as<int4>(),
                   p->row_list.as<int>(), p->bin_count.as<int>(), p->xl_base.as<int>(), p->c_tile_rowptr.as<int>(),
                   reinterpret_cast<long long *>(ctx->d_scalars), ctx->d_flags, p->pairs_offset.as<int>(), p->group_nnz.as<int>(), ngroups_reset,
                   p->blk_heads.as<int>(), nblk_reset, seg_on, p->seg_list.as<int2>());
    }
    PEM_TRY(exclusive_scan_i32(ctx, p->row_lbase.as<int>(), p->row_lbase.as<int>(), (size_t)mt, ctx->d_scalars));
    // one read-back: P, the bin populations and the product total of the oversized rows
    int64_t P = 0, Pall = 0;
    int counts[5];
    size_t n_xl;
    int nrows_xl = 0, max_xl = 0, nsegs = 0;
    if (p->warm_pass) {
        nsegs = p->w_nsegs;
        max_xl = p->w_max_xl;
        P = p->w_P;
        Pall = p->w_Pall;
        for (int b = 0; b < 5; ++b) counts[b] = p->w_counts[b];
        n_xl = (size_t)p->w_nxl;
        nrows_xl = p->w_nrows_xl;
    } else {
        int *hb = reinterpret_cast<int *>(ctx->h_scalars + 32);
        PEM_HIP(hipMemcpyAsync(hb, p->bin_count.p, sizeof(int) * BC_FAULT, hipMemcpyDeviceToHost, st));
        int64_t sc[4];
        PEM_TRY(read_scalars(ctx, ctx->d_scalars, 4, sc));
        P = sc[0];
        Pall = sc[3];
        for (int b = 0; b < 5; ++b) counts[b] = p->w_counts[b] = hb[b];
        nrows_xl = p->w_nrows_xl = hb[BC_XL_ROWS];
        n_xl = (size_t)hb[BC_XL_TOTAL];
        max_xl = p->w_max_xl = hb[BC_XL_MAX];
        nsegs = p->w_nsegs = hb[BC_SEGS];
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
        const size_t nblk = n / 256 + 1;
        PEM_TRY(arena_phase(ctx->arena, {{&p->pairs_a, sizeof(int) * (n + 4)}, {&p->pairs_b, sizeof(int) * (n + 4)},
                                         {&p->pair_col, sizeof(int) * (n + 4)}, {&p->blk_heads, sizeof(int) * (nblk + 4)}}));
        PEM_TRY(p->pairs_a.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->pairs_b.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->pair_col.reserve(sizeof(int) * (n + 4)));
        PEM_TRY(p->blk_heads.reserve(sizeof(int) * (nblk + 4)));
        // (a repeat pass had the counters cleared by the row classification; a first pass learns their number only here)
        if (!p->warm_pass) PEM_HIP(hipMemsetAsync(p->blk_heads.p, 0, sizeof(int) * (nblk + 4), st));
        // Oversized rows.  Up to S1_XLL_MAX live products each they are sorted where they lie, one workgroup per row: two
        // launches with no shared scratch, so the chain runs on a stream of its own BESIDE the row bins.  Larger ones go through
        // the global sort, after the bins (it uses the context's scan and sort scratch).
        const bool xl_local = n_xl > 0 && !p->opt_xl_global && max_xl <= S1_XLL_MAX;
        if (n_xl > 0) {
            PEM_TRY(p->prod_a.reserve(sizeof(int) * n_xl));
            PEM_TRY(p->prod_b.reserve(sizeof(int) * n_xl));
            PEM_TRY(p->sk0.reserve(sizeof(uint64_t) * n_xl));
            PEM_TRY(p->sk1.reserve(sizeof(uint64_t) * n_xl));
            PEM_TRY(p->sv0.reserve(sizeof(uint32_t) * n_xl));
            PEM_TRY(p->sv1.reserve(sizeof(uint32_t) * n_xl));
            PEM_TRY(p->xl_rowstart.reserve(sizeof(int) * ((size_t)mt + 4)));
        }
        const int *xl_rows = p->row_list.as<int>() + (size_t)5 * mt;
        auto xl_gather = [&](int local) {
            PEM_LAUNCH(ctx, s1_xl_gather_kernel, (unsigned)(nrows_xl > 0 ? nrows_xl : 1), 1024, xl_rows, nrows_xl, p->xl_base.as<int>(),
                       A->tile_rowptr.as<int>(), p->tr_lo, p->a_lo, p->aseg.as<int2>(), p->live_j.as<int>(), p->live_ab.as<int2>(), bits_tc, local,
                       p->sk0.as<uint64_t>(), p->sv0.as<uint32_t>(), p->prod_a.as<int>(), p->prod_b.as<int>());
        };
        S1Lanes lanes(ctx, p->opt_s1_serial != 0);
        launch_rowsorts(ctx, p, lanes, counts, mt, bits_tc, force64, nsegs);
        if (xl_local) {
            lanes.on(3);
            xl_gather(1);
            PEM_LAUNCH(ctx, s1_xl_rowsort_kernel, (unsigned)(nrows_xl > 0 ? nrows_xl : 1), 1024, xl_rows, nrows_xl, p->xl_base.as<int>(),
                       p->row_lbase.as<int>(), p->sk0.as<uint64_t>(), p->sk1.as<uint64_t>(), bits_tc, p->prod_a.as<int>(), p->prod_b.as<int>(),
                       p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(), p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>());
        }
        lanes.join();
        if (n_xl > 0 && !xl_local) {   // stable radix sort on (row, tile col) over all oversized rows together
            xl_gather(0);
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
                       p->pairs_a.as<int>(), p->pairs_b.as<int>(), p->pair_col.as<int>(), p->blk_heads.as<int>(), p->c_tile_rowptr.as<int>());
        }
        // first pairs per 256 pairs -> C tiles in front of every 256 pairs (step 2's dense tile index); and
        // _C_rowPtr = exclusive scan of the per-row tile counts (spgemm.cu:1168), total = T_C -- one launch where both are mid-size
        PEM_TRY(exclusive_scan_i32_two(ctx, p->blk_heads.as<int>(), p->blk_heads.as<int>(), nblk, nullptr, p->c_tile_rowptr.as<int>(),
                                       p->c_tile_rowptr.as<int>(), (size_t)mt, ctx->d_scalars + 1));
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
    const size_t ntc = (size_t)p->ntiles_c, n = (size_t)p->npairs;
    PEM_ENTER(ctx);
    PEM_TRY(p->c_tile_colidx.reserve(sizeof(int) * (ntc + 4)));
    PEM_TRY(p->pairs_offset.reserve(sizeof(int) * (ntc + 4)));
    if (n > 0)
        PEM_LAUNCH(ctx, s1_compact_kernel, (unsigned)((n + 255) / 256), 256, p->pair_col.as<int>(), (long long)n, p->blk_heads.as<int>(), (long long)ntc,
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
