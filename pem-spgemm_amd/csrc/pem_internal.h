// pem_internal.h -- shared host-side plumbing of libpemspgemm_hip.so (gfx950 only).
// Replaces the reference's rmm pools / thrust allocators / 19 cudaEvents / 5 streams
// (spgemm.cu:697-758, 808-817) with: one HIP stream per context, grow-only device buffers
// owned by the handles, one pinned scalar page for size read-backs, and an event pool for
// per-kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/pem_spgemm.h"
#include "../../include/pem_test.h"

namespace pem {

void set_error(const char *fmt, ...);

#define PEM_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            pem::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PEM_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define PEM_TRY(expr)                     \
    do {                                  \
        pem_status _s = (expr);           \
        if (_s != PEM_OK) return _s;      \
    } while (0)

// top of an ABI entry point that touches the device: the context's device is current and its arena bound
#define PEM_ENTER(ctx)                        \
    PEM_HIP(hipSetDevice((ctx)->device));     \
    pem::ArenaBind _arena_bind((ctx)->arena)

// ------------------------------------------------------------------------------------------
// Device memory arena of one context.  Replaces the reference's rmm pool_memory_resource
// (sized once, spgemm.cu:808-817) and the eleven cudaMallocAsync / cudaFreeAsync of every timed
// iteration (spgemm.cu:1138-1295, 1118-1131): the driver is asked for memory in few large slabs
// (hipMalloc clears and maps VRAM, milliseconds per GB), every buffer of the context's tilings,
// plans and temporaries is carved out of them, and memory a handle gives back stays in the arena for
// the next one -- so a fresh plan on a context that has run a product of the same size allocates
// nothing, and a first plan allocates ONE slab sized from upper bounds (pem_cplan_create).
// Ordering: a context is single-caller and all its device work is ordered on its stream (auxiliary
// streams are joined before a pass ends), so a block freed by the host may be handed out again at
// once: whatever still reads it was enqueued before whatever will write it.
// ------------------------------------------------------------------------------------------
class Arena {
public:
    explicit Arena(int device_) : device(device_) {}
    ~Arena();
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    // 256-byte aligned block of at least `bytes`; nullptr when the driver refuses (pem::set_error is set)
    void *alloc(size_t bytes);
    void free(void *p);
    // make sure ONE free block of at least `bytes` exists (at most one driver allocation)
    pem_status reserve(size_t bytes);
    // a sizing phase is about to carve buffers of these sizes: whatever part of them the free list cannot serve is
    // obtained from the driver in ONE allocation
    pem_status reserve_many(const size_t *sizes, int n);
    // give wholly free slabs back to the driver
    void trim();
    struct Stats {
        size_t slab_bytes, in_use_bytes, peak_in_use_bytes, largest_free_bytes;
        long long driver_allocs, block_allocs;
    };
    Stats stats();
    const int device;
    // bumped whenever a buffer of this arena is (re)allocated or freed, or slabs are trimmed: a captured pass (hipGraph) bakes
    // buffer addresses in.  Per arena (round 4): a process-wide counter made every allocation in ANY context retire every plan's
    // captured graph.  Atomic: handles may be destroyed by another thread than their context's.
    std::atomic<unsigned long long> generation{0};

private:
    struct Slab {
        char *base;
        size_t size;
    };
    void *take(size_t bytes);                  // from the free list, best fit; nullptr if none
    pem_status add_slab(size_t bytes);
    std::mutex mu;                             // handles may be destroyed by another thread than their context's
    std::vector<Slab> slabs;
    std::map<char *, size_t> free_blocks;      // by address, coalesced
    std::map<char *, size_t> used_blocks;
    size_t slab_bytes = 0, in_use = 0, peak = 0;
    long long n_driver = 0, n_block = 0;
};

// the arena device buffers of the calling thread are taken from: bound by every ABI entry point (PEM_ENTER)
std::shared_ptr<Arena> &current_arena();
struct ArenaBind {
    std::shared_ptr<Arena> prev;
    explicit ArenaBind(const std::shared_ptr<Arena> &a) : prev(current_arena()) { current_arena() = a; }
    ~ArenaBind() { current_arena() = prev; }
};

// Grow-only device buffer carved out of the bound arena.  Contents are NOT preserved across a growth.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    std::shared_ptr<Arena> arena;    // where p came from (keeps the slabs alive until the last buffer is gone)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), arena(std::move(o.arena))
    {
        o.p = nullptr;
        o.cap = 0;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p) {
            arena->free(p);
            ++arena->generation;
        }
        p = nullptr;
        cap = 0;
        arena.reset();
    }
    pem_status reserve(size_t bytes)
    {
        if (bytes <= cap) return PEM_OK;
        release();
        const std::shared_ptr<Arena> &a = current_arena();
        if (!a) {
            set_error("internal: device buffer requested outside an ABI call (no arena bound)");
            return PEM_E_INVALID;
        }
        size_t want = (bytes + 255) & ~size_t(255);
        ++a->generation;
        p = a->alloc(want);
        if (!p) return PEM_E_NOMEM;
        arena = a;
        cap = want;
        return PEM_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// One driver allocation per sizing phase of a first pass: the buffers listed are about to be grown to these sizes.
struct PhaseWant {
    const DevBuf *buf;
    size_t bytes;
};
pem_status arena_phase(const std::shared_ptr<Arena> &arena, std::initializer_list<PhaseWant> wants);

struct KernelStat {
    std::string name;
    int64_t calls = 0;
    double total_ms = 0.0;
};

struct PendingSpan {
    int stat;
    hipEvent_t e0, e1;
};

// device-side status words, zeroed at the start of every ABI call that launches kernels
enum DevFlag { FLAG_RANGE = 0, FLAG_DUP = 1, FLAG_OVERFLOW = 2, FLAG_CAPACITY = 3, FLAG_INTERNAL = 4, NUM_FLAGS = 8 };

}  // namespace pem

struct pem_ctx {
    int device = 0;
    std::shared_ptr<pem::Arena> arena;   // every device buffer of this context's handles and temporaries
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // pinned host page for scalar read-backs (replaces the reference's racy pageable
    // cudaMemcpyAsync of _C_nnz / d_pairs_count / C_nnz)
    std::vector<hipGraphExec_t> retired_graphs;   // graph executables plans no longer use: destroyed with the context (see retire_graph)
    bool dbg_destroy_graphs = false;   // diagnostic (PEM_DEBUG_GRAPH_DESTROY=1 when the context is created): destroy them at once instead
    int cu_count = 0;                  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    int64_t *h_scalars = nullptr;      // 64 slots
    volatile int *h_flags = nullptr;   // the status flags of a repeat pass, written by the pass's checking thread (host view; slots 56..)
    int *h_flags_dev = nullptr;        // ... the device's address of the same words (null: not mappable, copy instead)
    int64_t *d_scalars = nullptr;      // 64 slots on the device
    int *d_flags = nullptr;            // NUM_FLAGS ints
    // shared temporaries (grow-only, reused by every call on this context)
    pem::DevBuf scan_bsum;             // block sums of the device scan
    pem::DevBuf scan_state;            // mid-size scan: one published total per block + completion counter
    pem::DevBuf sort_hist;             // radix-sort histograms
    bool graph_replay = false;         // pem_set_graph_replay: repeat passes of pem_spgemm are replayed as one hipGraph
    bool capturing = false;            // a warm pass is being captured into a hipGraph: no timing events, no syncs
    bool chain_events = false;         // pem_spgemm: steps run back to back, boundary events are shared
    pem::DevBuf tmp[12];               // step/convert temporaries, see call sites
    int dbg_scan_force = 0;            // pem_debug_scan_i32: 0 automatic, 2 single-launch chained scan, 3 three-launch scan
    int dbg_scan_stall_ticket = -1;    // ... and the chained scan's block that stalls before it publishes
    // timing
    hipEvent_t ev[8] = {};             // step spans
    // fork/join inside a step: an independent long-tailed kernel runs on an auxiliary stream
    hipStream_t aux[3] = {};           // step 1's row bins and its oversized-row chain: with the main stream, the runtime's four hardware queues
    hipEvent_t ev_fork = nullptr, ev_join[3] = {};
    pem_timings timings = {};
    bool profiling = false;
    hipError_t launch_err = hipSuccess;   // first kernel launch the runtime refused since the last report (PEM_LAUNCH)
    const char *launch_name = nullptr;
    std::vector<pem::KernelStat> stats;
    std::vector<pem::PendingSpan> pending;
    std::vector<hipEvent_t> event_pool;
};

namespace pem {

// RAII kernel span: records two events around a launch when profiling is on.
struct KernelSpan {
    pem_ctx *ctx;
    int stat = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    KernelSpan(pem_ctx *c, const char *name);
    ~KernelSpan();
};
pem_status resolve_kernel_spans(pem_ctx *ctx);

// A launch the runtime refuses (a grid beyond its limits, an invalid configuration) would otherwise go unnoticed and the pass
// carry on with arrays nobody wrote: the first such error is kept in the context and reported -- PEM_E_HIP, with the kernel's
// name -- by the call's next synchronisation point (pem::launch_status; read_flags does it, so does the end of every pass).
#define PEM_LAUNCH(ctx, kernel, grid, block, ...)                                              \
    do {                                                                                       \
        pem::KernelSpan _span((ctx), #kernel);                                                 \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__);    \
        pem::note_launch((ctx), #kernel);                                                      \
    } while (0)

#define PEM_LAUNCH_NAMED(ctx, name, kernel, grid, block, ...)                                  \
    do {                                                                                       \
        pem::KernelSpan _span((ctx), name);                                                    \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__);    \
        pem::note_launch((ctx), name);                                                         \
    } while (0)

inline void note_launch(pem_ctx *ctx, const char *name)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && ctx->launch_err == hipSuccess) {
        ctx->launch_err = e;
        ctx->launch_name = name;
    }
}

inline unsigned grid_for(size_t n, unsigned per_block)
{
    size_t g = (n + per_block - 1) / per_block;
    return g == 0 ? 1u : (unsigned)g;
}

inline int bits_for(uint64_t count)   // bits needed to represent values in [0, count)
{
    int b = 0;
    while (b < 63 && (uint64_t(1) << b) < count) ++b;
    return b < 1 ? 1 : b;
}

// ---- device primitives (primitives.hip) -------------------------------------------------
// out[0..n) = exclusive prefix sums of in[0..n), out[n] = total.  in == out allowed.
// Both pointers 16-byte aligned.  Sets FLAG_OVERFLOW when the total exceeds INT32_MAX.
// If d_total64 != nullptr the 64-bit total is also stored there.
pem_status exclusive_scan_i32(pem_ctx *ctx, const int *in, int *out, size_t n, int64_t *d_total64);

// two independent arrays (any lengths): one launch where both are mid-size, else two
pem_status exclusive_scan_i32_two(pem_ctx *ctx, const int *in_a, int *out_a, size_t na, int64_t *d_total_a, const int *in_b, int *out_b, size_t nb,
                                  int64_t *d_total_b);

// two arrays of equal length scanned in place by one set of launches
pem_status exclusive_scan_i32_pair(pem_ctx *ctx, int *a, int *b, size_t n, int64_t *d_total_a, int64_t *d_total_b);

// Stable LSD radix sort of (key, payload) on key bits [first_bit, nbits).  Buffers ping-pong; on
// return *keys_out / *vals_out point at the buffers holding the sorted data.
pem_status radix_sort_u64_u32(pem_ctx *ctx, uint64_t *k0, uint64_t *k1, uint32_t *v0, uint32_t *v1,
                              size_t n, int nbits, uint64_t **keys_out, uint32_t **vals_out, int first_bit = 0);

pem_status zero_flags(pem_ctx *ctx);
pem_status read_flags(pem_ctx *ctx, int *host_flags /*NUM_FLAGS*/);   // synchronises the stream
pem_status launch_status(pem_ctx *ctx);            // PEM_E_HIP (once) if a kernel launch was refused since the last call
pem_status check_internal(const int *host_flags);   // PEM_E_HIP if a device primitive gave up (FLAG_INTERNAL): the arrays behind it are not valid
// copy `count` int64 scalars from device to the pinned page and synchronise
pem_status read_scalars(pem_ctx *ctx, const int64_t *d_src, int count, int64_t *host_dst);

}  // namespace pem

struct pem_tiled {
    int value_bytes = 8;              // 8: fp64 (the reference's ValueType, spgemm.cu:728), 4: fp32
    int rows = 0, cols = 0;
    int64_t nnz = 0;
    int tile_rows = 0, tile_cols = 0;
    int64_t ntiles = 0;
    double conv_ms = 0.0, conv_tile_kernel_ms = 0.0;
    pem::DevBuf tile_keys, tile_nnz_ptr, masks, rowptr, rowcolidx, vals, masks_t;
    pem::DevBuf tile_rowptr, tile_colidx, tile_colptr, tile_rowidx, tile_offsets;
    pem::DevBuf tile_rec;             // derived: uint32[16T] = masks[16t+r] | rowptr[16t+r] << 16 (one gather serves step 3)
    pem::DevBuf tile_rec_t;           // derived: uint32[16T] = masks_t[16t+c] | (entries in columns < c) << 16: the B side of step 3
    pem::DevBuf vals_t;               // derived: double[nnz], the tile's values in column-major order (read with tile_rec_t)
    pem::DevBuf tile_colocc;          // derived: int2[T] = (tile_colidx, tile_occ): step 1's expansion reads both of every B tile
    pem::DevBuf tile_occ;             // derived: uint32[T] = occupied columns (low 16 bits) | occupied rows << 16 (step-1 pruning)
    std::vector<int> h_tile_rowptr;   // host copy (tile_rows+1 ints) for plan creation / splits
};

struct pem_cplan {
    const pem_tiled *A = nullptr, *B = nullptr;
    int tr_lo = 0, tr_hi = 0;
    int a_lo = 0, a_hi = 0;            // A tile id range of the slice
    int max_row_tiles = 0;             // A tiles in the slice's longest tile row
    int state = 0;                     // 0 created, 1 step1 done, 2 step2 done, 3 step3 done
    // pem_option values (include/pem_spgemm.h); defaults come from the environment when the plan is created
    int opt_prune = 1, opt_key64 = 0, opt_xlcap = 0;
    int opt_band = 1;                                  // 0: many-pair tiles stay in the entry-per-lane kernel
    int opt_step1_esc = 0, opt_wide = 1, opt_warm = 1, opt_export_rows = 0, opt_s1_serial = 0;
    int opt_xl_global = 0;                             // oversized rows: the global (row, tile column) radix sort instead of one workgroup per row
    int opt_idx64 = 0;                                 // test hook: 64-bit addressing in the shallow step 3 whatever the sizes
    int opt_mark = 1;                                  // shallow, pruned plans: entry -> tile by marks + ballot instead of the shuffle search
    int opt_s3_xcd = 1;                                // step 3 (entry-per-lane kernels): XCD x takes the x-th contiguous eighth of C
    int opt_epw = 0;                                   // step 3: entries per wave / 256 (0: chosen from the C tiles' density)
    int opt_decode = 1;                                // shallow plans: step 3 reads (row, column) off the C masks, no Ctiles_rowColIdx on the pass
    int64_t ntiles_c = 0, npairs = 0, nnz_c = 0;
    pem::DevBuf c_tile_rowptr, c_tile_colidx;
    mutable pem::DevBuf c_tile_rowidx; // _C_tileRowIdx: on demand from c_tile_rowptr on the row-local path (no reader there)
    mutable bool c_rowidx_valid = false;
    pem::DevBuf pairs_offset, pairs_a, pairs_b;
    pem::DevBuf c_mask, c_tile_nnz_ptr, c_vals;
    mutable pem::DevBuf c_rowcolidx;   // Ctiles_rowColIdx: written by step 2 on deep plans, else on demand from c_mask (ensure_c_rowcolidx)
    mutable bool c_rowcolidx_valid = false;
    pem::DevBuf c_tile_cnt;            // uint16 per C tile: its entry count (s2_tiles_kernel -> s2_offsets_kernel), decode plans only
    bool s3_decode = false;            // this pass: step 3 reads (row, column) off the masks
    pem::DevBuf s3_chunk_tile;         // first C tile of every S3_CHUNK-entry chunk of C (written by step 2d for step 3's waves)
    mutable pem::DevBuf c_rowptr;      // Ctiles_rowPtr: materialised on demand from c_mask (nothing on the default path reads it)
    mutable bool c_rowptr_valid = false;
    // step-1 products kept for step 2 (expanded pair ids + the sorted permutation)
    pem::DevBuf prod_a, prod_b, aprod_off;
    pem::DevBuf lprod_off;             // like aprod_off, counting only products whose tiles can meet (live products)
    // (row-local step 1 keeps per-A-tile COUNTS in the two buffers above; only the global-sort path scans them)
    pem::DevBuf row_lbase;             // per tile row: live products -> exclusive scan = the row's first pair / C tile slot
    // the live list of step 1 (s1_expand_kernel): every live tile-level product once, in product order inside a 64-A-tile chunk
    pem::DevBuf live_j, live_ab;       // int[cap] tile column, int2[cap] (A tile, B tile); cap = all products of the slice (w_ntotal)
    pem::DevBuf aseg;                  // int2 per A tile of the slice: (where its live products start in the list, how many)
    pem::DevBuf chunk_seg, chunk_n;    // per chunk: int2 (start of its stretch, live products), int64 all products
    pem::DevBuf row_desc;              // int4 per tile row: (first piece, its length, second piece, live total) -- rows of one or two pieces
    int64_t w_ntotal = 0;              // tile-level products of the slice, counted on the plan's first pass: the live list's capacity
    int64_t npairs_all = 0;            // all tile-level products (the reference's P) -- npairs counts the live ones
    pem::DevBuf sk0, sk1, sv0, sv1;    // sort buffers
    uint32_t *sorted_perm = nullptr;   // points into sv0/sv1
    // row-local step 1
    pem::DevBuf row_list, bin_count, xl_base, xl_rowstart;
    pem::DevBuf pair_col;              // int per live pair (final order): tile column of its C tile, sign bit set on the first pair of every C tile
    pem::DevBuf blk_heads;             // int per 256 pairs: first pairs (C tiles) among them, then (scanned) C tiles in front of them
    bool pairs_ready = false;          // step 1 already wrote pairs_a / pairs_b
    bool flags_mirrored = false;       // ... and left the pass's status flags in ctx->h_flags
    bool verify_folded = false;        // this pass's size check ran inside s2_entries_kernel
    bool group_nnz_cleared = false;    // step 1's reset already zeroed group_nnz for this pass
    bool wide = true;                  // step 2 ran the fused kernel (step 3 then runs entry-per-lane); false: 16-lanes-per-tile baseline
    bool compact_valid = false;        // c_tile_colidx / pairs_offset hold the dense layout (else: row-local scratch, see ensure_compact)
    pem::DevBuf group_nnz;             // C entries per S2_GROUP tiles (s2_tiles_kernel -> scan -> s2_entries_kernel)
    // sizes of the last complete pass on this plan.  A and B are immutable, so a repeat pass has the same
    // sizes: it skips the three host read-backs and a device-side check compares them at the end instead.
    pem_ctx *owner = nullptr;              // the context the plan was created on
    hipGraphExec_t graph_exec = nullptr;   // PEM_GRAPH=1: the captured warm pass
    unsigned long long graph_gen = 0;      // the context arena's generation at capture time
    bool graph_failed = false;             // capture or instantiation failed once: plain launches from then on
    bool warm = false, warm_pass = false;
    int64_t w_P = 0, w_Pall = 0, w_TC = 0, w_nnz = 0;
    int w_counts[5] = {0, 0, 0, 0, 0};
    int w_nsegs = 0;                   // column-range segments of the big rows (s1_rowseg_kernel)
    pem::DevBuf seg_list;              // int2 per segment: (tile row, segment | segments of the row << 16)
    int opt_s2_transposed = 2;         // step 2's tile product from A's transposed masks: 0 never, 1 always, 2 on repeat passes of deep plans (see s2_pair_mask_t)
    int opt_s1_segments = 0;           // 1: rows above the 8192-key bin are sorted in column-range segments, one workgroup each (off by default: see s1_rowseg_kernel)
    int64_t w_nxl = 0;
    int w_nrows_xl = 0, w_max_xl = 0;
};
