/*
 * pem_spgemm.h -- C ABI of libpemspgemm_hip.so: the MI355X-native tiled SpGEMM hot path.
 *
 * The reference (stckvrflw/pem-spgemm) has no library boundary: `main()` launches its CUDA
 * kernels inline with raw device pointers (spgemm.cu:939-952, 1138-1336).  This header is
 * the boundary a maintainer would cut along those call sites; every entry point names the
 * reference lines it replaces.  Plain pointers and sizes only -- no C++/torch types.
 *
 * Conventions
 *   - every function returns pem_status (0 = ok, <0 = error); pem_last_error() gives the
 *     message of the calling thread's last failure.  The library never calls exit().
 *   - handles are opaque; the library owns all device memory behind them.  Host buffers
 *     passed in or out are caller-owned.  `*_device` variants take/return device pointers
 *     valid on the context's device and are ordered on the context's stream.
 *   - one context per GPU rank; a context is single-caller.  Distinct contexts are
 *     independent (thread-safe across contexts).
 *   - indices are int32 like the reference's (`int`), sizes crossing the ABI are int64.
 *   - tiles are 16x16, values are fp64 (spgemm.cu:727-728).
 */
#ifndef PEM_SPGEMM_H
#define PEM_SPGEMM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pem_status;
enum {
    PEM_OK = 0,
    PEM_E_INVALID = -1,     /* bad argument / shape mismatch / index out of range */
    PEM_E_DUPLICATE = -2,   /* duplicate (i,j) in the input (the reference mishandles these) */
    PEM_E_NOMEM = -3,       /* device or host allocation failed */
    PEM_E_OVERFLOW = -4,    /* a count does not fit the reference's int32 arrays */
    PEM_E_HIP = -5,         /* HIP runtime error */
    PEM_E_STATE = -6,       /* steps called out of order */
    PEM_E_NODEVICE = -7,    /* no usable GPU: the product path has no CPU fallback */
    PEM_E_IO = -8,          /* tiled-format cache: file missing, unreadable, truncated, corrupt or not a cache file */
    PEM_E_STALE = -9        /* tiled-format cache: the file is intact but was made from a different source */
};

typedef struct pem_ctx pem_ctx;
typedef struct pem_tiled pem_tiled;   /* one matrix in tiled-CSR form, usable as A or as B */
typedef struct pem_cplan pem_cplan;   /* C = A*B over a tile-row range of A: symbolic + numeric result */

const char *pem_last_error(void);
const char *pem_version(void);

/* ---- context ------------------------------------------------------------------------ */
/* Replaces the stream/event/pool set-up of spgemm.cu:730-758, 808-817.  `stream` is a
 * hipStream_t passed as void* (NULL: the context creates its own stream). */
pem_status pem_ctx_create(int device, pem_ctx **out);
pem_status pem_ctx_create_on_stream(int device, void *stream, pem_ctx **out);
pem_status pem_ctx_destroy(pem_ctx *ctx);
pem_status pem_ctx_synchronize(pem_ctx *ctx);

/* Device memory.  The reference sizes two rmm pools once (spgemm.cu:808-817) and takes the eleven per-iteration
 * buffers of its timed loop from a pool (cudaMallocAsync / cudaFreeAsync, spgemm.cu:1138-1295, 1118-1131).  Here every
 * context owns an arena: the driver is asked for memory in few large slabs, all buffers of the context's tilings, plans
 * and temporaries are carved out of them, and memory a destroyed handle gives back stays in the arena for the next one.
 * A pass on a new plan therefore makes at most one driver allocation per sizing phase (pairs / C tiles / C entries) and
 * none at all once the arena holds enough.  pem_ctx_reserve sizes the arena ahead of time (one driver allocation of
 * `bytes`, skipped if a free block that large exists); pem_ctx_trim returns wholly free slabs to the driver. */
pem_status pem_ctx_reserve(pem_ctx *ctx, int64_t bytes);
pem_status pem_ctx_trim(pem_ctx *ctx);
typedef struct {
    int64_t slab_bytes;         /* held from the driver                         */
    int64_t in_use_bytes;       /* handed out to live buffers                   */
    int64_t peak_in_use_bytes;
    int64_t largest_free_bytes; /* largest request served without the driver    */
    int64_t driver_allocs;      /* hipMalloc calls so far                       */
    int64_t block_allocs;       /* buffers carved so far                        */
} pem_memory_stats;
pem_status pem_ctx_memory_stats(pem_ctx *ctx, pem_memory_stats *out);

/* ---- a2-a7: COO / CSR -> tiled CSR ---------------------------------------------------- */
/* Replaces spgemm.cu:832-1066 (decide_which_tile, thrust sort/unique/reduce_by_key/scan,
 * COO->CSR, generate_tiles_csr, __transpose_B_mask, tile-level CSR/CSC + _B_tileOffsets).
 * transpose != 0 builds the tiling of M^T (spgemm.cu:788-792).  Every tiled matrix carries
 * both roles' metadata (A side: tile CSR; B side: transposed masks + tile CSC). */
pem_status pem_tiled_from_coo(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *I,
                              const int32_t *J, const double *V, int transpose, pem_tiled **out);
pem_status pem_tiled_from_coo_device(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *dI,
                                     const int32_t *dJ, const double *dV, int transpose, pem_tiled **out);
pem_status pem_tiled_from_csr(pem_ctx *ctx, int rows, int cols, const int32_t *rowptr,
                              const int32_t *colidx, const double *V, pem_tiled **out);
pem_status pem_tiled_destroy(pem_ctx *ctx, pem_tiled *t);

/* fp32 values.  The reference pins `ValueType = double` in main (spgemm.cu:728) although its kernels
 * are templates on it (spgemm.cu:137, 593).  The `_f32` entry points build a tiling whose values (and the values of
 * every C computed from it) are float: step 3 then runs one fmaf per product, in the same ascending-k order.  A and
 * B of one plan must share a value type; the fp64 export entry points refuse an fp32 plan and vice versa. */
pem_status pem_tiled_from_coo_f32(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *I,
                                  const int32_t *J, const float *V, int transpose, pem_tiled **out);
pem_status pem_tiled_from_coo_device_f32(pem_ctx *ctx, int rows, int cols, int64_t nnz, const int32_t *dI,
                                         const int32_t *dJ, const float *dV, int transpose, pem_tiled **out);
pem_status pem_tiled_from_csr_f32(pem_ctx *ctx, int rows, int cols, const int32_t *rowptr,
                                  const int32_t *colidx, const float *V, pem_tiled **out);

typedef struct {
    int32_t rows, cols;            /* of the tiled matrix (after transpose)           */
    int64_t nnz;
    int32_t tile_rows, tile_cols;  /* ceil(rows/16), ceil(cols/16)  spgemm.cu:840-843 */
    int64_t ntiles;                /* non-empty 16x16 tiles         spgemm.cu:871     */
    double conv_ms;                /* device time of the whole conversion             */
    double conv_tile_kernel_ms;    /* tile payload kernels only (CSV cols 5/6, spgemm.cu:938-978) */
    int32_t value_bytes;           /* 8: fp64 (the reference's ValueType), 4: fp32    */
    int32_t reserved;
} pem_tiled_info;
pem_status pem_tiled_get_info(const pem_tiled *t, pem_tiled_info *info);

/* Stage-level views (D2H copy of one array; `bytes` must equal the array's size). */
typedef enum {
    PEM_T_TILE_KEYS = 0,     /* int64[T]   (tileRow<<32)|tileCol, sorted         spgemm.cu:131-133, 869-877 */
    PEM_T_TILE_NNZ_PTR,      /* int32[T+1] perTileNnz exclusive scan             spgemm.cu:873-874 */
    PEM_T_MASKS,             /* uint16[16T] row bitmasks                         spgemm.cu:196-200 */
    PEM_T_ROWPTR,            /* uint8[16T] nnz before row r inside the tile      spgemm.cu:205-209 */
    PEM_T_ROWCOLIDX,         /* uint8[nnz] (r<<4)|c                              spgemm.cu:195, 221 */
    PEM_T_VALS,              /* double[nnz] (float[nnz] for an fp32 tiling) tile order, row-major inside a tile  spgemm.cu:220 */
    PEM_T_MASKS_T,           /* uint16[16T] transposed masks                     spgemm.cu:244-253 */
    PEM_T_TILE_ROWPTR,       /* int32[tile_rows+1]                               spgemm.cu:986-999 */
    PEM_T_TILE_COLIDX,       /* int32[T]                                         spgemm.cu:1001-1006 */
    PEM_T_TILE_COLPTR,       /* int32[tile_cols+1]                               spgemm.cu:1042-1055 */
    PEM_T_TILE_ROWIDX,       /* int32[T]                                         spgemm.cu:1056-1061 */
    PEM_T_TILE_OFFSETS       /* int32[T] CSC position -> CSR tile id             spgemm.cu:1034-1040 */
} pem_tiled_array;
pem_status pem_tiled_get_array(pem_ctx *ctx, const pem_tiled *t, pem_tiled_array which, void *host_dst, int64_t bytes);

/* ---- on-disk cache of the tiled format -------------------------------------
 * The reference parses the .mtx text and re-runs the whole conversion on every start, and its conversion clock
 * includes the parse (spgemm.cu:760-1066).  A cache file holds the sorted tile payload of one tiling -- the tile
 * list (spgemm.cu:869-877), perTileNnz (873-874), rowColIdx (195, 221) and vals (220) -- little-endian, with a
 * checksum.  Loading uploads the four arrays, checks them ON THE DEVICE (sorted distinct in-range tiles, 1..256
 * entries per tile, strictly ascending (r<<4|c) inside a tile, every entry inside rows x cols), then rebuilds the
 * masks, intra-tile row pointers, transposed masks and the tile-level CSR/CSC indices with the conversion's own
 * kernels: no parse, no sort.  A file that fails any check is rejected, never half-used.
 * `key` identifies what the tiling was made from (the caller's choice, e.g. size + mtime of the .mtx and the
 * transpose flag); load with a non-NULL `expect` returns PEM_E_STALE when the stored key differs. */
typedef struct {
    uint64_t source_size;      /* bytes of the source file                  */
    int64_t source_mtime_ns;   /* its modification time                     */
    uint32_t transpose;        /* the tiling is of the transposed source    */
    uint32_t reserved;         /* 0                                         */
} pem_cache_key;
pem_status pem_tiled_save(pem_ctx *ctx, const pem_tiled *t, const char *path, const pem_cache_key *key);
pem_status pem_tiled_load(pem_ctx *ctx, const char *path, const pem_cache_key *expect, pem_tiled **out);

/* ---- a8: flop count (spgemm.cu:1068-1079), computed on the device ------------------------ */
pem_status pem_flop_count(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, uint64_t *flop);

/* ---- a9-a13: the three steps ---------------------------------------------------------- */
/* A plan covers tile rows [tile_row_begin, tile_row_end) of A (tile_row_end < 0: all); the
 * multi-GPU row-block split gives each rank one plan over its own range, B replicated. */
pem_status pem_cplan_create(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B,
                            int32_t tile_row_begin, int32_t tile_row_end, pem_cplan **out);
pem_status pem_cplan_destroy(pem_ctx *ctx, pem_cplan *plan);

/* Plan options.  A new plan takes its defaults from the environment once, in pem_cplan_create (PEM_PRUNE=0, PEM_STEP1=esc,
 * PEM_WIDE=0, PEM_NO_WARM=1, PEM_S3_BAND=0, PEM_EXPORT=rows); after that only pem_cplan_set_option changes them -- no entry
 * point reads the environment at call time.  Changing an option sends the plan back to the last step the option does not touch
 * and makes the next pass a full (size-reading) one.  The values below are the public ones; the test hooks and tuning switches
 * of the kernels (same entry points, further values of pem_option) are declared in pem_test.h. */
typedef enum {
    PEM_OPT_PRUNE = 0,              /* 1 (default): drop tile products whose tiles cannot meet; 0: the reference's lists      */
    PEM_OPT_STEP1_GLOBAL_SORT = 1,  /* 0 (default): row-local LDS sorts; 1: global expand + radix sort (A/B baseline of step 1) */
    PEM_OPT_WIDE = 2,               /* 1 (default): fused step 2 + entry-per-lane step 3; 0: 16-lanes-per-tile baseline kernels */
    PEM_OPT_WARM = 3,               /* 1 (default): repeat passes re-use the previous pass's sizes (device-verified); 0: read back */
    PEM_OPT_S3_BAND = 4,            /* 1 (default): many-pair C tiles of deep plans go to the wave-per-tile kernel              */
    PEM_OPT_EXPORT_ROWS = 7,        /* 0 (default): balanced chunk export; 1: 16 lanes per tile row (A/B baseline)             */
    PEM_OPT__TEST_FIRST = 5,        /* (5, 6, 8-15: pem_test.h)                                                                 */
    PEM_OPT__LAST = 16
} pem_option;
pem_status pem_cplan_set_option(pem_cplan *plan, pem_option which, int64_t value);
pem_status pem_cplan_get_option(const pem_cplan *plan, pem_option which, int64_t *value);

/* step 1 (spgemm.cu:1141-1218: tile_spgemm_step1_*_spa_kernel or the NSPARSE symbolic
 * path): tile-level symbolic product -> C tile list sorted by (tile row, tile col).
 * By default products whose A tile's occupied columns miss the B tile's occupied rows are
 * dropped (they contribute nothing; the reference keeps them as empty pairs / empty C tiles).
 * The final C is identical; PEM_OPT_PRUNE = 0 reproduces the reference's lists. */
pem_status pem_spgemm_step1(pem_ctx *ctx, pem_cplan *plan);
/* step 2 (spgemm.cu:1220-1309: search_pairs<0/1>, compute_CMasksAndOffsets,
 * compute_CrowColIdx): pair lists, C tile bitmasks, per-tile nnz, intra-tile CSR. */
pem_status pem_spgemm_step2(pem_ctx *ctx, pem_cplan *plan);
/* step 3 (spgemm.cu:1313-1336: pem_spgemm_step3_accumulate): numeric values. */
pem_status pem_spgemm_step3(pem_ctx *ctx, pem_cplan *plan);
/* steps 1-3 back to back = one iteration of the reference's timed loop (spgemm.cu:1133-1341). */
pem_status pem_spgemm(pem_ctx *ctx, pem_cplan *plan);

typedef struct {
    int32_t tile_row_begin, tile_row_end;
    int32_t row_begin, row_end;    /* matrix rows of C this plan produces */
    int64_t ntiles_c;              /* T_C  (_C_nnz, spgemm.cu:1169)        */
    int64_t npairs;                /* pairs kept: products whose tiles can meet (== npairs_all with PEM_PRUNE=0) */
    int64_t nnz_c;                 /* C_nnz (spgemm.cu:1291)               */
    int64_t npairs_all;            /* P    (d_pairs_count, spgemm.cu:1246): every tile-level product, as the reference counts them */
} pem_cplan_info;
pem_status pem_cplan_get_info(const pem_cplan *plan, pem_cplan_info *info);

typedef enum {
    PEM_C_TILE_ROWPTR = 0,   /* int32[(tr_end-tr_begin)+1] _C_rowPtr            spgemm.cu:1166-1168 */
    PEM_C_TILE_ROWIDX,       /* int32[T_C] absolute tile row                    spgemm.cu:378 */
    PEM_C_TILE_COLIDX,       /* int32[T_C]                                      spgemm.cu:379 */
    PEM_C_PAIRS_OFFSET,      /* int32[T_C+1]                                    spgemm.cu:484, 1242 */
    PEM_C_PAIRS_A,           /* int32[P] A tile id, ascending k inside a C tile spgemm.cu:430 */
    PEM_C_PAIRS_B,           /* int32[P] B tile id                              spgemm.cu:428-431 */
    PEM_C_MASK,              /* uint32[8 T_C] (row 2q)<<16 | row 2q+1           spgemm.cu:533-543 */
    PEM_C_TILE_NNZ_PTR,      /* int32[T_C+1]                                    spgemm.cu:546, 1288 */
    PEM_C_ROWPTR,            /* uint8[16 T_C]                                   spgemm.cu:579-580 */
    PEM_C_ROWCOLIDX,         /* uint8[C_nnz] (on demand where step 3 decodes)   spgemm.cu:582-587 */
    PEM_C_VALS               /* double[C_nnz] (float for an fp32 plan)          spgemm.cu:643-656 */
} pem_cplan_array;
pem_status pem_cplan_get_array(pem_ctx *ctx, const pem_cplan *plan, pem_cplan_array which, void *host_dst, int64_t bytes);

/* Graph replay of repeat passes (off by default; PEM_GRAPH=1 in the environment turns it on at context creation).
 * The reference's timed loop re-launches every kernel of every pass (spgemm.cu:1133-1357).  With replay on, the second
 * and later pem_spgemm calls on an unchanged plan run as ONE hipGraph captured from the first repeat pass (same
 * kernels, grids and buffers: results are identical).  A replayed pass has no per-step events, so pem_timings reports
 * step1/2/3_ms = 0 for it and only spgemm_wall_ms; take the step split from a pass made with replay off. */
pem_status pem_set_graph_replay(pem_ctx *ctx, int on);

/* ---- a14: tiled C -> CSR / sorted COO (spgemm.cu:663-695, 1493-1543) ------------------- */
/* rowptr has (row_end-row_begin)+1 entries and is relative to the slice (rowptr[0] = 0);
 * column indices ascend inside a row.  COO rows are absolute, sorted by (row, col).
 * The _device forms write device arrays and return without waiting for the stream; pem_ctx_synchronize() is where a
 * failure inside their kernels (a device scan that gave up: PEM_E_HIP) is reported.  The host forms wait and report it. */
pem_status pem_c_export_csr(pem_ctx *ctx, const pem_cplan *plan, int64_t *nnz, int32_t *rowptr,
                            int32_t *colidx, double *vals);
pem_status pem_c_export_csr_device(pem_ctx *ctx, const pem_cplan *plan, int32_t *d_rowptr,
                                   int32_t *d_colidx, double *d_vals);
pem_status pem_c_export_coo(pem_ctx *ctx, const pem_cplan *plan, int64_t *nnz, int32_t *rows,
                            int32_t *cols, double *vals);
/* the same for a plan over fp32 tilings */
pem_status pem_c_export_csr_f32(pem_ctx *ctx, const pem_cplan *plan, int64_t *nnz, int32_t *rowptr,
                                int32_t *colidx, float *vals);
pem_status pem_c_export_csr_device_f32(pem_ctx *ctx, const pem_cplan *plan, int32_t *d_rowptr,
                                       int32_t *d_colidx, float *d_vals);
pem_status pem_c_export_coo_f32(pem_ctx *ctx, const pem_cplan *plan, int64_t *nnz, int32_t *rows,
                                int32_t *cols, float *vals);

/* ---- multi-GPU helper: balanced tile-row split (new) ----------------------- */
/* bounds[0..nparts] tile-row boundaries of A, balanced on the per-tile-row intermediate
 * product count (the quantity of spgemm_nsparse_kernel.h:135-151 at tile level). */
pem_status pem_split_tile_rows(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, int nparts, int32_t *bounds);
/* weights[0 .. A's tile rows): the per-tile-row weights pem_split_tile_rows balances (tile-level products + tiles + 1), for callers
 * that cut their own row blocks -- e.g. re-cut them from measured per-rank times (multigpu.tune_row_bounds). */
pem_status pem_tile_row_weights(pem_ctx *ctx, const pem_tiled *A, const pem_tiled *B, double *weights);

/* ---- timings (spgemm.cu:1343-1354: the per-step spans of the benchmark CSV) -------------- */
typedef struct {
    double step1_ms, step2_ms, step3_ms;   /* hipEvent spans on the context's stream       */
    double spgemm_wall_ms;                 /* host wall of the last pem_spgemm (CSV col 11)  */
    double export_ms;                      /* last pem_c_export_* device span               */
} pem_timings;
pem_status pem_get_timings(pem_ctx *ctx, pem_timings *t);

/* Per-kernel device time (hipEvent pairs around every launch on the context's stream).
 * Profiling serialises nothing extra but adds two event records per launch; off by default. */
pem_status pem_set_kernel_profiling(pem_ctx *ctx, int enabled);
pem_status pem_reset_kernel_stats(pem_ctx *ctx);
pem_status pem_kernel_stats_count(pem_ctx *ctx, int *n);
pem_status pem_kernel_stats_get(pem_ctx *ctx, int idx, char *name, int name_cap, int64_t *calls, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif
