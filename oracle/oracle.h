/*
 * oracle.h -- CPU restatement of the pem-spgemm hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is the *checker*: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under pem-spgemm_amd/ links,
 * imports or calls it; the product path fails loudly when the HIP library is absent.
 *
 * PARITY UNPINNED: the reference (stckvrflw/pem-spgemm) ships no tests, no golden
 * vectors and no recorded outputs, and it cannot be compiled here (CUDA/nvcc + rmm +
 * fast_matrix_market, none present).  This restatement follows the reference source
 * line by line (citations below, relative to /root/reference) and is cross-checked
 * against scipy.sparse fixtures generated in the build container (tests/golden/).
 *
 * Three independent pieces:
 *   mm_read.c         a1   Matrix-Market coordinate reader (spgemm.cu:43-110 + the
 *                          documented defaults of fast_matrix_market v1.7.6)
 *   ref_tiled_cpu.c   a2-a14  stage-by-stage tiled pipeline with the reference's
 *                          array layouts (spgemm.cu:112-695, 832-1062)
 *   ref_serial_csr.c  independent serial Gustavson CSRxCSR->CSR (the mathematical
 *                          definition of C; also the CPU baseline of SURVEY 8(d))
 */
#ifndef PEM_ORACLE_H
#define PEM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: Matrix Market reader ------------------------------------------------------- */
typedef struct {
    int rows, cols;
    int64_t nnz;         /* after symmetry expansion */
    int *I, *J;          /* 0-based, file order; mirrored entries follow their source */
    double *V;
    int symmetric;       /* header said symmetric / skew / hermitian (spgemm.cu:65-68) */
    int field;           /* 0 real, 1 integer, 2 pattern, 3 complex */
} oracle_coo;

int  oracle_mm_read(const char *path, oracle_coo *out);   /* 0 ok, <0 error */
void oracle_coo_free(oracle_coo *m);

/* ---- a2-a7: tiled CSR ---------------------------------------------------------------- */
typedef struct {
    int rows, cols, nnz;
    int tile_rows, tile_cols;     /* ceil(rows/16), ceil(cols/16)      spgemm.cu:840-843 */
    int ntiles;                   /* T = cntA / cntB                   spgemm.cu:871     */
    int64_t *tile_keys;           /* [T] (tileRow<<32)|tileCol sorted  spgemm.cu:131-133, 869-877 */
    int *tile_nnz_ptr;            /* [T+1] perTileNnz exclusive scan   spgemm.cu:873-874 */
    int *csr_rowptr;              /* [rows+1]                          spgemm.cu:894-910 */
    int *csr_col;                 /* [nnz] row-major sorted J          spgemm.cu:897     */
    double *csr_val;              /* [nnz]                                               */
    uint16_t *masks;              /* [16T] bit c of masks[16t+r] <=> (r,c) present  :196-200 */
    uint8_t *rowptr;              /* [16T] nnz in tile rows < r        spgemm.cu:205-209 */
    uint8_t *rowcolidx;           /* [nnz] (r<<4)|c, row-major in tile spgemm.cu:195,221 */
    double *vals;                 /* [nnz] tile order                  spgemm.cu:220     */
    uint16_t *masks_t;            /* [16T] BT[m] bit i = masks[i] bit m spgemm.cu:244-253 */
    int *tile_rowptr;             /* [tile_rows+1]                     spgemm.cu:986-999 */
    int *tile_colidx;             /* [T]                               spgemm.cu:1001-1006 */
    int *tile_colptr;             /* [tile_cols+1]                     spgemm.cu:1042-1055 */
    int *tile_rowidx;             /* [T] tile rows in CSC order        spgemm.cu:1056-1061 */
    int *tile_offsets;            /* [T] CSC position -> CSR tile id   spgemm.cu:1034-1040 */
} oracle_tiled;

/* transpose != 0 swaps I/J and rows/cols first (spgemm.cu:788-792). Duplicate (i,j)
 * entries are rejected (returns -2): the reference mishandles them. */
int  oracle_tiled_from_coo(int rows, int cols, int nnz, const int *I, const int *J,
                           const double *V, int transpose, oracle_tiled *out);
void oracle_tiled_free(oracle_tiled *t);

/* ---- a8: flop count (spgemm.cu:1068-1079) ------------------------------------------- */
uint64_t oracle_flop_count(const oracle_tiled *A, const oracle_tiled *B);

/* ---- a9-a13: the three steps --------------------------------------------------------- */
typedef struct {
    int tr_lo, tr_hi;             /* tile-row range of A this plan covers (multi-GPU slices) */
    int ntiles_c;                 /* T_C = _C_nnz                     spgemm.cu:1169 */
    int64_t npairs;               /* P = d_pairs_count                spgemm.cu:1246 */
    int64_t nnz_c;                /* C_nnz                            spgemm.cu:1291 */
    int *c_tile_rowptr;           /* [(tr_hi-tr_lo)+1] _C_rowPtr      spgemm.cu:1166-1168 */
    int *c_tile_rowidx;           /* [T_C] (absolute tile row)        spgemm.cu:378 */
    int *c_tile_colidx;           /* [T_C]                            spgemm.cu:379 */
    int *pairs_offset;            /* [T_C+1]                          spgemm.cu:484, 1242 */
    int *pairs_a;                 /* [P] A CSR tile id                spgemm.cu:430 */
    int *pairs_b;                 /* [P] B CSR tile id                spgemm.cu:428-431 */
    uint32_t *c_mask;             /* [8 T_C] (row 2q)<<16 | row 2q+1  spgemm.cu:533-543 */
    int *c_tile_nnz_ptr;          /* [T_C+1]                          spgemm.cu:546, 1288 */
    uint8_t *c_rowptr;            /* [16 T_C]                         spgemm.cu:579-580 */
    uint8_t *c_rowcolidx;         /* [C_nnz]                          spgemm.cu:582-587 */
    double *c_vals;               /* [C_nnz]                          spgemm.cu:643-656 */
} oracle_cplan;

int  oracle_spgemm_step1(const oracle_tiled *A, const oracle_tiled *B, int tr_lo, int tr_hi,
                         oracle_cplan *p);
int  oracle_spgemm_step2(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p);
int  oracle_spgemm_step3(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p);
/* fp32 chain: operands must be float-representable; c_vals holds the float results widened */
int  oracle_spgemm_step3_f32(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p);
void oracle_cplan_free(oracle_cplan *p);

/* a14: tiled C -> COO sorted by (row, col) (spgemm.cu:663-695, 1516-1519), caller
 * buffers of nnz_c; and the same as CSR over rows [16*tr_lo, min(16*tr_hi, rows_A)). */
int oracle_c_export_coo(const oracle_cplan *p, int *rows, int *cols, double *vals);
int oracle_c_export_csr(const oracle_cplan *p, int rows_a, int *rowptr, int *colidx, double *vals);

/* ---- independent serial CSR Gustavson ------------------------------------------------ */
typedef struct { int rows, cols; int64_t nnz; int *rowptr; int *col; double *val; } oracle_csr;

/* COO -> CSR with sorted columns (duplicates rejected: -2). */
int  oracle_csr_from_coo(int rows, int cols, int nnz, const int *I, const int *J,
                         const double *V, int transpose, oracle_csr *out);
/* C = A*B, sorted columns, ascending-k accumulation with one fma per product, structural
 * zeros kept.  threads<=1: serial; >1: OpenMP row-parallel (same arithmetic, same result). */
int  oracle_csr_spgemm(const oracle_csr *A, const oracle_csr *B, int threads, oracle_csr *C);
int  oracle_csr_spgemm_f32(const oracle_csr *A, const oracle_csr *B, int threads, oracle_csr *C);   /* float chain, fmaf */
void oracle_csr_free(oracle_csr *m);
int  oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
