/*
 * pem_host.h -- C ABI of libpemhost.so: host-side I/O of the `pemspgemm` command-line tool.
 * Replaces read_matrix_market<T> (spgemm.cu:43-110, which wraps fast_matrix_market v1.7.6)
 * and the result/CSV writers of spgemm.cu:1424-1450, 1527-1560.  No GPU code in here.
 */
#ifndef PEM_HOST_H
#define PEM_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t rows, cols;
    int64_t nnz;          /* after symmetry expansion */
    int32_t *I, *J;       /* 0-based triplets, file order; mirrored entries appended */
    double *V;            /* pattern -> 1.0, complex -> real part (spgemm.cu:99-107) */
    int32_t symmetric;    /* header symmetry != general (spgemm.cu:65-68) */
    int32_t field;        /* 0 real, 1 integer, 2 pattern, 3 complex */
} pem_coo;

/* 0 ok; <0 error, message via pem_host_last_error().  threads <= 0: all hardware threads. */
int pem_mm_read(const char *path, int threads, pem_coo *out);
void pem_coo_free(pem_coo *m);
const char *pem_host_last_error(void);

/* Seeded synthetic stand-ins for the SuiteSparse inputs BASELINE.json names (no SuiteSparse file exists
 * offline; "generator in C++, fixed seeds, values uniform in [-1,1) excluding 0, no duplicates, sorted rows").
 * name: one of pem_standin_names() ("cage4 scircuit webbase-1M mc2depi cage15"); scale in (0, 1] shrinks rows and
 * nnz together (tests).  The triplets come out sorted by (row, column); free with pem_coo_free.
 * 0 ok, -1 bad argument, -2 unknown name, -6 out of memory.  Models and calibration: host/standin.cpp. */
int pem_standin_generate(const char *name, double scale, pem_coo *out);
const char *pem_standin_names(void);

/* spgemm.cu:1527-1560: <dir>/SPGEMM_RESULT_{NNZ,ROWS,COLS,VALS}.txt -- NNZ one integer without
 * newline, ROWS/COLS one 0-based int per line, VALS fixed with 17 digits after the point. */
int pem_write_result_files(const char *dir, int64_t nnz, const int32_t *rows, const int32_t *cols, const double *vals);

/* SURVEY 8(f)-1 (new, beyond the reference): write a CSR matrix as a Matrix-Market coordinate real general file,
 * 1-based, rows ascending, columns ascending inside a row, values with 17 significant digits. */
int pem_write_mtx_csr(const char *path, int32_t rows, int32_t cols, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                      const char *comment);

/* spgemm.cu:1424-1450: append "\n" + the 14 reference fields (fixed, 2 decimals) to `path`;
 * `extra` (may be NULL) is appended verbatim after the 14th field. */
typedef struct {
    const char *matrix;                  /* file stem (spgemm.cu:1428-1431) */
    uint64_t flop;
    int64_t c_nnz;
    double compression_ratio;
    double a_conversion_kernel_ms, b_conversion_kernel_ms, total_conversion_ms;
    double step1_ms, step2_ms, step3_ms;
    double spgemm_ms, kernel_ms, malloc_ms, gflops;
} pem_csv_record;
int pem_csv_append(const char *path, const pem_csv_record *rec, const char *extra);

#ifdef __cplusplus
}
#endif
#endif
