/*
 * ref_serial_csr.c -- independent serial Gustavson CSR x CSR -> CSR.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED.
 *
 * This is the mathematical definition the tiled pipeline must reproduce: C = A*B with
 *   - structural nnz kept (numerical cancellation does not drop an entry; the reference's
 *     C masks are purely structural, spgemm.cu:533-543),
 *   - sorted column indices per row (the reference's final COO is sorted, spgemm.cu:1516-1519),
 *   - per C entry, products accumulated in ascending k with one fused multiply-add each
 *     (spgemm.cu:635-656: pairs ascending in k-tile, bits ascending inside a tile, `+= a*b`
 *     contracted to FMA by nvcc's default -fmad=true).
 * It shares no code with ref_tiled_cpu.c, so agreement between the two (and with the
 * scipy.sparse fixtures) checks the reading of the reference's tiled algorithm.
 * It is also the CPU baseline of SURVEY 8(d): the reference has no CPU SpGEMM path.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { int i, j; double v; } trip;
static int cmp_trip(const void *a, const void *b)
{
    const trip *x = (const trip *)a, *y = (const trip *)b;
    if (x->i != y->i) return (x->i > y->i) - (x->i < y->i);
    return (x->j > y->j) - (x->j < y->j);
}
static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_csr_free(oracle_csr *m) { free(m->rowptr); free(m->col); free(m->val); memset(m, 0, sizeof *m); }

int oracle_csr_from_coo(int rows, int cols, int nnz, const int *I_in, const int *J_in,
                        const double *V, int transpose, oracle_csr *out)
{
    memset(out, 0, sizeof *out);
    const int *I = transpose ? J_in : I_in, *J = transpose ? I_in : J_in;
    if (transpose) { int t = rows; rows = cols; cols = t; }
    trip *tr = (trip *)malloc(sizeof(trip) * (size_t)(nnz ? nnz : 1));
    for (int e = 0; e < nnz; ++e) {
        if (I[e] < 0 || I[e] >= rows || J[e] < 0 || J[e] >= cols) { free(tr); return -1; }
        tr[e].i = I[e]; tr[e].j = J[e]; tr[e].v = V[e];
    }
    qsort(tr, (size_t)nnz, sizeof(trip), cmp_trip);
    for (int e = 1; e < nnz; ++e)
        if (tr[e].i == tr[e - 1].i && tr[e].j == tr[e - 1].j) { free(tr); return -2; }
    out->rows = rows; out->cols = cols; out->nnz = nnz;
    out->rowptr = (int *)calloc((size_t)rows + 1, sizeof(int));
    out->col = (int *)malloc(sizeof(int) * (size_t)(nnz ? nnz : 1));
    out->val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    for (int e = 0; e < nnz; ++e) { out->rowptr[tr[e].i + 1]++; out->col[e] = tr[e].j; out->val[e] = tr[e].v; }
    for (int r = 0; r < rows; ++r) out->rowptr[r + 1] += out->rowptr[r];
    free(tr);
    return 0;
}

/* with g_f32 set the chain runs in float (operands are float values held in doubles, one fmaf per
 * product, result widened exactly) -- the checker of the fp32 product path. */
static int g_f32 = 0;

/* one row: returns the number of distinct columns; if col/val != NULL also writes them sorted */
static int row_product(const oracle_csr *A, const oracle_csr *B, int i, int *marker, double *acc,
                       int *list, int *col, double *val)
{
    int cnt = 0;
    for (int ea = A->rowptr[i]; ea < A->rowptr[i + 1]; ++ea) {        /* ascending k */
        int k = A->col[ea];
        double a = A->val[ea];
        for (int eb = B->rowptr[k]; eb < B->rowptr[k + 1]; ++eb) {
            int j = B->col[eb];
            if (marker[j] != i) { marker[j] = i; acc[j] = 0.0; list[cnt++] = j; }
            acc[j] = g_f32 ? (double)fmaf((float)a, (float)B->val[eb], (float)acc[j]) : fma(a, B->val[eb], acc[j]);
        }
    }
    if (col) {
        qsort(list, (size_t)cnt, sizeof(int), cmp_int);
        for (int n = 0; n < cnt; ++n) { col[n] = list[n]; val[n] = acc[list[n]]; }
    }
    return cnt;
}

int oracle_csr_spgemm(const oracle_csr *A, const oracle_csr *B, int threads, oracle_csr *C)
{
    memset(C, 0, sizeof *C);
    if (A->cols != B->rows) return -1;
    int m = A->rows, n = B->cols;
    C->rows = m; C->cols = n;
    C->rowptr = (int *)calloc((size_t)m + 1, sizeof(int));
    if (threads < 1) threads = 1;
#ifndef _OPENMP
    threads = 1;
#endif
    int64_t *cnt = (int64_t *)calloc((size_t)m + 1, sizeof(int64_t));
    /* symbolic pass: row sizes */
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        int *marker = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        double *acc = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
        int *list = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        for (int j = 0; j < n; ++j) marker[j] = -1;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int i = 0; i < m; ++i) cnt[i + 1] = row_product(A, B, i, marker, acc, list, NULL, NULL);
        free(marker); free(acc); free(list);
    }
    for (int i = 0; i < m; ++i) cnt[i + 1] += cnt[i];
    if (cnt[m] > 0x7FFFFFFF) { free(cnt); oracle_csr_free(C); return -4; }
    for (int i = 0; i <= m; ++i) C->rowptr[i] = (int)cnt[i];
    C->nnz = cnt[m];
    free(cnt);
    C->col = (int *)malloc(sizeof(int) * (size_t)(C->nnz ? C->nnz : 1));
    C->val = (double *)malloc(sizeof(double) * (size_t)(C->nnz ? C->nnz : 1));
    /* numeric pass */
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        int *marker = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        double *acc = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
        int *list = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        for (int j = 0; j < n; ++j) marker[j] = -1;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int i = 0; i < m; ++i)
            row_product(A, B, i, marker, acc, list, C->col + C->rowptr[i], C->val + C->rowptr[i]);
        free(marker); free(acc); free(list);
    }
    return 0;
}

int oracle_csr_spgemm_f32(const oracle_csr *A, const oracle_csr *B, int threads, oracle_csr *C)
{
    g_f32 = 1;
    int rc = oracle_csr_spgemm(A, B, threads, C);
    g_f32 = 0;
    return rc;
}
