/*
 * mm_read.c -- oracle restatement of row a1: Matrix-Market -> COO.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED.
 *
 * Follows read_matrix_market<T> (spgemm.cu:43-110), which delegates the parse to
 * fast_matrix_market v1.7.6 (spgemm.cu:60-61, 72-74, 79-81; the library itself is NOT in
 * /root/reference).  What the reference's call sites fix:
 *   - triplets come back 0-based in I/J (used directly as indices, spgemm.cu:126-127);
 *   - complex files keep the real part only (spgemm.cu:99-107);
 *   - pattern files get a placeholder value (fast_matrix_market's default is 1);
 *   - symmetric / skew-symmetric / hermitian storage is generalised to a full matrix.
 * fast_matrix_market's default for diagonal entries of symmetric coordinate files is an
 * extra explicit-zero duplicate; the reference cannot digest duplicates, so
 * this restatement emits each diagonal entry once -- the mathematically intended matrix.
 * Mirrored entries are appended after all file entries (order is irrelevant downstream:
 * every consumer sorts).
 */
#include "oracle.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void lower(char *s) { for (; *s; ++s) *s = (char)tolower((unsigned char)*s); }

int oracle_mm_read(const char *path, oracle_coo *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[1024], banner[64], object[64], format[64], field[64], symm[64];
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symm) != 5) { fclose(f); return -2; }
    lower(banner); lower(object); lower(format); lower(field); lower(symm);
    if (strcmp(banner, "%%matrixmarket") || strcmp(object, "matrix")) { fclose(f); return -2; }
    if (strcmp(format, "coordinate")) { fclose(f); return -3; }   /* dense 'array' files: out of scope */
    int fld = !strcmp(field, "real") ? 0 : !strcmp(field, "double") ? 0 : !strcmp(field, "integer") ? 1 :
              !strcmp(field, "pattern") ? 2 : !strcmp(field, "complex") ? 3 : -1;
    int sym = !strcmp(symm, "general") ? 0 : !strcmp(symm, "symmetric") ? 1 :
              !strcmp(symm, "skew-symmetric") ? 2 : !strcmp(symm, "hermitian") ? 3 : -1;
    if (fld < 0 || sym < 0) { fclose(f); return -2; }
    /* comments and blank lines */
    do { if (!fgets(line, sizeof line, f)) { fclose(f); return -2; } } while (line[0] == '%' || line[strspn(line, " \t\r\n")] == 0);
    long long rows, cols, n;
    if (sscanf(line, "%lld %lld %lld", &rows, &cols, &n) != 3) { fclose(f); return -2; }
    int64_t cap = sym ? 2 * n : n;
    int *I = (int *)malloc(sizeof(int) * (size_t)(cap ? cap : 1));
    int *J = (int *)malloc(sizeof(int) * (size_t)(cap ? cap : 1));
    double *V = (double *)malloc(sizeof(double) * (size_t)(cap ? cap : 1));
    int64_t cnt = 0;
    while (cnt < n && fgets(line, sizeof line, f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == 0 || *p == '\n' || *p == '\r' || *p == '%') continue;
        char *e;
        long long i = strtoll(p, &e, 10); p = e;
        long long j = strtoll(p, &e, 10); p = e;
        double v = 1.0;                                   /* pattern placeholder */
        if (fld != 2) v = strtod(p, &e);                  /* complex: real part is the first number */
        if (i < 1 || j < 1 || i > rows || j > cols) { free(I); free(J); free(V); fclose(f); return -4; }
        I[cnt] = (int)(i - 1); J[cnt] = (int)(j - 1); V[cnt] = v; ++cnt;
    }
    fclose(f);
    if (cnt != n) { free(I); free(J); free(V); return -5; }
    if (sym) {
        for (int64_t e = 0; e < n; ++e) {
            if (I[e] == J[e]) continue;
            I[cnt] = J[e]; J[cnt] = I[e];
            V[cnt] = (sym == 2) ? -V[e] : V[e];           /* hermitian: conj keeps the real part */
            ++cnt;
        }
    }
    out->rows = (int)rows; out->cols = (int)cols; out->nnz = cnt;
    out->I = I; out->J = J; out->V = V; out->symmetric = sym != 0; out->field = fld;
    return 0;
}

void oracle_coo_free(oracle_coo *m)
{
    free(m->I); free(m->J); free(m->V);
    memset(m, 0, sizeof *m);
}
