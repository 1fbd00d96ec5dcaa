/*
 * ref_tiled_cpu.c -- oracle restatement of rows a2-a14: the tiled pipeline, stage by stage,
 * with the reference's array layouts so every HIP stage can be diffed at its own boundary.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED (the reference has no tests).
 *
 * Each function cites the reference lines (relative to /root/reference) it follows.  Where
 * the reference uses a library primitive (thrust sort / unique / reduce_by_key / scan), the
 * restatement computes the same, uniquely determined result with plain loops + qsort.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TS 16

static int popc16(unsigned x) { return __builtin_popcount(x & 0xFFFFu); }

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

typedef struct { int i, j; double v; } trip;
static int cmp_trip(const void *a, const void *b)
{
    const trip *x = (const trip *)a, *y = (const trip *)b;
    if (x->i != y->i) return (x->i > y->i) - (x->i < y->i);
    return (x->j > y->j) - (x->j < y->j);
}

typedef struct { int64_t key; int id; } keyid;
static int cmp_keyid(const void *a, const void *b)
{
    const keyid *x = (const keyid *)a, *y = (const keyid *)b;
    if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
    return (x->id > y->id) - (x->id < y->id);
}

/* utilities.h:104-131 binarySearch: index of target in sorted arr, or -1 */
static int bsearch_int(const int *arr, int target, int len)
{
    int l = 0, r = len - 1;
    while (l <= r) {
        int m = l + (r - l) / 2;
        if (arr[m] == target) return m;
        if (arr[m] < target) l = m + 1; else r = m - 1;
    }
    return -1;
}

void oracle_tiled_free(oracle_tiled *t)
{
    free(t->tile_keys); free(t->tile_nnz_ptr); free(t->csr_rowptr); free(t->csr_col); free(t->csr_val);
    free(t->masks); free(t->rowptr); free(t->rowcolidx); free(t->vals); free(t->masks_t);
    free(t->tile_rowptr); free(t->tile_colidx); free(t->tile_colptr); free(t->tile_rowidx); free(t->tile_offsets);
    memset(t, 0, sizeof *t);
}

#define ALLOC(T, n) ((T *)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(T)))

int oracle_tiled_from_coo(int rows, int cols, int nnz, const int *I_in, const int *J_in,
                          const double *V, int transpose, oracle_tiled *out)
{
    memset(out, 0, sizeof *out);
    /* spgemm.cu:788-792: A*A^T = swap(B_I,B_J), swap(B_rows,B_cols) before upload */
    const int *I = transpose ? J_in : I_in, *J = transpose ? I_in : J_in;
    if (transpose) { int t = rows; rows = cols; cols = t; }
    out->rows = rows; out->cols = cols; out->nnz = nnz;
    out->tile_rows = (rows - 1 + TS) / TS;                    /* spgemm.cu:840-843 */
    out->tile_cols = (cols - 1 + TS) / TS;
    for (int e = 0; e < nnz; ++e)
        if (I[e] < 0 || I[e] >= rows || J[e] < 0 || J[e] >= cols) return -1;

    /* a2 decide_which_tile (spgemm.cu:112-135): key = (row>>4)<<32 | (col>>4) */
    int64_t *keys = ALLOC(int64_t, nnz);
    for (int e = 0; e < nnz; ++e) keys[e] = ((int64_t)(I[e] >> 4) << 32) | (int64_t)(J[e] >> 4);
    /* a3 (spgemm.cu:866-892): sort, unique_count, reduce_by_key(1), exclusive_scan, unique */
    qsort(keys, (size_t)nnz, sizeof(int64_t), cmp_i64);
    int T = 0;
    for (int e = 0; e < nnz; ++e) if (e == 0 || keys[e] != keys[e - 1]) ++T;
    out->ntiles = T;
    out->tile_keys = ALLOC(int64_t, T);
    out->tile_nnz_ptr = ALLOC(int, T + 1);
    {
        int t = -1;
        for (int e = 0; e < nnz; ++e) {
            if (e == 0 || keys[e] != keys[e - 1]) { ++t; out->tile_keys[t] = keys[e]; out->tile_nnz_ptr[t] = e; }
        }
        out->tile_nnz_ptr[T] = nnz;
    }
    free(keys);

    /* a4 (spgemm.cu:894-928): stable_sort zip(I,J,val) lexicographically; rowPtr by
     * reduce_by_key + scatter + exclusive_scan (rows without entries keep an empty range) */
    trip *tr = ALLOC(trip, nnz);
    for (int e = 0; e < nnz; ++e) { tr[e].i = I[e]; tr[e].j = J[e]; tr[e].v = V[e]; }
    qsort(tr, (size_t)nnz, sizeof(trip), cmp_trip);
    for (int e = 1; e < nnz; ++e)
        if (tr[e].i == tr[e - 1].i && tr[e].j == tr[e - 1].j) { free(tr); oracle_tiled_free(out); return -2; }
    out->csr_rowptr = ALLOC(int, rows + 1);
    out->csr_col = ALLOC(int, nnz);
    out->csr_val = ALLOC(double, nnz);
    for (int e = 0; e < nnz; ++e) { out->csr_rowptr[tr[e].i + 1]++; out->csr_col[e] = tr[e].j; out->csr_val[e] = tr[e].v; }
    for (int r = 0; r < rows; ++r) out->csr_rowptr[r + 1] += out->csr_rowptr[r];
    free(tr);

    /* a5 generate_tiles_csr (spgemm.cu:137-226): one block per tile, thread (r,c) */
    out->masks = ALLOC(uint16_t, 16 * (size_t)T);
    out->rowptr = ALLOC(uint8_t, 16 * (size_t)T);
    out->rowcolidx = ALLOC(uint8_t, nnz);
    out->vals = ALLOC(double, nnz);
    for (int t = 0; t < T; ++t) {
        int tx = (int)(out->tile_keys[t] & 0xFFFFFFFF), ty = (int)(out->tile_keys[t] >> 32);   /* :166-167 */
        int off_x = tx << 4, off_y = ty << 4;
        int slot = out->tile_nnz_ptr[t];                       /* :211 tile_offset; :216 block exclusive sum */
        int run = 0;
        for (int r = 0; r < TS; ++r) {
            unsigned mask = 0;
            int row = off_y + r;
            if (row < rows) {                                  /* :181 */
                int rp = out->csr_rowptr[row], len = out->csr_rowptr[row + 1] - rp;
                for (int c = 0; c < TS; ++c) {
                    int pos = bsearch_int(out->csr_col + rp, off_x + c, len);   /* :186 */
                    if (pos != -1) {
                        mask |= 1u << c;                       /* :196 ballot over the 16-lane row group */
                        out->vals[slot] = out->csr_val[rp + pos];               /* :220 */
                        out->rowcolidx[slot] = (uint8_t)((r << 4) | c);          /* :195, :221 */
                        ++slot;
                    }
                }
            }
            out->masks[16 * (size_t)t + r] = (uint16_t)mask;   /* :200 */
            out->rowptr[16 * (size_t)t + r] = (uint8_t)run;    /* :205-209 exclusive scan of row nnz */
            run += popc16(mask);
        }
        if (slot != out->tile_nnz_ptr[t + 1]) { oracle_tiled_free(out); return -3; }
    }

    /* a6 __transpose_B_mask (spgemm.cu:228-258): BT[m] bit i = masks[i] bit m */
    out->masks_t = ALLOC(uint16_t, 16 * (size_t)T);
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < 16; ++m) {
            unsigned tmp = 0;
            for (int i = 0; i < 16; ++i) tmp |= ((out->masks[16 * (size_t)t + i] >> m) & 1u) << i;   /* :250 */
            out->masks_t[16 * (size_t)t + m] = (uint16_t)tmp;
        }

    /* a7 tile-level CSR (spgemm.cu:986-1031) */
    out->tile_rowptr = ALLOC(int, out->tile_rows + 1);
    out->tile_colidx = ALLOC(int, T);
    for (int t = 0; t < T; ++t) {
        out->tile_rowptr[(int)(out->tile_keys[t] >> 32) + 1]++;
        out->tile_colidx[t] = (int)(out->tile_keys[t] & 0xFFFFFFFF);
    }
    for (int r = 0; r < out->tile_rows; ++r) out->tile_rowptr[r + 1] += out->tile_rowptr[r];

    /* a7 tile-level CSC + _B_tileOffsets (spgemm.cu:1033-1062): swap32 the keys, sort
     * zip(key, sequence) -> permutation CSC position -> CSR tile id */
    keyid *kc = ALLOC(keyid, T);
    for (int t = 0; t < T; ++t) {
        int64_t k = out->tile_keys[t];
        kc[t].key = ((k & 0xFFFFFFFF) << 32) | ((k >> 32) & 0xFFFFFFFF);     /* utilities.h swap32 */
        kc[t].id = t;
    }
    qsort(kc, (size_t)T, sizeof(keyid), cmp_keyid);
    out->tile_colptr = ALLOC(int, out->tile_cols + 1);
    out->tile_rowidx = ALLOC(int, T);
    out->tile_offsets = ALLOC(int, T);
    for (int t = 0; t < T; ++t) {
        out->tile_colptr[(int)(kc[t].key >> 32) + 1]++;
        out->tile_rowidx[t] = (int)(kc[t].key & 0xFFFFFFFF);
        out->tile_offsets[t] = kc[t].id;
    }
    for (int c = 0; c < out->tile_cols; ++c) out->tile_colptr[c + 1] += out->tile_colptr[c];
    free(kc);
    return 0;
}

/* a8 (spgemm.cu:1068-1079): flop = sum over nnz e of A of nnz(B[col(e),:]) */
uint64_t oracle_flop_count(const oracle_tiled *A, const oracle_tiled *B)
{
    uint64_t flop = 0;
    for (int e = 0; e < A->nnz; ++e) {
        int k = A->csr_col[e];
        if (k < B->rows) flop += (uint64_t)(B->csr_rowptr[k + 1] - B->csr_rowptr[k]);
    }
    return flop;
}

void oracle_cplan_free(oracle_cplan *p)
{
    free(p->c_tile_rowptr); free(p->c_tile_rowidx); free(p->c_tile_colidx);
    free(p->pairs_offset); free(p->pairs_a); free(p->pairs_b);
    free(p->c_mask); free(p->c_tile_nnz_ptr); free(p->c_rowptr); free(p->c_rowcolidx); free(p->c_vals);
    memset(p, 0, sizeof *p);
}

/* a9 step 1, SPA path (spgemm.cu:271-384).  The NSPARSE path used for wide B
 * (spgemm.cu:1142-1151, 1175-1195) has the same output contract: C tiles of every tile
 * row in ascending tile-column order (rank sort, spgemm_nsparse_kernel.h:788-798). */
int oracle_spgemm_step1(const oracle_tiled *A, const oracle_tiled *B, int tr_lo, int tr_hi,
                        oracle_cplan *p)
{
    memset(p, 0, sizeof *p);
    if (A->cols != B->rows || tr_lo < 0 || tr_hi > A->tile_rows || tr_lo > tr_hi) return -1;
    p->tr_lo = tr_lo; p->tr_hi = tr_hi;
    int mt = tr_hi - tr_lo, nt = B->tile_cols;
    int nmasks = (nt + 31) / 32;
    unsigned *bitmask = ALLOC(unsigned, nmasks);
    p->c_tile_rowptr = ALLOC(int, mt + 1);
    /* pass 1: tile_spgemm_step1_cuda_spa_kernel (:271-313) count, then exclusive scan (:1168) */
    for (int i = tr_lo; i < tr_hi; ++i) {
        memset(bitmask, 0, sizeof(unsigned) * (size_t)nmasks);
        for (int a = A->tile_rowptr[i]; a < A->tile_rowptr[i + 1]; ++a) {
            int k = A->tile_colidx[a];
            for (int b = B->tile_rowptr[k]; b < B->tile_rowptr[k + 1]; ++b) {
                int col = B->tile_colidx[b];
                bitmask[col / 32] |= 1u << (31 - col % 32);      /* :300-301 */
            }
        }
        int cnt = 0;
        for (int w = 0; w < nmasks; ++w) cnt += __builtin_popcount(bitmask[w]);
        p->c_tile_rowptr[i - tr_lo + 1] = cnt;
    }
    for (int i = 0; i < mt; ++i) p->c_tile_rowptr[i + 1] += p->c_tile_rowptr[i];
    p->ntiles_c = p->c_tile_rowptr[mt];
    p->c_tile_rowidx = ALLOC(int, p->ntiles_c);
    p->c_tile_colidx = ALLOC(int, p->ntiles_c);
    /* pass 2: ..._numeric_cuda_spa_kernel (:316-384): rebuild, walk bits MSB->LSB = ascending column */
    for (int i = tr_lo; i < tr_hi; ++i) {
        memset(bitmask, 0, sizeof(unsigned) * (size_t)nmasks);
        for (int a = A->tile_rowptr[i]; a < A->tile_rowptr[i + 1]; ++a) {
            int k = A->tile_colidx[a];
            for (int b = B->tile_rowptr[k]; b < B->tile_rowptr[k + 1]; ++b) {
                int col = B->tile_colidx[b];
                bitmask[col / 32] |= 1u << (31 - col % 32);
            }
        }
        int pos = p->c_tile_rowptr[i - tr_lo];
        for (int w = 0; w < nmasks; ++w)
            for (int bit = 0; bit < 32; ++bit)
                if ((bitmask[w] >> (31 - bit)) & 1u) {           /* :374-381 */
                    p->c_tile_rowidx[pos] = i;
                    p->c_tile_colidx[pos] = w * 32 + bit;
                    ++pos;
                }
    }
    free(bitmask);
    return 0;
}

/* a10 __find_pairs (spgemm.cu:387-439): iterate the shorter list, binarySearch the longer;
 * hits are appended in iteration order (= ascending k, both lists are sorted). */
static int find_pairs(int *pa, int *pb, const int *iter, int iter_len, const int *targ, int targ_len,
                      int iter_off, int targ_off, int AorB, const int *b_tile_offsets)
{
    int n = 0;
    for (int i = 0; i < iter_len; ++i) {
        int found = bsearch_int(targ, iter[i], targ_len);           /* :409 */
        if (found == -1) continue;
        if (pa) {
            int first = iter_off + i, second = targ_off + found;    /* :424-425 */
            if (!AorB) { int t = first; first = second; second = t; }   /* :426-427 order is A,B */
            pa[n] = first;                                          /* :430 A CSR tile id */
            pb[n] = b_tile_offsets[second];                         /* :428,431 CSC pos -> B CSR tile id */
        }
        ++n;
    }
    return n;
}

int oracle_spgemm_step2(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p)
{
    int TC = p->ntiles_c;
    /* a10 pass 0 (spgemm.cu:441-485, 1227-1242): counts -> exclusive scan */
    p->pairs_offset = ALLOC(int, TC + 1);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            int64_t run = 0;
            for (int t = 0; t < TC; ++t) { int c = p->pairs_offset[t]; p->pairs_offset[t] = (int)run; run += c; }
            p->pairs_offset[TC] = (int)run;
            p->npairs = run;
            if (run > 0x7FFFFFFF) return -4;
            p->pairs_a = ALLOC(int, run);
            p->pairs_b = ALLOC(int, run);
        }
        for (int t = 0; t < TC; ++t) {
            int crow = p->c_tile_rowidx[t], ccol = p->c_tile_colidx[t];          /* :467-468 */
            const int *aseg = A->tile_colidx + A->tile_rowptr[crow];              /* :469 */
            const int *bseg = B->tile_rowidx + B->tile_colptr[ccol];              /* :470 */
            int alen = A->tile_rowptr[crow + 1] - A->tile_rowptr[crow];
            int blen = B->tile_colptr[ccol + 1] - B->tile_colptr[ccol];
            int AorB = alen <= blen ? 1 : 0;                                      /* :473 */
            int *pa = pass ? p->pairs_a + p->pairs_offset[t] : NULL;
            int *pb = pass ? p->pairs_b + p->pairs_offset[t] : NULL;
            int n = AorB ? find_pairs(pa, pb, aseg, alen, bseg, blen, A->tile_rowptr[crow], B->tile_colptr[ccol], 1, B->tile_offsets)
                         : find_pairs(pa, pb, bseg, blen, aseg, alen, B->tile_colptr[ccol], A->tile_rowptr[crow], 0, B->tile_offsets);
            if (!pass) p->pairs_offset[t] = n;                                    /* :483-484 */
        }
    }

    /* a11 compute_CMasksAndOffsets (spgemm.cu:499-550): lane q owns rows 2q, 2q+1 */
    p->c_mask = ALLOC(uint32_t, 8 * (size_t)TC);
    p->c_tile_nnz_ptr = ALLOC(int, TC + 1);
    for (int t = 0; t < TC; ++t) {
        int nnz_t = 0;
        for (int q = 0; q < 8; ++q) {
            unsigned acc = 0;
            for (int pr = p->pairs_offset[t]; pr < p->pairs_offset[t + 1]; ++pr) {
                int a = p->pairs_a[pr], b = p->pairs_b[pr];
                unsigned cm = 0;
                for (int n = 0; n < 16; ++n)                                       /* :535 */
                    cm |= (unsigned)((A->masks[16 * (size_t)a + 2 * q] & B->masks_t[16 * (size_t)b + n]) != 0) << n;
                cm <<= 16;                                                         /* :536 */
                for (int n = 0; n < 16; ++n)                                       /* :538 */
                    cm |= (unsigned)((A->masks[16 * (size_t)a + 2 * q + 1] & B->masks_t[16 * (size_t)b + n]) != 0) << n;
                acc |= cm;                                                         /* :540 */
            }
            p->c_mask[8 * (size_t)t + q] = acc;                                    /* :543 */
            nnz_t += __builtin_popcount(acc);                                      /* :545 */
        }
        p->c_tile_nnz_ptr[t] = nnz_t;                                              /* :546 */
    }
    {   /* :1288 exclusive scan over T_C+1 */
        int64_t run = 0;
        for (int t = 0; t < TC; ++t) { int c = p->c_tile_nnz_ptr[t]; p->c_tile_nnz_ptr[t] = (int)run; run += c; }
        p->c_tile_nnz_ptr[TC] = (int)run;
        p->nnz_c = run;
        if (run > 0x7FFFFFFF) return -4;
    }

    /* a12 compute_CrowColIdx (spgemm.cu:552-591): lane = row */
    p->c_rowptr = ALLOC(uint8_t, 16 * (size_t)TC);
    p->c_rowcolidx = ALLOC(uint8_t, p->nnz_c);
    for (int t = 0; t < TC; ++t) {
        int off = p->c_tile_nnz_ptr[t], run = 0;
        for (int r = 0; r < 16; ++r) {
            unsigned m = p->c_mask[8 * (size_t)t + (r >> 1)];                      /* :574 */
            m >>= (((r % 2) ^ 1) << 4);                                            /* :575 */
            m &= 0xFFFF;                                                           /* :576 */
            p->c_rowptr[16 * (size_t)t + r] = (uint8_t)run;                        /* :579-580 */
            for (int c = 0; c < 16; ++c)                                           /* :582-587 n-th set bit, ascending */
                if ((m >> c) & 1u) p->c_rowcolidx[off + run++] = (uint8_t)((r << 4) | c);
        }
    }
    return 0;
}

/* a13 pem_spgemm_step3_accumulate (spgemm.cu:593-661).  One accumulator per C entry,
 * pairs in ascending k-tile order, bits of Amask[r]&BT[c] ascending, one multiply-add per
 * product (nvcc contracts `+= a*b` to an FMA by default).  The reference accumulates into
 * never-zeroed memory; the restatement starts from +0.0. */
static int step3_impl(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p, int f32)
{
    p->c_vals = ALLOC(double, p->nnz_c);
    for (int t = 0; t < p->ntiles_c; ++t) {
        int off = p->c_tile_nnz_ptr[t], nnz_t = p->c_tile_nnz_ptr[t + 1] - off;           /* :626-627 */
        for (int n = 0; n < nnz_t; ++n) {
            int r = p->c_rowcolidx[off + n] >> 4, c = p->c_rowcolidx[off + n] & 0xF;      /* :645-646 */
            double acc = 0.0;
            for (int pr = p->pairs_offset[t]; pr < p->pairs_offset[t + 1]; ++pr) {        /* :635 */
                int a = p->pairs_a[pr], b = p->pairs_b[pr];
                int aoff = A->tile_nnz_ptr[a], boff = B->tile_nnz_ptr[b];                 /* :639-640 */
                unsigned lane_mask = A->masks[16 * (size_t)a + r] & B->masks_t[16 * (size_t)b + c];   /* :647 */
                while (lane_mask) {
                    int ffs = __builtin_ctz(lane_mask);                                    /* :650 */
                    int a_o = popc16(A->masks[16 * (size_t)a + r] & (0xFFFFu >> (16 - ffs)));          /* :651 */
                    int b_o = popc16(B->masks[16 * (size_t)b + ffs] & (0xFFFFu >> (16 - c)));          /* :652 */
                    double av = A->vals[aoff + A->rowptr[16 * (size_t)a + r] + a_o];
                    double bv = B->vals[boff + B->rowptr[16 * (size_t)b + ffs] + b_o];
                    acc = f32 ? (double)fmaf((float)av, (float)bv, (float)acc) : fma(av, bv, acc);   /* :653 */
                    lane_mask &= ~(1u << ffs);                                             /* :655 */
                }
            }
            p->c_vals[off + n] = acc;
        }
    }
    return 0;
}

int oracle_spgemm_step3(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p) { return step3_impl(A, B, p, 0); }

/* the same chain in float -- operands are float values held in doubles, one fmaf per product (the
 * reference's kernel is a template on ValueType, spgemm.cu:593; only main pins double, spgemm.cu:728). */
int oracle_spgemm_step3_f32(const oracle_tiled *A, const oracle_tiled *B, oracle_cplan *p) { return step3_impl(A, B, p, 1); }

/* a14 sanitize_C (spgemm.cu:663-695) + stable_sort by (row, col, val) (spgemm.cu:1516-1519).
 * (row, col) pairs are unique, so the order is fully determined by (row, col). */
int oracle_c_export_coo(const oracle_cplan *p, int *rows, int *cols, double *vals)
{
    trip *tr = ALLOC(trip, p->nnz_c);
    for (int t = 0; t < p->ntiles_c; ++t) {
        int ty = p->c_tile_rowidx[t], tx = p->c_tile_colidx[t];
        for (int e = p->c_tile_nnz_ptr[t]; e < p->c_tile_nnz_ptr[t + 1]; ++e) {
            tr[e].i = (ty << 4) + (p->c_rowcolidx[e] >> 4);                 /* :689 */
            tr[e].j = (tx << 4) + (p->c_rowcolidx[e] & 0xF);                /* :690 */
            tr[e].v = p->c_vals[e];
        }
    }
    qsort(tr, (size_t)p->nnz_c, sizeof(trip), cmp_trip);
    for (int64_t e = 0; e < p->nnz_c; ++e) { rows[e] = tr[e].i; cols[e] = tr[e].j; vals[e] = tr[e].v; }
    free(tr);
    return 0;
}

int oracle_c_export_csr(const oracle_cplan *p, int rows_a, int *rowptr, int *colidx, double *vals)
{
    int r0 = p->tr_lo * 16, r1 = p->tr_hi * 16 < rows_a ? p->tr_hi * 16 : rows_a;
    int nrows = r1 - r0;
    int *rows = ALLOC(int, p->nnz_c);
    oracle_c_export_coo(p, rows, colidx, vals);
    for (int r = 0; r <= nrows; ++r) rowptr[r] = 0;
    for (int64_t e = 0; e < p->nnz_c; ++e) rowptr[rows[e] - r0 + 1]++;
    for (int r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
    free(rows);
    return 0;
}
