/*
 * pem_mgpu.h -- C ABI of libpemmgpu.so: the row-block SpGEMM across the GPUs of ONE process.
 *
 * The reference is single-GPU (one device, spgemm.cu:730-758); nothing in it is replaced here.  This is the
 * C/C++ side of north_star's "partitioned across the 8 GPUs of one node by 1-D row-block splits of A with B
 * replicated and per-rank C slices gathered over RCCL/xGMI": one pem_ctx per device, one host thread per device
 * driving it (contexts are independent, include/pem_spgemm.h), and one exchange step -- the gather of the CSR
 * slices to a root device with grouped ncclSend/ncclRecv (RCCL; each slice travels on its own xGMI link into
 * the root).  `pemspgemm --gpus N` is built on it; bench.py's N>1 path does the same with one PROCESS per GPU
 * over torch.distributed.
 */
#ifndef PEM_MGPU_H
#define PEM_MGPU_H
#include <stdint.h>
#include "pem_spgemm.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pem_mgpu pem_mgpu;

/* One context per listed device + an RCCL communicator over them (ncclCommInitAll).  ndev >= 1. */
pem_status pem_mgpu_create(int ndev, const int *devices, pem_mgpu **out);
pem_status pem_mgpu_destroy(pem_mgpu *m);
int pem_mgpu_size(const pem_mgpu *m);
pem_ctx *pem_mgpu_ctx(pem_mgpu *m, int rank);

/* Where slice g lands in the assembled CSR: exclusive prefix sums of the slices' row and entry counts
 * (row_off / nnz_off have n+1 entries, the last = totals).  Pure host arithmetic; exported so that it is tested
 * without a GPU. */
void pem_mgpu_slice_offsets(int n, const int64_t *nrows, const int64_t *nnz, int64_t *row_off, int64_t *nnz_off);
/* rowptr of the assembled CSR from the slices' relative row pointers (slice g: nrows[g]+1 entries starting at 0):
 * out[row_off[g] + r] = slice_g[r] + nnz_off[g], out[total rows] = total nnz.  Host arithmetic, tested on CPU. */
void pem_mgpu_rebase_rowptr(int n, const int64_t *nrows, const int32_t *const *slice_rowptr, const int64_t *row_off,
                            const int64_t *nnz_off, int32_t *out);

/* The exchange step.  plans[g] is rank g's finished plan (step 3 done) over its tile-row block of A, blocks in rank
 * order.  Every rank exports its slice as CSR on its own device; column indices and values travel to `root` in one
 * RCCL group (ncclSend / ncclRecv, slice g straight into its place in the root's arrays); the small row pointers are
 * copied out by their owners and rebased on the host.  Size query: rowptr == NULL -> only *nnz and *nrows are set.
 * With host buffers given, the assembled CSR is copied out of the root device into them.  gather_ms: host wall of
 * export + RCCL group on all ranks (what `t_gather` reports), excluding the final device-to-host copy. */
pem_status pem_mgpu_gather_csr(pem_mgpu *m, pem_cplan *const *plans, int root, int64_t *nrows, int64_t *nnz, int32_t *rowptr,
                               int32_t *colidx, double *vals, double *gather_ms);

#ifdef __cplusplus
}
#endif
#endif
