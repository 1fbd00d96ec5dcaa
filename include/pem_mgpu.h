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

/* Row-block boundaries re-cut from MEASURED per-rank pass times (setup work of an N-GPU run, before anything is timed).
 * weights[0 .. mt): pem_tile_row_weights; bounds[0 .. nparts]: the cut that was timed; ms[p]: rank p's time per pass on its
 * block.  Model: a rank's pass costs fixed_ms (launch structure, latency chains: what no row carries) plus its rows' weights at
 * the rate measured on that rank; rows keep the rate of the part they were timed in, and the new cut gives every rank the same
 * share of the summed cost.  Host arithmetic (tested on CPU against the Python harness' multigpu.recut_bounds). */
pem_status pem_mgpu_recut_bounds(int nparts, int mt, const double *weights, const int32_t *bounds, const double *ms, double fixed_ms,
                                 int32_t *out);

/* One pass of the whole product with the exchange overlapped: every rank's row block is cut into `nchunks` chunks
 * (plans[rank * nchunks + c], tile rows abutting in that order; pem_split_tile_rows over n * nchunks parts balances them),
 * one host thread per device computes its chunks in order, and chunk c's CSR export travels to `root` over RCCL -- straight
 * into its place in the root's assembled arrays -- while chunk c + 1 computes.  The first call on a set of plans runs them once
 * to learn the chunks' sizes (the root posts its receives by size); later calls are the overlapped pass alone.
 * pass_ms: from the moment all ranks start to the moment the last one has finished steps 1-3 of its last chunk;
 * tail_ms: from there until every transfer has landed (the part of the exchange the compute did not hide).
 * rowptr == NULL: the pass runs and the assembled C stays on the root device (sizes in *nrows / *nnz). */
pem_status pem_mgpu_spgemm_gather_chunked(pem_mgpu *m, pem_cplan *const *plans, int nchunks, int root, int64_t *nrows, int64_t *nnz,
                                          int32_t *rowptr, int32_t *colidx, double *vals, double *pass_ms, double *tail_ms);

#ifdef __cplusplus
}
#endif
#endif
