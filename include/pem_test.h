/* pem_test.h -- test hooks and tuning switches of libpemspgemm_hip.so (gfx950 / MI355X only).
 *
 * Not part of the drop-in boundary (pem_spgemm.h): nothing the reference's main() does maps to these.  They are what the
 * parity tests and the measurement tools use to push small inputs through the code a large input selects, to A/B a kernel
 * variant against its baseline, and to test a device primitive on its own.  Same library, same entry points
 * (pem_cplan_set_option / pem_cplan_get_option take these values of pem_option as well).
 */
#ifndef PEM_TEST_H
#define PEM_TEST_H
#include "pem_spgemm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* further values of pem_option (pem_spgemm.h holds 0-4 and 7) */
#define PEM_OPT_S1_FORCE_KEY64 ((pem_option)5)  /* test hook: 64-bit sort keys in step 1's row sorts whatever B's width                        */
#define PEM_OPT_S1_XLCAP       ((pem_option)6)  /* test hook: tile rows with more live products take the oversized-row path (0: off)           */
#define PEM_OPT_S1_SERIAL      ((pem_option)8)  /* diagnostic: step 1's row bins one after the other instead of concurrently                   */
#define PEM_OPT_S3_DECODE      ((pem_option)9)  /* 1 (default): on plans with < 2 pairs per C tile step 3 reads (row, column) off the C masks and
                                                   Ctiles_rowColIdx is materialised on demand; 0: step 2 writes it on every pass               */
#define PEM_OPT_S1_XL_GLOBAL   ((pem_option)10) /* 0 (default): oversized tile rows are sorted one workgroup per row where their products lie;
                                                   1: all of them through one global radix sort on (row, tile column) (rows above 2^18 always) */
#define PEM_OPT_S3_EPW         ((pem_option)11) /* step 3: C entries per wave / 256 (0, default: chosen from the C tiles' density)             */
#define PEM_OPT_S3_IDX64       ((pem_option)12) /* test hook: the mask-decoding step 3 addresses with 64-bit indices whatever the sizes        */
#define PEM_OPT_S3_MARK        ((pem_option)13) /* 1 (default; pruned plans): entry -> tile lookup by LDS marks + one ballot; 0: shuffle search */
#define PEM_OPT_S3_XCD         ((pem_option)14) /* 1 (default): step 3's entry-per-lane kernels give XCD x the x-th contiguous eighth of C      */
#define PEM_OPT_S1_SEGMENTS    ((pem_option)15) /* 0 (default): one workgroup per tile row above the 8192-key bin; 1: such rows are sorted in
                                                   column-range segments, one workgroup per segment -- pays where a plan holds a handful of
                                                   them (webbase-1M's directory rows in a 1/8 row block), costs where it holds hundreds         */
#define PEM_OPT_S2_TRANSPOSED  ((pem_option)16) /* step 2's boolean tile product: 0 from A's row masks (one trip per nonzero of the A tile), 1 from A's
                                                   transposed masks (one trip per inner index occupied on both sides), 2 (default): 1 on repeat
                                                   passes of plans with two or more pairs per C tile (cage15-class), else 0            */

/* The device exclusive scan (replaces thrust::exclusive_scan, spgemm.cu:1168, 1242, 1288, and NSPARSE/utils_cuda_scan.h)
 * on a caller's array: out[0..n] = exclusive prefix sums, out[n] = *total.  regime 0: chosen by n like the hot path;
 * 1: one block; 2: the single-launch chained scan (n <= 262144); 3: the three-launch scan.  in_place scans the device
 * copy in place (as the hot path does); stall_ticket >= 0 makes that block of the chained scan stall for ~0.1 ms before
 * it publishes, so every later block sits out a long wait. */
pem_status pem_debug_scan_i32(pem_ctx *ctx, const int32_t *in, int64_t n, int regime, int in_place, int stall_ticket,
                              int32_t *out, int64_t *total);

/* Launches a kernel with a block size the runtime must refuse (2048 threads) on the context's stream and returns PEM_OK, as
 * every asynchronous launch does: the refusal is kept in the context and comes back as PEM_E_HIP -- with the kernel's name in
 * pem_last_error() -- from the next call that synchronises (pem_ctx_synchronize, the end of a pass, a conversion). */
pem_status pem_debug_refused_launch(pem_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
