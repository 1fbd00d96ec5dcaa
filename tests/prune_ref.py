"""What the oracle's (reference-faithful) plan arrays look like once dead products are dropped.

The HIP step 1 prunes every tile-level product (A tile, B tile) whose A tile has no entry in a column
that is an occupied row of the B tile: it cannot contribute to C.  The reference keeps those products
as pairs and the C tiles made only of them as empty tiles.  This module derives, from the oracle's
arrays, the arrays the pruned pipeline must produce: same pairs minus the dead ones, same tiles minus
the ones left without a pair, offsets recomputed; masks / row pointers / entries / values untouched.
"""
import numpy as np

C_NAMES = ["c_tile_rowptr", "c_tile_rowidx", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr",
           "c_rowptr", "c_rowcolidx", "c_vals"]


def expected(op, oA, oB, prune=True):
    """-> (dict name -> array, (ntiles_c, npairs, nnz_c, npairs_all))"""
    if not prune:
        return {n: getattr(op, n) for n in C_NAMES}, (op.ntiles_c, op.npairs, op.nnz_c, op.npairs)
    am = oA.masks.reshape(-1, 16)
    bm = oB.masks.reshape(-1, 16)
    colocc = np.bitwise_or.reduce(am, axis=1).astype(np.uint32) if len(am) else np.zeros(0, np.uint32)
    rowocc = ((bm != 0).astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(axis=1).astype(np.uint32) if len(bm) else np.zeros(0, np.uint32)
    live = (colocc[op.pairs_a] & rowocc[op.pairs_b]) != 0
    TC = op.ntiles_c
    tile_of_pair = np.repeat(np.arange(TC), np.diff(op.pairs_offset))
    live_cnt = np.bincount(tile_of_pair[live], minlength=TC)
    keep = live_cnt > 0
    nnz_t = np.diff(op.c_tile_nnz_ptr)
    assert np.all(nnz_t[~keep] == 0) and np.all(nnz_t[keep] > 0), "a C tile is non-empty iff it has a live pair"
    mt = op.tr_hi - op.tr_lo
    rowidx = op.c_tile_rowidx[keep]
    out = dict(
        pairs_a=op.pairs_a[live], pairs_b=op.pairs_b[live],
        pairs_offset=np.concatenate([[0], np.cumsum(live_cnt[keep])]).astype(np.int32),
        c_tile_rowidx=rowidx, c_tile_colidx=op.c_tile_colidx[keep],
        c_tile_rowptr=np.concatenate([[0], np.cumsum(np.bincount(rowidx - op.tr_lo, minlength=mt))]).astype(np.int32),
        c_mask=op.c_mask.reshape(-1, 8)[keep].ravel(),
        c_tile_nnz_ptr=np.concatenate([[0], np.cumsum(nnz_t[keep])]).astype(np.int32),
        c_rowptr=op.c_rowptr.reshape(-1, 16)[keep].ravel(),
        c_rowcolidx=op.c_rowcolidx, c_vals=op.c_vals)
    return out, (int(keep.sum()), int(live.sum()), op.nnz_c, op.npairs)
