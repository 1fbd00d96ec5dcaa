"""1-D row-block SpGEMM across the GPUs of one node (new work -- the reference is
single-GPU).  One process per GPU; A is split by tile rows (multiples of 16 matrix rows),
B is replicated, every rank runs steps 1-3 on its slice with no communication, and the CSR
slices of C are gathered to rank 0 with point-to-point sends: on the xGMI full mesh each
slice travels on its own direct link into the root.

Backend-agnostic (RCCL on GPUs, gloo in the CPU tests): tensors live wherever the caller
put them.
"""
import torch
import torch.distributed as dist


def slice_bounds(bounds, rank):
    """tile-row range [lo, hi) of `rank` from the pem_split_tile_rows boundaries"""
    return int(bounds[rank]), int(bounds[rank + 1])


def row_bounds_from_transpose(b_tile_rowptr, b_tile_colptr, b_tile_rowidx, nparts):
    """pem_split_tile_rows for A = B^T without tiling A: A's tile (i, k) exists iff B's tile (k, i) does, so tile
    row i of A is tile column i of B (read off B's tile CSC) and its product count is the sum of the lengths of
    B's tile rows k over that column.  Same weights and cut rule as the device routine (products + tiles + 1)."""
    import numpy as np
    blen = np.diff(np.asarray(b_tile_rowptr, dtype=np.int64))
    colptr = np.asarray(b_tile_colptr, dtype=np.int64)
    mt = len(colptr) - 1
    col_of = np.repeat(np.arange(mt), np.diff(colptr))
    w = np.bincount(col_of, weights=blen[np.asarray(b_tile_rowidx, dtype=np.int64)], minlength=mt) + np.diff(colptr) + 1.0
    pre = np.concatenate([[0.0], np.cumsum(w)])
    bounds, row = [0], 0
    for g in range(1, nparts):
        target = pre[mt] * g / nparts
        while row < mt and pre[row + 1] <= target:
            row += 1
        bounds.append(row)
    bounds.append(mt)
    return np.asarray(bounds, dtype=np.int32)


def recut_bounds(weights, bounds, times, fixed=0.0):
    """Row-block boundaries re-cut from MEASURED per-rank pass times.  `weights`: the per-tile-row weights the first cut balanced
    (pem_tile_row_weights), `bounds`: the cut that was timed, `times[p]`: rank p's time per pass on bounds[p]..bounds[p+1].
    Model: a rank's pass costs `fixed` (launch structure, latency chains: what no row carries) plus its rows' weights at the rate
    measured on that rank, rate_p = (times[p] - fixed) / weight of part p.  Rows keep the rate of the part they were timed in;
    the new cut gives every rank the same share of the summed cost.  Pure arithmetic (no device, no communication)."""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    b = np.asarray(bounds, dtype=np.int64)
    t = np.asarray(times, dtype=np.float64)
    nparts, mt = len(b) - 1, len(w)
    assert b[0] == 0 and b[-1] == mt and len(t) == nparts and np.all(np.diff(b) >= 0)
    pre = np.concatenate([[0.0], np.cumsum(w)])
    part_w = pre[b[1:]] - pre[b[:-1]]
    var = np.maximum(t - fixed, 0.05 * t)                       # (a rank is never modelled as costing nothing)
    rate = np.where(part_w > 0, var / np.maximum(part_w, 1e-300), 0.0)
    cost = w * np.repeat(rate, np.diff(b))
    cpre = np.concatenate([[0.0], np.cumsum(cost)])
    out, row = [0], 0
    for g in range(1, nparts):
        target = cpre[mt] * g / nparts
        while row < mt and cpre[row + 1] <= target:
            row += 1
        out.append(row)
    out.append(mt)
    return np.asarray(out, dtype=np.int32)


def tune_row_bounds(pkg, ctx, A, B, bounds, rank, world, device, passes=20, rounds=3, graph=True, group=None, gain=0.03):
    """Measure-and-recut of the 1-D row split (setup work, before anything is timed): every rank times `passes` repeat passes of
    its block, the times are all-gathered, and the blocks are re-cut (recut_bounds) so that ranks whose rows cost more per unit
    of weight -- oversized tile rows with their serial sort chain, sparse C tiles -- get fewer of them.  The first-cut weights are
    tile-level product counts; on the webbase-1M stand-in they leave the eight ranks 0.26-0.31 ms apart (re-cut: 0.29 for the
    slowest).  Returns the cut with the smallest maximum over ranks -- the first cut unless a later one beat it by `gain` (pass
    times move by a few percent from run to run) -- and the history [(bounds, per-rank ms)] of every round."""
    import time
    import numpy as np
    weights = pkg.tile_row_weights(ctx, A, B)
    best, best_max, history = np.asarray(bounds, dtype=np.int32), None, []
    cur = best
    for rnd in range(rounds + 1):
        lo, hi = slice_bounds(cur, rank)
        plan = pkg.CPlan(ctx, A, B, lo, hi)
        if graph:
            ctx.set_graph_replay(True)
        for _ in range(3):
            plan.spgemm()
        ctx.synchronize()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(passes):
            plan.spgemm()
        ctx.synchronize()
        mine = (time.perf_counter() - t0) * 1e3 / passes
        plan.close()
        tt = torch.zeros(world, dtype=torch.float64, device=device)
        tt[rank] = mine
        dist.all_reduce(tt, group=group)
        times = tt.cpu().numpy()
        history.append((cur.copy(), times.copy()))
        if best_max is None:
            best, best_max, first_max = cur.copy(), float(times.max()), float(times.max())
        elif times.max() < best_max and times.max() < (1.0 - gain) * first_max:
            best, best_max = cur.copy(), float(times.max())
        if rnd == rounds or times.max() <= 1.04 * times.mean():
            break
        # the part of a pass that no row carries: what the fastest rank would still pay with no rows -- taken as 60 % of the
        # fastest time on the first round (launch structure + one row's latency chain), nothing later (the rates then hold it)
        cur = recut_bounds(weights, cur, times, fixed=0.6 * float(times.min()) if rnd == 0 else 0.5 * float(times.min()))
    return best, history


def _drain(t):
    """Block the host until the communication just waited on has really finished.  Under RCCL `req.wait()` only makes
    torch's current stream wait; the library exports into the send buffers on ITS OWN stream, so the next pass's export
    must not start before the transfer that reads them is done (gloo's wait() already blocks the host)."""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def gather_csr_slices(rowptr, colidx, vals, dst=0, group=None):
    """Gather per-rank CSR row slices (rowptr relative, starting at 0) into one CSR on `dst`.

    rowptr: int32 [nrows_r + 1], colidx: int32 [nnz_r], vals: float64 (float32 for fp32 tilings) [nnz_r] on this rank.
    Returns (rowptr, colidx, vals) of the whole C on `dst`, None elsewhere.  Slices are
    concatenated in rank order, so with tile-row-aligned splits the result equals the 1-GPU
    CSR bit for bit (structure and values).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = rowptr.device
    meta = torch.tensor([rowptr.numel() - 1, colidx.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu()
    nrows = metas[:, 0].tolist()
    nnzs = metas[:, 1].tolist()
    if rank != dst:
        ops = [dist.P2POp(dist.isend, rowptr, dst, group=group)]
        if nnzs[rank] > 0:
            ops.append(dist.P2POp(dist.isend, colidx, dst, group=group))
            ops.append(dist.P2POp(dist.isend, vals, dst, group=group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        _drain(rowptr)
        return None
    tot_rows, tot_nnz = sum(nrows), sum(nnzs)
    out_rp = torch.zeros(tot_rows + 1, dtype=torch.int32, device=dev)
    out_ci = torch.empty(tot_nnz, dtype=torch.int32, device=dev)
    out_v = torch.empty(tot_nnz, dtype=vals.dtype, device=dev)   # float64, or float32 for fp32 tilings
    rp_parts = [None] * world
    ops = []
    roff = noff = 0
    offs = []
    for r in range(world):
        offs.append((roff, noff))
        if r == dst:
            rp_parts[r] = rowptr
            out_ci[noff:noff + nnzs[r]] = colidx
            out_v[noff:noff + nnzs[r]] = vals
        else:
            rp_parts[r] = torch.empty(nrows[r] + 1, dtype=torch.int32, device=dev)
            ops.append(dist.P2POp(dist.irecv, rp_parts[r], r, group=group))
            if nnzs[r] > 0:
                ops.append(dist.P2POp(dist.irecv, out_ci[noff:noff + nnzs[r]], r, group=group))
                ops.append(dist.P2POp(dist.irecv, out_v[noff:noff + nnzs[r]], r, group=group))
        roff += nrows[r]
        noff += nnzs[r]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for r in range(world):   # rebase the relative row pointers by the nnz offset of the slice
        ro, no = offs[r]
        out_rp[ro + 1:ro + nrows[r] + 1] = rp_parts[r][1:] + no
    _drain(out_rp)               # the copies out of the caller's colidx / vals are done before it may re-export into them
    return out_rp, out_ci, out_v


# ------------------------------------------------------------------------------------------------------------------
# 2-D (row block of A) x (column block of B) partition, for inputs whose replicated B does not fit.
# Rank (i, j) of an R x Cb grid holds only rows block i of A and columns block j of B (both cut from the COO before
# upload, indices left global so tile ids of C are global) and computes the whole block C(i, j) = A(i, :) * B(:, j)
# with no communication: every C entry still sees its full k range in ascending order, so values are bit-identical
# to the 1-GPU result.  The blocks are gathered to one rank and spliced row by row (column blocks in order).
# ------------------------------------------------------------------------------------------------------------------
def grid_coords(rank, ncol_blocks):
    """(row block, column block) of `rank` in a row-major R x Cb grid"""
    return rank // ncol_blocks, rank % ncol_blocks


def balanced_tile_bounds(index, extent, nparts):
    """Boundaries (in tiles of 16) cutting [0, ceil(extent/16)) into `nparts` ranges of about equal nnz.
    index: row (or column) index of every nonzero; returns nparts+1 ascending tile indices."""
    import numpy as np
    ntile = (int(extent) + 15) // 16
    cnt = np.bincount(np.asarray(index, dtype=np.int64) >> 4, minlength=ntile).astype(np.int64)
    pre = np.concatenate([[0], np.cumsum(cnt)])
    bounds = [0]
    for g in range(1, nparts):
        target = pre[-1] * g / nparts
        t = int(np.searchsorted(pre, target, side="left"))
        bounds.append(min(max(t, bounds[-1]), ntile))
    bounds.append(ntile)
    return bounds


def restrict(index, lo_tile, hi_tile):
    """boolean mask of the nonzeros whose `index` (row or column) lies in tiles [lo_tile, hi_tile)"""
    import numpy as np
    index = np.asarray(index)
    return (index >= 16 * lo_tile) & (index < 16 * hi_tile)


def assemble_csr_blocks(blocks, ncol_blocks):
    """Splice the CSR blocks of an R x Cb grid (row-major list of (rowptr, colidx, vals); rowptr relative to the
    block's first row, column indices global and ascending across column blocks) into one CSR.
    Works on any device; returns (rowptr int32, colidx int32, vals)."""
    nrb = len(blocks) // ncol_blocks
    dev = blocks[0][0].device
    vdt = blocks[0][2].dtype
    rp_parts, entries = [], []
    row_base = 0
    lens_all = []
    for i in range(nrb):
        row = blocks[i * ncol_blocks:(i + 1) * ncol_blocks]
        nrows = row[0][0].numel() - 1
        lens = torch.stack([(b[0][1:] - b[0][:-1]).to(torch.int64) for b in row])      # [Cb, nrows]
        lens_all.append(lens)
        row_base += nrows
    tot_len = torch.cat([l.sum(0) for l in lens_all]) if lens_all else torch.zeros(0, dtype=torch.int64, device=dev)
    out_rp = torch.zeros(tot_len.numel() + 1, dtype=torch.int64, device=dev)
    out_rp[1:] = torch.cumsum(tot_len, 0)
    nnz = int(out_rp[-1])
    out_ci = torch.empty(nnz, dtype=torch.int32, device=dev)
    out_v = torch.empty(nnz, dtype=vdt, device=dev)
    row0 = 0
    for i in range(nrb):
        lens = lens_all[i]
        nrows = lens.shape[1]
        before = torch.cumsum(lens, 0) - lens                                            # entries of lower column blocks, per row
        for j in range(ncol_blocks):
            rp, ci, v = blocks[i * ncol_blocks + j]
            if ci.numel() == 0:
                continue
            rows = torch.repeat_interleave(torch.arange(nrows, device=dev), lens[j])     # block-local row of every entry
            within = torch.arange(ci.numel(), device=dev) - rp[:-1].to(torch.int64)[rows]
            dst = out_rp[row0 + rows] + before[j][rows] + within
            out_ci[dst] = ci
            out_v[dst] = v
        row0 += nrows
    return out_rp.to(torch.int32), out_ci, out_v


def gather_csr_blocks(rowptr, colidx, vals, ncol_blocks, dst=0, group=None):
    """2-D counterpart of gather_csr_slices: every rank contributes the CSR of its block (rank = i * Cb + j);
    `dst` receives them point to point and splices them.  Returns the whole CSR on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = rowptr.device
    meta = torch.tensor([rowptr.numel() - 1, colidx.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().tolist()
    if rank != dst:
        ops = [dist.P2POp(dist.isend, rowptr, dst, group=group)]
        if metas[rank][1] > 0:
            ops.append(dist.P2POp(dist.isend, colidx, dst, group=group))
            ops.append(dist.P2POp(dist.isend, vals, dst, group=group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        _drain(rowptr)
        return None
    blocks, ops = [], []
    for r in range(world):
        if r == dst:
            blocks.append((rowptr, colidx, vals))
            continue
        nr, nz = metas[r]
        b = (torch.empty(nr + 1, dtype=torch.int32, device=dev), torch.empty(nz, dtype=torch.int32, device=dev),
             torch.empty(nz, dtype=vals.dtype, device=dev))
        ops.append(dist.P2POp(dist.irecv, b[0], r, group=group))
        if nz > 0:
            ops.append(dist.P2POp(dist.irecv, b[1], r, group=group))
            ops.append(dist.P2POp(dist.irecv, b[2], r, group=group))
        blocks.append(b)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out = assemble_csr_blocks(blocks, ncol_blocks)
    _drain(out[0])
    return out


# ------------------------------------------------------------------------------------------------------------------
# SURVEY 8(f)-4, second half: overlap the gather with the computation of later tile-row chunks.  A rank's row block is
# cut into chunks (one plan each, so repeat passes stay warm); chunk c is exported on the device and handed to the
# communication backend -- asynchronous point-to-point, on the backend's own streams -- while chunk c+1 runs steps
# 1-3 on the library's stream.  The root receives into per-(rank, chunk) staging buffers and concatenates them at the
# end; chunk sizes are exchanged once (A and B are immutable, so they repeat) and cached.
# ------------------------------------------------------------------------------------------------------------------
def concat_csr_slices(slices):
    """row-wise concatenation of CSR slices [(rowptr relative, colidx, vals), ...] in list order"""
    dev = slices[0][0].device
    nrows = [s[0].numel() - 1 for s in slices]
    nnzs = [s[1].numel() for s in slices]
    out_rp = torch.zeros(sum(nrows) + 1, dtype=torch.int32, device=dev)
    out_ci = torch.empty(sum(nnzs), dtype=torch.int32, device=dev)
    out_v = torch.empty(sum(nnzs), dtype=slices[0][2].dtype, device=dev)
    ro = no = 0
    for (rp, ci, v), nr, nz in zip(slices, nrows, nnzs):
        out_rp[ro + 1:ro + nr + 1] = rp[1:] + no
        out_ci[no:no + nz] = ci
        out_v[no:no + nz] = v
        ro += nr
        no += nz
    return out_rp, out_ci, out_v


class ChunkedRowBlock:
    """One rank's row block as `nchunks` plans whose CSR exports are sent while the next chunk computes."""

    def __init__(self, pkg, ctx, A, B, bounds, rank, nchunks, torch_dtype, dst=0, group=None):
        # bounds: pem_split_tile_rows(ctx, A, B, world * nchunks) -- product-balanced over ALL chunks of all ranks
        self.pkg, self.ctx, self.dst, self.group = pkg, ctx, dst, group
        self.rank, self.world, self.nchunks = rank, dist.get_world_size(group), nchunks
        self.plans = [pkg.CPlan(ctx, A, B, int(bounds[rank * nchunks + c]), int(bounds[rank * nchunks + c + 1])) for c in range(nchunks)]
        self.vdt = torch_dtype
        self.bufs = [None] * nchunks
        self.metas = None          # [world][nchunks] (nrows, nnz), known after the first pass
        self.stage = None

    def _export(self, c):
        p = self.plans[c]
        info = p.info()
        nrows, nz = info["row_end"] - info["row_begin"], info["nnz_c"]
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.bufs[c] is None or self.bufs[c][1].numel() != nz:
            self.bufs[c] = (torch.empty(nrows + 1, dtype=torch.int32, device=dev), torch.empty(nz, dtype=torch.int32, device=dev),
                            torch.empty(nz, dtype=self.vdt, device=dev))
        rp, ci, v = self.bufs[c]
        dummy = rp   # export needs non-null pointers only when nnz > 0
        p.export_csr_device(rp.data_ptr(), ci.data_ptr() if nz else dummy.data_ptr(), v.data_ptr() if nz else dummy.data_ptr())
        self.ctx.synchronize()      # the export ran on the library's stream; the backend reads the buffers on its own
        return nrows, nz

    def _ensure_stage(self):
        if self.rank != self.dst or self.stage is not None:
            return
        dev = torch.device("cuda", torch.cuda.current_device())
        self.stage = [[None if r == self.dst else
                       (torch.empty(self.metas[r][c][0] + 1, dtype=torch.int32, device=dev),
                        torch.empty(self.metas[r][c][1], dtype=torch.int32, device=dev),
                        torch.empty(self.metas[r][c][1], dtype=self.vdt, device=dev)) for c in range(self.nchunks)]
                      for r in range(self.world)]

    def run_pass(self):
        """all chunks: compute, export, send (overlapped).  Returns the whole C as CSR on `dst`, None elsewhere."""
        first = self.metas is None      # the root cannot post receives before it knows the sizes
        works, my_meta = [], []
        if not first:
            self._ensure_stage()
        for c in range(self.nchunks):
            self.plans[c].spgemm()
            my_meta.append(self._export(c))
            if not first:
                works += self._post(c)          # chunk c travels while chunk c+1 computes
        if first:
            dev = torch.device("cuda", torch.cuda.current_device())
            t = torch.tensor(my_meta, dtype=torch.int64, device=dev)
            allm = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(allm, t, group=self.group)
            self.metas = [m.cpu().tolist() for m in allm]
            self._ensure_stage()
            for c in range(self.nchunks):
                works += self._post(c)
        for w in works:
            w.wait()
        out = self._assemble()
        # every send that reads this pass's export buffers, and the root's copies out of them, are finished before the
        # next pass exports into the same buffers on the library's stream
        torch.cuda.current_stream().synchronize()
        return out

    def _post(self, c):
        ops = []
        if self.rank != self.dst:
            rp, ci, v = self.bufs[c]
            ops.append(dist.P2POp(dist.isend, rp, self.dst, group=self.group))
            if ci.numel():
                ops.append(dist.P2POp(dist.isend, ci, self.dst, group=self.group))
                ops.append(dist.P2POp(dist.isend, v, self.dst, group=self.group))
        else:
            for r in range(self.world):
                if r == self.dst:
                    continue
                rp, ci, v = self.stage[r][c]
                ops.append(dist.P2POp(dist.irecv, rp, r, group=self.group))
                if ci.numel():
                    ops.append(dist.P2POp(dist.irecv, ci, r, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, v, r, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _assemble(self):
        if self.rank != self.dst:
            return None
        slices = []
        for r in range(self.world):
            for c in range(self.nchunks):
                slices.append(self.bufs[c] if r == self.dst else self.stage[r][c])
        return concat_csr_slices(slices)
