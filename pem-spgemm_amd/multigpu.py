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
    return out_rp, out_ci, out_v
