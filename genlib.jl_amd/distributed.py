"""Multi-GPU gen.phi with storage-sharded level matrices (one process per GPU, torch.distributed).

The default multi-GPU partition of this package needs no collective: every level that fits one GPU is
replicated and only the last level is row-sharded (bench.py).  This module is the other
regime: when two level matrices no longer fit one GPU's HBM (the reference keeps two dense matrices alive,
src/compute.jl:291,301), every rank stores a COLUMN PANEL of every level (1/world of it) and the ranks
exchange, before each level step, the parent columns of their new members -- one all-to-all per level
over RCCL/xGMI (backend "nccl"), or over gloo through host memory in rehearsals.

    rows, r0 = phi_panels(pedigree, probandIDs)      # this rank's rows [r0, r0 + len(rows)) of Phi
"""
import numpy as np

from . import _capi


def _exchange(dist, send, send_cols, recv_cols, col_floats, device):
    """All-to-all of whole columns: send_cols[d] columns to rank d, recv_cols[s] from rank s."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    n_recv = int(recv_cols.sum()) * col_floats
    recv = torch.empty(max(n_recv, 1), dtype=torch.float32, device=device)
    if world == 1 or (int(send_cols.sum()) == 0 and n_recv == 0 and False):
        return recv
    in_splits = [int(c) * col_floats for c in send_cols]
    out_splits = [int(c) * col_floats for c in recv_cols]
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(recv[:n_recv], send[:sum(in_splits)], output_split_sizes=out_splits, input_split_sizes=in_splits)
        return recv
    # gloo (rehearsals, CPU tensors): pairwise non-blocking sends / receives through host memory
    hs = send[:sum(in_splits)].cpu()
    hr = torch.empty(n_recv, dtype=torch.float32)
    rank = dist.get_rank()
    so = np.concatenate([[0], np.cumsum(in_splits)])
    ro = np.concatenate([[0], np.cumsum(out_splits)])
    hr[ro[rank]:ro[rank + 1]] = hs[so[rank]:so[rank + 1]]
    reqs = []
    for peer in range(world):
        if peer == rank:
            continue
        if out_splits[peer]:
            reqs.append(dist.irecv(hr[ro[peer]:ro[peer + 1]], src=peer))
        if in_splits[peer]:
            reqs.append(dist.isend(hs[so[peer]:so[peer + 1]].contiguous(), dst=peer))
    for r in reqs:
        r.wait()
    recv[:n_recv].copy_(hr)
    return recv


def panel_sweep(pl, dist, dev):
    """One gen.phi sweep on an existing PanelPlan: begin, then pack / all-to-all / compute per level
    step.  The result stays resident (this rank's column panel of Phi).  Returns bytes sent."""
    import torch
    pl.begin(device=dev.index)
    sent = 0
    for step in range(pl.n_steps):
        s_cols, r_cols, cf = pl.exchange_counts(step)
        send = torch.empty(max(int(s_cols.sum()) * cf, 1), dtype=torch.float32, device=dev)
        pl.pack(step, send.data_ptr())
        recv = _exchange(dist, send, s_cols, r_cols, cf, dev)
        torch.cuda.synchronize(dev)
        pl.compute(step, recv.data_ptr())
        sent += int(s_cols.sum()) * cf * 4
    return sent


def phi_panels(pedigree, probandIDs, dist=None, device=None, stats=None):
    """gen.phi(pedigree, probandIDs) with column-panel storage across the ranks of `dist`
    (torch.distributed, already initialised; None = a single rank).  Returns (rows, row_begin): this
    rank's block of rows of Phi in proband order, float32.  Bit-identical to gen.phi."""
    import torch
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    pl = _capi.PanelPlan(pedigree.ind, pedigree.father, pedigree.mother, np.asarray(probandIDs, dtype=np.int64), rank, world)
    try:
        sent = panel_sweep(pl, dist, dev)
        if stats is not None:
            stats["exchange_bytes_sent"] = sent
            stats["panel_device_bytes"] = pl.device_bytes
        r0, _ = pl.result_rows()
        return pl.result_to_host(), r0
    finally:
        pl.close()


def replicated_levels_fit(cut_sizes, free_bytes):
    """The size test of SURVEY.md 8(e): do two consecutive level matrices plus the result fit one GPU?
    (pitch = multiple of 64 floats >= n + 1, as the library lays them out)"""
    pitch = lambda n: (n + 1 + 63) // 64 * 64          # noqa: E731
    need = 0
    for a, b in zip(cut_sizes[:-1], cut_sizes[1:]):
        need = max(need, 4 * ((a + 1) * pitch(a) + (b + 1) * pitch(b)))
    return need <= 0.92 * free_bytes
