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


def _exchange(dist, send, recv, send_cols, recv_cols, col_floats):
    """All-to-all of whole columns: send_cols[d] columns to rank d, recv_cols[s] from rank s, between the
    preallocated device buffers `send` and `recv` (the used prefixes)."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    if world == 1:
        return
    in_splits = [int(c) * col_floats for c in send_cols]
    out_splits = [int(c) * col_floats for c in recv_cols]
    n_send, n_recv = sum(in_splits), sum(out_splits)
    if dist.get_backend() == "nccl":
        # RCCL over xGMI; every rank calls it at every step (a rank with nothing to move passes empty splits)
        dist.all_to_all_single(recv[:n_recv], send[:n_send], output_split_sizes=out_splits, input_split_sizes=in_splits)
        return
    # gloo (rehearsals, CPU tensors): pairwise non-blocking sends / receives through host memory
    hs = send[:n_send].cpu()
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


def panel_sweep(pl, dist, dev):
    """One gen.phi sweep on an existing PanelPlan: begin, then pack / all-to-all / compute per level
    step.  The result stays resident (this rank's column panel of Phi).  Returns bytes sent.

    The send / receive buffers are allocated once per plan at the largest step's size.  Ordering between
    the library's stream and torch's: genphi_panel_pack returns when the packed columns are complete
    (the collective may start), and the collective is complete on torch's current stream before
    genphi_panel_compute is enqueued (one stream synchronisation per step, no device-wide ones)."""
    import torch
    pl.begin(device=dev.index)
    counts = [pl.exchange_counts(step) for step in range(pl.n_steps)]
    bufs = getattr(pl, "_xbufs", None)
    if bufs is None:
        n_s = max([int(s.sum()) * cf for s, _, cf in counts] + [1])
        n_r = max([int(r.sum()) * cf for _, r, cf in counts] + [1])
        bufs = pl._xbufs = (torch.empty(n_s, dtype=torch.float32, device=dev), torch.empty(n_r, dtype=torch.float32, device=dev))
    import time
    send, recv = bufs
    sent = 0
    t_pack = t_xchg = t_comp = 0.0
    for step, (s_cols, r_cols, cf) in enumerate(counts):
        t0 = time.perf_counter()
        pl.pack(step, send.data_ptr())
        t1 = time.perf_counter()
        _exchange(dist, send, recv, s_cols, r_cols, cf)
        torch.cuda.current_stream(dev).synchronize()
        t2 = time.perf_counter()
        pl.compute(step, recv.data_ptr())
        t3 = time.perf_counter()
        t_pack += t1 - t0; t_xchg += t2 - t1; t_comp += t3 - t2
        sent += int(s_cols.sum()) * cf * 4
    # host wall time of the three phases of the last sweep (each ends with a stream synchronisation)
    pl.last_sweep = {"pack_ms": t_pack * 1e3, "exchange_ms": t_xchg * 1e3, "compute_ms": t_comp * 1e3}
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
            stats["step_modes"] = pl.step_modes()
        r0, _ = pl.result_rows()
        return pl.result_to_host(), r0
    finally:
        pl.close()


def replicated_bytes(cut_sizes, wide_last=False, both_counts=None):
    """Device bytes of the replicated path for these cuts, as genphi_compute_device allocates them
    (csrc/genphi_hip.hip: upload_plan / ensure_level_buffers / the result): two ping-pong buffers sized by
    the largest EVEN and the largest ODD intermediate cut (not by the largest consecutive pair), the
    N x pitch(N) result, a second copy of the last level when its step is WIDE (proband-order delivery),
    the tail padding of each buffer, and ~30 bytes of index arrays per member per level.  The compacted
    parent matrix of block-assembly steps is bounded by the largest intermediate level.  both_counts (members
    dragged along per step): all zero means no step assembles blocks below the LDS width and nothing stays in place."""
    pitch = lambda n: (n + 1 + 63) // 64 * 64          # noqa: E731
    tail = 64 * 1024
    need = [0, 0]
    for c, n in enumerate(cut_sizes[:-1]):
        need[c & 1] = max(need[c & 1], (n + 1) * pitch(n) + tail)
    n_last = cut_sizes[-1]
    total = need[0] + need[1] + n_last * pitch(n_last)
    if wide_last:
        total += (n_last + 1) * pitch(n_last) + tail
    dragged = both_counts is None or any(b > 0 for b in both_counts)
    if max(cut_sizes[:-1], default=0) > 36863 or dragged:    # block-assembly steps (WIDE, or a run kept in place): psi_p, the scatter buffer
        total += max(need)
        # runs of steps may stay in place: ONE slot matrix instead of two ping-pong ones -- the planner drops the runs
        # when that needs more than 1.2 x the plain buffers and more than 4 GiB (PlanOptions::stay_mem_ratio,
        # stay_mem_floor_bytes), so that is the bound
        total += max((need[0] + need[1]) // 5, (1 << 30) - (need[0] + need[1]))
    return 4 * total + 30 * sum(cut_sizes)


def replicated_levels_fit(cut_sizes, free_bytes, dist=None, device=None, both_counts=None):
    """The size test of SURVEY.md 8(e): do the replicated level matrices plus the result fit one GPU?
    With `dist` the answer is made COLLECTIVE (MIN over the ranks): ranks see different amounts of free
    memory, and a rank that went down the exchange path while its peers went down the replicated one
    would wait in a collective nobody else enters."""
    wide_last = len(cut_sizes) >= 2 and cut_sizes[-2] > 36863      # the last step reads rows of the cut before it
    fits = replicated_bytes(cut_sizes, wide_last=wide_last, both_counts=both_counts) <= 0.92 * free_bytes
    if dist is not None and dist.get_world_size() > 1:
        import torch
        t = torch.tensor([1 if fits else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        fits = bool(int(t.item()))
    return fits
