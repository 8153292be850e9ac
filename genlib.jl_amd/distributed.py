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


def comm_selftest(dist, dev):
    """Communicator check before the first sweep: one all_to_all_single of rank-stamped columns (the collective of
    _exchange, uneven splits included) and one MAX all-reduce on `dev` tensors, verified on every rank.  With the
    "nccl" backend this is RCCL -- also at world size 1, where the exchange lists themselves are empty."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    on_dev = dist.get_backend() == "nccl"
    d = dev if on_dev else torch.device("cpu")
    # rank r sends (d + 1) * 3 floats to rank d, each stamped 1000 r + d
    in_splits = [(q + 1) * 3 for q in range(world)]
    out_splits = [(rank + 1) * 3] * world
    send = torch.cat([torch.full((n,), 1000.0 * rank + q, dtype=torch.float32) for q, n in enumerate(in_splits)]).to(d)
    recv = torch.full((sum(out_splits),), -1.0, dtype=torch.float32, device=d)
    dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits)
    want = torch.cat([torch.full((out_splits[q],), 1000.0 * q + rank, dtype=torch.float32) for q in range(world)])
    ok = bool(torch.equal(recv.cpu(), want))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=d)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    return {"backend": dist.get_backend(), "world": world, "device_tensors": on_dev, "all_to_all_single_ok": ok}


def panel_sweep(pl, dist, dev, ordered=True):
    """One gen.phi sweep on an existing PanelPlan: begin, then pack / all-to-all / compute per level
    step.  The result stays resident (this rank's column panel of Phi).  Returns bytes sent.

    The send / receive buffers are allocated once per plan at the largest step's size.  Ordering between
    the library's stream and torch's is by EVENTS, not by host synchronisation (genphi_panel_pack_on /
    _compute_on): torch's current stream waits for the packed columns before the collective starts, the
    library's stream waits for the collective before it unpacks, and the host runs ahead enqueueing the
    next step; one synchronisation at the end of the sweep.  ordered=False is the round-3 form (a stream
    synchronisation after pack, after the collective and after the level kernels of every step: A/B)."""
    import time
    import torch
    pl.begin(device=dev.index)
    counts = getattr(pl, "_xcounts", None)
    if counts is None:
        counts = pl._xcounts = [pl.exchange_counts(step) for step in range(pl.n_steps)]
    bufs = getattr(pl, "_xbufs", None)
    if bufs is None:
        n_s = max([int(s.sum()) * cf for s, _, cf in counts] + [1])
        n_r = max([int(r.sum()) * cf for _, r, cf in counts] + [1])
        bufs = pl._xbufs = (torch.empty(n_s, dtype=torch.float32, device=dev), torch.empty(n_r, dtype=torch.float32, device=dev))
    send, recv = bufs
    sp, rp = send.data_ptr(), recv.data_ptr()
    sent = 0
    t_begin = time.perf_counter()
    if ordered:
        world = dist.get_world_size() if dist is not None else 1
        # raw hipStream_t the collective is enqueued on; one rank: no collective, nothing to order across streams (GENPHI_NO_STREAM)
        stream = torch.cuda.current_stream(dev).cuda_stream if world > 1 else -1
        for step, (s_cols, r_cols, cf) in enumerate(counts):
            pl.pack_on(step, sp, stream)
            _exchange(dist, send, recv, s_cols, r_cols, cf)
            pl.compute_on(step, rp, stream)
            sent += int(s_cols.sum()) * cf * 4
        t_enq = time.perf_counter()
        pl.sync()
        t_end = time.perf_counter()
        pl.last_sweep = {"ordered_by": "events", "host_enqueue_ms": (t_enq - t_begin) * 1e3, "sweep_wall_ms": (t_end - t_begin) * 1e3}
        return sent
    t_pack = t_xchg = t_comp = 0.0
    for step, (s_cols, r_cols, cf) in enumerate(counts):
        t0 = time.perf_counter()
        pl.pack(step, sp)
        t1 = time.perf_counter()
        _exchange(dist, send, recv, s_cols, r_cols, cf)
        torch.cuda.current_stream(dev).synchronize()
        t2 = time.perf_counter()
        pl.compute(step, rp)
        t3 = time.perf_counter()
        t_pack += t1 - t0; t_xchg += t2 - t1; t_comp += t3 - t2
        sent += int(s_cols.sum()) * cf * 4
    # host wall time of the three phases of the last sweep (each ends with a stream synchronisation)
    pl.last_sweep = {"ordered_by": "host synchronisation", "pack_ms": t_pack * 1e3, "exchange_ms": t_xchg * 1e3, "compute_ms": t_comp * 1e3,
                     "sweep_wall_ms": (time.perf_counter() - t_begin) * 1e3}
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
    dragged = both_counts is not None and any(b > 0 for b in both_counts)
    if max(cut_sizes[:-1], default=0) > 36863 or dragged or both_counts is None:    # block-assembly steps (WIDE, or a run kept in place): psi_p, the scatter buffer
        total += max(need)
        # runs of steps may stay in place: ONE slot matrix instead of two ping-pong ones -- the planner drops the runs
        # when that needs more than 1.2 x the plain buffers and more than 4 GiB (PlanOptions::stay_mem_ratio,
        # stay_mem_floor_bytes), so that is the bound when members are known to be dragged along; with both_counts unknown only
        # the 0.2 x term is charged (a tiny pedigree must not be charged 4 GiB).  This formula is the fallback of callers without
        # a plan: genphi_plan_device_bytes_needed (PhiPlan.device_bytes_needed) knows the slot capacities and the result's pitch.
        extra = (need[0] + need[1]) // 5
        if dragged:
            extra = max(extra, (1 << 30) - (need[0] + need[1]))
        total += extra
    return 4 * total + 30 * sum(cut_sizes)


def replicated_levels_fit(cut_sizes, free_bytes, dist=None, device=None, both_counts=None, plan_bytes=None):
    """The size test of SURVEY.md 8(e): do the replicated level matrices plus the result fit one GPU?
    With `dist` the answer is made COLLECTIVE (MIN over the ranks): ranks see different amounts of free
    memory, and a rank that went down the exchange path while its peers went down the replicated one
    would wait in a collective nobody else enters."""
    wide_last = len(cut_sizes) >= 2 and cut_sizes[-2] > 36863      # the last step reads rows of the cut before it
    # (plan_bytes: what the plan itself says it will allocate -- slot matrices, the result at its pitch --, PhiPlan.device_bytes_needed)
    need = plan_bytes if plan_bytes is not None else replicated_bytes(cut_sizes, wide_last=wide_last, both_counts=both_counts)
    fits = need <= 0.92 * free_bytes
    if dist is not None and dist.get_world_size() > 1:
        import torch
        t = torch.tensor([1 if fits else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        fits = bool(int(t.item()))
    return fits
