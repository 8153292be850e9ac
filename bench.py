#!/usr/bin/env python3
"""bench.py -- dense gen.phi throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the hot path: all level steps of gen.phi (Psi = 1/2 I ... the
N x N proband matrix, delivered in proband order) for the synthetic pedigree of the named
workload, with the pedigree's flat index arrays already resident in HBM and the result left
resident in HBM.  Default workload = the configuration the metric is quoted on
(BASELINE.json configs[3]: 1e6 individuals / 1e5 probands / 30 generations; fits one GPU).

N > 1: one process per GPU; the upper levels are replicated (a per-level exchange over xGMI
would cost more than recomputing them), the final level is row-sharded, no
data-path collective.  Total work is fixed as N grows => "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A/B runs of this script steer the library through GENPHI_* environment hooks, which it reads only under GENPHI_ENV_HOOKS=1
_NOT_AB = ("GENPHI_TRACE", "GENPHI_D2H_THREADS", "GENPHI_D2H_SYM", "GENPHI_D2H_TILE", "GENPHI_ENV_HOOKS", "GENPHI_PLAN_CACHE", "GENPHI_KEEP_MB")
if any(k.startswith("GENPHI_") and k not in _NOT_AB for k in os.environ):
    os.environ.setdefault("GENPHI_ENV_HOOKS", "1")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (n_ind, n_pro, n_gen[, skip_permille]) of synth.random_mating, or a special tag
    "cfg4": (1_000_000, 100_000, 30),
    # cfg4 with overlapping generations: 0.5 % of the parents come from generation g-2, so ancestors are
    # reached at several distances and are dragged through the cuts (cuts up to 123k members, 80 % of a
    # cut dragged along, B = 1.45 TB): the WIDE block-assembly levels.  Not a BASELINE.json config.
    "cfg4o": (1_000_000, 100_000, 30, 5),
    "cfg3": (100_000, 10_000, 20),
    # cfg3 exactly as SURVEY.md 8(d) words it: 5 % of the parents come from generation g-2 "to exercise the dragged
    # path" (cuts to 20,540 members, up to 91 % of a cut dragged along, B = 26.25 GB, 1.66e9 pair evaluations).
    "cfg3s": (100_000, 10_000, 20, 50),
    "cfg2": "genea140",
    # the same real genealogy with EVERY individual a proband (the full kinship matrix of a genealogy: 41,523 x 41,523): nobody ever
    # leaves the cuts, B = 101 GB of which all but the new rows and columns is dragged x dragged copy (src/compute.jl:108-110)
    "cfg2all": "genea140_all",
    # ... and with a random quarter of its individuals (10,380 probands, ancestors among them at every depth)
    "cfg2q": "genea140_quarter",
    "cfg5": "deep_inbred",
}


def measured_ceiling():
    """GB/s this access pattern (whole rows, 16-byte accesses, 3 reads : 2 writes, non-temporal stores,
    no reuse) moved in profiles/microbench/row_stream2.hip on an MI355X: parsed from the committed
    output of that run (context for roofline.frac, never the headline)."""
    path = os.path.join(ROOT, "profiles", "microbench", "out", "r02_row_stream2.out")
    try:
        for line in open(path):
            if "3 reads : 2 writes, nt stores" in line and "1024 thr" in line:
                return float(line.split()[-2]) * 1000.0, os.path.relpath(path, ROOT)
    except OSError:
        pass
    return None, None


def csrc_sha16(root=None):
    """Fingerprint of the kernel sources of the DENSE path (what the traffic files measure): ties a committed
    profiles/traffic_<workload>.json to the sources it was collected with (works on the GPU box, which has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root or ROOT, "genlib.jl_amd", "csrc")
    for name in ("genphi_hip.hip", "sparse_levels.hip", "sparse_levels.h", "planner.cpp", "planner.h", "panel_launch.h"):
        h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_workload(name):
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    w = WORKLOADS[name]
    if w == "genea140":
        ped = gen.genealogy(gen.genea140)
        return ped, gen.pro(ped), "genea140 bundled pedigree (41523 individuals, 140 probands)"
    if w in ("genea140_all", "genea140_quarter"):
        ped = gen.genealogy(gen.genea140)
        ids = np.asarray(ped.ind, dtype=np.int64)
        if w == "genea140_quarter":
            ids = np.sort(np.random.default_rng(7).choice(ids, size=len(ids) // 4, replace=False))
        return ped, ids, ("genea140 bundled pedigree (41523 individuals), " + ("every individual a proband" if w == "genea140_all" else
                          f"a random quarter of the individuals as probands ({len(ids)}, numpy default_rng(7))"))
    if w == "deep_inbred":
        ind, fa, mo, sex, pro = synth.deep_inbred(200, 50, 3)
        desc = "deep consanguineous synthetic pedigree (1e4 individuals, 200 generations x 50, 3 sires/generation)"
    else:
        skip = w[3] if len(w) > 3 else 0
        ind, fa, mo, sex, pro = synth.random_mating(w[0], w[1], w[2], skip_permille=skip)
        desc = (f"synthetic random-mating pedigree, {w[0]} individuals / {w[1]} probands / {w[2]} generations, "
                f"SplitMix64 seed {synth.SEED}" + (f", {skip} per mille of the parents from generation g-2" if skip else ""))
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    return ped, pro, desc


def cpu_baseline(ped, pro, sizes, budget_s=20.0):
    """The oracle (a C/OpenMP port of the reference algorithm, kind "port") timed on this host's cores on a bounded
    sample of the same workload: (i) the first level steps of the same pedigree, until about budget_s / 2 seconds of work,
    (ii) rows of the LAST level step (the proband matrix: a third of the evaluations of cfg4, random access into the widest
    source matrix) on the real index structure of that step.  Extrapolated to the whole sweep by (i <= j) kernel evaluations,
    the upper steps at the rate of (i), the last one at the rate of (ii)."""
    from oracle import oracle as O
    O.fit_threads_to_quota()                                   # (`cores` below = the threads actually used: the CPUs the container may use)
    oped = O.Pedigree(ped.ind, ped.father, ped.mother, sort=False)
    n_pro = sizes[-1]
    evals_upper = sum(n * (n + 1) // 2 for n in sizes[1:-1])
    evals_last = n_pro * (n_pro + 1) // 2
    # choose how many level steps to run: probe one step, then size the sample
    t0 = time.perf_counter()
    _, done1 = oped.phi(pro, stop_after_levels=1) if len(sizes) > 2 else (None, 0)
    t1 = time.perf_counter() - t0
    if len(sizes) <= 2 or done1 == 0:
        t0 = time.perf_counter()
        oped.phi(pro)
        t = time.perf_counter() - t0
        return {"value": n_pro * n_pro / t, "unit": "proband-pairs/s", "cores": O.num_threads(), "kind": "port",
                "sample": "the whole workload, one run of the C/OpenMP oracle"}
    k, t, done = 1, t1, done1
    while k < len(sizes) - 2 and t < budget_s / 3:
        # grow the sample geometrically in evaluations until it costs a few seconds
        target = max(done * 3, 1)
        acc, k2 = 0, 0
        for n in sizes[1:-1]:
            acc += n * (n + 1) // 2
            k2 += 1
            if acc >= target:
                break
        k = max(k + 1, min(k2, len(sizes) - 2))
        t0 = time.perf_counter()
        _, done = oped.phi(pro, stop_after_levels=k)
        t = time.perf_counter() - t0
    rate_upper = done / t
    # (ii) rows of the last level step, spread over the proband order, about a tenth of the budget
    n_rows = int(max(1, min(n_pro, 256)))
    rows = np.unique(np.linspace(0, n_pro - 1, n_rows).astype(np.int64))
    t_last, done_last = oped.time_level_rows(pro, len(sizes) - 2, rows)
    while t_last < budget_s / 20 and len(rows) < n_pro:
        rows = np.unique(np.linspace(0, n_pro - 1, min(n_pro, 4 * len(rows))).astype(np.int64))
        t_last, done_last = oped.time_level_rows(pro, len(sizes) - 2, rows)
    rate_last = done_last / t_last
    t_est = evals_upper / rate_upper + evals_last / rate_last
    return {"value": n_pro * n_pro / t_est, "unit": "proband-pairs/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"first {k} of {len(sizes) - 2} upper level steps of the same pedigree ({done:.3g} of {evals_upper:.3g} pair-kernel "
                      f"evaluations, {t:.1f} s) + {len(rows)} rows of the last level step ({done_last:.3g} evaluations on the step's real "
                      f"indices and a source matrix of the real size, {t_last:.2f} s; the step has {evals_last:.3g}); "
                      f"C/OpenMP oracle = port of src/compute.jl:105-158,233-304; extrapolated by evaluation count, each part at its own rate",
            "rates_evals_per_s": {"upper_levels": rate_upper, "last_level": rate_last}}


def sparse_summary(pl, sizes):
    """Which leading cuts the sweep keeps as lists of their non-zero entries, and how empty they are (counted on the GPU by the plan's
    calibrating run; csrc/sparse_levels.hip)."""
    k, nnz = pl.sparse_levels()
    return {"last_sparse_cut": k,
            "nonzero_frac": [round(nnz[c] / float(sizes[c]) ** 2, 6) if nnz[c] >= 0 else None for c in range(min(len(nnz), len(sizes)))]}


def level_bytes(pl, sizes, both, esz=4.0, product_sweep=True, sparse=True):
    """Per level step: the ALGORITHMIC bytes of SURVEY.md 8(d), esz (n_k^2 + n_{k+1}^2) -- the reference's own formulation, a fresh
    dense matrix per level, src/compute.jl:291,301 -- and the bytes this implementation still has to MOVE: the same, minus the
    dragged x dragged block of a step that stays in place (never copied: 2 esz n_dragged^2), and for the leading cuts kept as lists of
    their non-zero entries the lists instead of the matrices (8 bytes per entry; a list step reads the lists of both sources of every
    row, ~2 E_k entries, and writes E_{k+1}; the step that writes the first dense matrix reads ~2 E_k and writes n^2 floats)."""
    alg = [esz * (a * a + b * b) for a, b in zip(sizes[:-1], sizes[1:])]
    in_place = [bool(pl.step_slots(k)[0] & 1) for k in range(len(alg))]
    moved = [x - (esz * 2.0 * both[k] * both[k] if in_place[k] and product_sweep else 0.0) for k, x in enumerate(alg)]
    k_sp = pl.sparse_levels()[0] if sparse else -1         # (the Float32 product sweep and the Float64-storage sweep run them)
    if k_sp >= 1:
        ent = pl.sparse_entries()
        for k in range(k_sp + 1):
            if k < k_sp:
                moved[k] = 8.0 * (2.0 * ent[k] + ent[k + 1])
            else:
                moved[k] = 8.0 * 2.0 * ent[k] + esz * sizes[k + 1] * sizes[k + 1]
    return alg, moved, in_place


def settle_after_release(released_bytes):
    """Wait until the driver has cleared device memory this process just released.  Released VRAM is wiped in the background with the copy
    engines -- about 65 ms per GB -- and until that is done every device-to-host copy of the process runs at HALF its rate and a large
    hipMalloc waits (profiles/microbench/free_then_copy.hip, out/r05_free_then_copy.out: 256 MB in 8.9 instead of 4.7 ms after a 1 GB
    hipFree; the 40 GB result hipMalloc 0.3 ms -> ~1 s).  A measurement that follows the release of another workload's plan would time
    that, not its own call.  Returns the seconds waited (reported on the line)."""
    wait = min(8.0, float(released_bytes) / 12e9)
    if wait > 0.02:
        time.sleep(wait)
        return wait
    return 0.0


def call_walls(ped, pro, device, with_d2h=True, reps=3):
    """Wall clock of one-shot gen.phi calls on a warm device -- what a caller of the drop-in API pays per call, the sweep being a small
    part of it: `first` = a plan nobody has seen (planning, upload, calibration of the sparse cuts, sweep, copy to the host, release),
    `repeat` = the same call again (gen.phi keeps its last plans per pedigree: sweep + copy).  Medians of `reps` calls."""
    import genlib_jl_amd as gen
    first, repeat = [], []
    for _ in range(reps):
        p2 = gen.Pedigree(ped.ind, ped.father, ped.mother, ped.sex)          # a pedigree object without cached plans
        t0 = time.perf_counter()
        if with_d2h:
            res = gen.phi(p2, pro, device=device)         # (kept until the clock is read: unmapping a 400 MB array is 15-20 ms of the interpreter's, not the call's)
        else:
            pl = gen.plan(p2, pro); pl.compute_device(device=device); pl.close()
        first.append((time.perf_counter() - t0) * 1e3)
        res = None
        if with_d2h:
            t0 = time.perf_counter()
            res = gen.phi(p2, pro, device=device)
            repeat.append((time.perf_counter() - t0) * 1e3)
            res = None
    out = {"first_ms": float(np.median(first)), "includes_copy_to_host": bool(with_d2h)}
    if repeat:
        out["repeat_ms"] = float(np.median(repeat))
    return out


def quick_workload(name, device, steps=5):
    """One of the other BASELINE.json configurations, measured the way the headline is (per-level HIP events inside the library, K timed
    sweeps after a first call and a warm-up), in the same process: a compact record for `other_workloads`."""
    import genlib_jl_amd as gen
    ped, pro, desc = load_workload(name)
    t0 = time.perf_counter()
    pl = gen.plan(ped, pro)
    plan_ms = (time.perf_counter() - t0) * 1e3
    sizes, both = pl.levels()
    n = pl.n_probands
    t0 = time.perf_counter()
    pl.compute_device(device=device)
    first_call_ms = (time.perf_counter() - t0) * 1e3
    pl.compute_device(device=device)
    kernel_ms, perm_ms, level_ms = 0.0, 0.0, None
    t0 = time.perf_counter()
    for _ in range(steps):
        st = pl.compute_device(device=device, timing=True)
        kernel_ms += st.total_ms
        perm_ms += st.perm_ms
        lm = np.array(st.level_ms[:st.n_steps], dtype=np.float64)
        level_ms = lm if level_ms is None else level_ms + lm
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    for _ in range(2):
        pl.compute_device(device=device)
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.compute_device(device=device)
    replay_ms = (time.perf_counter() - t0) * 1e3 / steps
    byt_alg, byt, in_place = level_bytes(pl, sizes, both)
    ms = kernel_ms / steps
    alg_frac = (sum(byt_alg) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None
    moved_frac = (sum(byt) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None
    on_alg = alg_frac is not None and alg_frac <= 1.0 and not any(in_place)
    rec = {"workload": f"{name}: {desc}", "n_probands": n, "levels": len(sizes), "max_cut": max(sizes), "algorithmic_GB": sum(byt_alg) / 1e9,
           "ms": wall_ms, "kernel_ms": ms, "graph_replay_ms": replay_ms, "plan_ms": plan_ms, "first_call_ms": first_call_ms,
           "frac": alg_frac if on_alg else moved_frac, "frac_basis": "algorithmic bytes" if on_alg else "bytes still to be moved",
           "algorithmic_frac": alg_frac, "moved_frac": moved_frac,
           "in_place_steps": int(sum(in_place)), "pairs_per_s": n * n / (wall_ms * 1e-3),
           "level_ms": [round(float(x) / steps, 4) for x in level_ms] if level_ms is not None and len(level_ms) <= 40 else None}
    rec.update(sparse_summary(pl, sizes))
    tpath = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("csrc_sha16") == csrc_sha16() and tj.get("hbm_bytes_per_launch"):
                rec["real_traffic_frac"] = tj["hbm_bytes_per_launch"] * len(byt) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        except Exception:
            pass
    released = pl.device_bytes
    pl.close()
    settle_after_release(released)
    rec["call_wall"] = call_walls(ped, pro, device, with_d2h=n * n * 4 <= (2 << 30))
    settle_after_release(released)                        # (the one-shot calls released their plans too)
    return rec


def shard_rows(n, rank, world):
    """Final-level row shard [r0, r1) of `rank`: contiguous, disjoint, covering [0, n)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def dry_run(args, rank, world):
    """CPU rehearsal (gloo) of everything around the compute call for N > 1: the plan, the row
    shards, the barrier and the max-over-ranks reduction.  Prints one JSON line on rank 0."""
    import torch
    import torch.distributed as dist
    import genlib_jl_amd as gen
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    ped, pro, desc = load_workload(args.workload)
    pl = gen.plan(ped, pro)
    n = pl.n_probands
    r0, r1 = shard_rows(n, rank, world)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0 + 1e-3 * (rank + 1)          # distinct per rank: the MAX must win
    shards = [None] * world
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall_max = float(tt.item())
        dist.all_gather_object(shards, (r0, r1))
    else:
        wall_max, shards = wall, [(r0, r1)]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "n_probands": n, "shards": shards,
                          "wall_is_max": wall_max >= wall, "levels": len(pl.levels()[0])}), flush=True)
    pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_exchange(args, ped, pro, desc, dist, rank, local_rank, world, cut_sizes, fits):
    """Storage-sharded level matrices: every rank holds a column panel of every level and the ranks
    exchange parent columns before every level step (genlib_jl_amd/distributed.py).  world = 1 runs the
    same path on one rank (no exchange): the panel kernels' own number."""
    import torch
    from genlib_jl_amd import _capi, distributed as gdist
    dev = torch.device("cuda", local_rank)
    pl = _capi.PanelPlan(ped.ind, ped.father, ped.mother, np.asarray(pro, dtype=np.int64), rank, world)
    n = pl.n_probands

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    selftest = gdist.comm_selftest(dist, dev) if dist is not None else None
    for _ in range(max(args.warmup, 1)):
        sent = gdist.panel_sweep(pl, dist, dev, ordered=not args.panel_host_sync)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sent = gdist.panel_sweep(pl, dist, dev, ordered=not args.panel_host_sync)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([wall, float(sent)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, sent_max = float(tt[0].item()), float(tt[1].item())
    else:
        sent_max = float(sent)
    if rank == 0:
        K = args.steps
        B = sum(4.0 * (a * a + b * b) for a, b in zip(cut_sizes[:-1], cut_sizes[1:]))
        achieved = B / (wall / K) / 1e9
        # per rank (rank 0's own numbers): device time of every level step's kernels (HIP events inside the library),
        # against this rank's share (1 / world of the columns) of the level's algorithmic bytes
        step_ms = pl.step_ms()
        lvl_b = [4.0 * (a * a + b * b) / world for a, b in zip(cut_sizes[:-1], cut_sizes[1:])]
        big = int(np.argmax(lvl_b))
        dev_ms = float(sum(step_ms))
        per_rank = {"device_ms_per_sweep": dev_ms,
                    "frac": (sum(lvl_b) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dev_ms > 0 else None,
                    "largest_level": {"step": big, "GB": lvl_b[big] / 1e9, "ms": step_ms[big],
                                      "frac": lvl_b[big] / (step_ms[big] * 1e-3) / 1e9 / HBM_PEAK_GBS if step_ms[big] > 0 else None},
                    "host_wall_ms_last_sweep": getattr(pl, "last_sweep", None),
                    "note": "one rank's kernels timed with HIP events; in a single-device rehearsal the ranks' kernels share the GPU"}
        print(json.dumps({
            "metric": "proband-pairs/sec for dense Phi (gen.phi), 1e5 probands; % HBM roofline",
            "value": n * n / (wall / K), "unit": "proband-pairs/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "n_probands": n, "levels": len(cut_sizes), "storage": "f32",
                       "parallelism": f"column panels x{world}: every level storage-sharded, one all-to-all of parent columns per level step"
                                      + (" (all ranks share one GPU: rehearsal)" if args.single_device and world > 1 else ""),
                       "why_exchange": "forced" if fits else "two level matrices do not fit one GPU",
                       "exchange_bytes_sent_per_rank_max": sent_max, "comm_selftest": selftest,
                       # what the all-to-alls of a sweep would take at the xGMI peer bandwidth (7 links x ~153 GB/s per GPU, all used at once)
                       "exchange_ms_predicted_xgmi": sent_max / (7 * 153e9) * 1e3 if world > 1 else 0.0,
                       "panel_device_bytes": pl.device_bytes, "max_cut": max(cut_sizes),
                       "panel_step_modes": pl.step_modes() if hasattr(pl, "step_modes") else None,
                       "ordering": "host synchronisation per step" if args.panel_host_sync else "stream events (no host synchronisation inside a sweep)",
                       "timed": "host wall clock around whole sweeps (pack, exchange through "
                                + ("RCCL" if args.backend == "nccl" and world > 1 else "host memory (gloo)" if world > 1 else "nothing: one rank")
                                + ", unpack, level kernels), max over ranks"},
            "roofline": {"bound": "hbm", "kernel": "level_full_kernel / level_split_fast_kernel on column panels (panel_level_kernel beyond LDS)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS / (1 if args.single_device else world), "traffic": None,
                         "per_rank_kernels": per_rank,
                         "note": "achieved = algorithmic bytes of the whole job (4 sum(n_k^2 + n_{k+1}^2)) / time, exchange included; "
                                 "frac is per GPU (ranks that share one GPU in a rehearsal count as one)"},
        }), flush=True)
    pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_sparse(args):
    """gen.sparse_phi (src/compute.jl:321-447) on the GPU: one line with the device time of the sweep (HIP events
    inside the library), GB/s on algorithmic bytes 4 sum(n_old^2 + n_next^2) over the waves, and the largest wave's
    own fraction of the HBM peak.  --workload sparse140: genea140 with its 140 probands; sparse2k: a 2,000-proband
    synthetic pedigree (1e5 individuals, 20 generations)."""
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    if args.workload == "sparse140":
        ped = gen.genealogy(gen.genea140)
        pro, desc = gen.pro(ped), "genea140 bundled pedigree, gen.sparse_phi over its 140 probands"
    else:
        ind, fa, mo, sex, pro = synth.random_mating(100_000, 2_000, 20)
        ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
        desc = "synthetic random-mating pedigree, 100000 individuals / 2000 probands / 20 generations, gen.sparse_phi"
    best = None
    t_wall = []
    for k in range(args.warmup + args.steps):
        t0 = time.perf_counter()
        K = gen.sparse_phi(ped, pro, device=0)
        dt = time.perf_counter() - t0
        st = K.stats()
        if k >= args.warmup:
            t_wall.append(dt)
            if best is None or st["sweep_ms"] < best["sweep_ms"]:
                best = st
    n = len(pro)
    ms = float(np.mean([best["sweep_ms"]]))
    wm, wb = best["wave_ms"], best["wave_bytes"]
    big = int(np.argmax(wb))
    achieved = best["algorithmic_bytes"] / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "gen.sparse_phi sweep (secondary path): proband-pairs/s; % HBM roofline on algorithmic bytes",
        "value": n * n / (ms * 1e-3), "unit": "proband-pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "n_probands": n, "waves": best["n_waves"], "max_active": best["max_active"],
                   "algorithmic_GB": best["algorithmic_bytes"] / 1e9, "call_wall_ms_mean": float(np.mean(t_wall)) * 1e3,
                   "call_wall_note": "whole gen.sparse_phi call: pruning, queue simulation, upload, sweep, download (host + device)",
                   "wave_ms": [round(float(x), 4) for x in wm] if len(wm) <= 64 else None},
        "roofline": {"bound": "hbm", "kernel": "sparse_rows_compact_kernel + sparse_newnew_kernel + sparse_mirror_kernel",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "largest_wave": {"index": big, "GB": float(wb[big]) / 1e9, "ms": float(wm[big]),
                                      "frac": float(wb[big]) / (float(wm[big]) * 1e-3) / 1e9 / HBM_PEAK_GBS if wm[big] > 0 else None}},
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS) + ["sparse140", "sparse2k"])
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"],
                    help="f64: the secondary Float64 level sweep (GENPHI_FLAG_STORAGE_F64: what gen.f and pairwise phi(i, j) run); "
                         "N = 1 only, no D2H block, algorithmic bytes count 8 per entry")
    ap.add_argument("--no-d2h", action="store_true", help="skip the device-to-host copy of the end_to_end block")
    ap.add_argument("--no-others", action="store_true",
                    help="default run (cfg4, 1 GPU): do not measure the other BASELINE.json configurations (cfg2, cfg3, cfg3s, cfg5) for `other_workloads`")
    ap.add_argument("--no-call-wall", action="store_true", help="do not time one-shot gen.phi calls (profiling runs: only the sweeps of ONE plan in the process)")
    ap.add_argument("--no-sparse", action="store_true", help="A/B: every level as a dense matrix (GENPHI_FLAG_NO_SPARSE)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--exchange", action="store_true",
                    help="N > 1: storage-sharded levels (column panels) with an all-to-all before every level step, "
                         "instead of replicated levels; taken automatically when the level matrices do not fit one GPU "
                         "(also GENPHI_FORCE_EXCHANGE=1)")
    ap.add_argument("--pg", action="store_true",
                    help="initialise the torch.distributed process group at N = 1 too (with --exchange: the RCCL communicator is created "
                         "and a self-test all_to_all_single / all_reduce runs on it before the sweeps)")
    ap.add_argument("--panel-host-sync", action="store_true",
                    help="--exchange: order pack / collective / level kernels by host synchronisations per step (the round-3 form; A/B) "
                         "instead of stream events")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N > 1 plumbing: plan + shard + barrier + max-reduce, no compute")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if world > 1:
        # row shards: a proband cut that stayed in place would make every rank compute the whole last level before delivering its
        # rows; the proband step keeps its row kernel and its shards instead (the planner's cost model knows nothing about shards).
        # (Before the library reads its hooks: the gate is read once.)
        os.environ["GENPHI_STAY_LAST"] = "0"
        os.environ["GENPHI_ENV_HOOKS"] = "1"
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the gen.phi product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.pg and "RANK" not in os.environ:                     # not under the launcher: a one-rank group of its own
            os.environ.setdefault("MASTER_PORT", "29571")
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import genlib_jl_amd as gen
    if args.workload.startswith("sparse"):
        return run_sparse(args)
    ped, pro, desc = load_workload(args.workload)
    if world > 1 or args.exchange:
        from genlib_jl_amd import distributed as gdist
        probe = gen.plan(ped, pro)
        cut_sizes, both_counts = probe.levels()
        plan_bytes = probe.device_bytes_needed
        probe.close()
        # the decision is collective (MIN over the ranks): every rank must take the same path
        fits = gdist.replicated_levels_fit(cut_sizes, torch.cuda.mem_get_info()[0], dist=dist, device=torch.device("cuda", local_rank),
                                           both_counts=both_counts, plan_bytes=plan_bytes)
        if args.exchange or os.environ.get("GENPHI_FORCE_EXCHANGE") == "1" or not fits:
            return run_exchange(args, ped, pro, desc, dist, rank, local_rank, world, cut_sizes, fits)
    t_plan = time.perf_counter()
    pl = gen.plan(ped, pro)                                  # genphi_plan_create: levelisation + flat index arrays (host)
    plan_ms = (time.perf_counter() - t_plan) * 1e3
    sizes, both = pl.levels()
    n = pl.n_probands
    # final-level row shard of this rank (proband tiles across the GPUs; no collective)
    r0, r1 = shard_rows(n, rank, world)
    rows = (r0, r1) if world > 1 else None
    empty_shard = world > 1 and r1 == r0                     # more ranks than probands: nothing to compute here

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    f64 = args.storage == "f64"
    if f64 and world > 1:
        raise SystemExit("--storage f64 is a single-GPU secondary path")
    esz = 8.0 if f64 else 4.0

    def compute(**kw):
        return None if empty_shard else pl.compute_device(device=local_rank, kernel=args.kernel, rows=rows, storage64=f64, no_sparse=args.no_sparse, **kw)

    # the first call also uploads the index arrays and allocates the level matrices and the result
    t_first = time.perf_counter()
    compute()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(max(args.warmup - 1, 0)):
        compute()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, level_ms, perm_ms = 0.0, None, 0.0
    level_rows = None
    for _ in range(args.steps):
        st = compute(timing=True)
        if st is None:
            continue
        level_rows = [int(st.level_rows[k]) for k in range(st.n_steps)]
        kernel_ms += st.total_ms
        perm_ms += st.perm_ms
        lm = np.array(st.level_ms[:st.n_steps], dtype=np.float64)
        level_ms = lm if level_ms is None else level_ms + lm
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    # The timed region above runs the sweep eagerly with a HIP event after every level step (per-level times).
    # Outside it: the same sweep WITHOUT events, which the library replays from a captured hipGraph (sweeps of
    # >= 8 level steps) -- what a caller of gen.phi gets; matters for workloads of many short launches.
    replay_ms = None
    if world == 1 and not empty_shard:
        for _ in range(2):
            compute()                                        # an eager run with these arguments, then the capture
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        for _ in range(args.steps):
            compute()
        torch.cuda.synchronize()
        replay_ms = (time.perf_counter() - t0r) * 1e3 / args.steps

    # N > 1: every rank copies ITS row block to its host (the 40 GB copy of cfg4 split N ways: the end-to-end win of the row shards)
    shard_d2h_ms = None
    if world > 1 and not args.no_d2h and not f64:
        import psutil
        t_own = 0.0
        fits_host = psutil.virtual_memory().available > (r1 - r0) * n * 4 * world + (16 << 30)      # (all ranks of the node copy at once)
        barrier()                                          # (every rank, also one without rows: it is a collective)
        if not empty_shard and fits_host:
            t0d = time.perf_counter()
            host = pl.result_to_host()
            t_own = (time.perf_counter() - t0d) * 1e3
            del host
        tt = torch.tensor([t_own], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        shard_d2h_ms = float(tt.item())
    if rank == 0:
        K = args.steps
        ms_per_step = wall * 1e3 / K
        value = n * n / (wall / K)
        # roofline of the dominant kernel (the level kernel; every level step launches it once).
        # algorithmic bytes per launch = 4 (n_k^2 + n_{k+1}^2); for a sharded
        # final level the launch writes only its rows.
        # a launch that computes only part of a level's rows (the sharded last level; the upper levels
        # of a shard, restricted to its ancestors) is credited with the rows it wrote and the same
        # share of the source matrix: byt = 4 (n_k^2 + n_{k+1}^2) * rows / n_{k+1}
        byt = [esz * (a * a + b * b) for a, b in zip(sizes[:-1], sizes[1:])]
        if level_rows is not None:
            byt = [x * (min(r, b) / b if b else 1.0) for x, r, b in zip(byt, level_rows, sizes[1:])]
        lvl = (level_ms / K) if level_ms is not None and len(byt) else np.zeros(0)
        lvl_kernel = lvl.copy()
        if len(lvl_kernel):
            lvl_kernel[-1] -= perm_ms / K                 # the proband-order pass is a different kernel
        # the kernel that dominates the step: by accumulated time over the level steps of each kind
        modes = pl.step_modes()
        names = {0: "level_full_kernel", 1: "level_split_fast_kernel",
                 2: "block assembly (in place: rows_avg_t_kernel + the new x new sub-step; compacting: rows_compact_kernel + transposes)"}
        if f64:                                            # row-staged Float64 kernel up to 10,239-wide cuts, per-entry kernel beyond
            names = {m: "level_full64_kernel / level_naive64_kernel" for m in (0, 1, 2)}
        by_kernel = {}
        sp = sparse_summary(pl, sizes) if (args.kernel == 0 and not args.no_sparse) else {"last_sparse_cut": -1, "nonzero_frac": []}
        for k, t in enumerate(lvl_kernel):
            small = k < len(lvl_kernel) - 1 and sizes[k] <= 128 and sizes[k + 1] <= 128 and args.kernel == 0
            nm = "level_naive_kernel" if args.kernel == 1 else ("levels_small_kernel" if small else names.get(modes[k], "?"))
            if k <= sp["last_sparse_cut"]:
                nm = "sparse_step_kernel" if k < sp["last_sparse_cut"] else "sparse_dense_kernel"
            by_kernel[nm] = by_kernel.get(nm, 0.0) + float(t)
        dominant = max(by_kernel, key=by_kernel.get) if by_kernel else "level_split_kernel"
        # Levels that stay IN PLACE (persistent slots) never move their dragged x dragged block (src/compute.jl:108-110 copies it):
        # the algorithmic bytes 4 (n_k^2 + n_{k+1}^2) count it twice (read + write), so `frac` on them is no bandwidth figure and
        # can exceed 1.  For such steps `frac` / `achieved` use the bytes the formulation still has to move, 4 (n_k^2 + n_{k+1}^2)
        # - 8 n_dragged^2; the purely algorithmic figure stays under `algorithmic_frac`.  No in-place step: the two coincide.
        _, byt_moved, in_place = level_bytes(pl, sizes, both, esz, product_sweep=not f64 and args.kernel == 0, sparse=args.kernel == 0 and not args.no_sparse)
        share = [(min(r, b) / b if b else 1.0) for r, b in zip(level_rows, sizes[1:])] if level_rows is not None else [1.0] * len(byt)
        byt_alg = list(byt)
        byt = [x * f for x, f in zip(byt_moved, share)]
        tot_ms = float(lvl_kernel.sum())
        achieved_moved = float(sum(byt)) / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        achieved_alg = float(sum(byt_alg)) / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        # `frac`: on the algorithmic bytes (the bench contract; an implementation that elides traffic shows as algorithmic > measured,
        # SURVEY.md 8(d)) -- unless that is no bandwidth figure at all: steps that stay in place, or a sweep whose bytes are mostly zeros
        # that never move (genea140: 1.5 x the peak), then on the bytes still to be moved.  Both are always on the line.
        on_alg = achieved_alg <= HBM_PEAK_GBS and not (any(in_place) and not f64 and args.kernel == 0)
        achieved = achieved_alg if on_alg else achieved_moved
        tot_b = float(sum(byt_alg)) if on_alg else float(sum(byt))
        moved_total = pl.algorithmic_bytes * esz / 4.0 - (float(sum(byt_alg)) - float(sum(byt)))      # whole sweep: what still moves
        # end to end through the C-ABI: plan (host) + first call (upload, allocation,
        # one sweep) ... + a sweep + the device-to-host copy of the N x N result.  Never `value`.
        end_to_end = {"plan_ms": plan_ms, "first_call_ms": first_call_ms, "sweep_ms": ms_per_step}
        if shard_d2h_ms is not None:
            end_to_end["d2h_ms"] = shard_d2h_ms
            end_to_end["d2h_sample"] = f"every rank its own block of ~{(n + world - 1) // world} rows into a pageable host array, at the same time; MAX over the ranks"
            end_to_end["total_ms_plan_sweep_d2h"] = plan_ms + ms_per_step + shard_d2h_ms
            end_to_end["pairs_per_s"] = n * n / (end_to_end["total_ms_plan_sweep_d2h"] * 1e-3)
        if world == 1 and not args.no_d2h and not f64:
            import psutil
            need = n * n * 4
            if psutil.virtual_memory().available > need + (16 << 30):
                t0d = time.perf_counter()
                host = pl.result_to_host()
                end_to_end["d2h_ms"] = (time.perf_counter() - t0d) * 1e3
                end_to_end["d2h_GBs"] = need / (end_to_end["d2h_ms"] * 1e-3) / 1e9
                end_to_end["d2h_sample"] = "the whole N x N Float32 result into a pageable host array"
                del host
            else:
                k = max(1, min(n, (4 << 30) // max(n * 4, 1)))
                pl.compute_device(device=local_rank, kernel=args.kernel, rows=(0, k))
                t0d = time.perf_counter()
                host = pl.result_to_host()
                dt = time.perf_counter() - t0d
                end_to_end["d2h_ms"] = dt * 1e3 * n / k
                end_to_end["d2h_GBs"] = k * n * 4 / dt / 1e9
                end_to_end["d2h_sample"] = f"first {k} of {n} rows (host memory is short), scaled to N rows"
                del host
            tot = (plan_ms + ms_per_step + end_to_end["d2h_ms"]) * 1e-3
            end_to_end["total_ms_plan_sweep_d2h"] = tot * 1e3
            end_to_end["pairs_per_s"] = n * n / tot
        ceiling, ceiling_src = measured_ceiling()
        traffic, traffic_stale = None, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        # (the committed traffic files were collected with the default settings: not comparable under A/B hooks that change what moves)
        ab_hooks = [k for k in os.environ if k.startswith("GENPHI_") and k not in _NOT_AB]
        if os.path.exists(tpath) and not f64 and not ab_hooks:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_stale = tj.get("csrc_sha16") != csrc_sha16()      # collected with other kernel sources than these
            except Exception:
                traffic = None
        out = {
            "metric": ("Float64 level sweep (secondary path: gen.f, pairwise phi): proband-pairs/s; % HBM roofline on 8-byte entries" if f64
                       else "proband-pairs/sec for dense Phi (gen.phi), 1e5 probands; % HBM roofline"),
            "value": value, "unit": "proband-pairs/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "end_to_end": end_to_end,
            "config": {"workload": f"{args.workload}: {desc}", "n_probands": n, "levels": len(sizes),
                       "arithmetic": ("Float64 level matrices, Float64 accumulation (src/compute.jl:66-95 contract)" if f64 else
                                      "Float64 accumulation over Float32 level matrices (the reference's contract)"), "storage": args.storage,
                       "max_cut": max(sizes) if sizes else 0, "algorithmic_GB": pl.algorithmic_bytes * esz / 4.0 / 1e9,
                       "parallelism": f"final-level row shards x{world}, upper levels replicated" if world > 1 else "1 GPU",
                       "kernel_ms_per_step": kernel_ms / K, "proband_order_pass_ms": perm_ms / K,
                       "pairs_per_s_kernel_only": n * n / (kernel_ms / K * 1e-3) if kernel_ms > 0 else None,
                       "unordered_pairs_per_s": (n * (n + 1) // 2) / (wall / K),      # N(N+1)/2 entries
                       "level_ms": [round(float(x), 4) for x in lvl] if len(lvl) <= 64 else None,
                       "kernel_modes": pl.step_modes() if len(sizes) <= 65 else None,
                       # WIDE level steps that write their cut in place (persistent slots: only new rows / columns move)
                       "in_place_steps": sum(pl.step_slots(k)[0] & 1 for k in range(max(len(sizes) - 1, 0))),
                       # leading cuts kept as lists of their non-zero entries (sparse_levels.hip), and the share of non-zero entries
                       # the plan's calibrating run counted in them
                       "last_sparse_cut": sp["last_sparse_cut"], "nonzero_frac": sp["nonzero_frac"]},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_basis": "algorithmic bytes" if on_alg else "bytes still to be moved",
                         "algorithmic_frac": achieved_alg / HBM_PEAK_GBS, "moved_frac": achieved_moved / HBM_PEAK_GBS, "traffic": traffic,
                         # context, not the headline: what this access pattern (whole 96 KB rows, 16-byte
                         # accesses, 3 reads : 2 writes, no reuse) can move at all on this GPU, measured by
                         # profiles/microbench/row_stream.hip; and the rate of the REAL traffic when known
                         "measured_ceiling_GBs": ceiling, "measured_ceiling_source": ceiling_src,
                         "traffic_source": (f"profiles/traffic_{args.workload}.json (rocprofv3 PMC passes of this command; not re-collected by this run"
                                            + ("; STALE: collected with other kernel sources than this run's" if traffic_stale else "; same kernel sources as this run") + ")") if traffic else None,
                         "traffic_stale": traffic_stale,
                         # (the measured traffic covers EVERY kernel of a sweep, the proband-order pass included: its time counts here)
                         "real_traffic_GBs": (traffic * len(byt) / ((tot_ms + perm_ms / K) * 1e-3) / 1e9) if (traffic and tot_ms > 0) else None,
                         # the measured bytes against the peak: BELOW `frac` when levels stay in place (the dragged x dragged
                         # block counts in the algorithmic bytes but is never moved), above it when rows are re-read
                         "real_traffic_frac": (traffic * len(byt) / ((tot_ms + perm_ms / K) * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and tot_ms > 0) else None,
                         "launches_per_step": len(byt), "avg_launch_ms": tot_ms / max(len(byt), 1),
                         "largest_level": ({"step": int(np.argmax(byt)), "GB": max(byt) / 1e9, "ms": float(lvl_kernel[int(np.argmax(byt))]),
                                            "frac": max(byt) / (float(lvl_kernel[int(np.argmax(byt))]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                           if len(byt) and float(lvl_kernel[int(np.argmax(byt))]) > 0 else None),
                         "algorithmic_bytes_per_launch_avg": tot_b / max(len(byt), 1),
                         "whole_step_frac": (moved_total / (kernel_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if kernel_ms > 0 else None,
                         # host wall clock per sweep without per-level events (hipGraph replay for >= 8 level steps)
                         "graph_replay_ms_per_step": replay_ms,
                         "graph_replay_frac": (moved_total / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if replay_ms else None},
        }
        out["roofline"]["frac_note"] = ("`algorithmic_frac`: 4 sum(n_k^2 + n_{k+1}^2) (SURVEY.md 8(d)) over the level steps' time; `moved_frac`: the bytes this "
                                        "implementation still has to move (lists instead of matrices for the leading sparse cuts, no dragged x dragged block for "
                                        "steps that stay in place); `frac` = the algorithmic one unless steps stay in place or it exceeds 1 (then it is no "
                                        "bandwidth figure); `real_traffic_frac`: measured HBM traffic")
        if not args.no_cpu_baseline and world == 1 and not f64:       # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(ped, pro, sizes)
        settled_s = 0.0
        if world == 1:
            released = pl.device_bytes
            pl.close()                                     # (the measurements below make plans of their own: this one's memory first)
            pl = None
            settled_s += settle_after_release(released)
        if world == 1 and not f64 and args.kernel == 0 and not args.no_sparse and not args.no_call_wall:
            # a one-shot call through the drop-in API (result left resident when it is too large to copy twice within the run)
            out["end_to_end"]["call_wall"] = call_walls(ped, pro, local_rank, with_d2h=n * n * 4 <= (2 << 30), reps=3 if n <= 20000 else 1)
            settled_s += settle_after_release(released)
            out["end_to_end"]["call_wall"]["waited_for_released_memory_s"] = round(settled_s, 2)
        if args.workload == "cfg4" and world == 1 and not f64 and args.kernel == 0 and not args.no_others and not args.no_sparse and not ab_hooks:
            # the other BASELINE.json configurations in the same process (each a few ms per sweep): driver-timed evidence for them
            others = {}
            for w in ("cfg2", "cfg3", "cfg3s", "cfg5"):
                try:
                    others[w] = quick_workload(w, local_rank)
                except Exception as e:      # noqa: BLE001  (the headline must not die of a secondary measurement)
                    others[w] = {"error": f"{type(e).__name__}: {e}"}
            out["other_workloads"] = others
        print(json.dumps(out), flush=True)
    if pl is not None:
        pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
