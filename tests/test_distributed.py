"""N > 1 path.  gen.phi shards by final-level proband rows with NO data-path collective
(upper levels replicated), so what needs covering is the plumbing around the
compute call: row shards tile [0, N), the barrier / max-over-ranks timing, one JSON line from
rank 0.  CPU: world_size-2 gloo dry run.  GPU: two ranks sharing cuda:0 over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc)] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_shard_rows_tile_the_matrix():
    sys.path.insert(0, ROOT)
    import bench
    for n in (0, 1, 3, 140, 100_000):
        for world in (1, 2, 3, 8):
            sh = [bench.shard_rows(n, r, world) for r in range(world)]
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sh[:-1], sh[1:]))
            assert all(0 <= a <= b <= n for a, b in sh)


def test_replicated_memory_model_matches_the_allocations():
    """distributed.replicated_bytes models what genphi_compute_device allocates (two ping-pong buffers by
    the largest even / odd intermediate cut, the result, tail padding, index arrays), not the largest pair
    of consecutive levels: the size test that chooses between replicated levels and column panels."""
    sys.path.insert(0, ROOT)
    from genlib_jl_amd import distributed as gdist
    pitch = lambda n: (n + 1 + 63) // 64 * 64      # noqa: E731
    cuts = [10, 5000, 200, 7000, 3000, 9000, 100, 4000]           # the two largest intermediate cuts are not consecutive
    none = [0] * (len(cuts) - 1)                                  # no member dragged along: no block assembly, nothing in place
    b = gdist.replicated_bytes(cuts, both_counts=none)
    even, odd = max(cuts[0:-1:2]), max(cuts[1:-1:2])
    want = 4 * ((even + 1) * pitch(even) + (odd + 1) * pitch(odd) + 4000 * pitch(4000))
    assert want <= b <= want + 4 * 2 * 65536 + 30 * sum(cuts) + 1024
    pair = max(4 * ((a + 1) * pitch(a) + (c + 1) * pitch(c)) for a, c in zip(cuts[:-1], cuts[1:]))
    assert b > pair                                                # the old pairwise model under-estimated
    assert gdist.replicated_levels_fit(cuts, b / 0.92 + 1, both_counts=none) and not gdist.replicated_levels_fit(cuts, b / 0.92 - 1e6, both_counts=none)
    # members dragged along: the parent matrix of block-assembly steps and the slot matrix of in-place runs on top (the planner keeps
    # runs whose slot matrices stay below 4 GiB whatever the plain buffers are); unknown: only the proportional terms -- a tiny
    # pedigree is not charged 4 GiB (and a plan says exactly what it needs: PhiPlan.device_bytes_needed)
    some = [1] * (len(cuts) - 1)
    assert 4 * (1 << 30) <= gdist.replicated_bytes(cuts, both_counts=some) <= b + 4 * (1 << 30) + 4 * (9001 * pitch(9000) + 65536)
    assert b < gdist.replicated_bytes(cuts) <= b + 4 * (9001 * pitch(9000) + 65536) + b // 4 + 1024 < 4 * (1 << 30)
    # cfg4: 3.8 GB + 2.4 GB of level matrices + the 40 GB result
    cfg4 = [24301] * 24 + [24650, 25190, 26502, 30976, 100000]
    assert 45e9 < gdist.replicated_bytes(cfg4, both_counts=[0] * 28) < 48e9
    assert gdist.replicated_bytes(cfg4) < 55e9


def test_two_rank_gloo_dry_run():
    d = _launch(2, ["--workload", "cfg2", "--dry-run"], 29533)
    assert d["dry_run"] and d["n_gpus"] == 2 and d["n_probands"] == 140
    assert d["shards"] == [[0, 70], [70, 140]] and d["wall_is_max"] and d["levels"] == 18


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    d = _launch(2, ["--workload", "cfg3", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                    "--single-device", "--no-cpu-baseline"], 29534)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["unit"] == "proband-pairs/s" and d["roofline"]["bound"] == "hbm"


def test_panel_exchange_lists_are_consistent():
    """Storage-sharded gen.phi (column panels, include/genphi.h genphi_panel_*): host-side check, no
    GPU -- what rank a sends to rank b before a step is what b expects from a, the result row blocks
    tile [0, N), and a single rank exchanges nothing."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth, _capi
    ind, fa, mo, sex, pro = synth.random_mating(4000, 301, 9, skip_permille=100, seed=2)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    for world in (1, 2, 3, 8):
        plans = [_capi.PanelPlan(ped.ind, ped.father, ped.mother, pro, r, world) for r in range(world)]
        blocks = [p.result_rows() for p in plans]
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == len(pro)
        assert all(blocks[k][0] + blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
        total = 0
        for step in range(plans[0].n_steps):
            cnt = [p.exchange_counts(step) for p in plans]
            for a in range(world):
                assert cnt[a][0][a] == 0 and cnt[a][1][a] == 0                  # nothing to oneself
                for b in range(world):
                    assert cnt[a][0][b] == cnt[b][1][a]                          # a -> b is what b expects from a
                assert cnt[a][2] == cnt[0][2]
                total += int(cnt[a][0].sum())
        assert (total == 0) == (world == 1)
        if world > 1:                                  # a rank holds about 1 / world of what a single rank holds
            one = _capi.PanelPlan(ped.ind, ped.father, ped.mother, pro, 0, 1)
            assert all(0 < p.device_bytes < 0.9 * one.device_bytes for p in plans)
            one.close()
        for p in plans:
            p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [1, 2, 3])
def test_panel_mode_matches_the_oracle(nproc):
    """The exchange path end to end: `nproc` ranks (gloo; all on cuda:0) each hold a column panel of
    every level, exchange parent columns before every step, and their row blocks reassemble to the
    oracle's matrix bit for bit (geneaJi, overlapping generations, one-parent members, ancestors among
    the probands)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "panel_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(29540 + nproc), worker]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    assert d["ok"] and d["world"] == nproc, d
    assert (d["exchange_bytes_sent_rank0"] > 0) == (nproc > 1)
    assert d["modes_seen"] == [0, 1, 2]          # FULL and SPLIT row kernels on the panel's columns, and the per-entry fallback


@pytest.mark.gpu
def test_bench_exchange_mode_two_ranks_one_gpu():
    """bench.py --exchange: the storage-sharded path (column panels + an all-to-all per level step)
    through the same launcher contract, two gloo ranks sharing cuda:0."""
    d = _launch(2, ["--workload", "cfg5", "--steps", "1", "--warmup", "1", "--backend", "gloo", "--single-device",
                    "--no-cpu-baseline", "--exchange"], 29537)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["why_exchange"] == "forced"
    assert d["config"]["exchange_bytes_sent_per_rank_max"] > 0


@pytest.mark.gpu
def test_rccl_communicator_at_one_rank():
    """The collective of the exchange path on RCCL, as far as one GPU allows: bench.py --exchange under the launcher with ONE rank,
    backend "nccl" -- the process group and its RCCL communicator are created, distributed.comm_selftest runs the very
    all_to_all_single of _exchange (uneven splits) and a MAX all-reduce on device tensors, the sweeps run with the group alive,
    and the timing reduction goes through an RCCL all-reduce.  (More than one rank per GPU is refused by RCCL: the multi-rank
    rehearsals above use gloo.)"""
    d = _launch(1, ["--workload", "cfg3", "--steps", "3", "--warmup", "1", "--backend", "nccl", "--no-cpu-baseline", "--exchange", "--pg"], 29538)
    st = d["config"]["comm_selftest"]
    assert st == {"backend": "nccl", "world": 1, "device_tensors": True, "all_to_all_single_ok": True}, st
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["exchange_bytes_sent_per_rank_max"] == 0
    # no host synchronisation inside a sweep: the wall time of a sweep is the kernels' time plus little
    per_rank = d["roofline"]["per_rank_kernels"]
    assert d["ms_per_step"] <= 1.35 * per_rank["device_ms_per_sweep"] + 0.15, (d["ms_per_step"], per_rank)


def test_survey_literal_cfg4_needs_panels_and_fits_eight_ranks():
    """SURVEY.md 8(d)'s cfg4 taken literally (5 % of the parents from generation g-2): cuts to 206,808 members, B = 3.74 TB -- the
    north star's "N exceeds one GPU's HBM" case.  Host-side only: the replicated path does not fit 288 GB, a column panel at world = 8
    does with a wide margin, every rank's exchange volume and the largest step's are what DESIGN.md 6.2 tabulates (so that the first
    8-GPU lease is a measurement, not a debugging session): ~66 GB sent per rank and sweep, <= 4.1 GB in one step, i.e. ~61 ms of
    xGMI time per sweep at 7 links x 153 GB/s when the peers are evenly loaded."""
    sys.path.insert(0, ROOT)
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth, _capi, distributed as gdist
    ind, fa, mo, sex, pro = synth.random_mating(1_000_000, 100_000, 30, skip_permille=50)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    pl = gen.plan(ped, pro)
    sizes, both = pl.levels()
    assert max(sizes) == 206_808 and 3.7e12 < pl.algorithmic_bytes < 3.8e12
    pl.close()
    hbm = 288e9
    assert not gdist.replicated_levels_fit(sizes, hbm, both_counts=both)        # two level matrices of 2e5 members: 340 GB
    world = 8
    sent = []
    for r in (0, 5):
        pp = _capi.PanelPlan(ped.ind, ped.father, ped.mother, pro, r, world)
        assert pp.device_bytes < 0.25 * hbm, pp.device_bytes                   # 54.7 GB per rank, result block included
        assert pp.result_rows() == (12_500 * r, 12_500)
        assert set(pp.step_modes()) <= {0, 1}                                  # every step through the row kernels (panel rows fit in LDS)
        tot, worst = 0, 0
        for step in range(pp.n_steps):
            s_cols, r_cols, cf = pp.exchange_counts(step)
            assert s_cols[r] == 0 and r_cols[r] == 0
            tot += int(s_cols.sum()) * cf * 4
            worst = max(worst, int(s_cols.sum()) * cf * 4, int(r_cols.sum()) * cf * 4)
            # evenly loaded peers: no peer gets more than twice the mean share of a step that moves something real
            if s_cols.sum() > 7000:
                assert s_cols.max() <= 2 * s_cols.sum() / (world - 1)
        sent.append(tot)
        assert 60e9 < tot < 70e9 and worst < 4.2e9
        pp.close()
    xgmi_s = max(sent) / (7 * 153e9)
    assert 0.055 < xgmi_s < 0.066


def test_scale_script_rehearsal():
    """profiles/scale.sh -- the script the first 8-GPU lease runs -- rehearsed on the CPU: its launcher lines, ports and the summary
    table, with --dry-run ranks over gloo (N = 1, 2)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out", "scale_rehearsal")
    env = dict(os.environ, SCALE_NS="1 2", SCALE_DRY="1", SCALE_OUT=out_dir, SCALE_DRY_WORKLOAD="cfg2")
    r = subprocess.run(["bash", os.path.join(root, "profiles", "scale.sh")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rows_n1 rc=0" in r.stdout and "rows_n2 rc=0" in r.stdout, r.stdout + r.stderr
    assert "N=2: dry run ok" in r.stdout and "wall is max over ranks: True" in r.stdout, r.stdout
