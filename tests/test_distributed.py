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
