"""Generates tests/golden/cfg4_rowsample.npz: rows of the 1e5 x 1e5 proband matrix of the bench's headline workload (cfg4: 1e6
individuals / 1e5 probands / 30 generations, SplitMix64 seed 20241016) computed by the ORACLE -- every upper level step of
src/compute.jl:291-299 restated in full on the host (8.3e9 pair evaluations), the last step for the sampled rows -- so that the GPU
test compares against committed data instead of re-running the oracle on every lease (67 s of the GPU suite in round 4).

Stored per sampled row: its SHA-256 over the Float32 bytes, its Float64 sum, and Float64 sums of its 4,096-column blocks (to say
WHERE a row differs); the rows themselves would be 63 MB.  Which rows: the first and last 48 of the last step's work queue (host
planner), 59 spread over the proband order, three consecutive ones (also fetched as a row shard by the test).

    python tests/golden/make_rowsamples.py          (CPU only; ~3 min on 8 cores, 8 GB of memory)
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

BLOCK = 4096


def sample_rows(gen, ped, pro):
    pl = gen.plan(ped, pro)
    n = pl.n_probands
    modes = pl.step_modes()
    desc, seg, run = pl.step_walk(len(modes) - 1)
    order = desc[:, 1].astype(np.int64)
    pl.close()
    return np.unique(np.concatenate([order[:48], order[-48:], np.linspace(0, n - 1, 59).astype(np.int64), [50_000, 50_001, 50_002]]))


def digest(rows_f32):
    sha = np.array([hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest() for r in rows_f32])
    sums = rows_f32.astype(np.float64).sum(axis=1)
    nb = (rows_f32.shape[1] + BLOCK - 1) // BLOCK
    blocks = np.stack([rows_f32[:, b * BLOCK:(b + 1) * BLOCK].astype(np.float64).sum(axis=1) for b in range(nb)], axis=1)
    return sha, sums, blocks


def main():
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    from oracle import oracle as O
    O.fit_threads_to_quota()
    ind, fa, mo, sex, pro = synth.random_mating(1_000_000, 100_000, 30)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    rows = sample_rows(gen, ped, pro)
    t0 = time.time()
    want = O.Pedigree(ind, fa, mo).phi_rows(pro, rows)
    print(f"oracle: {len(rows)} rows of the {len(pro)} x {len(pro)} matrix in {time.time() - t0:.0f} s", flush=True)
    sha, sums, blocks = digest(want)
    np.savez_compressed(os.path.join(HERE, "cfg4_rowsample.npz"), rows=rows, sha256=sha, row_sum=sums, block_sum=blocks, block=np.int64(BLOCK),
                        head=want[:, :64].copy(), workload=np.array("synth.random_mating(1_000_000, 100_000, 30), seed 20241016"))
    print("wrote", os.path.join(HERE, "cfg4_rowsample.npz"))


if __name__ == "__main__":
    main()
