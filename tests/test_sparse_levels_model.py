"""The arithmetic of the zero-aware leading levels (genlib.jl_amd/csrc/sparse_levels.hip) restated in Python and checked
against the oracle on the CPU -- no GPU here; the kernels themselves are checked bit for bit in tests/test_gpu_parity.py.

What is verified: (i) a level step evaluated as a sparse product A Psi A^T over row lists, in INTEGER units of 2^-(2c+1) for
cut c, with the diagonal rule of src/compute.jl:148-154, gives exactly the Float32 matrix of the reference's level loop
(src/compute.jl:291-301 through the oracle) for pedigrees of at most 12 cuts; (ii) the bound the kernels rely on -- every entry
of cut c is an integer below 2^(2c+1) in those units -- holds on every cut of every case.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "genlib.jl_amd", "data")


def sparse_sweep(ped, pro):
    """Cuts as the oracle derives them (src/compute.jl:236-251); every level as {row: {column: integer units}}."""
    sizes, both, cuts = ped.levels(pro)
    fa = {int(i): int(f) for i, f in zip(ped.ind, ped.father)}
    mo = {int(i): int(m) for i, m in zip(ped.ind, ped.mother)}
    prev_pos = {int(x): k for k, x in enumerate(cuts[0])}
    rows = {k: {k: 1} for k in range(len(cuts[0]))}              # Psi_0 = 1/2 I in units of 2^-1
    for c in range(len(cuts) - 1):
        cut = [int(x) for x in cuts[c + 1]]
        src, new = [], []
        for x in cut:
            if x in prev_pos:
                src.append([prev_pos[x]]); new.append(False)
            else:
                src.append([prev_pos[p] for p in (fa[x], mo[x]) if p]); new.append(True)
        children = {}
        for i, ps in enumerate(src):
            for p in ps:
                children.setdefault(p, []).append((i, 1 if new[i] else 2))
        out = {}
        bound = 1 << (2 * (c + 1) + 1)
        for i, ps in enumerate(src):
            acc = {}
            wi = 1 if new[i] else 2
            for p in ps:
                for q, m in rows[p].items():
                    for j, wj in children.get(q, ()):
                        if new[i] and j == i:
                            continue
                        acc[j] = acc.get(j, 0) + m * wi * wj
            if new[i]:
                fm = rows[src[i][0]].get(src[i][1], 0) if len(src[i]) == 2 else 0
                acc[i] = (1 << (2 * c + 2)) + 2 * fm            # 1/2 + Psi[f][m]/2 only when both parents exist
            assert all(0 < v < bound for v in acc.values()), "an entry left [0, 1) or the integer grid of its cut"
            out[i] = acc
        rows = out
        prev_pos = {x: k for k, x in enumerate(cut)}
    n = len(cuts[-1])
    L = len(cuts) - 1
    phi = np.zeros((n, n), dtype=np.float32)
    for i, r in rows.items():
        for j, m in r.items():
            phi[i, j] = np.float32(m) * np.float32(2.0 ** -(2 * L + 1))
    return phi, [sum(len(r) for r in rows.values())]


def random_pedigree(rng, n_gen, per_gen, p_one_parent=0.1, p_skip=0.2, p_founder=0.05):
    ind, fa, mo = [], [], []
    gens = []
    nxt = 1
    for g in range(n_gen):
        ids = list(range(nxt, nxt + per_gen)); nxt += per_gen
        for x in ids:
            f = m = 0
            if g > 0 and rng.random() >= p_founder:
                gf = g - 2 if (g >= 2 and rng.random() < p_skip) else g - 1
                gm = g - 2 if (g >= 2 and rng.random() < p_skip) else g - 1
                f = int(rng.choice(gens[gf][0::2])); m = int(rng.choice(gens[gm][1::2]))
                u = rng.random()
                if u < p_one_parent / 2:
                    f = 0
                elif u < p_one_parent:
                    m = 0
            ind.append(x); fa.append(f); mo.append(m)
        gens.append(ids)
    return np.asarray(ind), np.asarray(fa), np.asarray(mo), np.asarray(gens[-1])


def test_geneaJi_sparse_product_is_the_reference_matrix():
    ped = O.Pedigree.from_file(os.path.join(DATA, "geneaJi.csv"))
    phi, _ = sparse_sweep(ped, ped.pro())
    ref = np.array([[0.591796875, 0.37109375, 0.072265625], [0.37109375, 0.591796875, 0.072265625],
                    [0.072265625, 0.072265625, 0.53515625]], dtype=np.float32)     # test/runtests.jl:50-52
    assert np.array_equal(phi, ref)
    assert np.array_equal(phi, ped.phi())


@pytest.mark.parametrize("seed", range(6))
def test_random_pedigrees_up_to_twelve_cuts(seed):
    rng = np.random.default_rng(seed)
    n_gen = int(rng.integers(3, 11))
    ind, fa, mo, last = random_pedigree(rng, n_gen, int(rng.integers(6, 40)))
    ped = O.Pedigree(ind, fa, mo)
    # probands at mixed depths: some ancestors of other probands (dragged through the cuts)
    pro = np.unique(np.concatenate([last, rng.choice(ind, size=min(5, len(ind)), replace=False)]))
    sizes, _, _ = ped.levels(pro)
    assert len(sizes) <= 12
    phi, _ = sparse_sweep(ped, pro)
    assert np.array_equal(phi, ped.phi(pro))


def test_genea140_leading_cuts():
    """The first cuts of genea140 (ancestors of the 140 probands eight generations up as the proband list): the sparse product
    equals the oracle's matrix of that cut, and the matrix is nearly empty."""
    ped = O.Pedigree.from_file(os.path.join(DATA, "genea140.csv"))
    sizes, both, cuts = ped.levels(ped.pro())
    members = cuts[4]                                            # cut 4: 4,357 members
    sub_sizes, _, sub_cuts = ped.levels(members)
    assert len(sub_sizes) <= 12
    phi, _ = sparse_sweep(ped, members)
    ref = ped.phi(members)
    assert np.array_equal(phi, ref)
    assert np.count_nonzero(ref) < 0.02 * ref.size
