"""The host side of gen.sparse_phi without a GPU: the order in which individuals leave the reference's queue and the moment each is
dropped from the live set (src/compute.jl:336-345, :397-439), as the product schedules them (genphi_sparse_schedule: a countdown of
known parents per child instead of a queue, retirement = the last child's processing index) against the literal restatement
(oracle/sparse_oracle.cpp: a deque, children_to_process counted down)."""
import numpy as np
import pytest


def _check(gen, oracle, ind, fa, mo, sex, pro, sort=True):
    from genlib_jl_amd import _capi
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex}, sort=sort)
    order, retire_at, wave = _capi.sparse_schedule(ped.ind, ped.father, ped.mother, pro)
    want = oracle.SparsePhi(oracle.Pedigree(ind, fa, mo, sort=sort), pro)
    assert np.array_equal(order, want.order())
    retired = want.retired()
    got = {int(i): int(r) for i, r in zip(order, retire_at) if r >= 0}
    assert got == retired
    pro_set = set(int(x) for x in pro)
    assert all((int(i) in pro_set) == (r < 0) for i, r in zip(order, retire_at))      # exactly the probands are never dropped
    # waves: contiguous in processing order, one per depth, every parent in an earlier wave
    assert np.all(np.diff(wave) >= 0) and wave[0] == 0 and set(np.diff(wave).tolist()) <= {0, 1}
    wave_of = dict(zip(order.tolist(), wave.tolist()))
    at = dict(zip(ped.ind.tolist(), range(len(ped.ind))))
    for i in order.tolist():
        for par in (int(ped.father[at[i]]), int(ped.mother[at[i]])):
            if par:
                assert wave_of[par] < wave_of[i]
    return order


def test_sparse_schedule_matches_the_reference_queue(gen, oracle):
    from genlib_jl_amd import synth
    ped = gen.genealogy(gen.geneaJi)
    g = oracle.read_tsv(gen.geneaJi)
    order = _check(gen, oracle, *g, gen.pro(ped))
    assert len(order) == 29 and order[:6].tolist() == [17, 19, 20, 23, 25, 26]        # the founders by ascending ID first (:336-339)
    for seed, (args, kw) in enumerate([((600, 60, 6), dict(skip_permille=100)), ((2000, 150, 8), dict(skip_permille=0)),
                                       ((1500, 100, 12), dict(skip_permille=200, seed=9)), ((900, 80, 7), dict(skip_permille=50, seed=3))]):
        ind, fa, mo, sex, pro = synth.random_mating(*args, **kw)
        if seed == 3:
            mo = mo.copy(); mo[::13] = 0                                              # one-parent individuals
        _check(gen, oracle, ind, fa, mo, sex, pro)
        i2, f2, m2, s2 = synth.parents_first_shuffle(ind, fa, mo, sex, seed=seed)     # rank = file position, depths interleaved
        extra = i2[np.random.default_rng(seed).integers(0, len(i2), 12)]              # ancestors among the probands
        for p in (pro, np.concatenate([pro[::2], extra, pro[:3]])):                   # (duplicates in the list too)
            _check(gen, oracle, i2, f2, m2, s2, p, sort=False)
            _check(gen, oracle, i2, f2, m2, s2, p, sort=True)
    g = oracle.read_tsv(gen.genea140)
    ped = gen.genealogy(gen.genea140)
    _check(gen, oracle, *g, gen.pro(ped)[:25])
    _check(gen, oracle, *g, gen.pro(ped)[100:140], sort=False)


def test_sparse_schedule_errors(gen):
    from genlib_jl_amd import _capi
    ped = gen.genealogy(gen.geneaJi)
    with pytest.raises(KeyError):
        _capi.sparse_schedule(ped.ind, ped.father, ped.mother, [999])                 # unknown proband: KeyError in the reference
    o, r, w = _capi.sparse_schedule(ped.ind, ped.father, ped.mother, np.zeros(0, np.int64))
    assert len(o) == 0
