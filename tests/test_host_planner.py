"""CPU tests of the host side: the C-ABI library loads and exports every symbol of
include/genphi.h, the planner's levelisation equals the oracle's literal restatement of
src/compute.jl:236-262, the Python mirror of genealogy/pro/founder equals the oracle's, and
the error behaviour matches the reference (KeyError).  No compute calls: no GPU here."""
import ctypes
import io
import os
import re
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "genlib.jl_amd", "data")      # the two bundled pedigrees (data files of the reference's tests)
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol(gen):
    from genlib_jl_amd import _capi
    header = open(os.path.join(ROOT, "include", "genphi.h")).read()
    declared = set(re.findall(r"\b(genphi_[a-z0-9_]+)\s*\(", header))
    declared -= {"genphi_opts", "genphi_stats", "genphi_plan"}
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    L = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in _capi.lib().genphi_version()


def _load(gen, path):
    return gen.genealogy(path)


@pytest.mark.parametrize("name", ["geneaJi.csv", "genea140.csv"])
def test_genealogy_pro_founder_match_oracle(gen, oracle, name):
    path = os.path.join(DATA, name)
    ped = _load(gen, path)
    oped = oracle.Pedigree.from_file(path)
    assert np.array_equal(ped.ind, oped.ind)            # same rank order (stable depth sort)
    assert np.array_equal(ped.father, oped.father) and np.array_equal(ped.mother, oped.mother)
    assert np.array_equal(gen.pro(ped), oped.pro())
    assert np.array_equal(gen.founder(ped), oped.founder())


def _check_levels(gen, oracle, ind, fa, mo, sex, pro):
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    oped = oracle.Pedigree(ind, fa, mo)
    assert np.array_equal(ped.ind, oped.ind)
    pl = gen.plan(ped, pro)
    sizes, both = pl.levels()
    osizes, oboth, ocuts = oped.levels(pro)
    assert sizes == osizes and both == oboth
    assert pl.n_probands == len(ocuts[-1])
    assert pl.algorithmic_bytes == 4.0 * sum(a * a + b * b for a, b in zip(sizes[:-1], sizes[1:]))
    pl.close()


def test_genealogy_from_a_table_in_memory(gen, oracle):
    """gen.genealogy(dataframe; sort) (src/create.jl:131-146 + the ordering of :196-254) through genphi_genealogy_order: the rank order
    of the oracle for shuffled files (stable on depth ties), IDs far apart (hash map instead of the direct table), a pandas DataFrame,
    sort=false, and the reference's errors."""
    import pandas as pd
    from genlib_jl_amd import synth
    for seed in range(3):
        ind, fa, mo, sex, _ = synth.random_mating(3000, 200, 9, seed=seed, skip_permille=150)
        perm = np.random.default_rng(seed).permutation(len(ind))                      # children before their parents in the file
        for mul in (1, 1_000_003):
            cols = [a[perm] * (mul if k < 3 else 1) for k, a in enumerate((ind, fa, mo, sex))]
            ped = gen.genealogy(dict(zip(("ind", "father", "mother", "sex"), cols)))
            oped = oracle.Pedigree(cols[0], cols[1], cols[2])
            assert np.array_equal(ped.ind, oped.ind) and np.array_equal(ped.father, oped.father) and np.array_equal(ped.mother, oped.mother)
            assert np.array_equal(ped.sex, cols[3][np.argsort(cols[0])][np.searchsorted(np.sort(cols[0]), ped.ind)])
        i2, f2, m2, s2 = synth.parents_first_shuffle(ind, fa, mo, sex, seed=seed)
        ped = gen.genealogy({"ind": i2, "father": f2, "mother": m2, "sex": s2}, sort=False)
        assert np.array_equal(ped.ind, i2) and np.array_equal(ped.father, f2)          # file order kept
    df = pd.DataFrame({"ind": [3, 1, 2], "father": [1, 0, 0], "mother": [2, 0, 0], "sex": [1, 1, 2]})
    assert gen.genealogy(df).ind.tolist() == [1, 2, 3]
    assert len(gen.genealogy({"ind": [], "father": [], "mother": [], "sex": []}).ind) == 0
    with pytest.raises(ValueError):                                                  # duplicate ID (refused, not last-wins)
        gen.genealogy({"ind": [1, 1], "father": [0, 0], "mother": [0, 0], "sex": [1, 1]})
    with pytest.raises(KeyError):                                                    # a parent that is not an individual
        gen.genealogy({"ind": [1, 2], "father": [0, 7], "mother": [0, 0], "sex": [1, 1]})
    with pytest.raises(ValueError):                                                  # a cycle
        gen.genealogy({"ind": [1, 2], "father": [2, 1], "mother": [0, 0], "sex": [1, 1]})
    with pytest.raises(KeyError):                                                    # sort=false: parent after child (src/create.jl:240-241)
        gen.genealogy({"ind": [2, 1], "father": [1, 0], "mother": [0, 0], "sex": [1, 1]}, sort=False)
    with pytest.raises(ValueError):
        gen.genealogy({"ind": [1, 2], "father": [0], "mother": [0, 0], "sex": [1, 1]})


def test_pro_by_flag_table_and_by_sets_agree(gen):
    """gen.pro (src/identify.jl:35-39: IDs without children, ascending) takes a flag table when the IDs lie in a moderate range and the
    two-sort set form otherwise: both against the definition, on dense IDs, on IDs spread out beyond the table's range, and on an
    empty pedigree's slice."""
    from genlib_jl_amd import synth
    def by_definition(ped):
        parents = set(ped.father.tolist()) | set(ped.mother.tolist())
        return np.array(sorted(x for x in ped.ind.tolist() if x not in parents), dtype=np.int64)
    for seed in range(3):
        ind, fa, mo, sex, _ = synth.random_mating(2500, 150, 7, seed=seed, skip_permille=100)
        for mul in (1, 37, 1_000_003):                                       # 1_000_003: far beyond 64 IDs' worth of table per individual
            ped = gen.genealogy({"ind": ind * mul, "father": fa * mul, "mother": mo * mul, "sex": sex})
            assert np.array_equal(gen.pro(ped), by_definition(ped)), (seed, mul)
    ped = gen.genealogy(gen.genea140)
    assert np.array_equal(gen.pro(ped), by_definition(ped)) and len(gen.pro(ped)) == 140


def test_levels_match_oracle_bundled(gen, oracle):
    for name in ["geneaJi.csv", "genea140.csv"]:
        ind, fa, mo, sex = oracle.read_tsv(os.path.join(DATA, name))
        _check_levels(gen, oracle, ind, fa, mo, sex, None if True else None)
    ind, fa, mo, sex = oracle.read_tsv(os.path.join(DATA, "genea140.csv"))
    ped = gen.genealogy(os.path.join(DATA, "genea140.csv"))
    pro = gen.pro(ped)
    # explicit proband subsets, shuffled order, with duplicates and a non-leaf proband
    rng = np.random.default_rng(1)
    sub = rng.permutation(pro)[:17].tolist()
    sub = sub + [sub[0], int(ped.father[np.flatnonzero(ped.ind == sub[1])[0]])]
    _check_levels(gen, oracle, ind, fa, mo, sex, np.array(sub))


def test_levels_match_oracle_synthetic(gen, oracle):
    from genlib_jl_amd import synth
    for args, kw in [((3000, 300, 12), dict(skip_permille=50)), ((5000, 500, 8), dict(skip_permille=0)),
                     ((2000, 100, 25), dict(skip_permille=200, seed=7))]:
        ind, fa, mo, sex, pro = synth.random_mating(*args, **kw)
        _check_levels(gen, oracle, ind, fa, mo, sex, pro)
    ind, fa, mo, sex, pro = synth.deep_inbred(60, 20, 3)
    _check_levels(gen, oracle, ind, fa, mo, sex, pro)
    ind, fa, mo, sex, pro = synth.chain_two_lines(40)
    _check_levels(gen, oracle, ind, fa, mo, sex, pro)


def test_verbose_lines_and_compute_false(gen):
    ped = gen.genealogy(gen.geneaJi)
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert gen.phi(ped, compute=False) is None       # src/compute.jl:264-266
    lines = buf.getvalue().splitlines()
    assert lines[0] == "Step 1 of 7: 2 founders, 4 probands, 2 both."      # format of :257-260
    assert lines[-1] == "Step 7 of 7: 4 founders, 3 probands, 0 both."
    assert len(lines) == 7


def test_error_behaviour(gen):
    ped = gen.genealogy(gen.geneaJi)
    with pytest.raises(KeyError):                        # unknown proband: KeyError (create.jl:70)
        gen.plan(ped, [1, 12345])
    with pytest.raises(KeyError):                        # parent listed after child, sort=false
        gen.genealogy({"ind": [1, 2], "father": [2, 0], "mother": [0, 0], "sex": [1, 1]}, sort=False)
    from genlib_jl_amd import _capi
    with pytest.raises(KeyError):                        # same through the raw C-ABI
        _capi.PhiPlan([1, 2], [2, 0], [0, 0], [1])
    with pytest.raises(ValueError):
        _capi.PhiPlan([1, 1], [0, 0], [0, 0], [1])
    # duplicates collapse, order = first occurrence
    pl = gen.plan(ped, [29, 1, 29, 2, 1])
    assert pl.n_probands == 3
    pl.close()
    # no probands: empty plan
    pl = gen.plan(ped, [])
    assert pl.levels() == ([], []) and pl.n_probands == 0
    pl.close()


def test_no_gpu_means_loud_failure(gen):
    """The product path has no CPU fallback: without a device compute must raise
    GenphiDeviceError (on a GPU box the call succeeds and the test is skipped)."""
    ped = gen.genealogy(gen.geneaJi)
    try:
        out = gen.phi(ped)
    except gen.GenphiDeviceError as exc:
        assert "no CPU fallback" in str(exc) or "HIP" in str(exc) or "hip" in str(exc)
        return
    assert out.shape == (3, 3)
    pytest.skip("GPU present: gen.phi ran on it")


def test_synthetic_generator_is_deterministic(gen):
    from genlib_jl_amd import synth
    a = synth.random_mating(1000, 100, 10)
    b = synth.random_mating(1000, 100, 10)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # SplitMix64 known answers (seed 1234567: first outputs of the published generator)
    out = synth.splitmix64(1234567, np.arange(3))
    assert [int(v) for v in out] == [6457827717110365317, 3203168211198807973, 9817491932198370423]
    ind, fa, mo, sex, pro = a
    assert np.all(fa < ind) and np.all(mo < ind) and len(pro) == 100
    assert np.all(sex[fa[fa > 0] - 1] == 1) and np.all(sex[mo[mo > 0] - 1] == 2)


def test_native_loader_matches_oracle_rank_order(gen, oracle, tmp_path):
    """genphi_genealogy_read (csrc/loader.cpp) = gen.genealogy(filename; sort): TSV parse
    (src/create.jl:161-189) + stable depth sort (:196-227), against the oracle's restatement."""
    from genlib_jl_amd import synth, _capi
    for name in ("geneaJi.csv", "genea140.csv"):
        path = os.path.join(DATA, name)
        ind, fa, mo, sex = _capi.genealogy_read(path)
        op = oracle.Pedigree.from_file(path)
        assert np.array_equal(ind, op.ind) and np.array_equal(fa, op.father) and np.array_equal(mo, op.mother)
    # shuffled file order: the depth sort has real work, ties keep file order
    ind, fa, mo, sex, _ = synth.random_mating(50_000, 5_000, 12, skip_permille=80)
    perm = np.random.default_rng(5).permutation(len(ind))
    path = str(tmp_path / "shuffled.tsv")
    synth.write_tsv(path, ind[perm], fa[perm], mo[perm], sex[perm])
    got = _capi.genealogy_read(path)
    op = oracle.Pedigree(ind[perm], fa[perm], mo[perm])
    assert np.array_equal(got[0], op.ind) and np.array_equal(got[1], op.father) and np.array_equal(got[2], op.mother)
    assert np.array_equal(np.sort(got[0]), np.sort(ind)) and set(np.unique(got[3])) <= {1, 2}
    # sort=False keeps file order and requires parents first (KeyError otherwise, as _finalize_pedigree)
    with pytest.raises(KeyError):
        _capi.genealogy_read(path, sort=False)
    ordered = str(tmp_path / "ordered.tsv")
    synth.write_tsv(ordered, ind, fa, mo, sex)
    assert np.array_equal(_capi.genealogy_read(ordered, sort=False)[0], ind)
    # error behaviour
    bad = tmp_path / "bad.tsv"
    bad.write_text("ind\tfather\tmother\tsex\n1\t0\t0\t1\n2\t1\t9\t2\n")
    with pytest.raises(KeyError):
        _capi.genealogy_read(str(bad))                 # unknown mother 9
    bad.write_text("ind\tfather\tmother\tsex\n1\t0\t0\n")
    with pytest.raises(ValueError):
        _capi.genealogy_read(str(bad))                 # 3 columns
    with pytest.raises(ValueError):
        _capi.genealogy_read(str(tmp_path / "missing.tsv"))
    bad.write_text("ind\tfather\tmother\tsex\n1\t2\t0\t1\n2\t1\t0\t1\n")
    with pytest.raises(ValueError):
        _capi.genealogy_read(str(bad))                 # cycle


def test_branching_matches_reference_checks_and_oracle(gen, oracle):
    """gen.branching (src/extract.jl:65-186) through the C-ABI (genphi_branching): the
    reference's own three checks (test/runtests.jl:69-74), then array equality with the
    oracle's restatement on genea140 and on a synthetic pedigree, all three argument cases."""
    ped = gen.genealogy(gen.geneaJi)
    assert gen.founder(gen.branching(ped, pro=[1])).tolist() == [17, 19, 20, 25, 26]
    assert gen.pro(gen.branching(ped, ancestors=[13])).tolist() == [1, 2]
    assert gen.branching(ped, pro=[1], ancestors=[13]).ind.tolist() == [13, 8, 4, 1]
    assert len(gen.branching(ped)) == 0
    with pytest.raises(KeyError):
        gen.branching(ped, pro=[12345])
    with pytest.raises(KeyError):
        gen.branching(ped, ancestors=[12345])

    from genlib_jl_amd import synth
    ind, fa, mo, sex, pro = synth.random_mating(3000, 300, 8, skip_permille=60)
    cases = [(gen.genealogy(gen.genea140), oracle.Pedigree.from_file(gen.genea140)),
             (gen.Pedigree(ind, fa, mo, sex), oracle.Pedigree(ind, fa, mo, sort=False))]
    rng = np.random.default_rng(11)
    for gp, op in cases:
        pr = rng.choice(gen.pro(gp), size=7, replace=False)
        an = rng.choice(gen.founder(gp), size=5, replace=False)
        for kw in ({"pro": pr}, {"ancestors": an}, {"pro": pr, "ancestors": an}, {"pro": np.zeros(0, np.int64)}):
            got = gen.branching(gp, **kw)
            want = op.branching(**kw)
            assert np.array_equal(got.ind, want[0]) and np.array_equal(got.father, want[1]) \
                and np.array_equal(got.mother, want[2]), kw.keys()
            # sex travels with the individual
            assert np.array_equal(got.sex, gp.sex[gp.positions(got.ind)])
        # pruning to the probands' ancestors does not change the levelisation gen.phi does
        sub = gen.branching(gp, pro=pr)
        a, b = gen.plan(gp, np.sort(pr)), gen.plan(sub, np.sort(pr))
        assert a.levels() == b.levels()
        a.close(); b.close()


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "phi_c_abi")
    lib_dir = os.path.join(ROOT, "genlib.jl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "phi_c_abi.c"), "-o", exe, "-L", lib_dir, "-lgenphi",
                           "-Wl,-rpath," + lib_dir])
    return exe


def test_c_abi_from_plain_c(gen, tmp_path):
    """include/genphi.h is a real C header: a C11 program (examples/phi_c_abi.c) compiles against it
    with -Werror, links libgenphi.so and levelises geneaJi (no GPU needed for --plan)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe, gen.geneaJi, "--plan"], capture_output=True, text=True, check=True).stdout
    assert out.splitlines()[0] == "Step 1 of 7: 2 founders, 4 probands, 2 both."
    assert len(out.splitlines()) == 7


@pytest.mark.gpu
def test_c_abi_from_plain_c_on_gpu(gen, tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe, gen.geneaJi], capture_output=True, text=True, check=True).stdout.splitlines()
    rows = [[float(x) for x in line.split()] for line in out[7:10]]
    assert rows == [[0.591796875, 0.37109375, 0.072265625], [0.37109375, 0.591796875, 0.072265625],
                    [0.072265625, 0.072265625, 0.53515625]]                 # test/runtests.jl:50-52
    assert out[10] == "phiMean 0.171875"                                       # :53


def test_hub_walk_work_lists(monkeypatch):
    """The work lists of the SPLIT kernels (csrc/planner.h, build_hub_walk): a walk over the parent graph of a cut's
    rows in which a workgroup keeps the expansion of one "hub" row in registers, finishes the rows that have only that
    source from it, stages the OTHER parent's row of every child, and goes on with the row staged last as the next hub.
    Invariants the kernels rely on, and the point of it: fewer staged rows than one group per father."""
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    monkeypatch.setenv("GENPHI_FULL_MAX_FLOATS", "0")                       # every level SPLIT
    for args, kw in [((6000, 700, 7), dict(skip_permille=30)), ((3000, 300, 12), dict(skip_permille=150, seed=11)), ((40_000, 4000, 10), dict())]:
        ind, fa, mo, sex, pro = synth.random_mating(*args, **kw)
        if kw.get("seed") == 11:
            mo = mo.copy(); mo[::17] = 0                                    # one-parent members
        ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
        for max_run in ("32", None, "5"):                                   # (None: the default = 1, no chaining)
            if max_run is None:
                monkeypatch.delenv("GENPHI_MAX_RUN", raising=False)
            else:
                monkeypatch.setenv("GENPHI_MAX_RUN", max_run)
            pl = gen.plan(ped, pro)
            sizes, both = pl.levels()
            staged_walk = staged_groups = 0
            for step, mode in enumerate(pl.step_modes()):
                if mode != 1:
                    continue
                desc, seg, run = pl.step_walk(step)
                n_prev, n = sizes[step], sizes[step + 1]
                none = n_prev
                assert sorted(desc[:, 0].tolist()) == list(range(n))                      # every row of the cut exactly once
                assert np.array_equal(desc[:, 0], desc[:, 1])                            # (no shard: output row = storage row)
                n_segs, n_runs = len(seg) - 2, len(run) - 1
                assert seg[n_segs, 0] == n and run[n_runs, 0] == n_segs                  # terminators
                assert np.all(np.diff(seg[: n_segs + 1, 0]) >= 0) and np.all(np.diff(run[:, 0]) > 0)
                starts = set(run[:n_runs, 0].tolist())
                for g in range(n_segs):
                    wb, hub, n0, typ = (int(x) for x in seg[g])
                    we = int(seg[g + 1, 0])
                    rows = desc[wb:we]
                    assert np.all(rows[:n0, 2] == none) and np.all(rows[n0:, 2] != none)  # rows without a row to stage lead the segment
                    assert 0 <= len(rows) - n0 <= 8                                       # <= 8 children with one (<= 4 where the exact kernel keeps rank masks)
                    assert n0 <= 8                                                        # (all parentless members share the hub "none": no serial tail)
                    assert (typ == 0) == (g in starts)
                    if typ == 0:
                        rr = run[sorted(starts).index(g)]                                # the run's entry repeats its first segment
                        assert (rr[1] & 0xffff, rr[1] >> 16, rr[2], rr[3]) == (hub, n0, wb, we)
                        if g + 1 < n_segs and seg[g + 1, 3] == 2:                         # a hub's children beyond 8: a new run
                            assert seg[g + 1, 1] == hub
                    if typ == 1:                                                         # the hub is the row staged last by the previous segment
                        assert desc[wb - 1, 2] == hub and wb > 0 and seg[g - 1, 0] < wb
                    if typ == 2:
                        assert seg[g - 1, 1] == hub and n0 == 0
                    if typ != 0 and g + 1 < n_segs and len(rows) == n0:
                        assert (g + 1) in starts                                         # a segment that stages nothing ends its run
                staged_walk += n_runs + int(np.count_nonzero(desc[:, 2] != none))
                if max_run is None:
                    assert not np.any(seg[:n_segs, 3] == 1)                              # no chain steps
            pl.close()
            if max_run == "32":
                full = staged_walk
            if max_run is None:
                assert full < 0.9 * staged_walk                                          # chaining stages fewer rows than one run per hub
    monkeypatch.delenv("GENPHI_FULL_MAX_FLOATS", raising=False)
    monkeypatch.delenv("GENPHI_MAX_RUN", raising=False)


def test_wide_runs_planned_in_place(monkeypatch):
    """Persistent slots (csrc/planner.h LevelStep::stay), planner side, no GPU: on overlapping generations whose cuts exceed a small
    LDS budget the planner keeps runs of WIDE steps in place; a step that writes in place reads by slot and so does the step after
    it; the slot capacity holds the cut and the new members; the cut sizes and dragged counts do not depend on any of it (they are
    the oracle-checked ones); GENPHI_NO_STAY and a tiny GENPHI_STAY_MAX_SLOTS switch it off; more head room makes longer runs."""
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    ind, fa, mo, sex, pro = synth.random_mating(30000, 400, 30, skip_permille=600, seed=3)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    monkeypatch.setenv("GENPHI_LDS_CAP_FLOATS", "2000")
    monkeypatch.setenv("GENPHI_STAY_MEM_PCT", "1000")

    def plan_of(**env):
        for k in ("GENPHI_NO_STAY", "GENPHI_STAY_MAX_SLOTS", "GENPHI_STAY_HEADROOM"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = gen.plan(ped, pro)
        sizes, both = pl.levels()
        modes = pl.step_modes()
        slots = [pl.step_slots(k) for k in range(len(modes))]
        pl.close()
        return sizes, both, modes, slots

    sizes, both, modes, slots = plan_of()
    stay = [k for k, f in enumerate(slots) if f[0] & 1]
    assert len(stay) >= 8 and all(modes[k] == 2 for k in stay)
    for k in stay:
        flags, P, p0, npad = slots[k]
        assert flags & 2 and (k + 1 == len(slots) or (slots[k + 1][0] & 2 and modes[k + 1] == 2))      # reads by slot; so does the next step, a WIDE one (if any)
        assert P % 64 == 0 and p0 % 64 == 0 and npad % 64 == 0 and p0 < P
        assert npad >= sizes[k + 1] - both[k] and P >= sizes[k] + npad           # room for the source cut and the new members
        assert sizes[k + 1] >= 2 * (sizes[k + 1] - both[k])                      # worth it: at least as many dragged as new members
    for k, f in enumerate(slots):
        if f[0] == 2:                                                            # the step that leaves a run: reads by slot, writes compactly
            assert slots[k - 1][0] & 1 and slots[k - 1][1] == f[1]
    for env in ({"GENPHI_NO_STAY": "1"}, {"GENPHI_STAY_MAX_SLOTS": "128"}):
        s2, b2, m2, sl2 = plan_of(**env)
        assert (s2, b2, m2) == (sizes, both, modes) and all(f == (0, 0, 0, 0) for f in sl2)
    s3, b3, m3, sl3 = plan_of(GENPHI_STAY_HEADROOM="3")
    assert (s3, b3, m3) == (sizes, both, modes) and sum(f[0] & 1 for f in sl3) >= len(stay)
    # the memory guard: runs whose slot matrices need more than the given share of the plain level buffers are dropped
    s4, b4, m4, sl4 = plan_of(GENPHI_STAY_HEADROOM="3", GENPHI_STAY_MEM_PCT="101")
    assert (s4, b4, m4) == (sizes, both, modes) and all(f == (0, 0, 0, 0) for f in sl4)


def test_narrow_runs_cost_model_and_proband_cut_in_place(monkeypatch):
    """Round 4, planner side, no GPU.  (i) SURVEY.md 8(d)'s cfg3 (5 % of the parents from g-2): runs of SPLIT-width cuts stay in place
    by the cost model -- 11 steps with the fixed cost per block-assembled step, 14 on bytes alone, none with GENPHI_STAY_NARROW=0 -- and
    cut sizes / dragged counts do not depend on any of it.  (ii) a real genealogy with every individual a proband: the run goes THROUGH
    the proband cut (the last step writes in place: flags & 1, nothing reads it), unless GENPHI_STAY_LAST=0 -- then the last step reads
    by slot and compacts.  (iii) genea140 with its 140 probands keeps its row kernels."""
    import numpy as np
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    knobs = ("GENPHI_STAY_NARROW", "GENPHI_STAY_OVERHEAD_K", "GENPHI_STAY_LAST", "GENPHI_NO_STAY")

    def plan_of(ped, pro, **env):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = gen.plan(ped, pro)
        sizes, both = pl.levels()
        modes = pl.step_modes()
        slots = [pl.step_slots(k) for k in range(len(modes))]
        pl.close()
        return sizes, both, modes, slots

    ind, fa, mo, sex, pro = synth.random_mating(100_000, 10_000, 20, skip_permille=50)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    sizes, both, modes, slots = plan_of(ped, pro)
    stay = [k for k, f in enumerate(slots) if f[0] & 1]
    assert max(sizes) == 20_540 and len(stay) == 11 and all(modes[k] == 2 and sizes[k] <= 36_863 for k in stay)
    assert slots[stay[-1] + 1][0] == 2 and modes[stay[-1] + 1] == 2 and modes[-1] == 0      # the compacting step behind the run; the proband step FULL
    assert all(sizes[k + 1] >= 2 * (sizes[k + 1] - both[k]) for k in stay)
    s2, b2, m2, sl2 = plan_of(ped, pro, GENPHI_STAY_OVERHEAD_K="0")
    assert (s2, b2) == (sizes, both) and sum(f[0] & 1 for f in sl2) == 14
    s3, b3, m3, sl3 = plan_of(ped, pro, GENPHI_STAY_NARROW="0")
    assert (s3, b3) == (sizes, both) and 2 not in m3 and all(f == (0, 0, 0, 0) for f in sl3)

    ped = gen.genealogy(gen.genea140)
    ids = np.asarray(ped.ind, dtype=np.int64)
    sizes, both, modes, slots = plan_of(ped, ids)
    assert sizes[-1] == 41_523 and slots[-1][0] & 1 and modes[-1] == 2                      # the proband step writes in place
    assert sum(f[0] & 1 for f in slots) == 11 and slots[-1][1] >= sizes[-1]                 # slot capacity holds every proband
    s4, b4, m4, sl4 = plan_of(ped, ids, GENPHI_STAY_LAST="0")
    assert (s4, b4) == (sizes, both) and sl4[-1][0] == 2 and sum(f[0] & 1 for f in sl4) == 10   # ... or reads by slot and compacts
    s5, b5, m5, sl5 = plan_of(ped, gen.pro(ped))
    assert 2 not in m5                                                                      # 140 probands: 36-57 % of a cut is new
    for k in knobs:
        monkeypatch.delenv(k, raising=False)


def test_tuning_through_the_abi_and_the_environment_gate():
    """Settings reach a plan through a genphi_tuning (genphi_plan_create_tuned), or through GENPHI_* environment variables -- which the
    library reads only under GENPHI_ENV_HOOKS=1: ambient variables in somebody's process change nothing."""
    import subprocess
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    ind, fa, mo, sex, pro = synth.random_mating(3000, 300, 8, skip_permille=100)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    pl = gen.plan(ped, pro, tuning={})
    default_modes = pl.step_modes()
    pl.close()
    assert 2 not in default_modes
    pl = gen.plan(ped, pro, tuning={"LDS_CAP_FLOATS": 64, "GENPHI_NO_STAY": 1})      # (with or without the prefix)
    assert 2 in pl.step_modes()                                      # rows no longer fit the LDS budget: block assembly
    pl.close()
    with pytest.raises(Exception):
        gen.plan(ped, pro, tuning={"NO_SUCH_KNOB": 1})
    code = ("import sys; sys.path.insert(0, %r); import genlib_jl_amd as gen; from genlib_jl_amd import synth; "
            "ind, fa, mo, sex, pro = synth.random_mating(3000, 300, 8, skip_permille=100); "
            "ped = gen.genealogy({'ind': ind, 'father': fa, 'mother': mo, 'sex': sex}); print(2 in gen.plan(ped, pro).step_modes())" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith("GENPHI_")}
    env["GENPHI_LDS_CAP_FLOATS"] = "64"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == "False", out.stdout + out.stderr      # the variable alone: ignored
    env["GENPHI_ENV_HOOKS"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == "True", out.stdout + out.stderr
