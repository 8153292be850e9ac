"""Random stress of the GPU path against the oracle (not part of the pytest run; run on a GPU box:
    python tests/stress_random.py [seconds] [seed]
Random pedigrees x kernel families (LDS budget forces FULL / SPLIT / WIDE) x certificate thresholds
(mixed fast / grouping-exact SPLIT levels) x workgroup variants x proband subsets x row shards x the
Float64 sweep x sparse_phi x genealogy(sort = true | false), every result compared with the oracle bit for bit.  Prints one line per
failure with everything needed to reproduce it, and a summary."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GENPHI_ENV_HOOKS"] = "1"      # the knobs below are environment hooks: read by the library only under this gate

KNOBS = ["GENPHI_LDS_CAP_FLOATS", "GENPHI_FULL_MAX_FLOATS", "GENPHI_CERT_MIN_EXP", "GENPHI_FAST_NT", "GENPHI_NO_FAST",
         "GENPHI_MAX_CPT", "GENPHI_NO_SMALL", "GENPHI_NO_SHARD_PRUNE", "GENPHI_MAX_GROUP", "GENPHI_WIDE_ROUTE", "GENPHI_MAX_RUN",
         "GENPHI_NO_STAY", "GENPHI_STAY_HEADROOM", "GENPHI_STAY_MEM_PCT", "GENPHI_STAY_SCATTER", "GENPHI_STAY_TWO_PASS",
         "GENPHI_STAY_NARROW", "GENPHI_STAY_NARROW_MIN", "GENPHI_STAY_MIN_RATIO_PCT", "GENPHI_STAY_SCALAR_T", "GENPHI_STAY_OVERHEAD_K", "GENPHI_STAY_LAST", "GENPHI_COLPERM_PLAIN", "GENPHI_STAY_TILE", "GENPHI_SPARSE_NO_FUSED",
         "GENPHI_SPARSE_K", "GENPHI_SPARSE_MIN_CUT", "GENPHI_SPARSE_CLASSES", "GENPHI_SPARSE_CHUNK", "GENPHI_SPARSE_ARENA"]


def make_case(case):
    """The pedigree, proband list and environment knobs of one random case (a pure function of `case`); `r` goes on drawing."""
    from genlib_jl_amd import synth
    r = np.random.default_rng(case)
    n_gen = int(r.integers(3, 16))
    n_pro = int(r.integers(5, 400))
    n_ind = n_pro + (n_gen - 1) * int(r.integers(20, 500))
    skip = int(r.choice([0, 0, 30, 150, 400, 650, 850]))
    ind, fa, mo, sex, pro = synth.random_mating(n_ind, n_pro, n_gen, seed=case, skip_permille=skip)
    if r.random() < 0.3:
        mo = mo.copy(); mo[:: int(r.integers(7, 40))] = 0                   # one-parent members
    if r.random() < 0.4:                                                  # probands: a subset, some ancestors, duplicates
        extra = ind[r.integers(0, len(ind), size=int(r.integers(1, 20)))]
        pro = np.concatenate([r.permutation(pro)[: max(2, n_pro // 2)], extra, pro[:2]])
    # (round 4) probands at every depth: a random share of ALL individuals, or every one of them -- nobody then leaves the cuts, runs of
    # in-place steps reach the proband cut and the result is delivered from the slot matrix
    env = {}
    if r.random() < 0.7:
        env["GENPHI_LDS_CAP_FLOATS"] = str(int(r.choice([64, 256, 700, 1500, 4096])))
    if r.random() < 0.3:
        env["GENPHI_FULL_MAX_FLOATS"] = "0"
    if r.random() < 0.5:
        env["GENPHI_CERT_MIN_EXP"] = str(int(r.integers(-27, 0)))
    if r.random() < 0.3:
        env["GENPHI_FAST_NT"] = str(int(r.choice([512, 1024])))
    if r.random() < 0.15:
        env["GENPHI_NO_FAST"] = "1"
    if r.random() < 0.3:
        env["GENPHI_MAX_CPT"] = str(int(r.choice([4, 8, 16])))
    if r.random() < 0.2:
        env["GENPHI_NO_SMALL"] = "1"
    if r.random() < 0.2:
        env["GENPHI_MAX_GROUP"] = str(int(r.choice([1, 3, 8])))
    if r.random() < 0.5:
        env["GENPHI_WIDE_ROUTE"] = str(r.choice(["A", "B"]))
    if r.random() < 0.5:
        env["GENPHI_MAX_RUN"] = str(int(r.choice([2, 7, 32, 1000])))             # hub walk with chain steps
    if r.random() < 0.15:
        env["GENPHI_NO_STAY"] = "1"                                               # WIDE levels never stay in place
    elif r.random() < 0.4:
        env["GENPHI_STAY_HEADROOM"] = str(int(r.choice([1, 2, 4])))               # longer in-place runs
    if r.random() < 0.3:
        env["GENPHI_STAY_TWO_PASS"] = "1"                                         # new x dragged and its transpose as two kernels
    if r.random() < 0.3:
        env["GENPHI_STAY_SCATTER"] = "1"                                          # the new x new block always through the compact buffer
    if r.random() < 0.8:
        env["GENPHI_STAY_MEM_PCT"] = "100000"                                     # (small cuts: the slot matrix is many times the plain buffers)
    # in-place runs at FULL / SPLIT widths (round 4): the cost model decides per run; small pedigrees need the width floor lowered
    if r.random() < 0.65:
        env["GENPHI_STAY_NARROW_MIN"] = str(int(r.choice([0, 16, 64, 200])))
        env["GENPHI_STAY_OVERHEAD_K"] = "0"                                       # (the cost model on bytes alone: tiny cuts would never pay a step's launches)
    elif r.random() < 0.3:
        env["GENPHI_STAY_NARROW"] = "0"
    if r.random() < 0.3:
        env["GENPHI_STAY_MIN_RATIO_PCT"] = str(int(r.choice([110, 150, 300])))       # in place with few dragged members too / only with many
    if r.random() < 0.2:
        pro = ind.copy() if r.random() < 0.5 else r.permutation(ind)[: max(3, len(ind) // int(r.integers(2, 6)))]
    if r.random() < 0.3:
        env["GENPHI_STAY_TILE"] = str(int(r.choice([128, 256])))                   # tile width of the fused in-place kernel
    if r.random() < 0.2:
        env["GENPHI_COLPERM_PLAIN"] = "1"                                         # the proband-order pass by the one-workgroup-per-row kernel
    if r.random() < 0.2:
        env["GENPHI_STAY_LAST"] = "0"                                             # the proband cut never stays in place
    if r.random() < 0.25:
        env["GENPHI_STAY_SCALAR_T"] = "1"                                         # the fused in-place kernel's transposed tile by 4-byte stores
    if r.random() < 0.3:
        env["GENPHI_SPARSE_NO_FUSED"] = "1"                                       # sparse_phi: a wave as rows + new x new kernels with T in HBM
    # (round 5) the leading cuts as lists of their non-zero entries (csrc/sparse_levels.hip): forced on small pedigrees, any last sparse
    # cut, the launch forms of a list step, small chunks of the sparse -> dense step.  Drawn from a generator of their own: the cases of
    # earlier rounds keep their pedigrees and knobs.
    r5 = np.random.default_rng([case, 5])
    if r5.random() < 0.7:
        env["GENPHI_SPARSE_MIN_CUT"] = "0"
        if r5.random() < 0.7:
            env["GENPHI_SPARSE_K"] = str(int(r5.choice([1, 2, 3, 5, 8, 11])))
        if r5.random() < 0.5:
            env["GENPHI_SPARSE_CLASSES"] = str(int(r5.choice([0, 1])))
        if r5.random() < 0.4:
            env["GENPHI_SPARSE_CHUNK"] = str(int(r5.choice([1024, 2048])))
        if r5.random() < 0.5:                                       # (small arenas: the calibration run enlarges them cut by cut)
            env["GENPHI_SPARSE_ARENA"] = str(int(r5.choice([64, 200, 1000, 20000])))
    elif r5.random() < 0.3:
        env["GENPHI_SPARSE_K"] = "-1"
    return r, n_gen, n_ind, n_pro, skip, ind, fa, mo, sex, pro, env


def run_case(case, gen, synth, O):
    """One case end to end: every product path it draws against the oracle, bit for bit.  Returns (failures, env, (n_gen, n_ind, n_pro,
    skip), in-place steps, sparse cuts)."""
    n_stay = n_sparse = 0
    r, n_gen, n_ind, n_pro, skip, ind, fa, mo, sex, pro, env = make_case(case)
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sort = bool(r.random() < 0.5)                                         # sort=false: the file order is the rank
    if not sort:
        ind, fa, mo, sex = synth.parents_first_shuffle(ind, fa, mo, sex, seed=case & 0xffff)
    env["sort"] = str(sort)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex}, sort=sort)
    oped = O.Pedigree(ind, fa, mo, sort=sort)
    want = oped.phi(pro)
    n = len(want)
    what = []
    try:
        pl = gen.plan(ped, pro)
        n_stay += sum(pl.step_slots(k)[0] & 1 for k in range(len(pl.step_modes())))
        if not np.array_equal(pl.compute(), want):
            what.append("full")
        n_sparse += pl.sparse_levels()[0] + 1
        if pl.sparse_levels()[0] >= 1 and not np.array_equal(pl.compute(no_sparse=True), want):
            what.append("same-plan-dense")
        if n > 2:
            a = int(r.integers(0, n - 1)); b = int(r.integers(a + 1, n + 1))
            if not np.array_equal(pl.compute(rows=(a, b)), want[a:b]):
                what.append(f"shard({a},{b})")
            if not np.array_equal(pl.compute(), want):
                what.append("full-after-shard")
        if r.random() < 0.2 and not np.array_equal(pl.compute(kernel=1), want):
            what.append("naive")
        pl.close()
        if r.random() < 0.25:
            ids = r.choice(ind, size=min(30, len(ind)), replace=False)
            if not np.array_equal(gen.f(ped, ids), oped.f(ids)):
                what.append("f")
        if r.random() < 0.25 and n_ind < 3000:
            sub = list(dict.fromkeys(int(x) for x in pro))[:40]
            K = gen.sparse_phi(ped, sub)
            Ko = O.SparsePhi(oped, sub)
            got = K.get(np.repeat(sub, len(sub)), np.tile(sub, len(sub))).reshape(len(sub), len(sub)).astype(np.float32)
            ge, we = K.entries(), Ko.entries()
            same_entries = sorted(zip(ge[0].tolist(), ge[1].tolist(), ge[2].tolist())) == sorted(zip(we[0].tolist(), we[1].tolist(), we[2].tolist()))
            if not np.array_equal(got, Ko.matrix()) or repr(K) != Ko.show() or not same_entries or gen.phiMean(K) != Ko.phi_mean():
                what.append("sparse")
    except Exception as e:          # noqa: BLE001
        what.append(f"exception {type(e).__name__}: {e}")
    return what, env, (n_gen, n_ind, n_pro, skip), n_stay, n_sparse


def main():
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    from oracle import oracle as O
    O.fit_threads_to_quota()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = np.random.default_rng(seed0)
    t0, n_cases, n_fail, n_stay, n_sparse = time.time(), 0, 0, 0, 0
    while time.time() - t0 < budget:
        case = int(rng.integers(1 << 30))
        what, env, shape, stay, sparse = run_case(case, gen, synth, O)
        n_gen, n_ind, n_pro, skip = shape
        n_stay += stay; n_sparse += sparse
        n_cases += 1
        if what:
            n_fail += 1
            print(f"FAIL case={case} gens={n_gen} n_ind={n_ind} n_pro={n_pro} skip={skip} env={env} -> {what}", flush=True)
        if n_cases % 20 == 0:
            print(f"... {n_cases} cases, {n_fail} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"stress: {n_cases} cases ({n_stay} in-place steps and {n_sparse} sparse cuts among them), {n_fail} failures in {time.time() - t0:.0f} s (seed {seed0})", flush=True)
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
