"""One-off extended stress (not collected by pytest): random pedigrees x kernel modes x shards x
proband subsets against the oracle.  usage (GPU box): python tests/stress_random.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import genlib_jl_amd as gen
from oracle import oracle
from test_gpu_parity import _random_pedigree

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, cases, fails = time.time(), 0, 0
while time.time() - t0 < budget:
    n = int(rng.choice([30, 120, 400, 900, 1500, 1500, 4000]))
    pf, p1, ps = rng.choice([0.01, 0.05, 0.3]), rng.choice([0.0, 0.1, 0.3]), rng.choice([0.0, 0.05])
    back = int(rng.choice([5, 50, 400, n]))
    ind, fa, mo, sex = _random_pedigree(rng, n, pf, p1, ps, back)
    oped = oracle.Pedigree(ind, fa, mo)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    cap = rng.choice([0, 0, 4096, 900, 150])
    os.environ.pop("GENPHI_LDS_CAP_FLOATS", None)
    if cap:
        os.environ["GENPHI_LDS_CAP_FLOATS"] = str(cap)
    os.environ.pop("GENPHI_FULL_MAX_FLOATS", None)
    if rng.random() < 0.3:
        os.environ["GENPHI_FULL_MAX_FLOATS"] = "64"          # SPLIT kernel on small rows
    pro = oped.pro() if rng.random() < 0.5 else rng.choice(ind, size=int(rng.integers(1, min(n, 300) + 1)), replace=True)
    want = oped.phi(pro)
    pl = gen.plan(ped, pro)
    N = pl.n_probands
    got = pl.compute()
    ok = np.array_equal(got, want)
    if N >= 3:
        a, b = sorted(int(x) for x in rng.choice(np.arange(1, N), size=2, replace=False))
        parts = np.concatenate([pl.compute(rows=r) for r in ((0, a), (a, b), (b, N))], axis=0)
        ok = ok and np.array_equal(parts, want)
    modes = sorted(set(pl.step_modes()))
    # the same plan again (hipGraph replay from the second untimed sweep on), the naive kernel,
    # point lookups and the on-device sums
    if rng.random() < 0.3:
        ok = ok and np.array_equal(pl.compute(), want) and np.array_equal(pl.compute(), want)
        ok = ok and np.array_equal(pl.compute(kernel=1), want)
        if N:
            r_, c_ = rng.integers(0, N, 5), rng.integers(0, N, 5)
            pl.compute_device()
            ok = ok and np.array_equal(pl.result_entries(r_, c_), want[r_, c_].astype(np.float64))
            a_, d_, _ = pl.result_sums()
            w64 = want.astype(np.float64)
            ok = ok and abs(a_ - w64.sum()) <= 1e-9 * max(1.0, w64.sum()) and abs(d_ - np.trace(w64)) <= 1e-9 * max(1.0, np.trace(w64))
    pl.close()
    cases += 1
    if cases % 25 == 0:
        print(f"... {cases} cases, {fails} mismatches, {time.time() - t0:.0f} s", flush=True)
    if not ok:
        fails += 1
        print("MISMATCH", dict(n=n, pf=pf, p1=p1, ps=ps, back=back, cap=int(cap), N=N, modes=modes), flush=True)
print(f"{cases} cases, {fails} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
