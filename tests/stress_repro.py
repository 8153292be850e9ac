"""Replays tests/stress_random.py with the same seed until the first mismatch and dissects it."""
import os, sys, time
os.environ["GENPHI_ENV_HOOKS"] = "1"      # environment hooks are read by the library only under this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import genlib_jl_amd as gen
from oracle import oracle
from test_gpu_parity import _random_pedigree

rng = np.random.default_rng(int(sys.argv[1]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
case = 0
while True:
    n = int(rng.choice([30, 120, 400, 900, 1500]))
    pf, p1, ps = rng.choice([0.01, 0.05, 0.3]), rng.choice([0.0, 0.1, 0.3]), rng.choice([0.0, 0.05])
    back = int(rng.choice([5, 50, 400, n]))
    ind, fa, mo, sex = _random_pedigree(rng, n, pf, p1, ps, back)
    cap = rng.choice([0, 0, 4096, 900, 150])
    fm = rng.random() < 0.3
    oped = oracle.Pedigree(ind, fa, mo)
    pro = oped.pro() if rng.random() < 0.5 else rng.choice(ind, size=int(rng.integers(1, min(n, 300) + 1)), replace=True)
    os.environ.pop("GENPHI_LDS_CAP_FLOATS", None); os.environ.pop("GENPHI_FULL_MAX_FLOATS", None)
    if cap: os.environ["GENPHI_LDS_CAP_FLOATS"] = str(cap)
    if fm: os.environ["GENPHI_FULL_MAX_FLOATS"] = "64"
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    want = oped.phi(pro)
    pl = gen.plan(ped, pro)
    N = pl.n_probands
    got = pl.compute()
    ok_full = np.array_equal(got, want)
    a = b = None
    ok_sh = True
    if N >= 3:
        a, b = sorted(int(x) for x in rng.choice(np.arange(1, N), size=2, replace=False))
        parts = [pl.compute(rows=r) for r in ((0, a), (a, b), (b, N))]
        ok_sh = np.array_equal(np.concatenate(parts, axis=0), want)
    case += 1
    if ok_full and ok_sh:
        pl.close(); continue
    print("case", case, dict(n=n, pf=pf, p1=p1, ps=ps, back=back, cap=int(cap), fm=fm, N=N, a=a, b=b), "full ok", ok_full, "shards ok", ok_sh)
    print("levels", pl.levels()[0][:40], "n_levels", len(pl.levels()[0]), "modes", pl.step_modes()[:60])
    def diff(name, m, ref):
        d = np.argwhere(m != ref)
        print(f"  {name}: {len(d)} entries differ; first {d[:5].tolist()} got {[float(m[tuple(x)]) for x in d[:3]]} want {[float(ref[tuple(x)]) for x in d[:3]]}")
    diff("full", got, want)
    if N >= 3:
        for r, prt in zip(((0, a), (a, b), (b, N)), parts):
            diff(f"shard {r}", prt, want[r[0]:r[1]])
    for env in ("GENPHI_NO_SHARD_PRUNE", "GENPHI_NO_SMALL", "GENPHI_NO_IDENTITY", "GENPHI_NO_GRAPH"):
        os.environ[env] = "1"
        p2 = gen.plan(ped, pro)
        g2 = p2.compute()
        s2 = np.concatenate([p2.compute(rows=r) for r in ((0, a), (a, b), (b, N))], axis=0) if N >= 3 else g2
        print(f"  with {env}: full ok {np.array_equal(g2, want)}  shards ok {np.array_equal(s2, want)}")
        p2.close(); os.environ.pop(env)
    sizes = pl.levels()[0]
    nst = len(sizes) - 1
    lo, hi = 0, nst            # smallest k such that pruning only steps >= k still fails... find boundary
    os.environ["GENPHI_NO_SMALL"] = "1"
    def shards_ok(k):
        os.environ["GENPHI_SHARD_PRUNE_MIN_STEP"] = str(k)
        p3 = gen.plan(ped, pro)
        r = np.array_equal(np.concatenate([p3.compute(rows=r) for r in ((0, a), (a, b), (b, N))], axis=0), want)
        p3.close()
        return r
    # pruning steps >= k: ok for k = nst (nothing pruned); find the largest k that fails
    bad = [k for k in range(nst) if not shards_ok(k)]
    print("  (per-level launches) pruning only steps >= k fails for k in", bad[:5], "...", bad[-5:], "of", nst, "steps")
    if bad:
        k = bad[-1]
        print("   last failing k =", k, "cut sizes around:", sizes[max(k - 1, 0):k + 3], "modes", pl.step_modes()[max(k - 1, 0):k + 2])
    if bad:
        k = bad[-1]
        os.environ["GENPHI_SHARD_PRUNE_MIN_STEP"] = str(k)
        fixes = []
        for row in range(sizes[k + 1]):
            os.environ["GENPHI_SHARD_FORCE"] = f"{k}:{row}"
            p3 = gen.plan(ped, pro)
            okk = np.array_equal(np.concatenate([p3.compute(rows=r) for r in ((0, a), (a, b), (b, N))], axis=0), want)
            p3.close()
            if okk: fixes.append(row)
        os.environ.pop("GENPHI_SHARD_FORCE", None)
        print("   forcing one extra row of cut", k + 1, "at step", k, "fixes it for rows", fixes)
    os.environ.pop("GENPHI_SHARD_PRUNE_MIN_STEP", None); os.environ.pop("GENPHI_NO_SMALL", None)
    p2 = gen.plan(ped, pro)
    print("  naive kernel full ok", np.array_equal(p2.compute(kernel=1), want))
    p2.close()
    if skip <= 0: break
    skip -= 1
