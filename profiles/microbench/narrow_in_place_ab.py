"""In-place runs at FULL / SPLIT widths: is the planner's choice (cost model + width floor) the faster one on mid-size pedigrees?
Sweep time (HIP events inside the library, best of 5) with the default plan, with GENPHI_STAY_NARROW=0 (row kernels everywhere), with the
cost model on bytes alone (GENPHI_STAY_OVERHEAD_K=0: the first version of round 4) and with twice the fixed cost per block-assembled step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import genlib_jl_amd as gen
from genlib_jl_amd import synth

CASES = [((20000, 300, 18), dict(skip_permille=500, seed=2)), ((30000, 400, 30), dict(skip_permille=600, seed=3)),
         ((12000, 500, 24), dict(skip_permille=700, seed=8)), ((60000, 3000, 12), dict(skip_permille=100, seed=5)),
         ((100000, 5000, 20), dict(skip_permille=50)), ((100000, 10000, 20), dict(skip_permille=50)),
         ((200000, 10000, 25), dict(skip_permille=100)), ((150000, 8000, 12), dict(skip_permille=200))]
for args, kw in CASES:
    ind, fa, mo, sex, pro = synth.random_mating(*args, **kw)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    res = {}
    for tag, env in (("default", {}), ("row kernels", {"GENPHI_STAY_NARROW": "0"}), ("bytes only", {"GENPHI_STAY_OVERHEAD_K": "0"}), ("overhead x2", {"GENPHI_STAY_OVERHEAD_K": "128000"})):
        for k in ("GENPHI_STAY_NARROW", "GENPHI_STAY_NARROW_MIN", "GENPHI_STAY_OVERHEAD_K"):
            os.environ.pop(k, None)
        os.environ.update(env)
        pl = gen.plan(ped, pro)
        sizes, both = pl.levels()
        n_stay = sum(pl.step_slots(k)[0] & 1 for k in range(len(sizes) - 1))
        pl.compute_device(device=0)
        best = min(pl.compute_device(device=0, timing=True).total_ms for _ in range(5))
        res[tag] = (best, n_stay)
        pl.close()
    print(args, kw, "max cut", max(sizes), " | ".join(f"{t}: {v[0]:.3f} ms ({v[1]} in place)" for t, v in res.items()), flush=True)
