"""What ONE rank of an N-GPU run does at cfg4 (its row shard of the last level + the upper levels
restricted to the shard's ancestors), timed on one GPU.  usage: python shard_time.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import genlib_jl_amd as gen
from genlib_jl_amd import synth
ind, fa, mo, sex, pro = synth.random_mating(1_000_000, 100_000, 30)
ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
pl = gen.plan(ped, pro)
n = pl.n_probands
for world in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    for prune in (True, False):
        if world == 1 and not prune:
            continue
        os.environ.pop("GENPHI_NO_SHARD_PRUNE", None)
        if not prune:
            os.environ["GENPHI_NO_SHARD_PRUNE"] = "1"
        rank = world // 2
        rows = None if world == 1 else ((n * rank) // world, (n * (rank + 1)) // world)
        pl.compute_device(rows=rows)
        best = min(pl.compute_device(rows=rows, timing=True).total_ms for _ in range(3))
        lm = [pl.stats.level_ms[k] for k in range(pl.stats.n_steps)]
        print(f"N={world} rank {rank} prune={prune}: {best:7.2f} ms  upper {sum(lm[:-1]):6.2f}  last {lm[-1]:6.2f}  "
              f"levels 20..28 {[round(x, 2) for x in lm[20:28]]}", flush=True)
