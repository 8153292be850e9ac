import sys, os
sys.path.insert(0, '.')
import numpy as np
import genlib_jl_amd as gen
from genlib_jl_amd import synth
from oracle import oracle as O

def check(name, ind, fa, mo, sex, pro):
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    ref = O.Pedigree(ind, fa, mo).phi(pro)
    pl = gen.plan(ped, pro)
    for k in (0, 1):
        out = pl.compute(kernel=k)
        nbad = int((out != ref).sum())
        print(name, "kernel", k, "modes", sorted(set(pl.step_modes())), "levels", len(pl.levels()[0]), "bad", nbad, flush=True)
    pl.close()

ind, fa, mo, sex = O.read_tsv(gen.geneaJi); ped = O.Pedigree(ind, fa, mo); check("geneaJi", ind, fa, mo, sex, ped.pro())
for args, kw in [((300, 30, 6), dict(skip_permille=100)), ((3000, 300, 12), dict(skip_permille=50)), ((5000, 500, 8), dict(skip_permille=0))]:
    check(str(args), *synth.random_mating(*args, **kw))
ind, fa, mo, sex = O.read_tsv(gen.genea140); ped = O.Pedigree(ind, fa, mo); pro = ped.pro()
for n in (2, 5, 20, 140):
    check(f"genea140[{n}]", ind, fa, mo, sex, pro[:n])
