"""How do cuts wider than one LDS row (HALF mode, > 36,864 members) perform?  A few generations of
WIDTH individuals each, N_PRO probands.  usage: python wide_cuts.py WIDTH N_PRO"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import genlib_jl_amd as gen
from genlib_jl_amd import synth
width, n_pro = int(sys.argv[1]), int(sys.argv[2])
ind, fa, mo, sex, pro = synth.random_mating(width * 4 + n_pro, n_pro, 5)
ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
pl = gen.plan(ped, pro)
sizes = pl.levels()[0]
for _ in range(2):
    st = pl.compute_device(timing=True)
ms = [st.level_ms[k] for k in range(st.n_steps)]
print("cuts", sizes, "modes", pl.step_modes())
for k, t in enumerate(ms):
    b = 4.0 * (sizes[k] ** 2 + sizes[k + 1] ** 2)
    print(f"  step {k}: {sizes[k]} -> {sizes[k+1]}  {t:8.3f} ms  {b / t / 1e9:8.2f} TB/s algorithmic  perm {st.perm_ms:.3f} ms" if k == len(ms) - 1 else
          f"  step {k}: {sizes[k]} -> {sizes[k+1]}  {t:8.3f} ms  {b / t / 1e9:8.2f} TB/s algorithmic")
