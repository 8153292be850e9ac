"""Time of the last level step only (4 generations, N_PRO probands) under an environment switch.
usage: python final_level_ab.py N_PRO [N_PRO ...]  (run on the GPU box).  Was used for the
GENPHI_TEAMS experiment (static chunk teams, DESIGN.md 5: removed again); VARIANTS is the
list of values tried for ENV_NAME."""
ENV_NAME, VARIANTS = "GENPHI_MAX_CPT", ("28", "16")
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    n_pro = int(sys.argv[2])
    ind, fa, mo, sex, pro = synth.random_mating(31034 * 3 + n_pro, n_pro, 4)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
    pl = gen.plan(ped, pro)
    best = 1e9
    for _ in range(4):
        st = pl.compute_device(timing=True)
        best = min(best, st.level_ms[st.n_steps - 1])
    print(json.dumps({"n_pro": n_pro, "cuts": pl.levels()[0], "final_ms": best, "sums": pl.result_sums()[:2]}))
else:
    for n in sys.argv[1:]:
        for t in VARIANTS:
            env = dict(os.environ, **{ENV_NAME: t})
            out = subprocess.run([sys.executable, __file__, "--child", n], env=env, capture_output=True, text=True)
            print(ENV_NAME + "=" + t, out.stdout.strip() or out.stderr[-400:])
