"""Replays ONE case of tests/stress_random.py (its `case=` number) and prints the plan and which sweeps differ from the oracle.
   python tests/stress_case.py 499794305 [KEY=VALUE ...]     (extra environment knobs override the case's)"""
import os, sys
os.environ["GENPHI_ENV_HOOKS"] = "1"      # environment hooks are read by the library only under this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import stress_random as S


def main():
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    from oracle import oracle as O
    case = int(sys.argv[1])
    r, n_gen, n_ind, n_pro, skip, ind, fa, mo, sex, pro, env = S.make_case(case)
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1)
        if v == "":
            env.pop(k, None)
        else:
            env[k] = v
    for k in S.KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sort = bool(r.random() < 0.5)
    if not sort:
        ind, fa, mo, sex = synth.parents_first_shuffle(ind, fa, mo, sex, seed=case & 0xffff)
    print("case", case, "gens", n_gen, "n_ind", n_ind, "n_pro", n_pro, "skip", skip, "sort", sort, env)
    ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex}, sort=sort)
    pl = gen.plan(ped, pro)
    sizes, both = pl.levels()
    modes = pl.step_modes()
    for k in range(len(modes)):
        print(" step", k, sizes[k], "->", sizes[k + 1], "mode", modes[k], "dragged", both[k], "info", pl.step_info(k), "slots", pl.step_slots(k))
    if "--plan" in os.environ.get("STRESS_CASE_FLAGS", ""):
        return 0
    want = O.Pedigree(ind, fa, mo, sort=sort).phi(pro)
    for name, kw in (("product", {}), ("per-entry", {"kernel": 1}), ("product again", {})):
        got = pl.compute(**kw)
        bad = np.argwhere(got != want)
        print(f" {name}: {len(bad)} of {want.size} entries differ" + (f"; first {bad[0]}, rows {np.unique(bad[:, 0])[:12]}" if len(bad) else ""))
    pl.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
