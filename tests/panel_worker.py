"""Worker of the storage-sharded gen.phi tests: one rank of a gloo group, every rank on cuda:0.
Computes its row block with genlib_jl_amd.distributed.phi_panels, gathers the blocks on rank 0, compares
with the oracle and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import genlib_jl_amd as gen
    from genlib_jl_amd import synth
    from genlib_jl_amd.distributed import phi_panels
    from oracle import oracle as O
    O.fit_threads_to_quota()
    torch.cuda.set_device(0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    d = dist if world > 1 else None
    rank = dist.get_rank() if world > 1 else 0
    cases = {"geneaJi": None, "synth": synth.random_mating(3000, 257, 9, skip_permille=120, seed=4)}
    one = synth.random_mating(1200, 90, 7, skip_permille=60, seed=8)
    mo1 = one[2].copy(); mo1[::11] = 0
    cases["one_parent"] = (one[0], one[1], mo1, one[3], np.concatenate([one[4], one[0][300:310]]))   # + ancestors among the probands
    # kinships down to 2^-81 (rows without an exactness certificate) next to an ordinary pedigree, in a
    # parents-first file order with interleaved depths (sort = false: the rank is the file position)
    base, tiny, tiny2 = synth.random_mating(2500, 200, 8, skip_permille=80, seed=6), synth.chain_two_lines(22), synth.chain_two_lines(40)
    off1, off2 = int(base[0].max()), int(base[0].max()) + int(tiny[0].max())
    rel = lambda a, o: np.where(a > 0, a + o, 0)                      # noqa: E731
    mi = np.concatenate([base[0], tiny[0] + off1, tiny2[0] + off2])
    mf = np.concatenate([base[1], rel(tiny[1], off1), rel(tiny2[1], off2)])
    mm = np.concatenate([base[2], rel(tiny[2], off1), rel(tiny2[2], off2)])
    ms = np.concatenate([base[3], tiny[3], tiny2[3]])
    mp = np.concatenate([base[4], tiny[4] + off1, tiny2[4] + off2])
    cases["tiny_unsorted"] = synth.parents_first_shuffle(mi, mf, mm, ms, seed=3) + (mp,)
    # kernel families of the panel level step, forced through the same hooks as for plans (read at create)
    knobs = ("GENPHI_FULL_MAX_FLOATS", "GENPHI_LDS_CAP_FLOATS", "GENPHI_NO_FAST", "GENPHI_CERT_MIN_EXP", "GENPHI_PANEL_NAIVE",
             "GENPHI_MAX_CPT", "GENPHI_FAST_NT")
    envs = [{}, {"GENPHI_FULL_MAX_FLOATS": "0"}, {"GENPHI_FULL_MAX_FLOATS": "0", "GENPHI_NO_FAST": "1"},
            {"GENPHI_FULL_MAX_FLOATS": "0", "GENPHI_CERT_MIN_EXP": "-7", "GENPHI_MAX_CPT": "4"},
            {"GENPHI_FULL_MAX_FLOATS": "0", "GENPHI_FAST_NT": "512", "GENPHI_CERT_MIN_EXP": "-3"},
            {"GENPHI_NO_FAST": "1"}, {"GENPHI_CERT_MIN_EXP": "-5"}, {"GENPHI_LDS_CAP_FLOATS": "200"}, {"GENPHI_PANEL_NAIVE": "1"}]
    ok, sent, modes_seen, bad, k_env = True, 0, set(), [], 0
    for name, c in [(n, c) for n, c in cases.items() for _ in envs]:
        env = envs[k_env % len(envs)]; k_env += 1
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        if c is None:
            ped = gen.genealogy(gen.geneaJi)
            pro = gen.pro(ped)
            oped = O.Pedigree.from_file(gen.geneaJi)
        else:
            ind, fa, mo, sex, pro = c
            srt = name != "tiny_unsorted"
            ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex}, sort=srt)
            oped = O.Pedigree(ind, fa, mo, sort=srt)
        st = {}
        rows, r0 = phi_panels(ped, pro, dist=d, device=0, stats=st)
        sent += st["exchange_bytes_sent"]
        modes_seen |= set(st.get("step_modes", []))
        parts = [None] * world
        if world > 1:
            dist.all_gather_object(parts, (r0, rows))
        else:
            parts = [(r0, rows)]
        if rank == 0:
            parts.sort(key=lambda t: t[0])
            full = np.concatenate([p[1] for p in parts], axis=0)
            want = oped.phi(pro)
            good = full.shape == want.shape and bool(np.array_equal(full, want))
            if not good:
                bad.append((name, env))
            ok &= good
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "world": world, "exchange_bytes_sent_rank0": int(sent), "modes_seen": sorted(modes_seen),
                          "bad": bad}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
