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
    ok, sent = True, 0
    for name, c in cases.items():
        if c is None:
            ped = gen.genealogy(gen.geneaJi)
            pro = gen.pro(ped)
            oped = O.Pedigree.from_file(gen.geneaJi)
        else:
            ind, fa, mo, sex, pro = c
            ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
            oped = O.Pedigree(ind, fa, mo)
        st = {}
        rows, r0 = phi_panels(ped, pro, dist=d, device=0, stats=st)
        sent += st["exchange_bytes_sent"]
        parts = [None] * world
        if world > 1:
            dist.all_gather_object(parts, (r0, rows))
        else:
            parts = [(r0, rows)]
        if rank == 0:
            parts.sort(key=lambda t: t[0])
            full = np.concatenate([p[1] for p in parts], axis=0)
            want = oped.phi(pro)
            ok &= full.shape == want.shape and bool(np.array_equal(full, want))
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "world": world, "exchange_bytes_sent_rank0": int(sent)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
