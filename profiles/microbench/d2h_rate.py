"""Device-to-host rate of genphi_result_to_host at cfg4 size (40 GB), pinned ring vs direct."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import genlib_jl_amd as gen
from genlib_jl_amd import synth
n_pro = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ind, fa, mo, sex, pro = synth.random_mating(31034 * 3 + n_pro, n_pro, 4)
ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
pl = gen.plan(ped, pro)
pl.compute_device()
ref = None
for mode in ("pinned", "pinned", "pageable", "pinned-16thr"):
    os.environ.pop("GENPHI_D2H_PAGEABLE", None); os.environ.pop("GENPHI_D2H_THREADS", None)
    if mode == "pageable": os.environ["GENPHI_D2H_PAGEABLE"] = "1"
    if mode == "pinned-16thr": os.environ["GENPHI_D2H_THREADS"] = "16"
    t0 = time.perf_counter(); a = pl.result_to_host(); dt = time.perf_counter() - t0
    chk = (float(a[0, :100].sum()), float(a[-1, -100:].sum()), float(a[n_pro // 2, ::997].sum()))
    if ref is None: ref = chk
    print(f"{mode:14s} {a.nbytes / 1e9:6.1f} GB in {dt:6.3f} s = {a.nbytes / dt / 1e9:6.1f} GB/s  same={chk == ref}", flush=True)
    del a
