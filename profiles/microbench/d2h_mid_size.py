"""Device-to-host copy of a mid-size result (cfg3: 1e4 x 1e4 Float32 = 400 MB) through genphi_result_to_host: the copy and the
release of the destination array timed apart, for ordinary (np.empty) and huge-page (GENPHI_HOST_HUGEPAGES=1) destinations, with
the previous result still alive during the copy (what `phi = gen.phi(...)` in a loop does) or released before it.
usage: GENPHI_ENV_HOOKS=1 python profiles/microbench/d2h_mid_size.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, genlib_jl_amd as gen
ped, pro, _ = bench.load_workload("cfg3")
now = time.perf_counter
for hp in ("0", "1", "0", "1"):
    os.environ["GENPHI_HOST_HUGEPAGES"] = hp
    for alive in (True, False):
        copy, free, keep = [], [], None
        for _ in range(6):
            pl = gen.plan(ped, pro); pl.compute_device()
            if not alive and keep is not None:
                t0 = now(); keep = None; free.append((now() - t0) * 1e3)
            t0 = now(); a = pl.result_to_host(); copy.append((now() - t0) * 1e3)
            pl.close()
            if alive and keep is not None:
                t0 = now(); keep = None; free.append((now() - t0) * 1e3)
            keep = a; a = None
        keep = None
        print("huge pages %s, previous result %s: copy %s ms; releasing a result %s ms" % (
            hp, "alive during the copy" if alive else "released before the copy", [round(t, 1) for t in copy], [round(t, 1) for t in free]), flush=True)
os.environ["GENPHI_PLAN_CACHE"] = "0"
for hp in ("0", "1"):
    os.environ["GENPHI_HOST_HUGEPAGES"] = hp
    whole, phi = [], None
    t00 = now()
    for _ in range(8):
        t0 = now(); r = gen.phi(ped, pro); whole.append((now() - t0) * 1e3); phi = r; r = None
    phi = None
    print("huge pages %s: gen.phi in a loop, result replaced every time: calls %s ms, loop %.1f ms per iteration" % (
        hp, [round(t, 1) for t in whole], (now() - t00) * 1e3 / 8), flush=True)
for thr in (1, 2, 4, 8, 16):
    os.environ["GENPHI_D2H_THREADS"] = str(thr); os.environ["GENPHI_HOST_HUGEPAGES"] = "0"
    pl = gen.plan(ped, pro); pl.compute_device()
    ts = []
    for _ in range(6):
        t0 = now(); a = pl.result_to_host(); ts.append((now() - t0) * 1e3); a = None
    pl.close()
    print("one plan, copies in a row, %2d copy threads: %s ms" % (thr, [round(t, 1) for t in ts]), flush=True)
