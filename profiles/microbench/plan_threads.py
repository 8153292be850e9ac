import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import genlib_jl_amd as gen
import bench
for w in (sys.argv[1:] or ("cfg4", "cfg3s")):
    ped, pro, desc = bench.load_workload(w)
    for thr in (1, 2, 4, 8, 16):
        os.environ["GENPHI_PLAN_THREADS"] = str(thr)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); pl = gen.plan(ped, pro); ts.append((time.perf_counter() - t0) * 1e3); pl.close()
        print(w, "threads", thr, "plan ms", [round(t, 2) for t in ts], "median", round(float(np.median(ts)), 2), flush=True)
