"""Wall clock of a ONE-SHOT gen.phi call on a warm device (plan + upload + sweep + device-to-host copy + teardown), per workload:
what a caller of the drop-in API pays per call, next to the sweep the bench line times.  usage: call_wall.py [workload ...]
GENPHI_TRACE=1 in the environment prints the phases of every call on stderr."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import genlib_jl_amd as gen  # noqa: E402

for w in (sys.argv[1:] or ["cfg2", "cfg3", "cfg3s", "cfg5"]):
    ped, pro, desc = bench.load_workload(w)
    gen.phi(ped, pro)                                   # warm: runtime, code objects, LDS opt-ins
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        phi = gen.phi(ped, pro)
        ts.append((time.perf_counter() - t0) * 1e3)
        phi = None                                      # (outside the timed interval: unmapping a 400 MB array takes the interpreter 15-20 ms)
    t0 = time.perf_counter(); pl = gen.plan(ped, pro); t_plan = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); pl.compute_device(); t_first = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); pl.compute_device(); t_second = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); res = pl.result_to_host(); t_d2h = (time.perf_counter() - t0) * 1e3
    res = None
    t0 = time.perf_counter(); pl.close(); t_close = (time.perf_counter() - t0) * 1e3
    print(f"{w}: gen.phi call wall median {np.median(ts):.2f} ms (min {min(ts):.2f}, max {max(ts):.2f}); "
          f"plan {t_plan:.2f}, first compute {t_first:.2f}, second compute {t_second:.2f}, to host {t_d2h:.2f}, close {t_close:.2f}", flush=True)
