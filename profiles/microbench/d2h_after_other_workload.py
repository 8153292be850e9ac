"""Does a workload run earlier in the process change the copy of a 400 MB result?  (call_wall.py measured cfg3's copy at 20 ms after cfg2
and 8.5 ms alone.)  usage: python profiles/microbench/d2h_after_other_workload.py [first workloads ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, genlib_jl_amd as gen
now = time.perf_counter
def copies(tag):
    ped, pro, _ = bench.load_workload("cfg3")
    time.sleep(float(os.environ.get("SLEEP_BEFORE", "0")))
    pl = gen.plan(ped, pro); pl.compute_device()
    time.sleep(float(os.environ.get("SLEEP_AFTER", "0")))
    ts = []
    for _ in range(6):
        t0 = now(); a = pl.result_to_host(); ts.append((now() - t0) * 1e3); a = None
    pl.close()
    print("%-40s cfg3 copies %s ms" % (tag, [round(t, 1) for t in ts]), flush=True)
copies("first thing in the process:")
for w in sys.argv[1:]:
    ped, pro, _ = bench.load_workload(w)
    mode = os.environ.get("HOW", "phi")
    if mode == "phi":
        gen.phi(ped, pro)
    elif mode == "device":
        pl = gen.plan(ped, pro); pl.compute_device(); pl.close()
    elif mode == "plan":
        gen.plan(ped, pro).close()
    copies("after %s (%s):" % (w, mode))
