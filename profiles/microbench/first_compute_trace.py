"""GENPHI_TRACE of the first sweep of a fresh plan on a warm device (runtime started, code objects loaded, blocks kept).
usage: GENPHI_TRACE=1 python profiles/microbench/first_compute_trace.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench, genlib_jl_amd as gen
for w in (sys.argv[1:] or ["cfg2", "cfg3"]):
    ped, pro, _ = bench.load_workload(w)
    for rep in range(3):
        pl = gen.plan(ped, pro)
        sys.stderr.write(f"==== {w}: fresh plan {rep}, first compute_device\n"); sys.stderr.flush()
        t0 = time.perf_counter(); pl.compute_device(); t = (time.perf_counter() - t0) * 1e3
        sys.stderr.write(f"==== {w}: first compute_device {t:.2f} ms\n"); sys.stderr.flush()
        pl.close()
