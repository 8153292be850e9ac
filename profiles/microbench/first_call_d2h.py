"""Where a first gen.phi call and the device-to-host copy of the result go (cfg4 size): GENPHI_TRACE marks of the first and
second compute call, then genphi_result_to_host into a fresh pageable array (cold: page faults) and into the same array again (warm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GENPHI_TRACE"] = "1"
import numpy as np
import torch                                     # (before the library, as in bench.py)
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()          # (context creation is not the library's)
import bench
import genlib_jl_amd as gen
from genlib_jl_amd import _capi
import ctypes as C
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
ped, pro, desc = bench.load_workload(wl)
t0 = time.perf_counter(); pl = gen.plan(ped, pro); print(f"plan {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for k in range(2):
    t0 = time.perf_counter(); pl.compute_device(device=0); print(f"compute call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
n = pl.n_probands
out = np.empty((n, n), dtype=np.float32)
for k in range(3):
    t0 = time.perf_counter()
    rc = _capi.lib().genphi_result_to_host(pl._h, out.ctypes.data_as(C.POINTER(C.c_float)))
    dt = time.perf_counter() - t0
    print(f"result_to_host #{k} ({'cold pages' if k == 0 else 'warm'}): {dt * 1e3:.1f} ms = {out.nbytes / dt / 1e9:.1f} GB/s rc={rc}", flush=True)
t0 = time.perf_counter(); out2 = np.empty((n, n), dtype=np.float32); out2[:] = 0; print(f"first touch of {out2.nbytes / 1e9:.0f} GB by one thread: {time.perf_counter() - t0:.2f} s")
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
