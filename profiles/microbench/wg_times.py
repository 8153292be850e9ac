"""Per-workgroup busy time and per-phase stage timing of the last level step.  Needs
genlib.jl_amd/lib/libgenphi_dbg.so = the three csrc files built with the flags of
__graft_entry__.build() plus -DGENPHI_WG_TIMES=1.  usage: python wg_times.py N_PRO"""
import os, sys, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import genlib_jl_amd as gen
from genlib_jl_amd import synth, _capi
_capi.LIB_PATH = os.path.join(root, "genlib.jl_amd", "lib", "libgenphi_dbg.so")
n_pro = int(sys.argv[1])
n_gen = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # GENPHI_DBG_STEP=k picks an earlier level step
ind, fa, mo, sex, pro = synth.random_mating(31034 * (n_gen - 1) + n_pro, n_pro, n_gen)
ped = gen.genealogy({"ind": ind, "father": fa, "mother": mo, "sex": sex})
pl = gen.plan(ped, pro)
for _ in range(3):
    st = pl.compute_device(timing=True)
buf = np.zeros((1024, 3), dtype=np.uint64)
L = _capi.lib()
L.genphi_debug_wg_times.argtypes = [C.c_void_p]
assert L.genphi_debug_wg_times(buf.ctypes.data) == 0
t = buf[:256].astype(np.float64)
t0 = t[:, 0].min()
start, end, items = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]     # wall_clock64: 100 MHz -> us
k_ = int(os.environ.get("GENPHI_DBG_STEP", st.n_steps - 1))
print("cuts", pl.levels()[0], "step", k_, "ms", st.level_ms[k_])
for x in range(8):
    m = np.arange(256) % 8 == x
    print(f"xcd {x}: start {start[m].min():8.1f}..{start[m].max():8.1f} us  end {end[m].min():9.1f}..{end[m].max():9.1f} us  items/WG {items[m].min():.0f}..{items[m].max():.0f}  total {items[m].sum():.0f}")
ck = np.zeros((1024, 2), dtype=np.uint64)
L.genphi_debug_wg_clk.argtypes = [C.c_void_p]
assert L.genphi_debug_wg_clk(ck.ctypes.data) == 0
ck = ck[:256].astype(np.float64)
mhz = (ck[:, 1] - ck[:, 0]) / ((t[:, 1] - t[:, 0]) / 100.0)
print(f"shader clock during the kernel (clock64 ticks per wall-clock us): min {mhz.min():.0f}  median {np.median(mhz):.0f}  max {mhz.max():.0f} MHz")
ph = np.zeros((2, 1024, 16), dtype=np.uint64)
L.genphi_debug_wg_phases.argtypes = [C.c_void_p]
assert L.genphi_debug_wg_phases(ph.ctypes.data) == 0
ph = ph[:, :256].astype(np.float64)
names = ["barrier 1 (others still gathering)", "wait prefetched row + LDS writes", "barrier 2", "index loads + prefetch issue",
         "gathers (+ combine + stores)", "bookkeeping"]
for who, label in ((0, "thread 0 (wave 0)"), (1, "last thread (wave 15)")):
    na, nb = ph[who, :, 6].sum(), ph[who, :, 7].sum()
    print(f"{label}: {na:.0f} A stages, {nb:.0f} B stages; mean us per stage by phase   A stage   B stage")
    for k, nm in enumerate(names):
        print(f"  {nm:40s} {ph[who, :, k].sum() / max(na, 1) / 100.0:9.3f} {ph[who, :, 8 + k].sum() / max(nb, 1) / 100.0:9.3f}")
    print(f"  {'total':40s} {ph[who, :, 0:6].sum() / max(na, 1) / 100.0:9.3f} {ph[who, :, 8:14].sum() / max(nb, 1) / 100.0:9.3f}")
