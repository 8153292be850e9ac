#!/usr/bin/env python3
"""Summarise a profiles/collect.sh run (gpurun_out/prof_<tag>_<workload>/) into the tracked files
    profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
    profiles/<tag>_<workload>_levels.csv         per level-kernel dispatch: ms, HBM read/write bytes, L2 hit %
    profiles/traffic_<workload>.json             what bench.py reports as roofline.traffic
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB and
collected in separate passes; on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane)
coalesced read stream -- the staging loads of the level kernels are that -- so it is doubled;
WRITE_SIZE is taken as is (it matches the known output bytes of these kernels to <1 %)."""
import collections
import csv
import re
import glob
import json
import os
import shutil
import sys

import hashlib

tag, wl = (sys.argv + ["r01", "cfg4"])[1:3]


def csrc_sha16(root=None):
    """Fingerprint of the kernel sources of the DENSE path (what the traffic files measure): ties a committed
    profiles/traffic_<workload>.json to the sources it was collected with (works on the GPU box, which has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "genlib.jl_amd", "csrc")
    for name in ("genphi_hip.hip", "sparse_levels.hip", "sparse_levels.h", "planner.cpp", "planner.h", "panel_launch.h"):
        h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")
dst = os.path.join(root, "profiles")


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)     # newest run in that directory


shutil.copy(one("kt/*/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))


def agg(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, {"name": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return d


F, W = agg(one("pmc_fetch/*/*counter_collection.csv")), agg(one("pmc_write/*/*counter_collection.csv"))
try:
    T = agg(one("pmc_tcp/*/*counter_collection.csv"))      # requests from the CUs into L2 (64-byte requests)
except Exception:
    T = {}
KERNELS = ("level_", "levels_small", "rows_compact", "rows_avg", "drag_rows", "transpose_block", "transpose_slots", "slots_scatter", "slots_clear",
           "copy_block", "pad_zero", "colperm", "group_split", "sparse_step", "sparse_dense", "sparse_identity")
rows, tot, sweeps, steps = [], 0.0, 0, 0
# A plan's sparse cuts (sparse_levels.hip): the first sparse_identity_kernel of the process starts the CALIBRATION run -- its list steps
# are not part of a sweep and are left out.  The FIRST sweep after it takes cut k's lists as the calibration run left them and starts
# at sparse_dense_kernel: a partial sweep, listed in the level table but kept out of the per-sweep average.  Every later sweep starts
# with a sparse_identity_kernel of its own.
n_sparse_ident, first_sweep = 0, False
for k in F:
    name = F[k]["name"]
    if not any(t in name for t in KERNELS) or k not in W:
        continue
    if "sparse_identity_kernel" in name:
        n_sparse_ident += 1
        first_sweep = False
        if n_sparse_ident >= 2:
            sweeps += 1
    if n_sparse_ident == 1 and "sparse_dense_kernel" in name:
        first_sweep = True
    if n_sparse_ident == 1 and "sparse_" in name and not first_sweep:
        continue
    m = re.search(r"\(anonymous namespace\)::([A-Za-z_0-9]+(?:<[^>]*>)?)\(", name)      # (the argument list names the namespace again)
    short = (m.group(1) if m else name.split("(")[0]).replace(", ", ",")
    if "level_identity_kernel" in name:
        sweeps += 1                                        # one per gen.phi sweep (level step 0)
    if F[k]["ms"] < 0.02 and "level_split_kernel" in name:
        continue            # the grouping-exact SPLIT kernel of a level whose groups are all certified: it finds an empty list and ends at once
    rd = F[k].get("FETCH_SIZE", 0.0) * 1024 * 2            # KiB -> B, gfx950 wide-read correction
    wr = W[k].get("WRITE_SIZE", 0.0) * 1024
    hit, miss = W[k].get("TCC_HIT_sum", 0.0), W[k].get("TCC_MISS_sum", 0.0)
    rows.append((k, short, round(F[k]["ms"], 4), int(rd), int(wr), round(100 * hit / max(hit + miss, 1), 1),
                 int(T.get(k, {}).get("TCP_TCC_READ_REQ_sum", -1)), int(T.get(k, {}).get("TCP_TCC_WRITE_REQ_sum", -1))))
    if not first_sweep:
        tot += rd + wr
with open(os.path.join(dst, f"{tag}_{wl}_levels.csv"), "w") as fh:
    fh.write("dispatch,kernel,ms_under_pmc,hbm_read_bytes,hbm_write_bytes,l2_hit_pct,tcp_tcc_read_req,tcp_tcc_write_req\n")
    for r in rows:
        fh.write(",".join(('"%s"' % x) if isinstance(x, str) else str(x) for x in r) + "\n")
# level steps per sweep: from the bench line of the kernel-trace run
n_steps = None
try:
    for line in open(os.path.join(src, "kt.log")):
        if line.startswith("{"):
            n_steps = json.loads(line)["roofline"]["launches_per_step"]
except Exception:
    pass
sweeps = max(sweeps, 1)
per_step = tot / sweeps / n_steps if n_steps else None
json.dump({"workload": wl, "tag": tag, "csrc_sha16": csrc_sha16(root), "sweeps_profiled": sweeps, "level_steps_per_sweep": n_steps,
           "hbm_bytes_per_sweep": tot / sweeps,
           "hbm_bytes_per_launch": per_step,
           "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-read correction) and --pmc WRITE_SIZE in separate passes, "
                     "KiB -> bytes, summed over every kernel of the profiled gen.phi sweeps; per launch = per level step "
                     "(a WIDE level step is several kernels)",
           "csrc_sha16_note": "fingerprint of genphi_hip.hip, sparse_levels.hip / .h, planner.cpp, planner.h, panel_launch.h as collected"},
          open(os.path.join(dst, f"traffic_{wl}.json"), "w"), indent=1)
print(f"{len(rows)} dispatches, {sweeps} sweeps, {tot / sweeps / 1e9:.2f} GB per sweep" + (f", {per_step / 1e9:.3f} GB per level step" if per_step else ""))
