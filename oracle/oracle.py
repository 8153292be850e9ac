"""ctypes front-end of the CPU oracle (oracle/genphi_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of genphi_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_I64P = C.POINTER(C.c_int64)
_F32P = C.POINTER(C.c_float)


def build():
    """Compile liboracle.so (gcc) and libsparse_oracle.so (g++) (idempotent)."""
    for name, srcname in (("libsparse_oracle.so", "sparse_oracle.cpp"), ("liboracle.so", "genphi_oracle.c")):
        so = os.path.join(_HERE, name)
        src = os.path.join(_HERE, srcname)
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", name])
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_rank_order.argtypes = [C.c_int64, _I64P, _I64P, _I64P, _I64P]
        L.oracle_ped_create.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.POINTER(C.c_void_p)]
        L.oracle_ped_free.argtypes = [C.c_void_p]
        L.oracle_ped_free.restype = None
        L.oracle_pro.argtypes = [C.c_void_p, _I64P, C.c_int64]
        L.oracle_pro.restype = C.c_int64
        L.oracle_founder.argtypes = [C.c_void_p, _I64P, C.c_int64]
        L.oracle_founder.restype = C.c_int64
        L.oracle_phi_pair.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
        L.oracle_f.argtypes = [C.c_void_p, C.c_int64, _I64P, _F32P]
        L.oracle_levels_create.argtypes = [C.c_void_p, C.c_int64, _I64P, C.POINTER(C.c_void_p)]
        L.oracle_levels_free.argtypes = [C.c_void_p]
        L.oracle_levels_free.restype = None
        L.oracle_levels_count.argtypes = [C.c_void_p]
        L.oracle_levels_count.restype = C.c_int32
        L.oracle_levels_cut_size.argtypes = [C.c_void_p, C.c_int32]
        L.oracle_levels_cut_size.restype = C.c_int64
        L.oracle_levels_both.argtypes = [C.c_void_p, C.c_int32]
        L.oracle_levels_both.restype = C.c_int64
        L.oracle_levels_cut_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, _I64P]
        L.oracle_levels_cut_ids.restype = None
        L.oracle_phi_compute.argtypes = [C.c_void_p, C.c_void_p, _F32P, C.c_int32, _I64P]
        L.oracle_phi_compute_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, _I64P, _F32P, _I64P]
        L.oracle_time_level_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, _I64P, C.POINTER(C.c_double)]
        L.oracle_time_level_rows.restype = C.c_int64
        L.oracle_phi_mean.argtypes = [_F32P, C.c_int64]
        L.oracle_phi_mean.restype = C.c_float
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_set_num_threads.restype = None
        _LIB = L
    return _LIB


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(_I64P)


class OracleError(KeyError):
    pass


def read_tsv(path):
    """src/create.jl:161-189: skip the header line, whitespace-split, four Ints per row."""
    rows = np.loadtxt(path, dtype=np.int64, skiprows=1, ndmin=2)
    return rows[:, 0].copy(), rows[:, 1].copy(), rows[:, 2].copy(), rows[:, 3].copy()


class Pedigree:
    """Rank-ordered pedigree as gen.genealogy(...; sort=true) builds it."""

    def __init__(self, ind, father, mother, sort=True):
        L = lib()
        ind, father, mother = _i64(ind), _i64(father), _i64(mother)
        n = len(ind)
        if sort:
            order = np.empty(n, dtype=np.int64)
            rc = L.oracle_rank_order(n, _p(ind), _p(father), _p(mother), _p(order))
            if rc:
                raise OracleError(f"oracle_rank_order rc={rc}")
            ind, father, mother = ind[order], father[order], mother[order]
        self.ind, self.father, self.mother = _i64(ind), _i64(father), _i64(mother)
        self.n = n
        h = C.c_void_p()
        rc = L.oracle_ped_create(n, _p(self.ind), _p(self.father), _p(self.mother), C.byref(h))
        if rc:
            raise OracleError(f"oracle_ped_create rc={rc}")
        self._h = h

    @classmethod
    def from_file(cls, path, sort=True):
        ind, father, mother, _ = read_tsv(path)
        return cls(ind, father, mother, sort=sort)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().oracle_ped_free(self._h)
                self._h = None
        except Exception:
            pass

    def pro(self):
        out = np.empty(self.n, dtype=np.int64)
        k = lib().oracle_pro(self._h, _p(out), self.n)
        return out[:k].copy()

    def founder(self):
        out = np.empty(self.n, dtype=np.int64)
        k = lib().oracle_founder(self._h, _p(out), self.n)
        return out[:k].copy()

    def phi_pair(self, i, j):
        v = C.c_double()
        rc = lib().oracle_phi_pair(self._h, int(i), int(j), C.byref(v))
        if rc:
            raise OracleError(f"unknown ID ({i}, {j})")
        return v.value

    def f(self, ids):
        ids = _i64(ids)
        out = np.empty(len(ids), dtype=np.float32)
        rc = lib().oracle_f(self._h, len(ids), _p(ids), out.ctypes.data_as(_F32P))
        if rc:
            raise OracleError("unknown ID")
        return out

    def levels(self, pro=None):
        """(cut_sizes, both_counts, cut_id_lists) as src/compute.jl:236-262 derives them."""
        L = lib()
        pro = self.pro() if pro is None else _i64(pro)
        lv = C.c_void_p()
        rc = L.oracle_levels_create(self._h, len(pro), _p(pro), C.byref(lv))
        if rc:
            raise OracleError("unknown proband ID")
        try:
            nl = L.oracle_levels_count(lv)
            sizes = [L.oracle_levels_cut_size(lv, k) for k in range(nl)]
            both = [L.oracle_levels_both(lv, k) for k in range(nl - 1)]
            cuts = []
            for k in range(nl):
                ids = np.empty(sizes[k], dtype=np.int64)
                L.oracle_levels_cut_ids(self._h, lv, k, _p(ids))
                cuts.append(ids)
        finally:
            L.oracle_levels_free(lv)
        return sizes, both, cuts

    def phi(self, pro=None, stop_after_levels=0):
        """gen.phi(ped, pro): Matrix{Float32}; with stop_after_levels>0 returns
        (None, entries_evaluated) after that many level steps (bench sampling)."""
        L = lib()
        pro = self.pro() if pro is None else _i64(pro)
        lv = C.c_void_p()
        rc = L.oracle_levels_create(self._h, len(pro), _p(pro), C.byref(lv))
        if rc:
            raise OracleError("unknown proband ID")
        try:
            nl = L.oracle_levels_count(lv)
            n = L.oracle_levels_cut_size(lv, nl - 1)
            done = C.c_int64(0)
            if stop_after_levels > 0 and stop_after_levels < nl - 1:
                rc = L.oracle_phi_compute(self._h, lv, None, stop_after_levels, C.byref(done))
                if rc:
                    raise MemoryError("oracle_phi_compute")
                return None, done.value
            out = np.empty((n, n), dtype=np.float32)
            rc = L.oracle_phi_compute(self._h, lv, out.ctypes.data_as(_F32P), 0, C.byref(done))
            if rc:
                raise MemoryError("oracle_phi_compute")
        finally:
            L.oracle_levels_free(lv)
        return out


    def phi_rows(self, pro, rows):
        """Rows `rows` (positions in `pro`) of gen.phi(ped, pro): every upper level step in full, the last one for the
        sampled rows only (what a test can afford at sizes whose whole matrix costs minutes).  (len(rows), N) Float32."""
        L = lib()
        pro, rows = _i64(pro), _i64(rows)
        lv = C.c_void_p()
        if L.oracle_levels_create(self._h, len(pro), _p(pro), C.byref(lv)):
            raise OracleError("unknown proband ID")
        try:
            nl = L.oracle_levels_count(lv)
            n = L.oracle_levels_cut_size(lv, nl - 1)
            if len(rows) and (rows.min() < 0 or rows.max() >= n):
                raise IndexError("row out of range")
            out = np.empty((len(rows), n), dtype=np.float32)
            done = C.c_int64(0)
            if L.oracle_phi_compute_rows(self._h, lv, len(rows), _p(rows), out.ctypes.data_as(_F32P), C.byref(done)):
                raise MemoryError("oracle_phi_compute_rows")
        finally:
            L.oracle_levels_free(lv)
        return out

    def time_level_rows(self, pro, step, rows):
        """Timing sample (bench.py cpu_baseline): seconds and pair-kernel evaluations of `rows` x all columns of level
        step `step`, real index structure, a matrix of the real size with arbitrary values (see the C source)."""
        L = lib()
        pro, rows = _i64(pro), _i64(rows)
        lv = C.c_void_p()
        if L.oracle_levels_create(self._h, len(pro), _p(pro), C.byref(lv)):
            raise OracleError("unknown proband ID")
        try:
            sec = C.c_double(0.0)
            done = L.oracle_time_level_rows(self._h, lv, step, len(rows), _p(rows), C.byref(sec))
            dt = sec.value
        finally:
            L.oracle_levels_free(lv)
        if done < 0:
            raise MemoryError("oracle_time_level_rows")
        return dt, int(done)

    def branching(self, pro=None, ancestors=None):
        """gen.branching (src/extract.jl:65-186), restated as written there: build the
        children lists (:73-92), mark the ancestors of every proband (:32-42, :93-99) and the
        descendants of every ancestor (:49-58, :100-106) by graph walks, then emit in pedigree
        order the individuals selected by the three cases (:108-184), cutting parents that
        fall outside the kept set.  Pure Python (small inputs).  Returns (ind, father, mother)."""
        n = self.n
        pos = {int(i): k for k, i in enumerate(self.ind)}
        fa = [pos[int(x)] if x else -1 for x in self.father]
        mo = [pos[int(x)] if x else -1 for x in self.mother]
        children = [[] for _ in range(n)]
        for k in range(n):
            if fa[k] >= 0:
                children[fa[k]].append(k)
            if mo[k] >= 0:
                children[mo[k]].append(k)
        is_anc, is_desc = [False] * n, [False] * n

        def walk(start, flags, nxt):                       # the reference recurses; same marks
            stack = [start]
            while stack:
                x = stack.pop()
                if not flags[x]:
                    flags[x] = True
                    stack.extend(nxt(x))

        if pro is not None:
            for i in pro:
                if int(i) not in pos:
                    raise OracleError(int(i))
                walk(pos[int(i)], is_anc, lambda x: [q for q in (fa[x], mo[x]) if q >= 0])
        if ancestors is not None:
            for i in ancestors:
                if int(i) not in pos:
                    raise OracleError(int(i))
                walk(pos[int(i)], is_desc, lambda x: children[x])
        if pro is not None and ancestors is not None:
            keep = [a and d for a, d in zip(is_anc, is_desc)]
        elif pro is not None:
            keep = is_anc
        elif ancestors is not None:
            keep = is_desc
        else:
            keep = [False] * n
        out = [(int(self.ind[k]),
                int(self.father[k]) if fa[k] >= 0 and keep[fa[k]] else 0,
                int(self.mother[k]) if mo[k] >= 0 and keep[mo[k]] else 0) for k in range(n) if keep[k]]
        a = np.array(out, dtype=np.int64).reshape(-1, 3)
        return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()


def phi_mean(phi):
    phi = np.ascontiguousarray(phi, dtype=np.float32)
    return float(lib().oracle_phi_mean(phi.ctypes.data_as(_F32P), phi.shape[0]))


def num_threads():
    return int(lib().oracle_num_threads())


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box gives a container 16 of its
    256 hardware threads: `cpu.max` = "1600000 100000"); one OpenMP thread per hardware thread is throttled there, not sped up."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def fit_threads_to_quota():
    """Sets the oracle's OpenMP thread count to usable_cpus() when that is fewer than the default; returns the threads in use."""
    n = usable_cpus()
    if n < num_threads():
        lib().oracle_set_num_threads(n)
    return num_threads()


_SLIB = None


def _slib():
    global _SLIB
    if _SLIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libsparse_oracle.so"))
        L.sparse_oracle_create.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P]
        L.sparse_oracle_create.restype = C.c_void_p
        L.sparse_oracle_free.argtypes = [C.c_void_p]
        L.sparse_oracle_free.restype = None
        L.sparse_oracle_get.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.sparse_oracle_get.restype = C.c_double
        L.sparse_oracle_info.argtypes = [C.c_void_p, _I64P, _I64P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.sparse_oracle_info.restype = None
        L.sparse_oracle_order.argtypes = [C.c_void_p, _I64P, C.c_int64]
        L.sparse_oracle_order.restype = C.c_int64
        L.sparse_oracle_retired.argtypes = [C.c_void_p, _I64P, _I64P, C.c_int64]
        L.sparse_oracle_retired.restype = C.c_int64
        L.sparse_oracle_entries.argtypes = [C.c_void_p, _I64P, _I64P, _F32P, C.c_int64]
        L.sparse_oracle_entries.restype = C.c_int64
        _SLIB = L
    return _SLIB


class SparsePhi:
    """sparse_phi(pedigree, probandIDs) of the reference (src/compute.jl:321-447), restated literally in
    oracle/sparse_oracle.cpp: the KinshipMatrix it returns."""

    def __init__(self, ped, pro=None):
        pro = ped.pro() if pro is None else np.asarray(pro, dtype=np.int64)
        self.pro = _i64(pro)
        self._h = _slib().sparse_oracle_create(len(ped.ind), _p(ped.ind), _p(ped.father), _p(ped.mother), len(self.pro), _p(self.pro))
        if not self._h:
            raise KeyError("unknown proband ID")

    def __del__(self):
        if getattr(self, "_h", None):
            _slib().sparse_oracle_free(self._h)
            self._h = None

    def __getitem__(self, ids):
        v = _slib().sparse_oracle_get(self._h, int(ids[0]), int(ids[1]))
        if v < 0:
            raise KeyError(ids)
        return v

    def matrix(self):
        """getindex for every pair of probands (float32; what a user sees through ϕ[ID1, ID2])."""
        n = len(self.pro)
        out = np.zeros((n, n), dtype=np.float32)
        for a in range(n):
            for b in range(n):
                out[a, b] = self[(self.pro[a], self.pro[b])]
        return out

    def info(self):
        """(rows, stored entries, sum of all stored values, sum of the diagonal values)."""
        nr, nz = C.c_int64(), C.c_int64()
        sa, sd = C.c_double(), C.c_double()
        _slib().sparse_oracle_info(self._h, C.byref(nr), C.byref(nz), C.byref(sa), C.byref(sd))
        return nr.value, nz.value, sa.value, sd.value

    def show(self):
        nr, nz, _, _ = self.info()
        return f"{nr}×{nr} KinshipMatrix with {nz} stored entries."

    def phi_mean(self):
        """phiMean(::KinshipMatrix) (src/compute.jl:467-472), sums in Float64."""
        nr, _, sa, sd = self.info()
        return np.float32((sa - sd) / (nr * (nr - 1) / 2))

    def order(self):
        cap = 1 << 22
        buf = np.zeros(cap, dtype=np.int64)
        n = _slib().sparse_oracle_order(self._h, _p(buf), cap)
        return buf[:n].copy()

    def retired(self):
        """{ID: 0-based processing index at which the reference drops it from the live set} (src/compute.jl:401-430)."""
        cap = 1 << 22
        ids, at = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
        n = _slib().sparse_oracle_retired(self._h, _p(ids), _p(at), cap)
        return dict(zip(ids[:n].tolist(), at[:n].tolist()))

    def entries(self):
        cap = _slib().sparse_oracle_entries(self._h, None, None, None, 0)
        r, c, v = np.zeros(cap, np.int64), np.zeros(cap, np.int64), np.zeros(cap, np.float32)
        _slib().sparse_oracle_entries(self._h, _p(r), _p(c), v.ctypes.data_as(_F32P), cap)
        return r, c, v
