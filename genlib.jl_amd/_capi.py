"""ctypes binding of include/genphi.h (the same symbols the Julia shim `ccall`s).

There is deliberately NO fallback: if the HIP library is missing or no GPU is usable the
calls raise -- the product path never routes through a CPU implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgenphi.so")

GENPHI_OK = 0
GENPHI_ERR_UNKNOWN_ID = 1
GENPHI_ERR_ORDER = 2
GENPHI_ERR_DUPLICATE_ID = 3
GENPHI_ERR_ALLOC = 4
GENPHI_ERR_DEVICE = 5
GENPHI_ERR_ARG = 6

GENPHI_MAX_STAT_LEVELS = 1024
GENPHI_FLAG_NO_GRAPH = 1
GENPHI_FLAG_STORAGE_F64 = 2
GENPHI_FLAG_NO_SPARSE = 4

_I64P = C.POINTER(C.c_int64)
_F32P = C.POINTER(C.c_float)


STEP_FN = C.CFUNCTYPE(None, C.c_int32, C.c_int32, C.c_void_p)      # genphi_step_fn


class GenphiOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("kernel", C.c_int32), ("row_begin", C.c_int64),
                ("row_end", C.c_int64), ("timing", C.c_int32), ("flags", C.c_int32)]


class GenphiStats(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("timed", C.c_int32), ("total_ms", C.c_double),
                ("final_ms", C.c_double), ("perm_ms", C.c_double), ("algorithmic_bytes", C.c_double), ("max_cut", C.c_int64),
                ("level_ms", C.c_float * GENPHI_MAX_STAT_LEVELS), ("level_rows", C.c_int64 * GENPHI_MAX_STAT_LEVELS)]


# every symbol include/genphi.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "genphi_plan_create", "genphi_plan_create_tuned", "genphi_tuning_create", "genphi_tuning_set", "genphi_tuning_destroy", "genphi_plan_levels", "genphi_plan_n_probands", "genphi_plan_step_mode", "genphi_plan_step_info", "genphi_plan_step_slots",
    "genphi_plan_algorithmic_bytes", "genphi_plan_device_bytes", "genphi_plan_device_bytes_needed", "genphi_plan_sparse_levels", "genphi_plan_step_walk", "genphi_plan_set_step_hook", "genphi_compute_device", "genphi_result_device",
    "genphi_result_to_host", "genphi_result_to_host_f64", "genphi_phi_pairs", "genphi_result_sums", "genphi_result_entries",
    "genphi_compute_f32",
    "genphi_genealogy_read", "genphi_branching", "genphi_free", "genphi_release_cached", "genphi_cached_bytes", "genphi_plan_release_device", "genphi_plan_destroy",
    "genphi_last_error",
    "genphi_version", "genphi_sparse_phi", "genphi_sparse_info", "genphi_sparse_stats", "genphi_sparse_schedule", "genphi_genealogy_order", "genphi_sparse_get", "genphi_sparse_entries", "genphi_sparse_destroy",
    "genphi_panel_create", "genphi_panel_step_mode", "genphi_panel_step_ms", "genphi_panel_n_steps", "genphi_panel_n_probands", "genphi_panel_result_rows", "genphi_panel_exchange_counts",
    "genphi_panel_device_bytes", "genphi_panel_begin", "genphi_panel_pack", "genphi_panel_compute", "genphi_panel_pack_on", "genphi_panel_compute_on", "genphi_panel_sync",
    "genphi_panel_result_to_host",
    "genphi_panel_destroy",
]

_lib = None


class GenphiLibraryMissing(RuntimeError):
    pass


class GenphiDeviceError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GenphiLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for gen.phi.")
        L = C.CDLL(LIB_PATH)
        L.genphi_plan_create.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.POINTER(C.c_void_p)]
        L.genphi_plan_create.restype = C.c_int
        L.genphi_plan_create_tuned.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.c_void_p, C.POINTER(C.c_void_p)]
        L.genphi_plan_create_tuned.restype = C.c_int
        L.genphi_tuning_create.argtypes = []
        L.genphi_tuning_create.restype = C.c_void_p
        L.genphi_tuning_set.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.genphi_tuning_set.restype = C.c_int
        L.genphi_tuning_destroy.argtypes = [C.c_void_p]
        L.genphi_tuning_destroy.restype = None
        L.genphi_plan_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(_I64P), C.POINTER(_I64P)]
        L.genphi_plan_levels.restype = C.c_int
        L.genphi_plan_n_probands.argtypes = [C.c_void_p]
        L.genphi_plan_n_probands.restype = C.c_int64
        L.genphi_plan_step_mode.argtypes = [C.c_void_p, C.c_int32]
        L.genphi_plan_step_mode.restype = C.c_int
        L.genphi_plan_step_info.argtypes = [C.c_void_p, C.c_int32, _I64P]
        L.genphi_plan_step_info.restype = C.c_int
        L.genphi_plan_step_slots.argtypes = [C.c_void_p, C.c_int32, _I64P]
        L.genphi_plan_step_slots.restype = C.c_int
        _I32P = C.POINTER(C.c_int32)
        # (every symbol is bound unconditionally: this binding needs the build it was written for -- include/genphi.h)
        L.genphi_plan_step_walk.argtypes = [C.c_void_p, C.c_int32, _I64P, _I64P, _I64P, _I32P, _I32P, _I32P]
        L.genphi_plan_step_walk.restype = C.c_int
        L.genphi_plan_set_step_hook.argtypes = [C.c_void_p, STEP_FN, C.c_void_p]
        L.genphi_plan_set_step_hook.restype = C.c_int
        L.genphi_plan_device_bytes_needed.argtypes = [C.c_void_p]
        L.genphi_plan_device_bytes_needed.restype = C.c_int64
        L.genphi_plan_device_bytes.argtypes = [C.c_void_p]
        L.genphi_plan_device_bytes.restype = C.c_int64
        L.genphi_release_cached.argtypes = []
        L.genphi_release_cached.restype = None
        L.genphi_cached_bytes.argtypes = []
        L.genphi_cached_bytes.restype = C.c_int64
        L.genphi_plan_sparse_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int32), _I64P, _I64P, C.c_int32]
        L.genphi_plan_sparse_levels.restype = C.c_int
        L.genphi_plan_algorithmic_bytes.argtypes = [C.c_void_p]
        L.genphi_plan_algorithmic_bytes.restype = C.c_double
        L.genphi_compute_device.argtypes = [C.c_void_p, C.POINTER(GenphiOpts), C.POINTER(GenphiStats)]
        L.genphi_compute_device.restype = C.c_int
        L.genphi_result_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _I64P, _I64P, _I64P]
        L.genphi_result_device.restype = C.c_int
        L.genphi_result_to_host.argtypes = [C.c_void_p, _F32P]
        L.genphi_result_to_host.restype = C.c_int
        L.genphi_result_to_host_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.genphi_result_to_host_f64.restype = C.c_int
        L.genphi_phi_pairs.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, _I64P, C.POINTER(C.c_double), C.c_int32]
        L.genphi_phi_pairs.restype = C.c_int
        L.genphi_result_sums.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), _I64P]
        L.genphi_result_sums.restype = C.c_int
        L.genphi_result_entries.argtypes = [C.c_void_p, C.c_int64, _I64P, _I64P, C.POINTER(C.c_double)]
        L.genphi_result_entries.restype = C.c_int
        L.genphi_branching.argtypes = [C.c_int64, _I64P, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.c_int64, _I64P,
                                       _I64P, C.POINTER(_I64P), C.POINTER(_I64P), C.POINTER(_I64P), C.POINTER(_I64P)]
        L.genphi_branching.restype = C.c_int
        L.genphi_compute_f32.argtypes = [C.c_void_p, _F32P, C.POINTER(GenphiOpts), C.POINTER(GenphiStats)]
        L.genphi_compute_f32.restype = C.c_int
        L.genphi_genealogy_read.argtypes = [C.c_char_p, C.c_int32, _I64P, C.POINTER(_I64P), C.POINTER(_I64P),
                                            C.POINTER(_I64P), C.POINTER(_I64P)]
        L.genphi_genealogy_read.restype = C.c_int
        L.genphi_genealogy_order.argtypes = [C.c_int64, _I64P, _I64P, _I64P, _I64P, C.c_int32, _I64P, C.POINTER(_I64P), C.POINTER(_I64P),
                                             C.POINTER(_I64P), C.POINTER(_I64P)]
        L.genphi_genealogy_order.restype = C.c_int
        L.genphi_free.argtypes = [C.c_void_p]
        L.genphi_free.restype = None
        L.genphi_plan_release_device.argtypes = [C.c_void_p]
        L.genphi_plan_release_device.restype = C.c_int
        L.genphi_plan_destroy.argtypes = [C.c_void_p]
        L.genphi_plan_destroy.restype = None
        L.genphi_sparse_phi.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.c_int32, C.POINTER(C.c_void_p)]
        L.genphi_sparse_phi.restype = C.c_int
        L.genphi_sparse_info.argtypes = [C.c_void_p, _I64P, _I64P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.genphi_sparse_info.restype = C.c_int
        L.genphi_sparse_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), _I64P, _F32P,
                                          C.POINTER(C.c_double), C.c_int32]
        L.genphi_sparse_stats.restype = C.c_int
        L.genphi_sparse_schedule.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.c_int64, _I64P, _I64P, C.POINTER(C.c_int32), _I64P]
        L.genphi_sparse_schedule.restype = C.c_int
        L.genphi_sparse_get.argtypes = [C.c_void_p, C.c_int64, _I64P, _I64P, C.POINTER(C.c_double)]
        L.genphi_sparse_get.restype = C.c_int
        L.genphi_sparse_entries.argtypes = [C.c_void_p, C.c_int64, _I64P, _I64P, _F32P]
        L.genphi_sparse_entries.restype = C.c_int64
        L.genphi_sparse_destroy.argtypes = [C.c_void_p]
        L.genphi_sparse_destroy.restype = None
        L.genphi_panel_create.argtypes = [C.c_int64, _I64P, _I64P, _I64P, C.c_int64, _I64P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.genphi_panel_create.restype = C.c_int
        L.genphi_panel_n_steps.argtypes = [C.c_void_p]
        L.genphi_panel_n_steps.restype = C.c_int64
        L.genphi_panel_step_ms.argtypes = [C.c_void_p, C.c_int32]
        L.genphi_panel_step_ms.restype = C.c_double
        L.genphi_panel_step_mode.argtypes = [C.c_void_p, C.c_int32]
        L.genphi_panel_step_mode.restype = C.c_int
        L.genphi_panel_n_probands.argtypes = [C.c_void_p]
        L.genphi_panel_n_probands.restype = C.c_int64
        L.genphi_panel_result_rows.argtypes = [C.c_void_p, _I64P, _I64P]
        L.genphi_panel_result_rows.restype = C.c_int
        L.genphi_panel_exchange_counts.argtypes = [C.c_void_p, C.c_int32, _I64P, _I64P, _I64P]
        L.genphi_panel_exchange_counts.restype = C.c_int
        L.genphi_panel_device_bytes.argtypes = [C.c_void_p]
        L.genphi_panel_device_bytes.restype = C.c_double
        L.genphi_panel_begin.argtypes = [C.c_void_p, C.c_int32]
        L.genphi_panel_begin.restype = C.c_int
        L.genphi_panel_pack.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.genphi_panel_pack.restype = C.c_int
        L.genphi_panel_compute.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.genphi_panel_compute.restype = C.c_int
        L.genphi_panel_pack_on.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.genphi_panel_pack_on.restype = C.c_int
        L.genphi_panel_compute_on.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.genphi_panel_compute_on.restype = C.c_int
        L.genphi_panel_sync.argtypes = [C.c_void_p]
        L.genphi_panel_sync.restype = C.c_int
        L.genphi_panel_result_to_host.argtypes = [C.c_void_p, _F32P]
        L.genphi_panel_result_to_host.restype = C.c_int
        L.genphi_panel_destroy.argtypes = [C.c_void_p]
        L.genphi_panel_destroy.restype = None
        L.genphi_last_error.restype = C.c_char_p
        L.genphi_version.restype = C.c_char_p
        _lib = L
    return _lib


def last_error():
    return lib().genphi_last_error().decode("utf-8", "replace")


def _raise(rc):
    msg = last_error()
    if rc in (GENPHI_ERR_UNKNOWN_ID, GENPHI_ERR_ORDER):
        raise KeyError(msg)                      # the reference raises KeyError for both
    if rc == GENPHI_ERR_DUPLICATE_ID or rc == GENPHI_ERR_ARG:
        raise ValueError(msg)
    if rc == GENPHI_ERR_ALLOC:
        raise MemoryError(msg)
    raise GenphiDeviceError(msg)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def release_cached():
    """Give back what the library keeps between calls (device blocks of released plans, idle streams, pinned staging): genphi_release_cached."""
    lib().genphi_release_cached()


def cached_bytes():
    return int(lib().genphi_cached_bytes())


def genealogy_read(path, sort=True):
    """(ind, father, mother, sex) int64 arrays in rank order, parsed and depth-sorted natively."""
    L = lib()
    n = C.c_int64()
    ptrs = [_I64P() for _ in range(4)]
    rc = L.genphi_genealogy_read(os.fsencode(path), 1 if sort else 0, C.byref(n), *[C.byref(p) for p in ptrs])
    if rc:
        _raise(rc)
    try:
        out = tuple(np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64) for p in ptrs)
    finally:
        for p in ptrs:
            L.genphi_free(p)
    return out


def genealogy_order(ind, father, mother, sex, sort=True):
    """(ind, father, mother, sex) int64 arrays in rank order from a table in memory (genphi_genealogy_order): the checks and the
    stable depth sort of gen.genealogy(dataframe; sort), natively."""
    L = lib()
    ind, father, mother, sex = _i64(ind), _i64(father), _i64(mother), _i64(sex)
    if not (len(ind) == len(father) == len(mother) == len(sex)):
        raise ValueError("ind, father, mother, sex must have the same length")
    n = C.c_int64()
    ptrs = [_I64P() for _ in range(4)]
    rc = L.genphi_genealogy_order(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P),
                                  sex.ctypes.data_as(_I64P), 1 if sort else 0, C.byref(n), *[C.byref(p) for p in ptrs])
    if rc:
        _raise(rc)
    try:
        out = tuple(np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64) for p in ptrs)
    finally:
        for p in ptrs:
            L.genphi_free(p)
    return out


def branching(ind, father, mother, sex, pro=None, ancestors=None):
    """(ind, father, mother, sex) of the pruned pedigree, rank order kept (genphi_branching)."""
    L = lib()
    ind, father, mother, sex = _i64(ind), _i64(father), _i64(mother), _i64(sex)
    pro_a = None if pro is None else _i64(pro)
    anc_a = None if ancestors is None else _i64(ancestors)
    # a zero-length "given" list must stay distinguishable from "not given" (NULL)
    keep = np.zeros(1, dtype=np.int64)
    ptr = lambda a: None if a is None else (a if len(a) else keep).ctypes.data_as(_I64P)  # noqa: E731
    n = C.c_int64()
    ptrs = [_I64P() for _ in range(4)]
    rc = L.genphi_branching(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P),
                            mother.ctypes.data_as(_I64P), sex.ctypes.data_as(_I64P),
                            0 if pro_a is None else len(pro_a), ptr(pro_a),
                            0 if anc_a is None else len(anc_a), ptr(anc_a),
                            C.byref(n), *[C.byref(p) for p in ptrs])
    if rc:
        _raise(rc)
    try:
        out = tuple(np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64) for p in ptrs)
    finally:
        for p in ptrs:
            L.genphi_free(p)
    return out


class PhiPlan:
    """Owns a genphi_plan: levelisation + flat index arrays (host), level matrices (device)."""

    def __init__(self, ind, father, mother, pro_ids, tuning=None):
        """tuning: None = genphi_plan_create (the library's defaults; GENPHI_* environment hooks only under GENPHI_ENV_HOOKS=1), or a dict
        of settings for this plan alone, e.g. {"SPARSE_K": -1} (genphi_plan_create_tuned; {} = the defaults whatever the environment says)."""
        L = lib()
        ind, father, mother, pro_ids = _i64(ind), _i64(father), _i64(mother), _i64(pro_ids)
        h = C.c_void_p()
        args = (len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P), len(pro_ids),
                pro_ids.ctypes.data_as(_I64P))
        if tuning is None:
            rc = L.genphi_plan_create(*args, C.byref(h))
        else:
            t = L.genphi_tuning_create()
            try:
                for k, v in tuning.items():
                    rc = L.genphi_tuning_set(t, str(k).encode(), str(v).encode())
                    if rc:
                        _raise(rc)
                rc = L.genphi_plan_create_tuned(*args, t, C.byref(h))
            finally:
                L.genphi_tuning_destroy(t)
        if rc:
            _raise(rc)
        self._h = h
        self.stats = None

    def close(self):
        if getattr(self, "_h", None):
            lib().genphi_plan_destroy(self._h)
            self._h = None

    def release_device(self):
        """Free the plan's GPU memory; the next compute uploads again (same or another device)."""
        rc = lib().genphi_plan_release_device(self._h)
        if rc:
            _raise(rc)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_probands(self):
        return int(lib().genphi_plan_n_probands(self._h))

    @property
    def device_bytes(self):
        """Device memory the plan holds right now (index arrays, level matrices, row lists, the resident result)."""
        return int(lib().genphi_plan_device_bytes(self._h))

    @property
    def device_bytes_needed(self):
        """Host only: device memory a full-result Float32 sweep of this plan will allocate (genphi_plan_device_bytes_needed)."""
        return int(lib().genphi_plan_device_bytes_needed(self._h))

    @property
    def algorithmic_bytes(self):
        return float(lib().genphi_plan_algorithmic_bytes(self._h))

    def levels(self):
        """(cut_sizes, both_counts): top founders first."""
        nl = C.c_int32()
        cs, bc = _I64P(), _I64P()
        rc = lib().genphi_plan_levels(self._h, C.byref(nl), C.byref(cs), C.byref(bc))
        if rc:
            _raise(rc)
        n = nl.value
        return [int(cs[k]) for k in range(n)], [int(bc[k]) for k in range(max(n - 1, 0))]

    def step_info(self, step):
        """(mode, dragged members, distinct parents of the new members, new x new sub-step mode) of a level step."""
        out = (C.c_int64 * 4)()
        rc = lib().genphi_plan_step_info(self._h, int(step), out)
        if rc:
            _raise(rc)
        return tuple(int(x) for x in out)

    def step_slots(self, step):
        """(flags: 1 = the step writes its cut in place | 2 = it reads a cut stored by slot, slot capacity, first slot and
        reserved slots of the new members) of a level step; zeros unless WIDE steps keep their members in place."""
        out = (C.c_int64 * 4)()
        rc = lib().genphi_plan_step_slots(self._h, int(step), out)
        if rc:
            _raise(rc)
        return tuple(int(x) for x in out)

    def step_walk(self, step):
        """Work lists of SPLIT level step `step` (the hub walk, csrc/planner.h): (desc, seg, run) as int32 arrays of
        shape (rows, 4), (segments + 2, 4), (runs + 1, 4) (terminators included; run = first segment, hub | n0 << 16, its rows)."""
        nr, ns, nu = C.c_int64(), C.c_int64(), C.c_int64()
        rc = lib().genphi_plan_step_walk(self._h, int(step), C.byref(nr), C.byref(ns), C.byref(nu), None, None, None)
        if rc:
            _raise(rc)
        desc = np.zeros((nr.value, 4), np.int32); seg = np.zeros((ns.value + 2, 4), np.int32); run = np.zeros((nu.value + 1, 4), np.int32)
        i32 = C.POINTER(C.c_int32)
        rc = lib().genphi_plan_step_walk(self._h, int(step), None, None, None, desc.ctypes.data_as(i32), seg.ctypes.data_as(i32), run.ctypes.data_as(i32))
        if rc:
            _raise(rc)
        return desc, seg, run

    def set_step_hook(self, fn):
        """fn(step, n_steps) is called right before each level step of a Float32 sweep is handed to the GPU (the reference prints
        its "Running step ..." lines there, src/compute.jl:280-285); None removes the hook."""
        self._hook = STEP_FN(lambda step, n, user: fn(int(step), int(n))) if fn is not None else STEP_FN()
        rc = lib().genphi_plan_set_step_hook(self._h, self._hook, None)
        if rc:
            _raise(rc)

    def step_modes(self):
        """Kernel family per level step: 0 FULL, 1 SPLIT, 2 WIDE."""
        n = len(self.levels()[0]) - 1
        return [int(lib().genphi_plan_step_mode(self._h, k)) for k in range(max(n, 0))]

    def sparse_levels(self):
        """(k, nnz): the last cut the Float32 sweep keeps as lists of its non-zero entries (-1: every level is a dense matrix, or no
        sweep has run yet) and the non-zero entries counted per cut (-1 = not counted), genphi_plan_sparse_levels."""
        k = C.c_int32(-1)
        buf = (C.c_int64 * 16)()
        m = lib().genphi_plan_sparse_levels(self._h, C.byref(k), buf, None, 16)
        return int(k.value), [int(buf[c]) for c in range(m)]

    def sparse_entries(self):
        """List entries ((column, value) pairs, 8 bytes each) every counted cut is stored as; -1 = not counted."""
        buf = (C.c_int64 * 16)()
        m = lib().genphi_plan_sparse_levels(self._h, None, None, buf, 16)
        return [int(buf[c]) for c in range(m)]

    def _opts(self, device, kernel, rows, timing, storage64=False, no_graph=False, no_sparse=False):
        o = GenphiOpts()
        o.device = -1 if device is None else int(device)
        o.kernel = int(kernel)
        o.row_begin, o.row_end = (0, 0) if rows is None else (int(rows[0]), int(rows[1]))
        o.timing = 1 if timing else 0
        o.flags = (GENPHI_FLAG_STORAGE_F64 if storage64 else 0) | (GENPHI_FLAG_NO_GRAPH if no_graph else 0) | (GENPHI_FLAG_NO_SPARSE if no_sparse else 0)
        return o

    def compute_device(self, device=None, kernel=0, rows=None, timing=False, storage64=False, no_graph=False, no_sparse=False):
        """Run all level steps on the GPU; the result stays resident in HBM.  storage64: Float64
        level matrices (the values of the reference's Float64 pairwise recursion).  no_sparse: every level as a dense matrix
        (GENPHI_FLAG_NO_SPARSE; same values)."""
        o = self._opts(device, kernel, rows, timing, storage64, no_graph, no_sparse)
        st = GenphiStats()
        rc = lib().genphi_compute_device(self._h, C.byref(o), C.byref(st))
        if rc:
            _raise(rc)
        self.stats = st
        self._rows = (self.n_probands if rows is None or int(rows[1]) <= 0 else int(rows[1]) - int(rows[0]))
        self._f64 = bool(storage64)
        return st

    def result_device(self):
        """(device pointer, row pitch in floats, first row, number of rows) of the resident result."""
        ptr, ld, r0, nr = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        rc = lib().genphi_result_device(self._h, C.byref(ptr), C.byref(ld), C.byref(r0), C.byref(nr))
        if rc:
            _raise(rc)
        return ptr.value, ld.value, r0.value, nr.value

    def result_to_host(self):
        nr = self._rows if getattr(self, "_f64", False) else self.result_device()[3]
        n = self.n_probands
        out = np.empty((nr, n), dtype=np.float32)
        rc = lib().genphi_result_to_host(self._h, out.ctypes.data_as(_F32P))
        if rc:
            _raise(rc)
        return out

    def result_to_host_f64(self):
        """The resident Float64 result (after compute_device(storage64=True)) as float64 (rows, N)."""
        n = self.n_probands
        out = np.empty((self._resident_rows(), n), dtype=np.float64)
        rc = lib().genphi_result_to_host_f64(self._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc:
            _raise(rc)
        return out

    def _resident_rows(self):
        return self._rows if getattr(self, "_rows", None) is not None else self.n_probands

    def result_sums(self):
        """(sum of all resident entries, sum of their diagonal entries, resident rows), Float64,
        reduced on the device."""
        a, d, nr = C.c_double(), C.c_double(), C.c_int64()
        rc = lib().genphi_result_sums(self._h, C.byref(a), C.byref(d), C.byref(nr))
        if rc:
            _raise(rc)
        return a.value, d.value, nr.value

    def result_entries(self, rows, cols):
        """Phi[rows[k], cols[k]] (0-based proband positions) read from the resident result, Float64."""
        rows, cols = _i64(rows), _i64(cols)
        if rows.shape != cols.shape or rows.ndim != 1:
            raise ValueError("rows and cols must be 1-D and of equal length")
        out = np.empty(len(rows), dtype=np.float64)
        rc = lib().genphi_result_entries(self._h, len(rows), rows.ctypes.data_as(_I64P), cols.ctypes.data_as(_I64P),
                                         out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc:
            _raise(rc)
        return out

    def phi_mean(self):
        """gen.phiMean of the resident (full) result without a device-to-host copy."""
        a, d, nr = self.result_sums()
        n = self.n_probands
        if nr != n:
            raise ValueError("phi_mean needs all rows resident; combine result_sums() of the shards instead")
        return np.float32((a - d) / (n * n - n))

    def compute(self, device=None, kernel=0, rows=None, timing=False, no_sparse=False):
        self.compute_device(device=device, kernel=kernel, rows=rows, timing=timing, no_sparse=no_sparse)
        return self.result_to_host()


def phi_pairs(ind, father, mother, id_i, id_j, device=None):
    """Float64 kinship of each pair (id_i[k], id_j[k]) (genphi_phi_pairs: one Float64 sweep)."""
    L = lib()
    ind, father, mother, id_i, id_j = _i64(ind), _i64(father), _i64(mother), _i64(id_i), _i64(id_j)
    if id_i.shape != id_j.shape or id_i.ndim != 1:
        raise ValueError("id_i and id_j must be 1-D and of equal length")
    out = np.empty(len(id_i), dtype=np.float64)
    rc = L.genphi_phi_pairs(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P),
                            len(id_i), id_i.ctypes.data_as(_I64P), id_j.ctypes.data_as(_I64P),
                            out.ctypes.data_as(C.POINTER(C.c_double)), -1 if device is None else int(device))
    if rc:
        _raise(rc)
    return out


def sparse_schedule(ind, father, mother, pro_ids):
    """Host only: (order IDs, retire_at, wave) of the sweep gen.sparse_phi would run (genphi_sparse_schedule): the order in which
    individuals leave the reference's queue, the processing index at which each is dropped from the live set (-1: a proband), its wave."""
    L = lib()
    ind, father, mother, pro_ids = _i64(ind), _i64(father), _i64(mother), _i64(pro_ids)
    cap = len(ind)
    order, retire, wave = np.zeros(max(cap, 1), np.int64), np.zeros(max(cap, 1), np.int64), np.zeros(max(cap, 1), np.int32)
    n = C.c_int64()
    rc = L.genphi_sparse_schedule(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P), len(pro_ids),
                                  pro_ids.ctypes.data_as(_I64P), cap, order.ctypes.data_as(_I64P), retire.ctypes.data_as(_I64P),
                                  wave.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n))
    if rc:
        _raise(rc)
    n = n.value
    return order[:n].copy(), retire[:n].copy(), wave[:n].copy()


class KinshipMatrix:
    """What gen.sparse_phi returns (src/compute.jl:31-46): kinships of the probands, accessed by IDs."""

    def __init__(self, ind, father, mother, pro_ids, device=None):
        L = lib()
        ind, father, mother, pro_ids = _i64(ind), _i64(father), _i64(mother), _i64(pro_ids)
        h = C.c_void_p()
        rc = L.genphi_sparse_phi(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P),
                                 len(pro_ids), pro_ids.ctypes.data_as(_I64P), -1 if device is None else int(device), C.byref(h))
        if rc:
            _raise(rc)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().genphi_sparse_destroy(self._h)
            self._h = None

    def info(self):
        """(rows, stored entries, Float64 sum of all stored values, Float64 sum of the self kinships)."""
        nr, nz, sa, sd = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        rc = lib().genphi_sparse_info(self._h, C.byref(nr), C.byref(nz), C.byref(sa), C.byref(sd))
        if rc:
            _raise(rc)
        return nr.value, nz.value, sa.value, sd.value

    def stats(self):
        """Measurement of the GPU sweep that built this matrix: dict(n_waves, sweep_ms, algorithmic_bytes, max_active,
        wave_ms, wave_bytes) -- device time from HIP events, bytes = 4 (n_old^2 + n_next^2) per wave."""
        nw, ms, ab, ma = C.c_int32(), C.c_double(), C.c_double(), C.c_int64()
        rc = lib().genphi_sparse_stats(self._h, C.byref(nw), C.byref(ms), C.byref(ab), C.byref(ma), None, None, 0)
        if rc:
            _raise(rc)
        wm, wb = np.zeros(nw.value, np.float32), np.zeros(nw.value, np.float64)
        rc = lib().genphi_sparse_stats(self._h, None, None, None, None, wm.ctypes.data_as(_F32P), wb.ctypes.data_as(C.POINTER(C.c_double)), nw.value)
        if rc:
            _raise(rc)
        return {"n_waves": nw.value, "sweep_ms": ms.value, "algorithmic_bytes": ab.value, "max_active": ma.value, "wave_ms": wm, "wave_bytes": wb}

    def get(self, id1, id2):
        id1, id2 = _i64(np.atleast_1d(id1)), _i64(np.atleast_1d(id2))
        out = np.empty(len(id1), dtype=np.float64)
        rc = lib().genphi_sparse_get(self._h, len(id1), id1.ctypes.data_as(_I64P), id2.ctypes.data_as(_I64P),
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc:
            _raise(rc)
        return out

    def __getitem__(self, ids):
        """phi[ID1, ID2] (getindex, src/compute.jl:36-40); KeyError for an ID that is not a proband."""
        return float(self.get([ids[0]], [ids[1]])[0])

    def entries(self):
        n = lib().genphi_sparse_entries(self._h, 0, None, None, None)
        r, c, v = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float32)
        lib().genphi_sparse_entries(self._h, n, r.ctypes.data_as(_I64P), c.ctypes.data_as(_I64P), v.ctypes.data_as(_F32P))
        return r, c, v

    def __repr__(self):
        nr, nz, _, _ = self.info()
        return f"{nr}×{nr} KinshipMatrix with {nz} stored entries."          # Base.show, src/compute.jl:42-46


class PanelPlan:
    """Storage-sharded gen.phi of one rank (column panels + exchange; include/genphi.h, csrc/panel_phi.hip)."""

    def __init__(self, ind, father, mother, pro_ids, rank, world):
        L = lib()
        ind, father, mother, pro_ids = _i64(ind), _i64(father), _i64(mother), _i64(pro_ids)
        h = C.c_void_p()
        rc = L.genphi_panel_create(len(ind), ind.ctypes.data_as(_I64P), father.ctypes.data_as(_I64P), mother.ctypes.data_as(_I64P),
                                   len(pro_ids), pro_ids.ctypes.data_as(_I64P), int(rank), int(world), C.byref(h))
        if rc:
            _raise(rc)
        self._h, self.rank, self.world = h, int(rank), int(world)

    def close(self):
        if getattr(self, "_h", None):
            lib().genphi_panel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_steps(self):
        return int(lib().genphi_panel_n_steps(self._h))

    @property
    def n_probands(self):
        return int(lib().genphi_panel_n_probands(self._h))

    @property
    def device_bytes(self):
        return float(lib().genphi_panel_device_bytes(self._h))

    def result_rows(self):
        a, b = C.c_int64(), C.c_int64()
        rc = lib().genphi_panel_result_rows(self._h, C.byref(a), C.byref(b))
        if rc:
            _raise(rc)
        return a.value, b.value

    def step_modes(self):
        """Kernel family of every level step on this rank's panel: 0 FULL, 1 SPLIT (the row kernels of the
        dense path on the local columns), 2 per-entry kernel (panel rows too long for LDS, or GENPHI_PANEL_NAIVE)."""
        return [int(lib().genphi_panel_step_mode(self._h, k)) for k in range(self.n_steps)]

    def step_ms(self):
        """Device time (ms) of every level step's kernels in the last sweep (unpack of the received columns + level kernel)."""
        return [float(lib().genphi_panel_step_ms(self._h, k)) for k in range(self.n_steps)]

    def exchange_counts(self, step):
        """(columns to send per rank, columns to receive per rank, floats per column) before level step `step`."""
        s, r = np.zeros(self.world, np.int64), np.zeros(self.world, np.int64)
        f = C.c_int64()
        rc = lib().genphi_panel_exchange_counts(self._h, int(step), s.ctypes.data_as(_I64P), r.ctypes.data_as(_I64P), C.byref(f))
        if rc:
            _raise(rc)
        return s, r, f.value

    def begin(self, device=None):
        rc = lib().genphi_panel_begin(self._h, -1 if device is None else int(device))
        if rc:
            _raise(rc)

    def pack(self, step, d_send_ptr):
        rc = lib().genphi_panel_pack(self._h, int(step), C.c_void_p(d_send_ptr))
        if rc:
            _raise(rc)

    def compute(self, step, d_recv_ptr):
        rc = lib().genphi_panel_compute(self._h, int(step), C.c_void_p(d_recv_ptr))
        if rc:
            _raise(rc)

    def pack_on(self, step, d_send_ptr, stream):
        """pack, ordered by streams: `stream` (a raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) waits for the packed columns."""
        rc = lib().genphi_panel_pack_on(self._h, int(step), C.c_void_p(d_send_ptr), C.c_void_p(stream))
        if rc:
            _raise(rc)

    def compute_on(self, step, d_recv_ptr, stream):
        """compute, ordered by streams: the panel's stream waits for what `stream` holds (the collective), nothing blocks the host."""
        rc = lib().genphi_panel_compute_on(self._h, int(step), C.c_void_p(d_recv_ptr), C.c_void_p(stream))
        if rc:
            _raise(rc)

    def sync(self):
        rc = lib().genphi_panel_sync(self._h)
        if rc:
            _raise(rc)

    def result_to_host(self):
        r0, nr = self.result_rows()
        out = np.empty((nr, self.n_probands), dtype=np.float32)
        rc = lib().genphi_panel_result_to_host(self._h, out.ctypes.data_as(_F32P))
        if rc:
            _raise(rc)
        return out
