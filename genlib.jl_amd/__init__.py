"""genlib.jl_amd -- MI355X-native drop-in for GenLib.jl's dense kinship matrix `gen.phi`.

Host-side mirror of the reference interface for this ONE path (the reference's host
language, Julia, is not available in the build image; the Julia shim that a GenLib.jl
maintainer would add is in julia/GenLibAMD.jl and INTEGRATION.md).  Same names, argument
meaning, printed lines and error behaviour as the reference:

    import genlib_jl_amd as gen            # loader shim at the repo root
    ped = gen.genealogy(gen.geneaJi)       # src/create.jl:161-189 (+ depth sort :196-254)
    gen.pro(ped); gen.founder(ped)         # src/identify.jl:35-39, :15-19
    phi = gen.phi(ped, verbose=True)       # src/compute.jl:233-304 -> float32 (N, N)
    gen.phi(ped[1], ped[2])                # src/compute.jl:66-95: pairwise, Float64 (one Float64 GPU sweep)
    gen.phiMean(phi)                       # src/compute.jl:454-459 (PhiPlan.phi_mean(): on the device)
    K = gen.sparse_phi(ped); K[1, 2]       # src/compute.jl:321-447, :31-46; gen.phiMean(K) :467-472
    gen.f(ped, [1])                        # src/compute.jl:500-511, from one Float64 GPU sweep over the parents
    gen.branching(ped, pro=[1])            # src/extract.jl:65-186, native pruning (csrc/loader.cpp)

All kinship arithmetic runs in hand-written HIP kernels behind the C-ABI in
include/genphi.h (csrc/genphi_hip.hip); there is no CPU fallback.
"""
import os
from collections import OrderedDict

import numpy as np

from . import _capi
from ._capi import PhiPlan, KinshipMatrix, GenphiDeviceError, GenphiLibraryMissing  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

# bundled example pedigrees (the reference exports the same names, src/GenLib.jl:27,43)
geneaJi = os.path.join(_HERE, "data", "geneaJi.csv")
genea140 = os.path.join(_HERE, "data", "genea140.csv")


class Individual:
    """What `pedigree[ID]` returns (the reference's immutable Individual, src/create.jl:39-46):
    enough of it for the path -- ID, rank, and the parents as handles (None = unknown)."""

    def __init__(self, pedigree, ID, pos):
        self.pedigree, self.ID, self.rank = pedigree, ID, pos + 1

    def _parent(self, arr):
        pid = int(arr[self.rank - 1])
        return None if pid == 0 else self.pedigree[pid]

    @property
    def father(self):
        return self._parent(self.pedigree.father)

    @property
    def mother(self):
        return self._parent(self.pedigree.mother)

    @property
    def sex(self):
        return int(self.pedigree.sex[self.rank - 1])

    def __repr__(self):
        return f"Individual({self.ID})"


class Pedigree:
    """Rank-ordered pedigree: what `gen.genealogy` returns (src/create.jl:60-74, :234-254).

    Arrays are in rank order (parents before children); `rank` of ind[k] is k + 1.
    Parent id 0 = unknown.
    """

    def __init__(self, ind, father, mother, sex):
        # (read-only, like the reference's immutable Individual structs, src/create.jl:39-46: gen.phi keeps plans per pedigree)
        self.ind, self.father, self.mother, self.sex = (np.array(a, dtype=np.int64, order="C") for a in (ind, father, mother, sex))
        for a in (self.ind, self.father, self.mother, self.sex):
            a.setflags(write=False)
        self._index = None
        self._plans = OrderedDict()          # gen.phi's plans for this pedigree, least recently used first (see _plan_for)

    def __len__(self):
        return len(self.ind)

    def __getitem__(self, ID):
        """pedigree[ID] -> Individual handle (src/create.jl:70); KeyError on an unknown ID."""
        return Individual(self, int(ID), int(self.positions([ID])[0]))

    def _idx(self):
        if self._index is None:
            order = np.argsort(self.ind, kind="stable")
            self._index = (self.ind[order], order)
        return self._index

    def positions(self, ids):
        """Rank positions of the given IDs; KeyError on an unknown ID (OrderedDict lookup)."""
        ids = np.asarray(ids, dtype=np.int64)
        keys, order = self._idx()
        k = np.searchsorted(keys, ids)
        k = np.clip(k, 0, len(keys) - 1) if len(keys) else k
        bad = (len(keys) == 0) | (keys[k] != ids) if len(ids) else np.zeros(0, bool)
        if np.any(bad):
            raise KeyError(int(ids[np.argmax(bad)]))
        return order[k]

    def __contains__(self, ID):
        keys, _ = self._idx()
        k = np.searchsorted(keys, ID)
        return k < len(keys) and keys[k] == ID

    def __repr__(self):
        return f"Pedigree({len(self)} individuals)"


def _read_table(source):
    if isinstance(source, (str, os.PathLike)):
        # src/create.jl:161-189: header skipped, whitespace-separated ind father mother sex
        rows = np.loadtxt(source, dtype=np.int64, skiprows=1, ndmin=2)
        return rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    # DataFrame-like / mapping with the reference's column names (src/create.jl:131-146)
    get = (lambda k: np.asarray(source[k])) if not hasattr(source, "to_dict") else (lambda k: source[k].to_numpy())
    return tuple(np.asarray(get(k), dtype=np.int64) for k in ("ind", "father", "mother", "sex"))


def genealogy(source, sort=True):
    """gen.genealogy(filename | dataframe; sort=true)  (src/create.jl:131-189).

    With sort=True individuals are ordered by maximum ancestral depth (founders = 1) with a
    STABLE sort over input order (src/create.jl:196-227); the position in that order is the
    `rank` the kinship recursion branches on.  With sort=False the input order is kept and a
    parent listed after its child raises KeyError, as `_finalize_pedigree` does.
    """
    if isinstance(source, (str, os.PathLike)):
        # files go through the native loader (parse + depth sort in C++, csrc/loader.cpp)
        return Pedigree(*_capi.genealogy_read(source, sort=sort))
    # tables in memory: the same checks and the stable depth sort natively (genphi_genealogy_order, csrc/loader.cpp; the numpy form of
    # this took 1.1 s at 1e6 individuals, ten times the planning of the whole sweep)
    ind, father, mother, sex = (np.ascontiguousarray(a, dtype=np.int64) for a in _read_table(source))
    return Pedigree(*_capi.genealogy_order(ind, father, mother, sex, sort=sort))


def pro(pedigree):
    """gen.pro: IDs of individuals without children, ascending (src/identify.jl:35-39)."""
    ind, fa, mo = pedigree.ind, pedigree.father, pedigree.mother
    if len(ind) == 0:
        return ind.copy()
    lo, hi = int(ind.min()), int(ind.max())
    if lo > 0 and hi < 64 * len(ind) + (1 << 20) and int(min(fa.min(), mo.min())) >= 0 and int(max(fa.max(), mo.max())) <= hi:
        # IDs in a moderate range (genea140: 41,523 IDs up to 900,506): one flag byte per ID value instead of two sorts
        # (3-5 ms of a 12 ms gen.phi(genea140) call)
        is_parent = np.zeros(hi + 1, dtype=bool)
        is_parent[fa] = True
        is_parent[mo] = True                                 # (ID 0 = unknown parent: flagged, never looked up since lo > 0)
        return np.sort(ind[~is_parent[ind]])
    parents = np.union1d(fa, mo)
    return np.sort(ind[~np.isin(ind, parents)])


def founder(pedigree):
    """gen.founder: IDs with neither parent known, ascending (src/identify.jl:15-19)."""
    return np.sort(pedigree.ind[(pedigree.father == 0) & (pedigree.mother == 0)])


def plan(pedigree, probandIDs=None, tuning=None):
    """Levelise `pedigree` for `probandIDs` (host only; no GPU needed).  tuning: a dict of settings for this plan (PhiPlan)."""
    probandIDs = pro(pedigree) if probandIDs is None else np.asarray(probandIDs, dtype=np.int64)
    return PhiPlan(pedigree.ind, pedigree.father, pedigree.mother, probandIDs, tuning=tuning)


# gen.phi keeps the plans of its last calls per pedigree: the host prologue of the reference's phi (src/compute.jl:236-262:
# levelisation, cut sets, index copy) depends on (pedigree, probandIDs) alone, so a repeated call -- another subset, then the first one
# again; a parameter sweep -- skips planning, upload and the calibration of the sparse levels and pays the sweep and the copy.
PLAN_CACHE_ENTRIES = 4                 # per pedigree; 0 disables (also GENPHI_PLAN_CACHE=0)
PLAN_CACHE_DEVICE_BYTES = 2 << 30      # a plan that holds more device memory than this is released at the end of its call


def _plan_for(pedigree, probandIDs, device):
    """(plan, keep): the cached plan of this call, or a new one."""
    probandIDs = pro(pedigree) if probandIDs is None else np.ascontiguousarray(probandIDs, dtype=np.int64)
    entries = 0 if os.environ.get("GENPHI_PLAN_CACHE") == "0" else PLAN_CACHE_ENTRIES
    cache = getattr(pedigree, "_plans", None)
    if entries <= 0 or cache is None:
        return PhiPlan(pedigree.ind, pedigree.father, pedigree.mother, probandIDs), None
    # (the library reads its GENPHI_* hooks when a plan is created: a plan made under other settings is another plan)
    key = (hash(probandIDs.tobytes()), len(probandIDs), device, tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("GENPHI_"))))
    hit = cache.get(key)
    if hit is not None and np.array_equal(hit[1], probandIDs):
        cache.move_to_end(key)
        return hit[0], key
    return PhiPlan(pedigree.ind, pedigree.father, pedigree.mother, probandIDs), key


def _keep_plan(pedigree, key, pl, probandIDs):
    cache = pedigree._plans
    if key in cache and cache[key][0] is pl:
        return True
    if pl.device_bytes > PLAN_CACHE_DEVICE_BYTES:
        return False
    old = cache.pop(key, None)
    if old is not None:
        old[0].close()
    cache[key] = (pl, np.array(probandIDs, dtype=np.int64))
    while len(cache) > PLAN_CACHE_ENTRIES:
        cache.popitem(last=False)[1][0].close()
    return True


def phi(pedigree, probandIDs=None, verbose=False, compute=True, device=None, kernel=0):
    """gen.phi(pedigree, probandIDs = pro(pedigree); verbose=false, compute=true), or the pairwise
    method gen.phi(individual_i, individual_j) (src/compute.jl:66-95) when called with two
    `pedigree[ID]` handles: the Float64 kinship of the pair, from one Float64 level sweep on the GPU
    (the reference's recursion is un-memoised and exponential on inbred pedigrees); bit-identical to
    the recursion while kinships are exactly representable in Float64 (pedigrees less than ~26
    generations deep), within 1e-15 relative beyond.

    Returns the square float32 matrix of pairwise kinship coefficients between probands
    (rows/columns in `probandIDs` order, duplicates collapsed), or None when compute=False.
    Prints the reference's cut-vertex lines (src/compute.jl:257-260, :281-284).
    Raises KeyError for an unknown proband ID.  Runs on the GPU (no CPU fallback).
    """
    if isinstance(pedigree, Individual):
        a, b = pedigree, probandIDs
        if not isinstance(b, Individual) or b.pedigree is not a.pedigree:
            raise TypeError("gen.phi(individual_i, individual_j) takes two individuals of one pedigree")
        ped = a.pedigree
        return float(_capi.phi_pairs(ped.ind, ped.father, ped.mother, [a.ID], [b.ID], device=device)[0])
    probandIDs = pro(pedigree) if probandIDs is None else np.ascontiguousarray(probandIDs, dtype=np.int64)
    pl, key = _plan_for(pedigree, probandIDs, device)
    keep = False
    try:
        if verbose or not compute:
            sizes, both = pl.levels()
            nsteps = max(len(sizes) - 1, 0)
            for i in range(nsteps):
                print(f"Step {i + 1} of {nsteps}: {sizes[i]} founders, {sizes[i + 1]} probands, {both[i]} both.")
        if not compute:
            return None
        if verbose:
            # the reference prints these inside its level loop (src/compute.jl:280-285): the library calls back right before it
            # hands each level step to the GPU
            pl.set_step_hook(lambda k, n: print(f"Running step {k + 1} of {n} ({sizes[k]} founders, {sizes[k + 1]} probands, {both[k]} both).", flush=True))
        try:
            out = pl.compute(device=device, kernel=kernel)
        finally:
            if verbose:
                pl.set_step_hook(None)               # (a hook left set would keep the plan off its captured graph)
        keep = key is not None and _keep_plan(pedigree, key, pl, probandIDs)
        return out
    finally:
        if not keep:
            if key is not None and key in pedigree._plans and pedigree._plans[key][0] is pl:
                del pedigree._plans[key]             # (a cached plan whose call failed)
            pl.close()


def f(pedigree, IDs, device=None):
    """gen.f(pedigree, IDs) (src/compute.jl:500-511): inbreeding coefficients (Float32 vector).

    F(x) = kinship of x's parents, 0 if a parent is unknown.  The reference evaluates each one
    with the un-memoised Float64 pairwise recursion (:66-95), exponential on deep inbred
    pedigrees, and rounds the result to Float32 once.  Here ONE level sweep over the set of
    parents runs on the GPU with Float64 level matrices (GENPHI_FLAG_STORAGE_F64), the
    (father, mother) entries are read back (`genphi_result_entries`) and rounded to Float32 once:
    the same values bit for bit while kinships are exactly representable in Float64 (pedigrees
    less than ~26 generations deep; beyond, within 1e-15 relative before the rounding).
    test/runtests.jl:47-48: f(ped, [1]) == [0.18359375], f(ped, [17]) == [0.].
    """
    IDs = np.asarray(IDs, dtype=np.int64)
    pos = pedigree.positions(IDs)                       # KeyError on an unknown ID
    fa, mo = pedigree.father[pos], pedigree.mother[pos]
    out = np.zeros(len(IDs), dtype=np.float32)
    both = (fa != 0) & (mo != 0)
    if not np.any(both):
        return out
    parents = np.unique(np.concatenate([fa[both], mo[both]]))     # sorted, like a proband list
    pl, key = _plan_for(pedigree, parents, device)                # (kept per pedigree like gen.phi's plans: f over the same IDs again pays the sweep)
    keep = False
    try:
        pl.compute_device(device=device, storage64=True)
        out[both] = pl.result_entries(np.searchsorted(parents, fa[both]), np.searchsorted(parents, mo[both])).astype(np.float32)
        keep = key is not None and _keep_plan(pedigree, key, pl, parents)
    finally:
        if not keep:
            if key is not None and key in pedigree._plans and pedigree._plans[key][0] is pl:
                del pedigree._plans[key]
            pl.close()
    return out


def branching(pedigree, pro=None, ancestors=None):
    """gen.branching(pedigree; pro=nothing, ancestors=nothing) (src/extract.jl:65-186).

    Pedigree of the individuals on the paths between the selected probands and ancestors
    (host-side pruning before gen.phi; native, csrc/loader.cpp).  KeyError on an unknown ID.
    """
    return Pedigree(*_capi.branching(pedigree.ind, pedigree.father, pedigree.mother, pedigree.sex,
                                     pro=pro, ancestors=ancestors))


def sparse_phi(pedigree, probandIDs=None, device=None):
    """gen.sparse_phi(pedigree, probandIDs = pro(pedigree)) (src/compute.jl:321-447): the reference's
    queue-driven kinship algorithm, returning a KinshipMatrix indexed by proband IDs (`K[1, 2]`,
    `repr(K)` = the reference's `show` line, `gen.phiMean(K)`).  Computed on the GPU one depth at a
    time on a dense active matrix (csrc/sparse_phi.hip); values, lookup behaviour and the number of
    stored entries are the reference's.  KeyError for an unknown proband ID."""
    probandIDs = pro(pedigree) if probandIDs is None else np.asarray(probandIDs, dtype=np.int64)
    return KinshipMatrix(pedigree.ind, pedigree.father, pedigree.mother, probandIDs, device=device)


def phiMean(phi_matrix):
    """gen.phiMean(::Matrix{Float32}) (src/compute.jl:454-459): mean off-diagonal kinship of a host
    matrix, accumulated in float32 like the reference.  numpy's float32 pairwise summation blocks
    differently from Julia's `sum`, so on large matrices the last bits can differ (about 1 ulp of
    Float32; exact whenever the sums are exact, e.g. 0.171875 on geneaJi, test/runtests.jl:53).
    `PhiPlan.phi_mean()` reduces the RESIDENT matrix on the device instead (Float64 accumulation,
    one rounding; no 40 GB device-to-host copy at N = 1e5)."""
    if isinstance(phi_matrix, KinshipMatrix):            # phiMean(::KinshipMatrix), src/compute.jl:467-472
        nr, _, total, diagonal = phi_matrix.info()
        return np.float32((total - diagonal) / (nr * (nr - 1) / 2))
    m = np.asarray(phi_matrix, dtype=np.float32)
    total = np.float32(m.sum(dtype=np.float32))
    diagonal = np.float32(np.diagonal(m).sum(dtype=np.float32))
    total = np.float32(total - diagonal)
    return np.float32(total / np.float32(m.size - m.shape[0]))
