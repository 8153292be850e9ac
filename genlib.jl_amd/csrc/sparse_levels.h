// sparse_levels.h -- internal interface between the gen.phi sweep (genphi_hip.hip) and the zero-aware leading
// levels (sparse_levels.hip).
//
// The level matrices of the first steps below the founders are almost empty: Psi_0 = 1/2 I, and an entry of cut c is
// non-zero only where the two members share an ancestor among the members above (cfg4: 0.02 % / 0.06 % / 0.25 % / 0.95 % /
// 3.6 % / 13 % of the entries of cuts 1..6; genea140: 0.2 % of cut 6, 1.2 % of cut 8).  The reference's second algorithm
// stores only `coefficient > 0.` for the same reason (src/compute.jl:391-394).  Here the leading cuts 0..k of a sweep are
// kept as ROW LISTS in HBM -- per row the (column, value) pairs of its non-zero entries, columns ascending -- and a level
// step is the sparse product A Psi A^T evaluated row by row (src/compute.jl:105-158 entry by entry gives the same sums):
//
//   steps 0 .. k-1   sparse_step_kernel       row lists of cut s  ->  row lists of cut s+1
//   step  k          sparse_dense_kernel      row lists of cut k  ->  the dense Float32 matrix of cut k+1, exactly what a
//                                             FULL / SPLIT row kernel would have written (rows [0, n) x columns [0, width)
//                                             and the all-zero "none" row n), so that step k+1 is an ordinary dense step
//
// Exactness (why any order of summation is the reference's value).  By induction over the cuts every entry of cut c is a
// multiple of 2^-(2c+1) in [0, 1): Psi_0 holds 0 and 1/2; an entry of cut c+1 is a quarter / a half / a copy of a sum of
// entries of cut c, or 1/2 + Psi[f][m]/2 (src/compute.jl:148-154).  For c <= 11 such a value is an integer below 2^23 in
// units of 2^-(2c+1): every partial sum is exact in 32-bit integer arithmetic, the conversion to Float32 is exact, and
// the reference's Float64 sum followed by RN32 (src/compute.jl:107,296) yields the same bits whatever its grouping.  The
// kernels therefore accumulate in integer units with LDS atomics, and sparse cuts end at cut 11 at the latest.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "planner.h"

namespace genphi {

constexpr int kSparseMaxLevel = 11;           // last cut whose entries are exact in 32-bit integer units (see above)
constexpr int kSparseMaxMembers = 65535;      // columns are kept as 16-bit words while a row is assembled

struct SparseStepDev {                        // device index arrays of level step s (per member of cut s+1)
    const int *srcA, *srcB, *ord;
    const int *work;                          // the planner's row order of the step (rows that share sources adjacent), or nullptr
};

struct SparseTuning {
    int max_permille = 200;    // the calibration run stops at the first cut with more than this share (in 1/1000) of non-zero entries; which
                               // of the cuts before it is the last sparse one is a matter of estimated times (sparse_levels.hip)
    int force_k = -2;          // test / A-B hook: -2 = by calibration; -1 = never sparse; k >= 0: cuts 0..k sparse whatever the counts say
                               // (clamped to what is eligible)
    int min_cut = 1536;        // ... and only when some cut of the sparse run has at least this many members (narrower levels are launch-bound)
    int chunk_cols = 12288;    // columns per workgroup of the sparse -> dense step (cfg4, same box: 0.62 ms at 8192, 0.56 at 12288, 0.79 at 4096)
    int long_batch = 4;        // list entries a thread of a four-wavefront row keeps in flight (4; 8 = A/B hook: measured SLOWER -- genea140's
                               // largest list steps +12..24 %, cfg3s +22 %, cfg4 the same: r05_ab_sparse_list_step_entries_in_flight_4_vs_8_*.out)
    int first_entries = 1 << 24;   // entries each row-list arena starts with (128 MB: genea140's and cfg3's lists fit, 11 M and 8.5 M entries); the calibration run enlarges them where a cut needs more (test hook: small values)
    int classes = -1;          // a launch per class of row lengths: -1 = where the rows of a cut differ much in length, 1 / 0 = always / never (A/B hook)
};

struct SparseLevels;           // opaque (sparse_levels.hip)

// Number of leading level steps that may run on row lists (a property of the plan alone): steps 0..S-1 are FULL / SPLIT
// steps that neither read nor write by slot, write cuts <= kSparseMaxLevel with < 65535 members, and are not the proband step.
int sparse_eligible_steps(const Plan &plan);

// Builds the children lists of the first S steps (host), uploads them and allocates the row-list arenas.  Returns nullptr
// and a message on failure.  `dev[s]` = device index arrays of step s (owned by the caller, alive as long as the handle).
SparseLevels *sparse_levels_create(const Plan &plan, int S, const std::vector<SparseStepDev> &dev, const SparseTuning &tun,
                                   hipStream_t stream, std::string &err);
void sparse_levels_destroy(SparseLevels *sl);

// One calibrating run of the sparse steps on `stream` (synchronises after every step): counts the non-zero entries of
// every cut and fixes k, the last sparse cut (-1: the sweep stays dense).  Values do not depend on k.
int sparse_levels_calibrate(SparseLevels *sl, hipStream_t stream, std::string &err);
int sparse_levels_k(const SparseLevels *sl);                  // last sparse cut, or -1
// step s of the sweep: s < k enqueues the row-list step (s == 0 also the lists of 1/2 I and the counters' reset) ...
int sparse_levels_enqueue_step(SparseLevels *sl, int s, hipStream_t stream, std::string &err);
// ... s == k the step that writes cut k+1 as a dense matrix: out = (n + 1) rows of pitch ld, columns [0, width) written; Float32 or
// (f64) Float64 entries; compact: rows and columns in the cut's storage order whatever the plan says about slots (the Float64 sweep)
int sparse_levels_enqueue_dense(SparseLevels *sl, void *out, bool f64, bool compact, long long ld, long long width, hipStream_t stream,
                                std::string &err);
// ... and after the last sparse step of a sweep: the error flags of the row-list steps are copied to the host (asynchronously); once the
// stream is synchronised sparse_levels_flags_ok says whether every row had the length the plan recorded for it
int sparse_levels_enqueue_flags(SparseLevels *sl, hipStream_t stream, std::string &err);
bool sparse_levels_flags_ok(const SparseLevels *sl);
// diagnostics: per cut c <= k (+1 from the calibration run) the number of non-zero entries (-1 unknown) and the longest row
int sparse_levels_counts(const SparseLevels *sl, int cap, long long *nnz, long long *entries, int *max_row);
double sparse_levels_device_bytes(const SparseLevels *sl);

}  // namespace genphi
