/*
 * genphi.h -- C-ABI of the MI355X-native gen.phi hot path (dense kinship matrix).
 *
 * Drop-in boundary.  The reference (GPhMorin/GenLib.jl v0.1.4) has no FFI: its boundary is
 * the Julia method
 *     phi(pedigree::Pedigree, probandIDs::Vector{Int} = pro(pedigree);
 *         verbose::Bool = false, compute::Bool = true)          src/compute.jl:233-304
 * A Julia shim with that exact signature (genlib.jl_amd/julia/GenLibAMD.jl, see
 * INTEGRATION.md) flattens the pedigree the way genout does (src/output.jl:24-29, kept at
 * 64 bit) and `ccall`s the entry points below; tests and bench.py bind the same symbols
 * through ctypes.  Plain pointers and sizes only; nothing throws across this boundary.
 *
 * Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - all arrays are caller-owned; the library copies what it needs and keeps no caller
 *     pointer after a call returns (Julia: GC.@preserve for the duration of the ccall);
 *   - individuals are passed in RANK ORDER (iteration order of the reference's Pedigree,
 *     parents before children, src/create.jl:234-254); parent id 0 = unknown;
 *   - calls on one plan are blocking and must be serialised by the caller; distinct plans
 *     may be used concurrently; HIP streams are private to the plan;
 *   - every function returning int returns GENPHI_OK (0) or an error code; the message is
 *     available from genphi_last_error() (thread-local).
 */
#ifndef GENPHI_H
#define GENPHI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENPHI_OK               0
#define GENPHI_ERR_UNKNOWN_ID   1   /* reference: KeyError (OrderedDict lookup, src/create.jl:70 via src/compute.jl:196) */
#define GENPHI_ERR_ORDER        2   /* parent listed after its child: KeyError in _finalize_pedigree, src/create.jl:240-241 */
#define GENPHI_ERR_DUPLICATE_ID 3
#define GENPHI_ERR_ALLOC        4
#define GENPHI_ERR_DEVICE       5   /* no usable GPU / HIP failure: the product has NO CPU fallback */
#define GENPHI_ERR_ARG          6

typedef struct genphi_plan genphi_plan;

typedef struct genphi_opts {
    int32_t device;        /* HIP device ordinal; -1 = current device                      */
    int32_t kernel;        /* 0 = default (LDS-staged rows); 1 = naive per-entry gather     */
    int64_t row_begin;     /* final-level row shard [row_begin,row_end) in proband order;   */
    int64_t row_end;       /*   row_end <= 0 means "all rows" (multi-GPU: one shard/rank)   */
    int32_t timing;        /* !=0: record per-level HIP-event timings into genphi_stats     */
    int32_t flags;         /* GENPHI_FLAG_* below, or-ed (0 = defaults)                      */
} genphi_opts;

#define GENPHI_FLAG_NO_GRAPH     1   /* never replay the sweep from a captured hipGraph                         */
#define GENPHI_FLAG_STORAGE_F64  2   /* Float64 level matrices: the values of the reference's Float64 pairwise
                                        recursion (src/compute.jl:66-95) instead of gen.phi's Float32-per-level
                                        matrices; for gen.f / pairwise queries (per-entry kernel, small sets)   */

#define GENPHI_FLAG_NO_SPARSE     4   /* every level as a dense matrix: the leading cuts, whose matrices are almost empty, are
                                        otherwise kept as lists of their non-zero entries (the idea of the reference's second
                                        algorithm, src/compute.jl:391-394; genphi_plan_sparse_levels).  Same values either way. */

#define GENPHI_MAX_STAT_LEVELS 1024
typedef struct genphi_stats {
    int32_t n_steps;                 /* level steps run (L-1)                               */
    int32_t timed;                   /* 1 if the ms fields below were measured              */
    double  total_ms;                /* first level kernel start -> last kernel end (HIP events on the plan's stream) */
    double  final_ms;                /* the last level step (+ proband-order pass) alone    */
    double  perm_ms;                 /* of which: the proband-order column pass (0 if none)  */
    double  algorithmic_bytes;       /* 4 * sum_k (n_k^2 + n_{k+1}^2), SURVEY.md 8(d)       */
    int64_t max_cut;                 /* largest cut size                                    */
    float   level_ms[GENPHI_MAX_STAT_LEVELS]; /* per level step (first n_steps entries)     */
    int64_t level_rows[GENPHI_MAX_STAT_LEVELS]; /* output rows each level step computed (a row shard
                                        computes, above the last level, only the rows it descends from) */
} genphi_stats;

/* Replaces the host prologue of phi(): levelisation by parent steps and the cut sets
 * (src/compute.jl:236-251, helper _previous_generation :193-207), plus the index copy
 * (_index_pedigree :165-186, founder_index assignment :287-289) in flat, device-ready form.
 * Pure host work: succeeds without a GPU.
 *   n_ind, ind/father/mother : the pedigree in rank order (0 = unknown parent)
 *   n_pro, pro_ids           : probandIDs (duplicates collapse, as `∩` does at :251)        */
int genphi_plan_create(int64_t n_ind, const int64_t *ind, const int64_t *father,
                       const int64_t *mother, int64_t n_pro, const int64_t *pro_ids,
                       genphi_plan **out);

/* Tuning, A/B and test settings of a plan without the environment.  The library has ~45 knobs (README.md, "Tuning hooks": LDS budget,
 * kernel families, in-place runs, sparse cuts, copy threads ...); none is needed in production.  They can be given as GENPHI_* environment
 * variables -- read ONLY when GENPHI_ENV_HOOKS=1 is set too, so that a library loaded into somebody's process never changes kernels on
 * ambient variables -- or, per plan, through a genphi_tuning: genphi_tuning_set(t, "SPARSE_K", "-1") (the hook's name with or without
 * the GENPHI_ prefix; unknown name -> GENPHI_ERR_ARG).  genphi_plan_create_tuned(..., NULL, ...) = the defaults whatever the
 * environment says.  The plan copies what it needs; the tuning may be destroyed right after.  No reference counterpart.        */
typedef struct genphi_tuning genphi_tuning;
genphi_tuning *genphi_tuning_create(void);
int genphi_tuning_set(genphi_tuning *t, const char *name, const char *value);
void genphi_tuning_destroy(genphi_tuning *t);
int genphi_plan_create_tuned(int64_t n_ind, const int64_t *ind, const int64_t *father,
                             const int64_t *mother, int64_t n_pro, const int64_t *pro_ids,
                             const genphi_tuning *tuning, genphi_plan **out);

/* Level description for the "Step i of n: a founders, b probands, c both." lines
 * (src/compute.jl:253-262 and :280-285).  cut_sizes has *n_levels entries (top founders
 * first, probands last), both_counts has *n_levels-1.  Pointers stay valid until
 * genphi_plan_destroy.                                                                      */
int genphi_plan_levels(const genphi_plan *plan, int32_t *n_levels,
                       const int64_t **cut_sizes, const int64_t **both_counts);

/* Number of distinct probands N (rows/columns of the result, proband first-occurrence order). */
int64_t genphi_plan_n_probands(const genphi_plan *plan);

/* Kernel family the planner chose for level step `step` (0-based, < n_levels-1):
 * 0 = FULL (both source rows of an output row in LDS), 1 = SPLIT (one row at a time),
 * 2 = WIDE (a source row does not fit in LDS: the level is assembled block by block from
 * streaming passes); -1 on a bad argument.  Diagnostic only (tests assert every variant is covered). */
int genphi_plan_step_mode(const genphi_plan *plan, int32_t step);

/* More of the same: info[0] = mode, info[1] = members dragged from the previous cut, info[2] =
 * distinct parents of the new members (WIDE steps, else 0), info[3] = how a WIDE step computes its
 * new x new block: 0 / 1 = a FULL / SPLIT sub-step on the compacted parent matrix, 3 = per-entry
 * kernel (parents too wide as well); -1 for the other modes.                                    */
int genphi_plan_step_info(const genphi_plan *plan, int32_t step, int64_t *info);

/* Diagnostic: how WIDE level step `step` stores its cuts in the Float32 sweep (persistent slots, csrc/planner.h LevelStep::stay).
 * info[0] = 1 when the step writes its cut IN PLACE (members keep their row / column slot in one matrix: only the new rows and
 * columns are written, the dragged x dragged block -- src/compute.jl:108-110 -- is not copied) | 2 when it reads a cut stored by
 * slot | 4 when the new members' slots are one stretch (their new x new block is then written in place instead of scattered);
 * info[1] = slot capacity (row pitch) of that matrix, info[2] / info[3] = first slot / reserved slots of the new members.
 * All zero for every other step.                                                                                               */
int genphi_plan_step_slots(const genphi_plan *plan, int32_t step, int64_t *info);

/* Diagnostic: the work lists of SPLIT level step `step` -- the hub walk over the parent graph of the cut's rows
 * (csrc/planner.h, WalkLists): desc4 = 4 ints per work row (storage row, output row, row to stage or n_prev, rank word),
 * seg4 = 4 ints per segment (first work row, hub row, leading rows without a row to stage, type 0 / 1 / 2) + 2
 * terminators, run4 = 4 ints per run (first segment, hub row | n0 << 16, first and end work row of that segment) + 1
 * terminator.  Call with NULL arrays for the counts.
 * Tests check its invariants (every row once; a type-1 hub is the row staged last; staged rows <= one per child + one per run). */
int genphi_plan_step_walk(const genphi_plan *plan, int32_t step, int64_t *n_rows, int64_t *n_segs, int64_t *n_runs,
                          int32_t *desc4, int32_t *seg4, int32_t *run4);

/* Progress hook for the "Running step k of n (...)" lines the reference prints INSIDE its level loop (src/compute.jl:280-285, verbose):
 * cb(step, n_steps, user) is called on the calling thread right before level step `step` (0-based) is handed to the GPU, in order.
 * While a hook is set the sweep is enqueued launch by launch (never replayed from a captured graph).  cb = NULL removes it.
 * The hook must not call back into the library with the same plan, and must not throw / raise across the C boundary (ctypes swallows
 * a Python exception raised inside it).  A run of tiny steps goes to the GPU as ONE launch: the hooks of all its steps are called, in
 * order, before that launch.                                                                                                     */
typedef void (*genphi_step_fn)(int32_t step, int32_t n_steps, void *user);
int genphi_plan_set_step_hook(genphi_plan *plan, genphi_step_fn cb, void *user);

/* Diagnostic: the zero-aware leading levels of the plan's Float32 sweep.  The level matrices right below the founders are almost
 * empty (Psi = 1/2 I at the top, src/compute.jl:271-274; an entry is non-zero only where two members share an ancestor above), so
 * the sweep keeps cuts 0..k as lists of their non-zero entries -- what the reference's sparse_phi stores, src/compute.jl:391-394 --
 * and step k writes cut k+1 as the first dense matrix.  k is fixed by the first genphi_compute_device of the plan, which counts the
 * non-zero entries of the leading cuts on the GPU.  *k_out = k (-1: every level is dense, or no sweep has run yet); nnz[c] = non-zero
 * entries of cut c for the cuts that were counted (-1 = not counted), entries[c] = the (column, value) pairs cut c is stored as (each
 * non-zero entry once per child of its column: 8 bytes each; what a sparse step reads and writes), at most `cap` of each (either
 * may be NULL); returns how many were filled.                                                                                   */
int genphi_plan_sparse_levels(const genphi_plan *plan, int32_t *k_out, int64_t *nnz, int64_t *entries, int32_t cap);

/* Device memory the plan holds right now, in bytes (level matrices, row lists of the sparse cuts, the resident result, the index
 * arrays): 0 before the first compute and after genphi_plan_release_device.  What a cache of plans budgets with.             */
int64_t genphi_plan_device_bytes(const genphi_plan *plan);

/* Host only: an UPPER BOUND of the device memory a full-result Float32 sweep of this plan allocates (level matrices -- slot matrices of
 * in-place runs included --, the result at ITS pitch, delivery buffers, index arrays, the row-list arenas of the sparse cuts): what a
 * caller compares with the free memory of a GPU before choosing between replicated levels and column panels.  A sweep
 * whose leading cuts run on row lists allocates level matrices only for the cuts that exist as matrices (genea140: 0.35 of 3.2 GB).  */
int64_t genphi_plan_device_bytes_needed(const genphi_plan *plan);

/* 4 * sum_k (n_k^2 + n_{k+1}^2): the algorithmic HBM bytes of one compute. */
double genphi_plan_algorithmic_bytes(const genphi_plan *plan);

/* Replaces src/compute.jl:269-303 (Psi = 1/2 I, the level loop with the per-pair kernel
 * :105-158 under Threads.@threads :291-299, Psi = phi).  Runs on the GPU; the result is
 * left resident in HBM inside the plan (Float32, N x N, proband order, row pitch
 * genphi_result_ld floats).  Float32 storage per level, Float64 accumulation, one RN
 * Float64->Float32 conversion per entry per level: bit-identical to the reference.
 * opts and stats may be NULL.                                                                */
int genphi_compute_device(genphi_plan *plan, const genphi_opts *opts, genphi_stats *stats);

/* Device pointer / row pitch (in floats) / first row and row count of the resident result
 * of the last genphi_compute_device (the rows of the shard it was asked for).               */
int genphi_result_device(const genphi_plan *plan, const float **d_ptr, int64_t *ld,
                         int64_t *row_begin, int64_t *n_rows);

/* Copies the resident rows to host: out is (n_rows x N) dense row-major Float32
 * (== column-major for the symmetric full matrix, so a Julia Matrix{Float32}(undef,N,N)
 * can be passed directly when all rows were computed).                                       */
int genphi_result_to_host(genphi_plan *plan, float *out);

/* Float64 counterpart for results computed with GENPHI_FLAG_STORAGE_F64: out is (n_rows x N) dense
 * row-major Float64.  (genphi_result_to_host on such a result delivers RN32 of these values: one
 * rounding, which is what gen.f returns, src/compute.jl:500-511.)                                  */
int genphi_result_to_host_f64(genphi_plan *plan, double *out);

/* phi(individual_i, individual_j) (src/compute.jl:66-95: the exact Float64 Karigl recursion,
 * un-memoised and exponential on inbred pedigrees) for n_pairs pairs of IDs at once: one Float64
 * level sweep over the individuals named, one lookup per pair.  out[k] = Phi(id_i[k], id_j[k])
 * (id_i[k] == id_j[k] gives the self-kinship 1/2 + Phi(father, mother)/2).  Bit-identical to the
 * recursion while every kinship is exactly representable in Float64 (pedigrees less than ~26
 * generations deep), within 1e-15 relative beyond.  Pedigree arguments as for genphi_plan_create;
 * unknown ID -> GENPHI_ERR_UNKNOWN_ID.                                                              */
int genphi_phi_pairs(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                     int64_t n_pairs, const int64_t *id_i, const int64_t *id_j, double *out, int32_t device);

/* On-device reduction for phiMean(::Matrix{Float32}) (src/compute.jl:454-459) without moving the
 * matrix to the host: Float64 sum of all resident entries and of their diagonal entries.
 * mean off-diagonal kinship = (sum_all - sum_diag) / (N*N - N); with row shards the ranks add
 * their partial sums.  The reference accumulates in Float32 (Julia's pairwise sum), so its
 * value agrees to Float32 rounding, exactly when the sums are exact (geneaJi: 0.171875).      */
int genphi_result_sums(genphi_plan *plan, double *sum_all, double *sum_diag, int64_t *n_rows);

/* Point lookups in the resident result without moving the matrix: out[k] = Phi[rows[k], cols[k]]
 * (0-based positions in proband order, duplicates collapsed as in genphi_plan_create; rows must
 * lie in the resident row range).  This is what gen.f(pedigree, IDs) (src/compute.jl:500-511)
 * needs: the inbreeding coefficient of x is the kinship of its parents, one entry of the sweep
 * over the parents instead of the reference's un-memoised pairwise recursion (:66-95).  Values
 * are the Float32 entries widened to Float64.                                                  */
int genphi_result_entries(genphi_plan *plan, int64_t n, const int64_t *rows, const int64_t *cols,
                          double *out);

/* Convenience = genphi_compute_device + genphi_result_to_host: what the Julia shim's
 * phi(...; compute=true) calls.  out: N x N Float32, caller-owned.                           */
int genphi_compute_f32(genphi_plan *plan, float *out, const genphi_opts *opts,
                       genphi_stats *stats);

/* Native loader for the step before the path: gen.genealogy(filename; sort) (src/create.jl:161-189:
 * header row skipped, four whitespace-separated integers `ind father mother sex` per row) plus,
 * with sort != 0, _ordered_pedigree (src/create.jl:196-227: stable sort by maximum ancestral
 * depth).  Returns the pedigree in RANK ORDER, ready for genphi_plan_create, in arrays
 * allocated by the library (release each with genphi_free).  sex_out may be NULL.  Unknown
 * parent -> GENPHI_ERR_UNKNOWN_ID; sort == 0 and a parent after its child -> GENPHI_ERR_ORDER. */
int genphi_genealogy_read(const char *path, int32_t sort, int64_t *n, int64_t **ind, int64_t **father,
                          int64_t **mother, int64_t **sex_out);
/* The same for a table already in memory (gen.genealogy(dataframe; sort), src/create.jl:131-146, ordered by :196-254): ind / father /
 * mother (/ sex, may be NULL) in file order -> malloc'ed arrays in rank order (release with genphi_free).  Errors as
 * genphi_genealogy_read: GENPHI_ERR_DUPLICATE_ID, GENPHI_ERR_UNKNOWN_ID (a parent that is not an individual: KeyError in the reference),
 * GENPHI_ERR_ARG (a cycle), and with sort = 0 GENPHI_ERR_ORDER (a parent listed after its child: KeyError, src/create.jl:240-241). */
int genphi_genealogy_order(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, const int64_t *sex,
                           int32_t sort, int64_t *n_out, int64_t **ind_out, int64_t **father_out, int64_t **mother_out,
                           int64_t **sex_out);
void genphi_free(void *ptr);

/* gen.branching(pedigree; pro, ancestors) (src/extract.jl:65-186), the pruning step before the
 * path: keeps the individuals on the paths between the selected probands and ancestors.
 * Input: the pedigree in rank order (as for genphi_plan_create) plus sex (may be NULL -> 0).
 * pro == NULL / ancestors == NULL mean "not given" (Julia `nothing`): with only pro, the
 * probands and all their ancestors are kept; with only ancestors, they and all their
 * descendants (parents outside the set become unknown); with both, the intersection (parents
 * outside it become unknown); with neither, nobody.  Output arrays are in the input's rank
 * order, allocated by the library (genphi_free each).  Unknown ID -> GENPHI_ERR_UNKNOWN_ID.   */
int genphi_branching(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                     const int64_t *sex, int64_t n_pro, const int64_t *pro, int64_t n_anc,
                     const int64_t *ancestors, int64_t *n_out, int64_t **ind_out, int64_t **father_out,
                     int64_t **mother_out, int64_t **sex_out);

/* What the library keeps between calls, per process: device blocks of released plans (up to GENPHI_KEEP_MB, default 8 GiB but at most 1/16 of the device's memory, per
 * device, handed to the next plan instead of hipMalloc / hipFree), idle streams, the pinned staging ring of genphi_result_to_host,
 * and the device side + pinned buffer of the last genphi_sparse_phi call (up to GENPHI_SPARSE_KEEP_MB, default 1024).  A one-shot
 * gen.phi call on a mid-size pedigree otherwise spends most of its time in the allocator.  genphi_release_cached gives all of it
 * back to the driver (plans in use are not touched); genphi_cached_bytes = device bytes kept right now.  The reference has no
 * counterpart: its matrices are garbage-collected Julia arrays (src/compute.jl:291,301).                                      */
void genphi_release_cached(void);
int64_t genphi_cached_bytes(void);

/* Releases everything the plan holds on its GPU (index arrays, level matrices, the resident
 * result, streams, captured graphs) and keeps the host-side plan: the next genphi_compute_device
 * uploads again, on the same or on another device (opts->device).  The reference has no
 * counterpart (its matrices are garbage-collected Julia arrays, src/compute.jl:291,301).      */
int genphi_plan_release_device(genphi_plan *plan);

/* ---- gen.sparse_phi / KinshipMatrix (src/compute.jl:321-447, :31-46, :467-472) -------------------
 * The reference's second kinship algorithm: individuals are processed one at a time in queue order,
 * each kinship is RN32(phi[father, j]/2 + phi[mother, j]/2), parents are dropped when their children
 * are done, and the result is a dictionary of the probands' non-zero kinships keyed by rank.  Here the
 * queue is simulated on the host (integers only) and all individuals of one depth are computed by two
 * streaming kernels on a dense "active" matrix in HBM (csrc/sparse_phi.hip).  Values, getindex
 * semantics (incl. the reference's (earlier, later) vs (smaller, larger rank) key behaviour), the
 * number of stored entries `show` prints and phiMean's sums are those of the reference.
 *   genphi_sparse_phi      sparse_phi(pedigree, probandIDs); pedigree in rank order as for
 *                          genphi_plan_create; runs on the GPU (no CPU fallback)
 *   genphi_sparse_info     n_rows x n_rows KinshipMatrix with n_stored entries; Float64 sums of all
 *                          stored values and of the self kinships (phiMean = (all - diag) / (n (n-1) / 2))
 *   genphi_sparse_get      getindex(phi, ID1, ID2) for n pairs; an ID that is not a proband ->
 *                          GENPHI_ERR_UNKNOWN_ID (KeyError in the reference)
 *   genphi_sparse_entries  every stored entry as (row rank, column rank, value); returns their number
 *                          (the arrays may be NULL / cap 0 to ask for it)                               */
typedef struct genphi_sparse genphi_sparse;
int genphi_sparse_phi(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                      int64_t n_pro, const int64_t *pro_ids, int32_t device, genphi_sparse **out);
int genphi_sparse_info(const genphi_sparse *h, int64_t *n_rows, int64_t *n_stored, double *sum_all, double *sum_diag);
/* Measurement of the sweep that built the handle: number of waves (depths), device time of the sweep and of each
 * wave (HIP events on its stream; the first `cap` waves), algorithmic bytes 4 (n_old^2 + n_next^2) per wave (the
 * active matrix read once, the next one written once) and in total, the largest active set.  Any pointer may be NULL. */
int genphi_sparse_stats(const genphi_sparse *h, int32_t *n_waves, double *sweep_ms, double *algorithmic_bytes,
                        int64_t *max_active, float *wave_ms, double *wave_bytes, int32_t cap);
/* Host only, no GPU needed (diagnostic, no reference counterpart as a function): the schedule genphi_sparse_phi follows -- the order
 * in which individuals leave the reference's queue (src/compute.jl:336-345 founders, :431-439 children) as IDs, for each the processing
 * index at which it is dropped from the live set (:401-430: when its last child is processed; -1 = a proband, never dropped) and its
 * wave (one wave per depth).  *n_out = the number of individuals processed (the probands and their ancestors); at most `cap` entries
 * of each array that is not NULL are filled.  Errors as genphi_sparse_phi (unknown proband, parent after child, duplicate ID). */
int genphi_sparse_schedule(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                           const int64_t *pro_ids, int64_t cap, int64_t *order_ids, int64_t *retire_at, int32_t *wave, int64_t *n_out);
int genphi_sparse_get(const genphi_sparse *h, int64_t n, const int64_t *id1, const int64_t *id2, double *out);
int64_t genphi_sparse_entries(const genphi_sparse *h, int64_t cap, int64_t *row_rank, int64_t *col_rank, float *val);
void genphi_sparse_destroy(genphi_sparse *h);

/* ---- storage-sharded gen.phi: column panels + an exchange step (csrc/panel_phi.hip) --------------------
 * For pedigrees whose level matrices do not fit one GPU (the reference keeps two dense level matrices
 * alive, src/compute.jl:291,301).  Rank r of `world` stores the column panel of the members it owns (all
 * rows x its columns: 1/world of every level matrix) and, before every level step, receives the parent
 * columns of its new members that other ranks own.  This library packs / consumes DEVICE buffers; the
 * all-to-all itself is issued by the host (torch.distributed over RCCL/xGMI: genlib.jl_amd/distributed.py).
 *   create            plan + ownership + exchange lists (host only; identical arguments on every rank)
 *   begin             upload, Psi_1 = 1/2 I on the local columns
 *   exchange_counts   columns to send to / receive from every rank before step `step`, floats per column
 *   pack / compute    fill the send buffer; unpack the received columns and run the level step: the FULL / SPLIT
 *                     row kernels of the dense path on this rank's local columns (source rows = rows of the
 *                     rank's extended panel), the per-entry kernel when a panel row does not fit in LDS
 *   result_to_host    this rank's row block [row_begin, row_begin + n_rows) of Phi (proband order)      */
typedef struct genphi_panel genphi_panel;
int genphi_panel_create(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                        int64_t n_pro, const int64_t *pro_ids, int32_t rank, int32_t world, genphi_panel **out);
int64_t genphi_panel_n_steps(const genphi_panel *p);
double genphi_panel_step_ms(const genphi_panel *p, int32_t step);  /* device time of the step's kernels (unpack + level) in the last sweep, HIP events; -1 bad argument */
int genphi_panel_step_mode(const genphi_panel *p, int32_t step);   /* 0 FULL / 1 SPLIT row kernels on the local columns, 2 per-entry kernel; -1 bad argument */
int64_t genphi_panel_n_probands(const genphi_panel *p);
int genphi_panel_result_rows(const genphi_panel *p, int64_t *row_begin, int64_t *n_rows);
int genphi_panel_exchange_counts(const genphi_panel *p, int32_t step, int64_t *send_cols, int64_t *recv_cols, int64_t *col_floats);
double genphi_panel_device_bytes(const genphi_panel *p);
int genphi_panel_begin(genphi_panel *p, int32_t device);
int genphi_panel_pack(genphi_panel *p, int32_t step, float *d_send);
int genphi_panel_compute(genphi_panel *p, int32_t step, const float *d_recv);
/* The same two calls ordered by streams instead of host synchronisations (what distributed.py uses): caller_stream is the
 * hipStream_t the host driver enqueues its collective on (torch's current stream).  pack_on makes caller_stream wait for the
 * packed columns; compute_on makes the panel's stream wait for what caller_stream holds (the collective) before it unpacks.
 * Neither blocks the host; genphi_panel_sync (or result_to_host / step_ms) waits for the sweep.  caller_stream =
 * GENPHI_NO_STREAM: no collective runs between the two calls (a single rank), nothing is ordered across streams.        */
#define GENPHI_NO_STREAM ((void *)(intptr_t)-1)
int genphi_panel_pack_on(genphi_panel *p, int32_t step, float *d_send, void *caller_stream);
int genphi_panel_compute_on(genphi_panel *p, int32_t step, const float *d_recv, void *caller_stream);
int genphi_panel_sync(genphi_panel *p);
int genphi_panel_result_to_host(genphi_panel *p, float *out);
void genphi_panel_destroy(genphi_panel *p);

/* Frees host and device memory of the plan (NULL is allowed). */
void genphi_plan_destroy(genphi_plan *plan);

/* Message of the last error on this thread ("" if none). */
const char *genphi_last_error(void);

/* Library version string. */
const char *genphi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GENPHI_H */
