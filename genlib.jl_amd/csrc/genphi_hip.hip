// genphi_hip.hip -- gfx950 (MI355X / CDNA4) kernels and the C-ABI of the gen.phi hot path.
//
// What runs here replaces src/compute.jl:269-303 of the reference (Psi = 1/2 I, the level
// loop with the recursive per-pair kernel :105-158 under Threads.@threads, Psi = phi).
//
// Arithmetic contract: level matrices are Float32 in HBM; each
// entry is a Float64 sum of <= 4 Float32 loads, grouped exactly as the reference's
// recursion groups them, scaled by an exact power of two and converted to Float32 once
// (round-to-nearest-even, subnormals kept).  No fast-math, no contraction.
//
// HBM layout of a level matrix (cut of n members): (n + 1) rows x ld floats, ld = multiple
// of 64 >= n + 1.  Row n and the columns >= n are all zero, so "no parent" is index n and
// every gather is unconditional.
//
// Kernels
//   level_full_kernel   one workgroup per output row: the (<= 2) source rows of the row's
//       member are staged whole into LDS with 16-byte coalesced loads, then every output
//       column gathers its (<= 4) terms from LDS and the row is written coalesced.
//   level_split_kernel  cuts of ~20k..40k members (two rows no longer fit in 160 KB of LDS):
//       row A is staged, the A-row terms of every column go to registers, row B is staged
//       into the same buffer, the B-row terms are gathered, combined and the row written.
//   WIDE levels (a source row no longer fits in LDS, > ~36.8k members): the cut is stored
//       [dragged..., new...] and the level matrix is assembled block by block from streaming
//       passes: rows_compact_kernel (dragged x dragged = a stream compaction of the previous
//       matrix; new x dragged = the compacted half sum of two rows; the parent x parent matrix),
//       transpose_block_kernel (dragged x new), a FULL / SPLIT sub-step on the compacted parent
//       matrix that writes the new x new block in place, pad_zero_kernel.  Levels with few dragged
//       members: drag_rows_kernel (a dragged row whole, its new columns gathered from the row's
//       parent entries in LDS) and the transpose the other way round -- the parent rows are not
//       streamed for the new x dragged block.
//   levels_small_kernel a RUN of consecutive steps with cuts <= 128 members in one persistent
//       launch: both level matrices live in LDS, one barrier per level (deep small pedigrees
//       are launch-bound otherwise).
//   level_naive_kernel             one thread per entry, four global gathers (reference
//       kernel for A/B comparisons; opts.kernel = 1).
//   colperm_kernel                 proband-order delivery of a final level computed in
//       [dragged, new] order (WIDE last step only): source row staged through LDS in segments,
//       the chunk's perm words and values in registers, coalesced 16-byte loads and stores.
//   No MFMA anywhere: this is a gather-average, HBM-bound.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/genphi.h"
#include "panel_launch.h"
#include "devcache.h"
#include "planner.h"
#include "sparse_levels.h"

using genphi::LevelStep;
using genphi::Plan;

// ---------------------------------------------------------------------------------------------
// device code
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kOrdMask = 0x7fffffff;

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef int i4_t __attribute__((ext_vector_type(4)));

// Float64 grouping of the reference recursion.
//   a = Psi[A_i][A_j]  b = Psi[A_i][B_j]  c = Psi[B_i][A_j]  d = Psi[B_i][B_j]
//   i climbs first (rank_i > rank_j): (a + b) + (c + d); otherwise (a + c) + (b + d).
// Halvings are exact, so they are applied once as `scale` (1, 1/2 or 1/4).
__device__ __forceinline__ float combine(float a, float b, float c, float d, bool i_hi, double scale)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;      // swap the middle terms, not the sums
    const double s = (static_cast<double>(a) + static_cast<double>(x)) +
                     (static_cast<double>(y) + static_cast<double>(d));
    return static_cast<float>(s * scale);
}

// same, with the exact power-of-two weight given as an exponent (0, -1 or -2)
__device__ __forceinline__ float combine_e(float a, float b, float c, float d, bool i_hi, int e)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;
    const double s = (static_cast<double>(a) + static_cast<double>(x)) +
                     (static_cast<double>(y) + static_cast<double>(d));
    return static_cast<float>(__builtin_ldexp(s, e));
}

// Exactness certificate of a level-matrix row: every entry is 0 or >= 2^-27 (entries are <= 1).
// Such a Float32 is a multiple of 2^-50, so ANY partial sum of up to four of them is a multiple of
// 2^-50 that is <= 4: exactly representable in Float64.  All groupings of the reference's
// recursion then give the same, exact, sum, and the entry may be computed in
// whatever order is cheapest -- the result is still bit-identical to the reference.  Rows without
// the certificate go through the grouping-exact path.
//   cert_key(v) = bits(v) - 1: 0 -> 0xffffffff (fine), anything below cert_thresh -> not certified
// threshold word in LevelArgs::cert_thresh = bits(2^-27) - 1 = 0x31ffffff (a test hook may raise it)
__device__ __forceinline__ unsigned cert_key(float v) { return __float_as_uint(v) - 1u; }

__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    // consecutive work items on one XCD (blocks b and b+8 share an XCD): rows that share a
    // source row then hit the same L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct LevelArgs {
    const float *psi;        // previous level matrix
    float *out;              // this level's matrix (or the shard buffer of the last level)
    long long ld_prev, ld;
    int n_prev, n;
    int width;               // columns a row kernel writes: [0, n) and the zero padding [n, width); = ld except in the new x new sub-step of a WIDE level
    const int *srcA, *srcB, *ord;   // per member of this cut (storage order)
    const unsigned *pk;      // srcA | srcB << 16 (FULL / SPLIT modes: n_prev < 65536)
    const int *rows;         // work list: storage row ids (n_rows entries)
    const int *out_rows;     // row of `out` for each work item; nullptr = same as storage row
    int n_rows;
    int lds_row;             // floats per staged row in LDS
    int chunk_cols;          // SPLIT: columns per chunk (multiple of blockDim)
    int n_chunks;            // SPLIT: column chunks per row (work item = sibling group x chunk)
    int n_groups;            // SPLIT: sibling groups (runs of equal A source in the work list)
    int slot_off;            // SPLIT: float offset in LDS of the two work-queue hand-over slots
    int dbg;                 // GENPHI_WG_TIMES builds: record this launch's workgroup timing
    int zero_row;            // !=0: this launch also zeroes the "none" row n of `out` (intermediate levels)
    // exactness certificates (see cert_bad): one word per row of the previous / this level matrix,
    // != 0 when the row holds an entry in (0, 2^-27); nullptr = not tracked
    const int *cert_prev;
    int *cert_out;
    const int *glist;        // SPLIT: this launch's sibling groups, compacted on the device (nullptr = all of grp[])
    unsigned chunk_magic;    // SPLIT: floor(2^32 / n_chunks) + 1, so that item / n_chunks = umulhi(item, chunk_magic); 0 when n_chunks == 1
    const int *gcnt;         // SPLIT: [0] certified groups, [1] the others (of this launch's group list)
    unsigned cert_thresh;    // bits of the smallest certified value minus one (test hook raises it)
    int cert_fast;           // FULL: rows whose source rows are certified take the grouping-free body (0: grouping-exact body only)
    // Rows and columns of a launch are the same members, except in a COLUMN PANEL (storage-sharded levels,
    // panel_phi.hip): there a launch computes every row of the cut for the rank's LOCAL columns only, and the
    // source "row" is a row of the rank's extended panel (own + received columns; pk indexes into it).
    const int *ord_col;      // rank word per column (= ord unless panel)
    const int *diag_col;     // panel: local column of row member i, or -1 (nullptr: column i)
    const int2 *pdesc;       // panel, SPLIT: per work row (local column of the member or -1, panel column of its other source); handed to the
                             // kernels as an argument of its own (const, restrict: scalar loads -- through this struct they are vector loads,
                             // and the wait for one drains vmcnt, i.e. the next row's prefetch, at the start of every stage)
    int zrow;                // index of the all-zero "none" row of `out` (= n unless panel: the cut's size)
};

// ---- shared pieces of the row kernels --------------------------------------------------------
struct RowCtx {
    int i, Ai, Bi, ord_i, dcol;
    bool new_i, hasB;
    const float *rowA, *rowB;
    float *orowp;
    double sc_i;
    float diag;
};

__device__ __forceinline__ RowCtx row_setup(const LevelArgs &p)
{
    RowCtx r;
    const int w = xcd_remap(blockIdx.x, p.n_rows);
    r.i = p.rows[w];
    const long long orow = p.out_rows ? p.out_rows[w] : r.i;
    r.Ai = p.srcA[r.i]; r.Bi = p.srcB[r.i];
    const int o = p.ord[r.i];
    r.new_i = o < 0;
    r.ord_i = o & kOrdMask;
    r.hasB = r.Bi != p.n_prev;                     // workgroup-uniform: dragged / one-parent rows stage one row
    r.rowA = p.psi + (long long)r.Ai * p.ld_prev;
    r.rowB = p.psi + (long long)r.Bi * p.ld_prev;
    r.orowp = p.out + orow * p.ld;
    r.sc_i = r.new_i ? 0.5 : 1.0;
    // diagonal of a new member: 1/2 + Psi[A][B]/2 (zero when a parent is missing)
    r.diag = 0.f;
    r.dcol = p.diag_col ? p.diag_col[r.i] : r.i;      // column that holds the member's own entry (-1: not among this launch's columns)
    if (r.new_i && r.dcol >= 0) {
        const int colB = p.diag_col ? static_cast<int>(p.pk[r.dcol] >> 16) : r.Bi;      // B's column in the staged row
        r.diag = static_cast<float>(0.5 + 0.5 * static_cast<double>(r.rowA[colB]));
    }
    return r;
}

// global -> LDS copy of nvec float4, all loads of a batch issued before the first LDS write
template <int BATCH>
__device__ __forceinline__ void stage_row(float *s, const float *g, int nvec, int tid, int nt)
{
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float4 *s4 = reinterpret_cast<float4 *>(s);
    for (int base = tid; base < nvec; base += BATCH * nt) {
        // unconditional (clamped) loads and stores: a per-element guard would push r[] to scratch
        float4 r[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) r[k] = g4[min(base + k * nt, nvec - 1)];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) s4[min(base + k * nt, nvec - 1)] = r[k];
    }
}

// ---- FULL: both source rows staged whole in LDS; pk[j] = srcA | srcB << 16 -------------------
template <int U, bool POS_ORD>
__global__ void __launch_bounds__(1024) level_full_kernel(const LevelArgs p)
{
    extern __shared__ float lds[];
    float *sA = lds;
    float *sB = lds + p.lds_row;
    if (blockIdx.x == (unsigned)p.n_rows) {        // extra block: the all-zero "none" row of this level
        float *zr = p.out + (long long)p.zrow * p.ld;
        for (long long j = threadIdx.x; j < p.ld; j += blockDim.x) zr[j] = 0.f;
        return;
    }
    const RowCtx r = row_setup(p);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nvec = p.lds_row >> 2;               // columns [0, lds_row) include the zero column
    // Certified source rows (see cert_key): every Float64 partial sum of an entry is exact, so the grouping of
    // the reference's recursion cannot change the result and the entry is RN32(((a + b) + (c + d)) 2^e) in any
    // order.  The two source rows are then ADDED WHILE THEY ARE STAGED -- S[k] = A[k] + B[k] as ONE Float64 per
    // source column, in the LDS bytes the two Float32 rows would take -- and an entry is RN32((S[A_j] + S[B_j]) 2^e):
    // two 8-byte gathers, one add, one scale, one conversion per entry instead of four gathers, four conversions,
    // three adds and the rank selects; no rank words are loaded.  Workgroup-uniform branch; rows with an
    // uncertified source take the grouping-exact body below (tests force both on the same inputs).
    if (p.cert_fast && (p.cert_prev[r.Ai] | p.cert_prev[r.Bi]) == 0) {
        double *S = reinterpret_cast<double *>(lds);
        const float4 *a4 = reinterpret_cast<const float4 *>(r.rowA), *b4 = reinterpret_cast<const float4 *>(r.rowB);
        double2 *S2 = reinterpret_cast<double2 *>(S);
        // a thread takes QUADS of columns (16-byte index loads, 16-byte row stores); the index words of its next quad are loaded one
        // iteration ahead -- the first one here, in front of the row loads, so that it overlaps the staging: a workgroup is alive for
        // ~10 us, and every L2 round trip it waits for alone is ~10 % of that.  (pk is padded with "zero column" words: the last quad
        // may run past n, the surplus entries are zeros inside the row's padding.)
        const int nq = (p.n + 3) >> 2;
        u4_t pkn = *reinterpret_cast<const u4_t *>(p.pk + 4 * max(min(tid, nq - 1), 0));      // (a rank's panel may have no column at all: n = 0)
        for (int base = tid; base < nvec; base += 2 * nt) {           // all loads of a batch in flight before the first LDS write
            const int k0 = base, k1 = min(base + nt, nvec - 1);
            const float4 x0 = a4[k0], x1 = a4[k1];
            float4 y0 = make_float4(0.f, 0.f, 0.f, 0.f), y1 = y0;
            if (r.hasB) { y0 = b4[k0]; y1 = b4[k1]; }
            S2[2 * k0] = make_double2(static_cast<double>(x0.x) + static_cast<double>(y0.x), static_cast<double>(x0.y) + static_cast<double>(y0.y));
            S2[2 * k0 + 1] = make_double2(static_cast<double>(x0.z) + static_cast<double>(y0.z), static_cast<double>(x0.w) + static_cast<double>(y0.w));
            S2[2 * k1] = make_double2(static_cast<double>(x1.x) + static_cast<double>(y1.x), static_cast<double>(x1.y) + static_cast<double>(y1.y));
            S2[2 * k1 + 1] = make_double2(static_cast<double>(x1.z) + static_cast<double>(y1.z), static_cast<double>(x1.w) + static_cast<double>(y1.w));
        }
        const double sc = r.new_i ? 0.25 : 0.5;    // row weight times the column's 1/2 (a dragged column is A = B = itself)
        unsigned ckf = 0xffffffffu;
        const int dq = (r.new_i && r.dcol >= 0) ? (r.dcol >> 2) : -1, de = r.dcol & 3;       // the quad / lane of the member's own entry
        __syncthreads();
        for (int q = tid; q < nq; q += nt) {
            const u4_t pk4 = pkn;
            if (q + nt < nq) pkn = *reinterpret_cast<const u4_t *>(p.pk + 4 * (q + nt));
            f4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = static_cast<float>((S[pk4[e] & 0xffff] + S[pk4[e] >> 16]) * sc);
            if (q == dq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (e == de) ? r.diag : v[e];
            }
            ckf = min(ckf, min(min(cert_key(v[0]), cert_key(v[1])), min(cert_key(v[2]), cert_key(v[3]))));
            *reinterpret_cast<f4_t *>(r.orowp + 4 * q) = v;
        }
        if (p.cert_out && ckf < p.cert_thresh) p.cert_out[r.i] = 1;
        for (long long j = 4LL * nq + tid; j < p.width; j += nt) r.orowp[j] = 0.f;
        return;
    }
    stage_row<4>(sA, r.rowA, nvec, tid, nt);
    if (r.hasB) stage_row<4>(sB, r.rowB, nvec, tid, nt);
    __syncthreads();
    const int e_ij = (r.new_i ? -1 : 0) - 1;       // every column has weight 1/2 (dragged: A = B = itself)
    unsigned ck = 0xffffffffu;
    for (int j0 = tid; j0 < p.n; j0 += U * nt) {
        unsigned pk[U]; int oj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(j0 + u * nt, p.n - 1);
            pk[u] = p.pk[j];
            oj[u] = POS_ORD ? 0 : p.ord_col[j];
        }
        float v[U];
        if (r.hasB) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int Aj = pk[u] & 0xffff, Bj = pk[u] >> 16;
                const float a = sA[Aj], b = sA[Bj], c = sB[Aj], d = sB[Bj];
                const bool i_hi = POS_ORD ? (j0 + u * nt < r.i) : (r.ord_i > (oj[u] & kOrdMask));
                v[u] = combine_e(a, b, c, d, i_hi, e_ij);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int Aj = pk[u] & 0xffff, Bj = pk[u] >> 16;
                v[u] = combine_e(sA[Aj], sA[Bj], 0.f, 0.f, true, e_ij);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * nt;
            if (j < p.n) {
                const float val = (j == r.dcol && r.new_i) ? r.diag : v[u];
                ck = min(ck, cert_key(val));
                r.orowp[j] = val;
            }
        }
    }
    if (p.cert_out && ck < p.cert_thresh) p.cert_out[r.i] = 1;      // exactness certificate of the row (see cert_key)
    // zero columns [n, ld): the "none" column of this level and its pitch padding
    for (long long j = p.n + tid; j < p.width; j += nt) r.orowp[j] = 0.f;
}

// ---- SPLIT: one whole source row in LDS at a time (cuts of ~20k..40k members) ----------------
// Persistent, software-pipelined.  A work item is (output row, column chunk); a workgroup
// walks its share of the items.  Per item: stage A lands in LDS, the A-row terms (a, b) of
// the chunk's columns go to registers, stage B lands in the same LDS buffer, the B-row terms
// are gathered, combined and written.  While a stage is being gathered from LDS the NEXT
// stage (row B, or the next item's row A) is already in flight from HBM/L2 into registers,
// so HBM never idles behind the LDS gathers.  Index words are loaded before the prefetch is
// issued: vmcnt retires in order, so the gathers only wait for the (older) index loads.
// The workgroups of one XCD walk consecutive items, so the chunks of one row (and rows that
// share a source row) are staged from that XCD's L2 after the first touch.
// Columns keep any order (no sorted-column requirement, no proband-order pass).
//   pk[j] = A_j | B_j << 16; a dragged column is stored as A = B = itself, so EVERY column has
//   weight 1/2 (x + x is exact).  POS_ORD: new members with both parents appear in rank order
//   along the storage order, so "row climbs first" is the position test j < i; otherwise the
//   per-column rank word ord[j] is loaded.
template <typename T>
__device__ __forceinline__ T ld_off(const void *sbase, unsigned byte_off)
{   // uniform base + 32-bit per-lane byte offset => one VGPR of address (saddr form)
    return *reinterpret_cast<const T *>(static_cast<const char *>(sbase) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(void *sbase, unsigned byte_off, T v)
{
    *reinterpret_cast<T *>(static_cast<char *>(sbase) + byte_off) = v;
}


// Row store of 4 consecutive columns.  AUX = 0: plain store; otherwise a buffer store with that
// cache-policy word (16 = sc1: write-through, the line is not kept in the XCD's L2, so a
// streamed output row does not evict the source rows other workgroups are about to stage).
#ifndef GENPHI_STORE_AUX
#define GENPHI_STORE_AUX 0
#endif
__device__ __forceinline__ void store_row4(float *row, unsigned row_bytes, unsigned byte_off, f4_t v)
{
#if GENPHI_STORE_AUX == 0 && !defined(GENPHI_STORE_PLAIN)
    // non-temporal row stores: an output row is never re-read by this launch, so it should not push
    // the source rows other workgroups are about to stage out of the XCD's L2 (same-box A/B on cfg4:
    // 44.2 -> 42.2 ms; the 3 : 2 read : write ceiling of this access pattern rises 5.3 -> 5.8 TB/s,
    // profiles/microbench/out/r02_*.out)
    (void)row_bytes;
    __builtin_nontemporal_store(v, reinterpret_cast<f4_t *>(reinterpret_cast<char *>(row) + byte_off));
#elif GENPHI_STORE_AUX == 0
    (void)row_bytes;
    st_off<f4_t>(row, byte_off, v);
#else
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrc, byte_off, 0, GENPHI_STORE_AUX);
#endif
}

// A work item is (sibling group, column chunk): the rows of a group share source row A, which
// is staged and gathered ONCE; then only row B is staged per child.  desc[w] = (storage row,
// output row, B source, ord word) of work row w; grp[g] = (first work row, A source), with a
// terminating entry.  Children with a B source come first in a group.
//
// The kernel is ONE flat loop over stages (A stage of an item, then one B stage per child):
// a single store_pre site and a single load_pre site, so that the staging registers are
// allocated once -- register pressure is the limiter here, and a spill is fatal to the
// pipeline (scratch reloads wait on vmcnt(0), i.e. on the prefetch in flight).
// Profiling aid (compile with -DGENPHI_WG_TIMES=1): per workgroup (start, end) wall-clock ticks
// and items started, of the LAST level_split_kernel launch; read with genphi_debug_wg_times.
#ifndef GENPHI_WG_TIMES
#define GENPHI_WG_TIMES 0
#endif
#if GENPHI_WG_TIMES
__device__ unsigned long long g_wg_times[1024][3];
__device__ unsigned long long g_wg_clk[1024][2];      // shader-clock ticks (clock64) at start / end: effective frequency
__device__ unsigned long long g_wg_phase[2][1024][16];   // [thread 0 | last thread]: ticks per phase, A stages in [0..5], B stages in [8..13]; [6] / [7] = A / B stage counts
#define GENPHI_PHASE(k) do { const unsigned long long t_ = wall_clock64(); \
        if (threadIdx.x == 0 || threadIdx.x == NT - 1) atomicAdd(&dbgl[(threadIdx.x ? 16 : 0) + (k) + ph_b], (unsigned)(t_ - t_ph)); \
        t_ph = t_; } while (0)
#else
#define GENPHI_PHASE(k) do { } while (0)
#endif

template <int NTHREADS, int CPT, int STG, bool POS_ORD>
__global__ void __launch_bounds__(NTHREADS)
level_split_kernel(const LevelArgs p, const int4 *__restrict__ desc, const int4 *__restrict__ seg, const int *__restrict__ glist,
                   const int2 *__restrict__ pdesc, int *queue)
{
    extern __shared__ float lds[];
    constexpr unsigned NT = NTHREADS;
    float *sR = lds;

    // Work split: the workgroups that share an XCD (blockIdx % 8) drain one contiguous slice of
    // the item list together, in order, through a per-slice atomic counter (queue[xcd], zeroed
    // by the host): items cost 1..5 stages, a static split leaves a tail.  Thread 0 draws the
    // item after next during stage A of each item and hands it over through an LDS slot, so
    // the draw's latency never sits in front of a prefetch.
    // When its own slice is drained a workgroup goes on with the slices of the other XCDs
    // (same counters): the XCDs do not run at the same speed (the last one finished 8 % after
    // the first on the final level of cfg4), and a slice boundary is only a locality hint.
    const int n_items = (p.gcnt ? p.gcnt[1] : p.n_groups) * p.n_chunks;
    const int xcd = blockIdx.x & 7;
    const int q = n_items >> 3, rem = n_items & 7;
    // With certificates on (p.glist), this kernel owns the sibling groups with an uncertified
    // source row and level_split_fast_kernel the others: group_split_kernel has compacted the two
    // group lists on the device (glist = this kernel's groups, gcnt[1] = how many); when every
    // group is certified this launch ends at once.
    auto draw = [&]() -> int {                            // global item index, or n_items when all is drawn
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            const int x = (xcd + t) & 7;
            const int l = atomicAdd(&queue[x], 1);
            if (l < q + (x < rem ? 1 : 0)) return x * q + min(x, rem) + l;
        }
        return n_items;
    };
    if (p.zero_row && blockIdx.x == 0) {                  // the all-zero "none" row of this level
        float *zr = p.out + (long long)p.zrow * p.ld;
        for (long long j = threadIdx.x; j < p.ld; j += NT) zr[j] = 0.f;
    }
    if (n_items == 0) return;                             // nothing uncertified in this launch
    int *slot = reinterpret_cast<int *>(lds + p.slot_off);   // queue hand-over slots (thread 0 draws one item ahead)
    if (threadIdx.x == 0) {
        slot[0] = draw();
        slot[1] = draw();
    }
    __syncthreads();
    int cur_l = __builtin_amdgcn_readfirstlane(slot[0]);  // this item / the next one (global item indices)
    int nxt_l = __builtin_amdgcn_readfirstlane(slot[1]);
    __syncthreads();                                      // thread 0 rewrites slot[0] in its first stage, ahead of that stage's first barrier
#if GENPHI_WG_TIMES
    const unsigned long long t_start = wall_clock64();
    if (threadIdx.x == 0 && p.dbg) { g_wg_times[blockIdx.x][0] = t_start; g_wg_times[blockIdx.x][1] = t_start; g_wg_times[blockIdx.x][2] = 0; g_wg_clk[blockIdx.x][0] = clock64(); }
#endif
    if (cur_l >= n_items) return;
    int kc = 0;                                           // items this workgroup has started
    unsigned tl = threadIdx.x;                            // re-materialised per stage (see asm below)

    // Staging registers: STG float4 per thread cover a whole source row.  Loads and LDS
    // writes are UNCONDITIONAL: level matrices and the LDS buffer are padded so that the
    // over-read / over-write past the row's end is harmless (a per-element guard makes hipcc
    // keep the array in scratch and wait on every load).
    f4_t pre[STG];
    unsigned pk[CPT];                                     // A_j | B_j << 16 of this thread's columns
    float pa[CPT], pb[CPT];                               // A-row terms of this thread's columns
    // generic layout: per child of the group (<= kMaxGroupGeneric), bit k = "the row member
    // has the larger rank than column k".  Built once per item in stage A, where the rank
    // words oj[] die before pa/pb are born: costs 4 VGPRs instead of CPT.
    unsigned hb0 = 0, hb1 = 0, hb2 = 0, hb3 = 0;
    static_assert(CPT <= 32, "one 32-bit mask per child");
    // column of element k of this thread: quads of 4 consecutive columns (16-byte index loads
    // and row stores), quad q at cb + q * 4 * NT + 4 * tl
    static_assert(CPT % 4 == 0, "columns are handled in quads");
    constexpr int NQ = CPT / 4;

    // ---- stage state (all wave-uniform) ----
    int it = cur_l;                                       // current item
    int gk = (p.chunk_magic ? static_cast<int>(__umulhi(static_cast<unsigned>(it), p.chunk_magic)) : it);   // position in this launch's group list
    int chunk = it - gk * p.n_chunks;
    int g = glist ? glist[gk] : gk;                       // (a restrict kernel argument: scalar loads, see level_split_fast_kernel)
    // an item of this kernel is ONE segment (hub row, its rows without B source first, then <= 4 children with one)
    int wb = seg[g].x, we = seg[g + 1].x, Ai = seg[g].y, n0 = seg[g].z;
    int w = wb;                                           // child whose B row is the current stage (B stages)
    bool stage_is_a = true;
    bool have_next = nxt_l < n_items;
    unsigned cb = (unsigned)chunk * (unsigned)p.chunk_cols;
    // columns [n, ld) (the "none" column and the pitch padding) are written as part of the
    // last chunk: their padded index words point at the zero column, so they come out as 0
    unsigned ce = min(cb + (unsigned)p.chunk_cols, (unsigned)p.width);

#pragma unroll
    for (int k_ = 0; k_ < STG; ++k_) pre[k_] = ld_off<f4_t>(p.psi + (long long)Ai * p.ld_prev, (tl + k_ * NT) * 16u);

#if GENPHI_WG_TIMES
    // phase accumulators live in LDS (32 words behind the queue slots): no registers, no scratch
    unsigned *dbgl = reinterpret_cast<unsigned *>(lds + p.slot_off) + 8;
    if (threadIdx.x < 32) dbgl[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t_ph = wall_clock64();
    int ph_b = 0;                                         // 0: A stage, 8: B stage
#endif
    for (;;) {
        // keep the per-column address arithmetic inside the loop: hoisted out it costs VGPRs
        // per column.  z0 is a zero the compiler cannot see through: XOR-ing the loop-carried
        // arrays with it stops hipcc from hoisting their Float64 conversions / LDS addresses.
        unsigned z0 = 0;
        asm volatile("" : "+s"(z0));
        asm volatile("" : "+v"(tl));
        __builtin_assume(tl < NT);

        // ---- part 0: the stage's scalar descriptors, issued BEFORE the barriers so that their
        //      latency (scalar cache / L2 round trips, ~0.8 us per stage when they sat in front
        //      of the prefetch) overlaps with the wait for the other waves and the LDS writes ----
        int4 dsc = make_int4(0, 0, 0, 0);
        int nextB;                                      // B source of the next child, or n_prev
        if (stage_is_a) {
            nextB = (wb + n0 < we) ? desc[wb + n0].z : p.n_prev;
        } else {
            dsc = desc[w];
            nextB = (w + 1 < we) ? desc[w + 1].z : p.n_prev;
        }
        GENPHI_PHASE(5);                                // stage bookkeeping
#if GENPHI_WG_TIMES
        ph_b = stage_is_a ? 0 : 8;
#endif
        // (the hand-over slot read below was written at least one barrier ago: stage A of the previous item)
        if (stage_is_a && kc > 0) {
            nxt_l = __builtin_amdgcn_readfirstlane(slot[(kc - 1) & 1]);
            have_next = nxt_l < n_items;
        }
        // next stage: B row of the next child that has one, else row A of the next item
        // (a dummy row when nothing is left: an unconditional prefetch keeps `pre` in one set)
        const int next_item = nxt_l;
        const int gkn = have_next ? (p.chunk_magic ? static_cast<int>(__umulhi(static_cast<unsigned>(next_item), p.chunk_magic)) : next_item) : gk;
        const int gn = glist ? glist[gkn] : gkn;
        const int nextAi = seg[gn].y;
        // ---- part 1: index loads of this stage and the queue draw, BEFORE the prefetch (vmcnt retires in
        //      order, so the gathers below only wait for these) and before the first barrier: their
        //      registers are dead here, and the prefetch can then follow the LDS writes piece by piece ----
        if (stage_is_a) {
            // the item after next: drawn now by thread 0 (oldest memory op of the stage, so
            // waiting for it never waits for the prefetch), read by everyone one item later
            if (threadIdx.x == 0) slot[kc & 1] = draw();
            ++kc;
            if (!POS_ORD) {
                // rank words first, alone: they are folded into the per-child masks and dead
                // before pk / the prefetch are even issued (one exposed L2 round trip per item)
                int oj[CPT];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const i4_t v = ld_off<i4_t>(p.ord_col, (cb + q * 4 * NT + tl * 4) * 4u);
                    oj[4 * q] = v.x; oj[4 * q + 1] = v.y; oj[4 * q + 2] = v.z; oj[4 * q + 3] = v.w;
                }
                const int wc = wb + n0;                 // the segment's children with a B source (<= 4)
                const int o0 = desc[min(wc, we - 1)].w & kOrdMask;
                const int o1 = desc[min(wc + 1, we - 1)].w & kOrdMask;
                const int o2 = desc[min(wc + 2, we - 1)].w & kOrdMask;
                const int o3 = desc[min(wc + 3, we - 1)].w & kOrdMask;
                hb0 = hb1 = hb2 = hb3 = 0;
#pragma unroll
                for (int k = CPT - 1; k >= 0; --k) {    // shift-accumulate: bit k ends up at position k
                    const int ok = oj[k] & kOrdMask;
                    hb0 = (hb0 << 1) | (o0 > ok ? 1u : 0u);
                    hb1 = (hb1 << 1) | (o1 > ok ? 1u : 0u);
                    hb2 = (hb2 << 1) | (o2 > ok ? 1u : 0u);
                    hb3 = (hb3 << 1) | (o3 > ok ? 1u : 0u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const u4_t v = ld_off<u4_t>(p.pk, (cb + q * 4 * NT + tl * 4) * 4u);
                pk[4 * q] = v.x; pk[4 * q + 1] = v.y; pk[4 * q + 2] = v.z; pk[4 * q + 3] = v.w;
            }
        }
        __syncthreads();                                // previous gathers are done with the buffer
        GENPHI_PHASE(0);
        {
            // every staging register is reloaded with the next row's piece as soon as its LDS write has read it
            const float *src = p.psi + (long long)(nextB != p.n_prev ? nextB : nextAi) * p.ld_prev;
#pragma unroll
            for (int k_ = 0; k_ < STG; ++k_) {
                *reinterpret_cast<f4_t *>(reinterpret_cast<char *>(sR) + (tl + k_ * NT) * 16u) = pre[k_];
                pre[k_] = ld_off<f4_t>(src, (tl + k_ * NT) * 16u);
            }
        }
        GENPHI_PHASE(1);                                // wait for the prefetched row + LDS writes + next prefetch issued
        __syncthreads();
        GENPHI_PHASE(2);
#if GENPHI_WG_TIMES
        if (threadIdx.x == 0 || threadIdx.x == NT - 1) atomicAdd(&dbgl[(threadIdx.x ? 16 : 0) + (stage_is_a ? 6 : 7)], 1u);
#endif

        GENPHI_PHASE(3);                                // index loads + prefetch issue
        // ---- part 2: gathers from the staged row ----
        int wfin_b, wfin_e;                             // children without a B source to finish now
        if (stage_is_a) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { pa[k] = sR[pk[k] & 0xffff]; pb[k] = sR[pk[k] >> 16]; }
            wfin_b = wb;                                // the hub's rows without a B source lead the segment
            wfin_e = wb + n0;
        } else {
            const int ri = dsc.x, orow = dsc.y, oi = dsc.w;
            const bool new_i = oi < 0;
            float *orowp = p.out + (long long)orow * p.ld;
            const int e_ij = (new_i ? -1 : 0) - 1;       // 2^e: row weight times the column's 1/2
            // diagonal of a new member: 1/2 + Psi[A][B]/2 = 1/2 + Psi[B][A]/2 (bit-symmetric)
            const int dcol = pdesc ? pdesc[w].x : ri;           // the member's own column (panel: local column or -1)
            const float diag = static_cast<float>(0.5 + 0.5 * static_cast<double>(sR[pdesc ? pdesc[w].y : Ai]));
            const int qk = w - (wb + n0);               // which child of the segment (wave-uniform)
            const unsigned hi_bits = POS_ORD ? 0u : (qk == 0 ? hb0 : (qk == 1 ? hb1 : (qk == 2 ? hb2 : hb3)));
            const unsigned row_bytes = (unsigned)p.ld * 4u;
            unsigned ck = 0xffffffffu;                   // smallest cert_key of the entries written
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned jq = cb + q * 4 * NT + tl * 4;
                f4_t vq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    const unsigned j = jq + e;
                    const float c = sR[pk[k] & 0xffff], d = sR[pk[k] >> 16];
                    const bool i_hi = POS_ORD ? (j < (unsigned)ri) : (bool)((hi_bits >> k) & 1u);     // (POS_ORD is never used on panels)
                    vq[e] = combine_e(pa[k], pb[k], c, d, i_hi, e_ij);
                }
                ck = min(ck, min(min(cert_key(vq[0]), cert_key(vq[1])), min(cert_key(vq[2]), cert_key(vq[3])))); asm volatile("" : "+v"(ck));
                // a ragged last quad spills into the padding columns [n, ld), zeroed afterwards
                if (jq < ce) store_row4(orowp, row_bytes, jq * 4u, vq);
            }
            if (p.cert_out && ck < p.cert_thresh) p.cert_out[ri] = 1;         // plain store: the words only ever go 0 -> 1
            // the diagonal entry of a new member is patched by the thread that owns its column
            // (same thread as the quad store above, so the two stores stay ordered)
            if (new_i && dcol >= 0) {
                const unsigned r = (unsigned)dcol - cb;
                if ((unsigned)dcol >= cb && (unsigned)dcol < ce && ((r >> 2) & (NT - 1)) == tl)
                    st_off<float>(orowp, (unsigned)dcol * 4u, diag);
            }
            wfin_b = 0;
            wfin_e = 0;
        }
        // rows without a B source (dragged or one-parent rows of the hub): finish from the A-row terms
        for (int wf = wfin_b; wf < wfin_e; ++wf) {
            unsigned z1 = 0, tlf = tl;                  // opaque again: nothing per-column may be hoisted
            asm volatile("" : "+s"(z1));
            asm volatile("" : "+v"(tlf));
            __builtin_assume(tlf < NT);
            const int4 df = desc[wf];
            const bool new_f = df.w < 0;
            const unsigned dcol_f = static_cast<unsigned>(pdesc ? pdesc[wf].x : df.x);     // (-1 matches no column)
            float *orowp = p.out + (long long)df.y * p.ld;
            const int e_ij = (new_f ? -1 : 0) - 1;
            const unsigned row_bytes = (unsigned)p.ld * 4u;
            unsigned ckf = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned jq = cb + q * 4 * NT + tlf * 4;
                f4_t vq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    const float v = combine_e(__uint_as_float(__float_as_uint(pa[k]) ^ z1),
                                              __uint_as_float(__float_as_uint(pb[k]) ^ z1), 0.f, 0.f, true, e_ij);
                    vq[e] = (jq + e == dcol_f && new_f) ? 0.5f : v;
                }
                ckf = min(ckf, min(min(cert_key(vq[0]), cert_key(vq[1])), min(cert_key(vq[2]), cert_key(vq[3])))); asm volatile("" : "+v"(ckf));
                if (jq < ce) store_row4(orowp, row_bytes, jq * 4u, vq);
            }
            if (p.cert_out && ckf < p.cert_thresh) p.cert_out[df.x] = 1;
        }

        GENPHI_PHASE(4);                                // gathers, combine, row stores
        // ---- advance the stage state ----
        if (nextB != p.n_prev) {                        // next stage: B row of the next child
            w = stage_is_a ? wb + n0 : w + 1;
            stage_is_a = false;
        } else {                                        // next stage: row A of the next item
            if (!have_next) break;
            it = next_item;
            g = gn; gk = gkn;
            chunk = it - gk * p.n_chunks;
            wb = seg[g].x; we = seg[g + 1].x; Ai = nextAi; n0 = seg[g].z;
            w = wb;
            stage_is_a = true;
            cb = (unsigned)chunk * (unsigned)p.chunk_cols;
            ce = min(cb + (unsigned)p.chunk_cols, (unsigned)p.width);
        }
    }
#if GENPHI_WG_TIMES
    if (threadIdx.x == 0 && p.dbg) { g_wg_times[blockIdx.x][1] = wall_clock64(); g_wg_times[blockIdx.x][2] = kc; g_wg_clk[blockIdx.x][1] = clock64(); }
    __syncthreads();
    if (threadIdx.x < 32 && p.dbg) g_wg_phase[threadIdx.x >> 4][blockIdx.x][threadIdx.x & 15] = dbgl[threadIdx.x];
#endif
}

// ---- SPLIT, certified rows: the grouping-free body ---------------------------------------------
// Same work items, queue, pipeline and LDS use as level_split_kernel, for the sibling groups whose
// source rows (A and every child's B) all carry the exactness certificate (cert_bad == 0): every
// Float64 partial sum of an entry is then exact, so the reference's rank-dependent grouping cannot
// change the result and the entry is simply  RN32( ((a + b) + (c + d)) / 4 ):
//   stage A keeps  pab = (a + b) / 4  as ONE Float64 per column (the two registers that hold a and b
//   in the grouping-exact kernel), stage B is  RN32( fma(c + d, 1/4, pab) )  -- 2 conversions, 1 add,
//   1 fma, 1 conversion per entry instead of 4 + 3 + 1 + 1 and the grouping selects; no rank words,
//   no per-child masks (so groups of up to 8 children in every level), fewer registers.
// Bit-identical to the grouping-exact kernel on certified rows (tests force both on the same input).
// CHAIN: the run lists may hold chain steps (type-1 segments: GENPHI_MAX_RUN > 1).  The default lists (one hub per run) never do,
// and their instantiation carries none of that state -- the kernel is at the SGPR limit.
template <int NTHREADS, int CPT, int STG, bool CERT, bool CHAIN>
__global__ void __launch_bounds__(NTHREADS)
level_split_fast_kernel(const LevelArgs p, const int4 *__restrict__ desc, const int4 *__restrict__ seg, const int4 *__restrict__ run,
                        const int *__restrict__ glist, const int2 *__restrict__ pdesc, int *queue)
{
    extern __shared__ float lds[];
    constexpr unsigned NT = NTHREADS;
    float *sR = lds;
    const int n_items = p.gcnt[0] * p.n_chunks;           // certified runs of this launch (group_split_kernel) x column chunks
    const int xcd = blockIdx.x & 7;
    const int q = n_items >> 3, rem = n_items & 7;
    auto draw = [&]() -> int {                            // next item of a certified run, or n_items
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            const int x = (xcd + t) & 7;
            const int l = atomicAdd(&queue[x], 1);
            if (l < q + (x < rem ? 1 : 0)) return x * q + min(x, rem) + l;
        }
        return n_items;
    };
    if (p.zero_row && blockIdx.x == 0) {                  // the all-zero "none" row of this level
        float *zr = p.out + (long long)p.zrow * p.ld;
        for (long long j = threadIdx.x; j < p.ld; j += NT) zr[j] = 0.f;
    }
    if (n_items == 0) return;                             // no certified run in this launch
    int *slot = reinterpret_cast<int *>(lds + p.slot_off);   // queue hand-over slots (thread 0 draws one item ahead)
    if (threadIdx.x == 0) {
        slot[0] = draw();
        slot[1] = draw();
    }
    __syncthreads();
    int cur_l = __builtin_amdgcn_readfirstlane(slot[0]);
    int nxt_l = __builtin_amdgcn_readfirstlane(slot[1]);
    __syncthreads();                                      // thread 0 rewrites slot[0] in its first stage, ahead of that stage's first barrier
    if (cur_l >= n_items) return;
    int kc = 0;
    unsigned tl = threadIdx.x;

    f4_t pre[STG];
    unsigned pk[CPT];                                     // A_j | B_j << 16 of this thread's columns
    double pab[CPT];                                      // (a + b) / 4 of this thread's columns: the expansion of the current hub row
    static_assert(CPT % 4 == 0, "columns are handled in quads");
    constexpr int NQ = CPT / 4;

    // ---- stage state (all wave-uniform).  An item is (run, column chunk); a run is a list of segments (see
    //      GroupLists): the first one stages its hub row (stage A), every child with a B source is a stage B, and
    //      the expansion of a segment's last B row becomes the hub of a following type-1 segment for free. ----
    // (few scalars are kept across stages -- the kernel is at the SGPR limit, and a scalar spilled to a VGPR lane costs
    // a vector register of the column state; what a stage needs beyond them is re-read from the scalar cache)
    int g, g_end, wb, we, n0, Ai;                         // current segment, end of the run, the segment's rows, its hub row
    unsigned cb, ce;                                      // column chunk [cb, ce)
    {
        const int rk = (p.chunk_magic ? static_cast<int>(__umulhi(static_cast<unsigned>(cur_l), p.chunk_magic)) : cur_l);   // position in this launch's run list
        const int chunk = cur_l - rk * p.n_chunks;
        // (glist is a kernel argument of its own, const and restrict: its loads are scalar.  Read through the
        // argument struct they were vector loads, and waiting for one drains vmcnt -- row stores included)
        const int4 rr = run[glist[rk]];                   // (first segment, hub | n0 << 16, its rows [z, w))
        g = rr.x; g_end = run[glist[rk] + 1].x;
        wb = rr.z; we = rr.w; n0 = rr.y >> 16; Ai = rr.y & 0xffff;
        cb = (unsigned)chunk * (unsigned)p.chunk_cols;
        ce = min(cb + (unsigned)p.chunk_cols, (unsigned)p.width);
    }
    int w = wb;                                           // child whose B row is the current stage (B stages)
    bool stage_is_a = true;
    bool have_next = nxt_l < n_items;

#pragma unroll
    for (int k_ = 0; k_ < STG; ++k_) pre[k_] = ld_off<f4_t>(p.psi + (long long)Ai * p.ld_prev, (tl + k_ * NT) * 16u);

    for (;;) {
        asm volatile("" : "+v"(tl));
        __builtin_assume(tl < NT);

        // ---- part 0: scalar descriptors of the stage, ahead of the barriers ----
        // (everything the NEXT stage needs is loaded here, one stage ahead and in front of the barriers: a scalar load
        // that misses the scalar cache takes ~1 us, and a chain of dependent ones at the start of a stage is exposed)
        int4 dsc = make_int4(0, 0, 0, 0);                 // (stage B) storage row, output row, staged row of child w
        int nextB;                                        // B row of the next stage, or n_prev when the next stage is another item's stage A
        bool seg_step = false, chain = false;             // this stage is the last child of its segment and the run goes on; ... with the staged row as the hub
        int4 nsg = make_int4(0, 0, 0, 0);                 // the segment the run goes on with
        int n_we = 0;
        if (stage_is_a) {
            nextB = (wb + n0 < we) ? desc[wb + n0].z : p.n_prev;
        } else {
            dsc = desc[w];
            if (w + 1 < we) {
                nextB = desc[w + 1].z;
            } else if (g + 1 < g_end) {
                seg_step = true;
                nsg = seg[g + 1];
                n_we = seg[g + 2].x;
                chain = CHAIN && nsg.w == 1;
                nextB = (nsg.x + nsg.z < n_we) ? desc[nsg.x + nsg.z].z : p.n_prev;
            } else {
                nextB = p.n_prev;
            }
        }
        const int ri = dsc.x, orow = dsc.y, Bi = dsc.z;
        // (the hand-over slot read below was written at least one barrier ago: stage A of the previous item)
        if (stage_is_a && kc > 0) {
            nxt_l = __builtin_amdgcn_readfirstlane(slot[(kc - 1) & 1]);
            have_next = nxt_l < n_items;
        }
        // the next item's run (a dummy one when nothing is left: an unconditional prefetch keeps `pre` in one set)
        const int rkn = have_next ? (p.chunk_magic ? static_cast<int>(__umulhi(static_cast<unsigned>(nxt_l), p.chunk_magic)) : nxt_l) : 0;
        const int rn = glist[rkn];
        const int4 nrun = run[rn];                        // its first segment: (index, hub row | n0 << 16, rows [z, w)); ONE load chain per
        const int gn_end = run[rn + 1].x;                 // item, issued a stage ahead: nothing is loaded when the item starts
        const int nextAi = nrun.y & 0xffff;
        // index loads and the queue draw BEFORE the prefetch (vmcnt retires in order); the prefetch itself
        // is issued piece by piece as the LDS writes free the staging registers, so it leads by the LDS
        // writes and the second barrier
        if (stage_is_a) {
            if (threadIdx.x == 0) slot[kc & 1] = draw();
            ++kc;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const u4_t v = ld_off<u4_t>(p.pk, (cb + q * 4 * NT + tl * 4) * 4u);
                pk[4 * q] = v.x; pk[4 * q + 1] = v.y; pk[4 * q + 2] = v.z; pk[4 * q + 3] = v.w;
            }
        }
        __syncthreads();                                // previous gathers are done with the buffer
        {
            const float *src = p.psi + (long long)(nextB != p.n_prev ? nextB : nextAi) * p.ld_prev;
#pragma unroll
            for (int k_ = 0; k_ < STG; ++k_) {
                *reinterpret_cast<f4_t *>(reinterpret_cast<char *>(sR) + (tl + k_ * NT) * 16u) = pre[k_];
                pre[k_] = ld_off<f4_t>(src, (tl + k_ * NT) * 16u);
            }
        }
        __syncthreads();

        // ---- part 2: gathers from the staged row ----
        int wfin_b = 0, wfin_e = 0;                     // rows without a B source to finish from pab after this stage
        if (!stage_is_a) {
            float *orowp = p.out + (long long)orow * p.ld;
            // a row with a B source is a new member with both parents: weight 1/2 x 1/2 per column
            const int dcol = pdesc ? pdesc[w].x : ri;           // the member's own column (panel: local column or -1)
            const float diag = static_cast<float>(0.5 + 0.5 * static_cast<double>(sR[pdesc ? pdesc[w].y : Ai]));
            const unsigned row_bytes = (unsigned)p.ld * 4u;
            unsigned ck = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned jq = cb + q * 4 * NT + tl * 4;
                f4_t vq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    const double cd = static_cast<double>(sR[pk[k] & 0xffff]) + static_cast<double>(sR[pk[k] >> 16]);
                    vq[e] = static_cast<float>(__builtin_fma(cd, 0.25, pab[k]));
                }
                if (CERT) { ck = min(ck, min(min(cert_key(vq[0]), cert_key(vq[1])), min(cert_key(vq[2]), cert_key(vq[3])))); asm volatile("" : "+v"(ck)); }
                if (jq < ce) store_row4(orowp, row_bytes, jq * 4u, vq);
            }
            if (dcol >= 0) {
                const unsigned rr = (unsigned)dcol - cb;
                if ((unsigned)dcol >= cb && (unsigned)dcol < ce && ((rr >> 2) & (NT - 1)) == tl)
                    st_off<float>(orowp, (unsigned)dcol * 4u, diag);
            }
            if (CERT) { if (ck < p.cert_thresh) p.cert_out[ri] = 1; }     // plain store: the words only ever go 0 -> 1
        }
        if (CHAIN ? (stage_is_a || chain) : stage_is_a) {
            // the expansion of the staged row: the hub of this segment (stage A), or -- the row is still in LDS -- of the
            // NEXT one (chain step: no hub row is staged for it).  ONE site writes pab: a second one inside the loop above
            // costs ~1.5 VGPRs per column (spills).
            unsigned z1 = 0, tlg = tl;                  // opaque: nothing of the loop above may be shared with this one
            asm volatile("" : "+s"(z1));
            asm volatile("" : "+v"(tlg));
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const unsigned pkk = pk[k] ^ z1;
                pab[k] = (static_cast<double>(sR[pkk & 0xffff]) + static_cast<double>(sR[pkk >> 16])) * 0.25;
            }
            // the (new) hub's rows without B source lead its segment
            const bool from_chain = CHAIN && !stage_is_a;
            wfin_b = from_chain ? nsg.x : wb;
            wfin_e = wfin_b + (from_chain ? nsg.z : n0);
        }
        // rows without a B source (dragged or one-parent rows of the hub): finish from pab alone
        for (int wf = wfin_b; wf < wfin_e; ++wf) {
            unsigned tlf = tl;
            asm volatile("" : "+v"(tlf));
            __builtin_assume(tlf < NT);
            const int4 df = desc[wf];
            const bool new_f = df.w < 0;
            const unsigned dcol_f = static_cast<unsigned>(pdesc ? pdesc[wf].x : df.x);     // (-1 matches no column)
            float *orowp = p.out + (long long)df.y * p.ld;
            const double sc = new_f ? 1.0 : 2.0;          // pab carries 1/4; a dragged row weighs 1, not 1/2
            const unsigned row_bytes = (unsigned)p.ld * 4u;
            unsigned ckf = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned jq = cb + q * 4 * NT + tlf * 4;
                f4_t vq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = static_cast<float>(pab[4 * q + e] * sc);
                    vq[e] = (jq + e == dcol_f && new_f) ? 0.5f : v;
                }
                if (CERT) { ckf = min(ckf, min(min(cert_key(vq[0]), cert_key(vq[1])), min(cert_key(vq[2]), cert_key(vq[3])))); asm volatile("" : "+v"(ckf)); }
                if (jq < ce) store_row4(orowp, row_bytes, jq * 4u, vq);
            }
            if (CERT) { if (ckf < p.cert_thresh) p.cert_out[df.x] = 1; }
        }

        // ---- advance the stage state ----
        if (stage_is_a) {
            w = wb + n0;
        } else if (w + 1 < we) {
            w = w + 1;
        } else if (seg_step) {                          // next segment of the run: hub = the row just staged (type 1) or the same hub (type 2)
            g = g + 1;
            if (CHAIN && chain) Ai = Bi;
            wb = nsg.x; we = n_we; n0 = nsg.z;
            w = wb + n0;
        }
        if (nextB != p.n_prev) {                        // next stage: B row of child w
            stage_is_a = false;
        } else {                                        // next stage: the first segment of the next item
            if (!have_next) break;
            const int chunk = nxt_l - rkn * p.n_chunks;
            g = nrun.x; g_end = gn_end;
            wb = nrun.z; we = nrun.w; n0 = nrun.y >> 16; Ai = nextAi;
            w = wb;
            stage_is_a = true;
            cb = (unsigned)chunk * (unsigned)p.chunk_cols;
            ce = min(cb + (unsigned)p.chunk_cols, (unsigned)p.width);
        }
    }
}

// Splits the runs of one SPLIT launch (see WalkLists in planner.h) by certificate, on the device: a run is
// certified when its first hub row and the B row of every one of its work rows carry the certificate
// (cert_prev == 0; hubs entered by a chain step are B rows).  Certified run indices are appended to list0 -- the
// items of level_split_fast_kernel; the 32 runs of a block stay contiguous and in order (block-level scan, one
// atomic per block), so the walk order survives up to the order in which the blocks arrive -- and the SEGMENTS of
// the other runs to list1, the items of the grouping-exact level_split_kernel.  cnt[0] / cnt[1] = their numbers.
// Eight lanes per run (its work rows strided over them: a run is up to ~40 rows, and one thread walking them alone
// costs two dependent loads per row at memory latency: 64 us per level where this takes 5), 32 runs per block.
constexpr int kSplitLanes = 8, kSplitRuns = 256 / kSplitLanes;
__global__ void __launch_bounds__(256)
group_split_kernel(const int4 *__restrict__ desc, const int4 *__restrict__ seg, const int4 *__restrict__ run, int n_runs,
                   const int *__restrict__ cert_prev, int *__restrict__ list0, int *__restrict__ list1, int *__restrict__ cnt)
{
    __shared__ int good[kSplitRuns];
    __shared__ int base;
    const int sub = threadIdx.x % kSplitLanes, lr = threadIdx.x / kSplitLanes;
    const int r = blockIdx.x * kSplitRuns + lr;
    const bool live = r < n_runs;
    int bad = 0, g0 = 0, g1 = 0;
    if (live) {
        g0 = run[r].x; g1 = run[r + 1].x;
        const int wb = seg[g0].x, we = seg[g1].x;
        if (sub == 0) bad = cert_prev[run[r].y & 0xffff];  // "none" (index n_prev) is never flagged
        for (int w = wb + sub; w < we; w += kSplitLanes) bad |= cert_prev[desc[w].z];
    }
#pragma unroll
    for (int d = 1; d < kSplitLanes; d <<= 1) bad |= __shfl_xor(bad, d);      // (the 8 lanes of a run sit in one wave)
    if (sub == 0) good[lr] = (live && bad == 0) ? 1 : 0;
    __syncthreads();
    int o0 = 0, t0 = 0;
    for (int k = 0; k < kSplitRuns; ++k) {
        if (k < lr) o0 += good[k];
        t0 += good[k];
    }
    if (threadIdx.x == 0) base = atomicAdd(&cnt[0], t0);
    __syncthreads();
    if (live && sub == 0) {
        if (bad == 0) list0[base + o0] = r;
        else {                                            // (rare: kinships below 2^-27 somewhere in the run)
            const int b1 = atomicAdd(&cnt[1], g1 - g0);
            for (int g = g0; g < g1; ++g) list1[b1 + g - g0] = g;
        }
    }
}

// ---- WIDE levels: streaming passes over level matrices whose rows do not fit in LDS ------------
// The cut is stored [dragged members by previous position..., new members by rank...], so
//   dragged x dragged  out[i][j] = Psi[s_i][s_j]                      s increasing: a stream compaction
//   new x dragged      out[x][j] = RN32((Psi[f_x][s_j] + Psi[m_x][s_j]) / 2)   (src/compute.jl:111-126:
//                      0. + h(Psi[s_j, f_x]) + h(Psi[s_j, m_x]); a two-term Float64 sum, order-free)
//   dragged x new      the transpose of new x dragged (every level matrix is bit-symmetric)
//   new x new          a FULL / SPLIT level step of its own on Psi_P = Psi[parents][parents]
// rows_compact_kernel does the first two and extracts Psi_P: for work row w,
//   out[orow(w)][k] = RN32((Psi[A(w)][idx[k]] + Psi[B(w)][idx[k]]) * scale(w)),  k < m,
// with idx ascending (coalescing survives: consecutive lanes read nearly consecutive floats) and
// B = none for single-source rows (uniform branch: the zero row is not streamed).
// rowdesc[w] = (A, B, output row, scale exponent: 0 -> x1, -1 -> x1/2).
__global__ void __launch_bounds__(256)
rows_compact_kernel(const float *__restrict__ psi, long long ld_prev, int none, const int4 *__restrict__ rowdesc,
                    const int *__restrict__ idx, int m, float *__restrict__ out, long long ld_out, int *__restrict__ cert_out,
                    unsigned cert_thresh)
{
    constexpr int U = 8;                                       // elements per thread, all loads in flight before the first use
    const int4 d = rowdesc[blockIdx.x];
    const float *ra = psi + (long long)d.x * ld_prev;
    const float *rb = psi + (long long)d.y * ld_prev;
    float *o = out + (long long)d.z * ld_out;
    const double sc = d.w == 0 ? 1.0 : 0.5;
    const int k0 = blockIdx.y * (256 * U) + threadIdx.x;
    int q[U];
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = idx[min(k0 + u * 256, m - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ra[q[u]];
    const bool two = d.y != none;                              // uniform: single-source rows do not stream the zero row
    if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = rb[q[u]];
    }
    unsigned ck = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + u * 256;
        const float v = two ? static_cast<float>((static_cast<double>(a[u]) + static_cast<double>(b[u])) * sc)
                            : static_cast<float>(static_cast<double>(a[u]) * sc);
        if (k < m) {
            ck = min(ck, cert_key(v));
            __builtin_nontemporal_store(v, o + k);
        }
    }
    if (cert_out && ck < cert_thresh) cert_out[d.z] = 1;
}

// A WIDE level that stays in place (LevelStep::stay): the new x dragged block of the new rows, written at the dragged members' own
// columns (their slots):  m[z(w)][idx[k]] = RN32((m[A(w)][idx[k]] + m[B(w)][idx[k]]) / 2),  k < n_idx.  Source and destination are
// the same matrix (no __restrict__): the new rows' slots hold no member of the source cut.  rowdesc as in rows_compact_kernel.
__global__ void __launch_bounds__(256)
rows_avg_kernel(const float *m, long long ld, int none, const int4 *__restrict__ rowdesc, const int *__restrict__ idx, int n_idx,
                float *mo, int *__restrict__ cert_out, unsigned cert_thresh)
{
    constexpr int U = 8;
    const int4 d = rowdesc[blockIdx.x];
    const float *ra = m + (long long)d.x * ld;
    const float *rb = m + (long long)d.y * ld;
    float *o = mo + (long long)d.z * ld;
    const double sc = d.w == 0 ? 1.0 : 0.5;
    const int k0 = blockIdx.y * (256 * U) + threadIdx.x;
    int q[U];
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = idx[min(k0 + u * 256, n_idx - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ra[q[u]];
    const bool two = d.y != none;                              // uniform
    if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = rb[q[u]];
    }
    unsigned ck = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float v = two ? static_cast<float>((static_cast<double>(a[u]) + static_cast<double>(b[u])) * sc)
                            : static_cast<float>(static_cast<double>(a[u]) * sc);
        if (k0 + u * 256 < n_idx) {
            ck = min(ck, cert_key(v));
            o[q[u]] = v;
        }
    }
    if (cert_out && ck < cert_thresh) cert_out[d.z] = 1;
}

// The dragged rows of a WIDE level in ONE pass over their source row: row j of the cut is the row of
// the same member in the previous cut,
//   columns [0, nd)   out[j][k] = Psi[a_j][idx[k]]                           (stream compaction)
//   columns [nd, n)   out[j][nd + i] = RN32((Psi[a_j][f_i] + Psi[a_j][m_i]) / 2)   (the dragged x new block)
// The row's entries at the PARENTS' positions are compacted into LDS while the row streams by -- the
// parents inside the source window of each chunk of 8 * NT output columns (pstart[c] .. pstart[c + 1]
// of the ascending `parents` list), so the second touch of a sector comes from L1 / L2 -- and the new
// columns are gathered from there: pkn[i] =
// (parent index of f_i) | (of m_i) << 16, n_par = none (a zero).
// Persistent and software-pipelined: a workgroup walks rows blockIdx.x, + gridDim.x, ...; its work
// units (row, chunk) form ONE stream with the index words two units ahead and the gathers one
// unit ahead of the unit being consumed -- across row boundaries too, so the start-up of a row (row
// pointer, first index words, first gathers) and its LDS-gather phase overlap the neighbouring rows'
// loads.  With one workgroup per CU (LDS) nothing else would hide them.
template <int NT>
__global__ void __launch_bounds__(NT)
drag_rows_kernel(const float *__restrict__ psi, long long ld_prev, const int *__restrict__ idx, int nd,
                 const int *__restrict__ parents, int n_par, const int *__restrict__ pstart, const unsigned *__restrict__ pkn,
                 int n_new, float *__restrict__ out, long long ld_out, int *__restrict__ cert_out, unsigned cert_thresh)
{
    extern __shared__ float lds[];
    float *P = lds;                                            // n_par + 1 floats
    constexpr int U = 8, PU = 4;
    const int tid = threadIdx.x, nb = gridDim.x;
    const int n_chunks = (nd + NT * U - 1) / (NT * U);
    struct Unit { int row, c; };                               // wave-uniform
    auto next = [&](Unit u) { Unit v; v.c = u.c + 1 == n_chunks ? 0 : u.c + 1; v.row = u.c + 1 == n_chunks ? u.row + nb : u.row; return v; };
    int qA[U], qB[U], pqA[PU], pqB[PU], srA, srB;              // index words, source row of the unit
    float aA[U], aB[U], paA[PU], paB[PU];
    auto issue_idx = [&](Unit u, int (&q)[U], int (&pq)[PU], int &sr) {
        const int row = min(u.row, nd - 1);                    // past the end: harmless duplicate loads
        sr = idx[row];
        const int k0 = u.c * (NT * U) + tid;
#pragma unroll
        for (int e = 0; e < U; ++e) q[e] = idx[min(k0 + e * NT, nd - 1)];
        const int pb = pstart[u.c];
#pragma unroll
        for (int e = 0; e < PU; ++e) pq[e] = parents[min(pb + tid + e * NT, n_par - 1)];
    };
    auto issue_gather = [&](const int (&q)[U], const int (&pq)[PU], int sr, float (&a)[U], float (&pa)[PU]) {
        const float *ra = psi + (long long)__builtin_amdgcn_readfirstlane(sr) * ld_prev;
#pragma unroll
        for (int e = 0; e < U; ++e) a[e] = ra[q[e]];
#pragma unroll
        for (int e = 0; e < PU; ++e) pa[e] = ra[pq[e]];
    };
    unsigned ck = 0xffffffffu;
    auto consume = [&](Unit u, const float (&a)[U], const float (&pa)[PU], int sr) {
        if (u.row >= nd) return;
        float *o = out + (long long)u.row * ld_out;
        const int k0 = u.c * (NT * U) + tid;
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const int k = k0 + e * NT;
            if (k < nd) {
                ck = min(ck, cert_key(a[e]));
                __builtin_nontemporal_store(a[e], o + k);
            }
        }
        const int pb = pstart[u.c], pe = pstart[u.c + 1];
#pragma unroll
        for (int e = 0; e < PU; ++e) {
            const int k = pb + tid + e * NT;
            if (k < pe) P[k] = pa[e];
        }
        if (pe - pb > PU * NT) {                               // a window with many parents (few dragged columns)
            const float *ra = psi + (long long)__builtin_amdgcn_readfirstlane(sr) * ld_prev;
            for (int k = pb + PU * NT + tid; k < pe; k += NT) P[k] = ra[parents[k]];
        }
        if (u.c + 1 < n_chunks) return;
        // last chunk of the row: the new columns from the parent entries in LDS
        if (tid == 0) P[n_par] = 0.f;
        __syncthreads();
        float *on = o + nd;
        for (int i0 = tid; i0 < n_new; i0 += 4 * NT) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pkn[min(i0 + e * NT, n_new - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * NT;
                const float v = static_cast<float>((static_cast<double>(P[w[e] & 0xffff]) + static_cast<double>(P[w[e] >> 16])) * 0.5);
                if (i < n_new) {
                    ck = min(ck, cert_key(v));
                    __builtin_nontemporal_store(v, on + i);
                }
            }
        }
        if (cert_out && ck < cert_thresh) cert_out[u.row] = 1;
        ck = 0xffffffffu;
        __syncthreads();                                       // P is rewritten by the next row's units
    };
    Unit u0; u0.row = blockIdx.x; u0.c = 0;
    if (u0.row >= nd) return;
    Unit u1 = next(u0);
    issue_idx(u0, qA, pqA, srA);
    issue_idx(u1, qB, pqB, srB);
    issue_gather(qA, pqA, srA, aA, paA);
    for (;;) {
        // in flight: gathers of u0 (set A), index words of u1 (set B)
        const int s0 = srA;
        Unit u2 = next(u1);
        issue_idx(u2, qA, pqA, srA);
        issue_gather(qB, pqB, srB, aB, paB);
        consume(u0, aA, paA, s0);
        if (u1.row >= nd) break;
        const int s1 = srB;
        Unit u3 = next(u2);
        issue_idx(u3, qB, pqB, srB);
        issue_gather(qA, pqA, srA, aA, paA);
        consume(u1, aB, paB, s1);
        if (u2.row >= nd) break;
        u0 = u2; u1 = u3;
    }
}

// dst[c][dst_col0 + r] = src[r][c] for r < rows, c < cols (64 x 64 tiles through LDS, both sides
// coalesced): the dragged x new block from the new x dragged block.  Flags the certificate of every
// destination row that receives an uncertified value.  (128 x 128 tiles -- 512-byte runs on both
// sides, 66 KB of LDS -- were measured 13 % slower on cfg4o: two workgroups per CU hide less latency.)
constexpr int kTT = 64;
// `shift` source rows of the first tile row are empty so that every destination run (64 floats) starts on
// a 128-byte line although dst_col0 is arbitrary: runs that straddle lines are partial-line writes from
// two workgroups at different times (same-box A/B on cfg4o: 367 -> 329 ms for the whole sweep,
// profiles/microbench/out/r02_ab_transpose_line_aligned_cfg4o.out; 128-row tiles: no further gain).
__global__ void __launch_bounds__(256)
transpose_block_kernel(const float *__restrict__ src, long long ld_src, int rows, int cols, float *__restrict__ dst,
                       long long ld_dst, int dst_col0, int shift, int *__restrict__ cert_out, unsigned cert_thresh)
{
    constexpr int TR = kTT;
    __shared__ float tile[TR][kTT + 1];
    const int r0 = (int)blockIdx.y * TR - shift, c0 = blockIdx.x * kTT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int kb = ty; kb < TR; kb += 32) {                   // 8 loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = min(max(r0 + kb + 4 * u, 0), rows - 1), c = min(c0 + tx, cols - 1);     // clamped: unconditional loads
            v[u] = src[(long long)r * ld_src + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tile[kb + 4 * u][tx] = v[u];
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < kTT; k += 4) {
        const int c = c0 + k;
#pragma unroll
        for (int h = 0; h < TR / 64; ++h) {
            const int r = r0 + h * 64 + tx;
            if (c < cols && r >= 0 && r < rows) {
                const float v = tile[h * 64 + tx][k];
                dst[(long long)c * ld_dst + dst_col0 + r] = v;
                if (cert_out && cert_key(v) < cert_thresh) cert_out[c] = 1;
            }
        }
    }
}

// In-place WIDE steps: new member r of the step sits at slot blk_slot[r / 64] + r % 64 (granules of 64 free slots).
// The new x new block, computed into a compact n_new x n_new buffer t (pitch ld_t) by the FULL / SPLIT sub-step, goes to the
// members' rows and columns of the level matrix; the rows' certificate words travel with them.  One workgroup per row.
__global__ void __launch_bounds__(256)
slots_scatter_kernel(const float *__restrict__ t, long long ld_t, int n_new, const int *__restrict__ blk_slot, float *__restrict__ m,
                     long long ld, const int *__restrict__ cert_t, int *__restrict__ cert_out)
{
    const int r = blockIdx.x;
    const int row = blk_slot[r >> 6] + (r & 63);
    const float4 *src = reinterpret_cast<const float4 *>(t + (long long)r * ld_t);
    float *dst = m + (long long)row * ld;
    const int n4 = (n_new + 3) / 4;                               // (the last quad may spill into the granule's padding slots)
    for (int c4 = threadIdx.x; c4 < n4; c4 += 256) {
        const int c = c4 * 4;
        *reinterpret_cast<float4 *>(dst + blk_slot[c >> 6] + (c & 63)) = src[c4];
    }
    if (threadIdx.x == 0 && cert_out) cert_out[row] = cert_t[r];
}
// certificate words of the new members' slots := 0 (before the step flags them)
// (and, in the same launch, the step's scratch certificate words -- those of Psi_P's rows and of the scatter buffer's: `extra`)
__global__ void __launch_bounds__(256) slots_clear_kernel(int *__restrict__ words, const int *__restrict__ blk_slot, int n_gran,
                                                          int *__restrict__ extra, int n_extra)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n_gran * 64) words[blk_slot[k >> 6] + (k & 63)] = 0;
    for (int q = k; q < n_extra; q += gridDim.x * 256) extra[q] = 0;
}
// dragged x new = (new x dragged)^T in place: source row r of the block = new member r (row slot(r) of m), destination
// column slot(r), for the columns / destination rows [c_lo, c_lo + cols).  64 x 64 tiles: a tile row is one granule.
__global__ void __launch_bounds__(256)
transpose_slots_kernel(float *m, long long ld, int n_new, const int *__restrict__ blk_slot, int c_lo, int cols, int *__restrict__ cert_out,
                       unsigned cert_thresh)
{
    __shared__ float tile[kTT][kTT + 1];
    const int g = blockIdx.y, c0 = c_lo + blockIdx.x * kTT;
    const int base = blk_slot[g];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int rows_here = min(64, n_new - g * 64);
    for (int kb = ty; kb < kTT; kb += 32) {                      // 8 loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rl = min(kb + 4 * u, rows_here - 1), c = min(c0 + tx, c_lo + cols - 1);       // clamped: unconditional loads
            v[u] = m[(long long)(base + rl) * ld + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tile[kb + 4 * u][tx] = v[u];
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < kTT; k += 4) {
        const int c = c0 + k;
        if (c < c_lo + cols && tx < rows_here) {
            const float v = tile[tx][k];
            m[(long long)c * ld + base + tx] = v;
            if (cert_out && cert_key(v) < cert_thresh) cert_out[c] = 1;
        }
    }
}

// In-place WIDE steps, new x dragged AND dragged x new in one pass: a workgroup computes a tile of 64 new rows (one granule) x
// kFT columns of a slot range that holds dragged members,
//   v = RN32((m[f_r][c] + m[m_r][c]) / 2)      (a wave reads 1 KB of a parent row: 256 columns)
// stores it into the new rows (m[slot(r)][c], 1 KB runs) and, through LDS, transposed into the new columns of the rows c
// (m[c][slot(r)], 256-byte runs) -- the new rows are not read back.  rowdesc: the step's new rows in granule order
// (A, B, output row = slot, scale).  Source and destination are the same matrix: the tile's rows and the new columns hold no
// member of the source cut.  Certificates of both the new rows and the rows c.
// (tile widths: the template parameter FT of rows_avg_t_kernel)
// rows of a wave's share whose parent-row loads are in flight together (2 x 16-byte loads per row and lane).  With the tile in LDS the
// kernel runs two waves per SIMD, so the bytes in flight have to come from each wave: same-box A/B, cfg3s 3.56 / 3.40 / 3.42 ms and
// cfg4o 155.4 / 148.8 / 147.9 ms at 4 / 8 / 16 rows (profiles/microbench/out/r04_ab_rows_avg_t_rows_in_flight_*.out)
#ifndef GENPHI_AVG_T_ROWS
#define GENPHI_AVG_T_ROWS 16
#endif
// FT = columns of a tile: 256 (a wave-instruction covers one row piece of 1 KB; the 66 KB tile holds the kernel to two workgroups per CU) or
// 128 (two row pieces of 512 bytes per wave-instruction; 33 KB: four workgroups per CU).
template <int FT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4)))
rows_avg_t_kernel(float *m, long long ld, int none, const int4 *__restrict__ rowdesc, int n_new, const int *__restrict__ blk_slot,
                  const int2 *__restrict__ tiles, int *__restrict__ cert_out, unsigned cert_thresh, int gran_fastest, int scalar_t)
{
    extern __shared__ float tile_dyn[];                          // [64][FT + 1]
    float (*tile)[FT + 1] = reinterpret_cast<float (*)[FT + 1]>(tile_dyn);
    constexpr int LPR = FT / 4, RPI = 64 / LPR;                  // lanes per row piece, rows per wave-instruction
    // (gran_fastest: consecutive workgroups take the SAME columns of different granules -- a parent's row piece is then read by
    // its children's workgroups close in time)
    const int g = gran_fastest ? blockIdx.x : blockIdx.y, ct = gran_fastest ? blockIdx.y : blockIdx.x;
    // the column tiles of ALL slot ranges that hold dragged members in one launch: tiles[ct] = (first column, end of its range)
    const int2 tl = tiles[ct];
    const int c0 = tl.x, c_hi = tl.y, c_lo = tl.x;
    const int base = blk_slot[g], rows_here = min(64, n_new - g * 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lc = (lane % LPR) * 4, lr = lane / LPR;            // this lane's four columns of the tile, its row inside a wave-instruction
    const int c = c0 + lc;                                       // (ranges are multiples of 64 columns)
    const bool in = c < c_hi;
    // ---- new x dragged: 16 rows per wave, RB at a time (2 RB / RPI 16-byte loads in flight per lane) ----
    constexpr int RB = GENPHI_AVG_T_ROWS / RPI;                  // wave-instructions per batch
    for (int r0 = w * 16; r0 < w * 16 + 16; r0 += RB * RPI) {
        int4 d[RB];
        f4_t a[RB], b[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            d[u] = rowdesc[g * 64 + min(r0 + u * RPI + lr, rows_here - 1)];
            const int cc = in ? c : c_lo;                        // (clamped: unconditional loads)
            a[u] = *reinterpret_cast<const f4_t *>(m + (long long)d[u].x * ld + cc);
            b[u] = *reinterpret_cast<const f4_t *>(m + (long long)d[u].y * ld + cc);       // ("none" is the all-zero row)
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + u * RPI + lr;
            if (r >= rows_here) continue;                        // (`continue`, not `break`: the loop must unroll, d / a / b are registers)
            const double sc = d[u].w == 0 ? 1.0 : 0.5;
            f4_t v;
            unsigned ck = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = static_cast<float>((static_cast<double>(a[u][e]) + static_cast<double>(b[u][e])) * sc);
                tile[r][lc + e] = v[e];
                ck = min(ck, cert_key(v[e]));
            }
            if (in) {
                *reinterpret_cast<f4_t *>(m + (long long)d[u].z * ld + c) = v;
                if (cert_out && ck < cert_thresh) cert_out[d[u].z] = 1;
            }
        }
    }
    __syncthreads();
    // ---- dragged x new: column k of the tile is a 256-byte run of row c0 + k ----
    if (rows_here == 64 && !scalar_t) {
        // 16-byte stores: a lane takes four consecutive new members (tile rows 4 q .. 4 q + 3) of one destination row, sixteen lanes
        // cover the row's 256-byte run, a wave four destination rows per instruction
        const int q = lane & 15, sub = lane >> 4;
        for (int k = w * 4 + sub; k < FT; k += 16) {
            const int cr = c0 + k;
            if (cr < c_hi) {
                f4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tile[4 * q + e][k];
                *reinterpret_cast<f4_t *>(m + (long long)cr * ld + base + 4 * q) = v;
                if (cert_out && min(min(cert_key(v[0]), cert_key(v[1])), min(cert_key(v[2]), cert_key(v[3]))) < cert_thresh) cert_out[cr] = 1;
            }
        }
    } else {
        const int tx = lane;
        for (int k = w; k < FT; k += 4) {
            const int cr = c0 + k;
            if (cr < c_hi && tx < rows_here) {
                const float v = tile[tx][k];
                m[(long long)cr * ld + base + tx] = v;
                if (cert_out && cert_key(v) < cert_thresh) cert_out[cr] = 1;
            }
        }
    }
    (void)none;
}

// zero padding of a level matrix: columns [n, ld) of rows 0..n-1 and the whole "none" row n
// (width <= ld: the columns the level owns -- the entry cut of an in-place run has the run's pitch, but only its own width is padded)
__global__ void __launch_bounds__(256) pad_zero_kernel(float *__restrict__ m, long long ld, int n, long long width)
{
    const int r = blockIdx.x;
    float *row = m + (long long)r * ld;
    for (long long j = (r < n ? n : 0) + threadIdx.x; j < width; j += 256) row[j] = 0.f;
}

__global__ void level_naive_kernel(const LevelArgs p)
{
    const int w = blockIdx.x;                        // grid.y is limited to 65535: rows go on x
    const long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= p.ld) return;
    const int i = p.rows[w];
    const long long orow = p.out_rows ? p.out_rows[w] : i;
    float v = 0.f;
    if (j < p.n) {
        const int Ai = p.srcA[i], Bi = p.srcB[i], oi = p.ord[i];
        const int Aj = p.srcA[j], Bj = p.srcB[j], oj = p.ord[j];
        const float *rowA = p.psi + (long long)Ai * p.ld_prev;
        const float *rowB = p.psi + (long long)Bi * p.ld_prev;
        if (j == i && oi < 0) {
            v = static_cast<float>(0.5 + 0.5 * static_cast<double>(rowA[Bi]));
        } else {
            const double sc = (oi < 0 ? 0.5 : 1.0) * (oj < 0 ? 0.5 : 1.0);
            v = combine(rowA[Aj], rowA[Bj], rowB[Aj], rowB[Bj], (oi & kOrdMask) > (oj & kOrdMask), sc);
        }
        if (p.cert_out && cert_key(v) < p.cert_thresh) p.cert_out[i] = 1;
    }
    p.out[orow * p.ld + j] = v;
}

// ---- Float64 storage (opts flag GENPHI_FLAG_STORAGE_F64) -----------------------------------------
// The pairwise recursion of the reference, phi(i::Individual, j::Individual) (src/compute.jl:66-95),
// and gen.f built on it (:500-511) work in Float64 throughout and never round to Float32 between
// generations.  The same level sweep with Float64 level matrices reproduces them: every kinship
// is a dyadic rational, exactly representable in Float64 while the pedigree is less than ~26
// generations deep (then the sweep and the recursion are bit-identical whatever their order of
// operations), and within a few ulp (<< 1e-12) beyond.  One thread per entry, four global gathers;
// meant for the small proband sets of gen.f / pairwise queries, not for throughput.
//   colmap: member index of output column j (the last level is delivered in proband order), or nullptr
__global__ void level_naive64_kernel(const double *__restrict__ psi, long long ld_prev, int n_prev, double *__restrict__ out,
                                     long long ld, int n, const int *__restrict__ srcA, const int *__restrict__ srcB,
                                     const int *__restrict__ ord, const int *__restrict__ rows, const int *__restrict__ out_rows,
                                     const int *__restrict__ colmap, int n_cols)
{
    const int w = blockIdx.x;
    const long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const int i = rows ? rows[w] : w;
    const long long orow = out_rows ? out_rows[w] : i;
    double v = 0.0;
    if (j < n_cols) {
        const int jm = colmap ? colmap[j] : static_cast<int>(j);
        const int Ai = srcA[i], Bi = srcB[i], oi = ord[i];
        const int Aj = srcA[jm], Bj = srcB[jm], oj = ord[jm];
        const double *rowA = psi + (long long)Ai * ld_prev;
        const double *rowB = psi + (long long)Bi * ld_prev;
        if (jm == i && oi < 0) {
            v = 0.5 + 0.5 * rowA[Bi];
        } else {
            const double sc = (oi < 0 ? 0.5 : 1.0) * (oj < 0 ? 0.5 : 1.0);
            const double a = rowA[Aj], b = rowA[Bj], c = rowB[Aj], d = rowB[Bj];
            const bool i_hi = (oi & kOrdMask) > (oj & kOrdMask);
            const double x = i_hi ? b : c, y = i_hi ? c : b;
            v = ((a + x) + (y + d)) * sc;
        }
    }
    (void)n_prev; (void)n;
    out[orow * ld + j] = v;
}

// Float64 storage, cuts whose two source rows fit in LDS (2 x 8 bytes x (n_prev + 1) <= 160 KB): the FULL
// kernel's shape -- both source rows of an output row staged whole with 16-byte coalesced loads, four 8-byte LDS
// gathers per entry, coalesced row stores -- with the reference's grouping (src/compute.jl:66-95 climbs the
// higher-ranked individual first) and no rounding to Float32.  Same arguments as level_naive64_kernel + the number
// of rows.  PERSISTENT: a workgroup walks rows w = blockIdx.x, + gridDim.x, ... and
//   - owns the same columns in every row, so their index words (sources, rank word, member) are loaded ONCE into
//     registers (they were three dependent global loads per entry),
//   - has the NEXT row pair in flight into registers while the current one is gathered from LDS (rows of 7k+ members
//     leave room for one workgroup per CU only: nothing else would hide the load latency).
typedef double d2_t __attribute__((ext_vector_type(2)));   // (HIP's d2_t is a struct of unions: arrays of it stay in scratch)
constexpr int kF64Cols = 20;      // columns per thread: ceil(10 304 / 512)  (ld of the widest cut whose rows fit)
constexpr int kF64Pre = 10;       // d2_t pieces per source row and thread: ceil(10 240 / 2 / 512)
__global__ void __launch_bounds__(512)
level_full64_kernel(const double *__restrict__ psi, long long ld_prev, int n_prev, double *__restrict__ out, long long ld,
                    const int *__restrict__ srcA, const int *__restrict__ srcB, const int *__restrict__ ord,
                    const int *__restrict__ rows, const int *__restrict__ out_rows, const int *__restrict__ colmap, int n_cols,
                    int lds_row, int n_rows)
{
    extern __shared__ double lds64[];
    double *sA = lds64, *sB = lds64 + lds_row;
    const int tid = threadIdx.x, nt = blockDim.x;
    unsigned pkj[kF64Cols];
    int oj[kF64Cols], jmv[kF64Cols];
    const int j0 = blockIdx.y * nt * kF64Cols;           // this workgroup's column chunk (outputs wider than 512 x 20 columns take several)
#pragma unroll
    for (int k = 0; k < kF64Cols; ++k) {
        const int j = j0 + tid + k * nt;
        pkj[k] = 0; oj[k] = 0; jmv[k] = -1;
        if (j < n_cols) {
            const int jm = colmap ? colmap[j] : j;
            jmv[k] = jm; pkj[k] = static_cast<unsigned>(srcA[jm]) | static_cast<unsigned>(srcB[jm]) << 16; oj[k] = ord[jm];
        }
    }
    const int nvec = lds_row >> 1;
    d2_t ra[kF64Pre], rb[kF64Pre];
#define GENPHI_F64_FETCH(W)                                                                                                    \
    {                                                                                                                          \
        const int fi = rows ? rows[W] : (W);                                                                                   \
        const d2_t *a2 = reinterpret_cast<const d2_t *>(psi + (long long)srcA[fi] * ld_prev);                            \
        const d2_t *b2 = reinterpret_cast<const d2_t *>(psi + (long long)srcB[fi] * ld_prev);  /* "none" = the zero row */ \
        _Pragma("unroll") for (int k = 0; k < kF64Pre; ++k) { const int q = min(tid + k * nt, nvec - 1); ra[k] = a2[q]; rb[k] = b2[q]; } \
    }
    int w = blockIdx.x;
    if (w >= n_rows) return;
    GENPHI_F64_FETCH(w)
    for (; w < n_rows; w += gridDim.x) {
        __syncthreads();                                  // the previous row's gathers are done
        {
            d2_t *sA2 = reinterpret_cast<d2_t *>(sA), *sB2 = reinterpret_cast<d2_t *>(sB);
#pragma unroll
            for (int k = 0; k < kF64Pre; ++k) { const int q = tid + k * nt; if (q < nvec) { sA2[q] = ra[k]; sB2[q] = rb[k]; } }
        }
        const int i = rows ? rows[w] : w;
        const long long orow = out_rows ? out_rows[w] : i;
        const int Bi = srcB[i], oi = ord[i];
        if (w + (int)gridDim.x < n_rows) GENPHI_F64_FETCH(w + (int)gridDim.x)      // in flight behind the gathers
        __syncthreads();
        const bool new_i = oi < 0;
        const int ord_i = oi & kOrdMask;
        double *orowp = out + orow * ld;
        unsigned z1 = 0;                                  // opaque zero: the LDS addresses of the 20 columns are row-invariant, and hipcc
        asm volatile("" : "+s"(z1));                      // would keep all 80 of them in registers across the row loop (spills)
#pragma unroll
        for (int k = 0; k < kF64Cols; ++k) {
            const int j = j0 + tid + k * nt;
            if (j >= ld) break;
            double v = 0.0;
            if (jmv[k] >= 0) {
                const unsigned pkz = pkj[k] ^ z1;
                const int ojz = oj[k] ^ static_cast<int>(z1);
                const unsigned Aj = pkz & 0xffffu, Bj = pkz >> 16;
                if (jmv[k] == i && new_i) {
                    v = 0.5 + 0.5 * sA[Bi];
                } else {
                    const double sc = (new_i ? 0.5 : 1.0) * (ojz < 0 ? 0.5 : 1.0);
                    const double a = sA[Aj], b = sA[Bj], c = sB[Aj], d = sB[Bj];
                    const bool i_hi = ord_i > (ojz & kOrdMask);
                    const double x = i_hi ? b : c, y = i_hi ? c : b;
                    v = ((a + x) + (y + d)) * sc;
                }
            }
            orowp[j] = v;
            if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // two columns' gathers in flight, not twenty (registers)
        }
    }
    (void)n_prev;
#undef GENPHI_F64_FETCH
}

// Float64 storage, cuts whose rows fit in LDS one at a time (8 bytes x (n_prev + 1) <= 160 KB: up to 20,479 members): the SPLIT
// kernels' shape in its simplest form.  A workgroup walks a contiguous piece of the row list (rows sharing the A source are
// adjacent in a step's work order) for one chunk of 512 x 14 columns: row A is staged and its terms (a, b) of the chunk's columns
// kept in registers for as long as the following rows share it; every row with a B source stages that row, gathers (c, d), combines
// with the reference's grouping and stores.  The next row to stage is in flight into registers behind the gathers.  Same arguments
// as level_full64_kernel.
constexpr int kS64Cols = 14;      // columns per thread and chunk (registers: 7 per column + the 80 of the row in flight: 246 VGPRs, no scratch; 16 spills).
                                  // 14 instead of round 3's 12: genea140's widest cuts (13.7k columns) take 2 chunks instead of 3, i.e. a third fewer stagings
constexpr int kS64Pre = 20;       // d2_t pieces per source row and thread: 20,480 / 2 / 512
__global__ void __launch_bounds__(512)
level_split64_kernel(const double *__restrict__ psi, long long ld_prev, int n_prev, double *__restrict__ out, long long ld,
                     const int *__restrict__ srcA, const int *__restrict__ srcB, const int *__restrict__ ord,
                     const int *__restrict__ rows, const int *__restrict__ out_rows, const int *__restrict__ colmap, int n_cols,
                     int lds_row, int n_rows)
{
    extern __shared__ double srow[];
    const int tid = threadIdx.x;
    constexpr int nt = 512;
    unsigned pkj[kS64Cols];
    int oj[kS64Cols], jmv[kS64Cols];
    double va[kS64Cols], vb[kS64Cols];
    const int j0 = blockIdx.y * nt * kS64Cols;
#pragma unroll
    for (int k = 0; k < kS64Cols; ++k) {
        const int j = j0 + tid + k * nt;
        pkj[k] = 0; oj[k] = 0; jmv[k] = -1; va[k] = vb[k] = 0.0;
        if (j < n_cols) {
            const int jm = colmap ? colmap[j] : j;
            jmv[k] = jm; pkj[k] = static_cast<unsigned>(srcA[jm]) | static_cast<unsigned>(srcB[jm]) << 16; oj[k] = ord[jm];
        }
    }
    const int per = (n_rows + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int w0 = blockIdx.x * per, w1 = min(n_rows, w0 + per);
    if (w0 >= w1) return;
    const int nvec = lds_row >> 1;
    d2_t pre[kS64Pre];
    auto fetch = [&](int r) {
        const d2_t *g2 = reinterpret_cast<const d2_t *>(psi + (long long)r * ld_prev);
#pragma unroll
        for (int k = 0; k < kS64Pre; ++k) pre[k] = g2[min(tid + k * nt, nvec - 1)];
    };
    auto to_lds = [&]() {
        __syncthreads();                                  // the gathers from the previous row are done
        d2_t *s2 = reinterpret_cast<d2_t *>(srow);
#pragma unroll
        for (int k = 0; k < kS64Pre; ++k) { const int q = tid + k * nt; if (q < nvec) s2[q] = pre[k]; }
        __syncthreads();
    };
    // the row staged after (w, A just staged?): B of w, else the first A change / B source of the following rows; -1: none left
    auto next_stage = [&](int w, bool after_a, int cur_a) -> int {
        if (after_a) { const int b = srcB[rows ? rows[w] : w]; if (b != n_prev) return b; }
        for (int v = w + 1; v < w1; ++v) {
            const int i2 = rows ? rows[v] : v;
            if (srcA[i2] != cur_a) return srcA[i2];
            if (srcB[i2] != n_prev) return srcB[i2];
        }
        return -1;
    };
    int cur_a = -1;
    {
        const int i0 = rows ? rows[w0] : w0;
        fetch(srcA[i0]);                                  // ("none" is the all-zero row: a parentless member's A)
    }
    for (int w = w0; w < w1; ++w) {
        const int i = rows ? rows[w] : w;
        const long long orow = out_rows ? out_rows[w] : i;
        const int Ai = srcA[i], Bi = srcB[i], oi = ord[i];
        const bool new_i = oi < 0;
        const int ord_i = oi & kOrdMask;
        unsigned z1 = 0;                                  // opaque zero (see level_full64_kernel)
        asm volatile("" : "+s"(z1));
        if (Ai != cur_a) {                                // stage A, keep its terms
            to_lds();
            cur_a = Ai;
            const int nx = next_stage(w, true, cur_a);
            if (nx >= 0) fetch(nx);
#pragma unroll
            for (int k = 0; k < kS64Cols; ++k) {
                const unsigned pkz = pkj[k] ^ z1;
                va[k] = srow[pkz & 0xffffu]; vb[k] = srow[pkz >> 16];
            }
        }
        const bool has_b = Bi != n_prev;
        if (has_b) {
            to_lds();                                     // row B
            const int nx = next_stage(w, false, cur_a);
            if (nx >= 0) fetch(nx);
        }
        double *orowp = out + orow * ld;
#pragma unroll
        for (int k = 0; k < kS64Cols; ++k) {
            const int j = j0 + tid + k * nt;
            if (j >= ld) break;
            double v = 0.0;
            if (jmv[k] >= 0) {
                const unsigned pkz = pkj[k] ^ z1;
                const int ojz = oj[k] ^ static_cast<int>(z1);
                if (jmv[k] == i && new_i) {
                    v = 0.5 + 0.5 * (has_b ? srow[Ai] : 0.0);     // Psi[B][A] = Psi[A][B] (bit-symmetric levels)
                } else {
                    const double sc = (new_i ? 0.5 : 1.0) * (ojz < 0 ? 0.5 : 1.0);
                    const double a = va[k], b = vb[k];
                    const double c = has_b ? srow[pkz & 0xffffu] : 0.0, d = has_b ? srow[pkz >> 16] : 0.0;
                    const bool i_hi = ord_i > (ojz & kOrdMask);
                    const double x = i_hi ? b : c, y = i_hi ? c : b;
                    v = ((a + x) + (y + d)) * sc;
                }
            }
            orowp[j] = v;
            if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ void half_identity64_kernel(double *m, long long ld, int n, const int *out_rows, int n_rows, const int *colmap)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    // row k of the output is member (colmap ? colmap[k'] ...): used only for the all-founders case,
    // where every member is a proband: diagonal of the delivered matrix
    (void)colmap; (void)out_rows; (void)n;
    m[(long long)k * ld + k] = 0.5;
}

__global__ void gather_entries64_kernel(const double *__restrict__ m, const long long *__restrict__ off, long long n,
                                        double *__restrict__ out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = m[off[k]];
}

// ---- SMALL: a run of consecutive level steps whose cuts have <= kSmallMax members, fused -----
// Deep small pedigrees (hundreds of generations of a few dozen individuals: breeding lines,
// cfg5) are launch-bound: a level is a few microseconds of work.  One workgroup keeps BOTH
// level matrices of such a run in LDS and walks the steps back to back: no launches, no HBM
// round trips, one barrier per level.  Same per-entry arithmetic as level_naive_kernel
// (src/compute.jl:105-158 on Float32 storage, one rounding per entry per level).
constexpr int kSmallMax = 128;                // members per cut
constexpr int kSmallPitch = kSmallMax + 4;    // row pitch in LDS; column / row n ("none") stay zero
struct SmallStep {
    const int *srcA, *srcB, *ord;
    int n_prev, n;
};

__global__ void __launch_bounds__(1024)
levels_small_kernel(const SmallStep *__restrict__ steps, int n_run, const float *__restrict__ in, long long ld_in,
                    int in_is_half_identity, float *__restrict__ out, long long ld_out, int *__restrict__ cert_out,
                    unsigned cert_thresh)
{
    extern __shared__ float lds[];
    constexpr int P = kSmallPitch;
    float *cur = lds, *nxt = lds + P * P;
    int *sa = reinterpret_cast<int *>(lds + 2 * P * P), *sb = sa + kSmallMax, *so = sb + kSmallMax;
    const int tid = threadIdx.x, nt = blockDim.x;

    // the run's input matrix, with its zero "none" row and column (index n_prev)
    const int n0 = steps[0].n_prev;
    for (int idx = tid; idx < (n0 + 1) * (n0 + 1); idx += nt) {
        const int i = idx / (n0 + 1), j = idx - i * (n0 + 1);
        float v = 0.f;
        if (i < n0 && j < n0) v = in_is_half_identity ? (i == j ? 0.5f : 0.f) : in[(long long)i * ld_in + j];
        cur[i * P + j] = v;
    }
    for (int s = 0; s < n_run; ++s) {
        const SmallStep st = steps[s];
        for (int k = tid; k < st.n; k += nt) { sa[k] = st.srcA[k]; sb[k] = st.srcB[k]; so[k] = st.ord[k]; }
        __syncthreads();                                        // `cur` and the index arrays are complete
        const int n = st.n, n1 = n + 1;
        for (int idx = tid; idx < n1 * n1; idx += nt) {
            const int i = idx / n1, j = idx - i * n1;
            float v = 0.f;                                      // row / column n: the next level's "none"
            if (i < n && j < n) {
                const int Ai = sa[i], Bi = sb[i], oi = so[i];
                if (j == i && oi < 0) {
                    v = static_cast<float>(0.5 + 0.5 * static_cast<double>(cur[Ai * P + Bi]));
                } else {
                    const int Aj = sa[j], Bj = sb[j], oj = so[j];
                    const double sc = (oi < 0 ? 0.5 : 1.0) * (oj < 0 ? 0.5 : 1.0);
                    v = combine(cur[Ai * P + Aj], cur[Ai * P + Bj], cur[Bi * P + Aj], cur[Bi * P + Bj],
                                (oi & kOrdMask) > (oj & kOrdMask), sc);
                }
            }
            nxt[i * P + j] = v;
        }
        __syncthreads();                                        // everyone is done reading `cur` / sa..so
        float *t = cur; cur = nxt; nxt = t;
    }
    // the run's result in the regular padded layout (rows 0..n incl. the zero row, whole pitch)
    const int n = steps[n_run - 1].n;
    for (long long idx = tid; idx < (long long)(n + 1) * ld_out; idx += nt) {
        const int i = static_cast<int>(idx / ld_out), j = static_cast<int>(idx - (long long)i * ld_out);
        const float v = (j <= n) ? cur[i * P + j] : 0.f;
        out[idx] = v;
        if (cert_out && i < n && cert_key(v) < cert_thresh) cert_out[i] = 1;      // exactness certificate of row i
    }
}

// ---- level step 0: the source matrix is Psi_1 = 1/2 I (src/compute.jl:271-274) -----------------
// Nothing needs to be read from HBM: Psi_1[x][y] = 1/2 when x == y is a real member, else 0.  Same
// per-entry arithmetic as level_naive_kernel with the four gathers replaced by index compares;
// one workgroup per output row, 16-byte index loads and stores.  Saves the 1/2 I memset (0.4 ms at cfg4) and
// every source-row read of the first level.
__global__ void __launch_bounds__(256) level_identity_kernel(const LevelArgs p)
{
    const int w = blockIdx.x;                          // one workgroup per output row
    const int i = p.rows[w];
    const long long orow = p.out_rows ? p.out_rows[w] : i;
    const int Ai = p.srcA[i], Bi = p.srcB[i], oi = p.ord[i];
    const int none = p.n_prev;
    auto h = [none](int x, int y) -> float { return (x == y && x != none) ? 0.5f : 0.f; };
    float *orowp = p.out + orow * p.ld;
    for (long long jq = (long long)threadIdx.x * 4; jq < p.width; jq += 4 * blockDim.x) {
        // an entry is non-zero only where row and column share a source (or on the diagonal):
        // rare, so most waves take the all-zero path and the kernel runs at the speed of its stores
        int Aj[4], Bj[4];
        if (jq + 3 < p.n) {                            // index arrays are 256-byte aligned, jq % 4 == 0
            const int4 a4 = *reinterpret_cast<const int4 *>(p.srcA + jq), b4 = *reinterpret_cast<const int4 *>(p.srcB + jq);
            Aj[0] = a4.x; Aj[1] = a4.y; Aj[2] = a4.z; Aj[3] = a4.w;
            Bj[0] = b4.x; Bj[1] = b4.y; Bj[2] = b4.z; Bj[3] = b4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = jq + e < p.n;
                Aj[e] = in ? p.srcA[jq + e] : none; Bj[e] = in ? p.srcB[jq + e] : none;
            }
        }
        bool hit[4], any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long j = jq + e;
            hit[e] = j < p.n && (j == i || (Ai != none && (Ai == Aj[e] || Ai == Bj[e])) || (Bi != none && (Bi == Aj[e] || Bi == Bj[e])));
            any |= hit[e];
        }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (__any(any)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long j = jq + e;
                if (hit[e]) {
                    const int oj = p.ord[j];
                    if (j == i && oi < 0) {
                        v[e] = static_cast<float>(0.5 + 0.5 * static_cast<double>(h(Ai, Bi)));
                    } else {
                        const double sc = (oi < 0 ? 0.5 : 1.0) * (oj < 0 ? 0.5 : 1.0);
                        v[e] = combine(h(Ai, Aj[e]), h(Ai, Bj[e]), h(Bi, Aj[e]), h(Bi, Bj[e]), (oi & kOrdMask) > (oj & kOrdMask), sc);
                    }
                }
            }
        }
        *reinterpret_cast<float4 *>(orowp + jq) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Psi_1 = 1/2 I over the top founders (src/compute.jl:271-274); buffer pre-zeroed.
__global__ void half_identity_kernel(float *m, long long ld, int n, const int *out_rows, int n_rows, int row_begin)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const int r = row_begin + k;                 // matrix row
    const long long orow = out_rows ? out_rows[k] : r;
    if (r < n) m[orow * ld + r] = 0.5f;
}

// point lookups in the resident result (genphi_result_entries): out[k] = m[off[k]] widened
__global__ void gather_entries_kernel(const float *__restrict__ m, const long long *__restrict__ off, long long n,
                                      double *__restrict__ out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = static_cast<double>(m[off[k]]);
}

// phiMean support: Float64 sum of each resident row and its diagonal entry (row r0 + k holds
// proband r0 + k).  One workgroup per row, fixed summation order => reproducible.
__global__ void row_sums_kernel(const float *m, long long ld, int n, int row_begin, double *row_sum, double *diag)
{
    __shared__ double part[256];
    const int k = blockIdx.x;
    const float *row = m + (long long)k * ld;
    double acc = 0.0;
    for (int j = threadIdx.x * 4; j < n; j += blockDim.x * 4) {       // ld is a multiple of 64 and columns >= n are zero
        const float4 v = *reinterpret_cast<const float4 *>(row + j);
        acc += (static_cast<double>(v.x) + static_cast<double>(v.y)) + (static_cast<double>(v.z) + static_cast<double>(v.w));
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = blockDim.x >> 1; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) { row_sum[k] = part[0]; diag[k] = static_cast<double>(row[row_begin + k]); }
}

// out[k][c] = in[perm[row_begin + k]][perm[c]]: the last level of a WIDE step, computed in storage
// order, delivered in proband order
// A per-element global gather runs at ~0.4 TB/s (one cache line per lane).  Here a workgroup
// owns (row, column chunk): it keeps the chunk's perm words and output values in registers,
// stages the source row through LDS one segment at a time (16-byte coalesced loads), picks the
// elements that fall in the segment, and writes the chunk with 16-byte coalesced stores.
template <int CPT>
__global__ void __launch_bounds__(1024)
colperm_kernel(const float *__restrict__ in, float *__restrict__ out, long long ld, int n, const int *__restrict__ perm,
               int n_chunks, int seg_floats, int n_segs, int row_begin)
{
    extern __shared__ float lds[];
    static_assert(CPT % 4 == 0, "columns are handled in quads");
    constexpr int NQ = CPT / 4;
    const int row = blockIdx.x / n_chunks, chunk = blockIdx.x - row * n_chunks;
    const int tid = threadIdx.x;
    const long long c0 = (long long)chunk * CPT * 1024;
    const float *src = in + (long long)perm[row_begin + row] * ld;      // rows too are in storage order
    int pidx[CPT];
    float val[CPT];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const long long j = c0 + ((long long)q * 1024 + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pidx[4 * q + e] = (j + e < n) ? perm[j + e] : -1;       // -1: padding column, stays 0
            val[4 * q + e] = 0.f;
        }
    }
    for (int sg = 0; sg < n_segs; ++sg) {
        const int base = sg * seg_floats;
        const int len = min(seg_floats, (int)ld - base);             // ld is a multiple of 64
        __syncthreads();                                            // previous segment fully consumed
        stage_row<4>(lds, src + base, len >> 2, tid, 1024);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const unsigned t = (unsigned)(pidx[k] - base);
            if (t < (unsigned)len) val[k] = lds[t];
        }
    }
    float *dst = out + (long long)row * ld;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const long long j = c0 + ((long long)q * 1024 + tid) * 4;
        if (j < ld) *reinterpret_cast<float4 *>(dst + j) = make_float4(val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]);
    }
}


// The same delivery, persistent and software-pipelined (round 4; rows of up to 45,056 result columns from source rows of < 65,535 floats):
// the column permutation is the SAME for every row, so a workgroup loads it once -- packed two 16-bit source columns per register --
// (512 threads x 88 columns: 44 + 88 + 72 staging registers of the 256 a thread has at two waves per SIMD; with 1024 threads the 128 do not hold
// them) and then walks rows blockIdx.x, + gridDim.x, ...: ONE flat loop over (row, segment) stages in which the next stage's piece of a source
// row is already in flight into registers while the current one is gathered from LDS and the finished row is stored (colperm_kernel has
// one workgroup per row and CU -- LDS -- and nothing overlaps its loads: 3.8 TB/s on the 41.5k x 41.5k result of cfg2all).
template <int NT, int CPT, int STG>
__global__ void __launch_bounds__(NT)
colperm_pipe_kernel(const float *__restrict__ in, float *__restrict__ out, long long ld, int n, const int *__restrict__ perm,
                    int seg_floats, int n_segs, int row_begin, int n_rows)
{
    extern __shared__ float lds[];
    static_assert(CPT % 4 == 0, "columns are handled in quads");
    constexpr int NQ = CPT / 4;
    const int tid = threadIdx.x;
    unsigned pp[CPT / 2];                                           // source column of result column k: 16 bits each, 0xffff = padding (stays 0)
    float val[CPT];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const long long j = ((long long)q * NT + tid) * 4;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { w[e] = (j + e < n) ? static_cast<unsigned>(perm[j + e]) : 0xffffu; val[4 * q + e] = 0.f; }
        pp[2 * q] = w[0] | (w[1] << 16); pp[2 * q + 1] = w[2] | (w[3] << 16);
    }
    int row = blockIdx.x;
    if (row >= n_rows) return;
    f4_t pre[STG];
    auto fetch = [&](int r, int sg) {                              // the piece [sg * seg_floats, ...) of result row r's source row
        const float *src = in + (long long)perm[row_begin + r] * ld + (long long)sg * seg_floats;
        const int nvec = (min(seg_floats, (int)ld - sg * seg_floats)) >> 2;
#pragma unroll
        for (int k = 0; k < STG; ++k) pre[k] = *reinterpret_cast<const f4_t *>(src + 4 * min(tid + k * NT, nvec - 1));      // (clamped: unconditional loads)
    };
    fetch(row, 0);
    int sg = 0;
    for (;;) {
        const int base = sg * seg_floats;
        const int len = min(seg_floats, (int)ld - base);           // ld is a multiple of 64
        __syncthreads();                                            // the previous stage's gathers are done with the buffer
#pragma unroll
        for (int k = 0; k < STG; ++k) { const int q4 = tid + k * NT; if (4 * q4 < len) *reinterpret_cast<f4_t *>(lds + 4 * q4) = pre[k]; }
        // the next stage: the next piece of this row, or the first piece of the workgroup's next row
        const bool row_done = sg + 1 == n_segs;
        const int nrow = row_done ? row + (int)gridDim.x : row, nsg = row_done ? 0 : sg + 1;
        if (nrow < n_rows) fetch(nrow, nsg);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            unsigned w = pp[k >> 1];
            asm volatile("" : "+v"(w));                             // (opaque: the unpacked halves are loop-invariant and would be hoisted into 88 registers)
            const unsigned t = ((w >> ((k & 1) * 16)) & 0xffffu) - (unsigned)base;
            if (t < (unsigned)len) val[k] = lds[t];
        }
        if (row_done) {
            float *dst = out + (long long)row * ld;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const long long j = ((long long)q * NT + tid) * 4;
                if (j < ld) __builtin_nontemporal_store(f4_t{val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]}, reinterpret_cast<f4_t *>(dst + j));
                val[4 * q] = val[4 * q + 1] = val[4 * q + 2] = val[4 * q + 3] = 0.f;
            }
            if (nrow >= n_rows) break;
        }
        row = nrow; sg = nsg;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side: plan object, device state, C-ABI
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }

using genphi::PhaseTrace;      // GENPHI_TRACE=1: wall-clock marks of the phases of a call on stderr (planner.h)
int genphi_set_error(int code, const std::string &msg) { return fail(code, msg); }   // for loader.cpp

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(GENPHI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

// level matrices carry a zeroed tail so that the SPLIT kernel's unconditional staging loads
// (STG * 1024 float4 per row, <= 160 KB) may run past the last row
constexpr size_t kTailPadFloats = 64 * 1024;
// pk / ord are padded so that the unrolled per-thread column loops need no clamp
constexpr size_t kIdxPad = 32 * 1024;
// ... which only the SPLIT kernels do (a chunk of up to 24 x 1024 columns whatever the cut's width); the FULL kernel reads whole quads
// (a deep pedigree of 200 tiny levels carried 52 MB of padding: 11 of the 14 ms of its first call)
static size_t idx_pad(const LevelStep &s) { return s.mode == genphi::kModeSplit ? kIdxPad : 1024; }

// Tuning, A/B and test hooks.  Every one is an environment variable that is read ONCE, when the plan is
// created (genphi_plan_create), and kept in the plan: a plan never changes behaviour under the caller's
// feet, and no launch path calls getenv.  The list (with what each is for) is in README.md,
// "Environment hooks"; none is needed in production.
struct Tuning {
    int lds_cap_floats = 0;        // GENPHI_LDS_CAP_FLOATS   test: LDS budget for staged rows (forces SPLIT / WIDE on small inputs)
    int full_max_floats = -1;      // GENPHI_FULL_MAX_FLOATS  tuning: FULL vs SPLIT threshold (row length in floats)
    bool no_stay = false;          // GENPHI_NO_STAY          A/B + test: WIDE levels never stay in place (every level is copied into the other buffer)
    int stay_max_slots = 0;        // GENPHI_STAY_MAX_SLOTS   test: largest slot capacity of an in-place run (default: planner.h)
    int stay_headroom = -1;        // GENPHI_STAY_HEADROOM    tuning: extra blocks of free slots per in-place run (longer runs, more memory)
    int stay_mem_pct = 0;          // GENPHI_STAY_MEM_PCT     test: in-place runs may need this % of the plain buffers' memory (default 120)
    int stay_min_ratio_pct = -1;   // GENPHI_STAY_MIN_RATIO_PCT tuning: a step stays in place while cut >= this % of its new members (default 200)
    int stay_slack_pct = -1;       // GENPHI_STAY_SLACK_PCT   tuning: free slots beyond the widest (cut + new members) of an in-place run, in % (default 6)
    int stay_narrow = -1;          // GENPHI_STAY_NARROW      A/B + test: 0 = only levels whose rows do not fit in LDS stay in place (the round-3 behaviour); 2 = in place wherever the ratio test allows, whatever the cost model says
    int stay_family = -1;          // GENPHI_STAY_FAMILY      A/B: 0 = new members of a leaving class in rank order instead of by family
    bool colperm_plain = false;    // GENPHI_COLPERM_PLAIN    A/B + test: the proband-order pass by the one-workgroup-per-row kernel (rounds 1-3)
    int stay_last = -1;            // GENPHI_STAY_LAST        A/B + test: 0 = the proband cut never stays in place (the step that reads a run's last cut compacts it,
                                   //                         then the proband-order pass: the form of rounds 3 and early 4)
    int stay_overhead_k = -1;      // GENPHI_STAY_OVERHEAD_K  tuning + test: fixed cost of a block-assembled step in the planner's cost model, in thousands of
                                   //                         matrix entries (default 64000; tests that put tiny cuts in place set 0)
    int stay_narrow_min = -1;      // GENPHI_STAY_NARROW_MIN  tuning + test: narrowest source cut of an in-place step at FULL / SPLIT widths (default 2048)
    int stay_tile = 0;             // GENPHI_STAY_TILE        tuning: columns per tile of the fused in-place kernel, 256 or 128 (default: by the launch's size)
    bool stay_scalar_t = false;    // GENPHI_STAY_SCALAR_T    A/B: the fused kernel writes its transposed tile with 4-byte stores (the round-3 form) instead of 16-byte ones
    bool stay_col_fastest = false; // GENPHI_STAY_COL_FASTEST A/B: fused kernel's workgroups ordered column-fastest instead of granule-fastest (same columns together)
    bool stay_two_pass = false;    // GENPHI_STAY_TWO_PASS    A/B + test: new x dragged and its transpose as two kernels (rows_avg + transpose_slots) instead of the fused one
    bool stay_scatter = false;     // GENPHI_STAY_SCATTER     A/B + test: the new x new block of an in-place step always goes through the compact buffer
    int max_group = 8;             // GENPHI_MAX_GROUP        tuning: children per segment of the SPLIT work lists (<= 8; <= 4 where rank masks are kept)
    int max_run = 1;               // GENPHI_MAX_RUN          tuning: stages per run of the hub walk.  1 (default): a run is one hub and its children;
                                   //                         larger: the walk chains from hub to hub (16-20 % fewer staged rows, measured no faster:
                                   //                         profiles/microbench/out/r03_ab_hub_walk_*.out)
    int full_bs = 0;               // GENPHI_FULL_BS          tuning: workgroup size of level_full_kernel
    bool no_identity = false;      // GENPHI_NO_IDENTITY      test: level step 0 on a materialised 1/2 I
    int cert_min_exp = -27;        // GENPHI_CERT_MIN_EXP     test: certificate threshold 2^e, e in [-27, 0] (always safe)
    int dbg_step = -1;             // GENPHI_DBG_STEP         GENPHI_WG_TIMES builds: the step whose workgroup timing is recorded
    bool no_fast = false;          // GENPHI_NO_FAST          test / A-B: grouping-exact SPLIT / FULL bodies only
    int max_cpt = 0;               // GENPHI_MAX_CPT          test / tuning: columns per thread of a SPLIT chunk
    int fast_nt = 0;               // GENPHI_FAST_NT          test / tuning: 512- or 1024-thread certified-rows kernel
    char wide_route = 0;           // GENPHI_WIDE_ROUTE       A-B: 'A' / 'B' route of the WIDE levels (0 = by cost)
    bool tt_noalign = false;       // GENPHI_TT_NOALIGN       A-B: transpose without line-aligned destination runs
    bool no_shard_prune = false;   // GENPHI_NO_SHARD_PRUNE   test: a row shard computes every row of the upper levels
    int shard_force_step = -1, shard_force_row = -1;   // GENPHI_SHARD_FORCE "step:row"  debugging aid
    int shard_prune_min_step = 0;  // GENPHI_SHARD_PRUNE_MIN_STEP  debugging aid
    bool no_small = false;         // GENPHI_NO_SMALL         test: no fused small-level runs
    bool no_graph = false;         // GENPHI_NO_GRAPH         A-B: never replay a captured hipGraph
    int d2h_threads = 0;           // GENPHI_D2H_THREADS      tuning: worker threads of genphi_result_to_host
    bool d2h_pageable = false;     // GENPHI_D2H_PAGEABLE     A-B: no pinned staging ring
    int d2h_sym = -1;              // GENPHI_D2H_SYM          opt-in: 1 = a full result crosses the link as upper-triangle tiles + a host mirror pass (default: every entry is copied)
    int d2h_tile_rows = 0, d2h_tile_cols = 0;   // GENPHI_D2H_TILE "RxC"  test + tuning: tile of the symmetric copy (default 256 x 8192)
    int fail_alloc_at = 0;         // GENPHI_TEST_FAIL_ALLOC  test: the k-th device allocation of an upload fails (error-path test)
    int sparse_k = -2;             // GENPHI_SPARSE_K         A/B + test: last cut kept as row lists (sparse_levels.h): -1 = none (every level dense), k >= 0 = cuts 0..k
                                   //                         whatever their density (clamped to the eligible steps); default: by the calibration run's counts
    int sparse_permille = -1;      // GENPHI_SPARSE_PERMILLE  tuning: a cut stays sparse while at most this share (1/1000) of its entries is non-zero
    int sparse_min_cut = -1;       // GENPHI_SPARSE_MIN_CUT   tuning + test: ... and only when a cut of the sparse run has this many members
    int sparse_chunk = 0;          // GENPHI_SPARSE_CHUNK     tuning: columns per workgroup of the sparse -> dense step
    int sparse_batch = 0;          // GENPHI_SPARSE_BATCH     A/B: list entries in flight per thread of a long row's workgroup, 4 or 8 (default 4; 8 measured slower)
    int sparse_arena = 0;          // GENPHI_SPARSE_ARENA     test: entries the row-list arenas start with (default 16 Mi; small values exercise their growth)
    int d2h_chunk_mb = 0;          // GENPHI_D2H_CHUNK_MB     tuning: size of a pinned staging chunk of genphi_result_to_host (default 16, 4 for results below 2 GB)
    int sparse_classes = -1;       // GENPHI_SPARSE_CLASSES   A/B + test: 1 / 0 = a row-list step is always / never one launch per class of row lengths (default: where lengths differ much)
};

// A set of "GENPHI_NAME" -> value settings handed to genphi_plan_create_tuned (include/genphi.h): the same knobs without the environment.
struct genphi_tuning {
    std::map<std::string, std::string> kv;
};

// every hook name a Tuning understands (genphi_tuning_set refuses anything else)
static const char *const kTuningNames[] = {
    "GENPHI_LDS_CAP_FLOATS", "GENPHI_FULL_MAX_FLOATS", "GENPHI_NO_STAY", "GENPHI_STAY_MAX_SLOTS", "GENPHI_STAY_HEADROOM", "GENPHI_STAY_MEM_PCT",
    "GENPHI_STAY_SCATTER", "GENPHI_STAY_TWO_PASS", "GENPHI_STAY_COL_FASTEST", "GENPHI_STAY_SCALAR_T", "GENPHI_STAY_TILE", "GENPHI_STAY_SLACK_PCT",
    "GENPHI_STAY_MIN_RATIO_PCT", "GENPHI_STAY_NARROW", "GENPHI_STAY_NARROW_MIN", "GENPHI_STAY_OVERHEAD_K", "GENPHI_STAY_LAST", "GENPHI_COLPERM_PLAIN",
    "GENPHI_STAY_FAMILY", "GENPHI_MAX_GROUP", "GENPHI_MAX_RUN", "GENPHI_FULL_BS", "GENPHI_NO_IDENTITY", "GENPHI_CERT_MIN_EXP", "GENPHI_DBG_STEP",
    "GENPHI_NO_FAST", "GENPHI_MAX_CPT", "GENPHI_FAST_NT", "GENPHI_WIDE_ROUTE", "GENPHI_TT_NOALIGN", "GENPHI_NO_SHARD_PRUNE", "GENPHI_SHARD_FORCE",
    "GENPHI_SHARD_PRUNE_MIN_STEP", "GENPHI_NO_SMALL", "GENPHI_NO_GRAPH", "GENPHI_D2H_THREADS", "GENPHI_D2H_PAGEABLE", "GENPHI_D2H_SYM", "GENPHI_D2H_TILE",
    "GENPHI_D2H_CHUNK_MB", "GENPHI_TEST_FAIL_ALLOC", "GENPHI_SPARSE_K", "GENPHI_SPARSE_PERMILLE", "GENPHI_SPARSE_MIN_CUT", "GENPHI_SPARSE_CHUNK", "GENPHI_SPARSE_CLASSES", "GENPHI_SPARSE_BATCH", "GENPHI_SPARSE_ARENA"};

// the settings of a plan: from a genphi_tuning when one is given, else from the environment -- which the library reads only under
// GENPHI_ENV_HOOKS=1 (planner.h: env_hook)
static Tuning tuning_from(const genphi_tuning *tu)
{
    Tuning t;
    auto look = [tu](const char *name) -> const char * {
        if (tu) {
            auto it = tu->kv.find(name);
            return it == tu->kv.end() ? nullptr : it->second.c_str();
        }
        return genphi::env_hook(name);
    };
    auto geti = [&](const char *name, int dflt) { const char *e = look(name); return e ? std::atoi(e) : dflt; };
    auto has = [&](const char *name) { return look(name) != nullptr; };
    t.lds_cap_floats = geti("GENPHI_LDS_CAP_FLOATS", 0);
    t.full_max_floats = geti("GENPHI_FULL_MAX_FLOATS", -1);
    t.no_stay = geti("GENPHI_NO_STAY", 0) != 0;
    t.stay_max_slots = geti("GENPHI_STAY_MAX_SLOTS", 0);
    t.stay_headroom = geti("GENPHI_STAY_HEADROOM", -1);
    t.stay_mem_pct = geti("GENPHI_STAY_MEM_PCT", 0);
    t.stay_scatter = geti("GENPHI_STAY_SCATTER", 0) != 0;
    t.stay_two_pass = geti("GENPHI_STAY_TWO_PASS", 0) != 0;
    t.stay_col_fastest = geti("GENPHI_STAY_COL_FASTEST", 0) != 0;
    t.stay_scalar_t = geti("GENPHI_STAY_SCALAR_T", 0) != 0;
    { const int v = geti("GENPHI_STAY_TILE", 0); t.stay_tile = v == 128 ? 128 : (v == 256 ? 256 : 0); }
    t.stay_slack_pct = geti("GENPHI_STAY_SLACK_PCT", -1);
    t.stay_min_ratio_pct = geti("GENPHI_STAY_MIN_RATIO_PCT", -1);
    t.stay_narrow = geti("GENPHI_STAY_NARROW", -1);
    t.stay_narrow_min = geti("GENPHI_STAY_NARROW_MIN", -1);
    t.stay_overhead_k = geti("GENPHI_STAY_OVERHEAD_K", -1);
    t.stay_last = geti("GENPHI_STAY_LAST", -1);
    t.colperm_plain = has("GENPHI_COLPERM_PLAIN");
    t.stay_family = geti("GENPHI_STAY_FAMILY", -1);
    t.max_group = std::max(1, geti("GENPHI_MAX_GROUP", 8));
    t.max_run = std::max(1, geti("GENPHI_MAX_RUN", 1));
    t.full_bs = geti("GENPHI_FULL_BS", 0);
    t.no_identity = has("GENPHI_NO_IDENTITY");
    t.cert_min_exp = geti("GENPHI_CERT_MIN_EXP", -27);
    t.dbg_step = geti("GENPHI_DBG_STEP", -1);
    t.no_fast = has("GENPHI_NO_FAST");
    t.max_cpt = geti("GENPHI_MAX_CPT", 0);
    t.fast_nt = geti("GENPHI_FAST_NT", 0);
    if (const char *e = look("GENPHI_WIDE_ROUTE")) t.wide_route = (e[0] == 'B' || e[0] == 'b') ? 'B' : 'A';
    t.tt_noalign = has("GENPHI_TT_NOALIGN");
    t.no_shard_prune = has("GENPHI_NO_SHARD_PRUNE");
    if (const char *e = look("GENPHI_SHARD_FORCE")) {
        int fs = -1, fr = -1;
        if (std::sscanf(e, "%d:%d", &fs, &fr) == 2) { t.shard_force_step = fs; t.shard_force_row = fr; }
    }
    t.shard_prune_min_step = geti("GENPHI_SHARD_PRUNE_MIN_STEP", 0);
    t.no_small = has("GENPHI_NO_SMALL");
    t.no_graph = has("GENPHI_NO_GRAPH");
    t.d2h_threads = geti("GENPHI_D2H_THREADS", 0);
    t.d2h_pageable = has("GENPHI_D2H_PAGEABLE");
    t.d2h_sym = geti("GENPHI_D2H_SYM", -1);
    if (const char *e = look("GENPHI_D2H_TILE")) {
        int r = 0, c = 0;
        if (std::sscanf(e, "%dx%d", &r, &c) == 2 && r >= 1 && c >= 1) { t.d2h_tile_rows = r; t.d2h_tile_cols = c; }
    }
    t.fail_alloc_at = geti("GENPHI_TEST_FAIL_ALLOC", 0);
    t.sparse_k = geti("GENPHI_SPARSE_K", -2);
    t.sparse_permille = geti("GENPHI_SPARSE_PERMILLE", -1);
    t.sparse_min_cut = geti("GENPHI_SPARSE_MIN_CUT", -1);
    t.sparse_chunk = geti("GENPHI_SPARSE_CHUNK", 0);
    t.sparse_classes = geti("GENPHI_SPARSE_CLASSES", -1);
    t.d2h_chunk_mb = geti("GENPHI_D2H_CHUNK_MB", 0);
    t.sparse_batch = geti("GENPHI_SPARSE_BATCH", 0);
    t.sparse_arena = geti("GENPHI_SPARSE_ARENA", 0);
    return t;
}

// Work lists of a SPLIT launch: the hub walk of planner.h (WalkLists) as device arrays.
//   desc : per work row (storage row, output row, B source, rank word)
//   seg  : per segment (first work row, hub row, leading rows without B source, type) + terminators
//   run  : first segment of every run + terminator
// level_split_fast_kernel takes runs as its items, level_split_kernel (grouping-exact: it keeps one 32-bit rank
// mask per child in 4 VGPRs, hence segments of at most 4 children) single segments.
struct GroupLists {
    genphi::WalkLists w;
};
struct DeviceGroups {
    int4 *desc = nullptr, *seg = nullptr;
    int4 *run = nullptr;       // (first segment, hub row | n0 << 16, its rows [z, w)) per run + terminator
    int n_segs = 0, n_runs = 0;
};

struct DeviceStep {
    int *srcA = nullptr, *srcB = nullptr, *ord = nullptr, *work = nullptr;
    unsigned *pk = nullptr;
    DeviceGroups groups;       // SPLIT
    // WIDE: row descriptors of rows_compact_kernel for the cut's own rows (A, B, row, scale exponent)
    // and for the parent rows of Psi_P; the parents' positions; the new rows as a work list
    int4 *rowdesc = nullptr, *pardesc = nullptr;
    int *parents = nullptr, *newrows = nullptr;
    int *pstart = nullptr;     // drag_rows_kernel: first parent of each chunk of 8192 dragged columns
    int *idx = nullptr;        // source column of every dragged member (= srcA; the sources' SLOTS when the source cut is stored by slot:
                               // then rowdesc / pardesc / parents hold slots too, see LevelStep::src_slots)
    int *blk_slot = nullptr;   // (in-place steps) first slot of every granule of 64 new members
    int2 *tiles = nullptr;     // (in-place steps) the kFT-column tiles of the slot ranges that hold dragged members: (first column, end of the range)
    int n_tiles = 0;
    int tile_cols = 256;       // ... their width: 128 for small launches (more, smaller workgroups: cfg3s -2.4 %), 256 otherwise (GENPHI_STAY_TILE forces one)
    int nn = -1;               // index of the new x new sub-step in genphi_plan::nn_steps / nn_dsteps
};

static void build_groups(const LevelStep &s, const int *rows, const int *out_rows, int n_rows, GroupLists &gl, const Tuning &tun)
{
    // segments are capped -- the grouping-exact kernel keeps one 32-bit rank mask per child in 4 VGPRs, so <= 4 children where it
    // needs them (cut not in rank order), <= 8 where the position test replaces them -- and so are runs: a workgroup walks a
    // run's stages one after the other, so one huge run would be a serial tail
    genphi::build_hub_walk(s.srcA.data(), s.srcB.data(), s.ord.data(), static_cast<int32_t>(s.n_prev), rows, out_rows, n_rows,
                           std::min(tun.max_group, s.pos_ord ? 8 : 4), tun.max_run, gl.w);
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }
static size_t groups_bytes(const GroupLists &gl)
{
    return al256(gl.w.desc4.size() * sizeof(int)) + al256(gl.w.seg4.size() * sizeof(int)) + al256(gl.w.run.size() * sizeof(int));
}
// put(src, bytes) copies into the host image of a device blob and returns the device address
template <class Put>
static void put_groups(const GroupLists &gl, DeviceGroups &d, Put &&put)
{
    d.desc = reinterpret_cast<int4 *>(put(gl.w.desc4.data(), gl.w.desc4.size() * sizeof(int)));
    d.seg = reinterpret_cast<int4 *>(put(gl.w.seg4.data(), gl.w.seg4.size() * sizeof(int)));
    d.run = reinterpret_cast<int4 *>(put(gl.w.run.data(), gl.w.run.size() * sizeof(int)));
    d.n_segs = static_cast<int>(gl.w.seg4.size() / 4) - 2;
    d.n_runs = static_cast<int>(gl.w.run.size() / 4) - 1;
}

struct genphi_plan {
    Plan plan;
    genphi::PlanOptions popt;
    Tuning tun;                                  // environment hooks, read once in genphi_plan_create
    // device state (created lazily by the first compute)
    bool on_device = false;
    int device = -1;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    char *idx_blob = nullptr;
    std::vector<char *> lazy_blobs;              // walk lists uploaded after the index blob (ensure_groups)
    size_t lazy_bytes = 0;
    size_t idx_blob_bytes = 0, psi_p_floats = 0, nn_tmp_floats = 0;      // (genphi_plan_device_bytes)
    std::vector<DeviceStep> dsteps;
    std::vector<const LevelStep *> nn_steps;     // new x new sub-steps of the WIDE steps (owned by the plan's steps)
    std::vector<DeviceStep> nn_dsteps;
    float *psi_p = nullptr;                      // WIDE: compacted parent matrix Psi[parents][parents]
    float *nn_tmp = nullptr;                     // in-place WIDE steps: the new x new block before it is scattered to the new members' slots
    int *d_cert_t = nullptr;                     // ... and its rows' certificate words
    int *d_cert_p = nullptr;                     // certificates of the rows of psi_p
    size_t cert_p_words = 0;
    int *d_final_perm = nullptr;
    int *d_final_slots = nullptr;                // (the proband cut stayed in place) slot of every proband, result order
    int *d_shard_rows = nullptr, *d_shard_out_rows = nullptr;
    int *d_queues = nullptr;        // 16 work-queue counters per slot (a level step or a new x new sub-step)
    size_t sweep_words = 0;         // ints of the array that holds d_queues, d_gcnt and d_cert (cleared by ONE memset per sweep)
    size_t n_slots = 0;
    SmallStep *d_small = nullptr;   // one entry per level step (levels_small_kernel)
    hipGraphExec_t graph_exec = nullptr;   // captured sweep (see genphi_compute_device)
    // key = (kernel, r0, r1, need_perm, alloc_gen).  A captured graph bakes raw device pointers into
    // its kernel nodes, so every (re)allocation of a buffer the sweep touches bumps alloc_gen:
    // the stale graph can then never be replayed, and a fresh eager run precedes the next capture.
    long long graph_key[5] = {0, 0, 0, 0, 0}, eager_key[5] = {0, 0, 0, 0, 0};
    long long alloc_gen = 1;
    bool level_bufs_ready = false;         // buf[] / psi_p exist (ensure_level_buffers)
    int level_bufs_from = 0;               // ... buf[] sized for the dense matrices of cuts >= this one (the cuts before it live as row lists)
    int alloc_count = 0;                   // device allocations of uploads so far (GENPHI_TEST_FAIL_ALLOC)
    char *scratch = nullptr;               // genphi_result_sums / _entries staging (grown on demand)
    size_t scratch_bytes = 0;
    bool eager_valid = false;
    DeviceGroups shard_groups;      // SPLIT lists of the last step restricted to the shard (arrays inside d_shard_blob)
    char *d_shard_blob = nullptr;
    size_t shard_blob_bytes = 0;
    // exactness certificates: one word per row of every level matrix (cert_off[c] = first word of
    // cut c), the per-launch group flags and [certified, other] group counts per step
    int *d_cert = nullptr, *d_glist = nullptr, *d_gcnt = nullptr;
    std::vector<size_t> cert_off;
    size_t cert_words = 0;
    // Row shards: an upper level only needs the rows its shard's last-level rows descend from
    // (~80 % of a cut at 8 shards of cfg4, 55-72 % in the two levels below the last).  Per
    // intermediate step: restricted work list (+ SPLIT descriptors) of the current shard.
    struct ShardStep { int *rows = nullptr; DeviceGroups groups; int n_rows = 0; };
    std::vector<ShardStep> sh_steps;
    char *sh_blob = nullptr;
    bool sh_valid = false;
    int64_t shard_cap = 0, shard_r0 = -1, shard_r1 = -1;
    float *buf[2] = {nullptr, nullptr};
    size_t buf_floats[2] = {0, 0};
    // which of the two level buffers holds cut c: alternating, except that a WIDE step that stays in place (LevelStep::stay)
    // writes into its source's buffer.  buf_of[0]: with in-place steps (the product sweep), buf_of[1]: plain alternation (the
    // per-entry kernel = 1 sweep, which knows no slots).  cert_cut[c]: the cut whose certificate words cut c shares in the
    // product sweep (the entry cut of its in-place run; c itself otherwise).
    std::vector<int> buf_of[2], cert_cut;
    bool stay_active = false;                    // this sweep keeps WIDE levels in place (set per compute call)
    genphi_step_fn step_hook = nullptr;          // genphi_plan_set_step_hook: called before every level step of a Float32 sweep
    void *step_hook_user = nullptr;
    float *result = nullptr, *final_tmp = nullptr;
    // Float64-storage sweeps (gen.f, pairwise phi): own level matrices and result
    double *buf64[2] = {nullptr, nullptr}, *result64 = nullptr;
    size_t buf64_doubles[2] = {0, 0}, result64_doubles = 0;
    int *d_perm_rows = nullptr;                  // storage member of each resident proband row (f64 sweeps)
    size_t perm_rows_cap = 0;
    bool res_f64 = false;                        // the resident result is the Float64 one
    size_t result_floats = 0, final_tmp_floats = 0;
    int64_t res_ld = 0, res_row_begin = 0, res_n_rows = 0;
    // zero-aware leading levels (sparse_levels.h): created and calibrated by the first product sweep of the plan
    genphi::SparseLevels *sparse = nullptr;
    bool sparse_tried = false;
    std::vector<hipEvent_t> events;
};

static void drop_graph(genphi_plan *p)
{
    if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
    std::memset(p->graph_key, 0, sizeof(p->graph_key));
    p->eager_valid = false;
    ++p->alloc_gen;
}

// Releases everything the plan holds on its device and returns it to the "never uploaded" state:
// every pointer nulled, every size / capacity / cache key reset, so that a later upload (same or
// another device) starts from scratch instead of trusting stale pointers.
static void free_device(genphi_plan *p)
{
    if (!p->on_device) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    drop_graph(p);
    auto release = [](auto *&ptr) { if (ptr) (void)genphi::cached_free(ptr); ptr = nullptr; };
    release(p->idx_blob); p->idx_blob_bytes = 0; p->psi_p_floats = p->nn_tmp_floats = 0;
    for (char *&b : p->lazy_blobs) release(b);
    p->lazy_blobs.clear(); p->lazy_bytes = 0;
    release(p->d_shard_rows); release(p->d_shard_out_rows); release(p->d_shard_blob);
    release(p->d_glist); p->d_cert = p->d_gcnt = nullptr; p->sweep_words = 0;      // (d_cert / d_gcnt live inside d_queues' array)
    p->shard_groups = DeviceGroups(); p->shard_blob_bytes = 0; p->cert_off.clear(); p->cert_words = 0;
    release(p->d_queues); release(p->d_small); release(p->sh_blob); release(p->scratch);
    release(p->buf[0]); release(p->buf[1]); release(p->result); release(p->final_tmp);
    release(p->psi_p); release(p->d_cert_p); release(p->nn_tmp); p->d_cert_t = nullptr;      // (d_cert_t lives inside d_cert_p's array)
    release(p->buf64[0]); release(p->buf64[1]); release(p->result64); release(p->d_perm_rows);
    p->buf64_doubles[0] = p->buf64_doubles[1] = 0; p->result64_doubles = 0; p->perm_rows_cap = 0; p->res_f64 = false;
    p->nn_steps.clear(); p->nn_dsteps.clear(); p->cert_p_words = 0;
    genphi::sparse_levels_destroy(p->sparse); p->sparse = nullptr; p->sparse_tried = false;      // (its index arrays live inside idx_blob)
    p->d_final_perm = nullptr;                       // lived inside idx_blob
    p->d_final_slots = nullptr;
    p->dsteps.clear(); p->sh_steps.clear();
    p->sh_valid = false;
    p->shard_cap = 0; p->shard_r0 = p->shard_r1 = -1;
    p->buf_floats[0] = p->buf_floats[1] = 0; p->level_bufs_ready = false;
    p->result_floats = p->final_tmp_floats = 0; p->scratch_bytes = 0;
    p->res_ld = 0; p->res_row_begin = 0; p->res_n_rows = 0;
    genphi::cached_stream_release(p->stream, p->device);
    p->stream = nullptr;
    p->on_device = false;
    p->device = -1;
}

extern "C" {

const char *genphi_last_error(void) { return g_last_error.c_str(); }
const char *genphi_version(void) { return "genphi-mi355x 0.1 (gfx950)"; }

// indices_only: cuts and per-member sources / rank words only (no pk words, work orders or walk lists): all a
// Float64-storage sweep needs (genphi_phi_pairs builds such a plan per call)
static int plan_create_impl(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                            int64_t n_pro, const int64_t *pro_ids, bool indices_only, genphi_plan **out, const genphi_tuning *tuning = nullptr)
{
    if (!out) return fail(GENPHI_ERR_ARG, "genphi_plan_create: out is NULL");
    *out = nullptr;
    genphi_plan *p = new (std::nothrow) genphi_plan();
    if (!p) return fail(GENPHI_ERR_ALLOC, "out of memory");
    p->tun = tuning_from(tuning);
    p->popt.indices_only = indices_only;
    if (p->tun.lds_cap_floats >= 16) p->popt.lds_cap_floats = p->tun.lds_cap_floats;
    if (p->tun.full_max_floats >= 0) p->popt.full_max_floats = p->tun.full_max_floats;
    p->popt.no_stay = p->tun.no_stay;
    p->popt.stay_scatter = p->tun.stay_scatter;
    if (p->tun.stay_slack_pct >= 0) p->popt.stay_slack_pct = p->tun.stay_slack_pct;
    if (p->tun.stay_min_ratio_pct >= 0) p->popt.stay_min_ratio_pct = p->tun.stay_min_ratio_pct;
    if (p->tun.stay_max_slots > 0) p->popt.stay_max_slots = p->tun.stay_max_slots;
    if (p->tun.stay_narrow >= 0) { p->popt.stay_narrow = p->tun.stay_narrow != 0; p->popt.stay_narrow_force = p->tun.stay_narrow == 2; }
    if (p->tun.stay_narrow_min >= 0) p->popt.stay_narrow_min = p->tun.stay_narrow_min;
    if (p->tun.stay_overhead_k >= 0) p->popt.stay_step_overhead = 1000.0 * p->tun.stay_overhead_k;
    if (p->tun.stay_last >= 0) p->popt.stay_last = p->tun.stay_last != 0;
    if (p->tun.stay_family >= 0) p->popt.stay_family_order = p->tun.stay_family != 0;
    if (p->tun.stay_headroom >= 0) p->popt.stay_headroom = p->tun.stay_headroom;
    if (p->tun.stay_mem_pct > 0) { p->popt.stay_mem_ratio = p->tun.stay_mem_pct / 100.0; p->popt.stay_mem_floor_bytes = 0.0; }   // (an explicit share is taken literally)
    std::string err;
    int rc;
    try {
        rc = genphi::build_plan(n_ind, ind, father, mother, n_pro, pro_ids, p->popt, p->plan, err);
    } catch (const std::bad_alloc &) {
        delete p;
        return fail(GENPHI_ERR_ALLOC, "out of memory while planning");
    }
    if (rc != GENPHI_OK) { delete p; return fail(rc, err); }
    *out = p;
    return GENPHI_OK;
}

int genphi_plan_create(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                       int64_t n_pro, const int64_t *pro_ids, genphi_plan **out)
{
    return plan_create_impl(n_ind, ind, father, mother, n_pro, pro_ids, false, out);
}

genphi_tuning *genphi_tuning_create(void) { return new (std::nothrow) genphi_tuning(); }
void genphi_tuning_destroy(genphi_tuning *t) { delete t; }
int genphi_tuning_set(genphi_tuning *t, const char *name, const char *value)
{
    if (!t || !name || !value) return fail(GENPHI_ERR_ARG, "genphi_tuning_set: null argument");
    std::string key = std::strncmp(name, "GENPHI_", 7) == 0 ? name : std::string("GENPHI_") + name;
    for (const char *k : kTuningNames)
        if (key == k) { t->kv[key] = value; return GENPHI_OK; }
    return fail(GENPHI_ERR_ARG, "genphi_tuning_set: unknown setting " + key);
}
int genphi_plan_create_tuned(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                             int64_t n_pro, const int64_t *pro_ids, const genphi_tuning *tuning, genphi_plan **out)
{
    static const genphi_tuning none;                       // (tuning == NULL: the defaults, whatever the environment says)
    return plan_create_impl(n_ind, ind, father, mother, n_pro, pro_ids, false, out, tuning ? tuning : &none);
}

int genphi_plan_levels(const genphi_plan *plan, int32_t *n_levels, const int64_t **cut_sizes,
                       const int64_t **both_counts)
{
    if (!plan) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (n_levels) *n_levels = plan->plan.n_levels;
    if (cut_sizes) *cut_sizes = plan->plan.cut_sizes.data();
    if (both_counts) *both_counts = plan->plan.both_counts.data();
    return GENPHI_OK;
}

int64_t genphi_plan_n_probands(const genphi_plan *plan) { return plan ? plan->plan.n_pro : -1; }
int genphi_plan_step_mode(const genphi_plan *plan, int32_t step)
{
    if (!plan || step < 0 || step >= static_cast<int32_t>(plan->plan.steps.size())) return -1;
    return plan->plan.steps[step].mode;
}
int genphi_plan_step_info(const genphi_plan *plan, int32_t step, int64_t *info)
{
    if (!plan || !info || step < 0 || step >= static_cast<int32_t>(plan->plan.steps.size()))
        return fail(GENPHI_ERR_ARG, "genphi_plan_step_info: bad argument");
    const LevelStep &s = plan->plan.steps[step];
    info[0] = s.mode; info[1] = s.n_dragged; info[2] = static_cast<int64_t>(s.parents.size());
    info[3] = s.mode != genphi::kModeWide ? -1 : (s.nn.empty() ? 3 : s.nn[0].mode);
    return GENPHI_OK;
}
int genphi_plan_step_slots(const genphi_plan *plan, int32_t step, int64_t *info)
{
    if (!plan || !info || step < 0 || step >= static_cast<int32_t>(plan->plan.steps.size()))
        return fail(GENPHI_ERR_ARG, "genphi_plan_step_slots: bad argument");
    const LevelStep &s = plan->plan.steps[step];
    info[0] = (s.stay ? 1 : 0) | (s.src_slots ? 2 : 0) | (s.stay && s.contig ? 4 : 0);
    info[1] = s.stay ? plan->plan.ld[step + 1] : (s.src_slots ? s.P : 0);
    info[2] = s.p0; info[3] = s.npad;
    return GENPHI_OK;
}
double genphi_plan_algorithmic_bytes(const genphi_plan *plan) { return plan ? plan->plan.algorithmic_bytes : 0.0; }
int64_t genphi_plan_device_bytes(const genphi_plan *p)
{
    if (!p || !p->on_device) return 0;
    const double b = static_cast<double>(p->idx_blob_bytes) + static_cast<double>(p->lazy_bytes) + static_cast<double>(p->shard_blob_bytes) + static_cast<double>(p->scratch_bytes) +
                     4.0 * (static_cast<double>(p->buf_floats[0]) + static_cast<double>(p->buf_floats[1]) + static_cast<double>(p->result_floats) +
                            static_cast<double>(p->final_tmp_floats) + static_cast<double>(p->psi_p_floats) + static_cast<double>(p->nn_tmp_floats) +
                            static_cast<double>(p->sweep_words)) +
                     8.0 * (static_cast<double>(p->buf64_doubles[0]) + static_cast<double>(p->buf64_doubles[1]) + static_cast<double>(p->result64_doubles)) +
                     genphi::sparse_levels_device_bytes(p->sparse);
    return static_cast<int64_t>(b);
}
int64_t genphi_plan_device_bytes_needed(const genphi_plan *p)
{
    if (!p) return 0;
    const Plan &pl = p->plan;
    const int L = pl.n_levels;
    if (L == 0) return 0;
    // what upload_plan / ensure_level_buffers / genphi_compute_device allocate for a full-result Float32 sweep of this plan
    double need[2] = {0.0, 0.0}, total = 0.0;
    int b = 0;
    for (int c = 0; c + 1 < L; ++c) {                      // (as ensure_level_buffers: each buffer serves the sweep with in-place steps AND
        if (c >= 1) b = pl.steps[c - 1].stay ? b : 1 - b;  //  the plain alternation of the per-entry sweep)
        const double rows = static_cast<double>(pl.steps[c].src_slots ? pl.steps[c].P : pl.cut_sizes[c]) + 1.0;
        const double fl = rows * static_cast<double>(pl.ld[c]) + static_cast<double>(kTailPadFloats);
        need[b] = std::max(need[b], fl);
        need[c & 1] = std::max(need[c & 1], fl);
    }
    total += need[0] + need[1];
    const double N = static_cast<double>(pl.n_pro), ldN = static_cast<double>(pl.ld[L - 1]);
    total += N * ldN;                                      // the result (its pitch is the run's when the proband cut stays in place)
    if (!pl.final_perm.empty() && pl.final_slots.empty()) total += (N + 1.0) * ldN + static_cast<double>(kTailPadFloats);
    double psi_p = 0.0, nn_tmp = 0.0, idx = 0.0;
    for (const LevelStep &st : pl.steps) {
        if (st.mode == genphi::kModeWide && !st.nn.empty())
            psi_p = std::max(psi_p, static_cast<double>(st.nn[0].n_prev + 1) * static_cast<double>(st.nn[0].ld_prev) + static_cast<double>(kTailPadFloats));
        if (st.stay) nn_tmp = std::max(nn_tmp, static_cast<double>(st.npad) * static_cast<double>(st.npad) + static_cast<double>(kTailPadFloats));
        idx += 5.0 * static_cast<double>(st.n) + 2.0 * static_cast<double>(idx_pad(st)) + (st.mode == genphi::kModeSplit ? 6.0 * static_cast<double>(st.n) : 0.0)
               + (st.mode == genphi::kModeWide ? 12.0 * static_cast<double>(st.n) : 0.0);
    }
    total += psi_p + nn_tmp + idx;
    double bytes = 4.0 * total;
    // the row-list arenas of the sparse leading cuts while they are being calibrated (sparse_levels.hip; shrunk afterwards)
    const int S = (p->tun.sparse_k == -1 || p->popt.indices_only) ? 0 : genphi::sparse_eligible_steps(pl);
    if (S >= 2) {
        double nmax = 0.0;
        for (int c = 0; c <= S; ++c) nmax = std::max(nmax, static_cast<double>(pl.cut_sizes[c]));
        bytes += 2.0 * 8.0 * std::min(0.6 * nmax * nmax + 8.0 * nmax + 1024.0, 4.0e9);
    }
    return static_cast<int64_t>(bytes);
}
int genphi_plan_sparse_levels(const genphi_plan *plan, int32_t *k_out, int64_t *nnz, int64_t *entries, int32_t cap)
{
    if (k_out) *k_out = -1;
    if (!plan || !plan->sparse) return 0;
    if (k_out) *k_out = genphi::sparse_levels_k(plan->sparse);
    std::vector<long long> v(std::max(cap, 0), -1), e(std::max(cap, 0), -1);
    const int m = genphi::sparse_levels_counts(plan->sparse, cap, v.data(), e.data(), nullptr);
    for (int c = 0; c < m; ++c) { if (nnz) nnz[c] = v[c]; if (entries) entries[c] = e[c]; }
    return m;
}
int genphi_plan_set_step_hook(genphi_plan *plan, genphi_step_fn cb, void *user)
{
    if (!plan) return fail(GENPHI_ERR_ARG, "plan is NULL");
    plan->step_hook = cb; plan->step_hook_user = user;
    return GENPHI_OK;
}

int genphi_plan_step_walk(const genphi_plan *plan, int32_t step, int64_t *n_rows, int64_t *n_segs, int64_t *n_runs, int32_t *desc4,
                          int32_t *seg4, int32_t *run4)
{
    if (!plan || step < 0 || step >= static_cast<int32_t>(plan->plan.steps.size()))
        return fail(GENPHI_ERR_ARG, "genphi_plan_step_walk: bad argument");
    const LevelStep &s = plan->plan.steps[step];
    if (s.mode != genphi::kModeSplit) return fail(GENPHI_ERR_ARG, "genphi_plan_step_walk: not a SPLIT step");
    GroupLists gl;
    build_groups(s, s.work.data(), nullptr, static_cast<int>(s.work.size()), gl, plan->tun);
    if (n_rows) *n_rows = static_cast<int64_t>(gl.w.desc4.size() / 4);
    if (n_segs) *n_segs = static_cast<int64_t>(gl.w.seg4.size() / 4) - 2;
    if (n_runs) *n_runs = static_cast<int64_t>(gl.w.run.size() / 4) - 1;
    if (desc4) std::memcpy(desc4, gl.w.desc4.data(), gl.w.desc4.size() * sizeof(int32_t));
    if (seg4) std::memcpy(seg4, gl.w.seg4.data(), gl.w.seg4.size() * sizeof(int32_t));
    if (run4) std::memcpy(run4, gl.w.run.data(), gl.w.run.size() * sizeof(int32_t));
    return GENPHI_OK;
}

void genphi_release_cached(void)
{
    genphi::sparse_phi_release_kept();
    genphi::release_cached();
}
int64_t genphi_cached_bytes(void) { return static_cast<int64_t>(genphi::cached_bytes()); }

int genphi_plan_release_device(genphi_plan *plan)
{
    if (!plan) return fail(GENPHI_ERR_ARG, "plan is NULL");
    free_device(plan);
    return GENPHI_OK;
}

void genphi_plan_destroy(genphi_plan *plan)
{
    if (!plan) return;
    free_device(plan);
    delete plan;
}

}  // extern "C"

// Device allocation of an upload; GENPHI_TEST_FAIL_ALLOC = k makes the k-th one of a plan fail (error-path test)
static hipError_t plan_malloc(genphi_plan *p, void **ptr, size_t bytes)
{
    if (p->tun.fail_alloc_at > 0 && ++p->alloc_count == p->tun.fail_alloc_at) return hipErrorOutOfMemory;
    return genphi::cached_malloc(ptr, bytes);
}

// upload the flat index arrays once
static int upload_plan_impl(genphi_plan *p, int device)
{
    if (p->on_device) {
        if (device >= 0 && device != p->device) free_device(p);
        else { HIP_TRY(hipSetDevice(p->device)); return GENPHI_OK; }
    }
    PhaseTrace trace;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GENPHI_ERR_DEVICE, "no HIP device available: the gen.phi product path has no CPU fallback");
    if (device < 0) { HIP_TRY(hipGetDevice(&device)); }
    if (device >= ndev) return fail(GENPHI_ERR_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    p->device = device;
    p->on_device = true;                              // (free_device releases whatever exists; upload_plan calls it on any failure below)
    HIP_TRY(genphi::cached_stream(&p->stream));
    {
        int cus = 0;                                  // (hipGetDeviceProperties fills a 1.5 KB struct from dozens of queries: tens of ms of a first call)
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        p->n_cus = std::max(8, cus / 8 * 8);
    }

    trace.mark("  upload: device, stream");
    const Plan &pl = p->plan;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t total = 256;
    // every step to upload: the plan's steps, then the new x new sub-steps of the WIDE ones
    p->nn_steps.clear();
    for (const LevelStep &s : pl.steps)
        if (s.mode == genphi::kModeWide && !s.nn.empty()) p->nn_steps.push_back(&s.nn[0]);
    const size_t n_main = pl.steps.size(), n_all = n_main + p->nn_steps.size();
    auto step_at = [&](size_t k) -> const LevelStep & { return k < n_main ? pl.steps[k] : *p->nn_steps[k - n_main]; };
    std::vector<GroupLists> step_groups(n_all);
    {   // the hub walks of the SPLIT steps are independent of each other: a few host threads (24 -> ~7 ms of a first call on cfg4)
        // (the leading steps that may run on row lists -- sparse_levels.h -- get their walk lists when a sweep first runs them densely:
        // ensure_groups; genea140: five of its SPLIT steps, 1-2 ms of every one-shot call, never do)
        const size_t lazy_upto = (p->tun.sparse_k == -1 || p->popt.indices_only) ? 0 : static_cast<size_t>(genphi::sparse_eligible_steps(pl));
        std::vector<size_t> split_steps;
        for (size_t k = 0; k < n_all; ++k) if (step_at(k).mode == genphi::kModeSplit && !(k < n_main && k < lazy_upto)) split_steps.push_back(k);
        std::sort(split_steps.begin(), split_steps.end(), [&](size_t a, size_t b) { return step_at(a).n > step_at(b).n; });      // the big last step first
        const int n_thr = static_cast<int>(std::min<size_t>(4, split_steps.size()));
        std::atomic<size_t> next{0};
        std::vector<char> oom(std::max(n_thr, 1), 0);
        auto work = [&](int t) {
            try {
                for (size_t q = next.fetch_add(1); q < split_steps.size(); q = next.fetch_add(1)) {
                    const LevelStep &s = step_at(split_steps[q]);
                    build_groups(s, s.work.data(), nullptr, static_cast<int>(s.work.size()), step_groups[split_steps[q]], p->tun);
                }
            } catch (const std::bad_alloc &) { oom[t] = 1; }
        };
        if (n_thr <= 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (char e : oom) if (e) return fail(GENPHI_ERR_ALLOC, "out of memory while building the work lists");
    }
    for (size_t k = 0; k < n_all; ++k) {
        const LevelStep &s = step_at(k);
        total += 3 * al(s.n * sizeof(int)) + 2 * al((s.n + idx_pad(s)) * sizeof(int));
        if (s.mode == genphi::kModeSplit && !step_groups[k].w.run.empty()) total += groups_bytes(step_groups[k]);
        if (s.mode == genphi::kModeWide)
            total += al(s.n * sizeof(int4)) + al(s.parents.size() * sizeof(int4)) + al(s.parents.size() * sizeof(int)) +
                     al((s.n - s.n_dragged) * sizeof(int)) + al((s.n_dragged / 8192 + 2) * sizeof(int)) + al(s.n_dragged * sizeof(int)) +
                     al(s.blk_slot.size() * sizeof(int)) + al((s.live_ranges.size() / 2 + static_cast<size_t>(s.stay ? s.P : 0) / 128 + 1) * sizeof(int2));
    }
    total += al(pl.final_perm.size() * sizeof(int)) + al(pl.final_slots.size() * sizeof(int));
    trace.mark("  upload: walk lists (host)");
    HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->idx_blob), total));
    p->idx_blob_bytes = total;
    trace.mark("  upload: hipMalloc index blob");
    std::vector<char> host(total, 0);
    size_t off = 0;
    auto put = [&](const void *src, size_t bytes, size_t pad_bytes = 0) -> char * {
        char *d = p->idx_blob + off;
        if (bytes) std::memcpy(host.data() + off, src, bytes);
        off += al(bytes + pad_bytes);
        return d;
    };
    p->dsteps.assign(n_main, DeviceStep());
    p->nn_dsteps.assign(n_all - n_main, DeviceStep());
    int nn_next = 0;
    for (size_t k = 0; k < n_all; ++k) {
        const LevelStep &s = step_at(k);
        DeviceStep &d = k < n_main ? p->dsteps[k] : p->nn_dsteps[k - n_main];
        d.srcA = reinterpret_cast<int *>(put(s.srcA.data(), s.n * sizeof(int)));
        d.srcB = reinterpret_cast<int *>(put(s.srcB.data(), s.n * sizeof(int)));
        d.ord = reinterpret_cast<int *>(put(s.ord.data(), s.n * sizeof(int), idx_pad(s) * sizeof(int)));
        d.work = reinterpret_cast<int *>(put(s.work.data(), s.work.size() * sizeof(int)));
        d.pk = reinterpret_cast<unsigned *>(put(s.pk.data(), s.pk.size() * sizeof(unsigned), idx_pad(s) * sizeof(unsigned)));
        if (!s.pk.empty()) {      // padding entries point both sources at the zero column
            unsigned *hp = reinterpret_cast<unsigned *>(host.data() + (reinterpret_cast<char *>(d.pk) - p->idx_blob));
            const unsigned zero_pk = static_cast<unsigned>(s.n_prev) | (static_cast<unsigned>(s.n_prev) << 16);
            for (size_t k2 = s.pk.size(); k2 < s.pk.size() + idx_pad(s); ++k2) hp[k2] = zero_pk;
        }
        if (s.mode == genphi::kModeSplit && !step_groups[k].w.run.empty()) put_groups(step_groups[k], d.groups, put);
        if (s.mode == genphi::kModeWide) {
            // (a source cut stored by slot: sources, parents and "none" are slots of its P x P matrix; a step that stays in
            // place writes row r of its cut to the member's slot)
            const int none = s.src_slots ? s.P : static_cast<int>(s.n_prev);
            const int32_t *sA = s.src_slots ? s.absA.data() : s.srcA.data(), *sB = s.src_slots ? s.absB.data() : s.srcB.data();
            const int32_t *par = s.src_slots ? s.parents_abs.data() : s.parents.data();
            std::vector<int4> rd(s.n), pd(s.parents.size());
            std::vector<int> nr(s.n - s.n_dragged);
            for (int64_t r = 0; r < s.n; ++r)              // dragged: weight 1; new: weight 1/2 (columns here are dragged: weight 1)
                rd[r] = make_int4(sA[r], sB[r], s.stay ? s.out_slots[r] : static_cast<int>(r), s.ord[r] < 0 ? -1 : 0);
            for (size_t u = 0; u < s.parents.size(); ++u) pd[u] = make_int4(par[u], none, static_cast<int>(u), 0);
            for (size_t r = 0; r < nr.size(); ++r) nr[r] = static_cast<int>(s.n_dragged + r);
            d.rowdesc = reinterpret_cast<int4 *>(put(rd.data(), rd.size() * sizeof(int4)));
            d.pardesc = reinterpret_cast<int4 *>(put(pd.data(), pd.size() * sizeof(int4)));
            d.parents = reinterpret_cast<int *>(put(par, s.parents.size() * sizeof(int)));
            d.newrows = reinterpret_cast<int *>(put(nr.data(), nr.size() * sizeof(int)));
            d.idx = reinterpret_cast<int *>(put(sA, s.n_dragged * sizeof(int)));
            d.blk_slot = reinterpret_cast<int *>(put(s.blk_slot.data(), s.blk_slot.size() * sizeof(int)));
            {   // column tiles of rows_avg_t_kernel over all live ranges (one launch per step)
                std::vector<int2> tl;
                long long cols_live = 0;
                for (size_t h = 0; h + 1 < s.live_ranges.size(); h += 2) cols_live += s.live_ranges[h + 1] - s.live_ranges[h];
                d.tile_cols = p->tun.stay_tile ? p->tun.stay_tile : ((cols_live / 256 + 1) * static_cast<long long>(s.blk_slot.size()) < 8192 ? 128 : 256);
                for (size_t h = 0; h + 1 < s.live_ranges.size(); h += 2)
                    for (int c0 = s.live_ranges[h]; c0 < s.live_ranges[h + 1]; c0 += d.tile_cols) tl.push_back(make_int2(c0, s.live_ranges[h + 1]));
                d.n_tiles = static_cast<int>(tl.size());
                d.tiles = reinterpret_cast<int2 *>(put(tl.data(), tl.size() * sizeof(int2)));
            }
            // drag_rows_kernel: parents inside the source window of each chunk of 8192 dragged columns
            const int64_t chunk = 8192, nch = (s.n_dragged + chunk - 1) / chunk;
            std::vector<int> ps(nch + 1, 0);
            size_t pi = 0;
            for (int64_t c = 1; c < nch; ++c) {
                while (pi < s.parents.size() && s.parents[pi] < s.srcA[c * chunk]) ++pi;
                ps[c] = static_cast<int>(pi);
            }
            ps[nch] = static_cast<int>(s.parents.size());
            d.pstart = reinterpret_cast<int *>(put(ps.data(), ps.size() * sizeof(int)));
            if (!s.nn.empty()) d.nn = nn_next++;
        }
    }
    p->d_final_perm = reinterpret_cast<int *>(put(pl.final_perm.data(), pl.final_perm.size() * sizeof(int)));
    p->d_final_slots = reinterpret_cast<int *>(put(pl.final_slots.data(), pl.final_slots.size() * sizeof(int)));
    trace.mark("  upload: host image");
    HIP_TRY(hipMemcpyAsync(p->idx_blob, host.data(), total, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));      // `host` goes out of scope
    trace.mark("  upload: copy to device");

    const size_t n_slots = n_all + 1;                    // queue / counter slots: one per (sub-)step
    p->n_slots = n_slots;
    {   // certificates: words [cert_off[c], cert_off[c] + n_c] belong to cut c (incl. its "none" row)
        p->cert_off.assign(pl.n_levels + 1, 0);
        size_t w = 0;
        p->cert_cut.assign(pl.n_levels + 1, 0);
        p->buf_of[0].assign(pl.n_levels, 0); p->buf_of[1].assign(pl.n_levels, 0);
        for (int c = 0; c < pl.n_levels; ++c) {
            p->cert_off[c] = w;
            // (a cut stored by slot -- the source of a step with src_slots -- has one word per slot + the "none" row P)
            const bool by_slot = c < static_cast<int>(pl.steps.size()) && pl.steps[c].src_slots;
            w += (by_slot ? static_cast<size_t>(pl.steps[c].P) : static_cast<size_t>(pl.cut_sizes[c])) + 1;
            const bool stays = c >= 1 && pl.steps[c - 1].stay;
            p->cert_cut[c] = stays ? p->cert_cut[c - 1] : c;
            if (c >= 1) { p->buf_of[0][c] = stays ? p->buf_of[0][c - 1] : 1 - p->buf_of[0][c - 1]; p->buf_of[1][c] = c & 1; }
        }
        p->cert_cut[pl.n_levels] = pl.n_levels;
        p->cert_off[pl.n_levels] = w;
        p->cert_words = w;
        // ONE array for what every sweep clears first -- the work-queue counters, the group counts, the certificate words -- so that a
        // sweep starts with one memset instead of three (a deep pedigree's whole sweep is ~190 us)
        p->sweep_words = n_slots * 16 + n_slots * 4 + std::max<size_t>(w, 1);
        HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->d_queues), p->sweep_words * sizeof(int)));
        p->d_gcnt = p->d_queues + n_slots * 16;
        p->d_cert = p->d_gcnt + n_slots * 4;
        HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->d_glist), 2 * ((static_cast<size_t>(pl.max_cut) + 64) / 64 * 64) * sizeof(int)));
    }
    {
        std::vector<SmallStep> sm(pl.steps.size() + 1);
        for (size_t k = 0; k < pl.steps.size(); ++k)
            sm[k] = SmallStep{p->dsteps[k].srcA, p->dsteps[k].srcB, p->dsteps[k].ord,
                              static_cast<int>(pl.steps[k].n_prev), static_cast<int>(pl.steps[k].n)};
        HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->d_small), sm.size() * sizeof(SmallStep)));
        HIP_TRY(hipMemcpyAsync(p->d_small, sm.data(), sm.size() * sizeof(SmallStep), hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    return GENPHI_OK;
}

// The Float32 level matrices (ping-pong buffers, the compacted parent matrix of the WIDE steps): allocated by
// the first Float32 sweep, not by the upload -- a Float64-storage sweep (gen.f, pairwise phi) never touches them.
static int ensure_level_buffers_impl(genphi_plan *p, int first_cut)
{
    const Plan &pl = p->plan;
    // ping-pong buffers for the intermediate cuts first_cut..L-2 (the cuts before live as row lists, sparse_levels.hip: genea140's widest
    // cuts -- 13,654 members, 2 x 745 MB of level buffers -- are among them; its first dense cut has 4,8xx: allocating, clearing and
    // giving back the large pair was 0.5 ms of every one-shot call, and released VRAM slows the next copies down, devcache.h)
    size_t need[2] = {0, 0};
    for (int c = first_cut; c + 1 < pl.n_levels; ++c) {
        // a cut stored by slot (the cuts of an in-place run of WIDE steps): P + 1 rows of pitch P = ld[c]
        const bool by_slot = pl.steps[c].src_slots;
        const size_t fl = static_cast<size_t>(((by_slot ? static_cast<int64_t>(pl.steps[c].P) : pl.cut_sizes[c]) + 1) * pl.ld[c]) + kTailPadFloats;
        for (int v = 0; v < 2; ++v) need[p->buf_of[v][c]] = std::max(need[p->buf_of[v][c]], fl);
    }
    for (int b = 0; b < 2; ++b) {
        if (need[b]) {
            HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->buf[b]), need[b] * sizeof(float)));
            // on the plan's own stream: hipMemset on the null stream is asynchronous to the host
            // and unordered with a non-blocking stream, so it could wipe level results later
            HIP_TRY(hipMemsetAsync(p->buf[b], 0, need[b] * sizeof(float), p->stream));
        }
        p->buf_floats[b] = need[b];
    }
    p->level_bufs_from = first_cut;
    if (p->level_bufs_ready) return GENPHI_OK;         // (only the pair was replaced: a sweep that needs earlier cuts densely after all)
    // WIDE steps: the compacted parent matrix and its rows' certificates
    {
        size_t need_p = 0, need_c = 0;
        for (const LevelStep *nn : p->nn_steps) {
            need_p = std::max(need_p, static_cast<size_t>((nn->n_prev + 1) * nn->ld_prev) + kTailPadFloats);
            need_c = std::max(need_c, static_cast<size_t>(nn->n_prev) + 1);
        }
        size_t need_t = 0;                                         // in-place steps: npad x npad block + its certificate words
        for (const LevelStep &st : pl.steps)
            if (st.stay) need_t = std::max(need_t, static_cast<size_t>(st.npad));
        if (need_t) {
            HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->nn_tmp), (need_t * need_t + kTailPadFloats) * sizeof(float)));
            HIP_TRY(hipMemsetAsync(p->nn_tmp, 0, (need_t * need_t + kTailPadFloats) * sizeof(float), p->stream));
            p->nn_tmp_floats = need_t * need_t + kTailPadFloats;
        }
        if (need_p) {
            HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->psi_p), need_p * sizeof(float)));
            HIP_TRY(hipMemsetAsync(p->psi_p, 0, need_p * sizeof(float), p->stream));
            p->psi_p_floats = need_p;
        }
        if (need_p || need_t) {                                    // (one array: an in-place step clears both in one launch)
            HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&p->d_cert_p), (need_c + need_t + 1) * sizeof(int)));
            p->d_cert_t = p->d_cert_p + need_c;
            p->cert_p_words = need_c + need_t + 1;
        }
    }
    p->level_bufs_ready = true;
    return GENPHI_OK;
}
static int ensure_level_buffers(genphi_plan *p, int first_cut)
{
    if (p->level_bufs_ready && p->level_bufs_from <= first_cut) return GENPHI_OK;
    if (p->level_bufs_ready) {                        // e.g. GENPHI_FLAG_NO_SPARSE on a plan whose sweeps ran their leading cuts on lists
        drop_graph(p);
        (void)hipStreamSynchronize(p->stream);
        for (int b = 0; b < 2; ++b) { if (p->buf[b]) (void)genphi::cached_free(p->buf[b]); p->buf[b] = nullptr; p->buf_floats[b] = 0; }
    }
    const int rc = ensure_level_buffers_impl(p, first_cut);
    if (rc != GENPHI_OK) {                            // same rule as upload_plan: no half-allocated plan survives a failure
        const std::string keep = g_last_error;
        free_device(p);
        g_last_error = keep;
    }
    return rc;
}

// A failed upload (out of memory is the expected failure of large plans) must not leave a plan that looks
// uploaded: everything allocated so far is released, so that a retry uploads from scratch.
static int upload_plan(genphi_plan *p, int device)
{
    const bool was = p->on_device && !(device >= 0 && device != p->device);
    const int rc = upload_plan_impl(p, device);
    if (rc != GENPHI_OK && !was) {
        const std::string keep = g_last_error;
        free_device(p);
        g_last_error = keep;
    }
    return rc;
}

static int ensure_floats(genphi_plan *p, float **ptr, size_t *have, size_t need)
{
    if (*have >= need && *ptr) return GENPHI_OK;
    drop_graph(p);                                   // a captured sweep points at the old buffer
    if (*ptr) { HIP_TRY(genphi::cached_free(*ptr)); *ptr = nullptr; *have = 0; }
    HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(ptr), std::max<size_t>(need, 1) * sizeof(float)));
    *have = need;
    return GENPHI_OK;
}

static int ensure_scratch(genphi_plan *p, size_t bytes)
{
    if (p->scratch_bytes >= bytes && p->scratch) return GENPHI_OK;
    if (p->scratch) { HIP_TRY(genphi::cached_free(p->scratch)); p->scratch = nullptr; p->scratch_bytes = 0; }
    const size_t want = std::max<size_t>(bytes, size_t(1) << 16);
    HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->scratch), want));
    p->scratch_bytes = want;
    return GENPHI_OK;
}

static hipError_t set_max_lds(const void *fn, size_t bytes)
{
    // dynamic LDS above 64 KB has to be opted into per kernel AND per device; remember the largest
    // request of each (device, kernel).  Plans on distinct devices may run on distinct host threads.
    struct Seen { int device; const void *fn; size_t bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (Seen &x : seen) {
        if (x.device == dev && x.fn == fn) {
            if (x.bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
            if (e == hipSuccess) x.bytes = bytes;
            return e;
        }
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e == hipSuccess) seen.push_back({dev, fn, bytes});
    return e;
}

// ---- SPLIT kernel instantiation table ---------------------------------------------------------
template <int C, int S, bool O>
static hipError_t launch_split_inst(int grid, size_t lds, hipStream_t stream, const LevelArgs &a, const int4 *desc,
                                    const int4 *grp, int *queue)
{
    hipError_t e = set_max_lds(reinterpret_cast<const void *>(level_split_kernel<1024, C, S, O>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((level_split_kernel<1024, C, S, O>), dim3(grid), dim3(1024), lds, stream, a, desc, grp, a.glist, a.pdesc, queue);
    return hipGetLastError();
}

template <bool O>
static hipError_t launch_split(int cpt, int stg, int grid, size_t lds, hipStream_t stream, const LevelArgs &a,
                               const int4 *desc, const int4 *grp, int *queue)
{
#define GENPHI_T(C, S) if (cpt <= C && stg == S) return launch_split_inst<C, S, O>(grid, lds, stream, a, desc, grp, queue)
#ifdef GENPHI_MIN_INST
    GENPHI_T(16, 2); GENPHI_T(20, 8);
#else
    GENPHI_T(8, 2); GENPHI_T(16, 2); GENPHI_T(24, 2);
    GENPHI_T(8, 4); GENPHI_T(16, 4); GENPHI_T(24, 4);
    GENPHI_T(8, 6); GENPHI_T(16, 6); GENPHI_T(20, 6); if constexpr (O) { GENPHI_T(24, 6); }
    GENPHI_T(8, 7); GENPHI_T(16, 7); GENPHI_T(20, 7); if constexpr (O) { GENPHI_T(24, 7); }
    GENPHI_T(8, 8); GENPHI_T(16, 8); GENPHI_T(20, 8);
    GENPHI_T(8, 9); GENPHI_T(16, 9);
#endif
#undef GENPHI_T
    return hipErrorInvalidValue;
}

// certified-rows kernel: 1024-thread workgroups (4 waves per SIMD, 128 VGPRs) or 512-thread ones
// (2 waves per SIMD, 256 VGPRs: the per-thread overhead is paid half as often, so a workgroup
// holds ~25 % more columns -- 4 column chunks instead of 5 for the 1e5-wide final level of cfg4)
template <int NT, int C, int S, bool CERT, bool CHAIN>
static hipError_t launch_fast_inst(int grid, size_t lds, hipStream_t stream, const LevelArgs &a, const int4 *desc,
                                   const int4 *grp, const int4 *run, int *queue)
{
    hipError_t e = set_max_lds(reinterpret_cast<const void *>(level_split_fast_kernel<NT, C, S, CERT, CHAIN>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((level_split_fast_kernel<NT, C, S, CERT, CHAIN>), dim3(grid), dim3(NT), lds, stream, a, desc, grp, run, a.glist, a.pdesc, queue);
    return hipGetLastError();
}

// columns per thread the instantiations of the certified-rows kernel afford without spilling
// (checked with -Rpass-analysis=kernel-resource-usage: 0 scratch in every one)
static int fast_max_cpt(int nt, int stg)
{
    if (nt == 1024) return stg <= 4 ? 28 : (stg <= 8 ? 24 : 20);
    return stg <= 12 ? 52 : (stg <= 16 ? 56 : 48);
}

template <bool CERT, bool CHAIN>
static hipError_t launch_fast(int nt, int cpt, int stg, int grid, size_t lds, hipStream_t stream, const LevelArgs &a,
                              const int4 *desc, const int4 *grp, const int4 *run, int *queue)
{
#define GENPHI_F(N, C, S) if (nt == N && cpt <= C && stg == S) return launch_fast_inst<N, C, S, CERT, CHAIN>(grid, lds, stream, a, desc, grp, run, queue)
#ifdef GENPHI_MIN_INST
    GENPHI_F(1024, 16, 2); GENPHI_F(1024, 24, 8); GENPHI_F(512, 56, 16);
#else
    GENPHI_F(1024, 8, 2); GENPHI_F(1024, 16, 2); GENPHI_F(1024, 28, 2);
    GENPHI_F(1024, 8, 4); GENPHI_F(1024, 16, 4); GENPHI_F(1024, 28, 4);
    GENPHI_F(1024, 8, 6); GENPHI_F(1024, 16, 6); GENPHI_F(1024, 24, 6);
    GENPHI_F(1024, 8, 7); GENPHI_F(1024, 16, 7); GENPHI_F(1024, 24, 7);
    GENPHI_F(1024, 8, 8); GENPHI_F(1024, 16, 8); GENPHI_F(1024, 24, 8);
    GENPHI_F(1024, 8, 9); GENPHI_F(1024, 20, 9);
    GENPHI_F(512, 32, 12); GENPHI_F(512, 48, 12); GENPHI_F(512, 52, 12);
    GENPHI_F(512, 32, 14); GENPHI_F(512, 56, 14);
    GENPHI_F(512, 32, 16); GENPHI_F(512, 52, 16); GENPHI_F(512, 56, 16);
    GENPHI_F(512, 32, 18); GENPHI_F(512, 48, 18);
#endif
#undef GENPHI_F
    return hipErrorInvalidValue;
}

static int block_size_for(int64_t n, const Tuning &tun)
{
    {
        const int v = tun.full_bs;
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) return v;
    }
    if (n <= 512) return 64;
    if (n <= 2048) return 256;
    return 512;       // 4 workgroups per CU overlap staging and gathers; 1024 threads measured 20 % slower (cfg3)
}

// level step 0 reads Psi_1 = 1/2 I: level_identity_kernel computes it from the indices alone
// (GENPHI_NO_IDENTITY: test hook, the regular kernels on a materialised 1/2 I)
static bool identity_source(int step, const Tuning &tun) { return step == 0 && !tun.no_identity; }

// item / n_chunks as a multiply-high: exact for item < 2^32 / n_chunks (items are < 2^31 and n_chunks
// is a handful); n_chunks == 1 has no 32-bit magic number and is flagged by 0
static unsigned chunk_magic_for(int n_chunks)
{
    return n_chunks <= 1 ? 0u : 0xffffffffu / static_cast<unsigned>(n_chunks) + 1u;
}

// bits(2^-27) - 1: entries below 2^-27 (other than 0) void a row's exactness certificate.  Test hook:
// GENPHI_CERT_MIN_EXP = e in [-27, 0] raises the bound to 2^e (always safe: fewer rows certified),
// which makes mixed certified / uncertified levels out of ordinary small pedigrees.
static unsigned cert_threshold(const Tuning &tun)
{
    const int e = std::max(-27, std::min(0, tun.cert_min_exp));
    return (static_cast<unsigned>(127 + e) << 23) - 1u;
}

// What a level launch works on: a step of the plan, or the new x new sub-step of a WIDE step.
struct LevelCtx {
    const LevelStep *s;
    const DeviceStep *d;
    int slot;                  // queue / counter slot (d_queues + 16 * slot, d_gcnt + 4 * slot)
    const int *cert_prev;      // certificates of the source rows / of the rows written (nullptr: last level)
    int *cert_out;
    bool identity;             // the source matrix is Psi_1 = 1/2 I, never materialised
    bool no_none_row;          // sub-step of a WIDE level: the "none" row belongs to the enclosing level
    bool dbg;                  // GENPHI_WG_TIMES builds: record this launch
};

static LevelCtx main_ctx(genphi_plan *p, int step)
{
    LevelCtx c;
    c.s = &p->plan.steps[step]; c.d = &p->dsteps[step]; c.slot = step;
    // (cuts of an in-place run share the words of the run's entry cut: a member keeps its slot)
    c.cert_prev = p->d_cert + p->cert_off[p->stay_active ? p->cert_cut[step] : step];
    c.cert_out = p->d_cert + p->cert_off[p->stay_active ? p->cert_cut[step + 1] : step + 1];      // (the last level's have no reader: harmless)
    c.identity = identity_source(step, p->tun);
    c.no_none_row = false;
    c.dbg = (p->tun.dbg_step >= 0 ? p->tun.dbg_step : static_cast<int>(p->plan.steps.size()) - 1) == step;     // default: the last step
    return c;
}

// Walk lists of a SPLIT step that the upload left out (a leading step that may run on row lists): built and uploaded when a sweep first
// runs the step densely.
static int ensure_groups(genphi_plan *p, int step)
{
    const LevelStep &s = p->plan.steps[step];
    DeviceStep &d = p->dsteps[step];
    if (s.mode != genphi::kModeSplit || d.groups.desc != nullptr) return GENPHI_OK;
    GroupLists gl;
    build_groups(s, s.work.data(), nullptr, static_cast<int>(s.work.size()), gl, p->tun);
    const size_t gb = groups_bytes(gl);
    char *blob = nullptr;
    HIP_TRY(plan_malloc(p, reinterpret_cast<void **>(&blob), gb));
    p->lazy_blobs.push_back(blob);
    p->lazy_bytes += gb;
    std::vector<char> img(gb, 0);
    size_t off = 0;
    auto put = [&](const void *src, size_t bytes) -> char * {
        char *dst = blob + off;
        if (bytes) std::memcpy(img.data() + off, src, bytes);
        off += al256(bytes);
        return dst;
    };
    put_groups(gl, d.groups, put);
    HIP_TRY(hipMemcpyAsync(blob, img.data(), gb, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));          // `img` goes out of scope
    return GENPHI_OK;
}

// What a FULL / SPLIT launch needs from its owner (a plan's level step, or a rank's column panel)
struct LaunchRes {
    hipStream_t stream;
    int n_cus;
    const Tuning *tun;
    int *queue;              // 16 work-queue counters, zeroed before the launch
    int *gcnt;               // 4 group counters, zeroed before the launch
    int *glist_f, *glist_s;  // device-compacted group lists of the certified-rows / grouping-exact SPLIT kernels
};

// FULL or SPLIT launch of a prepared LevelArgs.  src_width = floats of a staged source row including its zero
// column (n_prev + 1; on a panel: own + received columns + 1); width = columns a row kernel writes.
static int launch_rows(const LaunchRes &R, LevelArgs a, int mode, bool pos_ord, int64_t src_width, int64_t width, int kernel,
                       const DeviceGroups &dg)
{
    const int n_rows = a.n_rows;
    const int lds_row = static_cast<int>((src_width + 3) / 4 * 4);
    if (mode == genphi::kModeFull) {
        a.lds_row = lds_row;
        const size_t lds = 2 * static_cast<size_t>(lds_row) * sizeof(float);
        const int bs = block_size_for(std::max<int64_t>(a.n, src_width), *R.tun);
        if (pos_ord) {
            HIP_TRY(set_max_lds(reinterpret_cast<const void *>(level_full_kernel<4, true>), lds));
            hipLaunchKernelGGL((level_full_kernel<4, true>), dim3(n_rows + a.zero_row), dim3(bs), lds, R.stream, a);
        } else {
            HIP_TRY(set_max_lds(reinterpret_cast<const void *>(level_full_kernel<4, false>), lds));
            hipLaunchKernelGGL((level_full_kernel<4, false>), dim3(n_rows + a.zero_row), dim3(bs), lds, R.stream, a);
        }
    } else if (mode == genphi::kModeSplit) {
        a.lds_row = lds_row;
        const int per_row4 = lds_row / 4;                                // float4 per staged row
        const int stg1k = (per_row4 + 1023) / 1024;
        // staging instantiations; the planner keeps lds_row <= 36864 floats (9 * 1024 float4 + the queue slots <= 160 KB)
        const int stg_inst = stg1k <= 2 ? 2 : (stg1k <= 4 ? 4 : (stg1k <= 6 ? 6 : (stg1k <= 7 ? 7 : (stg1k <= 8 ? 8 : 9))));
        const bool no_fast = R.tun->no_fast;                                    // test / A-B hook: grouping-exact kernel only
        const bool certs = !no_fast && kernel == 0;
        int *queue = R.queue;                                                 // zeroed at the start of the sweep
        int *gcnt = R.gcnt;
        // geometry of the grouping-exact kernel (1024 threads)
        constexpr int nt_s = 1024;
        // LDS must also absorb the unconditional over-write past the row's end
        const size_t lds_stage_s = std::max(static_cast<size_t>(lds_row) * sizeof(float), static_cast<size_t>(stg_inst) * nt_s * 16);
        const size_t lds = lds_stage_s + 32 + (GENPHI_WG_TIMES ? 128 : 0);
        const int per_thread = static_cast<int>((width + nt_s - 1) / nt_s);     // the padding columns [n, ld) are written too
        // register budget of the instantiations (all spill-free: a spill stalls the pipeline)
        int max_cpt;
        if (pos_ord) max_cpt = stg_inst <= 7 ? 24 : (stg_inst == 8 ? 20 : 16);
        else           max_cpt = stg_inst <= 4 ? 24 : (stg_inst <= 8 ? 20 : 16);
        const int env_cpt = R.tun->max_cpt;                                     // tuning hook: smaller chunks
        if (env_cpt >= 4) max_cpt = std::min(max_cpt, env_cpt / 4 * 4);
        const int n_chunks = (per_thread + max_cpt - 1) / max_cpt;
        const int cpt = (per_thread + n_chunks - 1) / n_chunks;
        const long long n_items = static_cast<long long>(dg.n_segs) * n_chunks;
        if (certs) {
            // ---- certified groups: level_split_fast_kernel ----
            LevelArgs f = a;
            int f_nt = 1024, f_chunks = 1 << 30, f_stg = stg_inst;
            const int force_nt = R.tun->fast_nt;                                // test / tuning hook
            for (int nt : {1024, 512}) {
                if (force_nt && nt != force_nt) continue;
                int stg = (per_row4 + nt - 1) / nt;
                if (nt == 512) stg = stg <= 12 ? 12 : (stg <= 14 ? 14 : (stg <= 16 ? 16 : 18)); else stg = stg_inst;
                const int pt = static_cast<int>((width + nt - 1) / nt);
                int mc = fast_max_cpt(nt, stg);
                if (env_cpt >= 4) mc = std::min(mc, env_cpt * (1024 / nt) / 4 * 4);
                const int nch = (pt + mc - 1) / mc;
                if (nch < f_chunks) { f_chunks = nch; f_nt = nt; f_stg = stg; }
            }
            const int f_pt = static_cast<int>((width + f_nt - 1) / f_nt);
            const int f_cpt = ((f_pt + f_chunks - 1) / f_chunks + 3) / 4 * 4;
            const long long f_items = static_cast<long long>(dg.n_runs) * f_chunks;
            // certified runs -> glist_f / gcnt[0], the segments of the others -> glist_s / gcnt[1]: decided on the device, per launch
            int *glist_f = R.glist_f, *glist_s = R.glist_s;
            int *gcnt_s = gcnt;                                               // [1] = segments of the grouping-exact kernel
            hipLaunchKernelGGL(group_split_kernel, dim3((dg.n_runs + kSplitRuns - 1) / kSplitRuns), dim3(256), 0, R.stream, dg.desc, dg.seg, dg.run, dg.n_runs,
                               a.cert_prev, glist_f, glist_s, gcnt);
            HIP_TRY(hipGetLastError());
            f.glist = glist_f; f.gcnt = gcnt;
            f.chunk_magic = chunk_magic_for(f_chunks);
            const size_t f_lds_stage = std::max(static_cast<size_t>(lds_row) * sizeof(float), static_cast<size_t>(f_stg) * f_nt * 16);
            f.slot_off = static_cast<int>(f_lds_stage / sizeof(float));
            f.chunk_cols = f_cpt * f_nt;
            f.n_chunks = f_chunks;
            f.n_groups = dg.n_runs;
            const int f_grid = static_cast<int>(std::min<long long>(R.n_cus, (f_items + 7) / 8 * 8));
            // (the run lists hold chain steps only under GENPHI_MAX_RUN > 1: the default instantiation carries none of that state)
            hipError_t fe;
            if (R.tun->max_run > 1)
                fe = f.cert_out ? launch_fast<true, true>(f_nt, f_cpt, f_stg, f_grid, f_lds_stage + 32, R.stream, f, dg.desc, dg.seg, dg.run, queue)
                                : launch_fast<false, true>(f_nt, f_cpt, f_stg, f_grid, f_lds_stage + 32, R.stream, f, dg.desc, dg.seg, dg.run, queue);
            else
                fe = f.cert_out ? launch_fast<true, false>(f_nt, f_cpt, f_stg, f_grid, f_lds_stage + 32, R.stream, f, dg.desc, dg.seg, dg.run, queue)
                                : launch_fast<false, false>(f_nt, f_cpt, f_stg, f_grid, f_lds_stage + 32, R.stream, f, dg.desc, dg.seg, dg.run, queue);
            HIP_TRY(fe);
            a.zero_row = 0;                                                   // the fast launch wrote the "none" row
            a.glist = glist_s; a.gcnt = gcnt_s;
        }
        // ---- the other groups (all of them without certificates): level_split_kernel ----
        a.slot_off = static_cast<int>(lds_stage_s / sizeof(float));
        a.chunk_cols = (cpt + 3) / 4 * 4 * nt_s;                        // whole quads of columns per thread
        a.n_chunks = n_chunks;
        a.chunk_magic = chunk_magic_for(n_chunks);
        a.n_groups = dg.n_segs;
        const int grid = static_cast<int>(std::min<long long>(R.n_cus, (n_items + 7) / 8 * 8));   // persistent: one workgroup per CU
        HIP_TRY(pos_ord ? launch_split<true>(cpt, stg_inst, grid, lds, R.stream, a, dg.desc, dg.seg, queue + 8)
                          : launch_split<false>(cpt, stg_inst, grid, lds, R.stream, a, dg.desc, dg.seg, queue + 8));
    } else {
        return fail(GENPHI_ERR_ARG, "internal: launch_rows takes FULL and SPLIT steps");
    }
    HIP_TRY(hipGetLastError());
    return GENPHI_OK;
}

static int launch_level(genphi_plan *p, const LevelCtx &cx, const float *psi, float *out, const int *rows,
                        const int *out_rows, int n_rows, int kernel, const DeviceGroups &dg)
{
    const LevelStep &s = *cx.s;
    const DeviceStep &d = *cx.d;
    const int step = cx.slot;
    LevelArgs a;
    a.psi = psi; a.out = out; a.ld_prev = s.ld_prev; a.ld = s.ld; a.width = static_cast<int>(s.width);
    a.n_prev = static_cast<int>(s.n_prev); a.n = static_cast<int>(s.n);
    a.srcA = d.srcA; a.srcB = d.srcB; a.ord = d.ord; a.pk = d.pk;
    a.rows = rows; a.out_rows = out_rows; a.n_rows = n_rows;
    a.lds_row = 0; a.chunk_cols = 0;
    a.dbg = cx.dbg ? 1 : 0;
    a.zero_row = (out_rows == nullptr && kernel != 1 && !cx.no_none_row) ? 1 : 0;
    a.cert_prev = cx.cert_prev;
    a.cert_out = out_rows == nullptr ? cx.cert_out : nullptr;
    a.glist = nullptr; a.gcnt = nullptr; a.chunk_magic = 0;
    a.cert_thresh = cert_threshold(p->tun);
    a.cert_fast = (kernel == 0 && !p->tun.no_fast && cx.cert_prev != nullptr) ? 1 : 0;
    a.ord_col = d.ord; a.diag_col = nullptr; a.pdesc = nullptr; a.zrow = a.n;
    if (n_rows <= 0) {
        // nothing to compute (a row shard whose ancestors do not reach this level), but the next level
        // still reads this level's all-zero "none" row for its parentless members
        if (out_rows == nullptr && !cx.no_none_row)
            HIP_TRY(hipMemsetAsync(out + s.n * s.ld, 0, static_cast<size_t>(s.ld) * sizeof(float), p->stream));
        return GENPHI_OK;
    }
    if (kernel == 1) {
        dim3 grid(static_cast<unsigned>(n_rows), static_cast<unsigned>((s.ld + 255) / 256));
        hipLaunchKernelGGL(level_naive_kernel, grid, dim3(256), 0, p->stream, a);
    } else if (cx.identity) {
        hipLaunchKernelGGL(level_identity_kernel, dim3(static_cast<unsigned>(n_rows)), dim3(256), 0, p->stream, a);
        if (out_rows == nullptr)         // intermediate level: its all-zero "none" row
            HIP_TRY(hipMemsetAsync(out + s.n * s.ld, 0, static_cast<size_t>(s.ld) * sizeof(float), p->stream));
    } else if (s.mode == genphi::kModeFull || s.mode == genphi::kModeSplit) {
        LaunchRes R;
        R.stream = p->stream; R.n_cus = p->n_cus; R.tun = &p->tun;
        R.queue = p->d_queues + 16 * step; R.gcnt = p->d_gcnt + 4 * step;
        R.glist_f = p->d_glist; R.glist_s = p->d_glist + (static_cast<size_t>(p->plan.max_cut) + 64) / 64 * 64;
        return launch_rows(R, a, s.mode, s.pos_ord, s.n_prev + 1, s.width, kernel, dg);
    } else {
        return fail(GENPHI_ERR_ARG, "internal: WIDE steps go through launch_wide_level");
    }
    HIP_TRY(hipGetLastError());
    return GENPHI_OK;
}

// ---- column panels (panel_phi.hip): a level step of a rank's panel through the same row kernels -------------
const void *genphi::panel_tuning_create() { return new (std::nothrow) Tuning(tuning_from(nullptr)); }
void genphi::panel_tuning_destroy(const void *t) { delete static_cast<const Tuning *>(t); }
int genphi::panel_tuning_lds_cap(const void *t, int dflt)
{
    const Tuning *u = static_cast<const Tuning *>(t);
    return (u && u->lds_cap_floats >= 16) ? u->lds_cap_floats : dflt;
}
int genphi::panel_tuning_full_max(const void *t, int dflt)
{
    const Tuning *u = static_cast<const Tuning *>(t);
    return (u && u->full_max_floats >= 0) ? u->full_max_floats : dflt;
}
unsigned genphi::panel_tuning_cert_thresh(const void *t)
{
    static const Tuning dflt;
    return cert_threshold(t ? *static_cast<const Tuning *>(t) : dflt);
}

int genphi::launch_panel_level(const PanelLaunch &L)
{
    static const Tuning dflt;
    const Tuning &tun = L.tuning ? *static_cast<const Tuning *>(L.tuning) : dflt;
    LevelArgs a;
    std::memset(&a, 0, sizeof(a));
    a.psi = L.psi; a.out = L.out; a.ld_prev = L.ld_prev; a.ld = L.ld; a.width = static_cast<int>(L.ld);
    a.n_prev = L.n_prev; a.n = L.n_cols; a.zrow = L.n_cut;
    a.srcA = L.srcA; a.srcB = L.srcB; a.ord = L.ord; a.pk = L.pk_col;
    a.ord_col = L.ord_col; a.diag_col = L.diag_col; a.pdesc = L.pdesc;
    a.rows = L.work; a.out_rows = nullptr; a.n_rows = L.n_cut;
    a.zero_row = 1;
    a.cert_prev = L.cert_prev; a.cert_out = L.cert_out;
    a.cert_thresh = cert_threshold(tun);
    a.cert_fast = (!tun.no_fast && L.cert_prev != nullptr) ? 1 : 0;
    HIP_TRY(hipMemsetAsync(L.counters, 0, 20 * sizeof(int), L.stream));
    LaunchRes R;
    R.stream = L.stream; R.n_cus = L.n_cus; R.tun = &tun;
    R.queue = L.counters; R.gcnt = L.counters + 16;
    R.glist_f = L.glist; R.glist_s = L.glist + L.glist_cap;
    DeviceGroups dg;
    dg.desc = const_cast<int4 *>(L.desc); dg.seg = const_cast<int4 *>(L.seg); dg.run = const_cast<int4 *>(L.run);
    dg.n_segs = L.n_segs; dg.n_runs = L.n_runs;
    return launch_rows(R, a, L.mode, /*pos_ord=*/false, L.src_width, L.ld, /*kernel=*/0, dg);
}

// A WIDE level step (see rows_compact_kernel): out = the level matrix of the cut in its
// [dragged..., new...] storage order, all rows.
static int launch_wide_level(genphi_plan *p, int step, const float *psi, float *out, int kernel)
{
    const LevelStep &s = p->plan.steps[step];
    const DeviceStep &d = p->dsteps[step];
    const LevelCtx cx = main_ctx(p, step);
    const int n = static_cast<int>(s.n), nd = static_cast<int>(s.n_dragged), n_new = n - nd;
    // (a source cut stored by slot: "none" is row P of its matrix; kernel = 1 sweeps never store by slot)
    const bool by_slot = s.src_slots && kernel != 1;
    const int none = by_slot ? s.P : static_cast<int>(s.n_prev);
    const unsigned thr = cert_threshold(p->tun);
    int *cert_out = cx.cert_out;
    const bool stay = s.stay && kernel != 1;
    if (kernel == 1 || cx.identity) {
        // the per-entry kernel and the 1/2 I kernel take any cut width: all rows in one launch
        LevelCtx c2 = cx;
        const int rc = launch_level(p, c2, psi, out, d.work, nullptr, n, kernel, d.groups);
        if (rc) return rc;
        if (kernel == 1) HIP_TRY(hipMemsetAsync(out + s.n * s.ld, 0, static_cast<size_t>(s.ld) * sizeof(float), p->stream));
        return GENPHI_OK;
    }
    if (stay && n_new == 0) return GENPHI_OK;                   // the cut only lost members: nothing moves
    const int n_gran = static_cast<int>(s.blk_slot.size());
    if (stay) {                                                 // the new members' certificate words; those of Psi_P's rows and of the scatter buffer's
        hipLaunchKernelGGL(slots_clear_kernel, dim3(static_cast<unsigned>((n_gran * 64 + 255) / 256)), dim3(256), 0, p->stream, cert_out, d.blk_slot, n_gran,
                           p->d_cert_p, static_cast<int>(p->cert_p_words));
        HIP_TRY(hipGetLastError());
    }
    const bool nn_naive = s.nn_naive || s.nn.empty();
    if (n_new > 0 && !nn_naive) {
        // 1. Psi_P = Psi[parents][parents], with its zero padding and "none" row, and its rows' certificates
        const LevelStep &nn = s.nn[0];
        const DeviceStep &dn = p->nn_dsteps[d.nn];
        const int n_par = static_cast<int>(nn.n_prev);
        if (!stay) HIP_TRY(hipMemsetAsync(p->d_cert_p, 0, (static_cast<size_t>(n_par) + 1) * sizeof(int), p->stream));
        if (n_par > 0) {
            dim3 grid(static_cast<unsigned>(n_par), static_cast<unsigned>((n_par + 2047) / 2048));
            hipLaunchKernelGGL(rows_compact_kernel, grid, dim3(256), 0, p->stream, psi, static_cast<long long>(s.ld_prev), none,
                               d.pardesc, d.parents, n_par, p->psi_p, static_cast<long long>(nn.ld_prev), p->d_cert_p, thr);
        }
        hipLaunchKernelGGL(pad_zero_kernel, dim3(static_cast<unsigned>(n_par + 1)), dim3(256), 0, p->stream, p->psi_p,
                           static_cast<long long>(nn.ld_prev), n_par, static_cast<long long>(nn.ld_prev));
        HIP_TRY(hipGetLastError());
        // 2. the new x new block: a FULL / SPLIT level step on Psi_P, written IN PLACE.  The sub-step's
        //    member lead + r is member nd + r of the cut, so its matrix starts (nd - lead) rows and
        //    columns into the cut's; the `lead` placeholder columns (zeros) land on columns
        //    [nd - lead, nd) of the new rows, which pass 3 overwrites.  The sub-step also writes the
        //    zero padding [n, ld) of its rows.
        const bool scatter = stay && !s.contig;                 // the new members' slots are not one stretch: through a compact buffer
        const long long shift = stay ? s.p0 : nd - nn.lead;
        LevelCtx cn;
        cn.s = &nn; cn.d = &dn; cn.slot = static_cast<int>(p->plan.steps.size()) + 1 + d.nn;
        cn.cert_prev = p->d_cert_p;
        cn.cert_out = scatter ? p->d_cert_t : cert_out + shift;
        cn.identity = false; cn.dbg = false; cn.no_none_row = true;
        // (in place with scattered slots: the block goes to a compact buffer of its own -- the sub-step's pitch is npad -- and
        // from there to the new members' rows and columns; with the slots in one stretch it is written where it belongs)
        const int rc = launch_level(p, cn, p->psi_p, scatter ? p->nn_tmp : out + shift * (s.ld + 1), dn.work, nullptr, n_new, 0, dn.groups);
        if (rc) return rc;
        if (scatter) {
            hipLaunchKernelGGL(slots_scatter_kernel, dim3(static_cast<unsigned>(n_new)), dim3(256), 0, p->stream, p->nn_tmp, static_cast<long long>(nn.ld),
                               n_new, d.blk_slot, out, static_cast<long long>(s.ld), p->d_cert_t, cert_out);
            HIP_TRY(hipGetLastError());
        }
    }
    // Two routes for the blocks that involve dragged members:
    //   A  rows_compact_kernel on every row (dragged x dragged, new x dragged), then dragged x new as the
    //      transpose of new x dragged.  The new rows stream two whole parent rows each to keep nd columns.
    //   B  drag_rows_kernel: the dragged rows whole (compacted copy + the dragged x new block gathered from
    //      the row's entries at the parents' positions, which sit in LDS), then new x dragged as the
    //      transpose of dragged x new: the parent rows are not streamed at all, but the kernel is held to
    //      one workgroup per CU by its LDS and streams ~25 % slower than the 2-D grid of route A.
    //   B wins when new members are many and dragged ones few (the late levels of overlapping generations).
    //   Per level of cfg4o (profiles/microbench/out/r02_ab_wide_route_per_level_cfg4o.out) B is faster from
    //   nd / n_prev ~ 0.67 downwards (the level at 0.29: 4.07 -> 3.52 ms), slower above (the widest level:
    //   24.9 -> 27.8 ms).  GENPHI_WIDE_ROUTE = A | B forces one (A/B hook); default: B iff nd / n_prev < 2/3.
    if (stay) {
        if (nn_naive) return fail(GENPHI_ERR_ARG, "internal: an in-place WIDE step without a row kernel for its new x new block");
        if (!p->tun.stay_two_pass) {
            // 3S + 4S fused: new x dragged and its transpose in one pass over the parents' rows (rows_avg_t_kernel)
            const int ft = d.tile_cols;
            const size_t lds = 64 * static_cast<size_t>(ft + 1) * sizeof(float);
            HIP_TRY(set_max_lds(ft == 128 ? reinterpret_cast<const void *>(rows_avg_t_kernel<128>) : reinterpret_cast<const void *>(rows_avg_t_kernel<256>), lds));
            if (d.n_tiles > 0) {
                const int gf = p->tun.stay_col_fastest ? 0 : 1;
                const unsigned nct = static_cast<unsigned>(d.n_tiles);
                dim3 gt(gf ? static_cast<unsigned>(n_gran) : nct, gf ? nct : static_cast<unsigned>(n_gran));
                if (ft == 128)
                    hipLaunchKernelGGL(rows_avg_t_kernel<128>, gt, dim3(256), lds, p->stream, out, static_cast<long long>(s.ld), none, d.rowdesc + nd, n_new,
                                       d.blk_slot, d.tiles, cert_out, thr, gf, p->tun.stay_scalar_t ? 1 : 0);
                else
                    hipLaunchKernelGGL(rows_avg_t_kernel<256>, gt, dim3(256), lds, p->stream, out, static_cast<long long>(s.ld), none, d.rowdesc + nd, n_new,
                                       d.blk_slot, d.tiles, cert_out, thr, gf, p->tun.stay_scalar_t ? 1 : 0);
                HIP_TRY(hipGetLastError());
            }
            return GENPHI_OK;
        }
        // 3S. new x dragged: the new rows at the dragged members' columns, from the parents' rows of the same matrix
        {
            dim3 grid(static_cast<unsigned>(n_new), static_cast<unsigned>((nd + 2047) / 2048));
            hipLaunchKernelGGL(rows_avg_kernel, grid, dim3(256), 0, p->stream, psi, static_cast<long long>(s.ld_prev), none, d.rowdesc + nd,
                               d.idx, nd, out, cert_out, thr);
            HIP_TRY(hipGetLastError());
        }
        // 4S. dragged x new = (new x dragged)^T, over the slot ranges that hold the dragged members (the dead slots among them
        //     included: their rows and columns hold nothing that is read); a tile row is one granule of new members
        for (size_t h = 0; h + 1 < s.live_ranges.size(); h += 2) {
            const int c_lo = s.live_ranges[h], len = s.live_ranges[h + 1] - c_lo;
            dim3 gt(static_cast<unsigned>((len + kTT - 1) / kTT), static_cast<unsigned>(n_gran));
            hipLaunchKernelGGL(transpose_slots_kernel, gt, dim3(256), 0, p->stream, out, static_cast<long long>(s.ld), n_new, d.blk_slot, c_lo, len,
                               cert_out, thr);
            HIP_TRY(hipGetLastError());
        }
        return GENPHI_OK;
    }
    bool route_b = !nn_naive && nd > 0 && n_new > 0 && s.nn[0].n_prev > 0 && !by_slot;     // (route B's parent windows need ascending sources)
    if (route_b) route_b = p->tun.wide_route ? p->tun.wide_route == 'B' : (3LL * nd < 2LL * s.n_prev);
    if (route_b) {
        // 3B. the dragged rows, all n columns
        const LevelStep &nn = s.nn[0];
        const DeviceStep &dn = p->nn_dsteps[d.nn];
        const int n_par = static_cast<int>(nn.n_prev);
        const size_t lds = (static_cast<size_t>(n_par) + 4) / 4 * 4 * sizeof(float);
        const unsigned *pkn = dn.pk + nn.lead;
        HIP_TRY(set_max_lds(reinterpret_cast<const void *>(drag_rows_kernel<1024>), lds));
        const int grid = std::min(nd, p->n_cus * (lds <= 78 * 1024 ? 2 : 1));            // persistent
        hipLaunchKernelGGL(drag_rows_kernel<1024>, dim3(static_cast<unsigned>(grid)), dim3(1024), lds, p->stream, psi,
                           static_cast<long long>(s.ld_prev), d.srcA, nd, d.parents, n_par, d.pstart, pkn, n_new, out,
                           static_cast<long long>(s.ld), cert_out, thr);
        HIP_TRY(hipGetLastError());
        // 4B. new x dragged = (dragged x new)^T: destination runs start at column 0 (aligned)
        dim3 gt(static_cast<unsigned>((n_new + kTT - 1) / kTT), static_cast<unsigned>((nd + kTT - 1) / kTT));
        hipLaunchKernelGGL(transpose_block_kernel, gt, dim3(256), 0, p->stream, out + nd, static_cast<long long>(s.ld), nd, n_new,
                           out + static_cast<long long>(nd) * s.ld, static_cast<long long>(s.ld), 0, 0, cert_out + nd, thr);
        HIP_TRY(hipGetLastError());
    } else {
        // 3A. columns [0, nd) of every row: dragged rows are compacted copies, new rows compacted half sums
        const int rows_1 = nn_naive ? nd : n;                 // (naive fallback: the new rows come whole from the per-entry kernel)
        if (nd > 0 && rows_1 > 0) {
            dim3 grid(static_cast<unsigned>(rows_1), static_cast<unsigned>((nd + 2047) / 2048));      // 256 threads x 8 elements
            hipLaunchKernelGGL(rows_compact_kernel, grid, dim3(256), 0, p->stream, psi, static_cast<long long>(s.ld_prev), none,
                               d.rowdesc, d.idx, nd, out, static_cast<long long>(s.ld), cert_out, thr);
            HIP_TRY(hipGetLastError());
        }
        if (n_new > 0 && nn_naive) {
            LevelArgs a;
            std::memset(&a, 0, sizeof(a));
            a.psi = psi; a.out = out; a.ld_prev = s.ld_prev; a.ld = s.ld; a.width = static_cast<int>(s.ld); a.n_prev = none; a.n = n;
            a.srcA = d.srcA; a.srcB = d.srcB; a.ord = d.ord; a.rows = d.newrows; a.n_rows = n_new;
            a.cert_out = cert_out; a.cert_thresh = thr; a.ord_col = d.ord; a.zrow = n;
            dim3 grid(static_cast<unsigned>(n_new), static_cast<unsigned>((s.ld + 255) / 256));
            hipLaunchKernelGGL(level_naive_kernel, grid, dim3(256), 0, p->stream, a);
            HIP_TRY(hipGetLastError());
        }
        // 4A. dragged x new = (new x dragged)^T
        if (n_new > 0 && nd > 0) {
            const bool tt_align = !p->tun.tt_noalign;                                        // A/B hook
            const int shift = tt_align ? (nd & 31) : 0;                     // destination runs start on 128-byte lines
            dim3 gt(static_cast<unsigned>((nd + kTT - 1) / kTT), static_cast<unsigned>((n_new + shift + kTT - 1) / kTT));
            hipLaunchKernelGGL(transpose_block_kernel, gt, dim3(256), 0, p->stream, out + static_cast<long long>(nd) * s.ld,
                               static_cast<long long>(s.ld), n_new, nd, out, static_cast<long long>(s.ld), nd, shift, cert_out, thr);
            HIP_TRY(hipGetLastError());
        }
    }
    // 5. padding columns and the "none" row
    hipLaunchKernelGGL(pad_zero_kernel, dim3(static_cast<unsigned>(n + 1)), dim3(256), 0, p->stream, out, static_cast<long long>(s.ld), n,
                       static_cast<long long>(s.width));
    HIP_TRY(hipGetLastError());
    return GENPHI_OK;
}

// Zero-aware leading levels (sparse_levels.h): the first sweep of a plan that can use them builds the children lists of the eligible
// steps and runs them once to count the non-zero entries of every cut; that fixes k, the last cut kept as row lists, for the life of the
// plan (values do not depend on k; GENPHI_FLAG_NO_SPARSE runs the same plan densely).  The Float32 product sweep and the Float64-storage
// sweep share the lists (integer values: exact for both).
static int ensure_sparse_levels(genphi_plan *p, int kernel, PhaseTrace &trace)
{
    if (kernel != 0 || p->sparse_tried) return GENPHI_OK;
    p->sparse_tried = true;
    const Plan &pl = p->plan;
    const int S = p->tun.sparse_k == -1 ? 0 : genphi::sparse_eligible_steps(pl);
    if (S >= 2) {
        std::vector<genphi::SparseStepDev> dev(S);
        for (int s = 0; s < S; ++s)
            dev[s] = genphi::SparseStepDev{p->dsteps[s].srcA, p->dsteps[s].srcB, p->dsteps[s].ord,
                                           static_cast<int64_t>(pl.steps[s].work.size()) == pl.steps[s].n ? p->dsteps[s].work : nullptr};
        genphi::SparseTuning stn;
        stn.force_k = p->tun.sparse_k;
        if (p->tun.sparse_permille > 0) stn.max_permille = p->tun.sparse_permille;
        if (p->tun.sparse_min_cut >= 0) stn.min_cut = p->tun.sparse_min_cut;
        if (p->tun.sparse_chunk > 0) stn.chunk_cols = p->tun.sparse_chunk;
        stn.classes = p->tun.sparse_classes;
        if (p->tun.sparse_batch == 4 || p->tun.sparse_batch == 8) stn.long_batch = p->tun.sparse_batch;
        if (p->tun.sparse_arena > 0) stn.first_entries = p->tun.sparse_arena;
        std::string serr;
        p->sparse = genphi::sparse_levels_create(pl, S, dev, stn, p->stream, serr);
        if (p->sparse) {
            const int rc = genphi::sparse_levels_calibrate(p->sparse, p->stream, serr);
            if (rc) return fail(rc, "sparse levels: " + serr);
        } else {
            (void)hipGetLastError();          // (no memory for the row lists: the sweep stays dense)
        }
    }
    trace.mark("sparse levels: lists + calibration");
    return GENPHI_OK;
}

static int ensure_doubles(double **ptr, size_t *have, size_t need)
{
    if (*have >= need && *ptr) return GENPHI_OK;
    if (*ptr) { HIP_TRY(genphi::cached_free(*ptr)); *ptr = nullptr; *have = 0; }
    HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(ptr), std::max<size_t>(need, 1) * sizeof(double)));
    *have = need;
    return GENPHI_OK;
}

// The whole sweep with Float64 level matrices (see level_naive64_kernel): rows [r0, r1) of the
// proband matrix end up in p->result64 (row pitch = pitch of the last cut).
static int compute_f64(genphi_plan *p, int64_t r0, int64_t r1, int kernel, genphi_stats *stats, bool timing, bool no_sparse)
{
    const Plan &pl = p->plan;
    const int L = pl.n_levels, n_steps = L - 1;
    const int64_t N = pl.n_pro, ldN = pl.ld[L - 1], n_rows = r1 - r0;
    int rc = ensure_doubles(&p->result64, &p->result64_doubles, static_cast<size_t>(n_rows * ldN));
    if (rc) return rc;
    // storage member of each resident row (the last cut may be stored in [dragged, new] order)
    std::vector<int> srow(n_rows), orow(n_rows);
    for (int64_t k = 0; k < n_rows; ++k) { srow[k] = pl.final_perm.empty() ? static_cast<int>(r0 + k) : pl.final_perm[r0 + k]; orow[k] = static_cast<int>(k); }
    if (static_cast<size_t>(2 * n_rows) > p->perm_rows_cap) {
        if (p->d_perm_rows) { HIP_TRY(genphi::cached_free(p->d_perm_rows)); p->d_perm_rows = nullptr; }
        HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->d_perm_rows), 2 * n_rows * sizeof(int)));
        p->perm_rows_cap = static_cast<size_t>(2 * n_rows);
    }
    HIP_TRY(hipMemcpyAsync(p->d_perm_rows, srow.data(), n_rows * sizeof(int), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_perm_rows + n_rows, orow.data(), n_rows * sizeof(int), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));           // host vectors go out of scope
    if (n_steps == 0) {
        // all probands parentless: 1/2 I (src/compute.jl:271-274, loop skipped); members = probands in order
        HIP_TRY(hipMemsetAsync(p->result64, 0, static_cast<size_t>(n_rows * ldN) * sizeof(double), p->stream));
        std::vector<double> half(1, 0.5);
        for (int64_t k = 0; k < n_rows; ++k)
            HIP_TRY(hipMemcpyAsync(p->result64 + k * ldN + (r0 + k), half.data(), sizeof(double), hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return GENPHI_OK;
    }
    if (timing && n_steps + 2 > GENPHI_MAX_STAT_LEVELS) return fail(GENPHI_ERR_ARG, "too many levels for timing stats");
    if (timing)
        while (static_cast<int>(p->events.size()) < n_steps + 3) {
            hipEvent_t e; HIP_TRY(hipEventCreate(&e)); p->events.push_back(e);
        }
    // the leading cuts as lists of their non-zero entries here too (integer values are exact for Float64 as for Float32, sparse_levels.h):
    // the first dense matrix is written in Float64, compactly
    {
        PhaseTrace trace;
        rc = ensure_sparse_levels(p, kernel, trace);
        if (rc) return rc;
    }
    const int sparse_k = (kernel == 0 && p->sparse && !no_sparse) ? genphi::sparse_levels_k(p->sparse) : -1;
    {   // Float64 level matrices of the cuts that exist as matrices (cuts 0..sparse_k live as row lists): genea140 2 x 1.5 GB -> 2 x 0.2
        size_t need[2] = {0, 0};
        for (int c = sparse_k + 1; c + 1 < L; ++c) need[c & 1] = std::max(need[c & 1], static_cast<size_t>((pl.cut_sizes[c] + 1) * pl.ld[c]));
        for (int b = 0; b < 2; ++b)
            if (need[b]) { rc = ensure_doubles(&p->buf64[b], &p->buf64_doubles[b], need[b]); if (rc) return rc; }
    }
    const int64_t n0 = pl.cut_sizes[0], ld0 = pl.ld[0];
    if (timing) HIP_TRY(hipEventRecord(p->events[0], p->stream));
    if (sparse_k < 0) {
        HIP_TRY(hipMemsetAsync(p->buf64[0], 0, static_cast<size_t>((n0 + 1) * ld0) * sizeof(double), p->stream));
        hipLaunchKernelGGL(half_identity64_kernel, dim3(static_cast<unsigned>((n0 + 255) / 256)), dim3(256), 0, p->stream, p->buf64[0],
                           static_cast<long long>(ld0), static_cast<int>(n0), static_cast<const int *>(nullptr), static_cast<int>(n0),
                           static_cast<const int *>(nullptr));
        HIP_TRY(hipGetLastError());
    }
    for (int s = 0; s < n_steps; ++s) {
        const LevelStep &st = pl.steps[s];
        const DeviceStep &d = p->dsteps[s];
        if (s <= sparse_k) {
            std::string serr;
            rc = s < sparse_k ? genphi::sparse_levels_enqueue_step(p->sparse, s, p->stream, serr)
                              : genphi::sparse_levels_enqueue_dense(p->sparse, p->buf64[(s + 1) & 1], true, true, st.ld, st.ld, p->stream, serr);
            if (rc == GENPHI_OK && s == sparse_k) rc = genphi::sparse_levels_enqueue_flags(p->sparse, p->stream, serr);
            if (rc) return fail(rc, "sparse levels: " + serr);
            if (timing) HIP_TRY(hipEventRecord(p->events[s + 1], p->stream));
            if (stats && s < GENPHI_MAX_STAT_LEVELS) stats->level_rows[s] = st.n;
            continue;
        }
        const double *psi = p->buf64[s & 1];
        const bool last = s == n_steps - 1;
        double *out = last ? p->result64 : p->buf64[(s + 1) & 1];
        const int rows_n = last ? static_cast<int>(n_rows) : static_cast<int>(st.n);
        const int *k_rows = last ? p->d_perm_rows : static_cast<const int *>(nullptr);
        const int *k_orows = last ? p->d_perm_rows + n_rows : static_cast<const int *>(nullptr);
        const int *k_colmap = (last && !pl.final_perm.empty()) ? p->d_final_perm : static_cast<const int *>(nullptr);
        const int lds_row64 = static_cast<int>((st.n_prev + 1 + 1) / 2 * 2);
        const size_t lds64 = 2 * static_cast<size_t>(lds_row64) * sizeof(double);
        if (kernel != 1 && lds64 <= 160 * 1024 && rows_n > 0) {
            // both Float64 source rows fit in LDS (cuts up to 10,239 members): the row-staged kernel
            HIP_TRY(set_max_lds(reinterpret_cast<const void *>(level_full64_kernel), lds64));
            // (columns per thread <= kF64Cols, row pieces per thread <= kF64Pre for each of these block sizes)
            const int bs = st.ld <= 64 * kF64Cols && lds_row64 <= 2 * 64 * kF64Pre ? 64 : (st.ld <= 256 * kF64Cols && lds_row64 <= 2 * 256 * kF64Pre ? 256 : 512);
            const int chunks = static_cast<int>((st.ld + bs * kF64Cols - 1) / (bs * kF64Cols));
            // resident workgroups per CU: LDS, and the kernel's ~240 VGPRs (two waves per SIMD)
            const int per_cu = static_cast<int>(std::min<size_t>(512 / bs, std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds64, 1))));
            const int grid = std::min(rows_n, std::max(1, p->n_cus * per_cu / chunks));
            hipLaunchKernelGGL(level_full64_kernel, dim3(static_cast<unsigned>(grid), static_cast<unsigned>(chunks)), dim3(bs), lds64, p->stream, psi,
                               static_cast<long long>(st.ld_prev), static_cast<int>(st.n_prev), out, static_cast<long long>(st.ld), d.srcA, d.srcB,
                               d.ord, k_rows, k_orows, k_colmap, static_cast<int>(st.n), lds_row64, rows_n);
        } else if (kernel != 1 && static_cast<size_t>(lds_row64) * sizeof(double) <= 160 * 1024 && rows_n > 0 && st.n_prev < 65535) {
            // one Float64 source row fits (cuts up to 20,479 members): the one-row-at-a-time kernel, rows in the step's work order
            // (same A source adjacent) where the step has one
            const size_t lds1 = static_cast<size_t>(lds_row64) * sizeof(double);
            HIP_TRY(set_max_lds(reinterpret_cast<const void *>(level_split64_kernel), lds1));
            const int chunks = static_cast<int>((st.ld + 512 * kS64Cols - 1) / (512 * kS64Cols));
            const int *w_rows = last ? k_rows : ((st.mode != genphi::kModeWide && static_cast<int64_t>(st.work.size()) == st.n) ? d.work : static_cast<const int *>(nullptr));
            const int grid_x = std::min(rows_n, std::max(1, p->n_cus / chunks) * 2);     // (two pieces per CU and chunk: the tail)
            hipLaunchKernelGGL(level_split64_kernel, dim3(static_cast<unsigned>(grid_x), static_cast<unsigned>(chunks)), dim3(512), lds1, p->stream, psi,
                               static_cast<long long>(st.ld_prev), static_cast<int>(st.n_prev), out, static_cast<long long>(st.ld), d.srcA, d.srcB,
                               d.ord, w_rows, k_orows, k_colmap, static_cast<int>(st.n), lds_row64, rows_n);
        } else {
            dim3 grid(static_cast<unsigned>(rows_n), static_cast<unsigned>((st.ld + 255) / 256));
            hipLaunchKernelGGL(level_naive64_kernel, grid, dim3(256), 0, p->stream, psi, static_cast<long long>(st.ld_prev),
                               static_cast<int>(st.n_prev), out, static_cast<long long>(st.ld), static_cast<int>(st.n), d.srcA, d.srcB, d.ord,
                               k_rows, k_orows, k_colmap, static_cast<int>(st.n));
        }
        HIP_TRY(hipGetLastError());
        if (!last)           // the all-zero "none" row of this level
            HIP_TRY(hipMemsetAsync(out + st.n * st.ld, 0, static_cast<size_t>(st.ld) * sizeof(double), p->stream));
        if (timing) HIP_TRY(hipEventRecord(p->events[s + 1], p->stream));
        if (stats && s < GENPHI_MAX_STAT_LEVELS) stats->level_rows[s] = rows_n;
    }
    (void)N;
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (sparse_k >= 0 && !genphi::sparse_levels_flags_ok(p->sparse))
        return fail(GENPHI_ERR_DEVICE, "sparse levels: a row list did not have the length the plan recorded (internal error)");
    if (timing) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->events[0], p->events[n_steps]));
        stats->total_ms = ms;
        for (int s = 0; s < n_steps; ++s) {
            HIP_TRY(hipEventElapsedTime(&ms, p->events[s], p->events[s + 1]));
            stats->level_ms[s] = ms;
        }
        stats->final_ms = stats->level_ms[n_steps - 1];
        stats->timed = 1;
    }
    return GENPHI_OK;
}

extern "C" {

int genphi_compute_device(genphi_plan *p, const genphi_opts *opts, genphi_stats *stats)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    const Plan &pl = p->plan;
    const int device = opts ? opts->device : -1;
    const int kernel = opts ? opts->kernel : 0;
    const bool timing = opts && opts->timing && stats;
    int64_t r0 = 0, r1 = pl.n_pro;
    if (opts && opts->row_end > 0) { r0 = opts->row_begin; r1 = opts->row_end; }
    if (r0 < 0 || r1 > pl.n_pro || r0 > r1) return fail(GENPHI_ERR_ARG, "row shard out of range");
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n_steps = std::max(pl.n_levels - 1, 0);
        stats->algorithmic_bytes = pl.algorithmic_bytes;
        stats->max_cut = pl.max_cut;
    }
    p->res_row_begin = r0; p->res_n_rows = r1 - r0;
    const int L = pl.n_levels;
    if (L == 0 || r1 == r0) { p->res_ld = 0; return GENPHI_OK; }

    PhaseTrace trace;
    int rc = upload_plan(p, device);
    if (rc) return rc;
    trace.mark("upload_plan");
    p->res_f64 = opts && (opts->flags & GENPHI_FLAG_STORAGE_F64);
    if (p->res_f64) {
        p->res_ld = pl.ld[L - 1];
        return compute_f64(p, r0, r1, kernel, stats, timing, opts && (opts->flags & GENPHI_FLAG_NO_SPARSE));
    }
    if (p->popt.indices_only) return fail(GENPHI_ERR_ARG, "internal: an indices-only plan serves Float64-storage sweeps only");
    p->stay_active = kernel != 1;             // the per-entry kernel sweep (kernel = 1) knows no slots: every level is written compactly
    const int n_steps = L - 1;
    rc = ensure_sparse_levels(p, kernel, trace);
    if (rc) return rc;
    const int sparse_k = (kernel == 0 && p->sparse && !(opts && (opts->flags & GENPHI_FLAG_NO_SPARSE))) ? genphi::sparse_levels_k(p->sparse) : -1;
    rc = ensure_level_buffers(p, sparse_k + 1);       // (cuts 0..sparse_k never exist as matrices)
    if (rc) return rc;
    trace.mark("ensure_level_buffers");
    if (timing && n_steps + 2 > GENPHI_MAX_STAT_LEVELS) return fail(GENPHI_ERR_ARG, "too many levels for timing stats");
    if (timing) {
        while (static_cast<int>(p->events.size()) < n_steps + 3) {
            hipEvent_t e; HIP_TRY(hipEventCreate(&e)); p->events.push_back(e);
        }
    }

    const int64_t N = pl.n_pro, ldN = pl.ld[L - 1], n_rows = r1 - r0;
    rc = ensure_floats(p, &p->result, &p->result_floats, static_cast<size_t>(n_rows * ldN));
    if (rc) return rc;
    trace.mark("result buffer");
    p->res_ld = ldN;
    // last step WIDE: the whole level (every row, [dragged, new] storage order) goes to final_tmp,
    // then rows [r0, r1) are delivered in proband order by colperm_kernel
    const bool need_perm = !pl.final_perm.empty();
    // ... unless the proband cut STAYED IN PLACE at the end of a run (Plan::final_slots): the last step then writes only the new probands'
    // rows and columns into the run's matrix and colperm_kernel delivers from there, by slot -- one pass instead of a compaction + a
    // permutation (genea140 with every individual a proband: the last step 7.7 -> ... ms).  The per-entry sweep (kernel = 1) knows no slots.
    const bool last_by_slot = need_perm && !pl.final_slots.empty() && p->stay_active;
    if (need_perm && !last_by_slot) {
        rc = ensure_floats(p, &p->final_tmp, &p->final_tmp_floats, static_cast<size_t>((N + 1) * ldN) + kTailPadFloats);
        if (rc) return rc;
    }
    // shard row lists for the last step: storage row of proband r, output row r - r0
    if (n_rows > p->shard_cap) {
        drop_graph(p);
        if (p->d_shard_rows) { HIP_TRY(genphi::cached_free(p->d_shard_rows)); HIP_TRY(genphi::cached_free(p->d_shard_out_rows)); }
        p->d_shard_rows = p->d_shard_out_rows = nullptr;
        HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->d_shard_rows), n_rows * sizeof(int)));
        HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->d_shard_out_rows), n_rows * sizeof(int)));
        p->shard_cap = n_rows; p->shard_r0 = p->shard_r1 = -1;
    }
    // The whole result by a row-kernel last step in proband order: the step's own work order and walk lists (built by the planner and the
    // upload for exactly these rows) ARE the shard's -- no second reuse order, hub walk and upload (7 ms of a first call at 1e5 probands)
    const bool whole_by_rows = r0 == 0 && r1 == pl.n_pro && !need_perm && n_steps > 0 && pl.steps[n_steps - 1].mode != genphi::kModeWide &&
                               static_cast<int64_t>(pl.steps[n_steps - 1].work.size()) == n_rows &&
                               (pl.steps[n_steps - 1].mode != genphi::kModeSplit || p->dsteps[n_steps - 1].groups.desc != nullptr);
    if ((p->shard_r0 != r0 || p->shard_r1 != r1) && whole_by_rows) {
        drop_graph(p);
        const int *work = p->dsteps[n_steps - 1].work;
        HIP_TRY(hipMemcpyAsync(p->d_shard_rows, work, n_rows * sizeof(int), hipMemcpyDeviceToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->d_shard_out_rows, work, n_rows * sizeof(int), hipMemcpyDeviceToDevice, p->stream));      // (output row = storage row)
        p->shard_groups = p->dsteps[n_steps - 1].groups;
        (void)genphi::cached_free(p->sh_blob); p->sh_blob = nullptr; p->sh_valid = false;
        p->sh_steps.assign(std::max(n_steps, 1), genphi_plan::ShardStep());
        p->shard_r0 = r0; p->shard_r1 = r1;
    }
    if (p->shard_r0 != r0 || p->shard_r1 != r1) {
        drop_graph(p);                               // shard lists and sh_blob are rewritten / reallocated below
        // work order of the shard: the planner's reuse order (sibling groups, chained along
        // shared B sources) restricted to the shard's rows
        std::vector<int> rows(n_rows), orows(n_rows);
        std::vector<std::pair<int, int>> key(n_rows);   // (storage row, out row)
        for (int64_t k = 0; k < n_rows; ++k) {
            const int r = static_cast<int>(r0 + k);
            key[k] = {need_perm ? pl.final_perm[r] : r, static_cast<int>(k)};
        }
        if (n_steps > 0) {
            const LevelStep &s = pl.steps[n_steps - 1];
            if (s.mode != genphi::kModeWide) {          // (a WIDE last step computes every row, in storage order)
                std::vector<int32_t> ord(n_rows);
                std::vector<int> out_of(s.n, -1);
                for (int64_t k = 0; k < n_rows; ++k) { ord[k] = key[k].first; out_of[key[k].first] = key[k].second; }
                genphi::reuse_order(s, ord);
                for (int64_t k = 0; k < n_rows; ++k) key[k] = {ord[k], out_of[ord[k]]};
            }
        }
        for (int64_t k = 0; k < n_rows; ++k) { rows[k] = key[k].first; orows[k] = key[k].second; }
        HIP_TRY(hipMemcpyAsync(p->d_shard_rows, rows.data(), n_rows * sizeof(int), hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->d_shard_out_rows, orows.data(), n_rows * sizeof(int), hipMemcpyHostToDevice, p->stream));
        p->shard_groups = DeviceGroups();
        std::vector<char> gimg;
        if (n_steps > 0 && pl.steps[n_steps - 1].mode == genphi::kModeSplit) {
            GroupLists gl;
            build_groups(pl.steps[n_steps - 1], rows.data(), orows.data(), static_cast<int>(n_rows), gl, p->tun);
            const size_t gb = groups_bytes(gl);
            if (gb > p->shard_blob_bytes) {
                if (p->d_shard_blob) { HIP_TRY(genphi::cached_free(p->d_shard_blob)); p->d_shard_blob = nullptr; p->shard_blob_bytes = 0; }
                HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->d_shard_blob), gb));
                p->shard_blob_bytes = gb;
            }
            gimg.assign(gb, 0);
            size_t goff = 0;
            auto gput = [&](const void *src, size_t bytes) -> char * {
                char *d = p->d_shard_blob + goff;
                if (bytes) std::memcpy(gimg.data() + goff, src, bytes);
                goff += al256(bytes);
                return d;
            };
            put_groups(gl, p->shard_groups, gput);
            HIP_TRY(hipMemcpyAsync(p->d_shard_blob, gimg.data(), gb, hipMemcpyHostToDevice, p->stream));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));      // host vectors go out of scope
        p->shard_r0 = r0; p->shard_r1 = r1;

        // upper levels restricted to the ancestors of the shard (walk the sources backwards)
        (void)genphi::cached_free(p->sh_blob); p->sh_blob = nullptr; p->sh_valid = false;
        p->sh_steps.assign(std::max(n_steps, 1), genphi_plan::ShardStep());
        const bool sharded = n_rows < pl.n_pro && n_steps >= 2 && !p->tun.no_shard_prune;
        if (sharded) {
            std::vector<std::vector<int>> host_rows(n_steps);
            std::vector<GroupLists> host_gl(n_steps);
            std::vector<char> need(pl.steps[n_steps - 1].n_prev + 1, 0);            // members of cut n_steps-1
            {
                const LevelStep &sl = pl.steps[n_steps - 1];
                if (sl.mode == genphi::kModeWide) std::fill(need.begin(), need.end(), 1);
                else for (int64_t k = 0; k < n_rows; ++k) {
                    const int i = rows[k];
                    if (sl.srcA[i] < sl.n_prev) need[sl.srcA[i]] = 1;
                    if (sl.srcB[i] < sl.n_prev) need[sl.srcB[i]] = 1;
                }
            }
            size_t total = 256;
            auto al = [](size_t b) { return (b + 255) / 256 * 256; };
            for (int st = n_steps - 2; st >= 0; --st) {
                const LevelStep &sv = pl.steps[st];                                // produces cut st+1 (n = sv.n)
                std::vector<char> need_prev(sv.n_prev + 1, 0);
                std::vector<int> &rw = host_rows[st];
                if (p->tun.shard_force_step == st && p->tun.shard_force_row >= 0 && p->tun.shard_force_row < static_cast<int>(need.size()))
                    need[p->tun.shard_force_row] = 1;                                  // debugging aid
                if (sv.mode == genphi::kModeWide) {                                  // computes every row, reads every row
                    std::fill(need_prev.begin(), need_prev.end(), 1);
                    need.swap(need_prev);
                    continue;
                }
                for (int32_t i : sv.work)                                           // keep the planner's reuse order
                    if (need[i]) {
                        rw.push_back(i);
                        if (sv.srcA[i] < sv.n_prev) need_prev[sv.srcA[i]] = 1;
                        if (sv.srcB[i] < sv.n_prev) need_prev[sv.srcB[i]] = 1;
                    }
                if (sv.mode == genphi::kModeSplit) {
                    build_groups(sv, rw.data(), nullptr, static_cast<int>(rw.size()), host_gl[st], p->tun);
                    total += groups_bytes(host_gl[st]);
                }
                total += al(rw.size() * sizeof(int));
                need.swap(need_prev);
            }
            HIP_TRY(genphi::cached_malloc(reinterpret_cast<void **>(&p->sh_blob), total));
            std::vector<char> host(total, 0);
            size_t off = 0;
            auto put = [&](const void *src, size_t bytes) -> char * {
                char *d = p->sh_blob + off;
                if (bytes) std::memcpy(host.data() + off, src, bytes);
                off += al(bytes);
                return d;
            };
            for (int st = 0; st + 1 < n_steps; ++st) {
                genphi_plan::ShardStep &sh = p->sh_steps[st];
                sh.n_rows = static_cast<int>(host_rows[st].size());
                sh.rows = reinterpret_cast<int *>(put(host_rows[st].data(), host_rows[st].size() * sizeof(int)));
                if (!host_gl[st].w.run.empty()) put_groups(host_gl[st], sh.groups, put);
            }
            HIP_TRY(hipMemcpyAsync(p->sh_blob, host.data(), total, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->sh_valid = true;
        }
    }

    trace.mark("shard lists");
    // ---- the sweep: every launch of one gen.phi, in stream order ------------------------------
    const int prune_min_step = p->tun.shard_prune_min_step;                     // debugging aid
    const bool small_off = p->tun.no_small;                                      // test hook: per-level launches only
    std::vector<int> ev_after(std::max(n_steps, 1));                             // event recorded after step k (timing)
    for (int k = 0; k < static_cast<int>(ev_after.size()); ++k) ev_after[k] = k + 1;
    const std::vector<int> &bid = p->buf_of[p->stay_active ? 0 : 1];       // level buffer of every cut
    auto enqueue = [&]() -> int {
        HIP_TRY(hipMemsetAsync(p->d_queues, 0, p->sweep_words * sizeof(int), p->stream));      // queues, group counts, certificates: one array
        if (n_steps == 0) {
            // all probands parentless: result = 1/2 I (src/compute.jl:271-274, loop skipped)
            HIP_TRY(hipMemsetAsync(p->result, 0, static_cast<size_t>(n_rows * ldN) * sizeof(float), p->stream));
            hipLaunchKernelGGL(half_identity_kernel, dim3(static_cast<unsigned>((n_rows + 255) / 256)), dim3(256), 0,
                               p->stream, p->result, ldN, static_cast<int>(N), p->d_shard_out_rows,
                               static_cast<int>(n_rows), static_cast<int>(r0));
            HIP_TRY(hipGetLastError());
        } else {
            // Psi_1 = 1/2 I over the top founders
            // (materialised only for kernels that read it: level_identity_kernel and the fused
            //  small-level run start from the indices)
            const int64_t n0 = pl.cut_sizes[0], ld0 = pl.ld[0];
            if (kernel == 1 || (!identity_source(0, p->tun) && sparse_k < 0)) {
                HIP_TRY(hipMemsetAsync(p->buf[0], 0, static_cast<size_t>((n0 + 1) * ld0) * sizeof(float), p->stream));
                hipLaunchKernelGGL(half_identity_kernel, dim3(static_cast<unsigned>((n0 + 255) / 256)), dim3(256), 0,
                                   p->stream, p->buf[0], ld0, static_cast<int>(n0), static_cast<const int *>(nullptr),
                                   static_cast<int>(n0), 0);
                HIP_TRY(hipGetLastError());
            }
            for (int s = 0; s < n_steps; ++s) {
                const LevelStep &st = pl.steps[s];
                const float *psi = p->buf[bid[s]];
                const bool last = s == n_steps - 1;
                if (p->step_hook) p->step_hook(s, n_steps, p->step_hook_user);
                if (s <= sparse_k) {
                    // cuts 0..sparse_k are row lists: a list step, or (s == sparse_k) the step that writes cut s+1 as a dense matrix
                    std::string serr;
                    rc = s < sparse_k ? genphi::sparse_levels_enqueue_step(p->sparse, s, p->stream, serr)
                                      : genphi::sparse_levels_enqueue_dense(p->sparse, p->buf[bid[s + 1]], false, false, st.ld, st.width, p->stream, serr);
                    if (rc == GENPHI_OK && s == sparse_k) rc = genphi::sparse_levels_enqueue_flags(p->sparse, p->stream, serr);
                    if (rc) return fail(rc, "sparse levels: " + serr);
                    if (timing) HIP_TRY(hipEventRecord(p->events[s + 1], p->stream));
                    continue;
                }
                // a run of >= 2 small intermediate steps goes through ONE launch (levels_small_kernel)
                if (kernel == 0 && !small_off) {
                    int e = s;
                    // (a step that reads or writes by slot -- tiny LDS budgets in tests make WIDE steps of small cuts -- is not a small step)
                    while (e < n_steps - 1 && pl.steps[e].n_prev <= kSmallMax && pl.steps[e].n <= kSmallMax && !pl.steps[e].src_slots && !pl.steps[e].stay) ++e;
                    if (e - s >= 2) {
                        const size_t lds = (2 * kSmallPitch * kSmallPitch + 3 * kSmallMax) * sizeof(float);
                        HIP_TRY(set_max_lds(reinterpret_cast<const void *>(levels_small_kernel), lds));
                        // (the hook of every step the fused run covers, BEFORE the run is handed to the GPU: include/genphi.h promises that order)
                        if (p->step_hook) for (int k = s + 1; k < e; ++k) p->step_hook(k, n_steps, p->step_hook_user);
                        hipLaunchKernelGGL(levels_small_kernel, dim3(1), dim3(1024), lds, p->stream, p->d_small + s, e - s,
                                           psi, static_cast<long long>(pl.ld[s]), s == 0 ? 1 : 0, p->buf[bid[e]],
                                           static_cast<long long>(pl.ld[e]), p->d_cert + p->cert_off[e], cert_threshold(p->tun));
                        HIP_TRY(hipGetLastError());
                        // per-level timing: ONE event for the run, booked on its first step (an event
                        // record costs more than a fused level)
                        if (timing) {
                            HIP_TRY(hipEventRecord(p->events[e], p->stream));
                            for (int k = s; k < e; ++k) ev_after[k] = e;
                        }
                        s = e - 1;
                        continue;
                    }
                }
                if (!last) {
                    float *out = p->buf[bid[s + 1]];
                    if (st.mode == genphi::kModeWide) {
                        // the first in-place step of a run: the all-zero "none" row P of the run's matrix
                        if (p->stay_active && st.stay && !(s > 0 && pl.steps[s - 1].stay))
                            HIP_TRY(hipMemsetAsync(out + static_cast<long long>(st.P) * st.ld, 0, static_cast<size_t>(st.ld) * sizeof(float), p->stream));
                        rc = launch_wide_level(p, s, psi, out, kernel);
                    }
                    else if (p->sh_valid && p->sh_steps[s].rows && s >= prune_min_step)
                        rc = launch_level(p, main_ctx(p, s), psi, out, p->sh_steps[s].rows, nullptr, p->sh_steps[s].n_rows, kernel,
                                          p->sh_steps[s].groups);
                    else {
                        rc = ensure_groups(p, s);
                        if (rc == GENPHI_OK)
                            rc = launch_level(p, main_ctx(p, s), psi, out, p->dsteps[s].work, nullptr, static_cast<int>(st.n), kernel,
                                              p->dsteps[s].groups);
                    }
                    if (rc) return rc;
                    // the all-zero "none" row of this level (FULL / SPLIT / WIDE launches write it themselves)
                    if (kernel == 1 && st.mode != genphi::kModeWide)
                        HIP_TRY(hipMemsetAsync(out + st.n * st.ld, 0, static_cast<size_t>(st.ld) * sizeof(float), p->stream));
                } else {
                    float *lvl = last_by_slot ? p->buf[bid[s + 1]] : p->final_tmp;      // (in place: the run's own matrix)
                    if (need_perm)
                        rc = launch_wide_level(p, s, psi, lvl, kernel);
                    else
                        rc = launch_level(p, main_ctx(p, s), psi, p->result, p->d_shard_rows, p->d_shard_out_rows,
                                          static_cast<int>(n_rows), kernel, p->shard_groups);
                    if (rc) return rc;
                    if (timing) HIP_TRY(hipEventRecord(p->events[n_steps + 1], p->stream));
                    if (need_perm) {
                        // columns per thread (registers: 2 per column) and LDS segments of the source row
                        const int per_thread = static_cast<int>((ldN + 1023) / 1024);
                        const int n_chunks = (per_thread + 43) / 44;            // 44 columns per thread fit without spills (128 VGPRs; 48 spill):
                                                                                // one chunk -- every source row staged once -- up to 45,056 columns
                        const int cpt = ((per_thread + n_chunks - 1) / n_chunks + 3) / 4 * 4;
                        const int seg_floats = 36864;
                        const int n_segs = static_cast<int>((ldN + seg_floats - 1) / seg_floats);
                        const size_t lds = static_cast<size_t>(std::min<int64_t>(seg_floats, ldN)) * sizeof(float);
                        const dim3 grid(static_cast<unsigned>(n_rows * n_chunks));
                        if (n_chunks == 1 && ldN < 65535 && !p->tun.colperm_plain) {
                            // one chunk, 16-bit source columns: the persistent pipelined form
                            const int seg_p = 32768;                      // 16 float4 per thread and piece (18 -- 36,864 floats -- spill)
                            const int n_segs_p = static_cast<int>((ldN + seg_p - 1) / seg_p);
                            const size_t lds_p = static_cast<size_t>(std::min<int64_t>(seg_p, ldN)) * sizeof(float);
                            HIP_TRY(set_max_lds(reinterpret_cast<const void *>(colperm_pipe_kernel<512, 88, 16>), lds_p));
                            hipLaunchKernelGGL((colperm_pipe_kernel<512, 88, 16>), dim3(static_cast<unsigned>(std::min<int64_t>(n_rows, p->n_cus))), dim3(512), lds_p, p->stream,
                                               lvl, p->result, ldN, static_cast<int>(N), last_by_slot ? p->d_final_slots : p->d_final_perm, seg_p, n_segs_p,
                                               static_cast<int>(r0), static_cast<int>(n_rows));
                        } else
#define GENPHI_CP(C) if (cpt <= C) { HIP_TRY(set_max_lds(reinterpret_cast<const void *>(colperm_kernel<C>), lds)); \
                        hipLaunchKernelGGL(colperm_kernel<C>, grid, dim3(1024), lds, p->stream, lvl, p->result, ldN, \
                                           static_cast<int>(N), last_by_slot ? p->d_final_slots : p->d_final_perm, n_chunks, seg_floats, n_segs, static_cast<int>(r0)); } else
                        GENPHI_CP(8) GENPHI_CP(16) GENPHI_CP(24) GENPHI_CP(32) GENPHI_CP(40) GENPHI_CP(44)
                        return fail(GENPHI_ERR_ARG, "internal: colperm geometry");
#undef GENPHI_CP
                        HIP_TRY(hipGetLastError());
                    }
                }
                if (timing) HIP_TRY(hipEventRecord(p->events[s + 1], p->stream));
            }
        }
        return GENPHI_OK;
    };

    // A sweep is 1 + L-1 small launches; deep pedigrees (hundreds of tiny levels) are bound by
    // launch overhead.  After one eager run with the same arguments (which sizes buffers and
    // opts kernels into their LDS), the sweep is captured into a hipGraph and replayed.
    // Timing runs stay eager (they need events between the launches).
    const bool graphs_off = p->tun.no_graph;
    const long long key[5] = {kernel, static_cast<long long>(r0), static_cast<long long>(r1), (need_perm ? 1 : 0) | (sparse_k >= 0 ? 2 : 0), p->alloc_gen};
    const bool same_as_eager = p->eager_valid && std::memcmp(key, p->eager_key, sizeof(key)) == 0;
    const bool use_graph = !timing && !graphs_off && !(opts && (opts->flags & GENPHI_FLAG_NO_GRAPH)) && same_as_eager && n_steps >= 8 && !p->step_hook;
    if (timing) HIP_TRY(hipEventRecord(p->events[0], p->stream));
    if (use_graph) {
        if (!p->graph_exec || std::memcmp(key, p->graph_key, sizeof(key)) != 0) {
            if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
            HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
            const int erc = enqueue();
            hipGraph_t graph = nullptr;
            const hipError_t ce = hipStreamEndCapture(p->stream, &graph);
            if (erc != GENPHI_OK) { if (graph) (void)hipGraphDestroy(graph); return erc; }
            if (ce != hipSuccess) return fail(GENPHI_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            const hipError_t ie = hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { p->graph_exec = nullptr; return fail(GENPHI_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
            std::memcpy(p->graph_key, key, sizeof(key));
        }
        HIP_TRY(hipGraphLaunch(p->graph_exec, p->stream));
    } else {
        rc = enqueue();
        if (rc) return rc;
        std::memcpy(p->eager_key, key, sizeof(key));
        p->eager_valid = true;
    }
    if (stats) {
        for (int s = 0; s < n_steps && s < GENPHI_MAX_STAT_LEVELS; ++s) {
            const LevelStep &st = pl.steps[s];
            int64_t rows = st.n;
            if (s == n_steps - 1) rows = need_perm ? st.n : n_rows;
            else if (st.mode != genphi::kModeWide && p->sh_valid && p->sh_steps[s].rows && s >= prune_min_step) rows = p->sh_steps[s].n_rows;
            stats->level_rows[s] = rows;
        }
    }
    if (timing && n_steps == 0) HIP_TRY(hipEventRecord(p->events[1], p->stream));
    trace.mark("sweep enqueued");
    HIP_TRY(hipStreamSynchronize(p->stream));
    trace.mark("sweep done");
    if (sparse_k >= 0 && !genphi::sparse_levels_flags_ok(p->sparse))
        return fail(GENPHI_ERR_DEVICE, "sparse levels: a row list did not have the length the plan recorded (internal error)");
    if (timing) {
        const int ne = std::max(n_steps, 1);
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->events[0], p->events[ne]));
        stats->total_ms = ms;
        for (int s = 0; s < ne; ++s) {
            const int e0 = s == 0 ? 0 : ev_after[s - 1], e1 = ev_after[s];
            ms = 0.f;
            if (e1 != e0) HIP_TRY(hipEventElapsedTime(&ms, p->events[e0], p->events[e1]));
            stats->level_ms[s] = ms;
        }
        stats->final_ms = stats->level_ms[ne - 1];
        if (n_steps > 0) {
            HIP_TRY(hipEventElapsedTime(&ms, p->events[n_steps + 1], p->events[n_steps]));
            stats->perm_ms = ms;
        }
        stats->timed = 1;
    }
    return GENPHI_OK;
}

int genphi_result_device(const genphi_plan *p, const float **d_ptr, int64_t *ld, int64_t *row_begin, int64_t *n_rows)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (p->res_f64) return fail(GENPHI_ERR_ARG, "the resident result is Float64 (GENPHI_FLAG_STORAGE_F64): use genphi_result_to_host_f64 / genphi_result_entries");
    if (d_ptr) *d_ptr = p->result;
    if (ld) *ld = p->res_ld;
    if (row_begin) *row_begin = p->res_row_begin;
    if (n_rows) *n_rows = p->res_n_rows;
    return GENPHI_OK;
}

int genphi_result_to_host_f64(genphi_plan *p, double *out)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (p->res_n_rows == 0 || p->plan.n_pro == 0) return GENPHI_OK;
    if (!out) return fail(GENPHI_ERR_ARG, "out is NULL");
    if (!p->res_f64 || !p->on_device || !p->result64)
        return fail(GENPHI_ERR_ARG, "no resident Float64 result: call genphi_compute_device with GENPHI_FLAG_STORAGE_F64 first");
    HIP_TRY(hipSetDevice(p->device));
    const size_t N = static_cast<size_t>(p->plan.n_pro);
    HIP_TRY(hipMemcpy2D(out, N * sizeof(double), p->result64, static_cast<size_t>(p->res_ld) * sizeof(double), N * sizeof(double),
                        static_cast<size_t>(p->res_n_rows), hipMemcpyDeviceToHost));
    return GENPHI_OK;
}

}  // extern "C"
// tmp[q * nr + r] = src[r * w + q] for q < 16 and the first nr - nr % 8 rows r (genphi_result_to_host's mirror pass): 8 x 8
// transposes in AVX2 registers.  Returns the rows done.  (Host code of one function: the library is not built with -mavx2.)
__attribute__((target("avx2"))) static size_t mirror_gather16_avx2(const float *src, size_t w, size_t nr, float *tmp)
{
    size_t r = 0;
    for (; r + 8 <= nr; r += 8) {
        for (int h = 0; h < 2; ++h) {                       // columns [8 h, 8 h + 8)
            __m256 v[8];
            for (int k = 0; k < 8; ++k) v[k] = _mm256_loadu_ps(src + (r + k) * w + 8 * h);
            const __m256 t0 = _mm256_unpacklo_ps(v[0], v[1]), t1 = _mm256_unpackhi_ps(v[0], v[1]);
            const __m256 t2 = _mm256_unpacklo_ps(v[2], v[3]), t3 = _mm256_unpackhi_ps(v[2], v[3]);
            const __m256 t4 = _mm256_unpacklo_ps(v[4], v[5]), t5 = _mm256_unpackhi_ps(v[4], v[5]);
            const __m256 t6 = _mm256_unpacklo_ps(v[6], v[7]), t7 = _mm256_unpackhi_ps(v[6], v[7]);
            const __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xee);
            const __m256 u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xee);
            const __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xee);
            const __m256 u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xee);
            _mm256_storeu_ps(tmp + (8 * h + 0) * nr + r, _mm256_permute2f128_ps(u0, u4, 0x20));
            _mm256_storeu_ps(tmp + (8 * h + 1) * nr + r, _mm256_permute2f128_ps(u1, u5, 0x20));
            _mm256_storeu_ps(tmp + (8 * h + 2) * nr + r, _mm256_permute2f128_ps(u2, u6, 0x20));
            _mm256_storeu_ps(tmp + (8 * h + 3) * nr + r, _mm256_permute2f128_ps(u3, u7, 0x20));
            _mm256_storeu_ps(tmp + (8 * h + 4) * nr + r, _mm256_permute2f128_ps(u0, u4, 0x31));
            _mm256_storeu_ps(tmp + (8 * h + 5) * nr + r, _mm256_permute2f128_ps(u1, u5, 0x31));
            _mm256_storeu_ps(tmp + (8 * h + 6) * nr + r, _mm256_permute2f128_ps(u2, u6, 0x31));
            _mm256_storeu_ps(tmp + (8 * h + 7) * nr + r, _mm256_permute2f128_ps(u3, u7, 0x31));
        }
    }
    return r;
}
extern "C" {

int genphi_result_to_host(genphi_plan *p, float *out)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (p->res_n_rows == 0 || p->plan.n_pro == 0) return GENPHI_OK;
    if (!out) return fail(GENPHI_ERR_ARG, "out is NULL");
    if (p->res_f64) {
        // Float64 sweep: deliver RN32 of the Float64 values (ONE rounding, like gen.f, src/compute.jl:500-511)
        const size_t n = static_cast<size_t>(p->res_n_rows) * static_cast<size_t>(p->plan.n_pro);
        std::vector<double> tmp(n);
        const int rc = genphi_result_to_host_f64(p, tmp.data());
        if (rc) return rc;
        for (size_t k = 0; k < n; ++k) out[k] = static_cast<float>(tmp[k]);
        return GENPHI_OK;
    }
    if (!p->on_device || !p->result) return fail(GENPHI_ERR_DEVICE, "no resident result: call genphi_compute_device first");
    HIP_TRY(hipSetDevice(p->device));
    const size_t N = static_cast<size_t>(p->plan.n_pro);
    const size_t rows = static_cast<size_t>(p->res_n_rows);
    HIP_TRY(hipStreamSynchronize(p->stream));
    // Large results go through a ring of pinned staging buffers: every worker thread owns a
    // stream and two pinned chunks, the DMA engine fills one chunk (device pitch -> dense rows)
    // while the thread copies the other into the caller's pageable array.  A plain hipMemcpy2D
    // into pageable memory is staged by the runtime on ONE thread (~17 GB/s; 23 GB/s with 8
    // concurrent calls); the PCIe Gen5 link carries more than twice that.
    const size_t bytes = rows * N * sizeof(float);
    // A FULL result is bit-symmetric (every level is: both (i, j) and (j, i) are the same Float64 expression), and the plain copy is
    // bound by the PCIe link (55 GB/s into warm pages, 53 with first-touch page faults), so GENPHI_D2H_SYM=1 sends only the tiles on and
    // above the diagonal across the link and lets the worker threads mirror them into the lower triangle on the host.  OPT-IN: measured
    // at 1e5 probands (profiles/microbench/out/r04_d2h_symmetric_vs_plain_cfg4.out) it takes 440-1140 ms warm and 615-1420 ms into fresh
    // pages against a steady 724 / 756 ms for the plain copy -- the mirror pass makes the HOST the bottleneck, and a GPU box gives the
    // process 16 CPUs (cgroup quota): whenever the 16 workers, the Python thread and the runtime's helpers exceed it, the kernel throttles
    // the lot.  On a host with cores to spare it is the faster path; here it is not reliably so, hence not the default.
    bool sym = rows == N && p->res_row_begin == 0 && N >= 2 && !p->tun.d2h_pageable && p->tun.d2h_sym == 1;
    int n_thr = 1;
    if (bytes >= (size_t(64) << 20) || sym) {              // (one worker per 32 MB up to 8: the ring is kept between calls, so mid-size results use it too)
        n_thr = sym ? 16 : static_cast<int>(std::min<size_t>(8, bytes >> 25));
        if (p->tun.d2h_threads > 0) n_thr = std::max(1, std::min(32, p->tun.d2h_threads));
    }
    const size_t row_bytes = N * sizeof(float);
    const size_t tile_r = p->tun.d2h_tile_rows > 0 ? static_cast<size_t>(p->tun.d2h_tile_rows) : 256;
    const size_t tile_c = p->tun.d2h_tile_cols > 0 ? static_cast<size_t>(p->tun.d2h_tile_cols) : 8192;
    // (16 MB chunks; 4 MB for results below 2 GB, whose workers have only a few chunks each to overlap the DMA with the host copy)
    const size_t chunk_mb = p->tun.d2h_chunk_mb > 0 ? static_cast<size_t>(p->tun.d2h_chunk_mb) : (bytes < (size_t(2) << 30) ? 4 : 16);
    const size_t chunk_rows = std::max<size_t>(1, (chunk_mb << 20) / row_bytes);
    const size_t chunk_bytes = sym ? std::max<size_t>(tile_r * std::min(tile_c, N) * sizeof(float), 4096) : chunk_rows * row_bytes;
    bool pinned = (n_thr > 1 || sym) && !p->tun.d2h_pageable;
    // (the ring belongs to the device, not to the plan: pinning 256 MB costs 60-100 ms and unpinning them 80 ms -- per one-shot call
    // when every plan had its own; devcache.h)
    genphi::PinnedRing *ring = nullptr;
    std::unique_lock<std::mutex> ring_lock;
    if (pinned) {
        ring = &genphi::pinned_ring(p->device);
        ring_lock = std::unique_lock<std::mutex>(ring->mu);
        if (!genphi::pinned_ring_reserve(*ring, static_cast<size_t>(2 * n_thr), chunk_bytes, static_cast<size_t>(n_thr))) {
            pinned = false;                             // could not pin: fall back to direct copies
            ring_lock.unlock();
            ring = nullptr;
        }
    }
    if (!pinned) sym = false;
    std::vector<hipError_t> errs(n_thr, hipSuccess);
    if (sym) {
        // items: (row block I, column tile J) with the tile's columns clipped to [max(c0, a), c1): on or right of the diagonal block
        struct Item { uint32_t a, b, cs, c1; };
        std::vector<Item> items;
        for (size_t a = 0; a < N; a += tile_r) {
            const size_t b = std::min(N, a + tile_r);
            for (size_t c0 = a / tile_c * tile_c; c0 < N; c0 += tile_c) {
                const size_t cs = std::max(c0, a), c1 = std::min(N, c0 + tile_c);
                items.push_back({static_cast<uint32_t>(a), static_cast<uint32_t>(b), static_cast<uint32_t>(cs), static_cast<uint32_t>(c1)});
            }
        }
        std::atomic<size_t> next{0};
        const size_t src_pitch = static_cast<size_t>(p->res_ld) * sizeof(float);
        const bool avx2 = __builtin_cpu_supports("avx2") != 0;
        auto worker = [&](int t) {
            hipError_t e = hipSetDevice(p->device);
            if (e != hipSuccess) { errs[t] = e; return; }
            hipStream_t st = ring->stream[t];
            float *pb[2] = {static_cast<float *>(ring->chunk[2 * t]), static_cast<float *>(ring->chunk[2 * t + 1])};
            auto issue = [&](const Item &it, float *dst) {
                const size_t w = it.c1 - it.cs;
                return hipMemcpy2DAsync(dst, w * sizeof(float), p->result + static_cast<size_t>(it.a) * static_cast<size_t>(p->res_ld) + it.cs, src_pitch,
                                        w * sizeof(float), it.b - it.a, hipMemcpyDeviceToHost, st);
            };
            // (waits sleep instead of spinning -- a blocking-sync event per buffer: the host side is the bottleneck of this copy,
            // and a GPU box gives a process 16 CPUs; spinning waiters take them from the threads that mirror tiles)
            hipEvent_t evb[2] = {nullptr, nullptr};
            for (hipEvent_t &x : evb)
                if ((e = hipEventCreateWithFlags(&x, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) { errs[t] = e; return; }
            size_t cur = next.fetch_add(1);
            if (cur >= items.size()) { for (hipEvent_t x : evb) (void)hipEventDestroy(x); return; }
            e = issue(items[cur], pb[0]);
            if (e == hipSuccess) e = hipEventRecord(evb[0], st);
            for (int k = 0; e == hipSuccess; ++k) {
                e = hipEventSynchronize(evb[k & 1]);        // item `cur` has landed in pb[k & 1]
                if (e != hipSuccess) break;
                const size_t nxt = next.fetch_add(1);
                if (nxt < items.size()) {                   // the DMA engine fills the other buffer meanwhile
                    e = issue(items[nxt], pb[(k + 1) & 1]);
                    if (e == hipSuccess) e = hipEventRecord(evb[(k + 1) & 1], st);
                }
                const Item it = items[cur];
                const float *blk = pb[k & 1];
                const size_t w = it.c1 - it.cs, nr = it.b - it.a;
                for (size_t r = 0; r < nr; ++r)             // the tile itself
                    std::memcpy(out + (it.a + r) * N + it.cs, blk + r * w, w * sizeof(float));
                // its mirror image: columns right of the diagonal block become the rows' entries [a, b), 16 columns (a cache
                // line of every tile row) at a time through a small buffer, written as runs of nr floats
                const size_t ts = std::max<size_t>(it.cs, it.b);
                constexpr size_t KB = 16;
                static thread_local std::vector<float> tmp;
                tmp.resize(KB * nr);
                for (size_t cb0 = ts; cb0 < it.c1; cb0 += KB) {
                    const size_t nb = std::min(KB, it.c1 - cb0);
                    const float *src = blk + (cb0 - it.cs);
                    if (nb == KB) {
                        size_t r = 0;
                        if (avx2) r = mirror_gather16_avx2(src, w, nr, tmp.data());       // 8 x 8 register transposes, whole multiples of 8 rows
                        for (; r < nr; ++r) {
                            const float *sr = src + r * w;
#pragma unroll
                            for (size_t q = 0; q < KB; ++q) tmp[q * nr + r] = sr[q];
                        }
                    } else {
                        for (size_t r = 0; r < nr; ++r)
                            for (size_t q = 0; q < nb; ++q) tmp[q * nr + r] = src[r * w + q];
                    }
                    for (size_t q = 0; q < nb; ++q) std::memcpy(out + (cb0 + q) * N + it.a, tmp.data() + q * nr, nr * sizeof(float));
                }
                if (nxt >= items.size()) break;
                cur = nxt;
            }
            for (hipEvent_t x : evb) (void)hipEventDestroy(x);
            errs[t] = e;
        };
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t) th.emplace_back(worker, t);
        for (auto &x : th) x.join();
        for (hipError_t e : errs)
            if (e != hipSuccess) return fail(GENPHI_ERR_DEVICE, std::string("genphi_result_to_host: ") + hipGetErrorString(e));
        return GENPHI_OK;
    }
    const bool d2h_stats = genphi::env_hook("GENPHI_D2H_STATS") != nullptr;
    auto copy_block = [&](int t) {
        const size_t r0 = rows * t / n_thr, r1 = rows * (t + 1) / n_thr;
        if (r1 == r0) return;
        hipError_t e = hipSetDevice(p->device);
        if (e != hipSuccess) { errs[t] = e; return; }
        const size_t src_pitch = static_cast<size_t>(p->res_ld) * sizeof(float);
        if (!pinned) {
            errs[t] = hipMemcpy2D(out + r0 * N, row_bytes, p->result + r0 * static_cast<size_t>(p->res_ld), src_pitch,
                                  row_bytes, r1 - r0, hipMemcpyDeviceToHost);
            return;
        }
        hipStream_t st = ring->stream[t];
        char *pb[2] = {static_cast<char *>(ring->chunk[2 * t]), static_cast<char *>(ring->chunk[2 * t + 1])};
        const size_t n_chunks = (r1 - r0 + chunk_rows - 1) / chunk_rows;
        auto issue = [&](size_t c) {
            const size_t a = r0 + c * chunk_rows, b = std::min(r1, a + chunk_rows);
            return hipMemcpy2DAsync(pb[c & 1], row_bytes, p->result + a * static_cast<size_t>(p->res_ld), src_pitch,
                                    row_bytes, b - a, hipMemcpyDeviceToHost, st);
        };
        e = issue(0);
        double t_wait = 0.0, t_host = 0.0;
        auto clk = [] { return std::chrono::steady_clock::now(); };
        for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
            const auto t0 = clk();
            e = hipStreamSynchronize(st);               // chunk c has landed in pb[c & 1]
            if (e != hipSuccess) break;
            if (c + 1 < n_chunks) e = issue(c + 1);     // the DMA engine fills the other buffer meanwhile
            const auto t1 = clk();
            const size_t a = r0 + c * chunk_rows, b = std::min(r1, a + chunk_rows);
            std::memcpy(out + a * N, pb[c & 1], (b - a) * row_bytes);
            if (d2h_stats) { t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count(); t_host += std::chrono::duration<double, std::milli>(clk() - t1).count(); }
        }
        if (d2h_stats) std::fprintf(stderr, "[genphi d2h] thread %d: %zu chunks, waiting for the DMA %.2f ms, copying into the caller's array %.2f ms\n", t, n_chunks, t_wait, t_host);
        errs[t] = e;
    };
    if (n_thr == 1) copy_block(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t) th.emplace_back(copy_block, t);
        for (auto &x : th) x.join();
    }
    for (hipError_t e : errs)
        if (e != hipSuccess) return fail(GENPHI_ERR_DEVICE, std::string("genphi_result_to_host: ") + hipGetErrorString(e));
    return GENPHI_OK;
}

int genphi_result_sums(genphi_plan *p, double *sum_all, double *sum_diag, int64_t *n_rows_out)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (sum_all) *sum_all = 0.0;
    if (sum_diag) *sum_diag = 0.0;
    if (n_rows_out) *n_rows_out = p->res_n_rows;
    if (p->res_n_rows == 0 || p->plan.n_pro == 0) return GENPHI_OK;
    if (p->res_f64) return fail(GENPHI_ERR_ARG, "genphi_result_sums works on the Float32 result (phiMean's input type, src/compute.jl:454)");
    if (!p->on_device || !p->result) return fail(GENPHI_ERR_DEVICE, "no resident result: call genphi_compute_device first");
    HIP_TRY(hipSetDevice(p->device));
    const int64_t nr = p->res_n_rows;
    int rc = ensure_scratch(p, 2 * nr * sizeof(double));
    if (rc) return rc;
    double *d = reinterpret_cast<double *>(p->scratch);
    hipLaunchKernelGGL(row_sums_kernel, dim3(static_cast<unsigned>(nr)), dim3(256), 0, p->stream, p->result,
                       static_cast<long long>(p->res_ld), static_cast<int>(p->plan.n_pro), static_cast<int>(p->res_row_begin),
                       d, d + nr);
    hipError_t e = hipGetLastError();
    std::vector<double> h(2 * nr);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, 2 * nr * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) return fail(GENPHI_ERR_DEVICE, std::string("genphi_result_sums: ") + hipGetErrorString(e));
    double sa = 0.0, sd = 0.0;                      // fixed order: reproducible
    for (int64_t k = 0; k < nr; ++k) { sa += h[k]; sd += h[nr + k]; }
    if (sum_all) *sum_all = sa;
    if (sum_diag) *sum_diag = sd;
    return GENPHI_OK;
}

int genphi_result_entries(genphi_plan *p, int64_t n, const int64_t *rows, const int64_t *cols, double *out)
{
    if (!p) return fail(GENPHI_ERR_ARG, "plan is NULL");
    if (n < 0 || (n > 0 && (!rows || !cols || !out))) return fail(GENPHI_ERR_ARG, "genphi_result_entries: bad argument");
    if (n == 0) return GENPHI_OK;
    if (!p->on_device || !(p->res_f64 ? static_cast<const void *>(p->result64) : static_cast<const void *>(p->result)))
        return fail(GENPHI_ERR_DEVICE, "no resident result: call genphi_compute_device first");
    const int64_t N = p->plan.n_pro, r0 = p->res_row_begin, nr = p->res_n_rows;
    std::vector<long long> off(static_cast<size_t>(n));
    for (int64_t k = 0; k < n; ++k) {
        if (rows[k] < r0 || rows[k] >= r0 + nr || cols[k] < 0 || cols[k] >= N)
            return fail(GENPHI_ERR_ARG, "genphi_result_entries: entry (" + std::to_string(rows[k]) + ", " + std::to_string(cols[k]) +
                                        ") outside the resident rows [" + std::to_string(r0) + ", " + std::to_string(r0 + nr) + ") x [0, " + std::to_string(N) + ")");
        off[k] = static_cast<long long>(rows[k] - r0) * p->res_ld + cols[k];
    }
    HIP_TRY(hipSetDevice(p->device));
    const size_t off_bytes = (static_cast<size_t>(n) * sizeof(long long) + 255) / 256 * 256;
    int rc = ensure_scratch(p, off_bytes + static_cast<size_t>(n) * sizeof(double));
    if (rc) return rc;
    long long *d_off = reinterpret_cast<long long *>(p->scratch);
    double *d_val = reinterpret_cast<double *>(p->scratch + off_bytes);
    hipError_t e = hipMemcpyAsync(d_off, off.data(), n * sizeof(long long), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        if (p->res_f64)
            hipLaunchKernelGGL(gather_entries64_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, p->stream,
                               p->result64, d_off, n, d_val);
        else
            hipLaunchKernelGGL(gather_entries_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, p->stream,
                               p->result, d_off, n, d_val);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_val, n * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) return fail(GENPHI_ERR_DEVICE, std::string("genphi_result_entries: ") + hipGetErrorString(e));
    return GENPHI_OK;
}

#if GENPHI_WG_TIMES
int genphi_debug_wg_times(unsigned long long *out /* [1024][3] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 1024 * 3) == hipSuccess ? GENPHI_OK : GENPHI_ERR_DEVICE;
}
int genphi_debug_wg_clk(unsigned long long *out /* [1024][2] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_clk), sizeof(unsigned long long) * 1024 * 2) == hipSuccess ? GENPHI_OK : GENPHI_ERR_DEVICE;
}
int genphi_debug_wg_phases(unsigned long long *out /* [2][1024][16] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_phase), sizeof(unsigned long long) * 2 * 1024 * 16) == hipSuccess ? GENPHI_OK : GENPHI_ERR_DEVICE;
}
#endif

int genphi_phi_pairs(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pairs,
                     const int64_t *id_i, const int64_t *id_j, double *out, int32_t device)
{
    if (n_pairs < 0 || (n_pairs > 0 && (!id_i || !id_j || !out))) return fail(GENPHI_ERR_ARG, "genphi_phi_pairs: bad argument");
    if (n_pairs == 0) return GENPHI_OK;
    // probands = the individuals named (first occurrences); one Float64 sweep; one lookup per pair
    std::vector<int64_t> ids;
    ids.reserve(static_cast<size_t>(2 * n_pairs));
    for (int64_t k = 0; k < n_pairs; ++k) { ids.push_back(id_i[k]); ids.push_back(id_j[k]); }
    genphi_plan *pl = nullptr;
    int rc = plan_create_impl(n_ind, ind, father, mother, static_cast<int64_t>(ids.size()), ids.data(), /*indices_only=*/true, &pl);
    if (rc) return rc;
    // position of every named individual in the plan's proband order (duplicates collapsed in first-occurrence order)
    std::vector<int64_t> uniq;
    std::vector<int64_t> rows(n_pairs), cols(n_pairs);
    {
        std::vector<std::pair<int64_t, int64_t>> seen;      // (id, position); the sets are small
        auto pos_of = [&](int64_t id) -> int64_t {
            for (const auto &e : seen) if (e.first == id) return e.second;
            seen.emplace_back(id, static_cast<int64_t>(seen.size()));
            return seen.back().second;
        };
        if (ids.size() > 4096) {                            // many pairs: a sorted index instead of the linear scan
            std::vector<int64_t> first;
            std::vector<std::pair<int64_t, int64_t>> order;
            for (size_t k = 0; k < ids.size(); ++k) order.emplace_back(ids[k], static_cast<int64_t>(k));
            std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            std::vector<int64_t> first_occ(ids.size());
            for (size_t k = 0; k < order.size(); ++k)
                first_occ[order[k].second] = (k > 0 && order[k - 1].first == order[k].first) ? first_occ[order[k - 1].second] : order[k].second;
            std::vector<int64_t> rank(ids.size(), -1);
            int64_t next = 0;
            for (size_t k = 0; k < ids.size(); ++k) if (first_occ[k] == static_cast<int64_t>(k)) rank[k] = next++;
            for (int64_t k = 0; k < n_pairs; ++k) { rows[k] = rank[first_occ[2 * k]]; cols[k] = rank[first_occ[2 * k + 1]]; }
        } else {
            for (int64_t k = 0; k < n_pairs; ++k) { rows[k] = pos_of(id_i[k]); cols[k] = pos_of(id_j[k]); }
        }
    }
    genphi_opts o;
    std::memset(&o, 0, sizeof(o));
    o.device = device; o.flags = GENPHI_FLAG_STORAGE_F64;
    rc = genphi_compute_device(pl, &o, nullptr);
    if (rc == GENPHI_OK) rc = genphi_result_entries(pl, n_pairs, rows.data(), cols.data(), out);
    const std::string keep = g_last_error;
    genphi_plan_destroy(pl);
    if (rc) g_last_error = keep;
    return rc;
}

int genphi_compute_f32(genphi_plan *p, float *out, const genphi_opts *opts, genphi_stats *stats)
{
    int rc = genphi_compute_device(p, opts, stats);
    if (rc) return rc;
    return genphi_result_to_host(p, out);
}

}  // extern "C"
