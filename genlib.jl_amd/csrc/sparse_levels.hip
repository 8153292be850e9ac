// sparse_levels.hip -- the zero-aware leading levels of a gen.phi sweep (see sparse_levels.h).
//
// Replaces, for the cuts right below the founders, what src/compute.jl:291-301 does with a fresh dense matrix per level:
// the same entries (src/compute.jl:105-158 evaluated pair by pair) are produced row by row from the non-zero entries of
// the previous cut only.  A level step is Psi' = A Psi A^T with the diagonal rule of src/compute.jl:148-154, A's rows being a
// unit vector (member dragged along) or (e_father + e_mother) / 2 (new member; planner.h).  It is evaluated in two halves:
//
//     Y  = Psi A^T      row p of the SOURCE cut with its columns mapped to the columns of the cut being written: a non-zero
//                       (q, v) of row p goes to every CHILD of q (q itself when it is dragged along, weight 1; its new
//                       children, weight 1/2)
//     Psi'[i] = w_i (Y[A_i] + Y[B_i])       A_i, B_i = the sources of member i, w_i = 1 dragged, 1/2 new
//
// and the half that needs scattered index lookups -- the children of every column of a row -- is done ONCE per row, by the
// workgroup that has just built that row in LDS: a sparse cut is stored as its rows of Y.  The step that reads them streams
// two contiguous lists per output row; nothing else.
//
// Layout in HBM.  Sparse cut c: rowd[c][p] = (first entry, entries) per member p, ent[] = (column in cut c+1, integer value)
// pairs: the entries of row p of Y_c, unsorted, a column may occur twice (two parents of a child both related to p).  fm[c][i]
// per member i of cut c+1 = Psi_c[A_i][B_i], the kinship of i's parents, which its diagonal needs (written by the workgroup
// that built row A_i of Psi_c).  Two arenas alternate between consecutive cuts.  Where a row goes in the arena: the
// calibration run of a plan places rows with an atomic cursor and records (place, length) per row; every later sweep writes
// the row to that place -- lengths depend on the pedigree alone, and 24k same-address device-scope atomics per level cost
// more than the level itself (measured: 0.56 ms per level of cfg4 whatever its size).
//
// Arithmetic: integers.  Psi_c in units of 2^-(2c+1), Y_c in units of 2^-(2c+2) (exact for c <= 11, see sparse_levels.h),
// accumulated with LDS atomics; Float32 values are built by one exact conversion when the first dense matrix is written.  No
// rank words are needed: every grouping of the reference's sum is exact.
#include "sparse_levels.h"

#include "devcache.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../../include/genphi.h"

namespace genphi {

namespace {

typedef float f4_t __attribute__((ext_vector_type(4)));

struct SpArgs {
    // the source cut s as rows of Y_s (columns = members of cut s+1)
    const uint2 *ent_in;
    const uint2 *rowd_in;
    const unsigned *fm_in;        // per member i of cut s+1: Psi_s[A_i][B_i] in units of 2^-(2s+1)
    const int *srcA, *srcB, *ord; // per member of cut s+1: sources in cut s (n_prev = none), rank word (< 0: new)
    int n_prev, n;                // |cut s|, |cut s+1|
    unsigned half_out;            // 1/2 in units of cut s+1: 2^(2s+2)
    // row-list step: cut s+1 written as rows of Y_{s+1} (columns = members of cut s+2)
    uint2 *ent_out;
    uint2 *rowd_out;              // (place, length) of every row: lengths by the calibration run's counting launch, places by its scan;
    int count_only;               //   read by every other launch.  count_only != 0: the counting launch (no values, no lists written)
    const unsigned *stop;         // (calibration run) != nullptr: a word that names the first cut NOT to compute once a cut proved too dense (0: none
    unsigned level;               //   yet) -- launches for that cut and the ones behind it end at once; level = the cut this launch writes
    unsigned ent_cap;             // entries the arena written holds
    unsigned nz_lo, nz_hi;        // (calibration run, writing launches) nz_hi != 0: this launch writes the rows with nz_lo < non-zero entries <= nz_hi
                                  //   (nz_lo = 0: from empty rows on) -- the counting launch has left every row's count in rnz_out
    const int *chn_off;           // children of member q of cut s+1 in cut s+2: chn[chn_off[q] .. chn_off[q + 1])
    const unsigned *chn;          //   position in cut s+2 | 0x80000000 when the child is q itself (dragged: weight 1)
    const int *mt_off;            // mates of member p of cut s+1: mt[mt_off[p] .. mt_off[p + 1]) = (B, i) for every new member i of
    const uint2 *mt;              //   cut s+2 with A_i = p and a second parent B
    unsigned *fm_out;             // per member of cut s+2
    const int *rows;              // the members this launch computes (nullptr: member = workgroup index)
    int remap;                    // rows in the planner's work order: consecutive ones on one XCD (not for rows sorted by length: the long ones would share one)
    unsigned *rnz_out;            // (calibration run) non-zero entries of every row of Psi_{s+1}
    int wp;                       // bitmap words in LDS: workgroup size x an odd number
    int cap;                      // entries of one row of Psi_{s+1} the LDS holds
    unsigned *stat;               // [0] entries of Y written so far (calibration run), [2] 1 = a row or the arena overflowed, 2 = a row's length
                                  // differs from the plan's
    // sparse -> dense step
    void *out;                    // Float32 matrix (gen.phi) or Float64 (the Float64-storage sweep)
    long long ld;
    int width, chunk_cols, n_chunks;
    float unit_out;               // 2^-(2s+3): one integer unit of cut s+1
    const int *slot;              // the cut written is stored BY SLOT (a step that stays in place, planner.h: LevelStep::stay): row and column of
    int zrow;                     //   member i = slot[i]; nullptr: compactly (row = column = i).  zrow: the all-zero "none" row (n, or the slot capacity P)
};

// Y_0 = (1/2 I) A^T: row p is the children list of p itself; fm_0[i] = Psi_0[A_i][B_i] = 1/2 only when a member's two parents are
// the same founder; the counters of a sweep.  (src/compute.jl:271-274)
__global__ void __launch_bounds__(256) sparse_identity_kernel(uint2 *ent, uint2 *rowd, const int *ch_off, const unsigned *ch, int n0, int n_ch,
                                                              const int *srcA, const int *srcB, const int *ord, unsigned *fm, int n1,
                                                              unsigned *stat, int n_stat)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_ch) {
        const unsigned cw = ch[k];
        ent[k] = make_uint2(cw & 0x7fffffffu, (cw >> 31) + 1u);
    }
    if (k < n0) rowd[k] = make_uint2(static_cast<unsigned>(ch_off[k]), static_cast<unsigned>(ch_off[k + 1] - ch_off[k]));
    if (k < n1) fm[k] = (ord[k] < 0 && srcA[k] == srcB[k] && srcA[k] != n0) ? 1u : 0u;
    if (k < n_stat) stat[k] = 0u;
}

// consecutive work items on one XCD (workgroups b and b + 8 share an XCD and its L2): rows that share source lists then find them
// in that L2.  Bijective for any number of workgroups.
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct SrcRows {
    unsigned offA, lenA, offB, total;
};

__device__ __forceinline__ SrcRows src_rows(const SpArgs &a, int A, int B)
{
    const int none = a.n_prev;
    const uint2 ra = A != none ? a.rowd_in[A] : make_uint2(0u, 0u);
    const uint2 rb = B != none ? a.rowd_in[B] : make_uint2(0u, 0u);
    return SrcRows{ra.x, ra.y, rb.x, ra.y + rb.y};
}

// The entries of the (<= 2) source rows of an output row as ONE index space: contiguous, coalesced 8-byte loads.  The first kBatch
// entries of every thread stay in registers between the passes of the row-list step (most rows have no more); the rest is read again.
constexpr int kBatch = 4;

constexpr int kBatchDense = 16;      // the sparse -> dense step: one pass over the lists, so as many loads in flight as a thread has entries

template <int NT, int KB = kBatch>
struct EntryCache {
    uint2 en[KB];
};

__device__ __forceinline__ uint2 load_entry(const SpArgs &a, const SrcRows &r, unsigned e)
{
    return a.ent_in[e < r.lenA ? r.offA + e : r.offB + (e - r.lenA)];
}

template <int NT, int KB>
__device__ __forceinline__ void load_first(const SpArgs &a, const SrcRows &r, int tid, EntryCache<NT, KB> &c)
{
    if (r.total == 0u) return;
#pragma unroll
    for (int b = 0; b < KB; ++b) c.en[b] = load_entry(a, r, min(static_cast<unsigned>(tid + b * NT), r.total - 1u));
}

template <int NT, int KB, class F>
__device__ __forceinline__ void for_each_entry(const SpArgs &a, const SrcRows &r, int tid, const EntryCache<NT, KB> &c, F &&f)
{
#pragma unroll
    for (int b = 0; b < KB; ++b)
        if (static_cast<unsigned>(tid + b * NT) < r.total) f(c.en[b].x, c.en[b].y);
    for (unsigned e0 = tid + KB * NT; e0 < r.total; e0 += KB * NT) {
        uint2 en[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) en[b] = load_entry(a, r, min(e0 + b * NT, r.total - 1u));
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (e0 + b * NT < r.total) f(en[b].x, en[b].y);
    }
}

// exclusive prefix of v over the workgroup (thread order) and the workgroup's total; wsum: NT / 64 ints of LDS
template <int NT>
__device__ __forceinline__ int block_scan(int v, int tid, int *wsum, int &total)
{
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += up;
    }
    if constexpr (NT == 64) {
        total = __shfl(incl, 63);
        return incl - v;
    } else {
        __syncthreads();                                   // (wsum may still be read from the previous scan)
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            const int s = wsum[w];
            if (w < (tid >> 6)) base += s;
            total += s;
        }
        return base + incl - v;
    }
}

// ---- rows of Y_s -> rows of Y_{s+1}: one workgroup (a wavefront, or four for long rows) per member of cut s+1 ----------
// Pass 1 marks the columns of row i of Psi_{s+1} in an LDS bitmap; a scan over the bitmap gives every column its place in the
// (ascending) row and the row its length; pass 2 adds the contributions into the row's values in LDS (ds_add_u32); the
// diagonal of a new member is set from fm; then the row is used where it stands: the kinships its children's diagonals
// need are looked up (fm_out), and it leaves expanded to the columns of the next cut (a row of Y_{s+1}).
template <int NT, int KB>
__global__ void __launch_bounds__(NT) sparse_step_kernel(const SpArgs a)
{
    extern __shared__ unsigned lds_u[];
    unsigned *bm = lds_u;
    unsigned *vals = bm + a.wp;
    unsigned short *pre = reinterpret_cast<unsigned short *>(vals + a.cap);
    unsigned short *cols = pre + a.wp;
    __shared__ int wsum[NT / 64 + 1];
    const int tid = threadIdx.x;
    if (a.stop) {                                          // (workgroup-uniform)
        const unsigned first_skipped = *a.stop;
        if (first_skipped != 0u && a.level >= first_skipped) return;
    }
    const int w = a.remap ? xcd_remap(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x);
    const int i = a.rows ? a.rows[w] : w;
    if (a.nz_hi != 0u) {                                   // (workgroup-uniform) a row of another launch's class of lengths
        const unsigned nz = a.rnz_out[i];
        if (nz > a.nz_hi || (a.nz_lo != 0u && nz <= a.nz_lo)) return;
    }
    const int A = a.srcA[i], B = a.srcB[i];
    const bool new_i = a.ord[i] < 0;
    const int none = a.n_prev;
    const SrcRows r = src_rows(a, A, B);
    // (KB entries per thread in flight and in registers between the passes: four; eight for the four-wavefront rows was measured slower,
    // sparse_levels.h)
    EntryCache<NT, KB> ec;
    load_first(a, r, tid, ec);
    const uint2 place = a.count_only ? make_uint2(0u, 0u) : a.rowd_out[i];
    const unsigned fm_i = (new_i && A != none && B != none) ? a.fm_in[i] : 0u;
    const int mt0 = a.mt_off[i], mt1 = a.mt_off[i + 1];
    for (int w = tid; w < a.wp; w += NT) bm[w] = 0u;
    __syncthreads();
    // pass 1: which columns
    for_each_entry(a, r, tid, ec, [&](unsigned c, unsigned) { atomicOr(&bm[c >> 5], 1u << (c & 31u)); });
    if (new_i && tid == 0) atomicOr(&bm[i >> 5], 1u << (i & 31));
    __syncthreads();
    // every thread owns T consecutive bitmap words (T odd: no bank conflicts between the lanes)
    const int T = a.wp / NT, w0 = tid * T;
    int cnt = 0;
    for (int w = w0; w < w0 + T; ++w) cnt += __popc(bm[w]);
    int total;
    int pos = block_scan<NT>(cnt, tid, wsum, total);
    if (a.count_only) {
        // the counting launch of a calibration run: the row's non-zero entries and the length of its row of Y_{s+1} (every column's
        // children), straight from the bitmap -- no column list, so a row of any length is counted
        int len = 0;
        for (int w = w0; w < w0 + T; ++w) {
            unsigned bits = bm[w];
            while (bits) {
                const int q = w * 32 + __ffs(bits) - 1;
                len += a.chn_off[q + 1] - a.chn_off[q];
                bits &= bits - 1u;
            }
        }
        int ltot;
        (void)block_scan<NT>(len, tid, wsum, ltot);
        if (tid == 0) {
            const bool fits = total <= a.cap;              // (the launch that writes the row holds it in LDS: longer rows void the cut)
            a.rowd_out[i] = make_uint2(0u, fits ? static_cast<unsigned>(ltot) : 0u);
            a.rnz_out[i] = static_cast<unsigned>(total);
            if (!fits) atomicOr(&a.stat[2], 1u);
        }
        return;
    }
    if (total > a.cap) {                                   // (workgroup-uniform) the row does not fit: the cut is too dense to stay sparse
        if (tid == 0) atomicOr(&a.stat[2], 1u);
        return;
    }
    for (int w = w0; w < w0 + T; ++w) {
        pre[w] = static_cast<unsigned short>(pos);
        unsigned bits = bm[w];
        while (bits) {
            const int b = __ffs(bits) - 1;
            cols[pos++] = static_cast<unsigned short>(w * 32 + b);
            bits &= bits - 1u;
        }
    }
    for (int t = tid; t < total; t += NT) vals[t] = 0u;
    __syncthreads();
    // pass 2: the values, in units of 2^-(2s+3)
    const unsigned wi = new_i ? 1u : 2u;
    for_each_entry(a, r, tid, ec, [&](unsigned c, unsigned m) {
        if (new_i && c == static_cast<unsigned>(i)) return;               // the diagonal of a new member is not a sum of this kind
        const int at = pre[c >> 5] + __popc(bm[c >> 5] & ((1u << (c & 31u)) - 1u));
        atomicAdd(&vals[at], m * wi);
    });
    __syncthreads();
    if (new_i && tid == 0) {
        const int at = pre[i >> 5] + __popc(bm[i >> 5] & ((1u << (i & 31)) - 1u));
        vals[at] = a.half_out + 2u * fm_i;                                // 1/2 + Psi[A][B]/2 when both parents exist, src/compute.jl:148-154
    }
    __syncthreads();
    // the kinship of i with each of its mates: the diagonal of their children in the next cut
    for (int k = mt0 + tid; k < mt1; k += NT) {
        const uint2 m = a.mt[k];
        const unsigned word = bm[m.x >> 5], bit = 1u << (m.x & 31u);
        a.fm_out[m.y] = (word & bit) ? vals[pre[m.x >> 5] + __popc(word & (bit - 1u))] : 0u;
    }
    // the row leaves as a row of Y_{s+1}: every entry (q, v) goes to the children of q.  kBatch entries per thread at a time, their
    // children ranges loaded together (entry -> range -> children are dependent round trips through L2), one scan per batch; the
    // order of a row's entries in Y is free.
    auto expand = [&](unsigned off, unsigned limit) -> unsigned {
        unsigned run = 0u;
        for (int t0 = 0; t0 < total; t0 += KB * NT) {
            int k0[KB], k1[KB];
            unsigned v[KB];
            int mine = 0;
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int t = t0 + b * NT + tid;
                const bool ok = t < total;
                const int q = ok ? cols[t] : 0;
                v[b] = ok ? vals[t] : 0u;
                k0[b] = a.chn_off[q];
                k1[b] = ok ? a.chn_off[q + 1] : k0[b];
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) mine += k1[b] - k0[b];
            int it_total;
            unsigned at = run + static_cast<unsigned>(block_scan<NT>(mine, tid, wsum, it_total));
#pragma unroll
            for (int b = 0; b < KB; ++b)
                for (int k = k0[b]; k < k1[b]; ++k, ++at) {
                    const unsigned cw = a.chn[k];
                    if (at < limit) a.ent_out[off + at] = make_uint2(cw & 0x7fffffffu, v[b] * ((cw >> 31) + 1u));
                }
            run += static_cast<unsigned>(it_total);
        }
        return run;
    };
    const unsigned run = expand(place.x, place.y);
    if (run != place.y && tid == 0) atomicOr(&a.stat[2], 2u);
}

// places of the rows of a cut from their lengths (calibration run): rowd[i].x = sum of rowd[j].y over j < i; the total goes to stat[0];
// a cut whose lists do not fit the arena is voided (every length 0; flag 8 and the entries it needs in stat[1]: the host enlarges the
// arena and runs the cut again).  One workgroup: a cut has < 65,535 rows.
__global__ void __launch_bounds__(1024) sparse_place_kernel(uint2 *rowd, int n, unsigned ent_cap, unsigned long long stop_entries, unsigned *stat,
                                                            unsigned *stop, unsigned level)
{
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    if (*stop != 0u && level >= *stop) {                   // (a cut before this one was too dense: this one does not exist)
        if (tid == 0) atomicOr(&stat[2], 4u);
        return;
    }
    const int per = (n + 1023) / 1024, i0 = tid * per, i1 = min(n, i0 + per);
    unsigned long long sum = 0;
    for (int i = i0; i < i1; ++i) sum += rowd[i].y;
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scan of the 1024 partial sums
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = part[1023];
    const bool ok = total <= static_cast<unsigned long long>(ent_cap);
    unsigned long long off = part[tid] - sum;
    for (int i = i0; i < i1; ++i) {
        const unsigned len = rowd[i].y;
        rowd[i] = ok ? make_uint2(static_cast<unsigned>(off), len) : make_uint2(0u, 0u);
        off += len;
    }
    if (tid == 0) {
        stat[0] = ok ? static_cast<unsigned>(total) : 0u;
        if (!ok) {
            stat[1] = total > 0xffffffffull ? 0xffffffffu : static_cast<unsigned>(total);
            atomicOr(&stat[2], 8u);
        }
        // a void cut ends the run here; one too dense to be worth a list step behind it (its lists alone are a quarter of a dense level's
        // bytes) is still written -- it may be the last sparse one -- and ends the run behind it
        if (!ok || stat[2] != 0u) *stop = level;
        else if (total > stop_entries) *stop = level + 1u;
    }
}

// four consecutive entries of the dense matrix from their integer values (exact conversions), as non-temporal 16-byte stores; Float64
// for the Float64-storage sweep (gen.f, pairwise phi: src/compute.jl:66-95 works in Float64 throughout)
typedef double d2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store4(float *dst, unsigned x, unsigned y, unsigned z, unsigned w, float unit)
{
    const f4_t v = {static_cast<float>(x) * unit, static_cast<float>(y) * unit, static_cast<float>(z) * unit, static_cast<float>(w) * unit};
    __builtin_nontemporal_store(v, reinterpret_cast<f4_t *>(dst));
}
__device__ __forceinline__ void store4(double *dst, unsigned x, unsigned y, unsigned z, unsigned w, float unit)
{
    const double u = static_cast<double>(unit);
    const d2v_t a = {static_cast<double>(x) * u, static_cast<double>(y) * u}, b = {static_cast<double>(z) * u, static_cast<double>(w) * u};
    __builtin_nontemporal_store(a, reinterpret_cast<d2v_t *>(dst));
    __builtin_nontemporal_store(b, reinterpret_cast<d2v_t *>(dst + 2));
}

// ---- rows of Y_k -> the dense matrix of cut k+1: one workgroup per (row, chunk of columns) -------------------------------
// The chunk's entries are accumulated in LDS (integer units) and leave as whole 16-byte stores, zeros included: the step
// writes what a FULL / SPLIT row kernel writes (columns [0, width) of every row, the all-zero row n) and reads only lists.
template <typename OutT>
__global__ void __launch_bounds__(256) sparse_dense_kernel(const SpArgs a)
{
    extern __shared__ unsigned acc[];
    OutT *const outm = static_cast<OutT *>(a.out);
    const int tid = threadIdx.x;
    // Workgroups b, b + 8, b + 16 ... run on the same XCD: the chunks of one row go there back to back, and so do the rows that follow
    // it in the planner's work order (siblings adjacent, families chained along shared mothers) -- the lists of a row's sources are then
    // read from HBM once per XCD visit instead of once per chunk and child (measured on cfg4: 1.98 GB of HBM reads for 0.33 GB of lists).
    const int grp = blockIdx.x / (8 * a.n_chunks), rest = blockIdx.x - grp * 8 * a.n_chunks;
    const int w = grp * 8 + (rest & 7), chunk = rest >> 3;
    if (w > a.n) return;
    const int i = w < a.n ? (a.rows ? a.rows[w] : w) : a.n;
    const int c0 = chunk * a.chunk_cols;
    if (i >= a.n) {                                        // the "none" row: zeros over the whole pitch
        const int c1 = static_cast<int>(min(static_cast<long long>(c0) + a.chunk_cols, a.ld));
        OutT *orow = outm + static_cast<long long>(a.zrow) * a.ld;
        for (int j = c0 + 4 * tid; j < c1; j += 1024) store4(orow + j, 0u, 0u, 0u, 0u, a.unit_out);
        return;
    }
    const int c1 = min(c0 + a.chunk_cols, a.width);
    if (c0 >= c1) return;                                  // (a chunk that only the wider "none" row has)
    const int A = a.srcA[i], B = a.srcB[i];
    const bool new_i = a.ord[i] < 0;
    const int none = a.n_prev;
    const SrcRows r = src_rows(a, A, B);
    const int si = a.slot ? a.slot[i] : i;                 // row / column of member i in the matrix written
    const bool diag_here = new_i && si >= c0 && si < c1;
    const unsigned fm_i = (diag_here && A != none && B != none) ? a.fm_in[i] : 0u;
    EntryCache<256, kBatchDense> ec;
    load_first(a, r, tid, ec);                             // (in flight while the accumulators are cleared)
    for (int j = 4 * tid; j < c1 - c0; j += 1024) *reinterpret_cast<uint4 *>(acc + j) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const unsigned wi = new_i ? 1u : 2u;
    for_each_entry(a, r, tid, ec, [&](unsigned cu, unsigned m) {
        if (new_i && cu == static_cast<unsigned>(i)) return;
        const int c = a.slot ? a.slot[cu] : static_cast<int>(cu);
        if (c < c0 || c >= c1) return;
        atomicAdd(&acc[c - c0], m * wi);
    });
    __syncthreads();
    if (diag_here && tid == 0) acc[si - c0] = a.half_out + 2u * fm_i;
    __syncthreads();
    OutT *orow = outm + static_cast<long long>(si) * a.ld + c0;
    for (int j = 4 * tid; j < c1 - c0; j += 1024) {
        const uint4 u = *reinterpret_cast<const uint4 *>(acc + j);
        store4(orow + j, u.x, u.y, u.z, u.w, a.unit_out);
    }
}

inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

struct SparseLevels {
    int S = 0;                       // eligible steps 0..S-1
    int k = -1;                      // last sparse cut after calibration
    bool calibrated = false;
    SparseTuning tun;
    std::vector<int> n_of;           // members of cuts 0..S
    std::vector<SparseStepDev> dev;
    char *blob = nullptr;            // children and mate lists of the S steps
    std::vector<const int *> ch_off, mt_off;
    std::vector<const unsigned *> ch;
    std::vector<const uint2 *> mt;
    std::vector<int> n_ch;           // entries of ch per step
    std::vector<const int *> slot;   // per step that stays in place: slot of every member of the cut it writes (else nullptr)
    std::vector<int> zrow;           // per step: the all-zero "none" row of the matrix it writes
    std::vector<double> dense_ms;    // per step: estimated time of the step as the dense plan would run it (row kernels, block assembly, in place)
    std::vector<double> out_bytes;   // per step: bytes of the dense matrix the sparse -> dense step would write in its place
    uint2 *ent[2] = {nullptr, nullptr};
    uint2 *rowd_blob = nullptr;      // row descriptors of cuts 0..S-1, written by the calibration run and kept
    std::vector<uint2 *> rowd;
    unsigned *fm_blob = nullptr;     // fm[c], c = 0..S-1: one word per member of cut c+1
    std::vector<unsigned *> fm;
    // rows of a cut by length (calibration): rnz[c][i] = non-zero entries of row i of Psi_c; order[c] = the members of cut c, longest row
    // first; cls[c][j] = first position in order[c] of the rows of class j (0: more than 1024 entries, 1: more than 256, 2: the rest; [3] = n)
    unsigned *rnz_blob = nullptr;
    int *order_blob = nullptr;
    std::vector<unsigned *> rnz;
    std::vector<int *> order;
    std::vector<std::array<int, 4>> cls;
    size_t ent_cap[2] = {0, 0};      // entries each arena holds: small at first, enlarged by the calibration run where a cut needs it
    size_t ent_max = 0;              // ... up to this many (a cut that needs more is too dense to stay sparse)
    int n_grown = 0;                 // (trace) how often the calibration run enlarged an arena
    bool lists_fresh = false;        // the calibration run has just written the lists of cut k (its fill launches ARE the list steps of a sweep) and
                                     // nothing has touched them since: the first sweep starts at the step that turns them into a matrix
    unsigned *stat = nullptr;        // 4 words per cut
    unsigned *stat_host = nullptr;   // pinned copy of them, fetched at the end of a sweep
    std::vector<long long> nnz;      // non-zero entries of Psi_c per cut 0..S (-1 unknown)
    std::vector<long long> n_ent;    // entries of Y_c
    std::vector<int> max_row;
    int cap_cal = 0;
    double bytes = 0.0;
};

int sparse_eligible_steps(const Plan &plan)
{
    const int n_steps = plan.n_levels - 1;
    int S = 0;
    for (int s = 0; s + 1 < n_steps; ++s) {                // (the proband step keeps its row kernel: proband order, row shards)
        const LevelStep &st = plan.steps[s];
        if (s + 1 > kSparseMaxLevel) break;
        // (any kernel family of the dense plan: a list step needs the sources of the members only, and the step that writes the first
        // dense matrix writes it the way the plan stores that cut -- compactly, or by slot when the step stays in place)
        if (st.n >= kSparseMaxMembers || st.n_prev >= kSparseMaxMembers || st.n < 1 || st.n_prev < 1) break;
        if (st.stay && st.P >= kSparseMaxMembers) break;
        ++S;
    }
    return S;
}

static int wp_for(int n, int nt)
{
    const int W = (n + 31) / 32;
    const int T = ((W + nt - 1) / nt) | 1;
    return nt * T;
}

#define SP_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return GENPHI_ERR_DEVICE; } \
    } while (0)

SparseLevels *sparse_levels_create(const Plan &plan, int S, const std::vector<SparseStepDev> &dev, const SparseTuning &tun,
                                   hipStream_t stream, std::string &err)
{
    if (S < 2 || static_cast<int>(dev.size()) < S) { err = "sparse_levels_create: nothing eligible"; return nullptr; }
    PhaseTrace trace;
    SparseLevels *sl = new (std::nothrow) SparseLevels();
    if (!sl) { err = "out of memory"; return nullptr; }
    sl->S = S; sl->tun = tun; sl->dev.assign(dev.begin(), dev.begin() + S);
    sl->n_of.resize(S + 1);
    for (int c = 0; c <= S; ++c) sl->n_of[c] = static_cast<int>(plan.cut_sizes[c]);
    sl->nnz.assign(S + 1, -1); sl->n_ent.assign(S + 1, -1); sl->max_row.assign(S + 1, 0);
    sl->nnz[0] = sl->n_of[0]; sl->max_row[0] = 1;
    // children lists (counting sort of the members of cut s+1 by their sources) and mate lists (new members with two parents, by
    // their A source)
    size_t total = 256;
    std::vector<std::vector<int>> offs(S), moffs(S);
    std::vector<std::vector<unsigned>> chs(S);
    std::vector<std::vector<uint2>> mts(S);
    for (int s = 0; s < S; ++s) {
        const LevelStep &st = plan.steps[s];
        const int n_prev = static_cast<int>(st.n_prev), n = static_cast<int>(st.n);
        std::vector<int> &off = offs[s], &moff = moffs[s];
        off.assign(static_cast<size_t>(n_prev) + 2, 0);
        moff.assign(static_cast<size_t>(n_prev) + 2, 0);
        for (int i = 0; i < n; ++i) {
            if (st.srcA[i] != n_prev) off[st.srcA[i] + 1]++;
            if (st.srcB[i] != n_prev) off[st.srcB[i] + 1]++;
            if (st.ord[i] < 0 && st.srcA[i] != n_prev && st.srcB[i] != n_prev) moff[st.srcA[i] + 1]++;
        }
        for (int q = 0; q <= n_prev; ++q) { off[q + 1] += off[q]; moff[q + 1] += moff[q]; }
        std::vector<unsigned> &ch = chs[s];
        std::vector<uint2> &mt = mts[s];
        ch.resize(static_cast<size_t>(off[n_prev]) + 1);
        mt.resize(static_cast<size_t>(moff[n_prev]) + 1);
        std::vector<int> fill(off.begin(), off.end() - 1), mfill(moff.begin(), moff.end() - 1);
        for (int i = 0; i < n; ++i) {
            const unsigned w = static_cast<unsigned>(i) | (st.ord[i] < 0 ? 0u : 0x80000000u);
            if (st.srcA[i] != n_prev) ch[fill[st.srcA[i]]++] = w;
            if (st.srcB[i] != n_prev) ch[fill[st.srcB[i]]++] = w;
            if (st.ord[i] < 0 && st.srcA[i] != n_prev && st.srcB[i] != n_prev)
                mt[mfill[st.srcA[i]]++] = make_uint2(static_cast<unsigned>(st.srcB[i]), static_cast<unsigned>(i));
        }
        sl->n_ch.push_back(off[n_prev]);
        total += al256(off.size() * sizeof(int)) + al256(ch.size() * sizeof(unsigned)) + al256(moff.size() * sizeof(int)) + al256(mt.size() * sizeof(uint2));
        if (st.stay) total += al256(st.out_slots.size() * sizeof(int));
    }
    sl->n_ent[0] = sl->n_ch[0];
    trace.mark("  sparse: children, mates (host)");
    auto fail = [&](const std::string &m) { err = m; sparse_levels_destroy(sl); return static_cast<SparseLevels *>(nullptr); };
    if (cached_malloc(reinterpret_cast<void **>(&sl->blob), total) != hipSuccess) return fail("hipMalloc (children lists) failed");
    std::vector<char> host(total, 0);
    size_t o = 0;
    sl->ch_off.resize(S); sl->ch.resize(S); sl->mt_off.resize(S); sl->mt.resize(S); sl->slot.assign(S, nullptr); sl->zrow.assign(S, 0);
    auto put = [&](const void *src, size_t bytes) -> const char * {
        std::memcpy(host.data() + o, src, bytes);
        const char *d = sl->blob + o;
        o += al256(bytes);
        return d;
    };
    for (int s = 0; s < S; ++s) {
        sl->ch_off[s] = reinterpret_cast<const int *>(put(offs[s].data(), offs[s].size() * sizeof(int)));
        sl->ch[s] = reinterpret_cast<const unsigned *>(put(chs[s].data(), chs[s].size() * sizeof(unsigned)));
        sl->mt_off[s] = reinterpret_cast<const int *>(put(moffs[s].data(), moffs[s].size() * sizeof(int)));
        sl->mt[s] = reinterpret_cast<const uint2 *>(put(mts[s].data(), mts[s].size() * sizeof(uint2)));
        const LevelStep &st = plan.steps[s];
        if (st.stay) sl->slot[s] = reinterpret_cast<const int *>(put(st.out_slots.data(), st.out_slots.size() * sizeof(int)));
        // (a cut stored by slot -- written by a step that stays in place, or the compactly written entry cut of such a run -- has its
        // "none" row at the slot capacity P = its pitch)
        sl->zrow[s] = (st.stay || (s + 1 < static_cast<int>(plan.steps.size()) && plan.steps[s + 1].src_slots)) ? static_cast<int>(plan.ld[s + 1]) : static_cast<int>(st.n);
        // what the step costs as the dense plan runs it, in matrix entries moved (the planner's own cost model, planner.cpp): a row-kernel
        // level reads and writes whole matrices; block assembly moves the new members' blocks and, unless it stays in place, the dragged block
        const double n = static_cast<double>(st.n), np = static_cast<double>(st.n_prev), d = static_cast<double>(st.n_dragged), nu = n - d;
        const double q = static_cast<double>(st.parents.size());
        const double nn = 2.0 * q * q + 1.5 * nu * q + nu * nu;
        double entries = np * np + n * n, fixed = 0.008;
        if (st.mode == kModeWide) {
            entries = st.stay ? nn + nu * nu + 3.5 * nu * d : nn + (d + 2.0 * nu) * np + n * d + 2.0 * nu * d;
            fixed = 0.045;
        }
        sl->dense_ms.push_back(4.0 * entries / 4.6e9 + fixed);
        sl->out_bytes.push_back(4.0 * n * static_cast<double>(st.width));
    }
    if (hipMemcpyAsync(sl->blob, host.data(), total, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return fail("upload of the children lists failed");
    trace.mark("  sparse: lists to device");
    // arenas: the rows of Y of cuts 0..S-1 (a non-zero entry of Psi has about two children; the calibration also writes the cut
    // that turns out too dense, until the arena is full)
    int n_max = 0;
    for (int c = 0; c <= S; ++c) n_max = std::max(n_max, sl->n_of[c]);
    const double share = tun.force_k >= 0 ? 1.0 : std::min(1.0, std::max(1, tun.max_permille) / 1000.0);
    double want = 3.0 * share * static_cast<double>(n_max) * static_cast<double>(n_max) + 8.0 * n_max + 1024.0;
    want = std::min(want, 4.0e9);                      // (32-bit cursor)
    // `want` is the most an arena may grow to, not what it starts with: the lists of the leading cuts are tiny (genea140: 0.2 k ... 11 M
    // entries where `want` is 110 M), and a block allocated for the worst case has to be given back after the calibration run -- the
    // driver clears released VRAM with the copy engines, and for ~65 ms per GB released every device-to-host copy of the process runs
    // at HALF its rate (profiles/microbench/free_then_copy.hip: 256 MB in 8.9 instead of 4.7 ms; a one-shot gen.phi with a 400 MB
    // result right after genea140: 20.5 instead of 8.5 ms for the copy).
    sl->ent_max = static_cast<size_t>(want);
    const size_t first = std::min(sl->ent_max, std::max<size_t>(static_cast<size_t>(std::max(64, tun.first_entries)), static_cast<size_t>(sl->n_ch[0]) + 64));
    for (int b = 0; b < 2; ++b) {
        sl->ent_cap[b] = first;
        if (cached_malloc(reinterpret_cast<void **>(&sl->ent[b]), sl->ent_cap[b] * sizeof(uint2)) != hipSuccess) return fail("hipMalloc (row-list arena) failed");
    }
    size_t rowd_total = 0, fm_total = 0;
    auto pad32 = [](int n) { return (static_cast<size_t>(n) + 32) / 32 * 32; };
    for (int c = 0; c < S; ++c) { rowd_total += pad32(sl->n_of[c]); fm_total += pad32(sl->n_of[c + 1]); }
    if (cached_malloc(reinterpret_cast<void **>(&sl->rowd_blob), rowd_total * sizeof(uint2)) != hipSuccess) return fail("hipMalloc (row descriptors) failed");
    if (cached_malloc(reinterpret_cast<void **>(&sl->fm_blob), fm_total * sizeof(unsigned)) != hipSuccess) return fail("hipMalloc (parents' kinships) failed");
    if (cached_malloc(reinterpret_cast<void **>(&sl->rnz_blob), rowd_total * sizeof(unsigned)) != hipSuccess) return fail("hipMalloc (row lengths) failed");
    if (cached_malloc(reinterpret_cast<void **>(&sl->order_blob), rowd_total * sizeof(int)) != hipSuccess) return fail("hipMalloc (row order) failed");
    sl->rowd.resize(S); sl->fm.resize(S); sl->rnz.resize(S); sl->order.resize(S); sl->cls.assign(S, std::array<int, 4>{0, 0, 0, 0});
    for (size_t c = 0, at = 0, fat = 0; c < static_cast<size_t>(S); ++c) {
        sl->rowd[c] = sl->rowd_blob + at; sl->rnz[c] = sl->rnz_blob + at; sl->order[c] = sl->order_blob + at; at += pad32(sl->n_of[c]);
        sl->fm[c] = sl->fm_blob + fat; fat += pad32(sl->n_of[c + 1]);
    }
    if (cached_malloc(reinterpret_cast<void **>(&sl->stat), 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned)) != hipSuccess) return fail("hipMalloc (counters) failed");
    if (cached_pinned(reinterpret_cast<void **>(&sl->stat_host), 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned)) != hipSuccess)
        return fail("hipHostMalloc (counters) failed");
    std::memset(sl->stat_host, 0, 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned));
    sl->bytes = static_cast<double>(total) + static_cast<double>((sl->ent_cap[0] + sl->ent_cap[1]) * sizeof(uint2)) +
                static_cast<double>(rowd_total * sizeof(uint2) + fm_total * sizeof(unsigned));
    sl->cap_cal = std::min(8192, (n_max + 63) / 64 * 64);
    trace.mark("  sparse: arenas");
    if (static_cast<size_t>(sl->n_ch[0]) > sl->ent_cap[0]) return fail("sparse levels: arena smaller than the first cut");
    return sl;
}

void sparse_levels_destroy(SparseLevels *sl)
{
    if (!sl) return;
    if (sl->blob) (void)cached_free(sl->blob);
    for (int b = 0; b < 2; ++b) if (sl->ent[b]) (void)cached_free(sl->ent[b]);
    if (sl->rowd_blob) (void)cached_free(sl->rowd_blob);
    if (sl->fm_blob) (void)cached_free(sl->fm_blob);
    if (sl->rnz_blob) (void)cached_free(sl->rnz_blob);
    if (sl->order_blob) (void)cached_free(sl->order_blob);
    if (sl->stat) (void)cached_free(sl->stat);
    cached_pinned_release(sl->stat_host);
    delete sl;
}

// arguments of step s (source cut s as rows of Y_s in arena s & 1)
static SpArgs args_for(const SparseLevels *sl, int s)
{
    SpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ent_in = sl->ent[s & 1]; a.rowd_in = sl->rowd[s]; a.fm_in = sl->fm[s];
    a.srcA = sl->dev[s].srcA; a.srcB = sl->dev[s].srcB; a.ord = sl->dev[s].ord;
    a.n_prev = sl->n_of[s]; a.n = sl->n_of[s + 1];
    a.half_out = 1u << (2 * s + 2);
    a.unit_out = std::ldexp(1.0f, -(2 * s + 3));
    a.stat = sl->stat + 4 * (s + 1);
    if (s + 1 < sl->S) {                                   // (the row-list form of cut s+1 needs the lists of step s+1)
        a.ent_out = sl->ent[(s + 1) & 1]; a.rowd_out = sl->rowd[s + 1];
        a.chn_off = sl->ch_off[s + 1]; a.chn = sl->ch[s + 1];
        a.mt_off = sl->mt_off[s + 1]; a.mt = sl->mt[s + 1];
        a.fm_out = sl->fm[s + 1];
        a.rnz_out = sl->rnz[s + 1];
    }
    a.count_only = 0;
    a.ent_cap = static_cast<unsigned>(std::min<size_t>(sl->ent_cap[(s + 1) & 1], 0xffffffffu));
    return a;
}

static int launch_identity(SparseLevels *sl, hipStream_t stream, std::string &err)
{
    const int n0 = sl->n_of[0], n1 = sl->n_of[1], n_stat = 4 * (sl->S + 1);
    const int m = std::max(std::max(n0, n1), std::max(n_stat, sl->n_ch[0]));
    hipLaunchKernelGGL(sparse_identity_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, sl->ent[0], sl->rowd[0], sl->ch_off[0], sl->ch[0], n0,
                       sl->n_ch[0], sl->dev[0].srcA, sl->dev[0].srcB, sl->dev[0].ord, sl->fm[0], n1, sl->stat, n_stat);
    SP_TRY(hipGetLastError());
    return GENPHI_OK;
}

// one launch of the row-list step: n_rows members (rows == nullptr: all of them, in order), `cap` entries per row in LDS, four
// wavefronts per row when `wide`
static int launch_rows(SparseLevels *sl, int s, const int *rows, int n_rows, int cap, bool wide, hipStream_t stream, std::string &err,
                       bool count_only = false, bool calibrating = false, unsigned nz_lo = 0u, unsigned nz_hi = 0u)
{
    if (n_rows <= 0) return GENPHI_OK;
    SpArgs a = args_for(sl, s);
    a.count_only = count_only ? 1 : 0;
    a.nz_lo = nz_lo; a.nz_hi = nz_hi;
    a.stop = calibrating ? sl->stat + 3 : nullptr;         // (word 3 of cut 0's counters, cleared by sparse_identity_kernel)
    a.level = static_cast<unsigned>(s + 1);
    a.rows = rows;
    a.remap = (rows != nullptr && rows == sl->dev[s].work) ? 1 : 0;
    a.cap = cap;
    a.wp = wp_for(a.n, wide ? 256 : 64);
    // (the counting launch keeps only the bitmap in LDS)
    const size_t lds = count_only ? 4 * static_cast<size_t>(a.wp) : 6 * (static_cast<size_t>(a.wp) + static_cast<size_t>(a.cap));
    if (wide && sl->tun.long_batch >= 8) hipLaunchKernelGGL((sparse_step_kernel<256, 8>), dim3(static_cast<unsigned>(n_rows)), dim3(256), lds, stream, a);
    else if (wide) hipLaunchKernelGGL((sparse_step_kernel<256, kBatch>), dim3(static_cast<unsigned>(n_rows)), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((sparse_step_kernel<64, kBatch>), dim3(static_cast<unsigned>(n_rows)), dim3(64), lds, stream, a);
    SP_TRY(hipGetLastError());
    return GENPHI_OK;
}

// Rough times (ms) of the forms a level step can take, from the counts of the calibration run: what the choice of the last sparse cut
// rests on.  The rates are measured ones (MI355X, profiles/): a row-list step moves its lists at ~2 TB/s (it is bound by dependent
// round trips, not by bytes), the sparse -> dense step writes at ~5 TB/s and reads two lists per row, a dense level runs at ~0.6 of 8 TB/s.
static double t_list_step(const SparseLevels *sl, int s)
{
    return 8.0 * (2.0 * static_cast<double>(sl->n_ent[s]) + static_cast<double>(sl->n_ent[s + 1])) / 2.0e9 + 0.010;
}
static double t_dense_from_lists(const SparseLevels *sl, int k)
{
    return sl->out_bytes[k] / 5.0e9 + 16.0 * static_cast<double>(sl->n_ent[k]) / 3.5e9 + 0.008;
}
static double t_dense_step(const SparseLevels *sl, int s) { return sl->dense_ms[s]; }

// a plan whose sweep stays dense after all keeps its counts (diagnostics) but not the arenas
static void drop_arenas(SparseLevels *sl, hipStream_t stream)
{
    (void)hipStreamSynchronize(stream);
    for (int b = 0; b < 2; ++b) {
        if (sl->ent[b]) (void)cached_free(sl->ent[b]);
        sl->ent[b] = nullptr;
        sl->bytes -= static_cast<double>(sl->ent_cap[b] * sizeof(uint2));
        sl->ent_cap[b] = 0;
    }
}

int sparse_levels_calibrate(SparseLevels *sl, hipStream_t stream, std::string &err)
{
    if (!sl) { err = "sparse_levels_calibrate: null handle"; return GENPHI_ERR_ARG; }
    if (sl->calibrated) return GENPHI_OK;
    sl->k = -1;
    if (sl->tun.force_k == -1) { sl->calibrated = true; drop_arenas(sl, stream); return GENPHI_OK; }
    const bool trace = std::getenv("GENPHI_TRACE") != nullptr;
    // Every candidate cut in one go, without a word from the host in between: per cut a COUNTING launch (row lengths straight from the
    // bitmaps), a scan that turns lengths into places (sparse_place_kernel), and the launch that writes the lists where they belong --
    // the form every later sweep runs.  (The first version placed rows with an atomic cursor and synchronised after every cut: one
    // same-address device-scope atomic per row is ~26 ns -- 1.9 of the 2.3 ms it took on genea140.)  A cut that turns out too dense
    // voids itself (its rows do not fit the LDS of the writing launch, or its lists the arena): everything behind it sees empty lists.
    PhaseTrace tr;                                         // (GENPHI_TRACE: where the calibration run's time goes)
    int rc = launch_identity(sl, stream, err);
    if (rc) return rc;
    const int n_cand = sl->tun.force_k >= 0 ? std::min(sl->S - 1, sl->tun.force_k) : sl->S - 1;   // cuts 1..n_cand may be kept as lists
    std::vector<unsigned> st(4 * (static_cast<size_t>(sl->S) + 1), 0u), rnz_all;
    // (host images laid out like the device blobs -- cut c at the offset of sl->rnz[c] / sl->order[c] -- so that a leg's row counts
    // come back in ONE copy and the row orders go out in one: ten small pageable copies were 0.15 ms of genea140's run)
    std::vector<size_t> rnz_at(sl->S + 1, 0);
    for (int c = 0; c < sl->S; ++c) rnz_at[c] = static_cast<size_t>(sl->rnz[c] - sl->rnz_blob);
    const size_t blob_words = sl->S > 0 ? rnz_at[sl->S - 1] + static_cast<size_t>(sl->n_of[sl->S - 1]) : 1;
    rnz_all.resize(std::max<size_t>(blob_words, 1));
    int last = 0;                                          // last cut whose lists are valid and sparse enough
    std::vector<int> order, cnt;
    std::vector<std::vector<int>> orders(sl->S + 1);
    // The arenas start small (sparse_levels_create).  A cut whose lists do not fit voids itself and names the entries it needs; the host
    // then enlarges the arena that cut is written to (the other one holds its source), gives the other one room for the cut behind it
    // (moving the live lists), and the run resumes AT that cut: `from` = the first cut of the current leg.
    for (int from = 1; from <= n_cand;) {
        for (int s = from - 1; s < n_cand; ++s) {          // cut s+1 may be kept as lists only when step s+1 is eligible too
            const int n = sl->n_of[s + 1];
            const bool wide = sl->n_of[s] >= 4096;         // (list lengths are not known on the host yet)
            rc = launch_rows(sl, s, nullptr, n, sl->cap_cal, wide, stream, err, /*count_only=*/true, /*calibrating=*/true);
            if (rc) return rc;
            const unsigned long long stop_entries = sl->tun.force_k >= 0 ? ~0ull : static_cast<unsigned long long>(0.25 * static_cast<double>(n) * static_cast<double>(n));
            hipLaunchKernelGGL(sparse_place_kernel, dim3(1), dim3(1024), 0, stream, sl->rowd[s + 1], n,
                               static_cast<unsigned>(std::min<size_t>(sl->ent_cap[(s + 1) & 1], 0xffffffffu)), stop_entries, sl->stat + 4 * (s + 1), sl->stat + 3,
                               static_cast<unsigned>(s + 1));
            SP_TRY(hipGetLastError());
            // the writing launches by class of row lengths, as in a sweep (the counting launch has left every row's count on the device;
            // the host does not know them yet, so every launch is offered every row and a row picks its launch): short rows on one
            // wavefront with a small row buffer -- one launch with room for the longest row a cut may have ran at a third of a sweep's
            // occupancy (genea140: the run 1.2 -> ... ms of a one-shot call)
            const int cap = sl->cap_cal;
            rc = launch_rows(sl, s, nullptr, n, std::min(cap, 256), false, stream, err, false, true, 0u, 256u);
            if (rc == GENPHI_OK && cap > 256) rc = launch_rows(sl, s, nullptr, n, std::min(cap, 1024), true, stream, err, false, true, 256u, 1024u);
            if (rc == GENPHI_OK && cap > 1024) rc = launch_rows(sl, s, nullptr, n, cap, true, stream, err, false, true, 1024u, 0xffffffffu);
            if (rc) return rc;
        }
        // one round trip per leg: the counters of every cut and every row's number of non-zero entries
        SP_TRY(hipMemcpyAsync(st.data(), sl->stat, st.size() * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        if (from <= n_cand)
            SP_TRY(hipMemcpyAsync(rnz_all.data() + rnz_at[from], sl->rnz[from],
                                  (rnz_at[n_cand] + static_cast<size_t>(sl->n_of[n_cand]) - rnz_at[from]) * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        tr.mark("    calibration: a leg enqueued");
        SP_TRY(hipStreamSynchronize(stream));
        tr.mark("    calibration: a leg done on the GPU");
        int grow_at = 0;
        for (int s = from - 1; s < n_cand; ++s) {
            const int n = sl->n_of[s + 1];
            const unsigned flags = st[4 * (s + 1) + 2];
            if (flags == 8u && static_cast<size_t>(st[4 * (s + 1) + 1]) + 64 <= sl->ent_max) { grow_at = s + 1; break; }   // only the arena was too small
            if (flags != 0u) break;                        // a row overflowed, or a cut before it was too dense: no lists
            const unsigned *rnz = rnz_all.data() + rnz_at[s + 1];
            long long nnz = 0;
            unsigned longest = 0;
            for (int q = 0; q < n; ++q) { nnz += rnz[q]; longest = std::max(longest, rnz[q]); }
            sl->nnz[s + 1] = nnz;
            sl->n_ent[s + 1] = static_cast<long long>(st[4 * (s + 1) + 0]);
            sl->max_row[s + 1] = static_cast<int>(longest);
            // the rows of cut s+1 by length, longest first (a counting sort): a launch per class of lengths, each with the LDS its rows need
            order.resize(n);
            cnt.assign(static_cast<size_t>(longest) + 2, 0);
            for (int q = 0; q < n; ++q) cnt[longest - rnz[q] + 1]++;
            for (unsigned v = 0; v <= longest; ++v) cnt[v + 1] += cnt[v];
            for (int q = 0; q < n; ++q) order[cnt[longest - rnz[q]]++] = q;
            std::array<int, 4> &cl = sl->cls[s + 1];
            cl = {0, 0, 0, n};
            for (int q = 0; q < n; ++q) { if (rnz[order[q]] > 1024u) cl[1] = q + 1; else if (rnz[order[q]] > 256u) cl[2] = q + 1; else break; }
            cl[2] = std::max(cl[2], cl[1]);
            orders[s + 1] = order;
            const double dn = static_cast<double>(n);
            if (trace)
                std::fprintf(stderr, "[genphi trace]   sparse cut %2d: %6d members, %10lld non-zero (%.4f), %10lld list entries, longest row %5d, rows > 1024 / > 256: %d / %d; "
                             "est. list step %.3f ms, dense step %.3f ms\n", s + 1, n, sl->nnz[s + 1], static_cast<double>(sl->nnz[s + 1]) / (dn * dn), sl->n_ent[s + 1],
                             sl->max_row[s + 1], cl[1], cl[2], t_list_step(sl, s), t_dense_step(sl, s));
            if (sl->tun.force_k < 0 && static_cast<double>(nnz) > sl->tun.max_permille / 1000.0 * dn * dn) break;
            last = s + 1;
        }
        if (grow_at == 0) break;
        {
            const int c = grow_at, b = c & 1;
            const size_t need = static_cast<size_t>(st[4 * c + 1]);
            // Large lists only where they can pay: when the estimated times already say that cut c as lists (a list step into it and the dense
            // matrix of cut c+1 from it) loses against stopping at cut c-1, the run ends here instead of allocating for it (cfg4's cut 6: 2.4 GB).
            if (sl->tun.force_k < 0 && c >= 2 && need * sizeof(uint2) > (size_t(256) << 20)) {
                const long long keep = sl->n_ent[c];
                sl->n_ent[c] = static_cast<long long>(need);
                const bool pays = t_list_step(sl, c - 1) + t_dense_from_lists(sl, c) < t_dense_from_lists(sl, c - 1) + t_dense_step(sl, c);
                sl->n_ent[c] = keep;
                if (trace) std::fprintf(stderr, "[genphi trace]   sparse cut %2d needs %zu list entries: %s\n", c, need, pays ? "arena enlarged" : "not worth its lists, the run ends");
                if (!pays) break;
            }
            const size_t cap_b = std::min(sl->ent_max, need + need / 8 + 64);
            uint2 *larger = nullptr;
            if (cached_malloc(reinterpret_cast<void **>(&larger), cap_b * sizeof(uint2)) != hipSuccess) {
                (void)hipGetLastError();                   // (no memory for this cut's lists: the run ends at the cut before it, the sweep goes on densely from there)
                break;
            }
            (void)cached_free(sl->ent[b]);                 // (holds cut c-2: dead)
            sl->ent[b] = larger;
            sl->bytes += static_cast<double>(cap_b * sizeof(uint2)) - static_cast<double>(sl->ent_cap[b] * sizeof(uint2));
            sl->ent_cap[b] = cap_b;
            if (c < n_cand) {                              // room for the cut behind it in the other arena (lists grow up to ~4 x per cut, less and less)
                const size_t dense_next = static_cast<size_t>(sl->n_of[c + 1]) * static_cast<size_t>(c + 2 <= sl->S ? sl->n_of[c + 2] : sl->n_of[c + 1]) + 64;
                const size_t cap_o = std::min(std::min(sl->ent_max, dense_next), 3 * need);
                // (small lists only: a cut behind this one that needs more than 256 MB is looked at -- does it pay? -- before anything is allocated for it)
                if (cap_o > sl->ent_cap[b ^ 1] && cap_o * sizeof(uint2) <= (size_t(256) << 20)) {
                    uint2 *bigger = nullptr;
                    if (cached_malloc(reinterpret_cast<void **>(&bigger), cap_o * sizeof(uint2)) != hipSuccess) { (void)hipGetLastError(); bigger = nullptr; }
                    const size_t live = static_cast<size_t>(sl->n_ent[c - 1]);       // cut c-1: the source of the leg to come
                    if (bigger) {                            // (room for the next cut is a convenience: without it that cut asks for itself)
                        SP_TRY(hipMemcpyAsync(bigger, sl->ent[b ^ 1], live * sizeof(uint2), hipMemcpyDeviceToDevice, stream));
                        SP_TRY(hipStreamSynchronize(stream));
                        (void)cached_free(sl->ent[b ^ 1]);
                        sl->ent[b ^ 1] = bigger;
                        sl->bytes += static_cast<double>(cap_o * sizeof(uint2)) - static_cast<double>(sl->ent_cap[b ^ 1] * sizeof(uint2));
                        sl->ent_cap[b ^ 1] = cap_o;
                    }
                }
            }
            ++sl->n_grown;
            // the counters of cuts c.. and the stop word start over
            SP_TRY(hipMemsetAsync(sl->stat + 4 * c, 0, 4 * static_cast<size_t>(sl->S + 1 - c) * sizeof(unsigned), stream));
            SP_TRY(hipMemsetAsync(sl->stat + 3, 0, sizeof(unsigned), stream));
            from = c;
        }
    }
    if (trace && sl->n_grown) std::fprintf(stderr, "[genphi trace]   sparse: arenas enlarged %d times (%zu + %zu entries)\n", sl->n_grown, sl->ent_cap[0], sl->ent_cap[1]);
    tr.mark("    calibration: rows sorted by length (host)");
    if (last >= 1) {
        std::vector<int> ord_all(rnz_at[last] + static_cast<size_t>(sl->n_of[last]) - rnz_at[1], 0);
        for (int c = 1; c <= last; ++c) std::copy(orders[c].begin(), orders[c].end(), ord_all.begin() + static_cast<std::ptrdiff_t>(rnz_at[c] - rnz_at[1]));
        SP_TRY(hipMemcpyAsync(sl->order[1], ord_all.data(), ord_all.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        SP_TRY(hipStreamSynchronize(stream));              // (`ord_all` goes out of scope)
    }
    tr.mark("    calibration: row orders to the device");
    sl->calibrated = true;                                 // (from here on rows go where this run put them)
    if (last < 1) { drop_arenas(sl, stream); return GENPHI_OK; }
    int k = last;
    if (sl->tun.force_k < 0) {
        // the last sparse cut: the cheapest of "lists up to cut k, the dense matrix of cut k+1 from them, dense beyond"
        double best = 0.0;
        for (int s = 0; s <= last; ++s) best += t_dense_step(sl, s);
        k = -1;
        double lists = 0.0;                                // steps 0..k-1 as list steps
        for (int kk = 1; kk <= last; ++kk) {
            lists += t_list_step(sl, kk - 1);
            double t = lists + t_dense_from_lists(sl, kk);
            for (int s = kk + 1; s <= last; ++s) t += t_dense_step(sl, s);
            if (t < best) { best = t; k = kk; }
        }
        if (k < 1) { drop_arenas(sl, stream); return GENPHI_OK; }
        int widest = 0;
        for (int c = 0; c <= k + 1; ++c) widest = std::max(widest, sl->n_of[c]);
        if (widest < sl->tun.min_cut) { drop_arenas(sl, stream); return GENPHI_OK; }
    }
    sl->k = k;
    {   // the arenas of the calibration run were sized for cuts of unknown density: now each holds exactly the largest cut it is used for
        size_t need[2] = {1024, 1024};
        for (int c = 0; c <= k; ++c) need[c & 1] = std::max(need[c & 1], static_cast<size_t>(sl->n_ent[c]) + 64);
        SP_TRY(hipStreamSynchronize(stream));
        // (cut k's lists survive the run only if no later cut was written over them: cut k+2 shares their arena)
        sl->lists_fresh = true;
        for (int c = k + 2; c <= n_cand; c += 2)
            if (st[4 * c + 2] == 0u && st[4 * c + 0] != 0u) sl->lists_fresh = false;
        for (int b = 0; b < 2; ++b) {
            if (need[b] * 2 > sl->ent_cap[b]) continue;    // (not worth a reallocation)
            if (b == (k & 1)) sl->lists_fresh = false;     // (the lists of cut k go with their arena: the first sweep computes them again)
            (void)cached_free(sl->ent[b]);
            sl->ent[b] = nullptr;
            SP_TRY(cached_malloc(reinterpret_cast<void **>(&sl->ent[b]), need[b] * sizeof(uint2)));
            sl->bytes += static_cast<double>(need[b] * sizeof(uint2)) - static_cast<double>(sl->ent_cap[b] * sizeof(uint2));
            sl->ent_cap[b] = need[b];
        }
    }
    if (trace) std::fprintf(stderr, "[genphi trace]   sparse cuts 0..%d\n", k);
    return GENPHI_OK;
}

int sparse_levels_k(const SparseLevels *sl) { return sl ? sl->k : -1; }

int sparse_levels_enqueue_step(SparseLevels *sl, int s, hipStream_t stream, std::string &err)
{
    if (!sl || s < 0 || s >= sl->k) { err = "sparse_levels_enqueue_step: not a sparse step"; return GENPHI_ERR_ARG; }
    if (sl->lists_fresh) return GENPHI_OK;                 // (first sweep after the calibration run: cut k's lists are there -- genea140 0.3 ms of a one-shot call)
    int rc = s == 0 ? launch_identity(sl, stream, err) : GENPHI_OK;
    if (rc) return rc;
    const std::array<int, 4> &cl = sl->cls[s + 1];
    const int *ord = sl->order[s + 1];
    const int cap_max = std::min(sl->cap_cal, std::max(64, (sl->max_row[s + 1] + 63) / 64 * 64));
    // One launch for all rows when their lengths are alike (random mating: the longest row of a cut is less than twice the average) --
    // several launches cost their tails (cfg4: +12 % on its two largest list steps) -- and a launch per class of lengths when they are
    // not (a real genealogy: the longest row of genea140's cut 8 is 12 x the average; -11 % over its list steps).  classes = 1 / 0 force.
    const bool skewed = static_cast<long long>(sl->max_row[s + 1]) * sl->n_of[s + 1] > 3ll * sl->nnz[s + 1];
    if (sl->tun.classes == 0 || (sl->tun.classes < 0 && !skewed))      // (in the planner's work order: rows that share sources adjacent)
        return launch_rows(sl, s, sl->dev[s].work, sl->n_of[s + 1], cap_max, sl->n_ent[s] > 192ll * sl->n_of[s], stream, err);
    // rows of more than 1024 entries, of more than 256, the rest: each launch with the LDS its rows need (a launch sized for the
    // longest row of a real genealogy leaves room for four wavefronts per CU)
    rc = launch_rows(sl, s, ord + cl[0], cl[1] - cl[0], cap_max, true, stream, err);
    if (rc == GENPHI_OK) rc = launch_rows(sl, s, ord + cl[1], cl[2] - cl[1], std::min(cap_max, 1024), true, stream, err);
    if (rc == GENPHI_OK) rc = launch_rows(sl, s, ord + cl[2], cl[3] - cl[2], std::min(cap_max, 256), false, stream, err);
    return rc;
}

int sparse_levels_enqueue_dense(SparseLevels *sl, void *out, bool f64, bool compact, long long ld, long long width, hipStream_t stream, std::string &err)
{
    if (!sl || sl->k < 1) { err = "sparse_levels_enqueue_dense: no sparse cut"; return GENPHI_ERR_ARG; }
    sl->lists_fresh = false;                               // (from the next sweep on the list steps run)
    SpArgs a = args_for(sl, sl->k);
    a.out = out; a.ld = ld; a.width = static_cast<int>(width);
    // columns per workgroup: the member rows in as few equal chunks as the budget allows (the "none" row spans the pitch)
    const int budget = std::max(1024, std::min(sl->tun.chunk_cols, 15360) / 1024 * 1024);
    const int row_chunks = static_cast<int>((width + budget - 1) / budget);
    a.chunk_cols = static_cast<int>(((width + row_chunks - 1) / row_chunks + 255) / 256 * 256);
    a.n_chunks = static_cast<int>((std::max(ld, width) + a.chunk_cols - 1) / a.chunk_cols);
    const size_t lds = static_cast<size_t>(a.chunk_cols) * sizeof(unsigned);
    a.rows = sl->dev[sl->k].work;
    // (the Float64 sweep and the per-entry sweep store every cut compactly: no slots, the "none" row at n)
    a.slot = compact ? nullptr : sl->slot[sl->k];
    a.zrow = compact ? a.n : sl->zrow[sl->k];
    const dim3 grid(static_cast<unsigned>((a.n + 1 + 7) / 8 * 8) * a.n_chunks);
    if (f64) hipLaunchKernelGGL(sparse_dense_kernel<double>, grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(sparse_dense_kernel<float>, grid, dim3(256), lds, stream, a);
    SP_TRY(hipGetLastError());
    return GENPHI_OK;
}

int sparse_levels_enqueue_flags(SparseLevels *sl, hipStream_t stream, std::string &err)
{
    if (!sl || sl->k < 1) return GENPHI_OK;
    SP_TRY(hipMemcpyAsync(sl->stat_host, sl->stat, 4 * (static_cast<size_t>(sl->k) + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    return GENPHI_OK;
}

bool sparse_levels_flags_ok(const SparseLevels *sl)
{
    if (!sl || sl->k < 1) return true;
    for (int c = 1; c <= sl->k; ++c) if (sl->stat_host[4 * c + 2] != 0u) return false;
    return true;
}

int sparse_levels_counts(const SparseLevels *sl, int cap, long long *nnz, long long *entries, int *max_row)
{
    if (!sl) return 0;
    const int m = std::min(cap, sl->S + 1);
    for (int c = 0; c < m; ++c) {
        if (nnz) nnz[c] = sl->nnz[c];
        if (entries) entries[c] = sl->n_ent[c];
        if (max_row) max_row[c] = sl->max_row[c];
    }
    return m;
}

double sparse_levels_device_bytes(const SparseLevels *sl) { return sl ? sl->bytes : 0.0; }

}  // namespace genphi
