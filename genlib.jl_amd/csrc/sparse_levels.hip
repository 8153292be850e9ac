// sparse_levels.hip -- the zero-aware leading levels of a gen.phi sweep (see sparse_levels.h).
//
// Replaces, for the cuts right below the founders, what src/compute.jl:291-301 does with a fresh dense matrix per level:
// the same entries (src/compute.jl:105-158 evaluated pair by pair) are produced row by row from the non-zero entries of
// the previous cut only.  Row i of cut s+1 is
//     Psi'[i][j] = w_i w_j sum_{p in src(i)} sum_{q in src(j)} Psi[p][q]          (w = 1 dragged, 1/2 new; planner.h)
// i.e. every non-zero (q, v) of the source rows of i contributes w_i w_j v to the columns j that have q as a source -- the
// CHILDREN of q (q itself when it is dragged along, weight 1; its new children, weight 1/2).  The diagonal of a new member
// is 1/2 + Psi[f][m]/2 when both parents exist, 1/2 otherwise (src/compute.jl:148-154).
//
// Layout in HBM.  A sparse cut is two arrays: rowd[i] = (first entry, number of entries) per member and ent[] = (column,
// Float32 bits) pairs, the entries of a row contiguous and ascending by column.  Rows are placed by an atomic cursor, so their
// order in ent[] varies from run to run; their contents do not.  Two arenas alternate between consecutive cuts.
//
// Arithmetic: integer units of 2^-(2c+1) for cut c (exact for c <= 11, see sparse_levels.h), accumulated with LDS atomics;
// Float32 values are rebuilt by one exact conversion.  No rank words are needed: every grouping of the reference's sum is exact.
#include "sparse_levels.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/genphi.h"

namespace genphi {

namespace {

typedef float f4_t __attribute__((ext_vector_type(4)));

struct SpArgs {
    const uint2 *ent_in;          // source cut: (column, Float32 bits) ...
    const uint2 *rowd_in;         // ... and per row (first entry, entries)
    uint2 *ent_out;               // (row-list step) entries of the cut written
    uint2 *rowd_out;              // (row-list step) its row descriptors: written by the calibration run (rows placed by an atomic cursor),
    int fixed;                    //   read by every later sweep (fixed != 0: the row's place and length are the plan's; a length that differs is an error)
    const int *srcA, *srcB, *ord; // per member of the cut written: sources in the source cut (n_prev = none), rank word (< 0: new)
    const int *ch_off;            // children of member q of the source cut: ch[ch_off[q] .. ch_off[q + 1])
    const unsigned *ch;           // position in the cut written | 0x80000000 when the child is q itself (dragged: weight 1)
    int n_prev, n;
    int wp;                       // (row-list step) bitmap words in LDS: workgroup size x an odd number
    int cap;                      // (row-list step) entries of one row the LDS holds
    unsigned ent_cap;             // (row-list step) entries the arena written holds
    float scale_in;               // 2^(2c+1): a value of the source cut c in integer units
    float unit_out;               // 2^-(2c+3): one integer unit of the cut written
    unsigned half_out;            // 1/2 in units of the cut written
    unsigned *stat;               // (row-list step) [0] entries written so far, [1] longest row, [2] 1 = a row or the arena overflowed, 2 = a row's length changed
    // sparse -> dense step
    float *out;
    long long ld;
    int width, chunk_cols, n_chunks;
};

// the lists of Psi_0 = 1/2 I (src/compute.jl:271-274) and the counters of a sweep
__global__ void __launch_bounds__(256) sparse_identity_kernel(uint2 *ent, uint2 *rowd, int n0, unsigned *stat, int n_stat)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n0) {
        ent[k] = make_uint2(static_cast<unsigned>(k), __float_as_uint(0.5f));
        rowd[k] = make_uint2(static_cast<unsigned>(k), 1u);
    }
    if (k < n_stat) stat[k] = k == 0 ? static_cast<unsigned>(n0) : (k == 1 ? 1u : 0u);
}

// The entries of the (<= 2) source rows of an output row as ONE index space [0, lenA + lenB), walked kBatch entries per thread at a
// time with the loads of a batch issued together: entry -> children range -> children are three dependent round trips through
// L2, and a row's time is the number of such chains a thread walks one after the other.
constexpr int kBatch = 4;

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

// f(entry from row A?, q, integer value of the entry, first child, end child) for every entry of the two source rows
template <int NT, class F>
__device__ __forceinline__ void for_each_entry(const SpArgs &a, const SrcRows &r, int tid, F &&f)
{
    for (unsigned e0 = tid; e0 < r.total; e0 += kBatch * NT) {
        uint2 en[kBatch];
        bool ok[kBatch], fromA[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const unsigned e = e0 + b * NT;
            ok[b] = e < r.total;
            const unsigned ee = ok[b] ? e : r.total - 1u;
            fromA[b] = ee < r.lenA;
            en[b] = a.ent_in[fromA[b] ? r.offA + ee : r.offB + (ee - r.lenA)];
        }
        int k0[kBatch], k1[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            k0[b] = a.ch_off[en[b].x];
            k1[b] = a.ch_off[en[b].x + 1u];
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b)
            if (ok[b]) f(fromA[b], static_cast<int>(en[b].x), static_cast<unsigned>(__uint_as_float(en[b].y) * a.scale_in), k0[b], k1[b]);
    }
}

// ---- row lists of cut s -> row lists of cut s+1: one workgroup (a wavefront, or four for long rows) per row ----------
// Pass 1 marks the columns the row touches in an LDS bitmap; a scan over the bitmap gives every column its place in the
// (ascending) row and the row its length; pass 2 adds the contributions into the row's values in LDS (integer units,
// ds_add_u32); the row leaves as 8-byte pairs.  Where the row goes: the calibration run of a plan places rows with an
// atomic cursor and records (place, length) per row; every later sweep writes the row to that place (row lengths depend on
// the pedigree alone) -- 24k same-address device-scope atomics per level cost more than the level itself.
template <int NT>
__global__ void __launch_bounds__(NT) sparse_step_kernel(const SpArgs a)
{
    extern __shared__ unsigned lds_u[];
    unsigned *bm = lds_u;
    unsigned *vals = bm + a.wp;
    unsigned short *pre = reinterpret_cast<unsigned short *>(vals + a.cap);
    unsigned short *cols = pre + a.wp;
    __shared__ unsigned fm_slot, off_slot;
    __shared__ int wsum[NT / 64 + 1];
    const int tid = threadIdx.x;
    const int i = blockIdx.x;
    const int A = a.srcA[i], B = a.srcB[i];
    const bool new_i = a.ord[i] < 0;
    const int none = a.n_prev;
    const SrcRows r = src_rows(a, A, B);
    const uint2 place = a.fixed ? a.rowd_out[i] : make_uint2(0u, 0u);
    for (int w = tid; w < a.wp; w += NT) bm[w] = 0u;
    if (tid == 0) fm_slot = 0u;
    __syncthreads();
    // pass 1: which columns
    for_each_entry<NT>(a, r, tid, [&](bool fromA, int q, unsigned mv, int k0, int k1) {
        if (fromA && q == B) fm_slot = mv;                              // Psi[A][B] for the diagonal (one thread at most)
        for (int k = k0; k < k1; ++k) {
            const unsigned c = a.ch[k] & 0x7fffffffu;
            atomicOr(&bm[c >> 5], 1u << (c & 31u));
        }
    });
    if (new_i && tid == 0) atomicOr(&bm[i >> 5], 1u << (i & 31));
    __syncthreads();
    // every thread owns T consecutive bitmap words (T odd: no bank conflicts between the lanes)
    const int T = a.wp / NT, w0 = tid * T;
    int cnt = 0;
    for (int w = w0; w < w0 + T; ++w) cnt += __popc(bm[w]);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const int s = wsum[w];
        if (w < (tid >> 6)) base += s;
        total += s;
    }
    bool bad = total > a.cap;                              // (workgroup-uniform) the row does not fit: the cut is too dense to stay sparse
    unsigned off = place.x;
    if (!bad) {
        if (a.fixed) {
            bad = static_cast<unsigned>(total) != place.y;
        } else {
            if (tid == 0) off_slot = atomicAdd(&a.stat[0], static_cast<unsigned>(total));
            __syncthreads();
            off = off_slot;
            bad = off + static_cast<unsigned>(total) > a.ent_cap || off + static_cast<unsigned>(total) < off;
        }
    }
    if (bad) {
        if (tid == 0) {
            atomicOr(&a.stat[2], (a.fixed && total <= a.cap) ? 2u : 1u);
            if (!a.fixed) a.rowd_out[i] = make_uint2(0u, 0u);
        }
        return;
    }
    {
        int pos = base + incl - cnt;
        for (int w = w0; w < w0 + T; ++w) {
            pre[w] = static_cast<unsigned short>(pos);
            unsigned bits = bm[w];
            while (bits) {
                const int b = __ffs(bits) - 1;
                cols[pos++] = static_cast<unsigned short>(w * 32 + b);
                bits &= bits - 1u;
            }
        }
    }
    for (int t = tid; t < total; t += NT) vals[t] = 0u;
    __syncthreads();
    // pass 2: the values, in units of 2^-(2c+3)
    const unsigned wi = new_i ? 1u : 2u;
    for_each_entry<NT>(a, r, tid, [&](bool, int, unsigned mv, int k0, int k1) {
        const unsigned m = mv * wi;
        for (int k = k0; k < k1; ++k) {
            const unsigned cw = a.ch[k], c = cw & 0x7fffffffu;
            if (new_i && c == static_cast<unsigned>(i)) continue;       // the diagonal of a new member is not a sum of this kind
            const int pos = pre[c >> 5] + __popc(bm[c >> 5] & ((1u << (c & 31u)) - 1u));
            atomicAdd(&vals[pos], m * ((cw >> 31) + 1u));
        }
    });
    __syncthreads();
    if (new_i && tid == 0) {
        const int pos = pre[i >> 5] + __popc(bm[i >> 5] & ((1u << (i & 31)) - 1u));
        vals[pos] = a.half_out + ((A != none && B != none) ? 2u * fm_slot : 0u);      // 1/2 + Psi[A][B]/2, src/compute.jl:148-154
    }
    __syncthreads();
    for (int t = tid; t < total; t += NT)
        a.ent_out[off + t] = make_uint2(static_cast<unsigned>(cols[t]), __float_as_uint(static_cast<float>(vals[t]) * a.unit_out));
    if (!a.fixed && tid == 0) {
        a.rowd_out[i] = make_uint2(off, static_cast<unsigned>(total));
        atomicMax(&a.stat[1], static_cast<unsigned>(total));
    }
}

// ---- row lists of cut k -> the dense matrix of cut k+1: one workgroup per (row, chunk of columns) ---------------------
// The chunk's entries are accumulated in LDS (integer units) and leave as whole 16-byte stores, zeros included: the step
// writes what a FULL / SPLIT row kernel writes (columns [0, width) of every row, the all-zero row n) and reads only lists.
__global__ void __launch_bounds__(256) sparse_dense_kernel(const SpArgs a)
{
    extern __shared__ unsigned acc[];
    __shared__ unsigned fm_slot;
    const int tid = threadIdx.x;
    const int i = blockIdx.x / a.n_chunks, chunk = blockIdx.x - i * a.n_chunks;
    const int c0 = chunk * a.chunk_cols;
    if (i >= a.n) {                                        // the "none" row: zeros over the whole pitch
        const int c1 = static_cast<int>(min(static_cast<long long>(c0) + a.chunk_cols, a.ld));
        float *orow = a.out + static_cast<long long>(a.n) * a.ld;
        const f4_t z = {0.f, 0.f, 0.f, 0.f};
        for (int j = c0 + 4 * tid; j < c1; j += 1024) __builtin_nontemporal_store(z, reinterpret_cast<f4_t *>(orow + j));
        return;
    }
    const int c1 = min(c0 + a.chunk_cols, a.width);
    if (c0 >= c1) return;                                  // (a chunk that only the wider "none" row has)
    const int A = a.srcA[i], B = a.srcB[i];
    const bool new_i = a.ord[i] < 0;
    const int none = a.n_prev;
    const SrcRows r = src_rows(a, A, B);
    for (int j = 4 * tid; j < c1 - c0; j += 1024) *reinterpret_cast<uint4 *>(acc + j) = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) fm_slot = 0u;
    __syncthreads();
    const unsigned wi = new_i ? 1u : 2u;
    for_each_entry<256>(a, r, tid, [&](bool fromA, int q, unsigned mv, int k0, int k1) {
        if (fromA && q == B) fm_slot = mv;
        const unsigned m = mv * wi;
        for (int k = k0; k < k1; ++k) {
            const unsigned cw = a.ch[k];
            const int c = static_cast<int>(cw & 0x7fffffffu);
            if (c < c0 || c >= c1 || (new_i && c == i)) continue;
            atomicAdd(&acc[c - c0], m * ((cw >> 31) + 1u));
        }
    });
    __syncthreads();
    if (new_i && tid == 0 && i >= c0 && i < c1) acc[i - c0] = a.half_out + ((A != none && B != none) ? 2u * fm_slot : 0u);
    __syncthreads();
    float *orow = a.out + static_cast<long long>(i) * a.ld + c0;
    for (int j = 4 * tid; j < c1 - c0; j += 1024) {
        const uint4 u = *reinterpret_cast<const uint4 *>(acc + j);
        const f4_t v = {static_cast<float>(u.x) * a.unit_out, static_cast<float>(u.y) * a.unit_out, static_cast<float>(u.z) * a.unit_out,
                        static_cast<float>(u.w) * a.unit_out};
        __builtin_nontemporal_store(v, reinterpret_cast<f4_t *>(orow + j));
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
    char *blob = nullptr;            // children lists of the S steps
    std::vector<const int *> ch_off;
    std::vector<const unsigned *> ch;
    uint2 *ent[2] = {nullptr, nullptr};
    uint2 *rowd_blob = nullptr;      // row descriptors of cuts 0..S-1, written by the calibration run and kept
    std::vector<uint2 *> rowd;
    size_t ent_cap = 0;
    unsigned *stat = nullptr;        // 4 words per cut
    unsigned *stat_host = nullptr;   // pinned copy of them, fetched at the end of a sweep
    std::vector<long long> nnz;      // per cut 0..S (-1 unknown)
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
        if (st.mode == kModeWide || st.stay || st.src_slots) break;
        if (st.n >= kSparseMaxMembers || st.n_prev >= kSparseMaxMembers || st.n < 1 || st.n_prev < 1) break;
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
    SparseLevels *sl = new (std::nothrow) SparseLevels();
    if (!sl) { err = "out of memory"; return nullptr; }
    sl->S = S; sl->tun = tun; sl->dev.assign(dev.begin(), dev.begin() + S);
    sl->n_of.resize(S + 1);
    for (int c = 0; c <= S; ++c) sl->n_of[c] = static_cast<int>(plan.cut_sizes[c]);
    sl->nnz.assign(S + 1, -1); sl->max_row.assign(S + 1, 0);
    sl->nnz[0] = sl->n_of[0]; sl->max_row[0] = 1;
    // children lists (counting sort of the members of cut s+1 by their sources)
    size_t total = 256;
    std::vector<std::vector<int>> offs(S);
    std::vector<std::vector<unsigned>> chs(S);
    for (int s = 0; s < S; ++s) {
        const LevelStep &st = plan.steps[s];
        const int n_prev = static_cast<int>(st.n_prev), n = static_cast<int>(st.n);
        std::vector<int> &off = offs[s];
        off.assign(static_cast<size_t>(n_prev) + 2, 0);
        for (int i = 0; i < n; ++i) {
            if (st.srcA[i] != n_prev) off[st.srcA[i] + 1]++;
            if (st.srcB[i] != n_prev) off[st.srcB[i] + 1]++;
        }
        for (int q = 0; q <= n_prev; ++q) off[q + 1] += off[q];
        std::vector<unsigned> &ch = chs[s];
        ch.resize(static_cast<size_t>(off[n_prev]) + 1);
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int i = 0; i < n; ++i) {
            const unsigned w = static_cast<unsigned>(i) | (st.ord[i] < 0 ? 0u : 0x80000000u);
            if (st.srcA[i] != n_prev) ch[fill[st.srcA[i]]++] = w;
            if (st.srcB[i] != n_prev) ch[fill[st.srcB[i]]++] = w;
        }
        total += al256(off.size() * sizeof(int)) + al256(ch.size() * sizeof(unsigned));
    }
    auto fail = [&](const std::string &m) { err = m; sparse_levels_destroy(sl); return static_cast<SparseLevels *>(nullptr); };
    if (hipMalloc(reinterpret_cast<void **>(&sl->blob), total) != hipSuccess) return fail("hipMalloc (children lists) failed");
    std::vector<char> host(total, 0);
    size_t o = 0;
    sl->ch_off.resize(S); sl->ch.resize(S);
    for (int s = 0; s < S; ++s) {
        std::memcpy(host.data() + o, offs[s].data(), offs[s].size() * sizeof(int));
        sl->ch_off[s] = reinterpret_cast<const int *>(sl->blob + o);
        o += al256(offs[s].size() * sizeof(int));
        std::memcpy(host.data() + o, chs[s].data(), chs[s].size() * sizeof(unsigned));
        sl->ch[s] = reinterpret_cast<const unsigned *>(sl->blob + o);
        o += al256(chs[s].size() * sizeof(unsigned));
    }
    if (hipMemcpyAsync(sl->blob, host.data(), total, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return fail("upload of the children lists failed");
    // arenas: cuts 1..S-1 may be kept sparse; the calibration also writes the cut that turns out too dense (until the arena is full)
    int n_max = 0;
    for (int c = 0; c < S; ++c) n_max = std::max(n_max, sl->n_of[c]);
    const double share = tun.force_k >= 0 ? 1.0 : std::min(1.0, std::max(1, tun.max_permille) / 1000.0);
    double want = share * static_cast<double>(n_max) * static_cast<double>(n_max) + 2.0 * n_max + 1024.0;
    want = std::min(want, 4.0e9);                      // (32-bit cursor)
    sl->ent_cap = static_cast<size_t>(want);
    for (int b = 0; b < 2; ++b)
        if (hipMalloc(reinterpret_cast<void **>(&sl->ent[b]), sl->ent_cap * sizeof(uint2)) != hipSuccess) return fail("hipMalloc (row-list arena) failed");
    size_t rowd_total = 0;
    for (int c = 0; c < S; ++c) rowd_total += (static_cast<size_t>(sl->n_of[c]) + 32) / 32 * 32;
    if (hipMalloc(reinterpret_cast<void **>(&sl->rowd_blob), rowd_total * sizeof(uint2)) != hipSuccess) return fail("hipMalloc (row descriptors) failed");
    sl->rowd.resize(S);
    for (size_t c = 0, at = 0; c < static_cast<size_t>(S); ++c) { sl->rowd[c] = sl->rowd_blob + at; at += (static_cast<size_t>(sl->n_of[c]) + 32) / 32 * 32; }
    if (hipMalloc(reinterpret_cast<void **>(&sl->stat), 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned)) != hipSuccess) return fail("hipMalloc (counters) failed");
    if (hipHostMalloc(reinterpret_cast<void **>(&sl->stat_host), 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)
        return fail("hipHostMalloc (counters) failed");
    std::memset(sl->stat_host, 0, 4 * (static_cast<size_t>(S) + 1) * sizeof(unsigned));
    sl->bytes = static_cast<double>(total) + 2.0 * sl->ent_cap * sizeof(uint2) + static_cast<double>(rowd_total * sizeof(uint2));
    sl->cap_cal = std::min(8192, (std::max(n_max, sl->n_of[S]) + 63) / 64 * 64);
    return sl;
}

void sparse_levels_destroy(SparseLevels *sl)
{
    if (!sl) return;
    if (sl->blob) (void)hipFree(sl->blob);
    for (int b = 0; b < 2; ++b) if (sl->ent[b]) (void)hipFree(sl->ent[b]);
    if (sl->rowd_blob) (void)hipFree(sl->rowd_blob);
    if (sl->stat) (void)hipFree(sl->stat);
    if (sl->stat_host) (void)hipHostFree(sl->stat_host);
    delete sl;
}

static SpArgs args_for(const SparseLevels *sl, int s)
{
    SpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ent_in = sl->ent[s & 1]; a.rowd_in = sl->rowd[s];
    a.ent_out = sl->ent[(s + 1) & 1]; a.rowd_out = s + 1 < sl->S ? sl->rowd[s + 1] : nullptr;
    a.fixed = sl->calibrated ? 1 : 0;
    a.srcA = sl->dev[s].srcA; a.srcB = sl->dev[s].srcB; a.ord = sl->dev[s].ord;
    a.ch_off = sl->ch_off[s]; a.ch = sl->ch[s];
    a.n_prev = sl->n_of[s]; a.n = sl->n_of[s + 1];
    a.ent_cap = static_cast<unsigned>(std::min<size_t>(sl->ent_cap, 0xffffffffu));
    a.scale_in = std::ldexp(1.0f, 2 * s + 1);
    a.unit_out = std::ldexp(1.0f, -(2 * s + 3));
    a.half_out = 1u << (2 * s + 2);
    a.stat = sl->stat + 4 * (s + 1);
    return a;
}

static int launch_step(SparseLevels *sl, int s, int cap, hipStream_t stream, std::string &err)
{
    if (s == 0) {
        const int n0 = sl->n_of[0], n_stat = 4 * (sl->S + 1);
        hipLaunchKernelGGL(sparse_identity_kernel, dim3((std::max(n0, n_stat) + 255) / 256), dim3(256), 0, stream, sl->ent[0], sl->rowd[0], n0,
                           sl->stat, n_stat);
        SP_TRY(hipGetLastError());
    }
    SpArgs a = args_for(sl, s);
    a.cap = cap;
    // long source rows: four wavefronts per row (a thread walks entries / (4 x 256) dependent chains instead of entries / (4 x 64))
    const bool wide = sl->nnz[s] > 96ll * sl->n_of[s];
    a.wp = wp_for(a.n, wide ? 256 : 64);
    const size_t lds = 6 * (static_cast<size_t>(a.wp) + static_cast<size_t>(a.cap));
    if (wide) hipLaunchKernelGGL(sparse_step_kernel<256>, dim3(static_cast<unsigned>(a.n)), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(sparse_step_kernel<64>, dim3(static_cast<unsigned>(a.n)), dim3(64), lds, stream, a);
    SP_TRY(hipGetLastError());
    return GENPHI_OK;
}

int sparse_levels_calibrate(SparseLevels *sl, hipStream_t stream, std::string &err)
{
    if (!sl) { err = "sparse_levels_calibrate: null handle"; return GENPHI_ERR_ARG; }
    if (sl->calibrated) return GENPHI_OK;
    sl->k = -1;
    if (sl->tun.force_k == -1) { sl->calibrated = true; return GENPHI_OK; }
    int k = 0;
    for (int s = 0; s + 1 < sl->S; ++s) {                  // cut s+1 may be a sparse source only when step s+1 is eligible too
        if (sl->tun.force_k >= 0 && s + 1 > sl->tun.force_k) break;
        const int rc = launch_step(sl, s, sl->cap_cal, stream, err);
        if (rc) return rc;
        unsigned st[4] = {0, 0, 0, 0};
        SP_TRY(hipMemcpyAsync(st, sl->stat + 4 * (s + 1), sizeof(st), hipMemcpyDeviceToHost, stream));
        SP_TRY(hipStreamSynchronize(stream));
        const double n = static_cast<double>(sl->n_of[s + 1]);
        const bool ovf = st[2] != 0;
        sl->nnz[s + 1] = ovf ? -1 : static_cast<long long>(st[0]);
        sl->max_row[s + 1] = static_cast<int>(st[1]);
        if (ovf) break;
        if (sl->tun.force_k < 0 && static_cast<double>(st[0]) > sl->tun.max_permille / 1000.0 * n * n) break;
        k = s + 1;
    }
    sl->calibrated = true;                                 // (from here on rows go where this run put them)
    if (k < 1) return GENPHI_OK;
    if (sl->tun.force_k < 0) {
        int widest = 0;
        for (int c = 0; c <= k + 1; ++c) widest = std::max(widest, sl->n_of[c]);
        if (widest < sl->tun.min_cut) return GENPHI_OK;
    }
    sl->k = k;
    return GENPHI_OK;
}

int sparse_levels_k(const SparseLevels *sl) { return sl ? sl->k : -1; }

int sparse_levels_enqueue_step(SparseLevels *sl, int s, hipStream_t stream, std::string &err)
{
    if (!sl || s < 0 || s >= sl->k) { err = "sparse_levels_enqueue_step: not a sparse step"; return GENPHI_ERR_ARG; }
    const int cap = std::min(sl->cap_cal, std::max(64, (sl->max_row[s + 1] + 63) / 64 * 64));
    return launch_step(sl, s, cap, stream, err);
}

int sparse_levels_enqueue_dense(SparseLevels *sl, float *out, long long ld, long long width, hipStream_t stream, std::string &err)
{
    if (!sl || sl->k < 1) { err = "sparse_levels_enqueue_dense: no sparse cut"; return GENPHI_ERR_ARG; }
    SpArgs a = args_for(sl, sl->k);
    a.out = out; a.ld = ld; a.width = static_cast<int>(width);
    a.chunk_cols = std::max(1024, std::min(sl->tun.chunk_cols, 15360) / 1024 * 1024);
    // (the "none" row spans the pitch, the member rows the width the caller names)
    a.n_chunks = static_cast<int>((std::max(ld, width) + a.chunk_cols - 1) / a.chunk_cols);
    const size_t lds = static_cast<size_t>(a.chunk_cols) * sizeof(unsigned);
    hipLaunchKernelGGL(sparse_dense_kernel, dim3(static_cast<unsigned>((a.n + 1)) * a.n_chunks), dim3(256), lds, stream, a);
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

int sparse_levels_counts(const SparseLevels *sl, int cap, long long *nnz, int *max_row)
{
    if (!sl) return 0;
    const int m = std::min(cap, sl->S + 1);
    for (int c = 0; c < m; ++c) {
        if (nnz) nnz[c] = sl->nnz[c];
        if (max_row) max_row[c] = sl->max_row[c];
    }
    return m;
}

double sparse_levels_device_bytes(const SparseLevels *sl) { return sl ? sl->bytes : 0.0; }

}  // namespace genphi
