// sparse_phi.hip -- gen.sparse_phi / KinshipMatrix on the GPU.
//
// Reference: src/compute.jl:321-447 (sparse_phi), :31-46 (KinshipMatrix, getindex by rank),
// :467-472 (phiMean(::KinshipMatrix)); pruning by branching(pedigree, pro = ...) src/extract.jl:65-186.
//
// What the reference does: individuals leave a FIFO queue one at a time (founders first, a child
// once both its parents are done); each gets its self kinship and its kinship with every LIVE
// earlier individual as RN32(phi[father, j]/2 + phi[mother, j]/2) (Float64 sum of two Float32
// halves, one Float32 store), and a non-proband parent is dropped once all its children are done.
// Values are stored under (rank of the earlier processed, rank of the later processed) but looked
// up under (smaller rank, larger rank): whenever two individuals leave the queue in the opposite
// order of their ranks, their kinship is stored where no lookup finds it -- it reads as 0 from then
// on (tests/oracle restate that behaviour; this file reproduces it, it does not "fix" it).  With
// genealogy(...; sort=true) that only happens inside one depth; with sort=false (rank = file
// position) it happens across depths too.
//
// Design here (not a translation): the queue order is depth-sorted (a child is enqueued while its
// deepest parent is processed) WHATEVER the ranks are, so all individuals of one depth -- a WAVE --
// only need kinships with strictly older individuals and with each other through those:
//   T[i][q]  = RN32(L(f_i, q)/2 + L(m_i, q)/2)      new i x every live older q
//   S[i][j]  = RN32(L'(j, f_i)/2 + L'(j, m_i)/2)    new i x new j, j processed before i, where
//              L'(j, p) = T[j][p] if rank(p) < rank(j) (p left the queue before j), else 0
//   S[i][i]  = RN32(1/2 + L(f_i, m_i)/2)
// with L(a, b) = the stored value if the (earlier, later) key equals the (smaller rank, larger rank)
// key, else 0 (src/compute.jl:366-390 looks up phi[min rank][max rank], :392-394 stores under
// [rank of the live one][rank of the new one]).  The live set is a dense matrix in HBM ("active
// matrix"), rebuilt after every wave as [survivors..., new...] (retired parents leave), like the cuts
// of the dense path with other membership rules.  Per wave, all streaming, no host round trip, TWO launches:
//   sparse_row_fused_kernel  one workgroup per new row i: T[i][.] built in LDS from the two parent rows of the active matrix
//                            (16-byte loads, masked by the key rule), its surviving columns written straight into the next
//                            matrix (new x survivors), then new x new for the later partners j > i by two LDS gathers per
//                            entry, and the self kinship; T never exists in HBM.  The workgroups after the rows compact
//                            survivors x survivors (a stream compaction of the old matrix: both parts only read it)
//   sparse_mirror_kernel     survivors x new and the lower triangle by 64 x 64 tile transposition
// (old sets above 36,864 members -- a T row beyond 144 KB of LDS -- take the round-3 form: sparse_rows_compact_kernel writes T to
// HBM, sparse_newnew_kernel gathers from it; GENPHI_SPARSE_NO_FUSED=1 forces that form.)
// The host simulates the queue once (integers only: processing order, waves, the wave after which
// every individual retires) and uploads every wave's index arrays in ONE blob before the sweep; the
// sweep is one stream of launches with a single synchronisation at its end.
//
// Entries that outlive their column (src/compute.jl:401-430 deletes phi[rank_j][parent] only for
// rank_j < parent rank): a proband j that left the queue before a non-proband x and has the larger rank
// keeps phi[rank_j][rank_x] for good -- `show` counts it, phiMean sums it.  The kernel that computes such
// an entry appends it (when > 0) to a list through an atomic counter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "devcache.h"
#include "planner.h"
#include "../../include/genphi.h"

int genphi_set_error(int code, const std::string &msg);      // genphi_hip.hip

namespace {

// "half" as the reference computes it: Float32 / 2 in Float32 (exact unless the result is subnormal)
__device__ __forceinline__ float half32(float v) { return v / 2.0f; }

// meta = (2 * rank + is_proband, processing index).  Distinct individuals have distinct ranks, so the
// order of the first words is the order of the ranks.  Does a lookup find the kinship of two DISTINCT
// individuals?  Only if the earlier processed one has the smaller rank.
__device__ __forceinline__ bool key_found(int2 ma, int2 mb) { return (ma.y < mb.y) == (ma.x < mb.x); }

struct StaleOut {            // entries that outlive their column: (row rank, column rank, value) appended through cnt
    int *cnt;
    int cap;
    int2 *rc;
    float *val;
};

__device__ __forceinline__ void stale_append(int *cnt, int cap, int2 *rc, float *val, int row_rank, int col_rank, float v)
{
    const int k = atomicAdd(cnt, 1);
    if (k < cap) { rc[k] = make_int2(row_rank, col_rank); val[k] = v; }
}

// T[i][q] of one new row i and one old slot q (see sparse_rows_kernel); L(parent, q): the self entry of the parent is always
// found, any other only under the key rule
__device__ __forceinline__ float rows_entry(int q, int n_old, int2 mq, float a, float b, bool hasF, bool hasM, int fx, int mx, int2 mf, int2 mm)
{
    const float lf = (hasF && (q == fx || key_found(mf, mq))) ? a : 0.f;
    const float lm = (hasM && (q == mx || key_found(mm, mq))) ? b : 0.f;
    return q < n_old ? static_cast<float>(0.0 + static_cast<double>(half32(lf)) + static_cast<double>(half32(lm))) : 0.f;
}

// New rows against the old members.  par[i] = (father slot, mother slot, 2 rank + pro of the father, of the
// mother) of new individual i (slot n_old = none); newpos[q] = column of old slot q in the next matrix or -1.
//   T[i][q]                     = RN32(L(f_i, q)/2 + L(m_i, q)/2)   for every old q
//   next[n_surv + i][newpos[q]] = the same, where q survives
// Four consecutive old slots per thread: 16-byte loads of the parent rows and of the index words.  Rows of M
// and T are padded to a multiple of 64 floats and every index array to 256 bytes, so the quads need no clamp.
// kRowsPerBlock new rows per workgroup share the index words of the thread's four slots (meta: 32 bytes, newpos: 16 bytes --
// more than the 32 bytes of matrix data one row needs there).
constexpr int kRowsPerBlock = 4;
__device__ __forceinline__ void
sparse_rows_body(int bx, int by, const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old, const int4 *__restrict__ par,
                 const int2 *__restrict__ meta_new, int n_new, const int *__restrict__ newpos, float *__restrict__ T, long long ldT,
                 float *__restrict__ next, long long ld_next, int n_surv, int *__restrict__ so_cnt, int so_cap, int2 *__restrict__ so_rc,
                 float *__restrict__ so_val)
{
    const int q0 = (by * 256 + threadIdx.x) * 4;
    if (q0 >= n_old) return;
    const int4 m01 = *reinterpret_cast<const int4 *>(meta + q0), m23 = *reinterpret_cast<const int4 *>(meta + q0 + 2);
    const int4 np4 = *reinterpret_cast<const int4 *>(newpos + q0);
    // all source-row loads of the block's rows are issued before the first use (8 float4 loads in flight per lane)
    int4 pr[kRowsPerBlock];
    float4 a4s[kRowsPerBlock], b4s[kRowsPerBlock];
#pragma unroll
    for (int rr = 0; rr < kRowsPerBlock; ++rr) {
        const int i = min(bx * kRowsPerBlock + rr, n_new - 1);
        pr[rr] = par[i];
        // (unconditional loads from a valid row: a select between a global pointer and a zero constant becomes a FLAT load of a
        // private copy; rows_entry drops the values of a missing parent)
        a4s[rr] = *reinterpret_cast<const float4 *>(M + (long long)(pr[rr].x != n_old ? pr[rr].x : 0) * ld + q0);
        b4s[rr] = *reinterpret_cast<const float4 *>(M + (long long)(pr[rr].y != n_old ? pr[rr].y : 0) * ld + q0);
    }
#pragma unroll
    for (int rr = 0; rr < kRowsPerBlock; ++rr) {
        const int i = bx * kRowsPerBlock + rr;
        if (i >= n_new) break;                                       // (workgroup-uniform)
        const int4 p = pr[rr];
        const float4 a4 = a4s[rr], b4 = b4s[rr];
        const bool hasF = p.x != n_old, hasM = p.y != n_old;         // workgroup-uniform
        const int2 mf = hasF ? meta[p.x] : make_int2(0, 0), mm = hasM ? meta[p.y] : make_int2(0, 0);
        const int2 mi = meta_new[i];
        float *trow = T + (long long)i * ldT;
        float *orow = next + (long long)(n_surv + i) * ld_next;
        const float v0 = rows_entry(q0, n_old, make_int2(m01.x, m01.y), a4.x, b4.x, hasF, hasM, p.x, p.y, mf, mm);
        const float v1 = rows_entry(q0 + 1, n_old, make_int2(m01.z, m01.w), a4.y, b4.y, hasF, hasM, p.x, p.y, mf, mm);
        const float v2 = rows_entry(q0 + 2, n_old, make_int2(m23.x, m23.y), a4.z, b4.z, hasF, hasM, p.x, p.y, mf, mm);
        const float v3 = rows_entry(q0 + 3, n_old, make_int2(m23.z, m23.w), a4.w, b4.w, hasF, hasM, p.x, p.y, mf, mm);
        *reinterpret_cast<float4 *>(trow + q0) = make_float4(v0, v1, v2, v3);
        // next[new row][survivor column]; phi[rank_q][rank_i] outlives i's retirement when q is a proband with the larger rank (i a non-proband)
#define GENPHI_SPARSE_EMIT(U, V, MQX, NP)                                                                                         \
        if (q0 + U < n_old) {                                                                                                     \
            if (NP >= 0) orow[NP] = V;                                                                                            \
            if (V > 0.f && !(mi.x & 1) && (MQX & 1) && MQX > mi.x) stale_append(so_cnt, so_cap, so_rc, so_val, MQX >> 1, mi.x >> 1, V); \
        }
        GENPHI_SPARSE_EMIT(0, v0, m01.x, np4.x)
        GENPHI_SPARSE_EMIT(1, v1, m01.z, np4.y)
        GENPHI_SPARSE_EMIT(2, v2, m23.x, np4.z)
        GENPHI_SPARSE_EMIT(3, v3, m23.z, np4.w)
#undef GENPHI_SPARSE_EMIT
    }
}

// survivors x survivors: next[r][c] = M[keep[r]][keep[c]] (keep ascending: a stream compaction)
template <int NT = 256>
__device__ __forceinline__ void
sparse_compact_body(int bx, int by, const float *__restrict__ M, long long ld, const int *__restrict__ keep, int n_surv, float *__restrict__ next,
                    long long ld_next)
{
    constexpr int U = 8;
    const float *src = M + (long long)keep[bx] * ld;
    float *dst = next + (long long)bx * ld_next;
    const int c0 = by * (NT * U) + threadIdx.x;
    int q[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = keep[min(c0 + u * NT, n_surv - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[q[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * NT;
        if (c < n_surv) dst[c] = v[u];
    }
}

// One launch per wave for the two passes that only READ the old active matrix (round 4; four launches per wave -> three): the first
// rows_bx * rows_by workgroups compute the new rows (sparse_rows_body), the others compact survivors x survivors (sparse_compact_body).
// Both stream independent parts of the next matrix; the short waves of a deep pedigree are bound by their launches.
__global__ void __launch_bounds__(256)
sparse_rows_compact_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old, const int4 *__restrict__ par,
                           const int2 *__restrict__ meta_new, int n_new, const int *__restrict__ newpos, float *__restrict__ T, long long ldT,
                           float *__restrict__ next, long long ld_next, int n_surv, int *__restrict__ so_cnt, int so_cap, int2 *__restrict__ so_rc,
                           float *__restrict__ so_val, const int *__restrict__ keep, int rows_bx, int rows_by)
{
    const int b = blockIdx.x, rb = rows_bx * rows_by;
    if (b < rb) {
        sparse_rows_body(b % rows_bx, b / rows_bx, M, ld, meta, n_old, par, meta_new, n_new, newpos, T, ldT, next, ld_next, n_surv, so_cnt, so_cap, so_rc, so_val);
    } else {
        const int c = b - rb;
        sparse_compact_body(c % n_surv, c / n_surv, M, ld, keep, n_surv, next, ld_next);
    }
}

// new x new, in queue order (new index = position in the wave): row a holds the entries with the LATER ones
// b > a and its own self kinship,
//   next[n_surv + a][n_surv + b] = RN32(L'(a, f_b)/2 + L'(a, m_b)/2),  L'(a, p) = T[a][p] if rank(p) < rank(a) else 0
//   next[n_surv + a][n_surv + a] = RN32(1/2 + L(f_a, m_a)/2)
// Row a of T is staged in LDS when it fits (lds_floats >= n_old), else gathered from L2.
// partners b > a of new row a; `src` = T[a][.] (global) or its LDS copy -- two instantiations, so that neither gathers through FLAT loads
template <int NT = 256, typename Src>
__device__ __forceinline__ void newnew_partners(Src src, int a, int n_old, int n_new, int2 ma, const int4 *__restrict__ par,
                                                const int2 *__restrict__ meta_new, float *__restrict__ orow, const StaleOut &so)
{
    for (int b0 = a + 1 + threadIdx.x; b0 < n_new; b0 += 4 * NT) {
        int4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = par[min(b0 + u * NT, n_new - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * NT;
            if (b >= n_new) continue;
            // the parents left the queue before a (an earlier wave): T[a][parent] sits under (rank parent, rank a),
            // which the lookup (smaller rank, larger rank) finds only when rank parent < rank a
            const float tf = (p[u].x != n_old && p[u].z < ma.x) ? src[p[u].x] : 0.f;
            const float tm = (p[u].y != n_old && p[u].w < ma.x) ? src[p[u].y] : 0.f;
            const float v = static_cast<float>(0.0 + static_cast<double>(half32(tf)) + static_cast<double>(half32(tm)));
            orow[b] = v;
            if (v > 0.f && (ma.x & 1)) {
                const int2 mb = meta_new[b];                      // b is the later one: phi[rank_a][rank_b] outlives b's retirement
                if (!(mb.x & 1) && ma.x > mb.x) stale_append(so.cnt, so.cap, so.rc, so.val, ma.x >> 1, mb.x >> 1, v);      // if a is a proband with the larger rank
            }
        }
    }
}

__global__ void __launch_bounds__(256)
sparse_newnew_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old, const int4 *__restrict__ par,
                     const int2 *__restrict__ meta_new, const float *__restrict__ T, long long ldT, int n_new, int lds_floats,
                     float *__restrict__ next, long long ld_next, int n_surv, StaleOut so)
{
    extern __shared__ float srow[];
    const int a = blockIdx.x;
    const float *trow = T + (long long)a * ldT;
    // (staging the row costs n_old loads: not worth it for the last rows of the wave, which have few later partners)
    const bool in_lds = lds_floats >= n_old && 8 * (n_new - a - 1) >= n_old;
    if (in_lds) {                                                 // (uniform: a and n_new are)
        const float4 *g4 = reinterpret_cast<const float4 *>(trow);
        float4 *s4 = reinterpret_cast<float4 *>(srow);
        for (int k = threadIdx.x; k < (n_old + 3) / 4; k += 256) s4[k] = g4[k];       // (T's pitch is a multiple of 64 floats)
        __syncthreads();
    }
    const int2 ma = meta_new[a];
    float *orow = next + (long long)(n_surv + a) * ld_next + n_surv;
    if (threadIdx.x == 0) {
        const int4 p = par[a];
        double cf = 0.5;
        if (p.x != n_old && p.y != n_old && (p.x == p.y || key_found(meta[p.x], meta[p.y])))
            cf += static_cast<double>(half32(M[(long long)p.x * ld + p.y]));
        orow[a] = static_cast<float>(cf);
    }
    if (in_lds) newnew_partners(static_cast<const float *>(srow), a, n_old, n_new, ma, par, meta_new, orow, so);
    else newnew_partners(trow, a, n_old, n_new, ma, par, meta_new, orow, so);
}

// One new row end to end (round 4, second pass): T[a][.] is only ever read by row a's own new x new entries, so the row is built in
// LDS from the two parent rows of the old matrix (the same rows_entry as sparse_rows_body), its surviving columns go straight to the
// next matrix, and the partners b > a gather from LDS -- T never touches HBM (one write and one read of n_new x n_old floats less per
// wave) and a wave is two launches.  Workgroups past the n_new rows compact survivors x survivors as before.  Old sets of up to 36,864
// members (144 KB of LDS); wider ones keep the two-kernel form above.
// (measured: 1 quad per lane and trip with 256 threads beats 2 and 4 quads and 512 threads -- fewer registers, more resident
// workgroups: r04_ab_sparse_phi_fused_row_quads_per_lane_and_threads.out)
#ifndef GENPHI_SPARSE_QUADS_PER_LANE
#define GENPHI_SPARSE_QUADS_PER_LANE 1
#endif
#ifndef GENPHI_SPARSE_ROW_THREADS
#define GENPHI_SPARSE_ROW_THREADS 256
#endif
constexpr int kRowThreads = GENPHI_SPARSE_ROW_THREADS;
__global__ void __launch_bounds__(kRowThreads)
sparse_row_fused_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old, const int4 *__restrict__ par,
                        const int2 *__restrict__ meta_new, int n_new, const int *__restrict__ newpos, float *__restrict__ next, long long ld_next,
                        int n_surv, StaleOut so, const int *__restrict__ keep)
{
    extern __shared__ float srow[];
    if (static_cast<int>(blockIdx.x) >= n_new) {
        const int c = blockIdx.x - n_new;
        sparse_compact_body<kRowThreads>(c % n_surv, c / n_surv, M, ld, keep, n_surv, next, ld_next);
        return;
    }
    const int a = blockIdx.x;
    const int4 p = par[a];
    const bool hasF = p.x != n_old, hasM = p.y != n_old;             // workgroup-uniform
    const int2 mf = hasF ? meta[p.x] : make_int2(0, 0), mm = hasM ? meta[p.y] : make_int2(0, 0);
    const int2 mi = meta_new[a];
    const float *rowF = M + (long long)(hasF ? p.x : 0) * ld, *rowM = M + (long long)(hasM ? p.y : 0) * ld;
    float *orow_s = next + (long long)(n_surv + a) * ld_next;
    const int nquad = (n_old + 3) / 4;
    constexpr int QL = GENPHI_SPARSE_QUADS_PER_LANE;                  // quads per lane and trip: 2 x QL 16-byte matrix loads in flight
    for (int k0 = threadIdx.x; k0 < nquad; k0 += QL * kRowThreads) {
        float4 a4[QL], b4[QL];
        int4 m01[QL], m23[QL], np4[QL];
#pragma unroll
        for (int u = 0; u < QL; ++u) {
            const int q0 = 4 * min(k0 + u * kRowThreads, nquad - 1);
            a4[u] = *reinterpret_cast<const float4 *>(rowF + q0);
            b4[u] = *reinterpret_cast<const float4 *>(rowM + q0);
            m01[u] = *reinterpret_cast<const int4 *>(meta + q0); m23[u] = *reinterpret_cast<const int4 *>(meta + q0 + 2);
            np4[u] = *reinterpret_cast<const int4 *>(newpos + q0);
        }
#pragma unroll
        for (int u = 0; u < QL; ++u) {
            if (k0 + u * kRowThreads >= nquad) continue;
            const int q0 = 4 * (k0 + u * kRowThreads);
            const float v0 = rows_entry(q0, n_old, make_int2(m01[u].x, m01[u].y), a4[u].x, b4[u].x, hasF, hasM, p.x, p.y, mf, mm);
            const float v1 = rows_entry(q0 + 1, n_old, make_int2(m01[u].z, m01[u].w), a4[u].y, b4[u].y, hasF, hasM, p.x, p.y, mf, mm);
            const float v2 = rows_entry(q0 + 2, n_old, make_int2(m23[u].x, m23[u].y), a4[u].z, b4[u].z, hasF, hasM, p.x, p.y, mf, mm);
            const float v3 = rows_entry(q0 + 3, n_old, make_int2(m23[u].z, m23[u].w), a4[u].w, b4[u].w, hasF, hasM, p.x, p.y, mf, mm);
            *reinterpret_cast<float4 *>(srow + q0) = make_float4(v0, v1, v2, v3);
#define GENPHI_SPARSE_EMIT(U, V, MQX, NP)                                                                                         \
            if (q0 + U < n_old) {                                                                                                 \
                if (NP >= 0) orow_s[NP] = V;                                                                                      \
                if (V > 0.f && !(mi.x & 1) && (MQX & 1) && MQX > mi.x) stale_append(so.cnt, so.cap, so.rc, so.val, MQX >> 1, mi.x >> 1, V); \
            }
            GENPHI_SPARSE_EMIT(0, v0, m01[u].x, np4[u].x)
            GENPHI_SPARSE_EMIT(1, v1, m01[u].z, np4[u].y)
            GENPHI_SPARSE_EMIT(2, v2, m23[u].x, np4[u].z)
            GENPHI_SPARSE_EMIT(3, v3, m23[u].z, np4[u].w)
#undef GENPHI_SPARSE_EMIT
        }
    }
    __syncthreads();
    float *orow = orow_s + n_surv;
    if (threadIdx.x == 0) {
        double cf = 0.5;
        if (hasF && hasM && (p.x == p.y || key_found(mf, mm))) cf += static_cast<double>(half32(rowF[p.y]));
        orow[a] = static_cast<float>(cf);
    }
    newnew_partners<kRowThreads>(static_cast<const float *>(srow), a, n_old, n_new, mi, par, meta_new, orow, so);
}

// next[c][r] = next[r][c] for the new rows r >= n_surv and the columns c < n_surv (survivors x new) or c > r
// (lower triangle of new x new); 64 x 64 tiles through LDS, both sides coalesced
__global__ void __launch_bounds__(256)
sparse_mirror_kernel(float *__restrict__ next, long long ld, int n_surv, int n_next)
{
    __shared__ float tile[64][65];
    const int r0 = n_surv + blockIdx.y * 64, c0 = blockIdx.x * 64;
    if (c0 >= n_surv && c0 + 63 <= r0) return;                    // a tile on or below the diagonal of new x new with nothing to mirror
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                 // 8 loads in flight per thread (clamped: unconditional loads)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = min(r0 + ty + 4 * (8 * h + u), n_next - 1), c = min(c0 + tx, n_next - 1);
            v[u] = next[(long long)r * ld + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tile[ty + 4 * (8 * h + u)][tx] = v[u];
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int c = c0 + k, r = r0 + tx;
        if (r < n_next && c < n_next && (c < n_surv || c > r)) next[(long long)c * ld + r] = tile[tx][k];
    }
}

long long pitch_of(long long n) { return ((n + 1) + 63) / 64 * 64; }

struct Wave {
    int n_old = 0, n_new = 0, n_surv = 0;
    // byte offsets in the device blob (the host image is written in place, wave by wave):
    //   par       per new individual: (father slot, mother slot, 2 rank + pro of the father, of the mother); slots in the old active list
    //   keep      old slots that survive the wave, ascending
    //   newpos    per old slot: its slot in the next list, or -1
    //   meta_new  (2 rank + pro, processing index) of the new individuals, queue order
    //   meta_old  ... of the old active list, slot order
    size_t o_par = 0, o_keep = 0, o_newpos = 0, o_meta_new = 0, o_meta_old = 0;
};

// Host memory of a result block, recycled through a one-slot cache: a fresh 16 MB block costs its page faults again on every call
// (glibc maps and unmaps blocks of that size), 1-2 ms of a 9 ms call at 2,000 probands.  Uninitialised on purpose: every entry is written.
struct FloatBlock {
    float *p = nullptr;
    size_t cap = 0;
    FloatBlock() = default;
    FloatBlock(const FloatBlock &) = delete;
    FloatBlock &operator=(const FloatBlock &) = delete;
    ~FloatBlock() { give_back(); }
    float operator[](size_t i) const { return p[i]; }
    static std::mutex &mu() { static std::mutex m; return m; }
    static FloatBlock *&slot() { static FloatBlock *s = nullptr; return s; }
    bool take(size_t n)
    {
        {
            std::lock_guard<std::mutex> lock(mu());
            FloatBlock *&c = slot();
            if (c && c->cap >= n && c->cap <= 4 * n + (1u << 20)) { p = c->p; cap = c->cap; c->p = nullptr; c->cap = 0; delete c; c = nullptr; return true; }
        }
        p = static_cast<float *>(std::malloc(std::max<size_t>(n, 1) * sizeof(float)));
        cap = p ? n : 0;
        return p != nullptr;
    }
    void give_back()
    {
        if (!p) return;
        if (cap * sizeof(float) <= (size_t(1) << 30)) {
            std::lock_guard<std::mutex> lock(mu());
            FloatBlock *&c = slot();
            if (!c) { c = new (std::nothrow) FloatBlock(); if (c) { c->p = p; c->cap = cap; p = nullptr; cap = 0; return; } }
        }
        std::free(p);
        p = nullptr; cap = 0;
    }
};

// The device side of a call -- one pool allocation, the stream, the timing events -- kept for the next call on the same device
// (one slot; pools above GENPHI_SPARSE_KEEP_MB, default 1024, are freed as before): hipMalloc + hipStreamCreate + hipFree with its
// device-wide synchronisation + the event churn are 1.5-2 ms of a 9 ms call.
struct DeviceSide {
    char *pool = nullptr;
    size_t bytes = 0;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    int device = -1;
    void destroy()
    {
        if (pool) (void)hipFree(pool);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
        pool = nullptr; bytes = 0; st = nullptr; ev.clear();
    }
};
std::mutex g_side_mu;
DeviceSide g_side_kept;
// pinned staging buffer of the proband block, kept between calls; capped (kPinCap): larger blocks take the plain 2D copy
std::mutex g_pin_mu;
void *g_pin = nullptr;
size_t g_pin_bytes = 0;
constexpr size_t kPinCap = size_t(256) << 20;

}  // namespace

// what gen.sparse_phi keeps between calls (the device side of the last call, the pinned staging buffer) goes back to the
// driver: part of genphi_release_cached
void genphi::sparse_phi_release_kept()
{
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        if (g_side_kept.st || g_side_kept.pool) {
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (g_side_kept.device >= 0) (void)hipSetDevice(g_side_kept.device);
            g_side_kept.destroy();
            g_side_kept = DeviceSide();
            if (cur >= 0) (void)hipSetDevice(cur);
        }
    }
    std::lock_guard<std::mutex> lock(g_pin_mu);
    if (g_pin) (void)hipHostFree(g_pin);
    g_pin = nullptr; g_pin_bytes = 0;
}

struct genphi_sparse {
    int64_t n_pro = 0;                        // distinct probands
    std::vector<int64_t> ids;                 // proband IDs, first-occurrence order
    std::vector<int> rank, proc;              // of each proband (rank in the pruned pedigree, processing index)
    std::vector<int> slot;                    // row / column of each proband in S
    FloatBlock S;                             // n_pro x n_pro: stored value of every pair of probands
    std::vector<int> stale_row_rank, stale_col_rank;   // entries that survive in a proband's dictionary
    std::vector<float> stale_val;
    std::unordered_map<int64_t, int> pos;     // ID -> index into ids
    // measurement (genphi_sparse_stats)
    double sweep_ms = 0.0, algorithmic_bytes = 0.0;
    std::vector<float> wave_ms;
    std::vector<double> wave_bytes;
    int64_t max_active = 0;
};

extern "C" {

// The host's schedule of a sweep, for tests that need no GPU (genphi_sparse_schedule): per processed individual, in processing order,
// its ID, the processing index at which it leaves the live set (-1: a proband, never) and its wave.
struct ScheduleOut {
    int64_t cap = 0, n = 0;
    int64_t *ids = nullptr, *retire_at = nullptr;
    int32_t *wave = nullptr;
};

static int sparse_impl(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                       const int64_t *pro_ids, int32_t device, genphi_sparse **out, ScheduleOut *sched)
{
    if (!out) return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_phi: out is NULL");
    *out = nullptr;
    if (n_ind < 0 || n_pro < 0 || (n_ind > 0 && (!ind || !father || !mother)) || (n_pro > 0 && !pro_ids))
        return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_phi: null or negative argument");
    genphi_sparse *R = new (std::nothrow) genphi_sparse();
    if (!R) return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory");
    // GENPHI_TRACE=1: wall-clock marks of the phases of this call on stderr
    const bool tracing = std::getenv("GENPHI_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!tracing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[genphi trace] sparse: %-24s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    auto bail = [&](int code, const std::string &msg) { delete R; return genphi_set_error(code, msg); };

    // ---- the pruned pedigree: probands and their ancestors, in pedigree order (branching) ----------
    // ID -> position: pedigree IDs are usually small dense integers (a direct table then: a hash map of 1e5 IDs costs 2 ms of a 12 ms call)
    struct IdAt {
        std::vector<int> table;
        std::unordered_map<int64_t, int> map;
        bool direct = false;
        int find(int64_t id) const
        {
            if (direct) return (id < 0 || id >= static_cast<int64_t>(table.size())) ? -1 : table[id];
            auto it = map.find(id);
            return it == map.end() ? -1 : it->second;
        }
        bool insert(int64_t id, int v)
        {
            if (direct) { if (table[id] >= 0) return false; table[id] = v; return true; }
            return map.emplace(id, v).second;
        }
    } at;
    {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int64_t i = 0; i < n_ind; ++i) { lo = std::min(lo, ind[i]); hi = std::max(hi, ind[i]); }
        // (genea140: 41,523 IDs up to 900,506 -- a 3.6 MB table filled in 0.2 ms against 1.4 ms of hashing)
        if (n_ind > 0 && lo >= 0 && hi < 64 * n_ind + (1 << 20)) { at.table.assign(static_cast<size_t>(hi) + 1, -1); at.direct = true; }
        else at.map.reserve(static_cast<size_t>(n_ind) * 2);
    }
    std::vector<int> fa(n_ind, -1), mo(n_ind, -1);
    for (int64_t i = 0; i < n_ind; ++i) {
        if (father[i] != 0) { fa[i] = at.find(father[i]); if (fa[i] < 0) return bail(GENPHI_ERR_ORDER, "parent listed after its child or unknown"); }
        if (mother[i] != 0) { mo[i] = at.find(mother[i]); if (mo[i] < 0) return bail(GENPHI_ERR_ORDER, "parent listed after its child or unknown"); }
        if (!at.insert(ind[i], static_cast<int>(i))) return bail(GENPHI_ERR_DUPLICATE_ID, "duplicate individual ID " + std::to_string(ind[i]));
    }
    std::vector<char> keep(n_ind, 0), is_pro(n_ind, 0);
    for (int64_t k = 0; k < n_pro; ++k) {
        const int x = at.find(pro_ids[k]);
        if (x < 0) return bail(GENPHI_ERR_UNKNOWN_ID, "KeyError: proband " + std::to_string(pro_ids[k]) + " not found");
        keep[x] = 1;
        if (!is_pro[x]) { is_pro[x] = 1; R->pos.emplace(pro_ids[k], static_cast<int>(R->ids.size())); R->ids.push_back(pro_ids[k]); }
    }
    for (int64_t x = n_ind - 1; x >= 0; --x)                     // parents precede children: one reverse sweep
        if (keep[x]) { if (fa[x] >= 0) keep[fa[x]] = 1; if (mo[x] >= 0) keep[mo[x]] = 1; }
    mark("id map, pruning marks");
    std::vector<int> iso_of(n_ind, -1), orig;                    // pruned index <-> original index
    for (int64_t x = 0; x < n_ind; ++x) if (keep[x]) { iso_of[x] = static_cast<int>(orig.size()); orig.push_back(static_cast<int>(x)); }
    const int m = static_cast<int>(orig.size());                 // rank of pruned index u is u + 1
    R->n_pro = static_cast<int64_t>(R->ids.size());
    if (m == 0) { if (sched) { sched->n = 0; delete R; return GENPHI_OK; } *out = R; return GENPHI_OK; }
    if (m >= (1 << 30)) return bail(GENPHI_ERR_ARG, "genphi_sparse_phi: more than 2^30 individuals");
    std::vector<int> pf(m), pm(m), depth(m), nchild(m, 0);
    std::vector<char> pro_flag(m);
    for (int u = 0; u < m; ++u) {
        const int x = orig[u];
        pf[u] = fa[x] >= 0 ? iso_of[fa[x]] : -1;
        pm[u] = mo[x] >= 0 ? iso_of[mo[x]] : -1;
        pro_flag[u] = is_pro[x];
        depth[u] = 1 + std::max(pf[u] >= 0 ? depth[pf[u]] : 0, pm[u] >= 0 ? depth[pm[u]] : 0);
        if (pf[u] >= 0) nchild[pf[u]]++;
        if (pm[u] >= 0) nchild[pm[u]]++;
    }
    std::vector<int> cstart(m + 1, 0), cfill(m, 0), clist;
    for (int u = 0; u < m; ++u) cstart[u + 1] = cstart[u] + nchild[u];
    clist.resize(cstart[m]);
    for (int u = 0; u < m; ++u) {                                // children in pedigree order (src/compute.jl:178-185)
        if (pf[u] >= 0) clist[cstart[pf[u]] + cfill[pf[u]]++] = u;
        if (pm[u] >= 0) clist[cstart[pm[u]] + cfill[pm[u]]++] = u;
    }

    // ---- the queue, integers only: processing order and the processing index at which each
    //      non-proband is retired (src/compute.jl:336-345, :397-439) -----------------------------------
    // The reference pushes a child when its second known parent is processed (or its only one) and skips a second pop of the same
    // individual; a countdown of known parents per child gives the same pushes in the same positions, so the processing order IS the
    // queue array (a father who is also the mother lists the child twice in a row: two decrements, one push where the second would be).
    // An individual is retired when its last child has been processed: retire = the largest processing index among its children,
    // i.e. the last one written below (the reference counts the children down, src/compute.jl:425-433: same index).
    std::vector<int> proc(m, -1), retire(m, INT32_MAX), order(m + 1);
    {
        std::vector<unsigned char> need(m);
        std::vector<std::pair<int64_t, int>> founders;
        for (int u = 0; u < m; ++u) {
            need[u] = static_cast<unsigned char>((pf[u] >= 0) + (pm[u] >= 0));
            if (!need[u]) founders.emplace_back(ind[orig[u]], u);
        }
        std::sort(founders.begin(), founders.end());             // founder(): IDs ascending
        int tail = 0;
        for (auto &e : founders) order[tail++] = e.second;
        for (int head = 0; head < tail; ++head) {
            const int u = order[head];
            proc[u] = head;
            if (pf[u] >= 0) retire[pf[u]] = head;
            if (pm[u] >= 0) retire[pm[u]] = head;
            for (int k = cstart[u], ke = cstart[u + 1]; k < ke; ++k) {
                const int c = clist[k];
                order[tail] = c;                                 // (kept only if this was the child's last missing parent: no
                tail += (--need[c] == 0);                        //  data-dependent branch, the coin flip costs more than the store)
            }
        }
        order.resize(tail);
        for (int u = 0; u < m; ++u) if (pro_flag[u]) retire[u] = INT32_MAX;      // probands stay to the end
    }
    if (static_cast<int>(order.size()) != m) return bail(GENPHI_ERR_ARG, "internal: the queue did not reach every individual");
    for (int k = 1; k < m; ++k)
        if (depth[order[k]] < depth[order[k - 1]]) return bail(GENPHI_ERR_ARG, "internal: processing order is not depth-sorted");

    mark("children lists, queue");
    // ---- waves: active lists, parents' slots, survivors; the layout of ONE device blob of every index array ----
    auto meta_of = [&](int u) { return make_int2(2 * (u + 1) + (pro_flag[u] ? 1 : 0), proc[u]); };
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    std::vector<Wave> waves;
    std::vector<int> active, next;                               // pruned indices, slot order
    std::vector<int> slot_of(m, -1);
    // the host image of the device blob, written in place; the buffer is kept per thread between calls (no reallocation, no page
    // faults in the steady state; padding and the unused tail of a wave's keep list are never read on the device)
    static thread_local std::vector<char> blob_kept;
    std::vector<char> &blob = blob_kept;
    struct BlobTrim { std::vector<char> &b; ~BlobTrim() { if (b.capacity() > (size_t(64) << 20)) std::vector<char>().swap(b); } } blob_trim{blob};
    if (blob.size() < 256) blob.resize(256);
    size_t max_mat = 64, max_T = 64, blob_bytes = 256;
    for (int b = 0; b < m;) {
        int e = b;
        while (e < m && depth[order[e]] == depth[order[b]]) ++e;
        Wave w;
        w.n_old = static_cast<int>(active.size());
        w.n_new = e - b;
        const int last_proc = e - 1;
        w.o_par = blob_bytes; blob_bytes += al(static_cast<size_t>(w.n_new) * sizeof(int4));
        w.o_keep = blob_bytes; blob_bytes += al(static_cast<size_t>(w.n_old) * sizeof(int));        // (room for every old slot)
        w.o_newpos = blob_bytes; blob_bytes += al(static_cast<size_t>(w.n_old) * sizeof(int));
        w.o_meta_new = blob_bytes; blob_bytes += al(static_cast<size_t>(w.n_new) * sizeof(int2));
        w.o_meta_old = blob_bytes; blob_bytes += al(static_cast<size_t>(w.n_old) * sizeof(int2));
        if (blob.size() < blob_bytes) blob.resize(std::max(blob_bytes, blob.size() * 2));
        int4 *par = reinterpret_cast<int4 *>(blob.data() + w.o_par);
        int *keep = reinterpret_cast<int *>(blob.data() + w.o_keep), *newpos = reinterpret_cast<int *>(blob.data() + w.o_newpos);
        int2 *meta_new = reinterpret_cast<int2 *>(blob.data() + w.o_meta_new), *meta_old = reinterpret_cast<int2 *>(blob.data() + w.o_meta_old);
        for (int k = b; k < e; ++k) {
            const int u = order[k];
            const int f = pf[u], mth = pm[u];
            par[k - b] = make_int4(f >= 0 ? slot_of[f] : w.n_old, mth >= 0 ? slot_of[mth] : w.n_old, f >= 0 ? meta_of(f).x : 0, mth >= 0 ? meta_of(mth).x : 0);
            meta_new[k - b] = meta_of(u);
        }
        next.clear();
        for (int sl = 0; sl < w.n_old; ++sl) {
            const int u = active[sl];
            meta_old[sl] = meta_of(u);
            if (retire[u] > last_proc) { newpos[sl] = static_cast<int>(next.size()); keep[next.size()] = sl; slot_of[u] = static_cast<int>(next.size()); next.push_back(u); }
            else newpos[sl] = -1;
        }
        w.n_surv = static_cast<int>(next.size());
        for (int k = b; k < e; ++k) { slot_of[order[k]] = static_cast<int>(next.size()); next.push_back(order[k]); }
        max_mat = std::max(max_mat, static_cast<size_t>((next.size() + 1) * pitch_of(static_cast<long long>(next.size()))));
        max_T = std::max(max_T, static_cast<size_t>(w.n_new) * static_cast<size_t>(pitch_of(w.n_old)));
        R->max_active = std::max<int64_t>(R->max_active, static_cast<int64_t>(next.size()));
        R->wave_bytes.push_back(4.0 * (static_cast<double>(w.n_old) * w.n_old + static_cast<double>(next.size()) * next.size()));
        active.swap(next);
        waves.push_back(w);
        b = e;
    }
    // the final active list is exactly the probands
    if (static_cast<int64_t>(active.size()) != R->n_pro) return bail(GENPHI_ERR_ARG, "internal: final active set is not the proband set");
    for (double x : R->wave_bytes) R->algorithmic_bytes += x;

    mark("waves, blob image");
    if (sched) {                                                  // the schedule only: no device work
        sched->n = m;
        int64_t k = 0;
        for (size_t wi = 0; wi < waves.size(); ++wi)
            for (int j = 0; j < waves[wi].n_new; ++j, ++k)
                if (k < sched->cap) {
                    const int u = order[k];
                    if (sched->ids) sched->ids[k] = ind[orig[u]];
                    if (sched->retire_at) sched->retire_at[k] = retire[u] == INT32_MAX ? -1 : retire[u];
                    if (sched->wave) sched->wave[k] = static_cast<int32_t>(wi);
                }
        delete R;
        return GENPHI_OK;
    }
    // ---- the device sweep: every launch of every wave in stream order, one synchronisation at the end ------
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return bail(GENPHI_ERR_DEVICE, "no HIP device available: gen.sparse_phi has no CPU fallback");
    if (device >= 0) { if (device >= ndev) return bail(GENPHI_ERR_DEVICE, "device ordinal out of range"); (void)hipSetDevice(device); }
    float *dM[2] = {nullptr, nullptr}, *dT = nullptr, *d_sval = nullptr;
    char *d_blob = nullptr;
    int2 *d_src = nullptr;
    int *d_cnt = nullptr;
    DeviceSide side;
    {   // the device side kept by the previous call on this device, if any
        int dev_now = 0;
        (void)hipGetDevice(&dev_now);
        std::lock_guard<std::mutex> lock(g_side_mu);
        if (g_side_kept.st && g_side_kept.device == dev_now) { side = std::move(g_side_kept); g_side_kept = DeviceSide(); }
        side.device = dev_now;
    }
    hipStream_t &st = side.st;
    std::vector<hipEvent_t> &ev = side.ev;
    char *&pool = side.pool;
    static const size_t keep_bytes = [] { const char *e = std::getenv("GENPHI_SPARSE_KEEP_MB"); return static_cast<size_t>(e ? std::max(0L, std::atol(e)) : 1024L) << 20; }();
    auto cleanup = [&]() {                                        // on an error: everything goes
        side.destroy();
        dM[0] = dM[1] = dT = d_sval = nullptr; d_blob = nullptr; d_src = nullptr; d_cnt = nullptr;
    };
    auto release = [&]() {                                        // at the end of a call: kept for the next one if the slot is free
        if (side.bytes <= keep_bytes && keep_bytes > 0) {
            std::lock_guard<std::mutex> lock(g_side_mu);
            if (!g_side_kept.st) { g_side_kept = std::move(side); side = DeviceSide(); return; }
        }
        side.destroy();
    };
#define SP_GO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return bail(GENPHI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    // (GENPHI_SPARSE_STALE_CAP: the first sweep's room for entries that outlive their columns -- tests force the second sweep with it)
    int stale_cap = 1 << 16, n_stale = 0;
    if (const char *e = genphi::env_hook("GENPHI_SPARSE_STALE_CAP")) stale_cap = std::max(1, std::atoi(e));
    const bool timed = waves.size() <= 4096;
    size_t max_lds = 0;
    const bool no_fused = genphi::env_hook("GENPHI_SPARSE_NO_FUSED") != nullptr;      // (A/B and tests: the two-kernel form of a wave)
    for (const Wave &w : waves) if (w.n_new > 0 && w.n_old <= 36864) max_lds = std::max(max_lds, static_cast<size_t>((w.n_old + 3) / 4 * 4) * sizeof(float));
    for (int attempt = 0; attempt < 2; ++attempt) {               // (a second sweep only if the list of outliving entries overflowed)
        if (!st) SP_GO(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        {   // ONE allocation for the two matrices, T, the index blob and the outliving-entry lists
            const size_t b_M = al(max_mat * sizeof(float)), b_T = al(max_T * sizeof(float)), b_blob = al(blob_bytes);
            const size_t b_sv = al(static_cast<size_t>(stale_cap) * sizeof(float)), b_src = al(static_cast<size_t>(stale_cap) * sizeof(int2));
            const size_t need = 2 * b_M + b_T + b_blob + b_sv + b_src + 256;
            if (side.bytes < need) {
                if (pool) { (void)hipFree(pool); pool = nullptr; side.bytes = 0; }
                SP_GO(hipMalloc(reinterpret_cast<void **>(&pool), need));
                side.bytes = need;
            }
            dM[0] = reinterpret_cast<float *>(pool); dM[1] = reinterpret_cast<float *>(pool + b_M);
            dT = reinterpret_cast<float *>(pool + 2 * b_M);
            d_blob = pool + 2 * b_M + b_T;
            d_sval = reinterpret_cast<float *>(pool + 2 * b_M + b_T + b_blob);
            d_src = reinterpret_cast<int2 *>(pool + 2 * b_M + b_T + b_blob + b_sv);
            d_cnt = reinterpret_cast<int *>(pool + 2 * b_M + b_T + b_blob + b_sv + b_src);
        }
        mark("stream, allocation");
        SP_GO(hipMemcpyAsync(d_blob, blob.data(), blob_bytes, hipMemcpyHostToDevice, st));
        SP_GO(hipMemsetAsync(d_cnt, 0, sizeof(int), st));
        if (max_lds > 48 * 1024) {
            SP_GO(hipFuncSetAttribute(reinterpret_cast<const void *>(sparse_newnew_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(max_lds)));
            SP_GO(hipFuncSetAttribute(reinterpret_cast<const void *>(sparse_row_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(max_lds)));
        }
        if (timed) {
            while (ev.size() < waves.size() + 1) { hipEvent_t e; SP_GO(hipEventCreate(&e)); ev.push_back(e); }
            SP_GO(hipEventRecord(ev[0], st));
        }
        StaleOut so; so.cnt = d_cnt; so.cap = stale_cap; so.rc = d_src; so.val = d_sval;
        long long ld_cur = pitch_of(0);
        int cur = 0;
        for (size_t wi = 0; wi < waves.size(); ++wi) {
            const Wave &w = waves[wi];
            const int n_next = w.n_surv + w.n_new;
            const long long ldT = pitch_of(w.n_old), ld_next = pitch_of(n_next);
            const int4 *d_par = reinterpret_cast<const int4 *>(d_blob + w.o_par);
            const int *d_keep = reinterpret_cast<const int *>(d_blob + w.o_keep), *d_newpos = reinterpret_cast<const int *>(d_blob + w.o_newpos);
            const int2 *d_meta_new = reinterpret_cast<const int2 *>(d_blob + w.o_meta_new), *d_meta_old = reinterpret_cast<const int2 *>(d_blob + w.o_meta_old);
            const bool fused = !no_fused && w.n_new > 0 && w.n_old <= 36864;
            if (fused) {   // a new row end to end per workgroup (T stays in LDS) + survivors x survivors: ONE launch
                const long long cb = w.n_surv > 0 ? static_cast<long long>(w.n_surv) * ((w.n_surv + 8 * kRowThreads - 1) / (8 * kRowThreads)) : 0;
                const size_t lds = std::max<size_t>(16, static_cast<size_t>((w.n_old + 3) / 4 * 4) * sizeof(float));
                hipLaunchKernelGGL(sparse_row_fused_kernel, dim3(static_cast<unsigned>(w.n_new + cb)), dim3(kRowThreads), lds, st, dM[cur], ld_cur, d_meta_old, w.n_old,
                                   d_par, d_meta_new, w.n_new, d_newpos, dM[cur ^ 1], ld_next, w.n_surv, so, d_keep);
                SP_GO(hipGetLastError());
            } else {   // the new rows against the old members + survivors x survivors: ONE launch (both only read the old matrix)
                const int rows_bx = (w.n_new > 0 && w.n_old > 0) ? (w.n_new + kRowsPerBlock - 1) / kRowsPerBlock : 0;
                const int rows_by = rows_bx ? (w.n_old + 1023) / 1024 : 0;
                const long long cb = w.n_surv > 0 ? static_cast<long long>(w.n_surv) * ((w.n_surv + 2047) / 2048) : 0;
                const long long nb = static_cast<long long>(rows_bx) * rows_by + cb;
                if (nb > 0) {
                    hipLaunchKernelGGL(sparse_rows_compact_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, st, dM[cur], ld_cur, d_meta_old, w.n_old,
                                       d_par, d_meta_new, w.n_new, d_newpos, dT, ldT, dM[cur ^ 1], ld_next, w.n_surv, so.cnt, so.cap, so.rc, so.val,
                                       d_keep, rows_bx, rows_by);
                    SP_GO(hipGetLastError());
                }
            }
            if (w.n_new > 0) {
                if (!fused) {
                    const int lds_floats = w.n_old <= 36864 ? (w.n_old + 3) / 4 * 4 : 0;
                    hipLaunchKernelGGL(sparse_newnew_kernel, dim3(static_cast<unsigned>(w.n_new)), dim3(256), static_cast<size_t>(lds_floats) * sizeof(float), st,
                                       dM[cur], ld_cur, d_meta_old, w.n_old, d_par, d_meta_new, dT, ldT, w.n_new, w.n_old > 0 ? lds_floats : 0,
                                       dM[cur ^ 1], ld_next, w.n_surv, so);
                    SP_GO(hipGetLastError());
                }
                dim3 grid(static_cast<unsigned>((n_next + 63) / 64), static_cast<unsigned>((w.n_new + 63) / 64));
                hipLaunchKernelGGL(sparse_mirror_kernel, grid, dim3(256), 0, st, dM[cur ^ 1], ld_next, w.n_surv, n_next);
                SP_GO(hipGetLastError());
            }
            if (timed) SP_GO(hipEventRecord(ev[wi + 1], st));
            cur ^= 1;
            ld_cur = ld_next;
        }
        mark("upload + enqueue");
        SP_GO(hipMemcpyAsync(&n_stale, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
        SP_GO(hipStreamSynchronize(st));
        mark("sweep done");
        if (n_stale > stale_cap) {                                // more entries outlive their columns than the list holds: once more, sized exactly
            if (attempt == 1 || n_stale > (1 << 28)) { cleanup(); return bail(GENPHI_ERR_ALLOC, "genphi_sparse_phi: too many entries outlive their columns"); }
            stale_cap = n_stale;                                  // (the pool grows at the top of the second attempt)
            continue;
        }
        // ---- results: the proband x proband block, the remembered entries, the timings ------------------------
        const int64_t N = R->n_pro;
        if (!R->S.take(static_cast<size_t>(N * N))) { cleanup(); return bail(GENPHI_ERR_ALLOC, "out of memory"); }
        if (N > 0) {
            // through a pinned staging buffer kept for the life of the process (a 2D copy into pageable memory runs at ~6 GB/s:
            // 2.7 of the 12 ms of a call at 2,000 probands), in four row bands so that the host copy of a band runs under the next one's transfer
            // (capped at 256 MB -- 8,192 probands --: beyond, the plain 2D copy; released by genphi_release_cached)
            std::lock_guard<std::mutex> lock(g_pin_mu);
            const size_t need = static_cast<size_t>(N * N) * sizeof(float);
            if (g_pin_bytes < need && need <= kPinCap) {
                if (g_pin) (void)hipHostFree(g_pin);
                g_pin = nullptr; g_pin_bytes = 0;
                if (hipHostMalloc(&g_pin, need, hipHostMallocDefault) == hipSuccess) g_pin_bytes = need; else { (void)hipGetLastError(); g_pin = nullptr; }
            }
            void *pin = g_pin_bytes >= need ? g_pin : nullptr;
            if (pin) {
                constexpr int kBands = 4;
                const size_t ev0 = timed ? waves.size() + 1 : 0;             // (after the waves' timing events)
                while (ev.size() < ev0 + kBands) { hipEvent_t e; SP_GO(hipEventCreate(&e)); ev.push_back(e); }
                hipEvent_t *band_ev = ev.data() + ev0;
                const int64_t rows_band = (N + kBands - 1) / kBands;
                for (int bnd = 0; bnd < kBands; ++bnd) {
                    const int64_t r0 = bnd * rows_band, r1 = std::min<int64_t>(N, r0 + rows_band);
                    if (r1 > r0)
                        SP_GO(hipMemcpy2DAsync(static_cast<char *>(pin) + r0 * N * sizeof(float), N * sizeof(float), dM[cur] + r0 * ld_cur, ld_cur * sizeof(float),
                                               N * sizeof(float), r1 - r0, hipMemcpyDeviceToHost, st));
                    SP_GO(hipEventRecord(band_ev[bnd], st));
                }
                for (int bnd = 0; bnd < kBands; ++bnd) {
                    const int64_t r0 = bnd * rows_band, r1 = std::min<int64_t>(N, r0 + rows_band);
                    SP_GO(hipEventSynchronize(band_ev[bnd]));
                    if (r1 > r0) std::memcpy(R->S.p + r0 * N, static_cast<char *>(pin) + r0 * N * sizeof(float), static_cast<size_t>(r1 - r0) * N * sizeof(float));
                }
            } else {
                SP_GO(hipMemcpy2D(R->S.p, N * sizeof(float), dM[cur], ld_cur * sizeof(float), N * sizeof(float), N, hipMemcpyDeviceToHost));
            }
        }
        if (n_stale) {
            std::vector<int2> rc(n_stale);
            std::vector<float> sv(n_stale);
            SP_GO(hipMemcpy(rc.data(), d_src, static_cast<size_t>(n_stale) * sizeof(int2), hipMemcpyDeviceToHost));
            SP_GO(hipMemcpy(sv.data(), d_sval, static_cast<size_t>(n_stale) * sizeof(float), hipMemcpyDeviceToHost));
            // the append order depends on the scheduling of the workgroups: sort for a reproducible list
            std::vector<int> o(n_stale);
            for (int k = 0; k < n_stale; ++k) o[k] = k;
            std::sort(o.begin(), o.end(), [&](int x, int y) { return rc[x].x != rc[y].x ? rc[x].x < rc[y].x : rc[x].y < rc[y].y; });
            for (int k = 0; k < n_stale; ++k) { R->stale_row_rank.push_back(rc[o[k]].x); R->stale_col_rank.push_back(rc[o[k]].y); R->stale_val.push_back(sv[o[k]]); }
        }
        if (timed) {
            float ms = 0.f;
            SP_GO(hipEventElapsedTime(&ms, ev[0], ev[waves.size()]));
            R->sweep_ms = ms;
            R->wave_ms.resize(waves.size());
            for (size_t wi = 0; wi < waves.size(); ++wi) SP_GO(hipEventElapsedTime(&R->wave_ms[wi], ev[wi], ev[wi + 1]));
        }
        mark("results to host");
        release();
        mark("device side kept / freed");
        break;
    }
#undef SP_GO
    const int64_t N = R->n_pro;
    R->rank.resize(N); R->proc.resize(N); R->slot.resize(N);
    for (int64_t k = 0; k < N; ++k) {
        const int u = iso_of[at.find(R->ids[k])];
        R->rank[k] = u + 1; R->proc[k] = proc[u]; R->slot[k] = slot_of[u];
    }
    *out = R;
    return GENPHI_OK;
}

int genphi_sparse_phi(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                      const int64_t *pro_ids, int32_t device, genphi_sparse **out)
{
    return sparse_impl(n_ind, ind, father, mother, n_pro, pro_ids, device, out, nullptr);
}

/* Host only (no GPU needed): the schedule genphi_sparse_phi would follow -- the processing order of src/compute.jl:336-345 / :431-439
 * (IDs), for each the processing index at which it is dropped from the live set (:401-430; -1 = a proband, kept), and its wave (depth).
 * *n_out = the number of individuals processed (probands and their ancestors); at most `cap` entries are filled. */
int genphi_sparse_schedule(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                           const int64_t *pro_ids, int64_t cap, int64_t *order_ids, int64_t *retire_at, int32_t *wave, int64_t *n_out)
{
    if (!n_out) return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_schedule: n_out is NULL");
    *n_out = 0;
    ScheduleOut so;
    so.cap = cap; so.ids = order_ids; so.retire_at = retire_at; so.wave = wave;
    genphi_sparse *unused = nullptr;
    const int rc = sparse_impl(n_ind, ind, father, mother, n_pro, pro_ids, -1, &unused, &so);
    if (rc == GENPHI_OK) *n_out = so.n;
    return rc;
}

/* Measurement of the sweep that built the handle: waves (depths), device time of the whole sweep and of every
 * wave (HIP events on the sweep's stream), algorithmic bytes 4 (n_old^2 + n_next^2) per wave (the old active
 * matrix read once, the next one written once), the largest active set. */
int genphi_sparse_stats(const genphi_sparse *h, int32_t *n_waves, double *sweep_ms, double *algorithmic_bytes, int64_t *max_active,
                        float *wave_ms, double *wave_bytes, int32_t cap)
{
    if (!h) return genphi_set_error(GENPHI_ERR_ARG, "sparse handle is NULL");
    const int32_t nw = static_cast<int32_t>(h->wave_bytes.size());
    if (n_waves) *n_waves = nw;
    if (sweep_ms) *sweep_ms = h->sweep_ms;
    if (algorithmic_bytes) *algorithmic_bytes = h->algorithmic_bytes;
    if (max_active) *max_active = h->max_active;
    for (int32_t k = 0; k < nw && k < cap; ++k) {
        if (wave_ms) wave_ms[k] = k < static_cast<int32_t>(h->wave_ms.size()) ? h->wave_ms[k] : 0.f;
        if (wave_bytes) wave_bytes[k] = h->wave_bytes[k];
    }
    return GENPHI_OK;
}

int genphi_sparse_info(const genphi_sparse *h, int64_t *n_rows, int64_t *n_stored, double *sum_all, double *sum_diag)
{
    if (!h) return genphi_set_error(GENPHI_ERR_ARG, "sparse handle is NULL");
    const int64_t N = h->n_pro;
    int64_t nz = 0;
    double tot = 0.0, dg = 0.0;
    for (int64_t a = 0; a < N; ++a) {
        const float self = h->S[h->slot[a] * N + h->slot[a]];
        nz += 1; tot += self; dg += self;                        // the self entry is always stored (>= 1/2)
        for (int64_t b = a + 1; b < N; ++b) {
            const float v = h->S[h->slot[a] * N + h->slot[b]];
            if (v > 0.f) { nz += 1; tot += v; }                  // one entry per pair, under the (earlier, later) key
        }
    }
    for (float v : h->stale_val) if (v > 0.f) { nz += 1; tot += v; }
    if (n_rows) *n_rows = N;
    if (n_stored) *n_stored = nz;
    if (sum_all) *sum_all = tot;
    if (sum_diag) *sum_diag = dg;
    return GENPHI_OK;
}

int genphi_sparse_get(const genphi_sparse *h, int64_t n, const int64_t *id1, const int64_t *id2, double *out)
{
    if (!h || n < 0 || (n > 0 && (!id1 || !id2 || !out))) return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_get: bad argument");
    const int64_t N = h->n_pro;
    for (int64_t k = 0; k < n; ++k) {
        auto a = h->pos.find(id1[k]), b = h->pos.find(id2[k]);
        if (a == h->pos.end() || b == h->pos.end())
            return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: " + std::to_string(a == h->pos.end() ? id1[k] : id2[k]) + " is not a proband of this KinshipMatrix");
        const int pa = a->second, pb = b->second;
        float v = h->S[h->slot[pa] * N + h->slot[pb]];
        // getindex looks under (smaller rank, larger rank); the value sits under (earlier, later)
        if (pa != pb && ((h->proc[pa] < h->proc[pb]) != (h->rank[pa] < h->rank[pb]))) v = 0.f;
        out[k] = static_cast<double>(v);
    }
    return GENPHI_OK;
}

int64_t genphi_sparse_entries(const genphi_sparse *h, int64_t cap, int64_t *row_rank, int64_t *col_rank, float *val)
{
    if (!h) return -1;
    const int64_t N = h->n_pro;
    int64_t k = 0;
    auto put = [&](int64_t r, int64_t c, float v) { if (k < cap && row_rank && col_rank && val) { row_rank[k] = r; col_rank[k] = c; val[k] = v; } ++k; };
    for (int64_t a = 0; a < N; ++a)
        for (int64_t b = 0; b < N; ++b) {
            const float v = h->S[h->slot[a] * N + h->slot[b]];
            if (a == b) put(h->rank[a], h->rank[a], v);
            else if (h->proc[a] < h->proc[b] && v > 0.f) put(h->rank[a], h->rank[b], v);
        }
    for (size_t s = 0; s < h->stale_val.size(); ++s)
        if (h->stale_val[s] > 0.f) put(h->stale_row_rank[s], h->stale_col_rank[s], h->stale_val[s]);
    return k;
}

void genphi_sparse_destroy(genphi_sparse *h) { delete h; }

}  // extern "C"
