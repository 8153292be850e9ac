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
// up under (smaller rank, larger rank): whenever two individuals of equal depth leave the queue in
// the opposite order of their ranks, their kinship is stored where no lookup finds it -- it reads
// as 0 from then on (tests/oracle restate that behaviour; this file reproduces it, it does not
// "fix" it).
//
// Design here (not a translation): the queue order is depth-sorted (a child is enqueued while its
// deepest parent is processed) WHATEVER the ranks are (with genealogy(...; sort=false) the rank is the
// file position and need not follow the depth), so all individuals of one depth -- a WAVE -- only
// need kinships with strictly older individuals and with each other through those:
//   T[i][q]  = RN32(L(f_i, q)/2 + L(m_i, q)/2)      new i x every live older q   (one kernel)
//   S[i][j]  = RN32(L'(j, f_i)/2 + L'(j, m_i)/2)    new i x new j, j processed before i, where
//              L'(j, p) = T[j][p] if rank(p) < rank(j) (p left the queue before j), else 0
//   S[i][i]  = RN32(1/2 + L(f_i, m_i)/2)
// with L(a, b) = the stored value if the (earlier, later) key equals the (smaller rank, larger rank)
// key, else 0 (src/compute.jl:366-390 looks up phi[min rank][max rank], :392-394 stores under
// [rank of the live one][rank of the new one]).  The live set is a dense matrix in HBM ("active matrix"), compacted after every wave
// (retired parents leave), exactly like the cuts of the dense path with other membership rules.
// The host simulates the queue once (integers only) to get the processing order, the waves and the
// wave after which every individual retires; all kinship arithmetic runs in the kernels below.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/genphi.h"

int genphi_set_error(int code, const std::string &msg);      // genphi_hip.hip

namespace {

// "half" as the reference computes it: Float32 / 2 in Float32 (exact unless the result is subnormal)
__device__ __forceinline__ float half32(float v) { return v / 2.0f; }

// stored value of the pair of active slots (a, b), as a lookup sees it (see L above); M is the
// active matrix (pitch ld), meta[s] = (rank, processing index) of slot s; none = zero row / column
// meta = (rank, processing index): does a lookup find the kinship of two distinct individuals?
__device__ __forceinline__ bool key_found(int2 ma, int2 mb) { return (ma.y < mb.y) == (ma.x < mb.x); }

__device__ __forceinline__ float lookup(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int a, int b, int none)
{
    if (a == none || b == none) return 0.f;
    if (a != b && !key_found(meta[a], meta[b])) return 0.f;     // stored under a key no lookup uses
    return M[(long long)a * ld + b];
}

// T[i][q] for the wave's new individuals i and every live slot q (q == n_old: the zero column)
__global__ void __launch_bounds__(256)
sparse_new_old_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old,
                      const int2 *__restrict__ par, float *__restrict__ T, long long ldT)
{
    const int i = blockIdx.x;
    const int2 p = par[i];                                       // (father slot, mother slot), n_old = none
    for (int q = blockIdx.y * 256 + threadIdx.x; q < ldT; q += gridDim.y * 256) {
        float v = 0.f;
        if (q < n_old) {
            const double c = 0.0 + static_cast<double>(half32(lookup(M, ld, meta, p.x, q, n_old))) +
                             static_cast<double>(half32(lookup(M, ld, meta, p.y, q, n_old)));
            v = static_cast<float>(c);
        }
        T[(long long)i * ldT + q] = v;
    }
}

// the next active matrix: rows / columns = [survivors (old slots keep[s])..., new individuals...];
// row and column n_next and the pitch padding are zero
__global__ void __launch_bounds__(256)
sparse_assemble_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ meta, int n_old,
                       const int2 *__restrict__ par, const float *__restrict__ T, long long ldT,
                       const int *__restrict__ keep, int n_surv, int n_new, const int2 *__restrict__ meta_next,
                       float *__restrict__ out, long long ld_out)
{
    const int r = blockIdx.x;                                    // 0 .. n_surv + n_new (the last one is the zero row)
    const int n_next = n_surv + n_new;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < ld_out; c += gridDim.y * 256) {
        float v = 0.f;
        if (r < n_next && c < n_next) {
            if (r < n_surv && c < n_surv) {
                v = M[(long long)keep[r] * ld + keep[c]];
            } else if (r >= n_surv && c < n_surv) {
                v = T[(long long)(r - n_surv) * ldT + keep[c]];
            } else if (r < n_surv) {
                v = T[(long long)(c - n_surv) * ldT + keep[r]];
            } else {
                const int a = r - n_surv, b = c - n_surv;
                if (a == b) {
                    const int2 p = par[a];
                    double cf = 0.5;
                    if (p.x != n_old && p.y != n_old) cf += static_cast<double>(half32(lookup(M, ld, meta, p.x, p.y, n_old)));
                    v = static_cast<float>(cf);
                } else {
                    const int i = max(a, b), j = min(a, b);      // j left the queue before i
                    const int2 p = par[i];
                    // the parents left the queue before j (an earlier wave): T[j][parent] sits under the key
                    // (rank parent, rank j), which the lookup (smaller rank, larger rank) finds only when
                    // rank parent < rank j -- always true for depth-sorted ranks, not with sort = false
                    const int2 mj = meta_next[n_surv + j];
                    const float tf = (p.x == n_old || !key_found(meta[p.x], mj)) ? 0.f : T[(long long)j * ldT + p.x];
                    const float tm = (p.y == n_old || !key_found(meta[p.y], mj)) ? 0.f : T[(long long)j * ldT + p.y];
                    v = static_cast<float>(0.0 + static_cast<double>(half32(tf)) + static_cast<double>(half32(tm)));
                }
            }
        }
        out[(long long)r * ld_out + c] = v;
    }
}

__global__ void sparse_gather_kernel(const float *__restrict__ M, long long ld, const int2 *__restrict__ rc, int n, float *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = M[(long long)rc[k].x * ld + rc[k].y];
}

long long pitch_of(long long n) { return ((n + 1) + 63) / 64 * 64; }

struct Wave {
    int n_old = 0, n_new = 0, n_surv = 0;
    std::vector<int2> par;        // per new individual: (father slot, mother slot) in the old active list
    std::vector<int> keep;        // old slots that survive the wave, ascending
    std::vector<int2> meta_next;  // (rank, processing index) of the next active list
    std::vector<int2> stale;      // (row slot, column slot) in the NEXT active matrix of entries to remember
};

}  // namespace

struct genphi_sparse {
    int64_t n_pro = 0;                        // distinct probands
    std::vector<int64_t> ids;                 // proband IDs, first-occurrence order
    std::vector<int> rank, proc;              // of each proband (rank in the pruned pedigree, processing index)
    std::vector<int> slot;                    // row / column of each proband in S
    std::vector<float> S;                     // n_pro x n_pro: stored value of every pair of probands
    std::vector<int> stale_row_rank, stale_col_rank;   // entries that survive in a proband's dictionary
    std::vector<float> stale_val;
    std::unordered_map<int64_t, int> pos;     // ID -> index into ids
};

extern "C" {

int genphi_sparse_phi(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                      const int64_t *pro_ids, int32_t device, genphi_sparse **out)
{
    if (!out) return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_phi: out is NULL");
    *out = nullptr;
    if (n_ind < 0 || n_pro < 0 || (n_ind > 0 && (!ind || !father || !mother)) || (n_pro > 0 && !pro_ids))
        return genphi_set_error(GENPHI_ERR_ARG, "genphi_sparse_phi: null or negative argument");
    genphi_sparse *R = new (std::nothrow) genphi_sparse();
    if (!R) return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory");
    auto bail = [&](int code, const std::string &msg) { delete R; return genphi_set_error(code, msg); };

    // ---- the pruned pedigree: probands and their ancestors, in pedigree order (branching) ----------
    std::unordered_map<int64_t, int> at;
    at.reserve(static_cast<size_t>(n_ind) * 2);
    std::vector<int> fa(n_ind, -1), mo(n_ind, -1);
    for (int64_t i = 0; i < n_ind; ++i) {
        if (father[i] != 0) { auto it = at.find(father[i]); if (it == at.end()) return bail(GENPHI_ERR_ORDER, "parent listed after its child or unknown"); fa[i] = it->second; }
        if (mother[i] != 0) { auto it = at.find(mother[i]); if (it == at.end()) return bail(GENPHI_ERR_ORDER, "parent listed after its child or unknown"); mo[i] = it->second; }
        if (!at.emplace(ind[i], static_cast<int>(i)).second) return bail(GENPHI_ERR_DUPLICATE_ID, "duplicate individual ID " + std::to_string(ind[i]));
    }
    std::vector<char> keep(n_ind, 0), is_pro(n_ind, 0);
    for (int64_t k = 0; k < n_pro; ++k) {
        auto it = at.find(pro_ids[k]);
        if (it == at.end()) return bail(GENPHI_ERR_UNKNOWN_ID, "KeyError: proband " + std::to_string(pro_ids[k]) + " not found");
        keep[it->second] = 1;
        if (!is_pro[it->second]) { is_pro[it->second] = 1; R->pos.emplace(pro_ids[k], static_cast<int>(R->ids.size())); R->ids.push_back(pro_ids[k]); }
    }
    for (int64_t x = n_ind - 1; x >= 0; --x)                     // parents precede children: one reverse sweep
        if (keep[x]) { if (fa[x] >= 0) keep[fa[x]] = 1; if (mo[x] >= 0) keep[mo[x]] = 1; }
    std::vector<int> iso_of(n_ind, -1), orig;                    // pruned index <-> original index
    for (int64_t x = 0; x < n_ind; ++x) if (keep[x]) { iso_of[x] = static_cast<int>(orig.size()); orig.push_back(static_cast<int>(x)); }
    const int m = static_cast<int>(orig.size());                 // rank of pruned index u is u + 1
    R->n_pro = static_cast<int64_t>(R->ids.size());
    if (m == 0) { *out = R; return GENPHI_OK; }
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
    std::vector<int> proc(m, -1), retire(m, INT32_MAX), left(m, 0), order;
    order.reserve(m);
    {
        std::deque<int> queue;
        std::vector<std::pair<int64_t, int>> founders;
        for (int u = 0; u < m; ++u) if (pf[u] < 0 && pm[u] < 0) founders.emplace_back(ind[orig[u]], u);
        std::sort(founders.begin(), founders.end());             // founder(): IDs ascending
        for (auto &e : founders) queue.push_back(e.second);
        while (!queue.empty()) {
            const int u = queue.front(); queue.pop_front();
            if (proc[u] >= 0) continue;                          // (a child listed twice by one parent)
            proc[u] = static_cast<int>(order.size());
            order.push_back(u);
            left[u] = nchild[u];
            for (int par : {pf[u], pm[u]})
                if (par >= 0 && !pro_flag[par] && --left[par] == 0) retire[par] = proc[u];
            for (int k = cstart[u]; k < cstart[u + 1]; ++k) {
                const int c = clist[k];
                if (pf[c] >= 0 && pm[c] >= 0) { if (proc[pf[c]] >= 0 && proc[pm[c]] >= 0) queue.push_back(c); }
                else queue.push_back(c);
            }
        }
    }
    if (static_cast<int>(order.size()) != m) return bail(GENPHI_ERR_ARG, "internal: the queue did not reach every individual");
    for (int k = 1; k < m; ++k)
        if (depth[order[k]] < depth[order[k - 1]]) return bail(GENPHI_ERR_ARG, "internal: processing order is not depth-sorted");

    // ---- waves: active lists, parents' slots, survivors, entries to remember ------------------------
    std::vector<Wave> waves;
    std::vector<int> active;                                     // pruned indices, slot order
    std::vector<int> slot_of(m, -1);
    std::vector<int> live_pro_ranks;                             // pruned indices (= rank - 1) of the probands processed so far, ascending
    size_t max_mat = 64, max_T = 64, max_meta = 1, max_par = 1, n_stale = 0;
    for (int b = 0; b < m;) {
        int e = b;
        while (e < m && depth[order[e]] == depth[order[b]]) ++e;
        Wave w;
        w.n_old = static_cast<int>(active.size());
        w.n_new = e - b;
        const int last_proc = e - 1;
        for (int k = b; k < e; ++k) {
            const int u = order[k];
            w.par.push_back(make_int2(pf[u] >= 0 ? slot_of[pf[u]] : w.n_old, pm[u] >= 0 ? slot_of[pm[u]] : w.n_old));
        }
        std::vector<int> next;
        for (int s = 0; s < w.n_old; ++s)
            if (retire[active[s]] > last_proc) { w.keep.push_back(s); next.push_back(active[s]); }
        w.n_surv = static_cast<int>(next.size());
        for (int k = b; k < e; ++k) next.push_back(order[k]);
        for (size_t s = 0; s < next.size(); ++s) { slot_of[next[s]] = static_cast<int>(s); w.meta_next.push_back(make_int2(next[s] + 1, proc[next[s]])); }
        // Entries that outlive their column (src/compute.jl:401-430): when a non-proband x retires, phi[rank j][rank x]
        // is deleted only for live j with rank j < rank x.  The entry exists when j left the queue before x, so every
        // proband j with proc(j) < proc(x) and rank(j) > rank(x) keeps it for good (`show` counts it, phiMean sums it) --
        // j of an earlier wave included (only possible when ranks are not depth-sorted: sort = false).  Gathered right
        // after the wave that processes x, while x's row is in the active matrix.  live_pro_ranks: ranks of the probands
        // processed so far, ascending.
        for (int kx = b; kx < e; ++kx) {
            const int x = order[kx];
            if (pro_flag[x]) { live_pro_ranks.insert(std::upper_bound(live_pro_ranks.begin(), live_pro_ranks.end(), x), x); continue; }
            for (auto it = std::upper_bound(live_pro_ranks.begin(), live_pro_ranks.end(), x); it != live_pro_ranks.end(); ++it)
                w.stale.push_back(make_int2(slot_of[*it], slot_of[x]));
            if (n_stale + w.stale.size() > (size_t(1) << 28))
                return bail(GENPHI_ERR_ALLOC, "genphi_sparse_phi: more than 2^28 entries outlive their column (ranks far from depth order)");
        }
        n_stale += w.stale.size();
        max_mat = std::max(max_mat, static_cast<size_t>((next.size() + 1) * pitch_of(static_cast<long long>(next.size()))));
        max_T = std::max(max_T, static_cast<size_t>(w.n_new) * static_cast<size_t>(pitch_of(w.n_old)));
        max_meta = std::max(max_meta, next.size() + 1);
        max_par = std::max(max_par, w.par.size());
        active.swap(next);
        waves.push_back(std::move(w));
        b = e;
    }
    // the final active list is exactly the probands
    if (static_cast<int64_t>(active.size()) != R->n_pro) return bail(GENPHI_ERR_ARG, "internal: final active set is not the proband set");

    // ---- the device sweep ------------------------------------------------------------------------------
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return bail(GENPHI_ERR_DEVICE, "no HIP device available: gen.sparse_phi has no CPU fallback");
    if (device >= 0) { if (device >= ndev) return bail(GENPHI_ERR_DEVICE, "device ordinal out of range"); (void)hipSetDevice(device); }
    float *dM[2] = {nullptr, nullptr}, *dT = nullptr, *d_stale = nullptr;
    int2 *d_meta[2] = {nullptr, nullptr}, *d_par = nullptr, *d_rc = nullptr;
    int *d_keep = nullptr;
    hipStream_t st = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(dM[0]); (void)hipFree(dM[1]); (void)hipFree(dT); (void)hipFree(d_stale); (void)hipFree(d_meta[0]); (void)hipFree(d_meta[1]);
        (void)hipFree(d_par); (void)hipFree(d_rc); (void)hipFree(d_keep);
        if (st) (void)hipStreamDestroy(st);
    };
#define SP_GO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return bail(GENPHI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    SP_GO(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        SP_GO(hipMalloc(reinterpret_cast<void **>(&dM[k]), max_mat * sizeof(float)));
        SP_GO(hipMalloc(reinterpret_cast<void **>(&d_meta[k]), max_meta * sizeof(int2)));
    }
    SP_GO(hipMalloc(reinterpret_cast<void **>(&dT), max_T * sizeof(float)));
    SP_GO(hipMalloc(reinterpret_cast<void **>(&d_par), max_par * sizeof(int2)));
    SP_GO(hipMalloc(reinterpret_cast<void **>(&d_keep), max_meta * sizeof(int)));
    SP_GO(hipMalloc(reinterpret_cast<void **>(&d_stale), std::max<size_t>(n_stale, 1) * sizeof(float)));
    SP_GO(hipMalloc(reinterpret_cast<void **>(&d_rc), std::max<size_t>(n_stale, 1) * sizeof(int2)));
    SP_GO(hipMemsetAsync(dM[0], 0, max_mat * sizeof(float), st));           // the empty active set: a zero "none" row
    size_t stale_done = 0;
    long long ld_cur = pitch_of(0);
    int cur = 0;
    for (const Wave &w : waves) {
        const int n_next = w.n_surv + w.n_new;
        const long long ldT = pitch_of(w.n_old), ld_next = pitch_of(n_next);
        SP_GO(hipMemcpyAsync(d_par, w.par.data(), w.par.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        if (!w.keep.empty()) SP_GO(hipMemcpyAsync(d_keep, w.keep.data(), w.keep.size() * sizeof(int), hipMemcpyHostToDevice, st));
        SP_GO(hipMemcpyAsync(d_meta[cur ^ 1], w.meta_next.data(), w.meta_next.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        {
            dim3 grid(static_cast<unsigned>(w.n_new), static_cast<unsigned>(std::min<long long>((ldT + 255) / 256, 64)));
            hipLaunchKernelGGL(sparse_new_old_kernel, grid, dim3(256), 0, st, dM[cur], ld_cur, d_meta[cur], w.n_old, d_par, dT, ldT);
            SP_GO(hipGetLastError());
        }
        {
            dim3 grid(static_cast<unsigned>(n_next + 1), static_cast<unsigned>(std::min<long long>((ld_next + 255) / 256, 64)));
            hipLaunchKernelGGL(sparse_assemble_kernel, grid, dim3(256), 0, st, dM[cur], ld_cur, d_meta[cur], w.n_old, d_par, dT, ldT,
                               d_keep, w.n_surv, w.n_new, d_meta[cur ^ 1], dM[cur ^ 1], ld_next);
            SP_GO(hipGetLastError());
        }
        if (!w.stale.empty()) {
            SP_GO(hipMemcpyAsync(d_rc + stale_done, w.stale.data(), w.stale.size() * sizeof(int2), hipMemcpyHostToDevice, st));
            const int n = static_cast<int>(w.stale.size());
            hipLaunchKernelGGL(sparse_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dM[cur ^ 1], ld_next, d_rc + stale_done, n,
                               d_stale + stale_done);
            SP_GO(hipGetLastError());
            stale_done += w.stale.size();
        }
        SP_GO(hipStreamSynchronize(st));                         // the wave's host arrays are reused by the next one
        cur ^= 1;
        ld_cur = ld_next;
    }
    // ---- results: the proband x proband block and the remembered entries ---------------------------------
    const int64_t N = R->n_pro;
    R->S.resize(static_cast<size_t>(N * N));
    if (N > 0)
        SP_GO(hipMemcpy2D(R->S.data(), N * sizeof(float), dM[cur], ld_cur * sizeof(float), N * sizeof(float), N, hipMemcpyDeviceToHost));
    R->stale_val.resize(n_stale);
    if (n_stale) SP_GO(hipMemcpy(R->stale_val.data(), d_stale, n_stale * sizeof(float), hipMemcpyDeviceToHost));
    cleanup();
#undef SP_GO
    R->rank.resize(N); R->proc.resize(N); R->slot.resize(N);
    for (int64_t k = 0; k < N; ++k) {
        const int u = iso_of[at[R->ids[k]]];
        R->rank[k] = u + 1; R->proc[k] = proc[u]; R->slot[k] = slot_of[u];
    }
    {   // ranks of the remembered entries, in the order they were gathered: replay the active lists
        std::vector<int> active2;
        for (int b = 0, wi = 0; b < m; ++wi) {
            int e = b;
            while (e < m && depth[order[e]] == depth[order[b]]) ++e;
            std::vector<int> next;
            for (int s : waves[wi].keep) next.push_back(active2[s]);
            for (int q = b; q < e; ++q) next.push_back(order[q]);
            for (const int2 &rc : waves[wi].stale) { R->stale_row_rank.push_back(next[rc.x] + 1); R->stale_col_rank.push_back(next[rc.y] + 1); }
            active2.swap(next);
            b = e;
        }
    }
    *out = R;
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
