// panel_phi.hip -- gen.phi with STORAGE-SHARDED level matrices: the multi-GPU path for pedigrees whose
// level matrices do not fit one GPU ("4 (n_k^2 + n_{k+1}^2) exceeds one GPU's HBM";
// north star: "founder blocks broadcast over RCCL/xGMI only when N exceeds one GPU's 288 GB").
//
// Reference: the level loop of phi(), src/compute.jl:269-303, which keeps two dense level matrices
// alive (:291, :301) -- the memory model that stops scaling at one device.
//
// Partition: every member of every cut has an OWNER rank for as long as it stays in the cuts
// (probands: contiguous blocks of the proband order, so a rank ends up with a row block of the
// result; everybody else: round-robin in rank order).  Rank r stores, of each level matrix, the
// COLUMN PANEL of its members: all n_k + 1 rows x its columns -- 1/W of the matrix.
// A level step on rank r needs Psi_k[src(i)][src(j)] for every row i and every LOCAL column j:
//   rows    every source row is local (panels hold all rows);
//   columns a dragged column's source is itself: local.  A new column (father, mother) may need
//           columns owned by other ranks: THE EXCHANGE STEP -- before the step every rank packs the
//           columns its peers asked for (pack kernel: strided column -> contiguous), the ranks run
//           one all-to-all (RCCL over xGMI through torch.distributed in the host driver; this library
//           only fills / consumes device buffers, no collective is issued from here), and unpack
//           appends the received columns to the local panel as "extension" columns.
// Exchange volume per step: 2 n_new n_k floats in total (two parent columns per new member), against
// 4 (n_k^2 + n_{k+1}^2) / W bytes of matrix traffic per rank; a replicated level (what fits one GPU
// gets) needs no exchange at all, which is why this path is only taken when the matrices do not fit
// (or when forced: GENPHI_FORCE_EXCHANGE in bench.py, tests).
// Kernels: a panel holds whole ROWS of its columns, so a level step is the FULL / SPLIT row kernels of
// genphi_hip.hip (one or two source rows of the rank's extended panel staged in LDS, gathers at the panel
// columns of the local columns' sources, certified grouping-free bodies) run over the rank's local columns:
// launch_panel_level (panel_launch.h).  Panel rows too long for LDS (> 36,864 floats) and GENPHI_PANEL_NAIVE
// fall back to one thread per (row, local column) with four global gathers (panel_level_kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/genphi.h"
#include "panel_launch.h"
#include "planner.h"

int genphi_set_error(int code, const std::string &msg);      // genphi_hip.hip
extern "C" const char *genphi_last_error(void);

namespace {

constexpr int kOrdMask = 0x7fffffff;

__device__ __forceinline__ float combine4(float a, float b, float c, float d, bool i_hi, double scale)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;           // grouping of the reference's recursion
    const double s = (static_cast<double>(a) + static_cast<double>(x)) + (static_cast<double>(y) + static_cast<double>(d));
    return static_cast<float>(s * scale);
}

// out[i][jl] for every row i of the new cut and every local column jl; psi = the extended local
// panel of the previous level (pitch ldp; row n_prev and column `zcol` are zero)
__global__ void __launch_bounds__(256)
panel_level_kernel(const float *__restrict__ psi, long long ldp, int n_prev, float *__restrict__ out, long long ldo, int n,
                   const int *__restrict__ srcA, const int *__restrict__ srcB, const int *__restrict__ ord,
                   const int4 *__restrict__ col /* (A ext-local, B ext-local, member position, ord word) */, int n_cols)
{
    const int i = blockIdx.x;                                  // 0 .. n (row n: the zero row of the new level)
    for (long long jl = (long long)blockIdx.y * 256 + threadIdx.x; jl < ldo; jl += (long long)gridDim.y * 256) {
        float v = 0.f;
        if (i < n && jl < n_cols) {
            const int4 cj = col[jl];
            const int Ai = srcA[i], Bi = srcB[i], oi = ord[i];
            const float *rowA = psi + (long long)Ai * ldp;
            const float *rowB = psi + (long long)Bi * ldp;
            if (cj.z == i && oi < 0) {
                // diagonal of a new member: 1/2 + Psi[father][mother]/2; column cj.y of row A is Psi[A_i][B_i]
                // (for a member with a missing parent cj.y is the zero column: 1/2)
                v = static_cast<float>(0.5 + 0.5 * static_cast<double>(rowA[cj.y]));
            } else {
                const double sc = (oi < 0 ? 0.5 : 1.0) * (cj.w < 0 ? 0.5 : 1.0);
                v = combine4(rowA[cj.x], rowA[cj.y], rowB[cj.x], rowB[cj.y], (oi & kOrdMask) > (cj.w & kOrdMask), sc);
            }
        }
        (void)n_prev;
        out[(long long)i * ldo + jl] = v;
    }
}

// Psi_1 = 1/2 I restricted to the local columns (src/compute.jl:271-274)
__global__ void panel_identity_kernel(float *__restrict__ out, long long ldo, int n, const int *__restrict__ member, int n_cols)
{
    const int jl = blockIdx.x * blockDim.x + threadIdx.x;
    if (jl < n_cols && member[jl] < n) out[(long long)member[jl] * ldo + jl] = 0.5f;
}

// send[k][i] = panel[i][cols[k]] for the n_send columns asked for by the peers (i < n_rows)
__global__ void __launch_bounds__(256)
panel_pack_kernel(const float *__restrict__ panel, long long ldp, int n_rows, const int *__restrict__ cols, float *__restrict__ send)
{
    const int k = blockIdx.x;
    const float *src = panel + cols[k];
    float *dst = send + (long long)k * n_rows;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n_rows; i += gridDim.y * 256) dst[i] = src[(long long)i * ldp];
}

// panel[i][col0 + k] = recv[k][i] (the received columns become extension columns)
// (a received value in (0, 2^-27) voids the exactness certificate of the row it lands in: cert[i] = 1)
__global__ void __launch_bounds__(256)
panel_unpack_kernel(float *__restrict__ panel, long long ldp, int n_rows, int col0, int n_recv, const float *__restrict__ recv,
                    int *__restrict__ cert, unsigned cert_thresh)
{
    __shared__ float tile[64][65];
    const int k0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int q = ty; q < 64; q += 4) {
        const int k = k0 + q, i = i0 + tx;
        tile[q][tx] = (k < n_recv && i < n_rows) ? recv[(long long)k * n_rows + i] : 0.f;
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {
        const int i = i0 + q, k = k0 + tx;
        if (i < n_rows && k < n_recv) {
            const float v = tile[tx][q];
            panel[(long long)i * ldp + col0 + k] = v;
            if (cert && __float_as_uint(v) - 1u < cert_thresh) cert[i] = 1;
        }
    }
}

// rows of the result from the final column panel: out[jl][i] = panel[i][jl] (Phi is symmetric)
__global__ void __launch_bounds__(256)
panel_transpose_kernel(const float *__restrict__ panel, long long ldp, int n_rows, int n_cols, float *__restrict__ out, long long ldo)
{
    __shared__ float tile[64][65];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int q = ty; q < 64; q += 4) {
        const int i = i0 + q, j = j0 + tx;
        tile[q][tx] = (i < n_rows && j < n_cols) ? panel[(long long)i * ldp + j] : 0.f;
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {
        const int j = j0 + q, i = i0 + tx;
        if (j < n_cols && i < n_rows) out[(long long)j * ldo + i] = tile[tx][q];
    }
}

long long pitch(long long n) { return (n + 1 + 63) / 64 * 64; }

struct PanelStep {
    // exchange before the step: columns of the previous panel to send, grouped by destination rank
    std::vector<int> send_cols;                // local column indices (of the previous cut's panel)
    std::vector<int64_t> send_count, recv_count;   // per peer, in columns
    int n_ext = 0;                             // received columns (appended after the own columns)
    // the step itself
    std::vector<int4> col;                     // per local column of the new cut
    int n_cols = 0;
    // row-kernel form of the step (launch_panel_level)
    int mode = 2;                              // 0 FULL, 1 SPLIT, 2 per-entry kernel
    int src_width = 0;                         // floats of a source panel row incl. its zero column
    std::vector<unsigned> pk_col;              // panel column of A | of B << 16, per local column (+ padding)
    std::vector<int> ord_col;                  // rank word per local column (+ padding)
    std::vector<int> diag_col;                 // per row of the cut: its local column or -1
    std::vector<int> work;                     // rows sorted by (A source, B source)
    genphi::WalkLists walk;                    // SPLIT: the hub walk of the cut's rows
    std::vector<int2> pdesc;
    int n_segs = 0, n_runs = 0;
};

struct DevPanelStep {
    unsigned *pk_col = nullptr;
    int *ord_col = nullptr, *diag_col = nullptr, *work = nullptr;
    int4 *desc = nullptr, *seg = nullptr;
    int4 *run = nullptr;
    int2 *pdesc = nullptr;
};

}  // namespace

struct genphi_panel {
    genphi::Plan plan;
    int rank = 0, world = 1;
    std::vector<int> own_cols_per_cut;         // local columns per cut
    std::vector<std::vector<int>> member;      // per cut: member position of each local column
    std::vector<PanelStep> steps;              // L-1
    int64_t row_begin = 0, n_rows_res = 0;     // this rank's row block of the result (proband order)
    // device
    bool on_device = false;
    int device = -1;
    int fail_alloc_at = 0, alloc_count = 0;    // GENPHI_TEST_FAIL_ALLOC (read once in genphi_panel_create): the k-th device allocation fails
    hipStream_t stream = nullptr;
    float *panel[2] = {nullptr, nullptr};
    size_t panel_floats[2] = {0, 0};
    float *result = nullptr;
    std::vector<int *> d_srcA, d_srcB, d_ord, d_send_cols, d_member;
    std::vector<int4 *> d_col;
    std::vector<DevPanelStep> d_step;
    int *d_cert[2] = {nullptr, nullptr};       // exactness certificates of the rows of panel[0] / panel[1]
    int *d_counters = nullptr, *d_glist = nullptr;
    int glist_cap = 0, n_cus = 256;
    const void *tuning = nullptr;              // environment hooks (panel_tuning_create)
    bool naive = false;                        // GENPHI_PANEL_NAIVE: per-entry kernel on every step (A/B, tests)
    std::vector<hipEvent_t> ev;                // two per level step: around the step's kernels (unpack + level) of the last sweep
    hipEvent_t ev_x[2] = {nullptr, nullptr};   // ordering with the caller's stream: [0] packed columns complete, [1] received columns complete
    std::vector<char> ev_pending;              // step_ms[k] not read back from its events yet
    std::vector<float> step_ms;                // device time of every step's kernels (unpack + level) in the last sweep
    int cur = 0;                               // panel[cur] holds the level of the last step computed
};

static void panel_free_device(genphi_panel *p)
{
    if (!p->on_device) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    auto rel = [](auto *&q) { if (q) (void)hipFree(q); q = nullptr; };
    rel(p->panel[0]); rel(p->panel[1]); rel(p->result);
    rel(p->d_cert[0]); rel(p->d_cert[1]); rel(p->d_counters); rel(p->d_glist);
    for (hipEvent_t &e : p->ev) if (e) (void)hipEventDestroy(e);
    p->ev.clear();
    for (hipEvent_t &e : p->ev_x) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (DevPanelStep &d : p->d_step) { rel(d.pk_col); rel(d.ord_col); rel(d.diag_col); rel(d.work); rel(d.desc); rel(d.seg); rel(d.run); rel(d.pdesc); }
    p->d_step.clear();
    p->panel_floats[0] = p->panel_floats[1] = 0;
    for (auto &v : {&p->d_srcA, &p->d_srcB, &p->d_ord, &p->d_send_cols, &p->d_member}) { for (int *&q : *v) rel(q); v->clear(); }
    for (int4 *&q : p->d_col) rel(q);
    p->d_col.clear();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    p->stream = nullptr;
    p->on_device = false;
}

#define PN_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return genphi_set_error(GENPHI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int genphi_panel_create(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                        const int64_t *pro_ids, int32_t rank, int32_t world, genphi_panel **out)
{
    if (!out) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_create: out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_create: bad rank / world");
    genphi_panel *p = new (std::nothrow) genphi_panel();
    if (!p) return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory");
    genphi::PlanOptions opt;
    opt.indices_only = true;
    std::string err;
    int rc;
    try {
        rc = genphi::build_plan(n_ind, ind, father, mother, n_pro, pro_ids, opt, p->plan, err);
    } catch (const std::bad_alloc &) { delete p; return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory while planning"); }
    if (rc) { delete p; return genphi_set_error(rc, err); }
    p->rank = rank; p->world = world;
    if (const char *e = genphi::env_hook("GENPHI_TEST_FAIL_ALLOC")) p->fail_alloc_at = std::atoi(e);
    p->naive = genphi::env_hook("GENPHI_PANEL_NAIVE") != nullptr;
    p->tuning = genphi::panel_tuning_create();
    const genphi::Plan &pl = p->plan;
    const int L = pl.n_levels;
    const int64_t N = pl.n_pro;
    p->row_begin = N * rank / world;
    p->n_rows_res = N * (rank + 1) / world - p->row_begin;
    if (L == 0) { *out = p; return GENPHI_OK; }

    // ---- owners.  The plan's steps give, per member of cut c, its sources in cut c-1 and whether it
    //      is dragged (source = itself: it inherits the owner) or new (round-robin in storage order);
    //      the probands (last cut, proband order) own by contiguous blocks -- fixed FIRST, and
    //      propagated upwards to the cuts they are dragged through, so that ownership is persistent.
    std::vector<std::vector<int>> owner(L);
    for (int c = 0; c < L; ++c) owner[c].assign(pl.cut_sizes[c], -1);
    for (int r = 0; r < world; ++r)
        for (int64_t k = N * r / world; k < N * (r + 1) / world; ++k) owner[L - 1][k] = r;
    for (int c = L - 1; c >= 1; --c) {                       // a dragged member is the same individual as its source
        const genphi::LevelStep &st = pl.steps[c - 1];
        for (int64_t k = 0; k < st.n; ++k)
            if (st.ord[k] >= 0 && owner[c][k] >= 0) owner[c - 1][st.srcA[k]] = owner[c][k];
    }
    for (int c = 0; c < L; ++c) {                            // everybody else: inherited when dragged, else round-robin
        int rr = 0;
        for (int64_t k = 0; k < pl.cut_sizes[c]; ++k) {
            const bool dragged = c > 0 && pl.steps[c - 1].ord[k] >= 0;
            const int inherited = dragged ? owner[c - 1][pl.steps[c - 1].srcA[k]] : -1;
            if (owner[c][k] < 0) owner[c][k] = dragged ? inherited : (rr++ % world);
            else if (dragged && inherited != owner[c][k]) {
                delete p;
                return genphi_set_error(GENPHI_ERR_ARG, "internal: ownership of a dragged member is not persistent");
            }
        }
    }

    // ---- local columns per cut; per step: columns to fetch, columns to send, column descriptors ----
    p->member.resize(L);
    std::vector<std::vector<int>> lcol(L);                   // position in cut -> local column (or -1)
    for (int c = 0; c < L; ++c) {
        lcol[c].assign(pl.cut_sizes[c], -1);
        for (int64_t k = 0; k < pl.cut_sizes[c]; ++k)
            if (owner[c][k] == rank) { lcol[c][k] = static_cast<int>(p->member[c].size()); p->member[c].push_back(static_cast<int>(k)); }
    }
    p->steps.resize(L - 1);
    for (int c = 1; c < L; ++c) {
        const genphi::LevelStep &st = pl.steps[c - 1];
        PanelStep &ps = p->steps[c - 1];
        const int n_prev = static_cast<int>(st.n_prev);
        ps.send_count.assign(world, 0); ps.recv_count.assign(world, 0);
        // what every rank fetches: the non-local sources of its new local columns, ascending by (owner, position)
        std::vector<std::vector<std::vector<int>>> want(world, std::vector<std::vector<int>>(world));   // want[r][s]: positions r asks s for
        {
            std::vector<int> mark(n_prev, -1);
            for (int r = 0; r < world; ++r) {
                for (int64_t k = 0; k < st.n; ++k) {
                    if (owner[c][k] != r) continue;
                    for (int q : {st.srcA[k], st.srcB[k]}) {
                        if (q == n_prev || owner[c - 1][q] == r || mark[q] == r) continue;
                        mark[q] = r;
                        want[r][owner[c - 1][q]].push_back(q);
                    }
                }
                for (int s = 0; s < world; ++s) std::sort(want[r][s].begin(), want[r][s].end());
            }
        }
        // this rank sends want[d][rank] to every d, and receives want[rank][s] from every s (in peer order)
        for (int d = 0; d < world; ++d) {
            ps.send_count[d] = static_cast<int64_t>(want[d][rank].size());
            for (int q : want[d][rank]) ps.send_cols.push_back(lcol[c - 1][q]);
        }
        std::vector<int> ext_of(n_prev + 1, -1);             // position in cut c-1 -> extended-local column
        const int n_own_prev = static_cast<int>(p->member[c - 1].size());
        for (int k = 0; k < n_own_prev; ++k) ext_of[p->member[c - 1][k]] = k;
        int e = n_own_prev;
        for (int s = 0; s < world; ++s) {
            ps.recv_count[s] = static_cast<int64_t>(want[rank][s].size());
            for (int q : want[rank][s]) ext_of[q] = e++;
        }
        ps.n_ext = e - n_own_prev;
        const int zcol = e;                                  // the zero column of the extended panel
        ext_of[n_prev] = zcol;
        ps.n_cols = static_cast<int>(p->member[c].size());
        ps.col.resize(ps.n_cols);
        for (int jl = 0; jl < ps.n_cols; ++jl) {
            const int j = p->member[c][jl];
            // a dragged column: source = itself (local), B = none, weight 1 (st.srcB is none already)
            ps.col[jl] = make_int4(ext_of[st.srcA[j]], ext_of[st.srcB[j]], j, st.ord[j]);
        }
        // ---- the same step for the row kernels: a source "row" is a row of the extended panel (zcol + 1 floats) ----
        ps.src_width = zcol + 1;
        // (test hooks, as for plans: read once per handle, in panel_tuning_create)
        const int lds_cap = genphi::panel_tuning_lds_cap(p->tuning, genphi::kPanelSplitMaxFloats);
        const int full_max = genphi::panel_tuning_full_max(p->tuning, genphi::kPanelFullMaxFloats);
        const int row4 = (ps.src_width + 3) / 4 * 4;
        ps.mode = p->naive ? 2 : (2 * row4 <= lds_cap && row4 <= full_max ? 0 : (row4 <= lds_cap && zcol < 65536 ? 1 : 2));
        if (ps.mode != 2) {
            const size_t nc = static_cast<size_t>(ps.n_cols);
            ps.pk_col.assign(nc + genphi::kPanelIdxPad, static_cast<unsigned>(zcol) | (static_cast<unsigned>(zcol) << 16));
            ps.ord_col.assign(nc + genphi::kPanelIdxPad, 0);
            ps.diag_col.assign(st.n, -1);
            for (int jl = 0; jl < ps.n_cols; ++jl) {
                const int4 cj = ps.col[jl];
                // a dragged column is stored as A = B = itself, so that EVERY column has weight 1/2 (as in the plan's pk words)
                const bool dragged_col = cj.w >= 0;
                ps.pk_col[jl] = static_cast<unsigned>(cj.x) | (static_cast<unsigned>(dragged_col ? cj.x : cj.y) << 16);
                ps.ord_col[jl] = cj.w;
                ps.diag_col[cj.z] = jl;
            }
            ps.work.resize(st.n);
            for (int64_t k = 0; k < st.n; ++k) ps.work[k] = static_cast<int>(k);
            std::stable_sort(ps.work.begin(), ps.work.end(), [&](int x, int y) {
                return st.srcA[x] != st.srcA[y] ? st.srcA[x] < st.srcA[y] : st.srcB[x] < st.srcB[y];     // "no B" = n_prev sorts last
            });
            if (ps.mode == 1) {
                // the hub walk of the cut's rows (planner.h), and per work row where the member's own column is and which
                // panel column holds its OTHER source (the hub of its segment: the staged row is the B source, and the
                // self kinship 1/2 + Psi[B][hub]/2 is read from it)
                genphi::build_hub_walk(st.srcA.data(), st.srcB.data(), st.ord.data(), n_prev, ps.work.data(), nullptr, static_cast<int>(st.n), 4, 1, ps.walk);
                ps.n_segs = static_cast<int>(ps.walk.seg4.size() / 4) - 2;
                ps.n_runs = static_cast<int>(ps.walk.run.size() / 4) - 1;
                ps.pdesc.resize(st.n);
                for (int g = 0; g < ps.n_segs; ++g) {
                    const int hub = ps.walk.seg4[4 * g + 1];
                    for (int w = ps.walk.seg4[4 * g]; w < ps.walk.seg4[4 * (g + 1)]; ++w) {
                        const int i = ps.walk.desc4[4 * w];
                        ps.pdesc[w] = make_int2(ps.diag_col[i], ps.diag_col[i] >= 0 ? ext_of[hub] : zcol);
                    }
                }
            }
        }
    }
    *out = p;
    return GENPHI_OK;
}

int64_t genphi_panel_n_steps(const genphi_panel *p) { return p ? std::max(p->plan.n_levels - 1, 0) : -1; }
int64_t genphi_panel_n_probands(const genphi_panel *p) { return p ? p->plan.n_pro : -1; }
/* device time (ms, HIP events) of the kernels of level step `step` in the last sweep: unpack of the received columns + the level kernel */
double genphi_panel_step_ms(const genphi_panel *cp, int32_t step)
{
    genphi_panel *p = const_cast<genphi_panel *>(cp);
    if (!p || step < 0 || step >= static_cast<int32_t>(p->step_ms.size())) return -1.0;
    if (p->on_device && p->ev_pending[step]) {               // a stream-ordered sweep: the events are read when somebody asks
        (void)hipSetDevice(p->device);
        if (hipEventSynchronize(p->ev[2 * step + 1]) == hipSuccess && hipEventElapsedTime(&p->step_ms[step], p->ev[2 * step], p->ev[2 * step + 1]) == hipSuccess)
            p->ev_pending[step] = 0;
    }
    return static_cast<double>(p->step_ms[step]);
}
int genphi_panel_step_mode(const genphi_panel *p, int32_t step)
{
    return (!p || step < 0 || step >= static_cast<int32_t>(p->steps.size())) ? -1 : p->steps[step].mode;
}

int genphi_panel_result_rows(const genphi_panel *p, int64_t *row_begin, int64_t *n_rows)
{
    if (!p) return genphi_set_error(GENPHI_ERR_ARG, "panel handle is NULL");
    if (row_begin) *row_begin = p->row_begin;
    if (n_rows) *n_rows = p->n_rows_res;
    return GENPHI_OK;
}

/* Exchange geometry of step `step`: send_cols[d] / recv_cols[s] = columns this rank sends to rank d /
 * receives from rank s (arrays of `world` entries); col_floats = floats per column (rows of the level). */
int genphi_panel_exchange_counts(const genphi_panel *p, int32_t step, int64_t *send_cols, int64_t *recv_cols, int64_t *col_floats)
{
    if (!p || step < 0 || step >= static_cast<int32_t>(p->steps.size())) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_exchange_counts: bad argument");
    const PanelStep &ps = p->steps[step];
    for (int r = 0; r < p->world; ++r) { if (send_cols) send_cols[r] = ps.send_count[r]; if (recv_cols) recv_cols[r] = ps.recv_count[r]; }
    if (col_floats) *col_floats = p->plan.steps[step].n_prev;
    return GENPHI_OK;
}

/* Bytes of device memory this rank needs: its two panels (with their tail padding), its row block of the
 * result, the per-step index arrays and the certificates (the size test of the host driver). */
double genphi_panel_device_bytes(const genphi_panel *p)
{
    if (!p) return 0.0;
    double need[2] = {0, 0};
    const int L = p->plan.n_levels;
    for (int c = 0; c < L; ++c) {
        const long long ext = c + 1 < L ? p->steps[c].n_ext : 0;
        const double f = static_cast<double>(p->plan.cut_sizes[c] + 1) * static_cast<double>(pitch(static_cast<long long>(p->member[c].size()) + ext));
        need[c & 1] = std::max(need[c & 1], f);
    }
    double bytes = 4.0 * (need[0] + need[1] + 2.0 * 64 * 1024);
    bytes += 4.0 * static_cast<double>(p->n_rows_res) * static_cast<double>(pitch(p->plan.n_pro));            // result row block
    bytes += 2.0 * 4.0 * static_cast<double>(p->plan.max_cut + 1) + 8.0 * static_cast<double>((p->plan.max_cut + 64) / 64 * 64);   // certificates, group lists
    for (int c = 0; c + 1 < L; ++c) {
        const PanelStep &ps = p->steps[c];
        bytes += 12.0 * static_cast<double>(p->plan.steps[c].n) + 4.0 * static_cast<double>(ps.send_cols.size()) + 16.0 * static_cast<double>(ps.col.size());
        bytes += 4.0 * static_cast<double>(ps.pk_col.size() + ps.ord_col.size() + ps.diag_col.size() + ps.work.size()) +
                 4.0 * static_cast<double>(ps.walk.desc4.size() + ps.walk.seg4.size() + ps.walk.run.size()) + 8.0 * static_cast<double>(ps.pdesc.size());
    }
    return bytes;
}

static int panel_upload_impl(genphi_panel *p, int device)
{
    if (p->on_device) { PN_TRY(hipSetDevice(p->device)); return GENPHI_OK; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return genphi_set_error(GENPHI_ERR_DEVICE, "no HIP device available: the gen.phi product path has no CPU fallback");
    if (device < 0) PN_TRY(hipGetDevice(&device));
    if (device >= ndev) return genphi_set_error(GENPHI_ERR_DEVICE, "device ordinal out of range");
    PN_TRY(hipSetDevice(device));
    p->device = device; p->on_device = true;
    PN_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    const genphi::Plan &pl = p->plan;
    const int L = pl.n_levels;
    auto pmalloc = [&](void **dst, size_t bytes) -> hipError_t {
        if (p->fail_alloc_at > 0 && ++p->alloc_count == p->fail_alloc_at) return hipErrorOutOfMemory;
        return hipMalloc(dst, bytes);
    };
    size_t need[2] = {0, 0};
    for (int c = 0; c < L; ++c) {
        const long long ext = c + 1 < L ? p->steps[c].n_ext : 0;
        need[c & 1] = std::max(need[c & 1], static_cast<size_t>((pl.cut_sizes[c] + 1) * pitch(static_cast<long long>(p->member[c].size()) + ext)));
    }
    for (int b = 0; b < 2; ++b) {
        if (!need[b]) continue;
        need[b] += 64 * 1024;                               // zeroed tail: the SPLIT kernels stage whole float4 batches past the last row
        PN_TRY(pmalloc(reinterpret_cast<void **>(&p->panel[b]), need[b] * sizeof(float)));
        PN_TRY(hipMemsetAsync(p->panel[b], 0, need[b] * sizeof(float), p->stream));   // (on the handle's stream: the null stream is not ordered with it)
        p->panel_floats[b] = need[b];
    }
    auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
        hipError_t e = pmalloc(dst, std::max<size_t>(bytes, 16));
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    const size_t S = p->steps.size();
    p->d_srcA.assign(S, nullptr); p->d_srcB.assign(S, nullptr); p->d_ord.assign(S, nullptr); p->d_send_cols.assign(S, nullptr);
    p->d_col.assign(S, nullptr); p->d_member.assign(L, nullptr);
    for (size_t s = 0; s < S; ++s) {
        const genphi::LevelStep &st = pl.steps[s];
        PN_TRY(up(st.srcA.data(), st.n * sizeof(int), reinterpret_cast<void **>(&p->d_srcA[s])));
        PN_TRY(up(st.srcB.data(), st.n * sizeof(int), reinterpret_cast<void **>(&p->d_srcB[s])));
        PN_TRY(up(st.ord.data(), st.n * sizeof(int), reinterpret_cast<void **>(&p->d_ord[s])));
        PN_TRY(up(p->steps[s].send_cols.data(), p->steps[s].send_cols.size() * sizeof(int), reinterpret_cast<void **>(&p->d_send_cols[s])));
        PN_TRY(up(p->steps[s].col.data(), p->steps[s].col.size() * sizeof(int4), reinterpret_cast<void **>(&p->d_col[s])));
    }
    for (int c = 0; c < L; ++c)
        PN_TRY(up(p->member[c].data(), p->member[c].size() * sizeof(int), reinterpret_cast<void **>(&p->d_member[c])));
    {
        int cus = 0;
        PN_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        p->n_cus = std::max(8, cus / 8 * 8);
    }
    p->glist_cap = static_cast<int>((pl.max_cut + 64) / 64 * 64);
    p->ev.assign(2 * S, nullptr);
    for (hipEvent_t &e : p->ev) PN_TRY(hipEventCreate(&e));
    for (hipEvent_t &e : p->ev_x) PN_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->step_ms.assign(S, 0.f);
    p->ev_pending.assign(S, 0);
    for (int b = 0; b < 2; ++b) PN_TRY(pmalloc(reinterpret_cast<void **>(&p->d_cert[b]), (static_cast<size_t>(pl.max_cut) + 1) * sizeof(int)));
    PN_TRY(pmalloc(reinterpret_cast<void **>(&p->d_counters), 20 * sizeof(int)));
    PN_TRY(pmalloc(reinterpret_cast<void **>(&p->d_glist), 2 * static_cast<size_t>(p->glist_cap) * sizeof(int)));
    p->d_step.assign(S, DevPanelStep());
    for (size_t s = 0; s < S; ++s) {
        const PanelStep &ps = p->steps[s];
        DevPanelStep &d = p->d_step[s];
        if (ps.mode == 2) continue;
        PN_TRY(up(ps.pk_col.data(), ps.pk_col.size() * sizeof(unsigned), reinterpret_cast<void **>(&d.pk_col)));
        PN_TRY(up(ps.ord_col.data(), ps.ord_col.size() * sizeof(int), reinterpret_cast<void **>(&d.ord_col)));
        PN_TRY(up(ps.diag_col.data(), ps.diag_col.size() * sizeof(int), reinterpret_cast<void **>(&d.diag_col)));
        PN_TRY(up(ps.work.data(), ps.work.size() * sizeof(int), reinterpret_cast<void **>(&d.work)));
        if (ps.mode == 1) {
            PN_TRY(up(ps.walk.desc4.data(), ps.walk.desc4.size() * sizeof(int), reinterpret_cast<void **>(&d.desc)));
            PN_TRY(up(ps.walk.seg4.data(), ps.walk.seg4.size() * sizeof(int), reinterpret_cast<void **>(&d.seg)));
            PN_TRY(up(ps.walk.run.data(), ps.walk.run.size() * sizeof(int), reinterpret_cast<void **>(&d.run)));
            PN_TRY(up(ps.pdesc.data(), ps.pdesc.size() * sizeof(int2), reinterpret_cast<void **>(&d.pdesc)));
        }
    }
    return GENPHI_OK;
}

// Out of memory is the EXPECTED failure of the capacity path: a failed upload releases whatever it had
// allocated, so that the handle never looks uploaded with null or partial buffers and a retry (after the
// caller freed memory, or with more ranks) starts from scratch.
static int panel_upload(genphi_panel *p, int device)
{
    const bool was = p->on_device;
    const int rc = panel_upload_impl(p, device);
    if (rc != GENPHI_OK && !was) {
        const std::string keep = genphi_last_error();
        panel_free_device(p);
        genphi_set_error(rc, keep);
    }
    return rc;
}

/* Starts a sweep: Psi_1 = 1/2 I on the local columns of the top cut.  Then, for step = 0 .. n_steps-1:
 *   genphi_panel_pack(step, d_send)      d_send: sum(send_cols) x col_floats floats (device)
 *   <all-to-all of the packed columns: send_cols[d] * col_floats floats to rank d>
 *   genphi_panel_compute(step, d_recv)   d_recv: sum(recv_cols) x col_floats floats (device), peer order */
int genphi_panel_begin(genphi_panel *p, int32_t device)
{
    if (!p) return genphi_set_error(GENPHI_ERR_ARG, "panel handle is NULL");
    int rc = panel_upload(p, device);
    if (rc) return rc;
    const genphi::Plan &pl = p->plan;
    if (pl.n_levels == 0) return GENPHI_OK;
    const int n0 = static_cast<int>(pl.cut_sizes[0]), nc = static_cast<int>(p->member[0].size());
    const long long ext = pl.n_levels > 1 ? p->steps[0].n_ext : 0;
    const long long ld = pitch(nc + ext);
    PN_TRY(hipMemsetAsync(p->panel[0], 0, static_cast<size_t>((n0 + 1) * ld) * sizeof(float), p->stream));
    if (nc > 0) {
        hipLaunchKernelGGL(panel_identity_kernel, dim3((nc + 255) / 256), dim3(256), 0, p->stream, p->panel[0], ld, n0, p->d_member[0], nc);
        PN_TRY(hipGetLastError());
    }
    PN_TRY(hipMemsetAsync(p->d_cert[0], 0, (static_cast<size_t>(pl.max_cut) + 1) * sizeof(int), p->stream));   // 1/2 I: every row certified
    p->cur = 0;
    // (no synchronisation: everything that follows is ordered behind this on the panel's stream -- the blocking pack / compute calls
    // synchronise themselves, the stream-ordered ones end in genphi_panel_sync)
    return GENPHI_OK;
}

// Stream ordering with the host driver's collective (which runs on the DRIVER's stream, e.g. torch's current one):
//   genphi_panel_pack_on(.., caller_stream)     the pack kernel is enqueued on the panel's stream and caller_stream is made to wait
//                                               for it (an event): the collective enqueued next on caller_stream reads complete columns
//   genphi_panel_compute_on(.., caller_stream)  the panel's stream waits for what caller_stream holds so far (the collective), then
//                                               unpack + level kernels are enqueued; nothing blocks the host
// caller_stream == nullptr: the blocking forms (genphi_panel_pack / _compute): a stream synchronisation at the end of each call.
static int panel_pack_impl(genphi_panel *p, int32_t step, float *d_send, bool ordered, hipStream_t caller)
{
    if (!p || step < 0 || step >= static_cast<int32_t>(p->steps.size())) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_pack: bad argument");
    if (!p->on_device) return genphi_set_error(GENPHI_ERR_DEVICE, "genphi_panel_begin first");
    PN_TRY(hipSetDevice(p->device));
    const PanelStep &ps = p->steps[step];
    const int n_send = static_cast<int>(ps.send_cols.size());
    if (n_send > 0) {
        if (!d_send) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_pack: d_send is NULL");
        const int n_rows = static_cast<int>(p->plan.steps[step].n_prev);
        const long long ld = pitch(static_cast<long long>(p->member[step].size()) + ps.n_ext);
        dim3 grid(static_cast<unsigned>(n_send), static_cast<unsigned>(std::min((n_rows + 255) / 256, 64)));
        hipLaunchKernelGGL(panel_pack_kernel, grid, dim3(256), 0, p->stream, p->panel[step & 1], ld, n_rows, p->d_send_cols[step], d_send);
        PN_TRY(hipGetLastError());
    }
    if (ordered) {
        // (also with nothing to send: the driver's next collective overwrites the receive buffer, which the previous step's unpack reads;
        // GENPHI_NO_STREAM: no collective follows -- one rank -- and a cross-stream wait costs ~8 us of device time each)
        if (caller != static_cast<hipStream_t>(GENPHI_NO_STREAM)) {
            PN_TRY(hipEventRecord(p->ev_x[0], p->stream));
            PN_TRY(hipStreamWaitEvent(caller, p->ev_x[0], 0));
        }
    } else if (n_send > 0) {
        PN_TRY(hipStreamSynchronize(p->stream));               // the host driver's collective runs on another stream
    }
    return GENPHI_OK;
}
int genphi_panel_pack(genphi_panel *p, int32_t step, float *d_send) { return panel_pack_impl(p, step, d_send, false, nullptr); }
int genphi_panel_pack_on(genphi_panel *p, int32_t step, float *d_send, void *caller_stream)
{
    return panel_pack_impl(p, step, d_send, true, static_cast<hipStream_t>(caller_stream));
}

static int panel_compute_impl(genphi_panel *p, int32_t step, const float *d_recv, bool ordered, hipStream_t caller)
{
    if (!p || step < 0 || step >= static_cast<int32_t>(p->steps.size())) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_compute: bad argument");
    if (!p->on_device) return genphi_set_error(GENPHI_ERR_DEVICE, "genphi_panel_begin first");
    PN_TRY(hipSetDevice(p->device));
    const genphi::Plan &pl = p->plan;
    const genphi::LevelStep &st = pl.steps[step];
    const PanelStep &ps = p->steps[step];
    const int n_prev = static_cast<int>(st.n_prev), n = static_cast<int>(st.n);
    const int n_own = static_cast<int>(p->member[step].size());
    const long long ldp = pitch(static_cast<long long>(n_own) + ps.n_ext);
    float *psi = p->panel[step & 1];
    if (ordered && caller != static_cast<hipStream_t>(GENPHI_NO_STREAM)) {      // the received columns are complete on the caller's stream
        PN_TRY(hipEventRecord(p->ev_x[1], caller));
        PN_TRY(hipStreamWaitEvent(p->stream, p->ev_x[1], 0));
    }
    PN_TRY(hipEventRecord(p->ev[2 * step], p->stream));
    if (ps.n_ext > 0) {
        if (!d_recv) return genphi_set_error(GENPHI_ERR_ARG, "genphi_panel_compute: d_recv is NULL");
        dim3 grid(static_cast<unsigned>((ps.n_ext + 63) / 64), static_cast<unsigned>((n_prev + 63) / 64));
        hipLaunchKernelGGL(panel_unpack_kernel, grid, dim3(256), 0, p->stream, psi, ldp, n_prev, n_own, ps.n_ext, d_recv,
                           p->d_cert[step & 1], genphi::panel_tuning_cert_thresh(p->tuning));
        PN_TRY(hipGetLastError());
    }
    const bool last = step + 1 == static_cast<int32_t>(p->steps.size());
    const long long ext_next = last ? 0 : p->steps[step + 1].n_ext;
    const long long ldo = pitch(static_cast<long long>(ps.n_cols) + ext_next);
    float *out = p->panel[(step + 1) & 1];
    int *cert_out = p->d_cert[(step + 1) & 1];
    PN_TRY(hipMemsetAsync(cert_out, 0, (static_cast<size_t>(n) + 1) * sizeof(int), p->stream));
    if (ps.mode != 2) {
        // the FULL / SPLIT row kernels over this rank's local columns (panel_launch.h)
        const DevPanelStep &d = p->d_step[step];
        genphi::PanelLaunch L;
        L.stream = p->stream; L.n_cus = p->n_cus; L.tuning = p->tuning;
        L.psi = psi; L.out = out; L.ld_prev = ldp; L.ld = ldo;
        L.n_prev = n_prev; L.n_cut = n; L.n_cols = ps.n_cols; L.src_width = ps.src_width;
        L.srcA = p->d_srcA[step]; L.srcB = p->d_srcB[step]; L.ord = p->d_ord[step];
        L.pk_col = d.pk_col; L.ord_col = d.ord_col; L.diag_col = d.diag_col; L.work = d.work;
        L.mode = ps.mode; L.desc = d.desc; L.seg = d.seg; L.run = d.run; L.pdesc = d.pdesc; L.n_segs = ps.n_segs; L.n_runs = ps.n_runs;
        L.cert_prev = p->d_cert[step & 1]; L.cert_out = cert_out;
        L.counters = p->d_counters; L.glist = p->d_glist; L.glist_cap = p->glist_cap;
        const int rc = genphi::launch_panel_level(L);
        if (rc) return rc;
    } else {
        // panel rows too long for LDS: one thread per (row, local column); every row of the new panel counts as
        // uncertified (the kernel does not track small values), which only selects the grouping-exact bodies later
        dim3 grid(static_cast<unsigned>(n + 1), static_cast<unsigned>(std::min<long long>((ldo + 255) / 256, 64)));
        hipLaunchKernelGGL(panel_level_kernel, grid, dim3(256), 0, p->stream, psi, ldp, n_prev, out, ldo, n, p->d_srcA[step], p->d_srcB[step],
                           p->d_ord[step], p->d_col[step], ps.n_cols);
        PN_TRY(hipGetLastError());
        PN_TRY(hipMemsetAsync(cert_out, 0xff, static_cast<size_t>(n) * sizeof(int), p->stream));
    }
    PN_TRY(hipEventRecord(p->ev[2 * step + 1], p->stream));
    p->ev_pending[step] = 1;
    p->cur = (step + 1) & 1;
    if (!ordered) {
        PN_TRY(hipStreamSynchronize(p->stream));
        PN_TRY(hipEventElapsedTime(&p->step_ms[step], p->ev[2 * step], p->ev[2 * step + 1]));
        p->ev_pending[step] = 0;
    }
    return GENPHI_OK;
}
int genphi_panel_compute(genphi_panel *p, int32_t step, const float *d_recv) { return panel_compute_impl(p, step, d_recv, false, nullptr); }
int genphi_panel_compute_on(genphi_panel *p, int32_t step, const float *d_recv, void *caller_stream)
{
    return panel_compute_impl(p, step, d_recv, true, static_cast<hipStream_t>(caller_stream));
}
/* Waits for everything the panel's stream holds (the end of a stream-ordered sweep). */
int genphi_panel_sync(genphi_panel *p)
{
    if (!p) return genphi_set_error(GENPHI_ERR_ARG, "panel handle is NULL");
    if (!p->on_device) return GENPHI_OK;
    PN_TRY(hipSetDevice(p->device));
    PN_TRY(hipStreamSynchronize(p->stream));
    return GENPHI_OK;
}

/* This rank's row block of the result (rows [row_begin, row_begin + n_rows) of Phi in proband order,
 * n_rows x N dense row-major Float32), after the last genphi_panel_compute.                          */
int genphi_panel_result_to_host(genphi_panel *p, float *out)
{
    if (!p) return genphi_set_error(GENPHI_ERR_ARG, "panel handle is NULL");
    const genphi::Plan &pl = p->plan;
    const int64_t N = pl.n_pro, nr = p->n_rows_res;
    if (nr == 0 || N == 0) return GENPHI_OK;
    if (!out) return genphi_set_error(GENPHI_ERR_ARG, "out is NULL");
    if (!p->on_device) return genphi_set_error(GENPHI_ERR_DEVICE, "genphi_panel_begin first");
    PN_TRY(hipSetDevice(p->device));
    const int L = pl.n_levels;
    // local columns of the last cut are this rank's probands, in proband order: positions row_begin ..
    const long long ldp = pitch(static_cast<long long>(p->member[L - 1].size()));
    if (p->result) { PN_TRY(hipFree(p->result)); p->result = nullptr; }
    const long long ldo = pitch(N);
    PN_TRY(hipMalloc(reinterpret_cast<void **>(&p->result), static_cast<size_t>(nr * ldo) * sizeof(float)));
    dim3 grid(static_cast<unsigned>((nr + 63) / 64), static_cast<unsigned>((N + 63) / 64));
    hipLaunchKernelGGL(panel_transpose_kernel, grid, dim3(256), 0, p->stream, p->panel[(L - 1) & 1], ldp, static_cast<int>(N),
                       static_cast<int>(nr), p->result, ldo);
    PN_TRY(hipGetLastError());
    PN_TRY(hipStreamSynchronize(p->stream));
    PN_TRY(hipMemcpy2D(out, N * sizeof(float), p->result, ldo * sizeof(float), N * sizeof(float), nr, hipMemcpyDeviceToHost));
    return GENPHI_OK;
}

void genphi_panel_destroy(genphi_panel *p)
{
    if (!p) return;
    panel_free_device(p);
    genphi::panel_tuning_destroy(p->tuning);
    delete p;
}

}  // extern "C"
