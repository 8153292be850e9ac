// planner.cpp -- host-side level planner (see planner.h).
//
// Reference behaviour being replaced (GenLib.jl v0.1.4):
//   src/compute.jl:193-207  _previous_generation (parents of a set, first occurrences)
//   src/compute.jl:236-251  generations by parent steps, cut sets by union / intersect
//   src/compute.jl:165-186  _index_pedigree, :287-289 founder_index assignment
// Design here (not a translation): one pass over the generations stamps every individual
// with the first and last parent-step distance at which it is reached; a cut is then
// "everyone whose [first,last] interval covers that distance" (SURVEY.md A.2).  The order
// inside intermediate cuts is free (A.2), so it is chosen for the level kernels; only the last
// cut has a contractual order (proband first-occurrence order).
// Everything is linear: cuts are emitted already ordered (dragged members by scanning the
// previous cut, new members from per-distance buckets filled in rank order), orderings by small
// integer keys are counting sorts, and the per-step index arrays are built by a few threads.
#include "planner.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "../../include/genphi.h"

namespace genphi {

namespace {

// stable counting sort of `in` by key[in[k]] (keys in [0, n_keys)) into `out`
void counting_sort(const std::vector<int32_t> &in, std::vector<int32_t> &out, const std::vector<int32_t> &key,
                   int32_t n_keys, std::vector<int32_t> &cnt)
{
    cnt.assign(static_cast<size_t>(n_keys) + 1, 0);
    for (int32_t r : in) cnt[key[r] + 1]++;
    for (int32_t k = 0; k < n_keys; ++k) cnt[k + 1] += cnt[k];
    out.resize(in.size());
    for (int32_t r : in) out[cnt[key[r]]++] = r;
}

// ID -> rank index.  Pedigree IDs are usually small dense integers: a direct table then; anything
// else (sparse, huge or negative labels -- IDs are Julia Int) goes through an open-addressing hash table
// (genea140: 41,523 IDs up to 900,506 -- a direct table of 3.6 MB costs its page faults on every plan, std::unordered_map
// a node allocation per individual: 0.9 of the 3.7 ms of that plan).
class IdMap {
public:
    void init(int64_t n, const int64_t *ind)
    {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int64_t i = 0; i < n; ++i) { lo = std::min(lo, ind[i]); hi = std::max(hi, ind[i]); }
        if (n > 0 && lo >= 0 && hi < 3 * n + 1024) {
            table_.assign(static_cast<size_t>(hi) + 1, -1);
            direct_ = true;
        } else {
            int bits = 4;
            while ((int64_t(1) << bits) < 2 * n + 2) ++bits;
            shift_ = 64 - bits;
            keys_.assign(size_t(1) << bits, 0);
            vals_.assign(size_t(1) << bits, -1);
            mask_ = (size_t(1) << bits) - 1;
        }
    }
    // false when the ID is already present
    bool insert(int64_t id, int32_t rank)
    {
        if (direct_) {
            if (table_[id] >= 0) return false;
            table_[id] = rank;
            return true;
        }
        size_t h = slot(id);
        while (vals_[h] >= 0) {
            if (keys_[h] == id) return false;
            h = (h + 1) & mask_;
        }
        keys_[h] = id; vals_[h] = rank;
        return true;
    }
    int32_t find(int64_t id) const      // -1 when absent
    {
        if (direct_) return (id < 0 || id >= static_cast<int64_t>(table_.size())) ? -1 : table_[id];
        size_t h = slot(id);
        while (vals_[h] >= 0) {
            if (keys_[h] == id) return vals_[h];
            h = (h + 1) & mask_;
        }
        return -1;
    }
private:
    size_t slot(int64_t id) const { return static_cast<size_t>((static_cast<uint64_t>(id) * 0x9E3779B97F4A7C15ull) >> shift_); }
    bool direct_ = false;
    std::vector<int32_t> table_;
    std::vector<int64_t> keys_;
    std::vector<int32_t> vals_;
    size_t mask_ = 0;
    int shift_ = 0;
};

int mode_for(int64_t n_prev, const PlanOptions &opt)
{
    if (opt.indices_only) return kModeFull;
    const int64_t lds_row = (n_prev + 1 + 3) / 4 * 4;
    if (2 * lds_row <= opt.lds_cap_floats && lds_row <= opt.full_max_floats) return kModeFull;
    if (lds_row <= opt.lds_cap_floats && n_prev < 65535) return kModeSplit;
    return kModeWide;
}

}  // namespace

namespace {
// buffers of reuse_order, kept between calls of one planning run (fresh vectors cost page faults)
struct ReuseScratch {
    std::vector<int32_t> tmp, byA, byB, cnt, a_begin, a_end, b_begin, b_end, stack, out;
    std::vector<char> seen;
};

void reuse_order_impl(const LevelStep &st, std::vector<int32_t> &rows, ReuseScratch &w)
{
    const int64_t m = static_cast<int64_t>(rows.size());
    if (m == 0) return;
    const int32_t none = static_cast<int32_t>(st.n_prev);
    const int32_t n_keys = none + 1;
    // rows by (A, B) -- runs of equal A are the sibling groups, "no B" sorts last inside a group --
    // and by (B, A): two stable counting sorts each
    counting_sort(rows, w.tmp, st.srcB, n_keys, w.cnt);
    counting_sort(w.tmp, w.byA, st.srcA, n_keys, w.cnt);
    counting_sort(rows, w.tmp, st.srcA, n_keys, w.cnt);
    counting_sort(w.tmp, w.byB, st.srcB, n_keys, w.cnt);
    // start offsets of every A value / B value present (values are < n_prev + 1)
    w.a_begin.assign(n_keys + 1, -1); w.a_end.assign(n_keys + 1, -1);
    w.b_begin.assign(n_keys + 1, -1); w.b_end.assign(n_keys + 1, -1);
    for (int64_t k = 0; k < m; ++k) {
        const int32_t A = st.srcA[w.byA[k]], B = st.srcB[w.byB[k]];
        if (w.a_begin[A] < 0) w.a_begin[A] = static_cast<int32_t>(k);
        w.a_end[A] = static_cast<int32_t>(k + 1);
        if (w.b_begin[B] < 0) w.b_begin[B] = static_cast<int32_t>(k);
        w.b_end[B] = static_cast<int32_t>(k + 1);
    }
    w.seen.assign(n_keys + 1, 0);
    w.stack.clear();
    w.out.clear();
    w.out.reserve(m);
    for (int64_t k0 = 0; k0 < m; ++k0) {
        const int32_t A0 = st.srcA[w.byA[k0]];
        if (w.seen[A0]) continue;
        w.seen[A0] = 1;
        w.stack.push_back(A0);
        while (!w.stack.empty()) {
            const int32_t A = w.stack.back();
            w.stack.pop_back();
            for (int32_t k = w.a_begin[A]; k < w.a_end[A]; ++k) {
                const int32_t r = w.byA[k];
                w.out.push_back(r);
                const int32_t B = st.srcB[r];
                if (B == none) continue;
                for (int32_t t = w.b_begin[B]; t < w.b_end[B]; ++t) {      // the other children of this B source
                    const int32_t A2 = st.srcA[w.byB[t]];
                    if (!w.seen[A2]) { w.seen[A2] = 1; w.stack.push_back(A2); }
                }
            }
        }
    }
    rows.assign(w.out.begin(), w.out.end());
}
// pk words, the position-test flag and the row processing order of a FULL / SPLIT step whose
// srcA / srcB / ord are set
void finish_narrow_step(LevelStep &st, ReuseScratch &w)
{
    const int64_t n = st.n, n_prev = st.n_prev;
    st.pk.resize(n);
    // column role: a dragged member is stored as A = B = itself, so every column
    // carries weight 1/2 (x + x is exact) and the kernels need no per-column weight
    for (int64_t k = 0; k < n; ++k) {
        const uint32_t A = static_cast<uint32_t>(st.srcA[k]);
        const uint32_t B = (st.ord[k] < 0) ? static_cast<uint32_t>(st.srcB[k]) : A;
        st.pk[k] = A | (B << 16);
    }
    // position test usable instead of the rank word?  (new members with both parents must
    // appear in rank order along the storage order: always so for [dragged, new by rank] cuts,
    // a matter of luck for the proband order of the last cut)
    bool mono = true;
    int32_t last = -1;
    for (int64_t k = 0; k < n && mono; ++k) {
        if (st.ord[k] < 0 && st.srcB[k] != n_prev) {
            const int32_t rk = st.ord[k] & INT32_MAX;
            if (rk < last) mono = false;
            last = rk;
        }
    }
    st.pos_ord = mono;
    // row processing order: sibling groups (same A source) adjacent, groups chained
    // depth-first along shared B sources (see reuse_order)
    st.work.resize(n);
    std::iota(st.work.begin(), st.work.end(), 0);
    reuse_order_impl(st, st.work, w);
}

}  // namespace

void reuse_order(const LevelStep &st, std::vector<int32_t> &rows)
{
    ReuseScratch w;
    reuse_order_impl(st, rows, w);
}

void build_hub_walk(const int32_t *srcA, const int32_t *srcB, const int32_t *ord, int32_t none, const int *rows,
                    const int *out_rows, int n_rows, int seg_cap, int max_run, WalkLists &out)
{
    out.desc4.clear(); out.seg4.clear(); out.run.clear(); out.row_k.clear();
    out.desc4.reserve(static_cast<size_t>(n_rows) * 4);
    out.row_k.reserve(n_rows);
    seg_cap = std::max(1, seg_cap);
    max_run = std::max(1, max_run);
    // adjacency (CSR) over the nodes 0 .. none: a row is listed at its A source and, when it has one, at its B source
    // (A == none only for parentless members, whose hub is the all-zero row; A == B, selfing, is listed once)
    std::vector<int32_t> start(static_cast<size_t>(none) + 2, 0);
    for (int k = 0; k < n_rows; ++k) {
        const int i = rows[k];
        start[srcA[i] + 1]++;
        if (srcB[i] != none && srcB[i] != srcA[i]) start[srcB[i] + 1]++;
    }
    for (int32_t v = 0; v <= none; ++v) start[v + 1] += start[v];
    std::vector<int32_t> adj(start[none + 1]), fill(start.begin(), start.end() - 1), ptr(start.begin(), start.end() - 1);
    for (int k = 0; k < n_rows; ++k) {
        const int i = rows[k];
        adj[fill[srcA[i]]++] = k;
        if (srcB[i] != none && srcB[i] != srcA[i]) adj[fill[srcB[i]]++] = k;
    }
    std::vector<char> done(n_rows, 0);
    auto has_work = [&](int32_t v) {
        while (ptr[v] < start[v + 1] && done[adj[ptr[v]]]) ++ptr[v];
        return ptr[v] < start[v + 1];
    };
    auto put_row = [&](int k, int32_t b_src) {
        const int i = rows[k];
        out.desc4.push_back(i); out.desc4.push_back(out_rows ? out_rows[k] : i); out.desc4.push_back(b_src); out.desc4.push_back(ord[i]);
        out.row_k.push_back(k);
    };
    std::vector<int32_t> singles, edges, spill;                 // spill: (hub, row) pairs of rows without B source beyond kMaxSingles per hub visit
    constexpr size_t kMaxHubChildren = 8;
    constexpr size_t kMaxSingles = 8;                           // a workgroup finishes them one after the other: all parentless members of a cut
                                                                // share the hub "none", and hundreds of them in one run would be a serial tail
    for (int k0 = 0; k0 < n_rows; ++k0) {
        if (done[k0]) continue;
        int32_t hub = srcA[rows[k0]];
        out.run.push_back(static_cast<int32_t>(out.seg4.size() / 4)); out.run.push_back(hub); out.run.push_back(0); out.run.push_back(0);
        int type = 0, stages = 1;                               // (the staged hub row of the run's first segment)
        for (;;) {
            singles.clear(); edges.clear();
            for (int32_t t = ptr[hub]; t < start[hub + 1]; ++t) {
                const int k = adj[t];
                if (done[k]) continue;
                done[k] = 1;
                (srcB[rows[k]] == none ? singles : edges).push_back(k);
            }
            ptr[hub] = start[hub + 1];
            while (singles.size() > kMaxSingles) { spill.push_back(hub); spill.push_back(singles.back()); singles.pop_back(); }
            auto other = [&](int k) { const int i = rows[k]; return srcA[i] == hub ? srcB[i] : srcA[i]; };
            // the child whose other parent still has work goes last: its stage yields the next hub's expansion
            int32_t next_hub = -1;
            for (size_t e = edges.size(); e-- > 0;) {
                const int32_t o = other(edges[e]);
                if (o != hub && has_work(o)) { next_hub = o; std::swap(edges[e], edges.back()); break; }
            }
            size_t e = 0, in_run = 0;
            bool first = true;
            do {                                                // segments of this hub: its rows without B source lead the first one
                const size_t m = std::min(edges.size() - e, static_cast<size_t>(seg_cap));
                int seg_type = first ? type : 2;
                if (!first && in_run + m > kMaxHubChildren) {   // a hub with many children: a new run (the hub row staged again) every
                    out.run.push_back(static_cast<int32_t>(out.seg4.size() / 4)); out.run.push_back(hub);    // kMaxHubChildren of them, so that
                    out.run.push_back(0); out.run.push_back(0);
                    seg_type = 0; in_run = 0;                   // items stay short (tail of the work queue; the chunks of a run stay in step)
                }
                out.seg4.push_back(static_cast<int32_t>(out.desc4.size() / 4)); out.seg4.push_back(hub);
                out.seg4.push_back(first ? static_cast<int32_t>(singles.size()) : 0); out.seg4.push_back(seg_type);
                if (first) for (int32_t k : singles) put_row(k, none);
                for (size_t q = 0; q < m; ++q) put_row(edges[e + q], other(edges[e + q]));
                e += m; in_run += m;
                first = false;
            } while (e < edges.size());
            stages += static_cast<int>(edges.size());
            if (next_hub < 0 || stages >= max_run) break;
            hub = next_hub;
            type = 1;
        }
    }
    // the rows without B source that exceeded a hub visit's share: runs of their own (the hub row staged again, up to
    // kMaxSingles rows finished from it)
    for (size_t q = 0; q < spill.size();) {
        const int32_t hub = spill[q];
        size_t e = q;
        while (e < spill.size() && spill[e] == hub && (e - q) / 2 < kMaxSingles) e += 2;
        out.run.push_back(static_cast<int32_t>(out.seg4.size() / 4)); out.run.push_back(hub); out.run.push_back(0); out.run.push_back(0);
        out.seg4.push_back(static_cast<int32_t>(out.desc4.size() / 4)); out.seg4.push_back(hub);
        out.seg4.push_back(static_cast<int32_t>((e - q) / 2)); out.seg4.push_back(0);
        for (; q < e; q += 2) put_row(spill[q + 1], none);
    }
    out.run.push_back(static_cast<int32_t>(out.seg4.size() / 4)); out.run.push_back(0); out.run.push_back(n_rows); out.run.push_back(n_rows);
    out.seg4.push_back(n_rows); out.seg4.push_back(0); out.seg4.push_back(0); out.seg4.push_back(0);       // terminator (two, so that seg[g + 2] is readable)
    out.seg4.push_back(n_rows); out.seg4.push_back(0); out.seg4.push_back(0); out.seg4.push_back(0);
    // a run's entry also carries its first segment (one scalar load starts an item): rows [z, w), n0 in the high half of y
    for (size_t r = 0; r + 1 < out.run.size() / 4; ++r) {
        const int32_t g = out.run[4 * r];
        out.run[4 * r + 1] |= out.seg4[4 * g + 2] << 16;
        out.run[4 * r + 2] = out.seg4[4 * g];
        out.run[4 * r + 3] = out.seg4[4 * (g + 1)];
    }
}

int build_plan(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
               int64_t n_pro, const int64_t *pro_ids, const PlanOptions &opt, Plan &plan,
               std::string &err)
{
    if (n_ind < 0 || n_pro < 0 || (n_ind > 0 && (!ind || !father || !mother)) || (n_pro > 0 && !pro_ids)) {
        err = "genphi_plan_create: null or negative argument";
        return GENPHI_ERR_ARG;
    }
    if (n_ind >= (int64_t(1) << 31) - 1) { err = "pedigree too large (>= 2^31 individuals)"; return GENPHI_ERR_ARG; }

    PhaseTrace trace;
    // ---- id -> rank index; parents must precede children (src/create.jl:234-254) ----------
    IdMap rank_of;
    rank_of.init(n_ind, ind);
    std::vector<int32_t> fa(n_ind), mo(n_ind);
    for (int64_t i = 0; i < n_ind; ++i) {
        int32_t f = -1, m = -1;
        if (father[i] != 0) {
            f = rank_of.find(father[i]);
            if (f < 0) {
                err = "individual " + std::to_string(ind[i]) + ": father " + std::to_string(father[i]) +
                      " is unknown or listed after its child (pedigree must be in rank order)";
                return GENPHI_ERR_ORDER;
            }
        }
        if (mother[i] != 0) {
            m = rank_of.find(mother[i]);
            if (m < 0) {
                err = "individual " + std::to_string(ind[i]) + ": mother " + std::to_string(mother[i]) +
                      " is unknown or listed after its child (pedigree must be in rank order)";
                return GENPHI_ERR_ORDER;
            }
        }
        if (!rank_of.insert(ind[i], static_cast<int32_t>(i))) {
            err = "duplicate individual ID " + std::to_string(ind[i]);
            return GENPHI_ERR_DUPLICATE_ID;
        }
        fa[i] = f; mo[i] = m;
    }

    trace.mark("  plan: ids, parents");
    // ---- probands: first occurrences, in the caller's order (`∩` at src/compute.jl:251) ----
    std::vector<int32_t> tfirst(n_ind, -1), tlast(n_ind, -1), stamp(n_ind, -1);
    std::vector<int32_t> cur;
    cur.reserve(n_pro);
    for (int64_t k = 0; k < n_pro; ++k) {
        const int32_t r = rank_of.find(pro_ids[k]);
        if (r < 0) {
            err = "KeyError: proband " + std::to_string(pro_ids[k]) + " not found";
            return GENPHI_ERR_UNKNOWN_ID;
        }
        if (stamp[r] != 0) { stamp[r] = 0; cur.push_back(r); }
    }
    plan = Plan();
    plan.n_ind = n_ind;
    plan.n_pro = static_cast<int64_t>(cur.size());
    plan.final_members = cur;

    // ---- generations by parent steps; t = distance from the probands ---------------------
    int32_t t = 0;
    std::vector<int32_t> nxt;
    while (!cur.empty()) {
        nxt.clear();
        for (int32_t x : cur) {
            if (tfirst[x] < 0) tfirst[x] = t;
            tlast[x] = t;
            const int32_t f = fa[x], m = mo[x];
            if (f >= 0 && stamp[f] != t + 1) { stamp[f] = t + 1; nxt.push_back(f); }
            if (m >= 0 && stamp[m] != t + 1) { stamp[m] = t + 1; nxt.push_back(m); }
        }
        cur.swap(nxt);
        ++t;
    }
    trace.mark("  plan: generations");
    const int32_t L = t;                     // number of cuts (0 when there are no probands)
    plan.n_levels = L;
    if (L == 0) return GENPHI_OK;

    // cut c (c = 0 top founders ... L-1 probands) = { x : tfirst <= d_c <= tlast }, d_c = L-1-c.
    // x is NEW in cut c iff tlast[x] == d_c (it is not in cut c-1); bucket the individuals by
    // tlast while scanning them in rank order: every bucket comes out rank-sorted.
    std::vector<std::vector<int32_t>> new_of(L);
    {
        std::vector<int64_t> cnt(L, 0);
        for (int64_t x = 0; x < n_ind; ++x) if (tlast[x] >= 0) cnt[L - 1 - tlast[x]]++;
        for (int32_t c = 0; c < L; ++c) new_of[c].reserve(cnt[c]);
        for (int64_t x = 0; x < n_ind; ++x) if (tlast[x] >= 0) new_of[L - 1 - tlast[x]].push_back(static_cast<int32_t>(x));
    }

    // cut sizes (independent of any order): x is in the cuts c_new(x) = L-1-tlast .. c_last(x) = L-1-tfirst
    std::vector<int64_t> size_of(L, 0);
    {
        std::vector<int64_t> diff(static_cast<size_t>(L) + 1, 0);
        for (int64_t x = 0; x < n_ind; ++x) if (tlast[x] >= 0) { diff[L - 1 - tlast[x]]++; diff[L - tfirst[x]]--; }
        int64_t run = 0;
        for (int32_t c = 0; c < L; ++c) { run += diff[c]; size_of[c] = run; }
    }
    trace.mark("  plan: buckets, cut sizes");
    // Block assembly (kModeWide) per cut: blk[c] = the step that produces cut c assembles its level block by block.  Always so when a
    // source row does not fit in LDS; beyond that, steps of narrower cuts are switched to it when a run of them can be kept IN PLACE
    // (LevelStep::stay) and the bytes saved -- the dragged x dragged block is most of a level of overlapping generations, real
    // genealogies: src/compute.jl:108-110 -- outweigh the extra passes (the cost model below, in matrix entries moved).
    const bool stay_on = !opt.indices_only && !opt.no_stay;
    std::vector<char> blk(L, 0);
    for (int32_t c = 1; c < L; ++c) blk[c] = mode_for(size_of[c - 1], opt) == kModeWide ? 1 : 0;
    std::vector<int64_t> n_par_of(L, 0);                                 // distinct parents of the new members of cut c
    {
        std::vector<int32_t> seen(n_ind > 0 ? n_ind : 1, -1);
        for (int32_t c = 1; c < L && stay_on; ++c) {
            int64_t np = 0;
            for (int32_t x : new_of[c]) {
                if (fa[x] >= 0 && seen[fa[x]] != c) { seen[fa[x]] = c; ++np; }
                if (mo[x] >= 0 && seen[mo[x]] != c) { seen[mo[x]] = c; ++np; }
            }
            n_par_of[c] = np;
        }
    }
    // (the new x new block of the step itself and of the step that reads its cut must have a row kernel: the per-entry
    // fallback knows no slots; cut 1 is out because cut 0 = 1/2 I is normally never materialised)
    auto nn_ok = [&](int32_t c) { return new_of[c].empty() || mode_for(n_par_of[c], opt) != kModeWide; };
    // want[c]: cut c may be produced in place -- worth it while the dragged x dragged block that is not copied outweighs the new
    // rows (late levels of shrinking cuts have few dragged members, and reading a cut by slot costs the next step its faster
    // route).  run_end[c] > 0: a run of in-place cuts c .. run_end[c] is planned (the slot search below may still end it early).
    std::vector<char> want(L, 0);
    std::vector<int32_t> run_end(L, 0);
    if (stay_on) {
        auto nnew = [&](int32_t c) { return static_cast<double>(new_of[c].size()); };
        auto ndrag = [&](int32_t c) { return static_cast<double>(size_of[c]) - nnew(c); };
        // entries moved by the step that produces cut c: as a row-kernel level (every row staged and written whole) ...
        // (+ a fixed cost per step in the same unit: a row-kernel level is one or two launches, a block-assembled one six to eight
        // short ones -- clearing, Psi_P, padding, the sub-step's kernels, scatter, the new rows' pass -- about 50 us against 8, i.e.
        // ~64M against ~10M entries at 5 TB/s.  Without it the byte counts alone put cuts of 3-6k members in place, measured 11-45 %
        // SLOWER than their row kernels: profiles/microbench/out/r04_narrow_in_place_planner_choice.out)
        const double ov_rows = 0.15 * opt.stay_step_overhead, ov_blk = opt.stay_step_overhead;
        auto cost_rows = [&](int32_t c) {
            const double np = static_cast<double>(size_of[c - 1]), n = static_cast<double>(size_of[c]);
            return (ndrag(c) + 1.5 * nnew(c)) * np + n * n + ov_rows;
        };
        // ... the blocks that involve new members, common to both forms of block assembly (Psi_P, the new x new sub-step) ...
        auto cost_nn = [&](int32_t c) {
            const double q = static_cast<double>(n_par_of[c]);
            return 2.0 * q * q + 1.5 * nnew(c) * q + nnew(c) * nnew(c);
        };
        // ... assembled compactly (every row re-written at the dragged columns, the transposed block) ...
        auto cost_blk = [&](int32_t c) {
            const double np = static_cast<double>(size_of[c - 1]), n = static_cast<double>(size_of[c]);
            return cost_nn(c) + (ndrag(c) + 2.0 * nnew(c)) * np + n * ndrag(c) + 2.0 * nnew(c) * ndrag(c) + ov_blk;
        };
        // ... in place (only the new rows and columns move; the new x new block may go through the scatter buffer)
        auto cost_stay = [&](int32_t c) { return cost_nn(c) + nnew(c) * nnew(c) + 3.5 * nnew(c) * ndrag(c) + ov_blk; };
        // The proband cut itself may stay in place at the end of a run: its step then writes only the new probands' rows and columns
        // and the result is DELIVERED from the slot matrix by one permutation pass (colperm_kernel with the probands' slots) instead of
        // being compacted into storage order first and permuted then.  deliver = that pass (every proband's row of the slot matrix staged
        // once per chunk of 40k result columns, N^2 entries written), charged to the run that ends in the proband cut.
        const double n_last = static_cast<double>(size_of[L - 1]);
        const double deliver = n_last * 1.1 * static_cast<double>(*std::max_element(size_of.begin(), size_of.end())) * std::ceil(n_last / 40960.0) + n_last * n_last;
        if (opt.stay_last && L >= 3) {
            const int32_t c = L - 1;
            // (when the cut before the probands could stay in place as well, stopping short of the probands also costs the compacting
            // step that then has to produce that cut: the alternative to "through the proband cut" is dearer by that much)
            double stop_short = 0.0;
            if (c >= 3 && opt.stay_narrow && size_of[c - 2] >= opt.stay_narrow_min && size_of[c - 1] > static_cast<int64_t>(new_of[c - 1].size()) &&
                100 * size_of[c - 1] >= static_cast<int64_t>(opt.stay_min_ratio_pct) * static_cast<int64_t>(new_of[c - 1].size()) && nn_ok(c - 1))
                stop_short = std::max(0.0, cost_blk(c - 1) - cost_stay(c - 1));
            const bool narrow = opt.stay_narrow && size_of[c - 1] >= opt.stay_narrow_min &&
                                (opt.stay_narrow_force || cost_stay(c) + deliver < (blk[c] ? cost_blk(c) + deliver : cost_rows(c) + stop_short));
            want[c] = (blk[c] || narrow) && 100 * size_of[c] >= static_cast<int64_t>(opt.stay_min_ratio_pct) * static_cast<int64_t>(new_of[c].size()) &&
                      size_of[c] > static_cast<int64_t>(new_of[c].size()) && nn_ok(c);
        }
        for (int32_t c = 2; c + 1 < L; ++c) {
            const bool by_width = blk[c] && blk[c + 1];                  // both steps assemble blocks anyway
            const bool narrow = opt.stay_narrow && size_of[c - 1] >= opt.stay_narrow_min &&
                                (c + 2 < L || want[L - 1]) &&            // (the proband step reads a cut stored by slot only when it stays in place itself)
                                (opt.stay_narrow_force || cost_stay(c) < (blk[c] ? cost_blk(c) : cost_rows(c)));
            want[c] = (by_width || narrow) && 100 * size_of[c] >= static_cast<int64_t>(opt.stay_min_ratio_pct) * static_cast<int64_t>(new_of[c].size()) &&
                      size_of[c] > static_cast<int64_t>(new_of[c].size()) && nn_ok(c) && nn_ok(c + 1);
        }
        for (int32_t c = 2; c + 1 < L;) {
            if (!want[c]) { ++c; continue; }
            int32_t e = c;
            while (e + 1 < L && want[e + 1]) ++e;                        // (may reach the proband cut L - 1)
            double gain;
            bool all_wide;
            if (e == L - 1) {                                            // no step reads this run's last cut: the delivery pass instead
                gain = (blk[e] ? deliver : 0.0) - deliver;
                all_wide = true;
            } else {
                // the step that reads the run's last cut by slot assembles blocks: a loss when it would have been a row-kernel level
                gain = blk[e + 1] ? 0.0 : cost_rows(e + 1) - cost_blk(e + 1);
                all_wide = blk[e + 1] != 0;
            }
            for (int32_t cc = c; cc <= e; ++cc) gain += (blk[cc] ? cost_blk(cc) : cost_rows(cc)) - cost_stay(cc);
            for (int32_t cc = c; cc <= e; ++cc) all_wide = all_wide && blk[cc];
            if (all_wide || gain > 0.0 || opt.stay_narrow_force) run_end[c] = e;
            c = e + 1;
        }
    }
    // Cuts produced in place: their new members are ordered by the cut after which they leave, earliest first (rank order inside
    // a class), so that the slots of a block die from its start.
    if (stay_on) {
        std::vector<int32_t> key(n_ind > 0 ? n_ind : 1, 0), tmp, cnt;
        bool any_run = false;
        for (int32_t c = 2; c + 1 < L && !any_run; ++c) any_run = run_end[c] > 0;
        // (every block of such a plan, the cuts above the region too: the cut a run starts from is made of them)
        for (int32_t c = 0; c + 1 < L && any_run; ++c) {
            if (new_of[c].size() < 2) continue;
            // (inside a class: families together, by the father's rank, rank order inside a family -- the workgroup of
            // rows_avg_t_kernel that holds a granule of 64 new rows then reads a father's row once for his children)
            if (opt.stay_family_order)
                std::stable_sort(new_of[c].begin(), new_of[c].end(), [&](int32_t a, int32_t b) { return fa[a] < fa[b]; });
            for (int32_t x : new_of[c]) key[x] = L - 1 - tfirst[x];      // the last cut x is in: early leavers first (the slots are
                                                                         // a circular queue: what entered first, or sits lowest, dies first)
            counting_sort(new_of[c], tmp, key, L, cnt);
            new_of[c].swap(tmp);
        }
    }

    trace.mark("  plan: cost model, orders");
    // ---- storage order of every cut: [dragged by previous position..., new by rank...];
    //      the last cut keeps the proband order (contractual) unless its step is WIDE ----------
    std::vector<std::vector<int32_t>> cut(L);
    cut[0] = new_of[0];
    for (int32_t c = 1; c < L; ++c) {
        const int32_t d = L - 1 - c;
        std::vector<int32_t> &cc = cut[c];
        cc.reserve(cut[c - 1].size() + new_of[c].size());
        for (int32_t x : cut[c - 1]) if (tfirst[x] <= d) cc.push_back(x);       // dragged: still needed below
        cc.insert(cc.end(), new_of[c].begin(), new_of[c].end());
    }
    plan.cut_sizes.resize(L);
    plan.ld.resize(L);
    for (int32_t c = 0; c < L; ++c) {
        plan.cut_sizes[c] = static_cast<int64_t>(cut[c].size());
        plan.ld[c] = pitch_for(plan.cut_sizes[c]);
        plan.max_cut = std::max(plan.max_cut, plan.cut_sizes[c]);
    }
    trace.mark("  plan: cuts");
    // ---- runs of WIDE steps whose members stay in place (see LevelStep::stay) ---------------------------------
    // stay_c[c]: the step producing cut c writes in place; slotP[c] > 0: cut c is stored by slot, capacity slotP[c];
    // slots_c[c][k]: slot of member k of cut c; abs?_c[c][k]: slots of the sources (in cut c - 1) of member k of cut c
    std::vector<char> stay_c(L, 0), contig_c(L, 0);
    std::vector<int32_t> slotP(L, 0), npad_c(L, 0);
    std::vector<std::vector<int32_t>> slots_c(L), absA_c(L), absB_c(L), gran_c(L);
    if (stay_on) {
        auto pad64 = [](int64_t v) { return (v + 63) / 64 * 64; };
        std::vector<int32_t> slot_of(n_ind > 0 ? n_ind : 1, -1);
        std::vector<int32_t> last_at;                                // per slot: last cut of its occupant (-1: free)
        for (int32_t c = 1; c + 1 < L;) {
            if (run_end[c] <= 0) { ++c; continue; }
            const int32_t e = run_end[c];
            int64_t P = 0, blk_max = 0;
            for (int32_t cc = c; cc <= e; ++cc) {
                P = std::max(P, size_of[cc - 1] + pad64(static_cast<int64_t>(new_of[cc].size())));
                blk_max = std::max(blk_max, pad64(static_cast<int64_t>(new_of[cc].size())));
            }
            P = pad64(P + P * std::max(0, opt.stay_slack_pct) / 100 + 64) + std::max(0, opt.stay_headroom) * blk_max;      // slack for granules that are only partly dead
            if (P > opt.stay_max_slots || P >= (int64_t(1) << 30)) { c = e + 1; continue; }
            // the entry cut c - 1 sits at slots [0, n) of the matrix (written compactly, with pitch P, by its own step)
            const std::vector<int32_t> &ent = cut[c - 1];
            last_at.assign(static_cast<size_t>(P), -1);
            for (size_t k = 0; k < ent.size(); ++k) { slot_of[ent[k]] = static_cast<int32_t>(k); last_at[k] = L - 1 - tfirst[ent[k]]; }
            slots_c[c - 1].resize(ent.size());
            std::iota(slots_c[c - 1].begin(), slots_c[c - 1].end(), 0);
            // New members are placed 64 slots (a GRANULE) at a time: a granule is free when everything in it left the cuts before
            // the step's source cut.  The search goes round the slot space from where the last one ended, so the new members of a
            // step mostly sit in a few long stretches (what died together was born together: blocks are ordered by leaving time).
            const int64_t n_gran = P / 64;
            int64_t gpos = pad64(static_cast<int64_t>(ent.size())) / 64;
            int32_t done = c - 1;                                    // last cut stored by slot
            auto sources = [&](int32_t cc) {                         // slots of the sources of cut cc's members (cut cc - 1 is stored by slot)
                const int32_t d = L - 1 - cc;
                const size_t n = cut[cc].size();
                absA_c[cc].resize(n); absB_c[cc].resize(n);
                const int32_t none = static_cast<int32_t>(P);
                for (size_t k = 0; k < n; ++k) {
                    const int32_t x = cut[cc][k];
                    if (tlast[x] > d) { absA_c[cc][k] = slot_of[x]; absB_c[cc][k] = none; continue; }
                    const int32_t a = fa[x] >= 0 ? slot_of[fa[x]] : -1, b = mo[x] >= 0 ? slot_of[mo[x]] : -1;
                    if (a >= 0 && b >= 0) { absA_c[cc][k] = a; absB_c[cc][k] = b; }
                    else if (a >= 0) { absA_c[cc][k] = a; absB_c[cc][k] = none; }
                    else if (b >= 0) { absA_c[cc][k] = b; absB_c[cc][k] = none; }
                    else { absA_c[cc][k] = absB_c[cc][k] = none; }
                }
            };
            std::vector<int32_t> got;
            for (int32_t cc = c; cc <= e; ++cc) {
                const int64_t n_new = static_cast<int64_t>(new_of[cc].size()), need = (n_new + 63) / 64;
                got.clear();
                auto free_gran = [&](int64_t g) {
                    for (int64_t q = 64 * g; q < 64 * g + 64; ++q) if (last_at[q] >= cc - 1) return false;       // (free: gone before the source cut)
                    return true;
                };
                // first choice: ONE stretch of free granules (the new x new block is then written in place, not scattered)
                int64_t g = gpos;
                {
                    int64_t run_len = 0, s0 = -1;
                    for (int64_t scanned = 0, gg = gpos; scanned < n_gran + need && s0 < 0 && need > 0; ++scanned, gg = (gg + 1) % n_gran) {
                        if (gg == 0) run_len = 0;                    // a stretch does not wrap
                        run_len = free_gran(gg) ? run_len + 1 : 0;
                        if (run_len >= need) s0 = gg - need + 1;
                    }
                    if (s0 >= 0) { for (int64_t k = 0; k < need; ++k) got.push_back(static_cast<int32_t>(64 * (s0 + k))); g = (s0 + need) % n_gran; }
                }
                const bool contig = need > 0 && static_cast<int64_t>(got.size()) == need && !opt.stay_scatter;
                if (!contig) {
                    got.clear();
                    g = gpos;
                    for (int64_t scanned = 0; scanned < n_gran && static_cast<int64_t>(got.size()) < need; ++scanned, g = (g + 1) % n_gran)
                        if (free_gran(g)) got.push_back(static_cast<int32_t>(64 * g));
                }
                if (static_cast<int64_t>(got.size()) < need) break;  // the slot space is full: this step copies the cut out (and may start a new run)
                gpos = g;
                contig_c[cc] = contig ? 1 : 0;
                sources(cc);                                         // (before the new members get their slots)
                const size_t n = cut[cc].size(), nd = n - static_cast<size_t>(n_new);
                slots_c[cc].resize(n);
                for (size_t k = 0; k < nd; ++k) slots_c[cc][k] = slot_of[cut[cc][k]];
                for (int32_t base : got) for (int32_t q = base; q < base + 64; ++q) last_at[q] = -1;
                for (size_t k = nd; k < n; ++k) {
                    const int32_t x = cut[cc][k], q = got[(k - nd) / 64] + static_cast<int32_t>((k - nd) % 64);
                    slot_of[x] = q; slots_c[cc][k] = q; last_at[q] = L - 1 - tfirst[x];
                }
                stay_c[cc] = 1; gran_c[cc] = got; npad_c[cc] = static_cast<int32_t>(64 * need);
                done = cc;
            }
            if (done >= c) {
                // (a proband cut that stays in place: the RESULT takes the run's pitch too -- plan.ld[L - 1] is the result's row pitch)
                for (int32_t cc = c - 1; cc <= done; ++cc) { slotP[cc] = static_cast<int32_t>(P); plan.ld[cc] = P; }
                for (int32_t cc = c; cc <= std::min(done + 1, L - 1); ++cc) blk[cc] = 1;      // the run's steps and the step that leaves it assemble blocks
                if (done + 1 < L) sources(done + 1);                 // the step that leaves the run reads by slot
                else {                                               // the run ends in the proband cut: delivered from the probands' slots
                    plan.final_slots.resize(plan.final_members.size());
                    for (size_t k = 0; k < plan.final_members.size(); ++k) plan.final_slots[k] = slot_of[plan.final_members[k]];
                }
                // (a run the slot space ended early: what is left of it may start again behind the compacting step)
                if (done + 2 <= e && done + 2 + 1 < L) run_end[done + 2] = e;
                c = done + 2;
            } else {
                slots_c[c - 1].clear();
                if (c + 1 <= e) run_end[c + 1] = e;
                ++c;
            }
        }
    }
    // Memory: the two level buffers with in-place runs against plain alternation.  Usually less (one P x P matrix + the entry cut
    // instead of two matrices of the widest cuts); several runs with large P in both buffers can need more -- then not in place
    // (unless it is small change anyway: narrow cuts, stay_mem_floor_bytes).
    if (stay_on) {
        double need_s[2] = {0, 0}, need_p[2] = {0, 0};
        int b = 0;
        bool any = false;
        for (int32_t c = 0; c + 1 < L; ++c) {
            if (c >= 1) b = stay_c[c] ? b : 1 - b;
            any = any || stay_c[c];
            const double rows = static_cast<double>(slotP[c] > 0 ? slotP[c] : plan.cut_sizes[c]) + 1.0;
            need_s[b] = std::max(need_s[b], rows * static_cast<double>(plan.ld[c]));
            need_p[c & 1] = std::max(need_p[c & 1], (static_cast<double>(plan.cut_sizes[c]) + 1.0) * static_cast<double>(pitch_for(plan.cut_sizes[c])));
        }
        if (any && need_s[0] + need_s[1] > opt.stay_mem_ratio * (need_p[0] + need_p[1]) && 4.0 * (need_s[0] + need_s[1]) > opt.stay_mem_floor_bytes) {
            PlanOptions o2 = opt;
            o2.no_stay = true;
            return build_plan(n_ind, ind, father, mother, n_pro, pro_ids, o2, plan, err);
        }
    }
    const bool last_wide = L >= 2 && blk[L - 1];
    if (!last_wide) {
        cut[L - 1] = plan.final_members;
    } else {
        std::vector<int32_t> pos(n_ind, -1);
        for (size_t k = 0; k < cut[L - 1].size(); ++k) pos[cut[L - 1][k]] = static_cast<int32_t>(k);
        plan.final_perm.resize(plan.final_members.size());
        for (size_t k = 0; k < plan.final_members.size(); ++k) plan.final_perm[k] = pos[plan.final_members[k]];
    }
    plan.both_counts.assign(L - 1, 0);
    plan.steps.resize(L - 1);

    trace.mark("  plan: slots");
    // ---- positions chain through the cuts: one serial, linear pass snapshots, for every member
    //      of cut c, the positions of its sources in cut c-1 (-1 = none) ----------------------------
    std::vector<std::vector<int32_t>> posA(L), posB(L);
    {
        std::vector<int32_t> pos_even(n_ind, -1), pos_odd(n_ind, -1);
        auto pos_of = [&](int32_t c) -> std::vector<int32_t> & { return (c & 1) ? pos_odd : pos_even; };
        for (size_t k = 0; k < cut[0].size(); ++k) pos_of(0)[cut[0][k]] = static_cast<int32_t>(k);
        for (int32_t c = 1; c < L; ++c) {
            const std::vector<int32_t> &pp = pos_of(c - 1);
            const int32_t d = L - 1 - c;
            const size_t n = cut[c].size();
            posA[c].resize(n); posB[c].resize(n);
            for (size_t k = 0; k < n; ++k) {
                const int32_t x = cut[c][k];
                if (tlast[x] > d) { posA[c][k] = pp[x]; posB[c][k] = -1; }            // dragged: also in cut c-1
                else { posA[c][k] = fa[x] >= 0 ? pp[fa[x]] : -1; posB[c][k] = mo[x] >= 0 ? pp[mo[x]] : -1; }
            }
            std::vector<int32_t> &pc = pos_of(c);
            for (size_t k = 0; k < n; ++k) pc[cut[c][k]] = static_cast<int32_t>(k);
        }
    }

    trace.mark("  plan: positions");
    // ---- the flat index arrays of every step: independent of each other, built by a few threads ----
    auto build_step = [&](int32_t c, ReuseScratch &w) {
        LevelStep &st = plan.steps[c - 1];
        const int64_t n = plan.cut_sizes[c], n_prev = plan.cut_sizes[c - 1];
        const int32_t d = L - 1 - c;
        st.n_prev = n_prev; st.n = n;
        st.ld_prev = plan.ld[c - 1]; st.ld = plan.ld[c];
        // (the entry cut of an in-place run has the run's pitch P but is written compactly: no zero padding beyond its own width)
        st.width = (slotP[c] > 0 && !stay_c[c]) ? pitch_for(n) : st.ld;
        st.mode = blk[c] ? kModeWide : mode_for(n_prev, opt);
        st.srcA.resize(n); st.srcB.resize(n); st.ord.resize(n);
        const int32_t none = static_cast<int32_t>(n_prev);
        int64_t dragged = 0;
        for (int64_t k = 0; k < n; ++k) {
            const int32_t x = cut[c][k];
            const int32_t a = posA[c][k], b = posB[c][k];
            if (tlast[x] > d) { st.srcA[k] = a; st.srcB[k] = none; st.ord[k] = x; ++dragged; }
            else {
                if (a >= 0 && b >= 0) { st.srcA[k] = a; st.srcB[k] = b; }
                else if (a >= 0) { st.srcA[k] = a; st.srcB[k] = none; }
                else if (b >= 0) { st.srcA[k] = b; st.srcB[k] = none; }
                else { st.srcA[k] = st.srcB[k] = none; }
                st.ord[k] = x | kNewFlag;
            }
        }
        st.n_dragged = dragged;
        plan.both_counts[c - 1] = dragged;
        if (opt.indices_only) return;
        if (st.mode != kModeWide) { finish_narrow_step(st, w); return; }
        if (slotP[c - 1] > 0) {                                        // the source cut is stored by slot
            st.src_slots = true; st.P = slotP[c - 1];
            st.absA = std::move(absA_c[c]); st.absB = std::move(absB_c[c]);
        }
        if (stay_c[c]) {
            st.stay = true; st.npad = npad_c[c]; st.out_slots = slots_c[c]; st.blk_slot = gran_c[c]; st.contig = contig_c[c] != 0;
            st.p0 = st.blk_slot.empty() ? 0 : st.blk_slot[0];
            // slot ranges (whole granules) that hold the dragged members: what the new columns are written to.  Gaps of up to three
            // dead granules are bridged (fewer, longer launches; nothing reads a dead slot) -- never across a granule of this step's
            // new members: their columns belong to the new x new block.
            {
                const int32_t n_gran = st.P / 64;
                std::vector<char> gl(static_cast<size_t>(n_gran), 0);
                for (int64_t k = 0; k < dragged; ++k) gl[st.out_slots[k] / 64] = 1;
                for (int32_t b : st.blk_slot) gl[b / 64] = 2;
                for (int32_t g = 0; g < n_gran;) {
                    if (gl[g] != 1) { ++g; continue; }
                    int32_t e = g;                                   // last live granule of the range
                    for (int32_t h = g + 1; h < n_gran && h - e <= 4 && gl[h] != 2; ++h) if (gl[h] == 1) e = h;
                    st.live_ranges.push_back(64 * g); st.live_ranges.push_back(64 * (e + 1));
                    g = e + 1;
                }
            }
        }

        // WIDE: the cut is [dragged..., new...]; the new x new block becomes a step of its own
        // over the compacted parent x parent matrix Psi[parents][parents]
        st.work.resize(n);
        std::iota(st.work.begin(), st.work.end(), 0);
        const int64_t n_new = n - dragged;
        std::vector<char> is_par(n_prev + 1, 0);
        for (int64_t k = dragged; k < n; ++k) { is_par[st.srcA[k]] = 1; is_par[st.srcB[k]] = 1; }
        is_par[none] = 0;
        std::vector<int32_t> pidx(n_prev + 1, -1);
        for (int64_t q = 0; q < n_prev; ++q)
            if (is_par[q]) { pidx[q] = static_cast<int32_t>(st.parents.size()); st.parents.push_back(static_cast<int32_t>(q)); }
        const int64_t n_par = static_cast<int64_t>(st.parents.size());
        if (st.src_slots) {
            st.parents_abs.resize(n_par);
            for (int64_t u = 0; u < n_par; ++u) st.parents_abs[u] = slots_c[c - 1][st.parents[u]];
        }
        const int nn_mode = mode_for(n_par, opt);
        if (nn_mode == kModeWide) { st.nn_naive = true; return; }
        st.nn.resize(1);
        LevelStep &nn = st.nn[0];
        // written in place (rows / columns [dragged, n) of this cut's matrix): `lead` placeholder
        // members in front put the block's first column on a 16-byte boundary of the rows (128-byte
        // alignment measured no faster: profiles/microbench/out/r02_ab_nn_block_alignment_cfg4o.out)
        // (a step that stays in place: no placeholders, rows of npad columns ...
        const int64_t lead = st.stay ? 0 : dragged % 4;
        nn.lead = static_cast<int32_t>(lead);
        nn.n_prev = n_par; nn.n = lead + n_new;
        // (... and writes the block into a buffer of its own, pitch npad, that is scattered to the new members' slots afterwards)
        // (... unless the new members' slots are one stretch: then in place, at [p0, p0 + npad))
        nn.ld_prev = pitch_for(n_par); nn.ld = (st.stay && !st.contig) ? st.npad : st.ld; nn.width = st.stay ? st.npad : st.width - (dragged - lead);
        nn.mode = nn_mode;
        nn.srcA.resize(nn.n); nn.srcB.resize(nn.n); nn.ord.resize(nn.n);
        const int32_t pnone = static_cast<int32_t>(n_par);
        for (int64_t k = 0; k < lead; ++k) { nn.srcA[k] = nn.srcB[k] = pnone; nn.ord[k] = 0; }   // column of zeros, no row
        for (int64_t k = 0; k < n_new; ++k) {
            const int32_t A = st.srcA[dragged + k], B = st.srcB[dragged + k];
            nn.srcA[lead + k] = A == none ? pnone : pidx[A];
            nn.srcB[lead + k] = B == none ? pnone : pidx[B];
            nn.ord[lead + k] = st.ord[dragged + k];
        }
        nn.n_dragged = 0;
        finish_narrow_step(nn, w);
        if (lead) {                    // the placeholders have no row
            size_t o = 0;
            for (int32_t r : nn.work) if (r >= lead) nn.work[o++] = r;
            nn.work.resize(o);
        }
    };
    // Eight threads for plans of 150k members over all cuts and more, four from 50k on (measured on a GPU box,
    // profiles/microbench/out/r05_planner_threads_*.out: cfg4 44.5 ms with 1 thread, 28.5 with 4, 25.3 with 8, 23.5 with 16 -- the phases
    // above this one are serial; cfg3s 8.95 -> 7.1 with 8; cfg3 4.13 -> 2.46 with 8; genea140 2.45 -> 1.66 with 4, 1.61 with 8).  Smaller
    // plans (cfg5: 200 cuts of 50) stay on the calling thread: their steps are shorter than a thread's start.  (The build container's 8
    // CPUs gain nothing from the threads and lose nothing either.)  GENPHI_PLAN_THREADS overrides.
    int64_t members = 0;
    for (int32_t c = 0; c < L; ++c) members += plan.cut_sizes[c];
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency() / 2);
    int n_thr = members >= 150000 ? static_cast<int>(std::min(8u, hw)) : members >= 50000 ? static_cast<int>(std::min(4u, hw)) : 1;
    if (const char *e = env_hook("GENPHI_PLAN_THREADS")) n_thr = std::max(1, std::min(32, std::atoi(e)));
    n_thr = std::min(n_thr, std::max(1, L - 1));
    if (n_thr <= 1) {
        ReuseScratch w;
        for (int32_t c = 1; c < L; ++c) build_step(c, w);
    } else {
        std::vector<std::thread> th;
        std::vector<char> oom(n_thr, 0);
        for (int t2 = 0; t2 < n_thr; ++t2)
            th.emplace_back([&, t2]() {
                try {
                    ReuseScratch w;
                    for (int32_t c = L - 1 - t2; c >= 1; c -= n_thr) build_step(c, w);      // the big last step first
                } catch (const std::bad_alloc &) { oom[t2] = 1; }
            });
        for (auto &x : th) x.join();
        for (char e : oom) if (e) throw std::bad_alloc();
    }

    trace.mark("  plan: step arrays");
    double bytes = 0.0;
    for (int32_t c = 0; c + 1 < L; ++c) {
        const double a = static_cast<double>(plan.cut_sizes[c]), b = static_cast<double>(plan.cut_sizes[c + 1]);
        bytes += 4.0 * (a * a + b * b);
    }
    plan.algorithmic_bytes = bytes;
    return GENPHI_OK;
}

}  // namespace genphi
