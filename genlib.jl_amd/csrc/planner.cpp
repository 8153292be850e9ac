// planner.cpp -- host-side level planner (see planner.h).
//
// Reference behaviour being replaced (GenLib.jl v0.1.4):
//   src/compute.jl:193-207  _previous_generation (parents of a set, first occurrences)
//   src/compute.jl:236-251  generations by parent steps, cut sets by union / intersect
//   src/compute.jl:165-186  _index_pedigree, :287-289 founder_index assignment
// Design here (not a translation): one pass over the generations stamps every individual
// with the first and last parent-step distance at which it is reached; a cut is then
// "everyone whose [first,last] interval covers that distance" (SURVEY.md A.2).  The order
// inside intermediate cuts is free (A.2), so it is chosen for HBM/LDS locality of the level
// kernels; only the last cut has a contractual order (proband first-occurrence order).
#include "planner.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <unordered_map>

#include "../../include/genphi.h"

namespace genphi {

namespace {

struct Member {
    int32_t x;        // pedigree rank index
    int32_t A, B;     // sources in previous cut (n_prev = none)
    int32_t bucket;   // LDS window of B in the previous cut (HALF mode) else 0
    int32_t group;    // 1 = x is an LDS-side (B) source of the NEXT step (HALF mode) else 0
    bool is_new;
};

}  // namespace

void reuse_order(const LevelStep &st, std::vector<int32_t> &rows)
{
    const int64_t m = static_cast<int64_t>(rows.size());
    if (m == 0) return;
    const int32_t none = static_cast<int32_t>(st.n_prev);
    // rows by (A, B): runs of equal A are the sibling groups ("no B" sorts last inside a group)
    std::vector<int32_t> byA(rows);
    std::sort(byA.begin(), byA.end(), [&](int32_t a, int32_t b) {
        if (st.srcA[a] != st.srcA[b]) return st.srcA[a] < st.srcA[b];
        if (st.srcB[a] != st.srcB[b]) return st.srcB[a] < st.srcB[b];
        return a < b;
    });
    std::vector<int32_t> byB(rows);
    std::sort(byB.begin(), byB.end(), [&](int32_t a, int32_t b) {
        if (st.srcB[a] != st.srcB[b]) return st.srcB[a] < st.srcB[b];
        return st.srcA[a] < st.srcA[b];
    });
    // start offsets of every A value / B value present (values are < n_prev + 1)
    std::vector<int32_t> a_begin(st.n_prev + 2, -1), a_end(st.n_prev + 2, -1), b_begin(st.n_prev + 2, -1), b_end(st.n_prev + 2, -1);
    for (int64_t k = 0; k < m; ++k) {
        const int32_t A = st.srcA[byA[k]], B = st.srcB[byB[k]];
        if (a_begin[A] < 0) a_begin[A] = static_cast<int32_t>(k);
        a_end[A] = static_cast<int32_t>(k + 1);
        if (b_begin[B] < 0) b_begin[B] = static_cast<int32_t>(k);
        b_end[B] = static_cast<int32_t>(k + 1);
    }
    std::vector<char> seen(st.n_prev + 2, 0);
    std::vector<int32_t> stack, out;
    out.reserve(m);
    for (int64_t k0 = 0; k0 < m; ++k0) {
        const int32_t A0 = st.srcA[byA[k0]];
        if (seen[A0]) continue;
        seen[A0] = 1;
        stack.push_back(A0);
        while (!stack.empty()) {
            const int32_t A = stack.back();
            stack.pop_back();
            for (int32_t k = a_begin[A]; k < a_end[A]; ++k) {
                const int32_t r = byA[k];
                out.push_back(r);
                const int32_t B = st.srcB[r];
                if (B == none) continue;
                for (int32_t t = b_begin[B]; t < b_end[B]; ++t) {      // the other children of this B source
                    const int32_t A2 = st.srcA[byB[t]];
                    if (!seen[A2]) { seen[A2] = 1; stack.push_back(A2); }
                }
            }
        }
    }
    rows.swap(out);
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

    // ---- id -> rank index; parents must precede children (src/create.jl:234-254) ----------
    std::unordered_map<int64_t, int32_t> rank_of;
    rank_of.reserve(static_cast<size_t>(n_ind) * 2);
    std::vector<int32_t> fa(n_ind), mo(n_ind);
    for (int64_t i = 0; i < n_ind; ++i) {
        int32_t f = -1, m = -1;
        if (father[i] != 0) {
            auto it = rank_of.find(father[i]);
            if (it == rank_of.end()) {
                err = "individual " + std::to_string(ind[i]) + ": father " + std::to_string(father[i]) +
                      " is unknown or listed after its child (pedigree must be in rank order)";
                return GENPHI_ERR_ORDER;
            }
            f = it->second;
        }
        if (mother[i] != 0) {
            auto it = rank_of.find(mother[i]);
            if (it == rank_of.end()) {
                err = "individual " + std::to_string(ind[i]) + ": mother " + std::to_string(mother[i]) +
                      " is unknown or listed after its child (pedigree must be in rank order)";
                return GENPHI_ERR_ORDER;
            }
            m = it->second;
        }
        if (!rank_of.emplace(ind[i], static_cast<int32_t>(i)).second) {
            err = "duplicate individual ID " + std::to_string(ind[i]);
            return GENPHI_ERR_DUPLICATE_ID;
        }
        fa[i] = f; mo[i] = m;
    }

    // ---- probands: first occurrences, in the caller's order (`∩` at src/compute.jl:251) ----
    std::vector<int32_t> tfirst(n_ind, -1), tlast(n_ind, -1), stamp(n_ind, -1);
    std::vector<int32_t> cur;
    cur.reserve(n_pro);
    for (int64_t k = 0; k < n_pro; ++k) {
        auto it = rank_of.find(pro_ids[k]);
        if (it == rank_of.end()) {
            err = "KeyError: proband " + std::to_string(pro_ids[k]) + " not found";
            return GENPHI_ERR_UNKNOWN_ID;
        }
        if (stamp[it->second] != 0) { stamp[it->second] = 0; cur.push_back(it->second); }
    }
    plan = Plan();
    plan.n_ind = n_ind;
    plan.n_pro = static_cast<int64_t>(cur.size());
    plan.final_members = cur;

    // ---- generations by parent steps; t = distance from the probands ---------------------
    std::vector<int32_t> seen;              // everyone reached, in discovery order
    int32_t t = 0;
    std::vector<int32_t> nxt;
    while (!cur.empty()) {
        nxt.clear();
        for (int32_t x : cur) {
            if (tfirst[x] < 0) { tfirst[x] = t; seen.push_back(x); }
            tlast[x] = t;
            const int32_t f = fa[x], m = mo[x];
            if (f >= 0 && stamp[f] != t + 1) { stamp[f] = t + 1; nxt.push_back(f); }
            if (m >= 0 && stamp[m] != t + 1) { stamp[m] = t + 1; nxt.push_back(m); }
        }
        cur.swap(nxt);
        ++t;
    }
    const int32_t L = t;                     // number of cuts (0 when there are no probands)
    plan.n_levels = L;
    if (L == 0) return GENPHI_OK;

    // cut c (c = 0 top founders ... L-1 probands) = { x : tfirst <= L-1-c <= tlast }
    std::vector<std::vector<int32_t>> cut(L);
    {
        std::vector<int64_t> cnt(L, 0);
        for (int32_t x : seen) for (int32_t d = tfirst[x]; d <= tlast[x]; ++d) cnt[L - 1 - d]++;
        for (int32_t c = 0; c < L; ++c) cut[c].reserve(cnt[c]);
        for (int32_t x : seen) for (int32_t d = tfirst[x]; d <= tlast[x]; ++d) cut[L - 1 - d].push_back(x);
    }
    cut[L - 1] = plan.final_members;         // contractual order of the result

    plan.cut_sizes.resize(L);
    plan.ld.resize(L);
    for (int32_t c = 0; c < L; ++c) {
        plan.cut_sizes[c] = static_cast<int64_t>(cut[c].size());
        plan.ld[c] = pitch_for(plan.cut_sizes[c]);
        plan.max_cut = std::max(plan.max_cut, plan.cut_sizes[c]);
    }
    plan.both_counts.assign(L > 0 ? L - 1 : 0, 0);
    plan.steps.resize(L - 1);

    // x in cut c is "dragged" in step c-1 -> c iff it is also in cut c-1, i.e. tlast[x] > L-1-c
    auto is_dragged = [&](int32_t x, int32_t c) { return tlast[x] > L - 1 - c; };
    auto step_mode = [&](int32_t c_prev) {      // step c_prev -> c_prev+1
        const int64_t lds_row = (plan.cut_sizes[c_prev] + 1 + 3) / 4 * 4;
        if (2 * lds_row <= opt.lds_cap_floats && lds_row <= opt.full_max_floats) return int(kModeFull);
        if (lds_row <= opt.lds_cap_floats && plan.cut_sizes[c_prev] < 65535) return int(kModeSplit);
        return int(kModeHalf);
    };
    auto step_is_half = [&](int32_t c_prev) { return step_mode(c_prev) == kModeHalf; };

    // ---- order the cuts top-down and emit the flat index arrays ----------------------------
    std::vector<int32_t> pos_prev(n_ind, -1), pos_cur(n_ind, -1), mark(n_ind, -1);
    std::vector<Member> mem;
    int64_t s2_begin_prev = 0, win_len_prev = 0;   // LDS windows of the previous cut (HALF steps)
    for (int32_t c = 0; c < L; ++c) {
        const int64_t n = plan.cut_sizes[c];
        const int64_t n_prev = c > 0 ? plan.cut_sizes[c - 1] : 0;
        const bool in_half = c > 0 && step_is_half(c - 1);        // step producing this cut
        const bool out_half = c + 1 < L && step_is_half(c);       // step consuming this cut
        // who is an LDS-side (B) source of the next step?
        if (out_half) {
            for (int32_t y : cut[c + 1]) {
                if (!is_dragged(y, c + 1) && fa[y] >= 0 && mo[y] >= 0) mark[mo[y]] = c;
            }
        }
        mem.resize(n);
        for (int64_t k = 0; k < n; ++k) {
            Member &m = mem[k];
            const int32_t x = cut[c][k];
            m.x = x; m.bucket = 0; m.group = (out_half && mark[x] == c) ? 1 : 0;
            if (c == 0) { m.A = m.B = 0; m.is_new = true; continue; }
            if (is_dragged(x, c)) { m.is_new = false; m.A = pos_prev[x]; m.B = static_cast<int32_t>(n_prev); }
            else {
                m.is_new = true;
                const int32_t f = fa[x], mm = mo[x];
                if (f >= 0 && mm >= 0) { m.A = pos_prev[f]; m.B = pos_prev[mm]; }
                else if (f >= 0) { m.A = pos_prev[f]; m.B = static_cast<int32_t>(n_prev); }
                else if (mm >= 0) { m.A = pos_prev[mm]; m.B = static_cast<int32_t>(n_prev); }
                else { m.A = m.B = static_cast<int32_t>(n_prev); }
            }
            if (in_half && m.B != n_prev) m.bucket = static_cast<int32_t>((m.B - s2_begin_prev) / win_len_prev);
        }
        // storage order: intermediate cuts by (group, bucket, A, B); the last cut keeps the
        // proband order unless its step is HALF (then a locality order + final_perm).
        const bool reorder = (c < L - 1) || in_half;
        std::vector<int32_t> order(n);
        std::iota(order.begin(), order.end(), 0);
        // FULL / SPLIT steps: [dragged by previous position..., new by rank...] so that the
        // kernels need no per-column rank word; HALF steps keep their window-bucket order
        const bool pos_ord = reorder && c > 0 && !in_half && !out_half;
        if (pos_ord) {
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                const Member &p = mem[a], &q = mem[b];
                if (p.is_new != q.is_new) return !p.is_new;
                if (!p.is_new) return p.A < q.A;
                return p.x < q.x;
            });
        } else if (reorder && c > 0) {
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                const Member &p = mem[a], &q = mem[b];
                if (p.group != q.group) return p.group < q.group;
                if (p.bucket != q.bucket) return p.bucket < q.bucket;
                if (p.A != q.A) return p.A < q.A;
                return p.B < q.B;
            });
        } else if (reorder) {
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return mem[a].group < mem[b].group; });
        }
        for (int64_t k = 0; k < n; ++k) pos_cur[mem[order[k]].x] = static_cast<int32_t>(k);
        if (c == L - 1 && reorder) {
            plan.final_perm.resize(n);
            for (int64_t k = 0; k < n; ++k) plan.final_perm[k] = pos_cur[plan.final_members[k]];
        }

        if (c > 0) {
            LevelStep &st = plan.steps[c - 1];
            st.n_prev = n_prev; st.n = n;
            st.ld_prev = plan.ld[c - 1]; st.ld = plan.ld[c];
            st.mode = step_mode(c - 1);
            st.pos_ord = pos_ord;
            st.srcA.resize(n); st.srcB.resize(n); st.ord.resize(n);
            int64_t dragged = 0;
            for (int64_t k = 0; k < n; ++k) {
                const Member &m = mem[order[k]];
                st.srcA[k] = m.A; st.srcB[k] = m.B;
                st.ord[k] = m.x | (m.is_new ? kNewFlag : 0);
                dragged += m.is_new ? 0 : 1;
            }
            st.n_dragged = dragged;
            if (st.mode != kModeHalf) {
                st.pk.resize(n);
                // column role: a dragged member is stored as A = B = itself, so every column
                // carries weight 1/2 (x + x is exact) and the kernels need no per-column weight
                for (int64_t k = 0; k < n; ++k) {
                    const uint32_t A = static_cast<uint32_t>(st.srcA[k]);
                    const uint32_t B = (st.ord[k] < 0) ? static_cast<uint32_t>(st.srcB[k]) : A;
                    st.pk[k] = A | (B << 16);
                }
                // position test usable instead of the rank word?  (new members with both
                // parents must appear in rank order along the storage order)
                if (!st.pos_ord) {
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
                }
            }
            plan.both_counts[c - 1] = dragged;
            // row processing order: sibling groups (same A source) adjacent, groups chained
            // depth-first along shared B sources (see reuse_order)
            st.work.resize(n);
            std::iota(st.work.begin(), st.work.end(), 0);
            if (in_half) {
                std::stable_sort(st.work.begin(), st.work.end(), [&](int32_t a, int32_t b) {
                    if (st.srcA[a] != st.srcA[b]) return st.srcA[a] < st.srcA[b];
                    return st.srcB[a] < st.srcB[b];
                });
            } else {
                reuse_order(st, st.work);
            }
            if (in_half) {
                // column segments: runs of equal (group, bucket); processed bucket-major so a
                // window is staged once per row
                st.b_rel.resize(n);
                struct Run { int32_t b, e, bucket; };
                std::vector<Run> runs;
                for (int64_t k = 0; k < n;) {
                    const Member &m0 = mem[order[k]];
                    int64_t e = k + 1;
                    while (e < n && mem[order[e]].group == m0.group && mem[order[e]].bucket == m0.bucket) ++e;
                    runs.push_back({static_cast<int32_t>(k), static_cast<int32_t>(e), m0.bucket});
                    k = e;
                }
                std::stable_sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.bucket < b.bucket; });
                for (const Run &r : runs) {
                    Segment sg;
                    sg.col_begin = r.b; sg.col_end = r.e;
                    sg.win_begin = static_cast<int32_t>(s2_begin_prev + int64_t(r.bucket) * win_len_prev);
                    sg.win_len = static_cast<int32_t>(std::min<int64_t>(win_len_prev, n_prev - sg.win_begin));
                    // the zero slot sits right after the window: include the matrix's own zero
                    // column when the window reaches the end of the row
                    for (int32_t k = r.b; k < r.e; ++k) {
                        const int32_t B = st.srcB[k];
                        st.b_rel[k] = (B == n_prev) ? sg.win_len : B - sg.win_begin;
                    }
                    st.segs.push_back(sg);
                }
            }
        }
        // LDS windows over this cut for the next step
        if (out_half) {
            int64_t s2 = n;
            for (int64_t k = 0; k < n; ++k) if (mem[order[k]].group == 1) { s2 = k; break; }
            s2 = s2 / 4 * 4;
            const int64_t n_s2 = std::max<int64_t>(n - s2, 1);
            const int64_t wmax = std::max<int64_t>((opt.lds_cap_floats / 2 - 4) / 4 * 4, 4);
            const int64_t n_win = (n_s2 + wmax - 1) / wmax;
            int64_t w = (n_s2 + n_win - 1) / n_win;
            w = (w + 3) / 4 * 4;
            s2_begin_prev = s2; win_len_prev = w;
        } else { s2_begin_prev = 0; win_len_prev = 1; }
        pos_prev.swap(pos_cur);
    }

    double bytes = 0.0;
    for (int32_t c = 0; c + 1 < L; ++c) {
        const double a = static_cast<double>(plan.cut_sizes[c]), b = static_cast<double>(plan.cut_sizes[c + 1]);
        bytes += 4.0 * (a * a + b * b);
    }
    plan.algorithmic_bytes = bytes;
    return GENPHI_OK;
}

}  // namespace genphi
