/*
 * sparse_oracle.cpp -- CPU restatement of GenLib.jl's sparse_phi / KinshipMatrix.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/ may load this library (through oracle/oracle.py); it is
 * the checker for genphi_sparse_* in the product, never something the product calls.
 *
 * Literal restatement (same containers, same loops, same key order, same Float32 stores) of
 *   src/compute.jl:31-46    KinshipMatrix, getindex by rank (smaller rank outside)
 *   src/compute.jl:321-447  sparse_phi: branching, founders into a queue, per individual: self
 *                           kinship, kinship with every live rank, retirement of parents whose
 *                           children are all processed, children enqueued once both parents are done
 *   src/compute.jl:467-472  phiMean(::KinshipMatrix)
 *   src/extract.jl:65-186   branching(pedigree, pro = ...) (ancestors of the probands, re-ranked)
 * including the quirk that entries are STORED under (rank of the earlier processed, rank of the
 * later processed) but LOOKED UP under (smaller rank, larger rank) (:356-381): when two individuals
 * of the same depth leave the queue in the opposite order of their ranks, their kinship is stored
 * where no lookup finds it (it reads as 0 from then on, and stays in a proband's dictionary).
 * Restated, not fixed.
 *
 * Pinned by the reference's own test values (test/runtests.jl:54-57, on data/geneaJi.csv):
 * phiMean == 0.171875, [1, 2] == 0.37109375, "3×3 KinshipMatrix with 6 stored entries."
 * (tests/test_oracle_golden.py).
 */
#include <algorithm>
#include <cstdint>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Ind {                       /* IndexedIndividual, src/compute.jl:14-24 */
    int64_t ID = 0;
    int father = -1, mother = -1;  /* index into the isolated pedigree, -1 = nothing */
    std::vector<int> children;
    int rank = 0;                  /* 1-based */
    int founder_index = 0;
    bool is_proband = false;
    int children_to_process = 0;
};

struct Sparse {
    std::unordered_map<int, std::unordered_map<int, float>> dict;   /* Dict{Int32, Dict{Int32, Float32}} keyed by rank */
    std::unordered_map<int64_t, int> id_to_rank;                    /* probands only */
    std::vector<int64_t> order_ids;                                 /* processing order (IDs), for tests of the product's schedule */
    std::vector<int64_t> retired_ids, retired_at;                   /* who was dropped (:401-430), and at which processing index (0-based) */
};

}  // namespace

extern "C" {

/* ped arrays in rank order (parents before children), 0 = unknown parent.  Returns a handle or NULL
 * (unknown proband ID). */
void *sparse_oracle_create(int64_t n, const int64_t *ind, const int64_t *father, const int64_t *mother, int64_t n_pro,
                           const int64_t *pro)
{
    std::unordered_map<int64_t, int> at;
    for (int64_t i = 0; i < n; ++i) at[ind[i]] = static_cast<int>(i);
    /* branching(pedigree, pro = probandIDs), src/extract.jl:65-186: mark the probands' ancestors,
     * keep them in pedigree order, ranks 1.. in that order */
    std::vector<char> keep(n, 0);
    std::vector<int> stack;
    for (int64_t k = 0; k < n_pro; ++k) {
        auto it = at.find(pro[k]);
        if (it == at.end()) return nullptr;
        stack.push_back(it->second);
        while (!stack.empty()) {
            const int x = stack.back(); stack.pop_back();
            if (keep[x]) continue;
            keep[x] = 1;
            if (father[x] != 0) stack.push_back(at[father[x]]);
            if (mother[x] != 0) stack.push_back(at[mother[x]]);
        }
    }
    std::vector<Ind> iso;
    std::unordered_map<int64_t, int> iso_at;
    for (int64_t i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        Ind x;
        x.ID = ind[i];
        x.rank = static_cast<int>(iso.size()) + 1;
        x.father = father[i] != 0 ? iso_at[father[i]] : -1;
        x.mother = mother[i] != 0 ? iso_at[mother[i]] : -1;
        iso_at[x.ID] = static_cast<int>(iso.size());
        iso.push_back(x);
    }
    /* _index_pedigree, src/compute.jl:165-186: children in pedigree order */
    for (size_t i = 0; i < iso.size(); ++i) {
        if (iso[i].father >= 0) iso[iso[i].father].children.push_back(static_cast<int>(i));
        if (iso[i].mother >= 0) iso[iso[i].mother].children.push_back(static_cast<int>(i));
    }
    for (int64_t k = 0; k < n_pro; ++k) iso[iso_at[pro[k]]].is_proband = true;      /* :325-327 */

    Sparse *S = new Sparse();
    auto &phi = S->dict;
    std::unordered_set<int> ranks_to_visit;
    std::deque<int> queue;
    {   /* founder(isolated_pedigree): IDs ascending (src/identify.jl:15-19), :336-339 */
        std::vector<std::pair<int64_t, int>> f;
        for (size_t i = 0; i < iso.size(); ++i)
            if (iso[i].father < 0 && iso[i].mother < 0) f.emplace_back(iso[i].ID, static_cast<int>(i));
        std::sort(f.begin(), f.end());
        for (auto &e : f) queue.push_back(e.second);
    }
    while (!queue.empty()) {
        const int xi = queue.front(); queue.pop_front();
        Ind &I = iso[xi];
        S->order_ids.push_back(I.ID);
        const int rank_i = I.rank;
        const int father_rank = I.father < 0 ? 0 : iso[I.father].rank;
        const int mother_rank = I.mother < 0 ? 0 : iso[I.mother].rank;
        phi[rank_i] = std::unordered_map<int, float>();                               /* :348 */
        /* kinship with self, :349-361 */
        double coefficient = 0.5;
        if (father_rank != 0 && mother_rank != 0) {
            if (father_rank < mother_rank) {
                auto it = phi[father_rank].find(mother_rank);
                if (it != phi[father_rank].end()) coefficient += static_cast<double>(it->second / 2.0f);
            } else {
                auto it = phi[mother_rank].find(father_rank);
                if (it != phi[mother_rank].end()) coefficient += static_cast<double>(it->second / 2.0f);
            }
        }
        phi[rank_i][rank_i] = static_cast<float>(coefficient);
        /* kinship with previous individuals, :363-395 (each rank_j is independent of the others, so
         * the iteration order of the Set does not matter) */
        for (int rank_j : ranks_to_visit) {
            coefficient = 0.0;
            if (father_rank != 0) {
                if (rank_j < father_rank) {
                    auto it = phi[rank_j].find(father_rank);
                    if (it != phi[rank_j].end()) coefficient += static_cast<double>(it->second / 2.0f);
                } else {
                    auto it = phi[father_rank].find(rank_j);
                    if (it != phi[father_rank].end()) coefficient += static_cast<double>(it->second / 2.0f);
                }
            }
            if (mother_rank != 0) {
                if (rank_j < mother_rank) {
                    auto it = phi[rank_j].find(mother_rank);
                    if (it != phi[rank_j].end()) coefficient += static_cast<double>(it->second / 2.0f);
                } else {
                    auto it = phi[mother_rank].find(rank_j);
                    if (it != phi[mother_rank].end()) coefficient += static_cast<double>(it->second / 2.0f);
                }
            }
            if (coefficient > 0.0) phi[rank_j][rank_i] = static_cast<float>(coefficient);   /* :392-394: key = (earlier, later) */
        }
        ranks_to_visit.insert(rank_i);                                                /* :397 */
        I.founder_index = 1;
        I.children_to_process = static_cast<int>(I.children.size());
        /* retire parents whose children are all processed, :401-430 */
        for (int side = 0; side < 2; ++side) {
            const int pidx = side == 0 ? I.father : I.mother;
            const int prank = side == 0 ? father_rank : mother_rank;
            if (prank == 0) continue;
            Ind &P = iso[pidx];
            if (P.is_proband) continue;
            P.children_to_process -= 1;
            if (P.children_to_process == 0) {
                S->retired_ids.push_back(P.ID);
                S->retired_at.push_back(static_cast<int64_t>(S->order_ids.size()) - 1);
                ranks_to_visit.erase(prank);
                phi.erase(prank);
                for (int rank_j : ranks_to_visit)
                    if (rank_j < prank) phi[rank_j].erase(prank);
            }
        }
        /* children whose parents are both done, :431-439 */
        for (int c : I.children) {
            const Ind &Cc = iso[c];
            if (Cc.father >= 0 && Cc.mother >= 0) {
                if (iso[Cc.father].founder_index != 0 && iso[Cc.mother].founder_index != 0) queue.push_back(c);
            } else {
                queue.push_back(c);
            }
        }
    }
    for (int64_t k = 0; k < n_pro; ++k) S->id_to_rank[pro[k]] = iso[iso_at[pro[k]]].rank;   /* :442-445 */
    return S;
}

void sparse_oracle_free(void *h) { delete static_cast<Sparse *>(h); }

/* getindex(ϕ::KinshipMatrix, ID1, ID2), src/compute.jl:36-40; -1 for an ID that is not a proband */
double sparse_oracle_get(void *h, int64_t id1, int64_t id2)
{
    Sparse *S = static_cast<Sparse *>(h);
    auto a = S->id_to_rank.find(id1), b = S->id_to_rank.find(id2);
    if (a == S->id_to_rank.end() || b == S->id_to_rank.end()) return -1.0;
    int r1 = a->second, r2 = b->second;
    if (r1 > r2) std::swap(r1, r2);
    auto row = S->dict.find(r1);
    if (row == S->dict.end()) return -1.0;                 /* KeyError in the reference */
    auto it = row->second.find(r2);
    return it == row->second.end() ? 0.0 : static_cast<double>(it->second);
}

/* what `show` prints (:42-46): rows, stored entries; and the sums phiMean uses (:467-472), in Float64 */
void sparse_oracle_info(void *h, int64_t *n_rows, int64_t *n_stored, double *sum_all, double *sum_diag)
{
    Sparse *S = static_cast<Sparse *>(h);
    int64_t nz = 0;
    double tot = 0.0, dg = 0.0;
    for (auto &row : S->dict) {
        nz += static_cast<int64_t>(row.second.size());
        for (auto &e : row.second) tot += static_cast<double>(e.second);
        dg += static_cast<double>(row.second.at(row.first));
    }
    *n_rows = static_cast<int64_t>(S->dict.size());
    *n_stored = nz;
    *sum_all = tot;
    *sum_diag = dg;
}

/* processing order (IDs in the order they left the queue) */
/* the individuals dropped from the live set (:401-430) and the 0-based processing index of the child that dropped each */
int64_t sparse_oracle_retired(void *h, int64_t *ids, int64_t *at, int64_t cap)
{
    Sparse *S = static_cast<Sparse *>(h);
    const int64_t n = static_cast<int64_t>(S->retired_ids.size());
    for (int64_t k = 0; k < n && k < cap; ++k) { ids[k] = S->retired_ids[k]; at[k] = S->retired_at[k]; }
    return n;
}

int64_t sparse_oracle_order(void *h, int64_t *out, int64_t cap)
{
    Sparse *S = static_cast<Sparse *>(h);
    const int64_t n = static_cast<int64_t>(S->order_ids.size());
    for (int64_t k = 0; k < n && k < cap; ++k) out[k] = S->order_ids[k];
    return n;
}

/* every stored entry as (row rank, column rank, value); returns the count */
int64_t sparse_oracle_entries(void *h, int64_t *row_rank, int64_t *col_rank, float *val, int64_t cap)
{
    Sparse *S = static_cast<Sparse *>(h);
    int64_t k = 0;
    for (auto &row : S->dict)
        for (auto &e : row.second) {
            if (k < cap) { row_rank[k] = row.first; col_rank[k] = e.first; val[k] = e.second; }
            ++k;
        }
    return k;
}

}  // extern "C"
