// loader.cpp -- native pedigree loader and pruning: the steps BEFORE the gen.phi hot path
// (genealogy(filename) and branching).
//
// Replaces, for callers that want it, the reference's genealogy(filename; sort)
// (src/create.jl:161-189: header line skipped, four whitespace-separated Ints per row) followed
// by _ordered_pedigree (src/create.jl:196-227: stable sort by maximum ancestral depth, founders
// depth 1).  The output is exactly what genphi_plan_create wants: ind / father / mother (/ sex)
// in rank order.  Own design: one pass over a memory-resident buffer with a hand-rolled integer
// scanner, iterative depth computation (no recursion: depth-1e6 chains are fine), counting sort.
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/genphi.h"

extern int genphi_set_error(int code, const std::string &msg);   // genphi_hip.hip

namespace {

struct Cols { std::vector<int64_t> ind, father, mother, sex; };

bool parse_tsv(const char *buf, size_t len, Cols &c, std::string &err)
{
    size_t i = 0;
    while (i < len && buf[i] != '\n') ++i;          // header row
    if (i < len) ++i;
    size_t line = 1;
    while (i < len) {
        ++line;
        int64_t v[4];
        int got = 0;
        while (i < len && buf[i] != '\n') {
            while (i < len && (buf[i] == ' ' || buf[i] == '\t' || buf[i] == '\r')) ++i;
            if (i >= len || buf[i] == '\n') break;
            bool neg = false;
            if (buf[i] == '-' || buf[i] == '+') { neg = buf[i] == '-'; ++i; }
            if (i >= len || buf[i] < '0' || buf[i] > '9') { err = "line " + std::to_string(line) + ": not an integer"; return false; }
            int64_t x = 0;
            while (i < len && buf[i] >= '0' && buf[i] <= '9') { x = x * 10 + (buf[i] - '0'); ++i; }
            if (got < 4) v[got] = neg ? -x : x;
            ++got;
        }
        if (i < len) ++i;                            // newline
        if (got == 0) continue;                      // blank line
        if (got != 4) { err = "line " + std::to_string(line) + ": expected 4 columns (ind father mother sex), got " + std::to_string(got); return false; }
        c.ind.push_back(v[0]); c.father.push_back(v[1]); c.mother.push_back(v[2]); c.sex.push_back(v[3]);
    }
    return true;
}

}  // namespace

// Everything after the table is in memory: ID lookup, checks, depth sort (src/create.jl:196-254), output arrays in rank order.
static int order_and_emit(const Cols &c, int32_t sort, int64_t *n_out, int64_t **ind_out, int64_t **father_out, int64_t **mother_out,
                          int64_t **sex_out)
{
    const int64_t n = static_cast<int64_t>(c.ind.size());

    // file position of every id; parents must exist (KeyError in the reference).  IDs in a moderate range go through a direct table
    // (1e6 dense IDs: 10 ms instead of 150 ms of hashing), anything else through a hash map.
    std::vector<int32_t> table;
    std::unordered_map<int64_t, int32_t> pos;
    bool direct = false;
    {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int64_t i = 0; i < n; ++i) { lo = c.ind[i] < lo ? c.ind[i] : lo; hi = c.ind[i] > hi ? c.ind[i] : hi; }
        if (n > 0 && lo >= 0 && hi < 64 * n + (1 << 20)) { table.assign(static_cast<size_t>(hi) + 1, -1); direct = true; }
        else pos.reserve(static_cast<size_t>(n) * 2);
    }
    auto find = [&](int64_t id) -> int32_t {
        if (direct) return (id < 0 || id >= static_cast<int64_t>(table.size())) ? -1 : table[id];
        auto it = pos.find(id);
        return it == pos.end() ? -1 : it->second;
    };
    for (int64_t i = 0; i < n; ++i) {
        bool fresh;
        if (direct) { fresh = table[c.ind[i]] < 0; if (fresh) table[c.ind[i]] = static_cast<int32_t>(i); }
        else fresh = pos.emplace(c.ind[i], static_cast<int32_t>(i)).second;
        if (!fresh) return genphi_set_error(GENPHI_ERR_DUPLICATE_ID, "duplicate individual ID " + std::to_string(c.ind[i]));
    }
    std::vector<int32_t> pf(n, -1), pm(n, -1);
    for (int64_t i = 0; i < n; ++i) {
        if (c.father[i] != 0) {
            pf[i] = find(c.father[i]);
            if (pf[i] < 0) return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: father " + std::to_string(c.father[i]) + " of " + std::to_string(c.ind[i]) + " not found");
        }
        if (c.mother[i] != 0) {
            pm[i] = find(c.mother[i]);
            if (pm[i] < 0) return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: mother " + std::to_string(c.mother[i]) + " of " + std::to_string(c.ind[i]) + " not found");
        }
    }
    std::vector<int64_t> order(n);
    if (sort) {
        // depth(x) = 1 + max(depth(father), depth(mother)); explicit stack; a cycle is an error
        std::vector<int32_t> depth(n, 0), state(n, 0), stack;
        for (int64_t s0 = 0; s0 < n; ++s0) {
            if (depth[s0]) continue;
            stack.push_back(static_cast<int32_t>(s0));
            while (!stack.empty()) {
                const int32_t x = stack.back();
                const int32_t f = pf[x], m = pm[x];
                state[x] = 1;
                if (f >= 0 && !depth[f]) { if (state[f]) return genphi_set_error(GENPHI_ERR_ARG, "pedigree contains a cycle"); stack.push_back(f); continue; }
                if (m >= 0 && !depth[m]) { if (state[m]) return genphi_set_error(GENPHI_ERR_ARG, "pedigree contains a cycle"); stack.push_back(m); continue; }
                const int32_t fd = f >= 0 ? depth[f] : 0, md = m >= 0 ? depth[m] : 0;
                depth[x] = (fd > md ? fd : md) + 1;
                stack.pop_back();
            }
        }
        int32_t maxd = 0;
        for (int64_t i = 0; i < n; ++i) maxd = depth[i] > maxd ? depth[i] : maxd;
        std::vector<int64_t> cnt(static_cast<size_t>(maxd) + 2, 0);
        for (int64_t i = 0; i < n; ++i) cnt[depth[i] + 1]++;
        for (int32_t d = 1; d <= maxd + 1; ++d) cnt[d] += cnt[d - 1];
        for (int64_t i = 0; i < n; ++i) order[cnt[depth[i]]++] = i;         // stable: file order on ties
    } else {
        for (int64_t i = 0; i < n; ++i) {
            order[i] = i;
            if ((pf[i] >= 0 && pf[i] >= i) || (pm[i] >= 0 && pm[i] >= i))
                return genphi_set_error(GENPHI_ERR_ORDER, "KeyError: a parent of " + std::to_string(c.ind[i]) + " is listed after it (sort=false)");
        }
    }
    auto emit = [&](const std::vector<int64_t> &src) -> int64_t * {
        int64_t *dst = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * static_cast<size_t>(n > 0 ? n : 1)));
        if (dst) for (int64_t k = 0; k < n; ++k) dst[k] = src[order[k]];
        return dst;
    };
    int64_t *a = emit(c.ind), *b = emit(c.father), *d = emit(c.mother), *e = (sex_out && c.sex.size() == c.ind.size()) ? emit(c.sex) : nullptr;
    if (!a || !b || !d || (sex_out && c.sex.size() == c.ind.size() && !e)) { std::free(a); std::free(b); std::free(d); std::free(e); return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory"); }
    *n_out = n; *ind_out = a; *father_out = b; *mother_out = d;
    if (sex_out) *sex_out = e;
    return GENPHI_OK;
}


extern "C" {

void genphi_free(void *ptr) { std::free(ptr); }

int genphi_genealogy_read(const char *path, int32_t sort, int64_t *n_out, int64_t **ind_out, int64_t **father_out,
                          int64_t **mother_out, int64_t **sex_out)
{
    if (!path || !n_out || !ind_out || !father_out || !mother_out) return genphi_set_error(GENPHI_ERR_ARG, "genphi_genealogy_read: NULL argument");
    *n_out = 0; *ind_out = *father_out = *mother_out = nullptr;
    if (sex_out) *sex_out = nullptr;
    std::FILE *fh = std::fopen(path, "rb");
    if (!fh) return genphi_set_error(GENPHI_ERR_ARG, std::string("cannot open ") + path);
    std::string buf;
    {
        std::fseek(fh, 0, SEEK_END);
        const long sz = std::ftell(fh);
        std::fseek(fh, 0, SEEK_SET);
        buf.resize(sz > 0 ? static_cast<size_t>(sz) : 0);
        const size_t rd = buf.empty() ? 0 : std::fread(&buf[0], 1, buf.size(), fh);
        std::fclose(fh);
        if (rd != buf.size()) return genphi_set_error(GENPHI_ERR_ARG, std::string("short read on ") + path);
    }
    Cols c;
    std::string err;
    if (!parse_tsv(buf.data(), buf.size(), c, err)) return genphi_set_error(GENPHI_ERR_ARG, std::string(path) + ": " + err);
    return order_and_emit(c, sort, n_out, ind_out, father_out, mother_out, sex_out);
}

/* The same for a table already in memory (gen.genealogy(dataframe; sort), src/create.jl:131-146 + :196-254): ind / father / mother
 * (/ sex, may be NULL) in file order -> rank order.  Errors as genphi_genealogy_read: duplicate ID, unknown parent (KeyError in the
 * reference), a cycle, and with sort = 0 a parent listed after its child (KeyError). */
int genphi_genealogy_order(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother, const int64_t *sex, int32_t sort,
                           int64_t *n_out, int64_t **ind_out, int64_t **father_out, int64_t **mother_out, int64_t **sex_out)
{
    if (n_ind < 0 || (n_ind > 0 && (!ind || !father || !mother)) || !n_out || !ind_out || !father_out || !mother_out)
        return genphi_set_error(GENPHI_ERR_ARG, "genphi_genealogy_order: NULL or negative argument");
    *n_out = 0; *ind_out = *father_out = *mother_out = nullptr;
    if (sex_out) *sex_out = nullptr;
    Cols c;
    c.ind.assign(ind, ind + n_ind); c.father.assign(father, father + n_ind); c.mother.assign(mother, mother + n_ind);
    if (sex) c.sex.assign(sex, sex + n_ind);
    return order_and_emit(c, sort, n_out, ind_out, father_out, mother_out, sex_out);
}

// gen.branching (src/extract.jl:65-186).  The reference marks ancestors / descendants with two
// recursive walks over a pointer graph; here the pedigree is already in rank order (parents
// before children), so both marks are single linear passes: ancestors in one reverse sweep
// (a marked child marks its parents), descendants in one forward sweep (a marked parent marks
// its children).  No recursion, no per-individual allocation, O(N).
int genphi_branching(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
                     const int64_t *sex, int64_t n_pro, const int64_t *pro, int64_t n_anc,
                     const int64_t *ancestors, int64_t *n_out, int64_t **ind_out, int64_t **father_out,
                     int64_t **mother_out, int64_t **sex_out)
{
    if (!n_out || !ind_out || !father_out || !mother_out || n_ind < 0 || (n_ind > 0 && (!ind || !father || !mother)) ||
        n_pro < 0 || n_anc < 0)
        return genphi_set_error(GENPHI_ERR_ARG, "genphi_branching: bad argument");
    *n_out = 0; *ind_out = *father_out = *mother_out = nullptr;
    if (sex_out) *sex_out = nullptr;
    const int64_t n = n_ind;
    std::unordered_map<int64_t, int64_t> pos;
    pos.reserve(static_cast<size_t>(n) * 2);
    for (int64_t i = 0; i < n; ++i)
        if (!pos.emplace(ind[i], i).second)
            return genphi_set_error(GENPHI_ERR_DUPLICATE_ID, "duplicate individual ID " + std::to_string(ind[i]));
    std::vector<int64_t> pf(n, -1), pm(n, -1);
    for (int64_t i = 0; i < n; ++i) {
        for (int side = 0; side < 2; ++side) {
            const int64_t pid = side ? mother[i] : father[i];
            if (pid == 0) continue;
            auto it = pos.find(pid);
            if (it == pos.end()) return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: parent " + std::to_string(pid) + " of " + std::to_string(ind[i]) + " not found");
            if (it->second >= i) return genphi_set_error(GENPHI_ERR_ORDER, "KeyError: a parent of " + std::to_string(ind[i]) + " is listed after it");
            (side ? pm : pf)[i] = it->second;
        }
    }
    std::vector<uint8_t> is_anc(n, 0), is_desc(n, 0);
    if (pro) {
        for (int64_t k = 0; k < n_pro; ++k) {
            auto it = pos.find(pro[k]);
            if (it == pos.end()) return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: proband " + std::to_string(pro[k]) + " not found");
            is_anc[it->second] = 1;
        }
        for (int64_t i = n - 1; i >= 0; --i)
            if (is_anc[i]) {
                if (pf[i] >= 0) is_anc[pf[i]] = 1;
                if (pm[i] >= 0) is_anc[pm[i]] = 1;
            }
    }
    if (ancestors) {
        for (int64_t k = 0; k < n_anc; ++k) {
            auto it = pos.find(ancestors[k]);
            if (it == pos.end()) return genphi_set_error(GENPHI_ERR_UNKNOWN_ID, "KeyError: ancestor " + std::to_string(ancestors[k]) + " not found");
            is_desc[it->second] = 1;
        }
        for (int64_t i = 0; i < n; ++i)
            if (!is_desc[i] && ((pf[i] >= 0 && is_desc[pf[i]]) || (pm[i] >= 0 && is_desc[pm[i]]))) is_desc[i] = 1;
    }
    // kept set; a parent outside it becomes unknown (only possible when `ancestors` is given)
    std::vector<uint8_t> keep(n, 0);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        keep[i] = (pro && ancestors) ? (is_anc[i] && is_desc[i]) : (pro ? is_anc[i] : (ancestors ? is_desc[i] : 0));
        m += keep[i];
    }
    const size_t bytes = sizeof(int64_t) * static_cast<size_t>(m > 0 ? m : 1);
    int64_t *a = static_cast<int64_t *>(std::malloc(bytes)), *b = static_cast<int64_t *>(std::malloc(bytes));
    int64_t *d = static_cast<int64_t *>(std::malloc(bytes)), *e = sex_out ? static_cast<int64_t *>(std::malloc(bytes)) : nullptr;
    if (!a || !b || !d || (sex_out && !e)) { std::free(a); std::free(b); std::free(d); std::free(e); return genphi_set_error(GENPHI_ERR_ALLOC, "out of memory"); }
    int64_t o = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        a[o] = ind[i];
        b[o] = (pf[i] >= 0 && keep[pf[i]]) ? father[i] : 0;
        d[o] = (pm[i] >= 0 && keep[pm[i]]) ? mother[i] : 0;
        if (e) e[o] = sex ? sex[i] : 0;
        ++o;
    }
    *n_out = m; *ind_out = a; *father_out = b; *mother_out = d;
    if (sex_out) *sex_out = e;
    return GENPHI_OK;
}

}  // extern "C"
