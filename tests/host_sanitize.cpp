// Host-side sanitizer run ("use -fsanitize=address for the host planner"): the native loader and the planner, the only
// host code on the gen.phi path that indexes by pedigree data, driven over the bundled genealogies and over random pedigrees with
// every planner option, under AddressSanitizer + UndefinedBehaviorSanitizer.  Built and run by tests/test_host_sanitizers.py (CPU
// only; GPU sanitizers are not available on this pool).  Links planner.cpp and loader.cpp directly -- no HIP, no oracle.
//   usage: host_sanitize <genea140.csv> <geneaJi.csv>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../genlib.jl_amd/csrc/planner.h"
#include "../include/genphi.h"

static std::string g_err;
int genphi_set_error(int code, const std::string &msg) { g_err = msg; return code; }      // (lives in genphi_hip.hip in the library)

using genphi::Plan;
using genphi::PlanOptions;

static unsigned long long checksum(const Plan &p)
{
    unsigned long long s = static_cast<unsigned long long>(p.n_levels);
    for (int64_t c : p.cut_sizes) s = s * 31u + static_cast<unsigned long long>(c);
    for (const genphi::LevelStep &st : p.steps) {
        s = s * 31u + static_cast<unsigned long long>(st.n + st.n_prev + st.mode);
        for (int32_t v : st.srcA) s += static_cast<unsigned>(v);
        for (int32_t v : st.srcB) s += static_cast<unsigned>(v);
        for (int32_t v : st.work) s += static_cast<unsigned>(v);
    }
    for (int32_t v : p.final_members) s += static_cast<unsigned>(v);
    for (int32_t v : p.final_perm) s += static_cast<unsigned>(v);
    return s;
}

static int plan_all_ways(const std::vector<int64_t> &ind, const std::vector<int64_t> &fa, const std::vector<int64_t> &mo,
                         const std::vector<int64_t> &pro, const char *what)
{
    unsigned long long sum = 0;
    int n_ok = 0;
    for (int variant = 0; variant < 7; ++variant) {
        PlanOptions o;
        if (variant == 1) o.indices_only = true;
        if (variant == 2) o.no_stay = true;
        if (variant == 3) { o.lds_cap_floats = 256; o.full_max_floats = 0; }
        if (variant == 4) { o.stay_scatter = true; o.stay_min_ratio_pct = 110; o.stay_mem_ratio = 1000.0; }
        if (variant == 5) { o.stay_narrow = false; o.full_max_floats = 64; }
        if (variant == 6) { o.lds_cap_floats = 700; o.stay_slack_pct = 0; o.stay_mem_ratio = 1000.0; }
        Plan plan;
        std::string err;
        const int rc = genphi::build_plan(static_cast<int64_t>(ind.size()), ind.data(), fa.data(), mo.data(), static_cast<int64_t>(pro.size()),
                                          pro.data(), o, plan, err);
        if (rc != GENPHI_OK) { std::fprintf(stderr, "%s, variant %d: build_plan failed: %s\n", what, variant, err.c_str()); return 1; }
        sum += checksum(plan);
        // the hub walk of every SPLIT step (what the upload builds from the plan)
        for (const genphi::LevelStep &st : plan.steps) {
            if (st.mode != genphi::kModeSplit || st.work.empty()) continue;
            genphi::WalkLists w;
            std::vector<int> rows(st.work.begin(), st.work.end());
            genphi::build_hub_walk(st.srcA.data(), st.srcB.data(), st.ord.data(), static_cast<int32_t>(st.n_prev), rows.data(), nullptr,
                                   static_cast<int>(rows.size()), 4, 1, w);
            for (int32_t v : w.desc4) sum += static_cast<unsigned>(v);
            for (int32_t v : w.run) sum += static_cast<unsigned>(v);
        }
        ++n_ok;
    }
    std::printf("%s: %d plans, checksum %llu\n", what, n_ok, sum);
    return 0;
}

static std::vector<int64_t> probands_of(const std::vector<int64_t> &ind, const std::vector<int64_t> &fa, const std::vector<int64_t> &mo)
{
    std::vector<int64_t> parents(fa);
    parents.insert(parents.end(), mo.begin(), mo.end());
    std::sort(parents.begin(), parents.end());
    std::vector<int64_t> pro;
    for (int64_t x : ind) if (!std::binary_search(parents.begin(), parents.end(), x)) pro.push_back(x);
    return pro;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: host_sanitize genea140.csv geneaJi.csv\n"); return 2; }
    int bad = 0;
    for (int f = 1; f <= 2; ++f)
        for (int sort = 0; sort <= 1; ++sort) {
            int64_t n = 0, *ind = nullptr, *fa = nullptr, *mo = nullptr, *sex = nullptr;
            const int rrc = genphi_genealogy_read(argv[f], sort, &n, &ind, &fa, &mo, &sex);
            if (rrc == GENPHI_ERR_ORDER && sort == 0) { std::printf("%s, sort = false: %s (expected: the file lists a child before a parent)\n", argv[f], g_err.c_str()); continue; }
            if (rrc != GENPHI_OK) { std::fprintf(stderr, "read %s: %s\n", argv[f], g_err.c_str()); return 1; }
            std::vector<int64_t> vi(ind, ind + n), vf(fa, fa + n), vm(mo, mo + n), vs(sex, sex + n);
            genphi_free(ind); genphi_free(fa); genphi_free(mo); genphi_free(sex);
            const std::vector<int64_t> pro = probands_of(vi, vf, vm);
            bad |= plan_all_ways(vi, vf, vm, pro, argv[f]);
            // a share of all individuals as probands (ancestors of other probands among them), duplicates, reverse order
            std::vector<int64_t> some;
            for (int64_t k = 0; k < n; k += 7) some.push_back(vi[n - 1 - k]);
            some.push_back(some.front());
            bad |= plan_all_ways(vi, vf, vm, some, "  every seventh individual as a proband");
            // the table again through genphi_genealogy_order, and a branching on the first probands and their founders
            int64_t n2 = 0, *i2 = nullptr, *f2 = nullptr, *m2 = nullptr, *s2 = nullptr;
            if (genphi_genealogy_order(n, vi.data(), vf.data(), vm.data(), vs.data(), 1, &n2, &i2, &f2, &m2, &s2) != GENPHI_OK || n2 != n) { std::fprintf(stderr, "order: %s\n", g_err.c_str()); return 1; }
            genphi_free(i2); genphi_free(f2); genphi_free(m2); genphi_free(s2);
            std::vector<int64_t> few(pro.begin(), pro.begin() + std::min<size_t>(pro.size(), 5));
            if (genphi_branching(n, vi.data(), vf.data(), vm.data(), vs.data(), static_cast<int64_t>(few.size()), few.data(), 0, nullptr, &n2, &i2, &f2, &m2, &s2) != GENPHI_OK) {
                std::fprintf(stderr, "branching: %s\n", g_err.c_str()); return 1;
            }
            std::vector<int64_t> bi(i2, i2 + n2), bf(f2, f2 + n2), bm(m2, m2 + n2);
            genphi_free(i2); genphi_free(f2); genphi_free(m2); genphi_free(s2);
            bad |= plan_all_ways(bi, bf, bm, few, "  branching on five probands");
        }
    // error paths: unknown parent, duplicate ID, unknown proband, a cycle
    {
        const std::vector<int64_t> i{1, 2, 3}, fdup{0, 0, 1}, m0{0, 0, 2};
        int64_t n2 = 0, *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr;
        const std::vector<int64_t> idup{1, 1, 3}, funk{0, 0, 9}, fcyc{3, 0, 1};
        if (genphi_genealogy_order(3, idup.data(), fdup.data(), m0.data(), nullptr, 1, &n2, &a, &b, &c, &d) == GENPHI_OK) { std::fprintf(stderr, "duplicate ID accepted\n"); bad = 1; }
        if (genphi_genealogy_order(3, i.data(), funk.data(), m0.data(), nullptr, 1, &n2, &a, &b, &c, &d) == GENPHI_OK) { std::fprintf(stderr, "unknown father accepted\n"); bad = 1; }
        if (genphi_genealogy_order(3, i.data(), fcyc.data(), m0.data(), nullptr, 1, &n2, &a, &b, &c, &d) == GENPHI_OK) { std::fprintf(stderr, "cycle accepted\n"); bad = 1; }
        Plan plan; std::string err;
        const std::vector<int64_t> pro_unk{7};
        if (genphi::build_plan(3, i.data(), fdup.data(), m0.data(), 1, pro_unk.data(), PlanOptions(), plan, err) == GENPHI_OK) { std::fprintf(stderr, "unknown proband accepted\n"); bad = 1; }
        Plan empty;
        if (genphi::build_plan(3, i.data(), fdup.data(), m0.data(), 0, nullptr, PlanOptions(), empty, err) != GENPHI_OK) { std::fprintf(stderr, "no probands: %s\n", err.c_str()); bad = 1; }
    }
    // random pedigrees: overlapping generations, one-parent individuals, probands at every depth
    std::mt19937_64 rng(20261005);
    for (int rep = 0; rep < 24; ++rep) {
        const int gens = 2 + static_cast<int>(rng() % 14), per = 3 + static_cast<int>(rng() % 400);
        const int skip = static_cast<int>(rng() % 400);
        std::vector<int64_t> ind, fa, mo;
        std::vector<std::vector<int64_t>> g(gens);
        int64_t next = 1;
        for (int k = 0; k < gens; ++k)
            for (int q = 0; q < per; ++q) {
                const int64_t id = next++ * 3 + 1;                       // (IDs that are not positions)
                int64_t f = 0, m = 0;
                if (k > 0) {
                    const int kf = (k >= 2 && static_cast<int>(rng() % 1000) < skip) ? k - 2 : k - 1;
                    const int km = (k >= 2 && static_cast<int>(rng() % 1000) < skip) ? k - 2 : k - 1;
                    f = g[kf][rng() % g[kf].size()]; m = g[km][rng() % g[km].size()];
                    const unsigned u = static_cast<unsigned>(rng() % 100);
                    if (u < 4) f = 0; else if (u < 8) m = 0; else if (u < 10) f = m = 0;
                }
                ind.push_back(id); fa.push_back(f); mo.push_back(m);
                g[k].push_back(id);
            }
        std::vector<int64_t> pro(g[gens - 1]);
        for (int q = 0; q < 6; ++q) pro.push_back(ind[rng() % ind.size()]);
        bad |= plan_all_ways(ind, fa, mo, pro, "random pedigree");
    }
    std::printf(bad ? "FAILED\n" : "host sanitizer run: ok\n");
    return bad;
}
