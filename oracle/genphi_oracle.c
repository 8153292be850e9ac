/*
 * genphi_oracle.c -- CPU ORACLE for the gen.phi hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm (GenLib.jl v0.1.4,
 * /root/reference).  It is NOT part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and there only as the checker / the
 * reported CPU baseline.  The product path (genlib.jl_amd/csrc) never links or calls it.
 *
 * Parity pinning: checked in tests/test_oracle_golden.py against every value the
 * reference's own test-suite holds for this path (test/runtests.jl:41,42,47-53,58-60)
 * on the bundled data/geneaJi.csv, and against the survey-derived genea140 values
 * (flagged there as not reference-pinned).  The reference is
 * Julia-only and there is no Julia in the build container, so it could not be run here.
 *
 * Every function cites the reference file:line it follows.  The restatement is literal
 * on purpose (same recursion, same branch order, same Float64 accumulator, same
 * Float64 -> Float32 store per level, founder_index never reset) so that it can be
 * trusted as "what the reference computes", not as a fast implementation.
 *
 * Individuals are addressed by their 0-based position in the rank-ordered pedigree
 * (rank = position + 1, src/create.jl:234-254), so `rank_i > rank_j` is `i > j`.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_UNKNOWN_ID 1     /* KeyError in the reference */
#define ORACLE_ERR_ORDER 2          /* parent after child: KeyError in _finalize_pedigree */
#define ORACLE_ERR_ALLOC 3
#define ORACLE_ERR_DUP_ID 4

/* ------------------------------------------------------------------------------------ */
/* id -> position map (open addressing); containers only, no reference arithmetic here. */
typedef struct { int64_t *keys; int32_t *vals; uint64_t mask; } idmap;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static int idmap_init(idmap *m, int64_t n) {
    uint64_t cap = 16; while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
    m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
    m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->keys || !m->vals) return ORACLE_ERR_ALLOC;
    for (uint64_t i = 0; i < cap; i++) m->vals[i] = -1;
    m->mask = cap - 1; return ORACLE_OK;
}
static void idmap_free(idmap *m) { free(m->keys); free(m->vals); }
static int idmap_put(idmap *m, int64_t k, int32_t v) {
    uint64_t h = mix64((uint64_t)k) & m->mask;
    while (m->vals[h] >= 0) { if (m->keys[h] == k) return ORACLE_ERR_DUP_ID; h = (h + 1) & m->mask; }
    m->keys[h] = k; m->vals[h] = v; return ORACLE_OK;
}
static int32_t idmap_get(const idmap *m, int64_t k) {
    uint64_t h = mix64((uint64_t)k) & m->mask;
    while (m->vals[h] >= 0) { if (m->keys[h] == k) return m->vals[h]; h = (h + 1) & m->mask; }
    return -1;
}

/* ------------------------------------------------------------------------------------ */
/* src/create.jl:196-227  _max_depth! + _ordered_pedigree.
 * In : n individuals in FILE order (ind, father, mother; 0 = unknown parent).
 * Out: order[k] = file position of the individual that gets rank k+1
 *      (stable sort of max ancestral depth; founders have depth 1).
 * The reference recursion is memoised; restated iteratively with an explicit stack so
 * that a 1e6-deep chain cannot overflow the C stack (same values). */
int oracle_rank_order(int64_t n, const int64_t *ind, const int64_t *father,
                      const int64_t *mother, int64_t *order)
{
    idmap m; int rc = idmap_init(&m, n); if (rc) return rc;
    for (int64_t i = 0; i < n; i++) { rc = idmap_put(&m, ind[i], (int32_t)i); if (rc) { idmap_free(&m); return rc; } }
    int32_t *depth = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    int32_t *pf = (int32_t *)malloc(n * sizeof(int32_t)), *pm = (int32_t *)malloc(n * sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) {
        depth[i] = -1;
        pf[i] = father[i] != 0 ? idmap_get(&m, father[i]) : -1;
        pm[i] = mother[i] != 0 ? idmap_get(&m, mother[i]) : -1;
        if ((father[i] != 0 && pf[i] < 0) || (mother[i] != 0 && pm[i] < 0)) {
            free(depth); free(stack); free(pf); free(pm); idmap_free(&m); return ORACLE_ERR_UNKNOWN_ID;
        }
    }
    for (int64_t s = 0; s < n; s++) {
        if (depth[s] != -1) continue;
        int64_t sp = 0; stack[sp++] = (int32_t)s;
        while (sp > 0) {
            int32_t x = stack[sp - 1];
            int32_t f = pf[x], mo = pm[x];
            if (f >= 0 && depth[f] == -1) { stack[sp++] = f; continue; }
            if (mo >= 0 && depth[mo] == -1) { stack[sp++] = mo; continue; }
            int32_t fd = f >= 0 ? depth[f] : 0, md = mo >= 0 ? depth[mo] : 0;   /* :198-206 */
            depth[x] = (fd > md ? fd : md) + 1; sp--;
        }
    }
    /* stable sortperm(depths) (:220): counting sort by depth keeps file order on ties */
    int32_t maxd = 0; for (int64_t i = 0; i < n; i++) if (depth[i] > maxd) maxd = depth[i];
    int64_t *cnt = (int64_t *)calloc((size_t)maxd + 2, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) cnt[depth[i] + 1]++;
    for (int32_t d = 1; d <= maxd + 1; d++) cnt[d] += cnt[d - 1];
    for (int64_t i = 0; i < n; i++) order[cnt[depth[i]]++] = i;
    free(cnt); free(depth); free(stack); free(pf); free(pm); idmap_free(&m);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------ */
/* The rank-ordered pedigree the hot path consumes (src/create.jl:234-254).
 * father/mother are positions (0-based rank index) or -1. */
typedef struct {
    int64_t n;
    int64_t *id;
    int32_t *father, *mother;
    int32_t *nchildren;
    idmap map;
    /* IndexedIndividual.founder_index (src/compute.jl:14-24,165-186): 0 = not in the
     * previous cut; 1-based otherwise; NEVER reset between levels (:287-289). */
    int32_t *founder_index;
} oracle_ped;

void oracle_ped_free(oracle_ped *p) {
    if (!p) return;
    free(p->id); free(p->father); free(p->mother); free(p->nchildren); free(p->founder_index);
    idmap_free(&p->map); free(p);
}

/* src/create.jl:234-254 _finalize_pedigree: individuals given IN RANK ORDER; a parent that
 * has not been inserted yet is a KeyError there -> ORACLE_ERR_ORDER / UNKNOWN_ID here. */
int oracle_ped_create(int64_t n, const int64_t *ind, const int64_t *father,
                      const int64_t *mother, oracle_ped **out)
{
    oracle_ped *p = (oracle_ped *)calloc(1, sizeof(oracle_ped));
    if (!p) return ORACLE_ERR_ALLOC;
    p->n = n;
    p->id = (int64_t *)malloc(n * sizeof(int64_t));
    p->father = (int32_t *)malloc(n * sizeof(int32_t));
    p->mother = (int32_t *)malloc(n * sizeof(int32_t));
    p->nchildren = (int32_t *)calloc(n, sizeof(int32_t));
    p->founder_index = (int32_t *)calloc(n, sizeof(int32_t));
    int rc = idmap_init(&p->map, n); if (rc) { oracle_ped_free(p); return rc; }
    for (int64_t i = 0; i < n; i++) {
        p->id[i] = ind[i];
        int32_t f = -1, mo = -1;
        if (father[i] != 0) { f = idmap_get(&p->map, father[i]); if (f < 0) { oracle_ped_free(p); return ORACLE_ERR_ORDER; } }
        if (mother[i] != 0) { mo = idmap_get(&p->map, mother[i]); if (mo < 0) { oracle_ped_free(p); return ORACLE_ERR_ORDER; } }
        p->father[i] = f; p->mother[i] = mo;
        rc = idmap_put(&p->map, ind[i], (int32_t)i); if (rc) { oracle_ped_free(p); return rc; }
        if (f >= 0) p->nchildren[f]++;
        if (mo >= 0) p->nchildren[mo]++;
    }
    *out = p; return ORACLE_OK;
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return (x > y) - (x < y);
}

/* src/identify.jl:35-39 pro(): IDs of individuals without children, ascending.
 * Returns the count; writes at most cap IDs. */
int64_t oracle_pro(const oracle_ped *p, int64_t *out, int64_t cap) {
    int64_t k = 0;
    for (int64_t i = 0; i < p->n; i++) if (p->nchildren[i] == 0) { if (k < cap) out[k] = p->id[i]; k++; }
    if (k <= cap) qsort(out, (size_t)k, sizeof(int64_t), cmp_i64);
    return k;
}

/* src/identify.jl:15-19 founder(): IDs with neither parent, ascending. */
int64_t oracle_founder(const oracle_ped *p, int64_t *out, int64_t cap) {
    int64_t k = 0;
    for (int64_t i = 0; i < p->n; i++) if (p->father[i] < 0 && p->mother[i] < 0) { if (k < cap) out[k] = p->id[i]; k++; }
    if (k <= cap) qsort(out, (size_t)k, sizeof(int64_t), cmp_i64);
    return k;
}

/* ------------------------------------------------------------------------------------ */
/* src/compute.jl:66-95  phi(::Individual, ::Individual): exact Float64 pairwise Karigl,
 * un-memoised (exponential on inbred deep pedigrees: small inputs only). */
static double pair_rec(const oracle_ped *p, int32_t i, int32_t j) {
    double value = 0.;
    if (i > j) {                                                   /* :68-76 */
        if (p->father[i] >= 0) value += pair_rec(p, p->father[i], j) / 2;
        if (p->mother[i] >= 0) value += pair_rec(p, p->mother[i], j) / 2;
    } else if (j > i) {                                            /* :77-85 */
        if (p->father[j] >= 0) value += pair_rec(p, p->father[j], i) / 2;
        if (p->mother[j] >= 0) value += pair_rec(p, p->mother[j], i) / 2;
    } else {                                                       /* :86-92 */
        value += 0.5;
        if (p->father[i] >= 0 && p->mother[i] >= 0) value += pair_rec(p, p->father[i], p->mother[i]) / 2;
    }
    return value;
}
int oracle_phi_pair(const oracle_ped *p, int64_t id_i, int64_t id_j, double *out) {
    int32_t i = idmap_get(&p->map, id_i), j = idmap_get(&p->map, id_j);
    if (i < 0 || j < 0) return ORACLE_ERR_UNKNOWN_ID;
    *out = pair_rec(p, i, j); return ORACLE_OK;
}
/* src/compute.jl:500-511 f(): Float32 vector of phi(father, mother), 0 if a parent is missing. */
int oracle_f(const oracle_ped *p, int64_t n_ids, const int64_t *ids, float *out) {
    for (int64_t k = 0; k < n_ids; k++) {
        int32_t i = idmap_get(&p->map, ids[k]); if (i < 0) return ORACLE_ERR_UNKNOWN_ID;
        if (p->father[i] < 0 || p->mother[i] < 0) out[k] = (float)0.;
        else out[k] = (float)pair_rec(p, p->father[i], p->mother[i]);
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------ */
/* Ordered "vectors used as sets" exactly as the Julia Base functions behave on Vector{Int}. */
typedef struct { int32_t *v; int64_t n; } ivec;

/* src/compute.jl:193-207 _previous_generation: father then mother of every ID in order,
 * then unique! (first occurrences kept). stamp[] is a scratch array of size ped->n. */
static ivec previous_generation(const oracle_ped *p, ivec next, int32_t *stamp, int32_t tag) {
    ivec r; r.v = (int32_t *)malloc((size_t)(2 * next.n + 1) * sizeof(int32_t)); r.n = 0;
    for (int64_t k = 0; k < next.n; k++) {
        int32_t x = next.v[k];
        int32_t f = p->father[x], mo = p->mother[x];
        if (f >= 0 && stamp[f] != tag) { stamp[f] = tag; r.v[r.n++] = f; }
        if (mo >= 0 && stamp[mo] != tag) { stamp[mo] = tag; r.v[r.n++] = mo; }
    }
    return r;
}
/* Base.union(a, b): unique elements of a in order, then elements of b not yet present. */
static ivec vec_union(ivec a, ivec b, int32_t *stamp, int32_t tag) {
    ivec r; r.v = (int32_t *)malloc((size_t)(a.n + b.n + 1) * sizeof(int32_t)); r.n = 0;
    for (int64_t k = 0; k < a.n; k++) if (stamp[a.v[k]] != tag) { stamp[a.v[k]] = tag; r.v[r.n++] = a.v[k]; }
    for (int64_t k = 0; k < b.n; k++) if (stamp[b.v[k]] != tag) { stamp[b.v[k]] = tag; r.v[r.n++] = b.v[k]; }
    return r;
}
/* Base.intersect(a, b): unique elements of a, in a's order, that are also in b. */
static ivec vec_intersect(ivec a, ivec b, int32_t *stamp, int32_t tag_b, int32_t tag_seen) {
    ivec r; r.v = (int32_t *)malloc((size_t)(a.n + 1) * sizeof(int32_t)); r.n = 0;
    for (int64_t k = 0; k < b.n; k++) stamp[b.v[k]] = tag_b;
    for (int64_t k = 0; k < a.n; k++) if (stamp[a.v[k]] == tag_b) { stamp[a.v[k]] = tag_seen; r.v[r.n++] = a.v[k]; }
    return r;
}

typedef struct {
    int32_t n_levels;      /* L = length(cut_vertices) */
    ivec *cut;             /* cut[0] = top founders ... cut[L-1] = probands (positions) */
    int64_t *both;         /* both[k] = |cut[k] ∩ cut[k+1]|, k < L-1 (:260, :284) */
} oracle_levels;

void oracle_levels_free(oracle_levels *lv) {
    if (!lv) return;
    for (int32_t k = 0; k < lv->n_levels; k++) free(lv->cut[k].v);
    free(lv->cut); free(lv->both); free(lv);
}
int32_t oracle_levels_count(const oracle_levels *lv) { return lv->n_levels; }
int64_t oracle_levels_cut_size(const oracle_levels *lv, int32_t k) { return lv->cut[k].n; }
int64_t oracle_levels_both(const oracle_levels *lv, int32_t k) { return lv->both[k]; }
/* IDs of cut k in the reference's order */
void oracle_levels_cut_ids(const oracle_ped *p, const oracle_levels *lv, int32_t k, int64_t *out) {
    for (int64_t t = 0; t < lv->cut[k].n; t++) out[t] = p->id[lv->cut[k].v[t]];
}

/* src/compute.jl:236-251: generations by parent steps, then cut sets
 * cut[i] = top_down[i] ∩ bottom_up[i]. */
int oracle_levels_create(const oracle_ped *p, int64_t n_pro, const int64_t *pro_ids, oracle_levels **out)
{
    int32_t *stamp = (int32_t *)malloc((size_t)(p->n + 1) * sizeof(int32_t));
    for (int64_t i = 0; i < p->n; i++) stamp[i] = -1;
    int32_t tag = 0;
    /* cut_vertices = [probandIDs]  (raw list, duplicates kept at this point, :236) */
    int64_t cap = 16, L = 0;
    ivec *gen_rev = (ivec *)malloc((size_t)cap * sizeof(ivec));   /* bottom-up; reversed later */
    ivec g0; g0.v = (int32_t *)malloc((size_t)(n_pro + 1) * sizeof(int32_t)); g0.n = n_pro;
    for (int64_t k = 0; k < n_pro; k++) {
        int32_t x = idmap_get(&p->map, pro_ids[k]);
        if (x < 0) { free(g0.v); free(gen_rev); free(stamp); return ORACLE_ERR_UNKNOWN_ID; }   /* KeyError :196 */
        g0.v[k] = x;
    }
    gen_rev[L++] = g0;
    ivec prev = previous_generation(p, g0, stamp, tag++);          /* :237 */
    while (prev.n > 0) {                                           /* :238-241 */
        if (L == cap) { cap *= 2; gen_rev = (ivec *)realloc(gen_rev, (size_t)cap * sizeof(ivec)); }
        gen_rev[L++] = prev;
        prev = previous_generation(p, prev, stamp, tag++);
    }
    free(prev.v);
    /* top-down order: cut_vertices[1] = highest founders */
    ivec *cutv = (ivec *)malloc((size_t)L * sizeof(ivec));
    for (int64_t k = 0; k < L; k++) cutv[k] = gen_rev[L - 1 - k];
    free(gen_rev);
    /* :243-250 cumulative unions */
    ivec *top_down = (ivec *)malloc((size_t)L * sizeof(ivec));
    ivec *bottom_up = (ivec *)malloc((size_t)L * sizeof(ivec));   /* stored already re-reversed */
    top_down[0] = cutv[0];
    for (int64_t i = 0; i + 1 < L; i++) top_down[i + 1] = vec_union(cutv[i + 1], top_down[i], stamp, tag++);
    bottom_up[L - 1] = cutv[L - 1];
    for (int64_t i = L - 1; i > 0; i--) bottom_up[i - 1] = vec_union(cutv[i - 1], bottom_up[i], stamp, tag++);
    /* :251 */
    oracle_levels *lv = (oracle_levels *)calloc(1, sizeof(oracle_levels));
    lv->n_levels = (int32_t)L;
    lv->cut = (ivec *)malloc((size_t)L * sizeof(ivec));
    lv->both = (int64_t *)calloc((size_t)L, sizeof(int64_t));
    for (int64_t i = 0; i < L; i++) { lv->cut[i] = vec_intersect(top_down[i], bottom_up[i], stamp, tag, tag + 1); tag += 2; }
    for (int64_t i = 0; i + 1 < L; i++) {
        ivec b = vec_intersect(lv->cut[i], lv->cut[i + 1], stamp, tag, tag + 1); tag += 2;
        lv->both[i] = b.n; free(b.v);
    }
    for (int64_t i = 0; i < L; i++) {
        if (i > 0) free(top_down[i].v);
        if (i < L - 1) free(bottom_up[i].v);
        free(cutv[i].v);
    }
    free(top_down); free(bottom_up); free(cutv); free(stamp);
    *out = lv; return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------ */
/* src/compute.jl:105-158  phi(::IndexedIndividual, ::IndexedIndividual, Ψ::Matrix{Float32}).
 * Literal: Float64 accumulator starting at 0., `/ 2` per climb, same branch order.
 * Psi is n_prev x n_prev Float32 (bit-symmetric, so row/column-major is immaterial). */
static double level_rec(const oracle_ped *p, int32_t i, int32_t j, const float *Psi, int64_t ld)
{
    double value = 0.;
    const int32_t fi = p->founder_index[i], fj = p->founder_index[j];
    if (fi != 0 && fj != 0) {                                      /* :108-110 */
        value += Psi[(int64_t)(fi - 1) * ld + (fj - 1)];
    } else if (fi != 0) {                                          /* :111-118 */
        if (p->father[j] >= 0) value += level_rec(p, i, p->father[j], Psi, ld) / 2;
        if (p->mother[j] >= 0) value += level_rec(p, i, p->mother[j], Psi, ld) / 2;
    } else if (fj != 0) {                                          /* :119-126 */
        if (p->father[i] >= 0) value += level_rec(p, j, p->father[i], Psi, ld) / 2;
        if (p->mother[i] >= 0) value += level_rec(p, j, p->mother[i], Psi, ld) / 2;
    } else {
        if (i > j) {                                               /* rank_i > rank_j :130-138 */
            if (p->father[i] >= 0) value += level_rec(p, p->father[i], j, Psi, ld) / 2;
            if (p->mother[i] >= 0) value += level_rec(p, p->mother[i], j, Psi, ld) / 2;
        } else if (j > i) {                                        /* :139-147 */
            if (p->father[j] >= 0) value += level_rec(p, p->father[j], i, Psi, ld) / 2;
            if (p->mother[j] >= 0) value += level_rec(p, p->mother[j], i, Psi, ld) / 2;
        } else {                                                   /* :148-154 */
            value += 0.5;
            if (p->father[i] >= 0 && p->mother[i] >= 0)
                value += level_rec(p, p->father[i], p->mother[i], Psi, ld) / 2;
        }
    }
    return value;
}

/* src/compute.jl:269-303: the level sweep.  out: n_L x n_L Float32 (row-major; symmetric).
 * stop_after_levels > 0 stops after that many level steps (bench sampling only) and
 * returns without touching `out`; entries_done (optional) receives the number of
 * (i <= j) kernel evaluations performed. */
int oracle_phi_compute(oracle_ped *p, const oracle_levels *lv, float *out,
                       int32_t stop_after_levels, int64_t *entries_done)
{
    const int32_t L = lv->n_levels;
    memset(p->founder_index, 0, (size_t)p->n * sizeof(int32_t));   /* fresh _index_pedigree :269 */
    int64_t n1 = lv->cut[0].n;
    float *Psi = (float *)calloc((size_t)(n1 * n1 + 1), sizeof(float));  /* :271 zeros(Float32) */
    if (!Psi) return ORACLE_ERR_ALLOC;
    for (int64_t i = 0; i < n1; i++) Psi[i * n1 + i] = 0.5f;       /* :272-274 */
    int64_t ld = n1, done = 0;
    for (int32_t k = 0; k + 1 < L; k++) {                          /* :276 */
        if (stop_after_levels > 0 && k >= stop_after_levels) break;
        const ivec prev = lv->cut[k], next = lv->cut[k + 1];
        for (int64_t t = 0; t < prev.n; t++) p->founder_index[prev.v[t]] = (int32_t)(t + 1);  /* :287-289 */
        const int64_t n = next.n;
        float *phi = (float *)malloc((size_t)(n * n + 1) * sizeof(float));                     /* :291 */
        if (!phi) { free(Psi); return ORACLE_ERR_ALLOC; }
        /* threads only where a level is big enough to pay for the fork/join (deep pedigrees have
         * hundreds of tiny levels; the reference's Threads.@threads has the same per-level cost) */
        #pragma omp parallel for schedule(dynamic, 8) if (n >= 512)
        for (int64_t i = 0; i < n; i++) {                          /* :293-299 */
            for (int64_t j = i; j < n; j++) {
                float v = (float)level_rec(p, next.v[i], next.v[j], Psi, ld);  /* Float64 -> Float32 RN */
                phi[i * n + j] = v; phi[j * n + i] = v;            /* :296 */
            }
        }
        done += n * (n + 1) / 2;
        free(Psi); Psi = phi; ld = n;                              /* :301 */
    }
    if (entries_done) *entries_done = done;
    if (stop_after_levels <= 0 || stop_after_levels >= L - 1) {
        int64_t nL = lv->cut[L - 1].n;
        if (out) memcpy(out, Psi, (size_t)(nL * nL) * sizeof(float));
    }
    free(Psi);
    return ORACLE_OK;
}

/* Row samples of the LAST level step (test infrastructure for sizes at which the whole matrix costs minutes):
 * every upper level step in full, exactly as oracle_phi_compute, then only the given rows of the proband matrix
 * (positions in probandIDs order).  Entry (r, j) is evaluated as the reference evaluates it -- the pair in index
 * order, phi(probands[min(r, j)], probands[max(r, j)], Psi) (src/compute.jl:295-296: only i <= j, mirrored store).
 * out: n_rows x n_L Float32, row-major. */
int oracle_phi_compute_rows(oracle_ped *p, const oracle_levels *lv, int64_t n_rows, const int64_t *rows, float *out,
                            int64_t *entries_done)
{
    const int32_t L = lv->n_levels;
    memset(p->founder_index, 0, (size_t)p->n * sizeof(int32_t));   /* fresh _index_pedigree :269 */
    int64_t n1 = lv->cut[0].n;
    float *Psi = (float *)calloc((size_t)(n1 * n1 + 1), sizeof(float));  /* :271 */
    if (!Psi) return ORACLE_ERR_ALLOC;
    for (int64_t i = 0; i < n1; i++) Psi[i * n1 + i] = 0.5f;       /* :272-274 */
    int64_t ld = n1, done = 0;
    for (int32_t k = 0; k + 1 < L; k++) {                          /* :276 */
        const ivec prev = lv->cut[k], next = lv->cut[k + 1];
        for (int64_t t = 0; t < prev.n; t++) p->founder_index[prev.v[t]] = (int32_t)(t + 1);  /* :287-289 */
        const int64_t n = next.n;
        if (k + 2 == L) {                                          /* the last step: the sampled rows only */
            #pragma omp parallel for schedule(dynamic, 1)
            for (int64_t q = 0; q < n_rows; q++) {
                const int64_t r = rows[q];
                for (int64_t j = 0; j < n; j++) {
                    const int64_t a = r < j ? r : j, b = r < j ? j : r;
                    out[q * n + j] = (float)level_rec(p, next.v[a], next.v[b], Psi, ld);
                }
            }
            done += n_rows * n;
            break;
        }
        float *phi = (float *)malloc((size_t)(n * n + 1) * sizeof(float));                     /* :291 */
        if (!phi) { free(Psi); return ORACLE_ERR_ALLOC; }
        #pragma omp parallel for schedule(dynamic, 8) if (n >= 512)
        for (int64_t i = 0; i < n; i++) {                          /* :293-299 */
            for (int64_t j = i; j < n; j++) {
                float v = (float)level_rec(p, next.v[i], next.v[j], Psi, ld);
                phi[i * n + j] = v; phi[j * n + i] = v;            /* :296 */
            }
        }
        done += n * (n + 1) / 2;
        free(Psi); Psi = phi; ld = n;                              /* :301 */
    }
    if (L == 1) for (int64_t q = 0; q < n_rows; q++) memcpy(out + q * n1, Psi + rows[q] * n1, (size_t)n1 * sizeof(float));
    if (entries_done) *entries_done = done;
    free(Psi);
    return ORACLE_OK;
}

/* Timing sample for bench.py's cpu_baseline (never a result): the pair kernel over n_rows rows x all columns of level
 * step k (cut k -> cut k + 1) with the real index structure of that step, on a matrix Psi of the real size that holds
 * arbitrary values -- the control flow and the memory accesses of src/compute.jl:105-158 depend on founder_index, father,
 * mother and rank only, not on the values.  Returns the evaluations done and, in *seconds, the wall time of the evaluations alone (the matrix is filled before). */
int64_t oracle_time_level_rows(oracle_ped *p, const oracle_levels *lv, int32_t k, int64_t n_rows, const int64_t *rows, double *seconds)
{
    if (k < 0 || k + 1 >= lv->n_levels) return 0;
    memset(p->founder_index, 0, (size_t)p->n * sizeof(int32_t));
    const ivec prev = lv->cut[k], next = lv->cut[k + 1];
    for (int64_t t = 0; t < prev.n; t++) p->founder_index[prev.v[t]] = (int32_t)(t + 1);
    const int64_t ld = prev.n, n = next.n;
    float *Psi = (float *)malloc((size_t)(ld * ld + 1) * sizeof(float));
    if (!Psi) return -1;
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ld * ld; i++) Psi[i] = 0.25f;           /* (first touch spreads the pages over the threads) */
    double sink = 0.;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    #pragma omp parallel for schedule(dynamic, 1) reduction(+ : sink)
    for (int64_t q = 0; q < n_rows; q++) {
        const int64_t r = rows[q];
        for (int64_t j = 0; j < n; j++) {
            const int64_t a = r < j ? r : j, b = r < j ? j : r;
            sink += (float)level_rec(p, next.v[a], next.v[b], Psi, ld);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (seconds) *seconds = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
    free(Psi);
    return sink >= 0. ? n_rows * n : -2;
}

/* src/compute.jl:454-459 phiMean(::Matrix{Float32}) with plain left-to-right Float32 sums.
 * NOTE: Julia's sum() is pairwise-blocked; this matches it exactly only when the sums are
 * exact in Float32 (true for geneaJi: 0.171875, test/runtests.jl:53). */
float oracle_phi_mean(const float *phi, int64_t n) {
    float total = 0.f, diagonal = 0.f;
    for (int64_t i = 0; i < n * n; i++) total += phi[i];
    for (int64_t i = 0; i < n; i++) diagonal += phi[i * n + i];
    total -= diagonal;
    return total / (float)(n * n - n);
}

/* Threads of the level loops from here on (bench.py / the tests size it to the CPUs the process may actually use: a container
 * with a CPU quota below the machine's core count is throttled, not sped up, by one thread per core). */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
