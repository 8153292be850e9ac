/* Plain-C consumer of include/genphi.h: the same calls the Julia shim makes with ccall.
 *
 *   gcc -std=c11 -O2 -I include examples/phi_c_abi.c -o phi_c_abi \
 *       -L genlib.jl_amd/lib -lgenphi -Wl,-rpath,$PWD/genlib.jl_amd/lib
 *   ./phi_c_abi genlib.jl_amd/data/geneaJi.csv  # prints the 3 x 3 matrix of test/runtests.jl:50-52
 *   ./phi_c_abi genlib.jl_amd/data/geneaJi.csv --plan  # levelisation only (works without a GPU)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "genphi.h"

static int die(const char *what, int rc)
{
    fprintf(stderr, "%s failed (code %d): %s\n", what, rc, genphi_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s pedigree.tsv [--plan]\n", argv[0]); return 2; }
    const int plan_only = argc > 2 && strcmp(argv[2], "--plan") == 0;

    int64_t n = 0, *ind = NULL, *father = NULL, *mother = NULL;
    int rc = genphi_genealogy_read(argv[1], 1, &n, &ind, &father, &mother, NULL);       /* gen.genealogy(file) */
    if (rc) return die("genphi_genealogy_read", rc);

    /* gen.pro: individuals without children, ascending (the arrays are in rank order, IDs are arbitrary) */
    int64_t *pro = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1)), n_pro = 0;
    for (int64_t i = 0; i < n; ++i) {
        int has_child = 0;
        for (int64_t k = 0; k < n && !has_child; ++k) has_child = father[k] == ind[i] || mother[k] == ind[i];
        if (!has_child) pro[n_pro++] = ind[i];
    }
    for (int64_t a = 1; a < n_pro; ++a)                                                   /* insertion sort: tiny lists */
        for (int64_t b = a; b > 0 && pro[b - 1] > pro[b]; --b) { int64_t t = pro[b]; pro[b] = pro[b - 1]; pro[b - 1] = t; }

    genphi_plan *plan = NULL;
    rc = genphi_plan_create(n, ind, father, mother, n_pro, pro, &plan);
    if (rc) return die("genphi_plan_create", rc);
    int32_t n_levels = 0;
    const int64_t *cut = NULL, *both = NULL;
    rc = genphi_plan_levels(plan, &n_levels, &cut, &both);
    if (rc) return die("genphi_plan_levels", rc);
    for (int32_t k = 0; k + 1 < n_levels; ++k)                                            /* src/compute.jl:257-260 */
        printf("Step %d of %d: %lld founders, %lld probands, %lld both.\n", k + 1, n_levels - 1,
               (long long)cut[k], (long long)cut[k + 1], (long long)both[k]);

    if (!plan_only) {
        const int64_t N = genphi_plan_n_probands(plan);
        float *phi = malloc(sizeof(float) * (size_t)(N * N > 0 ? N * N : 1));
        rc = genphi_compute_f32(plan, phi, NULL, NULL);                                   /* the GPU level sweep */
        if (rc) return die("genphi_compute_f32", rc);
        for (int64_t i = 0; i < N && N <= 16; ++i) {
            for (int64_t j = 0; j < N; ++j) printf("%.9g ", phi[i * N + j]);
            printf("\n");
        }
        double all = 0, diag = 0;
        rc = genphi_result_sums(plan, &all, &diag, NULL);                                 /* phiMean without a D2H */
        if (rc) return die("genphi_result_sums", rc);
        if (N > 1) printf("phiMean %.9g\n", (all - diag) / (double)(N * N - N));
        free(phi);
    }
    genphi_plan_destroy(plan);
    genphi_free(ind); genphi_free(father); genphi_free(mother); free(pro);
    return 0;
}
