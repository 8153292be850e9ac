// Would triangular storage of the level matrices pay?  (round-4 question: the reference evaluates only i <= j and mirrors the store,
// src/compute.jl:295-296; every level matrix here is bit-symmetric.)  The row kernels stage WHOLE source rows: Psi[A][0 .. n).  If
// only the entries on and right of the diagonal were stored, a staged row would be
//     Psi[A][A .. n)      contiguous in row A                      (16-byte loads, as today)
//     Psi[0 .. A)[A]      column A of the rows above: one 4-byte element per row, stride = the row pitch
// This measures what staging costs in both forms, on a 24,320 x 24,320 matrix (an upper level of cfg4), one workgroup of 1024
// threads per staged row, 256 workgroups walking random rows, the row summed so that nothing is optimised away:
//     full rows        every row staged contiguously (the product's layout)
//     triangular       the column part gathered with stride-ld loads, the row part contiguous
// Useful bytes per staged row are n * 4 in both; the triangular form touches one 32-byte sector (a 128-byte line at the fabric)
// per element of the column part.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool TRI>
__global__ void __launch_bounds__(1024) stage(const float *__restrict__ m, long long ld, int n, int stages, float *__restrict__ out)
{
    float acc = 0.f;
    unsigned r = blockIdx.x * 97u;
    for (int s = 0; s < stages; ++s) {
        r = (r + 256u * 37u + 11u) % (unsigned)n;
        const float *row = m + (long long)r * ld;
        if (!TRI) {
            for (int k = threadIdx.x * 4; k < n; k += 4096) {
                const f4 v = *reinterpret_cast<const f4 *>(row + k);
                acc += v.x + v.y + v.z + v.w;
            }
        } else {
            const int a = (int)r & ~3;                                   // (quad-aligned start of the row part)
            for (int k = a + threadIdx.x * 4; k < n; k += 4096) {
                const f4 v = *reinterpret_cast<const f4 *>(row + k);
                acc += v.x + v.y + v.z + v.w;
            }
            // column part: elements [k][r], k < a -- eight loads in flight per thread
            for (int k0 = threadIdx.x; k0 < a; k0 += 8 * 1024) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int k = min(k0 + u * 1024, a - 1); v[u] = m[(long long)k * ld + r]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (k0 + u * 1024 < a) acc += v[u];
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const int n = 24320;
    const long long ld = 24384;
    float *m, *out;
    (void)hipMalloc(reinterpret_cast<void **>(&m), (size_t)n * ld * 4);
    (void)hipMalloc(reinterpret_cast<void **>(&out), 64);
    (void)hipMemset(m, 0, (size_t)n * ld * 4);
    const int stages = 64, grid = 256;
    for (int tri = 0; tri < 2; ++tri) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a);
            if (tri) hipLaunchKernelGGL(stage<true>, dim3(grid), dim3(1024), 0, 0, m, ld, n, stages, out);
            else hipLaunchKernelGGL(stage<false>, dim3(grid), dim3(1024), 0, 0, m, ld, n, stages, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double useful = (double)grid * stages * n * 4.0;
        printf("%-44s %8.3f ms for %d staged rows = %7.1f ns per row per workgroup, %6.2f TB/s of useful bytes\n",
               tri ? "triangular storage (row part + column part)" : "full rows (the product's layout)", best, grid * stages,
               best * 1e6 / stages, useful / (best * 1e-3) / 1e12);
    }
    return 0;
}
