// What besides gathers + combine makes a real B stage of level_split_kernel take ~6 us where the
// bare body takes 3 us?  Adds the stage's other traffic step by step (same shape: 256 workgroups x
// 1024 threads, 24 columns per thread, 24576-float rows):
//   MODE 0: gathers + combine only                      MODE 1: + 6 row stores (16 B per lane) per stage
//   MODE 2: + 6 prefetch loads of the next row (global)  MODE 3: + staging the prefetched row into LDS
//   between two barriers (= a whole stage, without the queue / descriptor logic)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off stage_parts.hip -o stage_parts
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ROW = 24576, CPT = 24, STAGES = 256, NT = 1024;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float combine_e(float a, float b, float c, float d, bool i_hi, int e)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;
    const double s = (static_cast<double>(a) + static_cast<double>(x)) + (static_cast<double>(y) + static_cast<double>(d));
    return static_cast<float>(__builtin_ldexp(s, e));
}

template <int MODE>
__global__ void __launch_bounds__(1024) stage(const float *__restrict__ psi, float *__restrict__ out, unsigned seed, int ri0, int n_rows)
{
    extern __shared__ float sR[];
    for (int i = threadIdx.x; i < ROW; i += NT) sR[i] = 1.0f / (1 + (i & 1023));
    __syncthreads();
    unsigned pk[CPT];
    float pa[CPT], pb[CPT];
    unsigned x = seed + (blockIdx.x * NT + threadIdx.x) * 2654435761u;
    for (int k = 0; k < CPT; ++k) {
        x = x * 1664525u + 1013904223u; const unsigned A = (x >> 8) % ROW;
        x = x * 1664525u + 1013904223u; const unsigned B = (x >> 8) % ROW;
        pk[k] = A | B << 16; pa[k] = sR[A]; pb[k] = sR[B];
    }
    unsigned tl = threadIdx.x;
    unsigned row = (blockIdx.x * 7919u) % n_rows;
    f4 pre[6];
    float acc = 0.f;
    if (MODE >= 2)
        for (int k = 0; k < 6; ++k) pre[k] = *reinterpret_cast<const f4 *>(psi + (size_t)row * ROW + (tl + k * NT) * 4);
    for (int s = 0; s < STAGES; ++s) {
        asm volatile("" : "+v"(tl));
        if (MODE >= 3) {
            __syncthreads();
            for (int k = 0; k < 6; ++k) *reinterpret_cast<f4 *>(sR + (tl + k * NT) * 4) = pre[k];
            __syncthreads();
        }
        row = (row * 1103515245u + 12345u) % n_rows;
        if (MODE >= 2)
            for (int k = 0; k < 6; ++k) pre[k] = *reinterpret_cast<const f4 *>(psi + (size_t)row * ROW + (tl + k * NT) * 4);
        float *orow = out + (size_t)((row + 17) % n_rows) * ROW;
        const int ri = ri0 + s * 37;
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            const unsigned jq = q * 4096 + tl * 4;
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                const float c = sR[pk[k] & 0xffff], d = sR[pk[k] >> 16];
                v[e] = combine_e(pa[k], pb[k], c, d, jq + e < (unsigned)ri, -2);
            }
            if (MODE >= 1) *reinterpret_cast<f4 *>(orow + jq) = v;
            else acc += (v[0] + v[1]) + (v[2] + v[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE >= 2) for (int k = 0; k < 6; ++k) acc += pre[k][0];
    if (acc == 12345.f) out[0] = acc;
}

int main()
{
    const int n_rows = 24576;                          // a 2.4 GB level matrix
    float *psi, *out;
    (void)hipMalloc(&psi, (size_t)n_rows * ROW * 4); (void)hipMalloc(&out, (size_t)n_rows * ROW * 4);
    (void)hipMemset(psi, 0, (size_t)n_rows * ROW * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    void (*ks[])(const float *, float *, unsigned, int, int) = {stage<0>, stage<1>, stage<2>, stage<3>};
    const char *names[] = {"gathers + combine", "+ row stores", "+ prefetch loads", "+ LDS staging and 2 barriers"};
    for (int m = 0; m < 4; ++m) {
        (void)hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, ROW * 4);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(ks[m], dim3(256), dim3(1024), ROW * 4, 0, psi, out, 7u, 12000, n_rows);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        }
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-32s %7.3f ms = %6.3f us per stage  (%.2f TB/s of row traffic)\n", names[m], ms, ms * 1e3 / STAGES,
               (m >= 1 ? 1 : 0) * 256.0 * STAGES * ROW * 4 * (m >= 2 ? 2 : 1) / (ms * 1e-3) / 1e12);
    }
    return 0;
}
