// Where does a B stage of level_split_kernel spend its on-chip time?  Same shape as the kernel
// (256 workgroups x 1024 threads, a 24576-float row in LDS, CPT columns per thread with random
// packed indices, quads separated by sched_barrier) but no global traffic in the loop.
//   MODE 0: LDS gathers + combine (the real stage body)     MODE 1: combine only (c, d from ALU)
//   MODE 2: gathers only (c + d in f32)                      MODE 3: gathers + combine, exponent
//   adjusted with integer ops instead of v_ldexp_f64         MODE 4: like 0 with wave-uniform i_hi
//   MODE 5: like 4 with one-instruction LDS addresses (v_mad_u32_u16 on the packed halves)
//   MODE 6: like 0 with the one-instruction LDS addresses
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off combine_rates.hip -o combine_rates
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ROW = 24576, CPT = 24, STAGES = 512;

__device__ __forceinline__ float combine_e(float a, float b, float c, float d, bool i_hi, int e)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;
    const double s = (static_cast<double>(a) + static_cast<double>(x)) + (static_cast<double>(y) + static_cast<double>(d));
    return static_cast<float>(__builtin_ldexp(s, e));
}
__device__ __forceinline__ float combine_int(float a, float b, float c, float d, bool i_hi, int e)
{
    const float x = i_hi ? b : c, y = i_hi ? c : b;
    const double s = (static_cast<double>(a) + static_cast<double>(x)) + (static_cast<double>(y) + static_cast<double>(d));
    // s >= 0 and either 0 or >= 2^-149: scale by 2^e (e <= 0) on the exponent field, saturating at 0
    unsigned long long u = __double_as_longlong(s);
    unsigned hi = static_cast<unsigned>(u >> 32);
    const unsigned dec = static_cast<unsigned>(-e) << 20;
    hi = hi > dec ? hi - dec : 0u;
    u = (static_cast<unsigned long long>(hi) << 32) | (u & 0xffffffffull);
    return static_cast<float>(__longlong_as_double(u));
}

template <int MODE>
__global__ void __launch_bounds__(1024) stage(float *out, unsigned seed, int ri0)
{
    extern __shared__ float sR[];
    for (int i = threadIdx.x; i < ROW; i += 1024) sR[i] = 1.0f / (1 + (i & 1023));
    __syncthreads();
    unsigned pk[CPT];
    float pa[CPT], pb[CPT], acc[CPT / 4];
    unsigned x = seed + (blockIdx.x * 1024 + threadIdx.x) * 2654435761u;
    for (int k = 0; k < CPT; ++k) {
        x = x * 1664525u + 1013904223u;
        const unsigned A = (x >> 8) % ROW;
        x = x * 1664525u + 1013904223u;
        const unsigned B = (x >> 8) % ROW;
        pk[k] = A | B << 16;
        pa[k] = sR[A]; pb[k] = sR[B];
    }
    for (int q = 0; q < CPT / 4; ++q) acc[q] = 0.f;
    unsigned tl = threadIdx.x;
    for (int s = 0; s < STAGES; ++s) {
        asm volatile("" : "+v"(tl));
        const int ri = ri0 + s * 37;
        const bool wave_hi = (s & 1);
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            const unsigned jq = q * 4096 + tl * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                float c, d;
                if (MODE == 1) { c = __uint_as_float(__float_as_uint(pa[k]) ^ (unsigned)s); d = __uint_as_float(__float_as_uint(pb[k]) + (unsigned)s); }
                else if (MODE == 5 || MODE == 6) {
                    unsigned ac, ad;                      // byte address = half * 4 + 0, one VALU op each
                    asm("v_mad_u32_u16 %0, %1, 4, 0" : "=v"(ac) : "v"(pk[k]));
                    asm("v_mad_u32_u16 %0, %1, 4, 0 op_sel:[1,0,0,0]" : "=v"(ad) : "v"(pk[k]));
                    c = *reinterpret_cast<float *>(reinterpret_cast<char *>(sR) + ac);
                    d = *reinterpret_cast<float *>(reinterpret_cast<char *>(sR) + ad);
                }
                else { c = sR[pk[k] & 0xffff]; d = sR[pk[k] >> 16]; }
                const bool i_hi = (MODE == 4 || MODE == 5) ? wave_hi : (jq + e < (unsigned)ri);
                if (MODE == 2) v[e] = c + d;
                else if (MODE == 3) v[e] = combine_int(pa[k], pb[k], c, d, i_hi, -2);
                else v[e] = combine_e(pa[k], pb[k], c, d, i_hi, -2);
            }
            acc[q] += (v[0] + v[1]) + (v[2] + v[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0;
    for (int q = 0; q < CPT / 4; ++q) t += acc[q];
    out[blockIdx.x * 1024 + threadIdx.x] = t;
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    void (*ks[])(float *, unsigned, int) = {stage<0>, stage<1>, stage<2>, stage<3>, stage<4>, stage<5>, stage<6>};
    const char *names[] = {"gathers + combine", "combine only", "gathers only", "gathers + combine (int exponent)", "gathers + combine (wave-uniform i_hi)", "uniform i_hi + mad_u16 addresses", "per-lane i_hi + mad_u16 addresses"};
    for (int m = 0; m < 7; ++m) {
        (void)hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, ROW * 4);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(ks[m], dim3(256), dim3(1024), ROW * 4, 0, out, 7u, 12000);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        }
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-40s %7.3f ms  = %6.3f us per stage of %d columns/thread = %6.1f clk@2.4GHz per column per wave (4 waves/SIMD)\n",
               names[m], ms, ms * 1e3 / STAGES, CPT, ms * 1e-3 / STAGES * 2.4e9 / (CPT * 4));
    }
    return 0;
}
