// Instruction-rate microbenchmark for the VALU mix of the level kernels (gfx950).
// Each kernel runs ITER iterations of 8 independent chains of one instruction per thread;
// 256 workgroups x 1024 threads (4 waves per SIMD, like level_split_kernel).  Reported:
// wave-instructions per clock per SIMD (1/4 = "full rate" for wave64 on a 16-lane SIMD).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, float seed, int e)
{
    float f[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + threadIdx.x * 1e-3f + i; d[i] = f[i]; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "s"(e));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(seed) : "vcc");
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(f[i]) : "v"(seed));
            if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

// random LDS gather: cost of one ds_read_b32 with 64 random addresses in a ROWS-float window
template <int ROWS>
__global__ void __launch_bounds__(1024) gather(float *out, unsigned seed)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < ROWS; i += 1024) lds[i] = i;
    __syncthreads();
    unsigned x = seed + threadIdx.x * 2654435761u;
    float acc = 0;
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            acc += lds[(x >> 8) % ROWS];
        }
    }
    if (acc == 1.5f) out[0] = acc;
}

int main()
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char *names[] = {"v_add_f32", "v_add_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_ldexp_f64", "v_cndmask_b32",
                           "v_fma_f64", "v_lshl_add_u32", "v_mul_f64"};
    void (*ks[])(float *, float, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
    printf("clock %d kHz (nominal)\n", clk_khz);
    for (int op = 0; op < 9; ++op) {
        hipLaunchKernelGGL(ks[op], dim3(256), dim3(1024), 0, 0, out, 1.0f, -2);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(ks[op], dim3(256), dim3(1024), 0, 0, out, 1.0f, -2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per SIMD: 4 waves x ITER x 8 wave-instructions
        const double winst = 4.0 * ITER * 8;
        const double clks = ms * 1e-3 * clk_khz * 1e3;
        printf("%-16s %8.3f ms  %6.2f clk per wave-instruction per SIMD\n", names[op], ms, clks / winst);
    }
    {
        const int rows = 24576;
        hipFuncSetAttribute((const void *)gather<24576>, hipFuncAttributeMaxDynamicSharedMemorySize, rows * 4);
        hipLaunchKernelGGL(gather<24576>, dim3(256), dim3(1024), rows * 4, 0, out, 7u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(gather<24576>, dim3(256), dim3(1024), rows * 4, 0, out, 7u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double winst = 16.0 * (ITER / 4) * 8;      // per CU: 16 waves
        const double clks = ms * 1e-3 * clk_khz * 1e3;
        printf("random ds_read_b32 over %d floats: %8.3f ms  %6.2f clk per wave-gather per CU (incl. ~4 VALU of index math)\n",
               rows, ms, clks / winst);
    }
    return 0;
}
