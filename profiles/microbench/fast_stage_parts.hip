// Where does a B stage of level_split_fast_kernel<512, 52, 16> (the 1e5-wide last level of cfg4) spend its time on the CU?
// Same shape (40 columns per thread instead of 52: this plain body needs a few more registers than the product kernel): 256 workgroups x 512 threads, a 30 976-float source row (124 KB) in LDS, pab = one double per
// column, v = RN32(fma(c + d, 1/4, pab)).  Parts are added one by one:
//   MODE 0: the two LDS gathers per column only (summed in f32)
//   MODE 1: VALU only (conversions + add + fma + conversion on register values, no LDS)
//   MODE 2: gathers + VALU (the body)
//   MODE 3: + the 10 row stores per thread
//   MODE 4: + staging a row from registers into LDS between two barriers (no global loads)
//   MODE 5: + loading the next row from global memory into the staging registers (a whole stage without queue logic)
//   MODE 6: MODE 5 with every workgroup loading row 0 (the loads always hit)
//   MODE 7: MODE 5 with an s_waitcnt vmcnt(0) in front of the LDS writes: the acknowledgement of the stage's row stores is waited
//           for (what the product kernel does: its stores are conditional, so hipcc cannot count them out of the wait for the
//           older prefetch loads; in MODE 5 the stores are unconditional and the wait is a counted vmcnt(N))
//   MODE 8: MODE 5 with the row staged by ds_write_addtid_b32 (4 bytes per lane at M0 + offset + 4 * lane: 128 B/clk against the
//           ~79 B/clk of ds_write_b128; the LDS image is then a permutation of the row, which the product would fold into its index words)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fast_stage_parts.hip -o fast_stage_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ROWF = 30976, LD = 31040, CPT = 40, STG = 16, NT = 512, STAGES = 512, OUTW = NT * CPT;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(NT) stage(const float *__restrict__ psi, float *__restrict__ out, unsigned seed, int n_rows, int n_out_rows, int share)
{
    extern __shared__ float sR[];
    for (int i = threadIdx.x; i < LD; i += NT) sR[i] = 1.0f / (1 + (i & 1023));
    __syncthreads();
    unsigned pk[CPT];
    double pab[CPT];
    unsigned x = seed + (blockIdx.x * NT + threadIdx.x) * 2654435761u;
    for (int k = 0; k < CPT; ++k) {
        x = x * 1664525u + 1013904223u; const unsigned A = (x >> 8) % ROWF;
        x = x * 1664525u + 1013904223u; const unsigned B = (x >> 8) % ROWF;
        pk[k] = A | B << 16; pab[k] = (static_cast<double>(sR[A]) + static_cast<double>(sR[B])) * 0.25;
    }
    unsigned tl = threadIdx.x;
    // workgroups b, b + 8, b + 16, ... sit on one XCD; `share` consecutive ones of them walk the same rows (as the column chunks
    // of one work item do in the product kernel): their loads can hit L2
    unsigned row = (((blockIdx.x >> 3) / share * 8 + (blockIdx.x & 7)) * 7919u) % n_rows;
    f4 pre[STG];
    float acc = 0.f;
    constexpr bool kLoads = MODE >= 5, kStage = MODE >= 4, kAddTid = MODE == 8;
    if (kStage)
        for (int k = 0; k < STG; ++k) {
            const unsigned o = (tl + k * NT) * 4;
            pre[k] = o < (unsigned)LD ? *reinterpret_cast<const f4 *>(psi + (size_t)row * LD + o) : f4{0, 0, 0, 0};
        }
    for (int s = 0; s < STAGES; ++s) {
        asm volatile("" : "+v"(tl));
        row = MODE == 6 ? 0u : (row * 1103515245u + 12345u) % n_rows;
        if (kStage) {
            __syncthreads();
            if (MODE == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < STG; ++k) {
                const unsigned o = (tl + k * NT) * 4;
                if (k < STG - 1 || o < (unsigned)LD) {          // (only the last piece can lie past the row's end)
                    if (kAddTid) {
                        // byte address of (wave w, piece k, component c) = w * 1024 + k * 8192 + c * 256 (+ 4 * lane by the hardware):
                        // M0 and the offset are 16 bits each, so the upper half of the row goes through a second M0 value
                        const unsigned wv = __builtin_amdgcn_readfirstlane(tl >> 6);
                        const unsigned m0v = wv * 1024u + (k < 8 ? 0u : 58116u);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" :: "v"(pre[k][c]), "s"(m0v),
                                         "n"(k * 8192 + c * 256 - (k < 8 ? 0 : 58116)) : "memory");
                    } else
                    *reinterpret_cast<f4 *>(sR + o) = pre[k];
                    if (kLoads) pre[k] = *reinterpret_cast<const f4 *>(psi + (size_t)row * LD + o);
                    else asm volatile("" : "+v"(pre[k]));
                }
            }
            __syncthreads();
        }
        float *orow = out + (size_t)((blockIdx.x * 61u + s) % n_out_rows) * OUTW;
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            const unsigned jq = q * (NT * 4) + tl * 4;
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                float c, d;
                if (MODE == 1) { c = __uint_as_float(pk[k] + s); d = __uint_as_float(pk[k] ^ tl); }
                else { c = sR[pk[k] & 0xffff]; d = sR[pk[k] >> 16]; }
                if (MODE == 0) v[e] = c + d;
                else v[e] = static_cast<float>(__builtin_fma(static_cast<double>(c) + static_cast<double>(d), 0.25, pab[k]));
            }
            if (MODE >= 3) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(orow + jq));
            else acc += (v[0] + v[1]) + (v[2] + v[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (kStage) for (int k = 0; k < STG; ++k) acc += pre[k][0];
    if (acc == 12345.f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 256;       // workgroups (default: one per CU); 128 = half the chip: less bandwidth contention
    printf("%d workgroups\n", grid);
    const int n_rows = 30976, n_out_rows = 16384;      // a 3.8 GB source level, 1.7 GB of output rows
    float *psi, *out;
    (void)hipMalloc(&psi, (size_t)n_rows * LD * 4); (void)hipMalloc(&out, (size_t)n_out_rows * OUTW * 4);
    (void)hipMemset(psi, 0, (size_t)n_rows * LD * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    void (*ks[])(const float *, float *, unsigned, int, int, int) = {stage<0>, stage<1>, stage<2>, stage<3>, stage<4>, stage<5>, stage<6>, stage<7>, stage<8>};
    const char *names[] = {"gathers only", "VALU only", "gathers + VALU", "+ row stores", "+ LDS staging, 2 barriers", "+ loads of the next row",
                           "same, every load hits (row 0)", "MODE 5 + vmcnt(0) before the LDS writes", "MODE 5, row staged by ds_write_addtid_b32"};
    for (int m = 0; m < 9; ++m) {
        (void)hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, LD * 4);
        for (int share = 1; share <= (m == 5 || m == 7 || m == 8 ? 4 : 1); share *= 2) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(ks[m], dim3(grid), dim3(NT), LD * 4, 0, psi, out, 7u, n_rows, n_out_rows, share);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%-42s %s %8.3f ms = %6.3f us per stage\n", names[m], share == 1 ? "          " : (share == 2 ? "(2 share) " : "(4 share) "), best, best * 1e3 / STAGES);
        }
    }
    return 0;
}
