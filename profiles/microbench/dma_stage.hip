// Would the 1e5-wide last level of cfg4 run in THREE column chunks if the source rows went into LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, so 68 columns per thread instead of 52)?  The catch: the next row can only
// land where the current one is no longer gathered from, i.e. after the gathers, and its arrival is exposed in every stage.
// This measures a stage of that shape (256 workgroups x 512 threads, 68 columns per thread = 34 816 columns per chunk,
// 124 KB rows) next to the register-staged stage of fast_stage_parts (MODE 5 there):
//   MODE 0: gathers + stores; barrier; LDS-DMA of the whole next row; vmcnt(0); barrier              (nothing overlaps)
//   MODE 1: the first 32 KB of the next row are DMA'd into the 32 KB of LDS the (padded) row image leaves free WHILE the gathers run;
//           after the barrier the other 89 KB go straight to their place and the 32 KB are moved there through registers
//   SHARE 1/2/4: that many workgroups of an XCD walk the same row sequence (the column chunks of one work item do) -> L2 hits
// Verified: the last stage's output of workgroup 0 is recomputed on the host from the row it must have gathered from.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dma_stage.hip -o dma_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWF = 30976, LD = 31040, CPT = 68, NT = 512, STAGES = 384, OUTW = NT * CPT;
constexpr int ROW_BYTES = LD * 4, NP = ((ROW_BYTES + 1023) / 1024 + 7) / 8 * 8;   // 1-KiB pieces (one wave-instruction each), the same number for every wave:
                                                                          // the last ones read into the following row (harmless) so that vmcnt can be counted
constexpr int P0 = 32;                                                  // pieces that go to the slack first (MODE 1)
constexpr int SLACK_OFF = NP * 1024;                                    // bytes; row image at [0, 128 KB)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__host__ __device__ inline unsigned next_row(unsigned row, unsigned n_rows) { return (row * 1103515245u + 12345u) % n_rows; }
__host__ __device__ inline unsigned col_pk(unsigned seed, unsigned gtid, int k)
{
    unsigned x = seed + gtid * 2654435761u + k * 40503u;
    x = x * 1664525u + 1013904223u; const unsigned A = (x >> 8) % ROWF;
    x = x * 1664525u + 1013904223u; const unsigned B = (x >> 8) % ROWF;
    return A | B << 16;
}

__device__ __forceinline__ void glds_piece(const float *rowp, float *sdst_piece, unsigned lane)
{
    __builtin_amdgcn_global_load_lds((glb_void *)(rowp + lane * 4), (lds_void *)(sdst_piece), 16, 0, 0);
}

template <int MODE>
__global__ void __launch_bounds__(NT) stage(const float *__restrict__ psi, float *__restrict__ out, unsigned seed, int n_rows, int n_out_rows,
                                            int share, int row0)
{
    extern __shared__ float sR[];
    unsigned pk[CPT];
    double pab[CPT];
    unsigned tl = threadIdx.x;
    const unsigned lane = tl & 63, wave = tl >> 6;
    // workgroups b, b + 8, b + 16, ... sit on one XCD; `share` consecutive ones of them walk the same rows
    const unsigned team = (blockIdx.x >> 3) / share * 8 + (blockIdx.x & 7);
    unsigned row = (team * 7919u) % n_rows;
    for (int k = 0; k < CPT; ++k) { pk[k] = col_pk(seed, blockIdx.x * NT + tl, k); pab[k] = 0.125 * k; }
    // first row: plain DMA + wait
    {
        const float *rp = psi + (size_t)row * LD;
        for (int i = 0; i < NP / 8; ++i) { const int p = wave + 8 * i; glds_piece(rp + p * 256, sR + p * 256, lane); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int s = 0; s < STAGES; ++s) {
        asm volatile("" : "+v"(tl));
        const unsigned nrow = row0 ? 0u : next_row(row, n_rows);
        const float *np_ = psi + (size_t)nrow * LD;
        if (MODE == 1) {                                 // the head of the next row into the slack, behind the gathers
#pragma unroll
            for (int i = 0; i < P0 / 8; ++i) { const int p = wave + 8 * i; glds_piece(np_ + p * 256, sR + SLACK_OFF / 4 + p * 256, lane); }
        }
        float *orow = out + (size_t)((blockIdx.x * 48u + s % 48) % n_out_rows) * OUTW;      // 48 rows of its own per workgroup
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            const unsigned jq = q * (NT * 4) + tl * 4;
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                const float c = sR[pk[k] & 0xffff], d = sR[pk[k] >> 16];
                v[e] = static_cast<float>(__builtin_fma(static_cast<double>(c) + static_cast<double>(d), 0.25, pab[k]));
            }
            __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(orow + jq));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave is done with the row
        if (MODE == 0) {
            for (int i = 0; i < NP / 8; ++i) { const int p = wave + 8 * i; glds_piece(np_ + p * 256, sR + p * 256, lane); }
        } else {
            for (int i = P0 / 8; i < NP / 8; ++i) { const int p = wave + 8 * i; glds_piece(np_ + p * 256, sR + p * 256, lane); }
            // (the slack pieces of THIS wave were issued first: vmcnt counts them out in order)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP / 8 - P0 / 8) : "memory");
            // ... a wave moves its own pieces: no barrier needed in between.  (In asm: hipcc puts a vmcnt(0) in front of any LDS
            // read it emits itself while an LDS-DMA is in flight, which would wait for the 12 pieces just issued.)
            {
                f4 t[P0 / 8];
                const unsigned sa = SLACK_OFF + wave * 1024 + lane * 16, da = wave * 1024 + lane * 16;
#pragma unroll
                for (int i = 0; i < P0 / 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"(sa), "n"(i * 8192) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < P0 / 8; ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(da), "v"(t[i]), "n"(i * 8192) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        row = nrow;
    }
}

int main(int argc, char **argv)
{
    const int n_rows = 30976, n_out_rows = 12288;
    float *psi, *out;
    const size_t psi_n = (size_t)(n_rows + 2) * LD;
    (void)hipMalloc(&psi, psi_n * 4); (void)hipMalloc(&out, (size_t)n_out_rows * OUTW * 4);
    std::vector<float> h(psi_n);
    for (size_t i = 0; i < psi_n; ++i) h[i] = (float)((i * 2654435761ull >> 7) & 0xfffff) / 1048576.0f;
    (void)hipMemcpy(psi, h.data(), psi_n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    void (*ks[])(const float *, float *, unsigned, int, int, int, int) = {stage<0>, stage<1>};
    const char *names[] = {"LDS-DMA, nothing overlapped", "LDS-DMA, 32 KB of the next row ahead"};
    const int lds_bytes = 160 * 1024;
    (void)argc; (void)argv;
    for (int m = 0; m < 2; ++m) {
        if (hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) { printf("LDS attr failed\n"); return 1; }
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int share = cfg == 3 ? 1 : (1 << cfg), row0 = cfg == 3;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(ks[m], dim3(256), dim3(NT), lds_bytes, 0, psi, out, 7u, n_rows, n_out_rows, share, row0);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            // verify workgroup 0's last stage
            unsigned row = 0;                                     // team 0 starts at row 0
            for (int s = 0; s < STAGES - 1; ++s) row = row0 ? 0u : next_row(row, n_rows);
            if (row0) row = 0;
            const size_t orow = (size_t)((STAGES - 1) % 48) * OUTW;
            std::vector<float> o(OUTW);
            (void)hipMemcpy(o.data(), out + orow, OUTW * 4, hipMemcpyDeviceToHost);
            long bad = 0;
            for (int t = 0; t < NT; ++t)
                for (int k = 0; k < CPT; ++k) {
                    const unsigned pkk = col_pk(7u, t, k);
                    const float c = h[(size_t)row * LD + (pkk & 0xffff)], d = h[(size_t)row * LD + (pkk >> 16)];
                    const float want = (float)__builtin_fma((double)c + (double)d, 0.25, 0.125 * k);
                    const int q = k / 4, e = k % 4;
                    if (o[q * (NT * 4) + t * 4 + e] != want) ++bad;
                }
            const double out_b = 256.0 * STAGES * OUTW * 4, in_b = 256.0 * STAGES * ROW_BYTES;
            printf("%-38s %s %8.3f ms = %6.3f us per stage; %5.2f TB/s written, %5.2f TB/s of rows requested; %ld wrong\n", names[m],
                   row0 ? "row 0 only " : (share == 1 ? "own rows   " : (share == 2 ? "2 share    " : "4 share    ")), best,
                   best * 1e3 / STAGES, out_b / (best * 1e-3) / 1e12, in_b / (best * 1e-3) / 1e12, bad);
        }
    }
    return 0;
}
