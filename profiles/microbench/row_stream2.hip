// Follow-up to row_stream.hip (round 2): does the memory ceiling of the level kernels' access pattern
// move with (a) non-temporal row stores, (b) 512-thread workgroups (two per CU), (c) the write-heavy
// mix of the final level of cfg4 (1 row read : ~3 rows written)?  Whole rows of 96 KB / 124 KB,
// 16-byte accesses, random rows, no reuse, no compute.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT, int RD, int WR, int K, bool NTS, bool NTL = false>   // K float4 per thread per row; NTS / NTL: non-temporal stores / loads
__global__ void __launch_bounds__(NT) stream(const float *__restrict__ in, float *__restrict__ out, int n_rows, int stages)
{
    constexpr int ROW = NT * K * 4;
    const unsigned tl = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned r = blockIdx.x;
    for (int s = 0; s < stages; ++s) {
        for (int a = 0; a < RD; ++a) {
            r = (r + 256u * 37u) % n_rows;
            const float *src = in + (size_t)r * ROW;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const f4 *q = reinterpret_cast<const f4 *>(src + (tl + k * NT) * 4);
                const f4 v = NTL ? __builtin_nontemporal_load(q) : *q;
                acc += v;
            }
        }
        for (int a = 0; a < WR; ++a) {
            r = (r + 256u * 37u) % n_rows;
            float *dst = out + (size_t)r * ROW;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                f4 *p = reinterpret_cast<f4 *>(dst + (tl + k * NT) * 4);
                if (NTS) __builtin_nontemporal_store(acc + (float)k, p); else *p = acc + (float)k;
            }
        }
    }
    if (acc[0] == 12345.f) out[0] = acc[1];
}

template <int NT, int RD, int WR, int K, bool NTS, bool NTL = false>
static void run(const char *name, const float *in, float *out, size_t bytes_buf, int stages, int grid)
{
    constexpr int ROW = NT * K * 4;
    const int n_rows = (int)(bytes_buf / ((size_t)ROW * 4));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((stream<NT, RD, WR, K, NTS, NTL>), dim3(grid), dim3(NT), 0, 0, in, out, n_rows, stages);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)grid * stages * (RD + WR) * ROW * 4;
    printf("%-64s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}

// Pipelined variant (what the level kernels do since the staging registers are reloaded behind the LDS
// writes): the K loads of the NEXT read are in flight while the rows of this step are written.
template <int NT, int RD, int WR, int K>
__global__ void __launch_bounds__(NT) stream_pipe(const float *__restrict__ in, float *__restrict__ out, int n_rows, int stages)
{
    constexpr int ROW = NT * K * 4;
    const unsigned tl = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned r = blockIdx.x, rr = blockIdx.x * 7919u;
    f4 v[K];
    rr = (rr + 256u * 37u) % n_rows;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const f4 *>(in + (size_t)rr * ROW + (tl + k * NT) * 4);
    for (int s = 0; s < stages * RD; ++s) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
        rr = (rr + 256u * 37u) % n_rows;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const f4 *>(in + (size_t)rr * ROW + (tl + k * NT) * 4);
        // WR rows are written per RD rows read: spread the writes over the reads
        const int w0 = s * WR / RD, w1 = (s + 1) * WR / RD;
        for (int a = w0; a < w1; ++a) {
            r = (r + 256u * 37u) % n_rows;
            float *dst = out + (size_t)r * ROW;
#pragma unroll
            for (int k = 0; k < K; ++k)
                __builtin_nontemporal_store(acc + (float)k, reinterpret_cast<f4 *>(dst + (tl + k * NT) * 4));
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
    if (acc[0] == 12345.f) out[0] = acc[1];
}

template <int NT, int RD, int WR, int K>
static void run_pipe(const char *name, const float *in, float *out, size_t bytes_buf, int stages, int grid)
{
    constexpr int ROW = NT * K * 4;
    const int n_rows = (int)(bytes_buf / ((size_t)ROW * 4));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((stream_pipe<NT, RD, WR, K>), dim3(grid), dim3(NT), 0, 0, in, out, n_rows, stages);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)grid * stages * (RD + WR) * ROW * 4;
    printf("%-64s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t buf = (size_t)3 << 30;
    float *in, *out;
    (void)hipMalloc(&in, buf); (void)hipMalloc(&out, buf);
    (void)hipMemset(in, 0, buf); (void)hipMemset(out, 0, buf);
    run<1024, 1, 0, 6, false>("1024 thr, 96 KB rows: reads only", in, out, buf, 96, 256);
    run<1024, 0, 1, 6, false>("1024 thr, 96 KB rows: writes only", in, out, buf, 96, 256);
    run<1024, 0, 1, 6, true >("1024 thr, 96 KB rows: writes only, nt", in, out, buf, 96, 256);
    run<1024, 3, 2, 6, false>("1024 thr, 96 KB rows: 3 reads : 2 writes", in, out, buf, 48, 256);
    run<1024, 3, 2, 6, true >("1024 thr, 96 KB rows: 3 reads : 2 writes, nt stores", in, out, buf, 48, 256);
    run<1024, 1, 3, 8, false>("1024 thr, 128 KB rows: 1 read : 3 writes (final level)", in, out, buf, 48, 256);
    run<1024, 1, 3, 8, true >("1024 thr, 128 KB rows: 1 read : 3 writes, nt stores", in, out, buf, 48, 256);
    run<1024, 1, 0, 6, false, true>("1024 thr, 96 KB rows: reads only, nt loads", in, out, buf, 96, 256);
    run<1024, 3, 2, 6, true, true>("1024 thr, 96 KB rows: 3 reads : 2 writes, nt stores + nt loads", in, out, buf, 48, 256);
    run<1024, 1, 3, 8, true, true>("1024 thr, 128 KB rows: 1 read : 3 writes, nt stores + nt loads", in, out, buf, 48, 256);
    run<512, 3, 2, 12, false>("512 thr x 256 WG, 96 KB rows: 3 reads : 2 writes", in, out, buf, 48, 256);
    run<512, 3, 2, 12, false>("512 thr x 512 WG, 96 KB rows: 3 reads : 2 writes", in, out, buf, 24, 512);
    run<512, 1, 3, 16, false>("512 thr x 256 WG, 128 KB rows: 1 read : 3 writes", in, out, buf, 48, 256);
    run_pipe<1024, 3, 2, 6>("1024 thr, 96 KB rows: 3 reads : 2 writes, nt stores, pipelined", in, out, buf, 48, 256);
    run_pipe<512, 3, 2, 12>("512 thr, 96 KB rows: 3 reads : 2 writes, nt stores, pipelined", in, out, buf, 48, 256);
    run_pipe<1024, 1, 3, 8>("1024 thr, 128 KB rows: 1 read : 3 writes, nt stores, pipelined", in, out, buf, 48, 256);
    run_pipe<512, 1, 3, 16>("512 thr, 128 KB rows: 1 read : 3 writes, nt stores, pipelined", in, out, buf, 48, 256);
    run<256, 1, 1, 6, false>("256 thr x 1024 WG, 24 KB rows: 1 read : 1 write", in, out, buf, 96, 1024);
    run<256, 1, 1, 6, false>("256 thr x 2048 WG, 24 KB rows: 1 read : 1 write", in, out, buf, 48, 2048);
    return 0;
}
