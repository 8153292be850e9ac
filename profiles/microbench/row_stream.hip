// The memory ceiling for the level kernels' access pattern: 256 persistent workgroups x 1024
// threads, each moving whole 96 KB rows of a 2.4 GB matrix with 16-byte accesses -- random rows,
// no two workgroups on the same row (no L2 reuse), no compute.
//   reads only / writes only / one read + one write per stage (a B stage) / the 1.5 : 1 mix of an
//   upper level of cfg4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ROW = 24576, NT = 1024;

template <int RD, int WR>   // rows read / written per stage (x2 to allow 3:2)
__global__ void __launch_bounds__(1024) stream(const float *__restrict__ in, float *__restrict__ out, int n_rows, int stages)
{
    const unsigned tl = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned r = blockIdx.x;
    for (int s = 0; s < stages; ++s) {
        for (int a = 0; a < RD; ++a) {
            r = (r + 256u * 37u) % n_rows;                 // distinct rows per workgroup and stage
            const float *src = in + (size_t)r * ROW;
#pragma unroll
            for (int k = 0; k < 6; ++k) { const f4 v = *reinterpret_cast<const f4 *>(src + (tl + k * NT) * 4); acc += v; }
        }
        for (int a = 0; a < WR; ++a) {
            r = (r + 256u * 37u) % n_rows;
            float *dst = out + (size_t)r * ROW;
#pragma unroll
            for (int k = 0; k < 6; ++k) *reinterpret_cast<f4 *>(dst + (tl + k * NT) * 4) = acc + (float)k;
        }
    }
    if (acc[0] == 12345.f) out[0] = acc[1];
}

template <int RD, int WR>
static void run(const char *name, const float *in, float *out, int n_rows, int stages)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((stream<RD, WR>), dim3(256), dim3(1024), 0, 0, in, out, n_rows, stages);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * stages * (RD + WR) * ROW * 4;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const int n_rows = 24576;
    float *in, *out;
    (void)hipMalloc(&in, (size_t)n_rows * ROW * 4); (void)hipMalloc(&out, (size_t)n_rows * ROW * 4);
    (void)hipMemset(in, 0, (size_t)n_rows * ROW * 4);
    run<1, 0>("row reads only", in, out, n_rows, 96);
    run<0, 1>("row writes only", in, out, n_rows, 96);
    run<1, 1>("1 read : 1 write (a B stage)", in, out, n_rows, 96);
    run<3, 2>("3 reads : 2 writes (upper level of cfg4)", in, out, n_rows, 48);
    return 0;
}
