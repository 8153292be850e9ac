// After hipFree of a large device block, are device-to-host copies slower for a while?  (The driver clears released VRAM in the
// background with the copy engines.)  build: hipcc --offload-arch=gfx950 -O2 free_then_copy.hip -o free_then_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float *p, size_t n) { for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = 1.0f; }
int main(int argc, char **argv)
{
    const size_t MB = size_t(1) << 20, copy_bytes = 256 * MB;
    void *src = nullptr, *dst = nullptr;
    CK(hipMalloc(&src, copy_bytes));
    CK(hipHostMalloc(&dst, copy_bytes, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto copy = [&]() -> double { const double t0 = now_ms(); (void)hipMemcpyAsync(dst, src, copy_bytes, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); return now_ms() - t0; };
    for (int i = 0; i < 3; ++i) copy();
    std::printf("256 MB device -> pinned host, idle device: %.2f %.2f %.2f ms\n", copy(), copy(), copy());
    for (int a = 1; a < argc; ++a) {
        const size_t gb = std::strtoull(argv[a], nullptr, 10);
        void *big = nullptr;
        CK(hipMalloc(&big, gb << 30));
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, static_cast<float *>(big), (gb << 30) / 4);
        CK(hipStreamSynchronize(st));
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        const double c0 = copy();
        const double t0 = now_ms();
        CK(hipFree(big));
        const double t_free = now_ms() - t0;
        std::printf("%zu GB: copy before the free %.2f ms; hipFree returned after %.2f ms; copies after it (ms, [started at]):", gb, c0, t_free);
        for (int i = 0; i < 14; ++i) { const double at = now_ms() - t0; std::printf(" %.1f[%.0f]", copy(), at); }
        std::printf("\n");
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
    }
    return 0;
}
