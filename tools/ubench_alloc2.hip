// Does a slow hipMalloc (VRAM an earlier process used: cleared by the driver when handed out again) on a second thread hold up
// kernels the first thread keeps launching?   hipcc --offload-arch=gfx950 -O2 -o ubench_alloc2 tools/ubench_alloc2.hip -lpthread; run it twice
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t bytes) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 256;
    for (; i < bytes; i += (size_t)gridDim.x * blockDim.x * 256) p[i] += 1;
}
int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30;
    const size_t big = (argc > 1 ? atoll(argv[1]) : 80) * GB, pool = (argc > 2 ? atoll(argv[2]) : 42) * GB;
    hipStream_t s;
    hipStreamCreate(&s);
    void *m = nullptr;
    hipMalloc(&m, big);
    touch<<<4096, 256, 0, s>>>((char *)m, big);
    hipStreamSynchronize(s);
    std::atomic<int> done{0};
    double t_malloc = 0;
    void *p = nullptr;
    std::thread th([&] {
        hipSetDevice(0);
        const double t = now();
        hipMalloc(&p, pool);
        t_malloc = (now() - t) * 1e3;
        done = 1;
    });
    int launches = 0;
    double worst = 0;
    const double t0 = now();
    while (!done || launches < 20) {
        const double t = now();
        touch<<<4096, 256, 0, s>>>((char *)m, 8 * GB);
        hipStreamSynchronize(s);
        worst = std::max(worst, (now() - t) * 1e3);
        ++launches;
        if (now() - t0 > 20) break;
    }
    th.join();
    printf("hipMalloc of %zu GB on the second thread: %.1f ms; meanwhile %d kernels + syncs on the first, the slowest %.2f ms, %.2f ms each on average\n",
           pool / GB, t_malloc, launches, worst, (now() - t0) * 1e3 / launches);
    touch<<<4096, 256, 0, s>>>((char *)p, pool);
    hipStreamSynchronize(s);
    return 0;
}
