// How long does device memory take to come by?  hipMalloc / hipMallocAsync / hipMemCreate of tens of GB on a fresh process, after an
// 80 GB allocation like the matrix's; and the first kernel touch of the memory.   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_alloc tools/ubench_alloc.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t bytes) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096;
    for (; i < bytes; i += (size_t)gridDim.x * blockDim.x * 4096) p[i] = 1;
}
int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30;
    const size_t big = (argc > 1 ? atoll(argv[1]) : 80) * GB, pool = (argc > 2 ? atoll(argv[2]) : 42) * GB;
    hipStream_t s;
    hipStreamCreate(&s);
    void *m = nullptr;
    double t = now();
    hipMalloc(&m, big);
    printf("hipMalloc %zu GB: %.1f ms\n", big / GB, (now() - t) * 1e3);
    t = now();
    touch<<<4096, 256, 0, s>>>((char *)m, big);
    hipStreamSynchronize(s);
    printf("  first touch: %.1f ms\n", (now() - t) * 1e3);
    for (int rep = 0; rep < 2; ++rep) {
        void *p = nullptr;
        t = now();
        hipError_t e = hipMalloc(&p, pool);
        printf("hipMalloc %zu GB (%s): %.1f ms\n", pool / GB, hipGetErrorString(e), (now() - t) * 1e3);
        t = now();
        touch<<<4096, 256, 0, s>>>((char *)p, pool);
        hipStreamSynchronize(s);
        printf("  first touch: %.1f ms\n", (now() - t) * 1e3);
        t = now();
        hipFree(p);
        printf("  hipFree: %.1f ms\n", (now() - t) * 1e3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        void *p = nullptr;
        t = now();
        hipError_t e = hipMallocAsync(&p, pool, s);
        hipStreamSynchronize(s);
        printf("hipMallocAsync %zu GB (%s): %.1f ms\n", pool / GB, hipGetErrorString(e), (now() - t) * 1e3);
        if (e == hipSuccess) {
            t = now();
            touch<<<4096, 256, 0, s>>>((char *)p, pool);
            hipStreamSynchronize(s);
            printf("  first touch: %.1f ms\n", (now() - t) * 1e3);
            t = now();
            hipFreeAsync(p, s);
            hipStreamSynchronize(s);
            printf("  hipFreeAsync: %.1f ms\n", (now() - t) * 1e3);
        }
    }
    return 0;
}
