// Which runtime calls on the first thread wait for a slow hipMalloc (VRAM an earlier process used) on a second thread?  Each kind of
// call is timed alone in a loop while the allocation runs; printed: the slowest instance of each.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_alloc3.co tools/ubench_alloc3.hip -lpthread; run after another process has used the memory
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t bytes) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 256;
    for (; i < bytes; i += (size_t)gridDim.x * blockDim.x * 256) p[i] += 1;
}
int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30;
    const size_t big = (argc > 1 ? atoll(argv[1]) : 80) * GB, pool = (argc > 2 ? atoll(argv[2]) : 42) * GB;
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    hipStream_t s;
    hipStreamCreate(&s);
    char *m = nullptr;
    hipMalloc(&m, big);
    touch<<<4096, 256, 0, s>>>(m, big);
    hipStreamSynchronize(s);
    void *pinned = nullptr;
    hipHostMalloc(&pinned, 1 << 20);
    std::vector<char> pageable(1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // every kind of call once before the allocation starts: queues, staging buffers and signals exist
    hipMemcpyAsync(pinned, m, 65536, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    hipMemcpy(pageable.data(), m, 65536, hipMemcpyDeviceToHost);
    hipMemcpyAsync(m, pinned, 65536, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    hipMemcpy(m, pageable.data(), 65536, hipMemcpyHostToDevice);
    hipMemsetAsync(m, 0, 1 << 20, s); hipStreamSynchronize(s);
    hipMemcpyAsync(m + GB, m, 1 << 20, hipMemcpyDeviceToDevice, s); hipStreamSynchronize(s);
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
    const char *names[] = {"kernel launch", "hipStreamSynchronize", "hipMemcpyAsync D2H pinned 64 KB + sync", "hipMemcpy D2H pageable 64 KB", "hipMemcpyAsync H2D pinned + sync",
                           "hipEventRecord x2 + hipEventSynchronize + ElapsedTime", "hipMemsetAsync + sync", "hipMalloc+hipFree 1 MB", "hipHostMalloc+hipHostFree 1 MB",
                           "hipStreamQuery", "hipMemcpyAsync D2D 1 MB + sync", "hipMemcpy H2D pageable 64 KB"};
    const int kinds = 12;
    double worst[kinds] = {0};
    int count[kinds] = {0};
    const double t0 = now();
    int round = 0;
    while (!done && now() - t0 < 30) {
        for (int k = 0; k < kinds; ++k) {
            if (only >= 0 && k != only && k > 1) continue;
            const double t = now();
            switch (k) {
            case 0: touch<<<4096, 256, 0, s>>>(m, GB); break;
            case 1: hipStreamSynchronize(s); break;
            case 2: hipMemcpyAsync(pinned, m, 65536, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); break;
            case 3: hipMemcpy(pageable.data(), m, 65536, hipMemcpyDeviceToHost); break;
            case 4: hipMemcpyAsync(m, pinned, 65536, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); break;
            case 5: { hipEventRecord(e0, s); touch<<<64, 256, 0, s>>>(m, 1 << 20); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); break; }
            case 6: hipMemsetAsync(m, 0, 1 << 20, s); hipStreamSynchronize(s); break;
            case 7: { void *q = nullptr; hipMalloc(&q, 1 << 20); hipFree(q); break; }
            case 8: { void *q = nullptr; hipHostMalloc(&q, 1 << 20); hipHostFree(q); break; }
            case 9: (void)hipStreamQuery(s); break;
            case 10: hipMemcpyAsync(m + GB, m, 1 << 20, hipMemcpyDeviceToDevice, s); hipStreamSynchronize(s); break;
            case 11: hipMemcpy(m, pageable.data(), 65536, hipMemcpyHostToDevice); break;
            }
            worst[k] = std::max(worst[k], (now() - t) * 1e3);
            ++count[k];
        }
        ++round;
    }
    th.join();
    printf("hipMalloc of %zu GB on the second thread: %.1f ms; %d rounds on the first meanwhile\n", pool / GB, t_malloc, round);
    for (int k = 0; k < kinds; ++k)
        if (count[k]) printf("  %-58s slowest %9.2f ms  (%d calls)\n", names[k], worst[k], count[k]);
    return 0;
}
