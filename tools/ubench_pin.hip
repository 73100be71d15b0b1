// How long page-locked staging costs, and what it could cost: hipHostMalloc / hipHostFree of the ingest's staging size against
// mmap (+ transparent huge pages, touched by several threads) + hipHostRegister; H2D rate from each.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_pin tools/ubench_pin.hip -lpthread && ./ubench_pin [MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 272) << 20;
    void *d = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipDeviceSynchronize());
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[256] = "?";
    if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
    printf("transparent_hugepage: %s", line);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void *h = nullptr;
        CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        double t1 = now();
        memset(h, 1, bytes);
        double t2 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t3 = now();
        CK(hipHostFree(h));
        double t4 = now();
        printf("hipHostMalloc %.1f ms, first touch %.1f ms, H2D %.1f ms (%.1f GB/s), hipHostFree %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
               bytes / 1e9 / (t3 - t2), (t4 - t3) * 1e3);
    }
    for (int huge = 0; huge < 2; ++huge)
        for (int threads : {1, 8}) {
            double t0 = now();
            const size_t al = (size_t)2 << 20;
            char *raw = (char *)mmap(nullptr, bytes + al, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (raw == MAP_FAILED) return 1;
            char *h = (char *)(((uintptr_t)raw + al - 1) & ~(uintptr_t)(al - 1));
            if (huge) madvise(h, bytes, MADV_HUGEPAGE);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([=] {
                    const size_t a = bytes * t / threads, b = bytes * (t + 1) / threads;
                    for (size_t i = a; i < b; i += 4096) h[i] = 1;
                });
            for (auto &x : th) x.join();
            double t1 = now();
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            double t2 = now();
            CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
            double t3 = now();
            CK(hipHostUnregister(h));
            double t4 = now();
            munmap(raw, bytes + al);
            double t5 = now();
            printf("mmap%s + touch on %d thread(s) %.1f ms, hipHostRegister %.1f ms, H2D %.1f ms (%.1f GB/s), unregister %.1f ms, munmap %.1f ms\n", huge ? " + MADV_HUGEPAGE" : "",
                   threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, bytes / 1e9 / (t3 - t2), (t4 - t3) * 1e3, (t5 - t4) * 1e3);
        }
    {   // pageable memory straight into hipMemcpy
        char *h = (char *)malloc(bytes);
        memset(h, 1, bytes);
        double t0 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        printf("pageable H2D %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / 1e9 / (t1 - t0));
        free(h);
    }
    {   // big device allocations
        for (size_t gb : {1, 3}) {
            double t0 = now();
            void *p = nullptr;
            CK(hipMalloc(&p, gb << 30));
            double t1 = now();
            CK(hipFree(p));
            double t2 = now();
            printf("hipMalloc %zu GiB %.1f ms, hipFree %.1f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
        }
    }
    CK(hipFree(d));
    return 0;
}
