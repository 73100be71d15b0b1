// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes the coded kernels use (MI355X_MICROARCH.md, HBM: "on gfx950
// FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Kernels W = 1, 2, 4: every lane streams 4 x W bytes per load, coalesced
// (a wavefront's load = 256 x W contiguous bytes), over BYTES bytes in total -- far beyond the Infinity Cache -- and stores one dword per
// thread.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o gpurun_out/ubench_fetch && rocprofv3 --kernel-trace --pmc FETCH_SIZE -- gpurun_out/ubench_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(256) void stream_read(const unsigned *__restrict__ src, unsigned *__restrict__ out, long long words_per_block)
{
    typedef unsigned vec __attribute__((ext_vector_type(W)));
    const vec *p = reinterpret_cast<const vec *>(src + (long long)blockIdx.x * words_per_block);
    const long long n = words_per_block / W;
    unsigned acc = 0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const vec v = __builtin_nontemporal_load(p + i);
        if (W == 1) acc += v[0];
        else if (W == 2) acc += v[0] ^ v[1];
        else acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const long long bytes = 8ll << 30;                       // 8 GiB read once per launch
    const int blocks = 8192;
    const long long words_per_block = bytes / 4 / blocks;
    unsigned *src = nullptr, *out = nullptr;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&out, sizeof(unsigned) * blocks * 256));
    CHECK(hipMemset(src, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_read<1>, dim3(blocks), dim3(256), 0, 0, src, out, words_per_block);
        hipLaunchKernelGGL(stream_read<2>, dim3(blocks), dim3(256), 0, 0, src, out, words_per_block);
        hipLaunchKernelGGL(stream_read<4>, dim3(blocks), dim3(256), 0, 0, src, out, words_per_block);
    }
    CHECK(hipDeviceSynchronize());
    printf("{\"bytes_read_per_launch\": %lld, \"bytes_written_per_launch\": %lld}\n", bytes, (long long)sizeof(unsigned) * blocks * 256);
    return 0;
}
