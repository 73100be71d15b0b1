// Accuracy of v_rcp_f64 (and of the Newton refinements built on it) on gfx950, for the EM kernel's correctly
// rounded quotient (csrc/em_kernels.hip: div_exact).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rcp.hip -o gpurun_out/ubench_rcp && ./gpurun_out/ubench_rcp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned int mix32(unsigned int x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// res[0..2]: max relative error (as double bits via atomicMax on the ordered integer image) of the raw seed, after one
// and after two Newton refinements, relative to the exactly rounded 1/x; res[3..4]: quotient mismatches vs IEEE divide
// with one / two refinements + residual correction.
__global__ void k(unsigned long long seed, int per_thread, unsigned long long *res)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int st = mix32((unsigned int)(tid ^ seed) + 0x9e3779b9u * (unsigned int)(seed >> 32));
    double e0 = 0, e1 = 0, e2 = 0;
    unsigned long long bad1 = 0, bad2 = 0;
    for (int it = 0; it < per_thread; ++it) {
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int a = st;
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int b = st;
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int c = st;
        // den: a float32 in [2^-60, 4) widened to double (what (p0+p1)+p2 is); num: a double in [0, 2*den]
        const float s = __uint_as_float((((a >> 23) % 62u + 67u) << 23) | (a & 0x7FFFFFu));
        const double den = (double)s;
        const float p1 = __uint_as_float((((b >> 23) % 62u + 67u) << 23) | (b & 0x7FFFFFu));
        const float p2 = __uint_as_float((((c >> 23) % 62u + 67u) << 23) | (c & 0x7FFFFFu));
        const double num = __builtin_fma(2.0, (double)p2, (double)p1);
        const double exact = 1.0 / den;
        double r = __builtin_amdgcn_rcp(den);
        e0 = fmax(e0, fabs(r - exact) / exact);
        double e = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e, r);
        e1 = fmax(e1, fabs(r - exact) / exact);
        {
            double q = num * r;
            const double rem = __builtin_fma(-den, q, num);
            q = __builtin_fma(rem, r, q);
            bad1 += __double_as_longlong(q) != __double_as_longlong(num / den);
        }
        e = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e, r);
        e2 = fmax(e2, fabs(r - exact) / exact);
        {
            double q = num * r;
            const double rem = __builtin_fma(-den, q, num);
            q = __builtin_fma(rem, r, q);
            bad2 += __double_as_longlong(q) != __double_as_longlong(num / den);
        }
    }
    atomicMax(res + 0, (unsigned long long)__double_as_longlong(e0));
    atomicMax(res + 1, (unsigned long long)__double_as_longlong(e1));
    atomicMax(res + 2, (unsigned long long)__double_as_longlong(e2));
    if (bad1) atomicAdd(res + 3, bad1);
    if (bad2) atomicAdd(res + 4, bad2);
}

int main()
{
    unsigned long long *d, h[5] = {0, 0, 0, 0, 0};
    CHECK(hipMalloc(&d, sizeof h));
    CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    const int per = 4096;
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, 20260313ull, per, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double e[3];
    for (int i = 0; i < 3; ++i) memcpy(&e[i], &h[i], 8);
    printf("pairs %.3g\n", 8192.0 * 256 * per);
    printf("v_rcp_f64 seed        max rel err %.3g = 2^%.1f\n", e[0], log2(e[0]));
    printf("after 1 refinement    max rel err %.3g = 2^%.1f   quotient mismatches vs IEEE: %llu\n", e[1], log2(e[1]), h[3]);
    printf("after 2 refinements   max rel err %.3g = 2^%.1f   quotient mismatches vs IEEE: %llu\n", e[2], log2(e[2]), h[4]);
    return 0;
}
