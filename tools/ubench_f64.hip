// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the
// exact EM / assignment kernels are made of, on gfx950.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o gpurun_out/ubench_f64 && ./gpurun_out/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;

// 8 independent chains per op so that dependent-issue latency does not limit a single wave.
#define BODY8(ASM) \
    asm volatile(ASM(0,8) ASM(1,9) ASM(2,10) ASM(3,11) ASM(4,12) ASM(5,13) ASM(6,14) ASM(7,15) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
                 "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));

#define A_MUL(i,j) "v_mul_f64 %" #i ", %" #i ", %16\n"
#define A_ADD(i,j) "v_add_f64 %" #i ", %" #i ", %16\n"
#define A_FMA(i,j) "v_fma_f64 %" #i ", %" #i ", %16, %16\n"
#define A_RCP(i,j) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_CVT_D2F(i,j) "v_cvt_f32_f64 %" #j ", %" #i "\n"   /* f <- d */
#define A_FIXUP(i,j) "v_div_fixup_f64 %" #i ", %" #i ", %16, %16\n"
#define A_FMAS(i,j) "v_div_fmas_f64 %" #i ", %" #i ", %16, %16\n"
#define A_SCALE(i,j) "v_div_scale_f64 %" #i ", vcc, %" #i ", %16, %" #i "\n"
#define A_LOG32(i,j) "v_log_f32 %" #j ", %" #j "\n"
#define A_ADD32(i,j) "v_add_f32 %" #j ", %" #j ", %" #j "\n"
#define A_FMA32(i,j) "v_fma_f32 %" #j ", %" #j ", %" #j ", %" #j "\n"
#define A_MOV(i,j) "v_mov_b32 %" #j ", %" #j "\n"
#define A_LDEXP(i,j) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define A_FREXPM(i,j) "v_frexp_mant_f64 %" #i ", %" #i "\n"
#define A_RNDNE(i,j) "v_rndne_f64 %" #i ", %" #i "\n"
#define A_MAX(i,j) "v_max_f64 %" #i ", %" #i ", %16\n"
#define A_PKFMA(i,j) "v_pk_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define A_PKMUL(i,j) "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
#define A_PKADD(i,j) "v_pk_add_f32 %" #i ", %" #i ", %16\n"

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, long long *cyc)
{
    double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    float f0 = (float)d0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    double c = 1.0000001;
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) { BODY8(A_MUL) }
        if (OP == 1) { BODY8(A_ADD) }
        if (OP == 2) { BODY8(A_FMA) }
        if (OP == 3) { BODY8(A_RCP) }
        if (OP == 4) { BODY8(A_CVT_D2F) }
        if (OP == 5) {   // v_cvt_f64_f32: d(i) <- f(8+i)
            asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                         "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        }
        if (OP == 6) { BODY8(A_FIXUP) }
        if (OP == 7) { BODY8(A_FMAS) }
        if (OP == 8) { BODY8(A_SCALE) }
        if (OP == 9) { BODY8(A_LOG32) }
        if (OP == 10) { BODY8(A_ADD32) }
        if (OP == 11) { BODY8(A_FMA32) }
        if (OP == 12) { BODY8(A_MOV) }
        if (OP == 13) { BODY8(A_LDEXP) }
        if (OP == 14) { BODY8(A_FREXPM) }
        if (OP == 15) { BODY8(A_RNDNE) }
        if (OP == 16) { BODY8(A_MAX) }
        if (OP == 17) { BODY8(A_PKFMA) }
        if (OP == 18) { BODY8(A_PKMUL) }
        if (OP == 19) { BODY8(A_PKADD) }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(double *, double, long long *);
struct Op { const char *name; kern_t fn; };

int main()
{
    Op ops[] = {{"v_mul_f64", k<0>}, {"v_add_f64", k<1>}, {"v_fma_f64", k<2>}, {"v_rcp_f64", k<3>}, {"v_cvt_f32_f64", k<4>},
                {"v_cvt_f64_f32", k<5>}, {"v_div_fixup_f64", k<6>}, {"v_div_fmas_f64", k<7>}, {"v_div_scale_f64", k<8>},
                {"v_log_f32", k<9>}, {"v_add_f32", k<10>}, {"v_fma_f32", k<11>}, {"v_mov_b32", k<12>}, {"v_ldexp_f64", k<13>},
                {"v_frexp_mant_f64", k<14>}, {"v_rndne_f64", k<15>}, {"v_max_f64", k<16>}, {"v_pk_fma_f32", k<17>},
                {"v_pk_mul_f32", k<18>}, {"v_pk_add_f32", k<19>}};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    CHECK(hipMalloc(&cyc, sizeof(long long) * cus * 8));
    printf("%-18s %8s %8s %8s   (cycles per wave-instruction per SIMD; s_memtime clock at 100 MHz -> x clk ratio unknown, use wall)\n", "op", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto &op : ops) {
        printf("%-18s", op.name);
        for (int bpc = 1; bpc <= 4; bpc *= 2) {   // blocks of 256 threads per CU = waves per SIMD
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(op.fn, dim3(cus * bpc), dim3(256), 0, 0, out, 1.5, cyc);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(op.fn, dim3(cus * bpc), dim3(256), 0, 0, out, 1.5, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // wave-instructions per SIMD = ITERS*8*bpc ; assume 2.4 GHz for the conversion (reported as-is)
            const double cyc_per = (double)ms * 1e-3 * 2.4e9 / ((double)ITERS * 8 * bpc);
            printf(" %8.2f", cyc_per);
        }
        printf("\n");
    }
    return 0;
}
