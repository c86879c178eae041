// sf_mfma_probe.hip -- what the fp64 pipes of this device deliver (development tool, gfx950).
//
//  rate:   wall-clock TFLOP/s and shader cycles per instruction of v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and
//          v_fma_f64 with 1, 2 and 4 waves per SIMD on every CU, and of MFMA + FMA issued together by one wave
//  layout: the lane maps of v_mfma_f64_4x4x4_4b_f64 (operand lane -> (block, row, k) / (block, k, col), result lane ->
//          (block, row, col)), found with one-hot operands -- the guides list the 16x16x4 maps only
// Usage: sf_mfma_probe [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                      \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);    \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

enum Mode
{
    M16 = 0,  // v_mfma_f64_16x16x4_f64: 2048 flop
    M4B = 1,  // v_mfma_f64_4x4x4_4b_f64: 512 flop
    VFMA = 2, // v_fma_f64: 128 flop
    MIX16 = 3, // one 16x16x4 MFMA + 16 FMAs per group (equal flops on both pipes)
    MIX4 = 4,  // one 4x4x4_4b MFMA + 4 FMAs per group
    M16F = 5,  // v_mfma_f32_16x16x4_f32: 2048 flop
    PKF = 6    // v_pk_fma_f32: 256 flop
};

template <int MODE> __global__ __launch_bounds__(64) void rate_kernel(int iters, double *sink, long long *cycles)
{
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double4_t c16[4] = {};
    double c4[8]     = {};
    double v[16];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f4 f16[4] = {};
    f2 pk[16], pka = {(float)a, (float)b}, pkb = {(float)b, (float)a};
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
        v[k]  = k;
        pk[k] = f2{(float)k, (float)k};
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (MODE == M16 || MODE == MIX16)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                c16[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c16[k], 0, 0, 0);
                if constexpr (MODE == MIX16)
                {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        v[j] = __builtin_fma(v[j], a, b);
                }
            }
        }
        if constexpr (MODE == M4B || MODE == MIX4)
        {
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                c4[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[k], 0, 0, 0);
                if constexpr (MODE == MIX4)
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[(4 * k + j) & 15] = __builtin_fma(v[(4 * k + j) & 15], a, b);
                }
            }
        }
        if constexpr (MODE == M16F)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                f16[k] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, f16[k], 0, 0, 0);
        }
        if constexpr (MODE == PKF)
        {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[j]) : "v"(pka), "v"(pkb));
        }
        if constexpr (MODE == VFMA)
        {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = __builtin_fma(v[j], a, b);
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        s += c16[k][0] + c16[k][1] + c16[k][2] + c16[k][3];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        s += c4[k];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        s += v[k] + pk[k].x + pk[k].y;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        s += f16[k][0] + f16[k][1] + f16[k][2] + f16[k][3];
    sink[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

template <int MODE> static void rate(const char *name, int cus, int wps, int iters, double *sink, long long *cyc)
{
    const int blocks = cus * 4 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, 64>>>(iters / 8, sink, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE><<<blocks, 64>>>(iters, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long c : h)
        mean += (double)c;
    mean /= blocks;
    double mfma_flop = 0, valu_flop = 0, n_mfma = 0, n_valu = 0;
    if (MODE == M16 || MODE == MIX16)
        mfma_flop = 4 * 2048.0, n_mfma = 4;
    if (MODE == M4B || MODE == MIX4)
        mfma_flop = 8 * 512.0, n_mfma = 8;
    if (MODE == VFMA)
        valu_flop = 16 * 128.0, n_valu = 16;
    if (MODE == M16F)
        mfma_flop = 4 * 2048.0, n_mfma = 4;
    if (MODE == PKF)
        valu_flop = 16 * 256.0, n_valu = 16;
    if (MODE == MIX16)
        valu_flop = 64 * 128.0, n_valu = 64;
    if (MODE == MIX4)
        valu_flop = 32 * 128.0, n_valu = 32;
    const double flops = (mfma_flop + valu_flop) * (double)iters * blocks;
    std::printf("%-22s %d wave(s)/SIMD  %8.3f ms  %7.2f TFLOP/s (mfma %6.2f + valu %6.2f)  wave-clocks/iter %8.1f "
                "(%4.0f MFMA + %4.0f FMA per iter)  counter clocks/ms %.0f\n",
                name, wps, ms, flops / ms * 1e-9, mfma_flop * iters * blocks / ms * 1e-9,
                valu_flop * iters * blocks / ms * 1e-9, mean / iters, n_mfma, n_valu, mean / ms);
    std::fflush(stdout);
}

// one-hot layout probe: D[la][lb][lane] for A = [lane == la], B = [lane == lb]
__global__ __launch_bounds__(64) void layout_kernel(double *d4, double *d16)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
        {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            d4[(la * 64 + lb) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const double4_t r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, double4_t{0, 0, 0, 0}, 0, 0, 0);
            for (int k = 0; k < 4; ++k)
                d16[((la * 64 + lb) * 64 + lane) * 4 + k] = r[k];
        }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device: %s, %d CUs, clock64 counter; iters %d\n", prop.gcnArchName, cus, iters);
    double *sink;
    long long *cyc;
    CK(hipMalloc((void **)&sink, sizeof(double) * 64 * cus * 4 * 8));
    CK(hipMalloc((void **)&cyc, sizeof(long long) * cus * 4 * 8));
    for (int wps : {1, 2, 4})
    {
        rate<M16>("mfma_f64_16x16x4", cus, wps, iters, sink, cyc);
        rate<M4B>("mfma_f64_4x4x4_4b", cus, wps, iters, sink, cyc);
        rate<M16F>("mfma_f32_16x16x4", cus, wps, iters, sink, cyc);
        rate<PKF>("pk_fma_f32", cus, wps, iters, sink, cyc);
        rate<VFMA>("v_fma_f64", cus, wps, iters, sink, cyc);
        rate<MIX16>("16x16x4 + 16 fma", cus, wps, iters / 4, sink, cyc);
        rate<MIX4>("4x4x4_4b + 4 fma", cus, wps, iters / 2, sink, cyc);
    }

    // ---- lane maps ---------------------------------------------------------------------------------------------
    double *d4, *d16;
    CK(hipMalloc((void **)&d4, sizeof(double) * 64 * 64 * 64));
    CK(hipMalloc((void **)&d16, sizeof(double) * 64 * 64 * 64 * 4));
    layout_kernel<<<1, 64>>>(d4, d16);
    CK(hipDeviceSynchronize());
    std::vector<double> h4(64 * 64 * 64), h16(64 * 64 * 64 * 4);
    CK(hipMemcpy(h4.data(), d4, sizeof(double) * h4.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h16.data(), d16, sizeof(double) * h16.size(), hipMemcpyDeviceToHost));
    std::printf("\nv_mfma_f64_4x4x4_4b_f64, one-hot A at lane la x one-hot B at lane lb -> result lanes (value 1):\n");
    std::printf("  (pairs with no result are different blocks or different k)\n");
    for (int la = 0; la < 64; ++la)
    {
        std::printf("  la %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h4[(la * 64 + lb) * 64 + l] != 0.0)
                    std::printf(" (lb %d -> lane %d)", lb, l);
        std::printf("\n");
    }
    std::printf("\nv_mfma_f64_16x16x4_f64 (cross-check of the documented maps), la = 0, 1, 16, 17 only:\n");
    for (int la : {0, 1, 16, 17})
    {
        std::printf("  la %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                for (int k = 0; k < 4; ++k)
                    if (h16[((la * 64 + lb) * 64 + l) * 4 + k] != 0.0)
                        std::printf(" (lb %d -> lane %d reg %d)", lb, l, k);
        std::printf("\n");
    }
    return 0;
}
