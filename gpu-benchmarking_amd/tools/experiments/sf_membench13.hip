// sf_membench13.hip -- the arithmetic-free traffic shape of 3D nq = 12 / 14 / 16 (one element per workgroup: 11-27 KB read,
// 14-32 KB written) against the waves per element and the workgroups a CU holds: what can the matrix-core kernels reach?
// (sf_membench8's kernel at 131 072 elements.)
// Arithmetic-free copies of the one-element-per-wave traffic shape (element e: IN_D doubles read, OUT_D doubles
// written, both on the 16-byte word grid so every access is a whole aligned word; words that straddle two elements
// are read by both owners, as in the product kernel) with an element handled by WPE = 1, 2 or 4 waves, EPB elements
// per workgroup, XCD runs of 64 workgroups as the product kernels use, and a dynamic-LDS pad that sets the occupancy.
// The stores of a wave wait for all of its loads (the sweeps need the whole element), plus a workgroup barrier when
// several waves share the element.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>

#include "../../csrc/sf_common.h"

typedef double d2 __attribute__((ext_vector_type(2)));
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

template <int IN_D, int OUT_D, int WPE, int EPB>
__global__ __launch_bounds__(64 * WPE *EPB) void shape(const d2 *__restrict__ in, d2 *__restrict__ out, uint64_t nelmt)
{
    extern __shared__ double pad[];
    constexpr int T   = 64 * WPE;                // threads per element
    constexpr int NLD = (IN_D / 2 + 2 + T - 1) / T;
    constexpr int NST = (OUT_D / 2 + 2 + T - 1) / T;
    const int t       = threadIdx.x % T;
    const uint64_t e  = sf::logical_block<64>() * EPB + threadIdx.x / T;
    if (e >= nelmt)
        return; // whole waves leave together (T is a multiple of 64); no barrier below when WPE == 1
    const uint64_t iw0 = e * IN_D / 2, iw1 = ((e + 1) * IN_D + 1) / 2; // word range of the element's input
    const uint64_t ow0 = (e * OUT_D + 1) / 2, ow1 = (e + 1) * OUT_D / 2; // whole words of its output
    d2 x[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k)
    {
        x[k] = d2{0, 0};
        if (iw0 + k * T + t < iw1)
            x[k] = __builtin_nontemporal_load(in + iw0 + k * T + t);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
        s += x[k].x + x[k].y;
    if (s == 123.456)
        pad[threadIdx.x] = s;
    if constexpr (WPE > 1)
        __syncthreads();
#pragma unroll
    for (int k = 0; k < NST; ++k)
        if (ow0 + k * T + t < ow1)
            __builtin_nontemporal_store(x[k % NLD], out + ow0 + k * T + t);
}

static hipEvent_t e0, e1;
static void run(const char *label, double bytes, int reps, const std::function<void()> &f)
{
    f();
    CK(hipDeviceSynchronize());
    double tmin = 1e30, tsum = 0;
    for (int r = 0; r < reps; ++r)
    {
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tmin = ms < tmin ? ms : tmin;
        tsum += ms;
    }
    CK(hipGetLastError());
    std::printf("%-58s %8.1f GB/s (min) %8.1f GB/s (mean)  %.3f of 8 TB/s\n", label, bytes / tmin * 1e-6,
                bytes / (tsum / reps) * 1e-6, bytes / (tsum / reps) * 1e-6 / 8000.0);
    std::fflush(stdout);
}

static d2 *g_in, *g_out;
static const uint64_t kElmt = 131072;

template <int IN_D, int OUT_D, int WPE> void go(int lds_per_wg)
{
    auto kern = shape<IN_D, OUT_D, WPE, 1>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    char label[128];
    std::snprintf(label, sizeof label, "in %4d out %4d  %d wave(s)/element  LDS %6d B/workgroup (%d per CU)", IN_D, OUT_D, WPE,
                  lds_per_wg, 160 * 1024 / lds_per_wg);
    run(label, 8.0 * kElmt * (IN_D + OUT_D), 20, [&] { kern<<<(unsigned)kElmt, 64 * WPE, lds_per_wg>>>(g_in, g_out, kElmt); });
}

template <int IN_D, int OUT_D> void rows(int image)
{
    go<IN_D, OUT_D, 1>(image);
    go<IN_D, OUT_D, 2>(image);
    go<IN_D, OUT_D, 4>(image);
    go<IN_D, OUT_D, 1>(image / 2);
    go<IN_D, OUT_D, 2>(image / 2);
    go<IN_D, OUT_D, 4>(image / 2);
    go<IN_D, OUT_D, 1>(image / 4);
    go<IN_D, OUT_D, 2>(image / 4);
    go<IN_D, OUT_D, 4>(image / 4);
    go<IN_D, OUT_D, 4>(image / 8);
    std::printf("\n");
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc((void **)&g_in, 8ull * kElmt * 3375 + 256));
    CK(hipMalloc((void **)&g_out, 8ull * kElmt * 4096 + 256));
    CK(hipMemset(g_in, 0, 8ull * kElmt * 3375 + 256));
    CK(hipMemset(g_out, 0, 8ull * kElmt * 4096 + 256));
    rows<1331, 1728>(13824);  // nq 12
    rows<2197, 2744>(21952);  // nq 14
    rows<3375, 4096>(32768);  // nq 16
    return 0;
}
