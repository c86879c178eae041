// sf_tune_mfma4.hip -- configuration sweep of the 4x4x4_4b matrix-core kernel (bwdtrans_mfma4.h) against the shipped
// AUTO kernel of the same order (development tool).  Built per order: -DTUNE_NQ=N.  Usage: sf_tune_mfma4_N [nelmt] [reps]
#include "../csrc/sf_dispatch.h"
#include "../csrc/wave_launch.h"
#include "tune_guard.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#ifndef TUNE_NQ
#define TUNE_NQ 28
#endif

using namespace sf;

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

static int g_reps = 15;
static hipEvent_t g_e0, g_e1;
static double *g_ref; // output of the reference variant, for a max-abs-difference check
static size_t g_nout;

__global__ void maxdiff_kernel(const double *a, const double *b, size_t n, double *res)
{
    double m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const double d = fabs(a[i] - b[i]);
        m              = fmax(m, d == d ? d : 1e300); // NaN (an output nobody wrote) must not hide
    }
    for (int off = 32; off; off >>= 1)
        m = fmax(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax((unsigned long long *)res, (unsigned long long)__double_as_longlong(m)); // m >= 0: order-preserving
}

template <class F> static void run(const char *label, const QuadArgs &a, F launch, bool is_ref = false)
{
    const int NQ = TUNE_NQ;
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(double) * a.nelmt * (NQ - 1) * (NQ - 1), sizeof(double) * a.nelmt * NQ * NQ,
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    CK(hipMemset(a.out, 0xff, sizeof(double) * g_nout)); // NaN pattern: unwritten outputs show up in the check
    int rc = launch();
    CK(hipDeviceSynchronize());
    if (rc != 0)
    {
        std::printf("%-44s rc=%d\n", label, rc);
        return;
    }
    double *dres, hres = 0;
    CK(hipMalloc((void **)&dres, sizeof(double)));
    CK(hipMemset(dres, 0, sizeof(double)));
    if (is_ref)
        CK(hipMemcpy(g_ref, a.out, sizeof(double) * g_nout, hipMemcpyDeviceToDevice));
    maxdiff_kernel<<<1024, 256>>>(a.out, g_ref, g_nout, dres);
    CK(hipMemcpy(&hres, dres, sizeof(double), hipMemcpyDeviceToHost));
    CK(hipFree(dres));
    std::vector<double> t;
    for (int r = 0; r < g_reps; ++r)
    {
        CK(hipEventRecord(g_e0, 0));
        launch();
        CK(hipEventRecord(g_e1, 0));
        CK(hipEventSynchronize(g_e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    double sum = 0;
    for (double v : t)
        sum += v;
    const double tmin = t[0], tmean = sum / t.size();
    const double dof = a.nelmt * nm * nm, bytes = a.nelmt * 8.0 * (nm * nm + (double)NQ * NQ);
    std::printf("%-44s min %8.4f mean %8.4f ms | %7.2f / %7.2f GDOF/s | %7.1f GB/s = %.3f of 8 TB/s (mean) | max|d| %.2e\n",
                label, tmin, tmean, dof / (tmin * 1e-3) * 1e-9, dof / (tmean * 1e-3) * 1e-9,
                bytes / (tmean * 1e-3) * 1e-9, bytes / (tmean * 1e-3) * 1e-9 / 8000.0, hres);
    std::fflush(stdout);
}

template <int EB, int WPB, int MW, int GJ, int K, int XG, int DYNB = 0, bool PEEL = true, bool SPLIT = false, int XR = 0>
static void m4(const QuadArgs &a)
{
    constexpr int NQ = TUNE_NQ;
    if constexpr (mfma4_lds_bytes<NQ, EB, WPB, true>() <= 160 * 1024 && mfma4_lds_bytes<NQ, EB, WPB>() > 160 * 1024)
    {
        char label[96];
        std::snprintf(label, sizeof label, "quad nq%d MFMA4 EB%d WPB%d MW%d GJ%d K%d xg%d shb lds %zu", NQ, EB, WPB, MW, GJ,
                      K, XG, mfma4_lds_bytes<NQ, EB, WPB, true>());
        run(label, a, [&]() { return launch_quad_mfma4<NQ, EB, WPB, MW, GJ, K, XG, true>(a, 0); });
    }
    else if constexpr (mfma4_lds_bytes<NQ, EB, WPB>() <= 160 * 1024)
    {
        char label[96];
        std::snprintf(label, sizeof label, "quad nq%d MFMA4 EB%d WPB%d MW%d GJ%d K%d xg%d dyn%d%s xr%d lds %zu", NQ, EB, WPB, MW, GJ, K,
                      XG, DYNB, PEEL ? (SPLIT ? " split" : "") : " nopeel", XR, mfma4_lds_bytes<NQ, EB, WPB>());
        run(label, a, [&]() { return launch_quad_mfma4<NQ, EB, WPB, MW, GJ, K, XG, false, DYNB, PEEL, SPLIT, XR>(a, 0); });
    }
}

// ST ticket counters in lines of their own (quad_mfma4_kernel's ST): small batches without the one-line atomic limit
template <int EB, int WPB, int MW, int DYNB, int ST, int GW = 0> static void m4s(const QuadArgs &a)
{
    constexpr int NQ = TUNE_NQ;
    constexpr size_t lds = mfma4_lds_bytes<NQ, EB, WPB, true>();
    if constexpr (lds <= 160 * 1024)
    {
        auto kern = quad_mfma4_kernel<NQ, EB, WPB, MW, 4, 0, 0, true, DYNB, true, false, false, false, 0, ST>;
        if (lds > 48 * 1024)
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int bpc = 0, cus = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, kWave * WPB, lds));
        CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
        const uint64_t nchunk = (a.nelmt + EB - 1) / EB, need = (nchunk + WPB - 1) / WPB;
        uint64_t grid         = (uint64_t)(GW > 0 ? GW : bpc) * cus;
        if (grid > need)
            grid = need;
        static unsigned long long *ctr = nullptr;
        if (!ctr)
            CK(hipMalloc((void **)&ctr, 128 * 64));
        char label[96];
        std::snprintf(label, sizeof label, "quad nq%d MFMA4 EB%d WPB%d MW%d K0 dyn%d striped%d (%d wg/CU) lds %zu", NQ, EB, WPB, MW, DYNB,
                      ST, GW > 0 ? GW : bpc, lds);
        run(label, a, [&]() {
            (void)hipMemsetAsync(ctr, 0, 128 * 64, 0);
            kern<<<(unsigned)grid, kWave * WPB, lds>>>(a.b0, a.b1, a.in, a.out, a.nelmt, ctr, nullptr);
            return (int)hipGetLastError();
        });
    }
}

// shader-clock time per phase of the chunk loop (quad_mfma4_kernel with STAMP): persistent grid fed by the batch counter,
// or one chunk per wave
template <int EB, int WPB, int MW, int K, int XG, int DYNB, bool EFL = false> static void phases(const QuadArgs &a)
{
    constexpr int NQ = TUNE_NQ;
    auto kern        = quad_mfma4_kernel<NQ, EB, WPB, MW, 4, K, XG, true, DYNB, true, false, true, EFL>;
    constexpr size_t lds = mfma4_lds_bytes<NQ, EB, WPB, true>();
    if (lds > 48 * 1024)
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0, cus = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, kWave * WPB, lds));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const uint64_t nchunk = (a.nelmt + EB - 1) / EB;
    const uint64_t need   = (nchunk + (uint64_t)WPB * (K > 0 ? K : 1) - 1) / ((uint64_t)WPB * (K > 0 ? K : 1));
    uint64_t grid         = (uint64_t)bpc * cus;
    if (K != 0 || grid > need)
        grid = need;
    const size_t nslot = (size_t)grid * WPB * 8;
    unsigned long long *dev, *ctr, host[8] = {};
    CK(hipMalloc((void **)&dev, nslot * sizeof(unsigned long long)));
    CK(hipMalloc((void **)&ctr, 64));
    float ms = 0;
    CK(hipMemset(a.out, 0xff, sizeof(double) * g_nout)); // NaN pattern: unwritten outputs show up in the check
    for (int rep = 0; rep < 3; ++rep) // the last repetition is reported
    {
        CK(hipMemset(dev, 0, nslot * sizeof(unsigned long long)));
        CK(hipMemset(ctr, 0, 64));
        CK(hipEventRecord(g_e0, 0));
        kern<<<(unsigned)grid, kWave * WPB, lds>>>(a.b0, a.b1, a.in, a.out, a.nelmt, ctr, dev);
        CK(hipEventRecord(g_e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    }
    std::vector<unsigned long long> all(nslot);
    CK(hipMemcpy(all.data(), dev, nslot * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t w = 0; w < nslot / 8; ++w)
        for (int k = 0; k < 6; ++k)
            host[k] += all[8 * w + k];
    CK(hipFree(dev));
    CK(hipFree(ctr));
    double *dres, hres = 0;
    CK(hipMalloc((void **)&dres, sizeof(double)));
    CK(hipMemset(dres, 0, sizeof(double)));
    maxdiff_kernel<<<1024, 256>>>(a.out, g_ref, g_nout, dres);
    CK(hipMemcpy(&hres, dres, sizeof(double), hipMemcpyDeviceToHost));
    CK(hipFree(dres));
    std::printf("(stamped launch: %.3f ms, max|d| %.2e) ", ms, hres);
    const double n = (double)host[5];
    std::printf("phases%s nq%d EB%d WPB%d MW%d K%d dyn%d (%d blocks/CU): shader clocks per chunk: stage+issue %.0f | step 1 %.0f | "
                "step 2 %.0f | flush %.0f | wait next %.0f | sum %.0f (%llu chunks)\n",
                EFL ? " [stores per j group]" : "", NQ, EB, WPB, MW, K, DYNB, bpc, host[0] / n, host[1] / n, host[2] / n, host[3] / n, host[4] / n,
                (host[0] + host[1] + host[2] + host[3] + host[4]) / n, host[5]);
}

int main(int argc, char **argv)
{
    constexpr int NQ = TUNE_NQ, NM = NQ - 1;
    const size_t nelmt = argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)1 << 20;
    g_reps             = argc > 2 ? std::atoi(argv[2]) : 15;
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    double *b0, *b1, *in, *out;
    g_nout = nelmt * NQ * NQ;
    CK(hipMalloc((void **)&b0, sizeof(double) * NM * NQ));
    CK(hipMalloc((void **)&b1, sizeof(double) * NM * NQ));
    CK(hipMalloc((void **)&in, sizeof(double) * nelmt * NM * NM));
    CK(hipMalloc((void **)&out, sizeof(double) * g_nout));
    CK(hipMalloc((void **)&g_ref, sizeof(double) * g_nout));
    tune::capacity() = {sizeof(double) * nelmt * NM * NM, sizeof(double) * g_nout, sizeof(double) * NM * NQ};
    fill_random(b0, NM * NQ, 11, 0, 0);
    CK(hipFree(b1));
    b1 = b0; // the isotropic case of every benchmark run: both directions share one basis array
    fill_random(in, nelmt * NM * NM, 0x5F3759DF, 0, 0);
    CK(hipDeviceSynchronize());
    QuadArgs a{b0, b1, in, nullptr, out, nelmt};
    std::printf("nq %d, %zu elements, %d reps; reference = the generic LDS kernel (first row), then the shipped kernels\n",
                NQ, nelmt, g_reps);
    run("generic block/LDS (reference result)", a, [&]() { return launch_quad_generic(SF_VARIANT_BLOCK_LDS, NQ, NQ, a, 0); },
        true);
    if (argc > 3 && std::string(argv[3]) == "striped")
    {
        for (int rep = 0; rep < 2; ++rep)
        {
            m4<2, 4, 2, 4, 0, 0, 4>(a);
            m4<2, 4, 2, 4, 0, 0, 8>(a);
            m4s<2, 4, 2, 4, 1>(a); // the single counter through this launcher
            m4s<2, 4, 2, 4, 16>(a);
            m4s<2, 4, 2, 2, 16>(a);
            m4s<2, 4, 2, 1, 16>(a);
            m4s<2, 4, 2, 1, 64>(a);
            m4s<2, 4, 2, 2, 64>(a);
            m4s<2, 4, 2, 1, 8>(a);
            m4s<4, 4, 1, 1, 16>(a);
            m4s<4, 4, 1, 2, 16>(a);
            m4s<4, 4, 1, 1, 64>(a);
            m4s<1, 4, 4, 1, 64>(a);
            m4s<1, 4, 4, 2, 64>(a);
            m4s<1, 4, 3, 1, 64>(a);
        }
        return 0;
    }
    phases<2, 4, 2, 0, 0, 4>(a);
    phases<2, 4, 2, 1, 64, 0>(a);
    phases<2, 4, 2, 2, 64, 0>(a);
    phases<1, 4, 4, 1, 64, 0>(a);
    phases<4, 4, 1, 0, 0, 4>(a);
    phases<2, 4, 2, 0, 0, 4, true>(a);
    phases<2, 4, 2, 1, 64, 0, true>(a);
    phases<2, 4, 2, 2, 64, 0, true>(a);
    phases<1, 4, 4, 1, 64, 0, true>(a);
    if (argc > 3) // phases only
        return 0;
    for (int rep = 0; rep < 2; ++rep)
    {
        run("shipped wave kernel", a, [&]() { return launch_quad_wave_nq(NQ, a, 0); });
        run("shipped 16x16x4 matrix-core kernel", a, [&]() { return launch_quad_mfma_nq(NQ, a, 0); });
        m4<2, 4, 2, 4, 1, 64>(a);
        m4<2, 4, 2, 4, 2, 64>(a);
        m4<2, 4, 2, 4, 0, 0>(a);
        m4<2, 4, 2, 4, 0, 0, 4>(a);
        m4<2, 4, 2, 4, 0, 0, 4, false>(a); // the same without the peeled k remainder (differs at nm = 1, 2 mod 4 only)
        m4<2, 4, 2, 4, 0, 0, 4, true, true>(a); // ... with the unpaired i tile's spare blocks splitting the q tiles (odd tile counts)
        m4<2, 4, 2, 4, 1, 64, 0, true, true>(a);
        m4<2, 4, 2, 4, 2, 64, 0, true, true>(a);
        m4<2, 4, 2, 4, 1, 64, 0, false>(a);
        m4<2, 4, 2, 4, 0, 0, 8>(a);
        // batch tickets per XCD: runs of 4 / 16 / 64 neighbouring batches on one XCD
        m4<2, 4, 2, 4, 0, 0, 4, true, false, 4>(a);
        m4<2, 4, 2, 4, 0, 0, 4, true, false, 16>(a);
        m4<2, 4, 2, 4, 0, 0, 4, true, false, 64>(a);
        m4<2, 4, 2, 4, 0, 0, 2, true, false, 32>(a);
        m4<2, 4, 2, 4, 0, 0, 1, true, false, 64>(a);
        m4<4, 4, 1, 4, 0, 0, 4, true, false, 16>(a);
        m4<2, 8, 2, 4, 0, 0, 8>(a);
        m4<4, 4, 1, 4, 0, 0>(a);
        m4<4, 4, 1, 4, 0, 0, 4>(a);
        m4<4, 4, 1, 4, 0, 0, 8>(a);
        // short-lived workgroups at more independent phases per CU (round 3): one element per wave at 3 / 4 waves per
        // SIMD, two-wave and one-wave workgroups of two-element chunks
        m4<1, 4, 4, 4, 1, 64>(a);
        m4<1, 4, 3, 4, 1, 64>(a);
        m4<1, 4, 4, 4, 2, 64>(a);
        m4<1, 8, 4, 4, 1, 64>(a);
        m4<2, 2, 2, 4, 1, 64>(a);
        m4<2, 1, 2, 4, 1, 64>(a);
        m4<2, 3, 3, 4, 1, 64>(a);
        m4<2, 2, 2, 4, 2, 64>(a);
        // three waves per SIMD by registers (<= 168 VGPRs): with one shared basis copy three four-wave workgroups fit
        // the LDS up to nq = 26
        m4<2, 4, 3, 4, 1, 64>(a);
        m4<2, 4, 3, 4, 2, 64>(a);
        m4<2, 4, 3, 2, 1, 64>(a);
        m4<2, 4, 3, 4, 0, 0, 4>(a); // persistent, three waves per SIMD
        m4<2, 4, 3, 4, 0, 0, 8>(a);
        m4<2, 4, 3, 4, 4, 64>(a);
    }
    return 0;
}
