// sf_tune.hip -- variant sweep of the wave kernels on the current device (development tool).
// Built once per (TUNE_DIM, TUNE_NQ): sf_tune_hex8, sf_tune_quad8, ...   Usage: sf_tune_X [nelmt] [reps]
// For every case of tools/tune_cases.h it prints kernel time (hipEvent: min / median / mean over reps),
// GDOF/s (min and mean), algorithmic GB/s (8*(nm^d+nq^d) B/element) and sqrt(sum out^2) as a sanity value.
#include "../csrc/sf_dispatch.h"
#include "../csrc/wave_launch.h"
#include "experiments/hex_mfma2.h"
#include "experiments/hex_mfma4_pair.h"
#include "tune_guard.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#ifndef TUNE_DIM
#define TUNE_DIM 3
#endif
#ifndef TUNE_NQ
#define TUNE_NQ 8
#endif
#include "tune_cases.h"

using namespace sf;

#define CK(x)                                                                                      \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__,       \
                         __LINE__);                                                                \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

static int g_reps = 15;
static hipEvent_t g_e0, g_e1;

static void run(const char *label, double dof, double bytes, double *out, size_t nout,
                const std::function<int()> &launch)
{
    int rc = launch(); // warm-up (+ occupancy query)
    CK(hipDeviceSynchronize());
    if (rc != 0)
    {
        std::printf("%-44s rc=%d\n", label, rc);
        return;
    }
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
    const double tmin = t[0], tmed = t[t.size() / 2], tmean = sum / t.size();
    double ss = 0;
    sumsq_blocking(out, nout, &ss, 0);
    std::printf("%-44s min %8.4f med %8.4f mean %8.4f ms | %7.2f / %7.2f GDOF/s (min/mean) | %7.1f GB/s | norm %.10g\n",
                label, tmin, tmed, tmean, dof / (tmin * 1e-3) * 1e-9, dof / (tmean * 1e-3) * 1e-9,
                bytes / (tmin * 1e-3) * 1e-9, std::sqrt(ss));
    std::fflush(stdout);
}

static const char *out_name(int o)
{
    return o == OUT_ST8 ? "st8 " : (o == OUT_ST16 ? "st16" : "lds ");
}

template <int NQ, int EC, int WPB, int BM, int MW, int KM, int OUT, int MEMF = 0>
void hex_case(const HexArgs &a)
{
    char label[96];
    char xg[16] = "";
    if (MEMF >= 16)
        std::snprintf(xg, sizeof xg, " xg%d%s", (MEMF >> 4) & 0xfff, (MEMF >> 16) & 1 ? " coop" : "");
    std::snprintf(label, sizeof label, "hex nq%d EC%d WPB%d %s MW%d K%d %s%s%s", NQ, EC, WPB,
                  BM == BASIS_LDS ? "lds " : (BM == BASIS_SMEM ? "smem" : (BM == BASIS_SMEM_COLS ? "sc8 " : "sc16")),
                  MW, KM, out_name(OUT),
                  (MEMF & 12) == 12 ? " al-io" : ((MEMF & 12) == 8 ? " al-o" : ((MEMF & 12) == 4 ? " al-i" : "")), xg);
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * a.nelmt * tune::ipow(NQ, 3),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ * NQ,
        [&]() { return launch_hex_wave<NQ, EC, WPB, BM, MW, KM, OUT, MEMF>(a, 0); });
}

template <int NQ, int EC, int WPB, int BM, int MW, int KM, int OUT, int MEMF = 0>
void quad_case(const QuadArgs &a)
{
    char label[96];
    std::snprintf(label, sizeof label, "quad nq%d EC%d WPB%d %s MW%d K%d %s mf%d", NQ, EC, WPB,
                  BM == BASIS_LDS ? "lds " : (BM == BASIS_SMEM ? "smem" : (BM == BASIS_SMEM_COLS ? "sc8 " : "sc16")), MW, KM,
                  out_name(OUT), MEMF);
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 2), sizeof(double) * a.nelmt * tune::ipow(NQ, 2),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm, a.nelmt * 8.0 * (nm * nm + (double)NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ, [&]() { return launch_quad_wave<NQ, EC, WPB, BM, MW, KM, OUT, MEMF>(a, 0); });
}

template <int NQ, int EC, int WPB, int MW, int KM, bool OL = false> void quad_mfma_case(const QuadArgs &a)
{
    char label[96];
    std::snprintf(label, sizeof label, "quad nq%d MFMA EC%d WPB%d MW%d K%d %s", NQ, EC, WPB, MW, KM,
                  OL ? "lds" : "st8");
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 2), sizeof(double) * a.nelmt * tune::ipow(NQ, 2),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm, a.nelmt * 8.0 * (nm * nm + (double)NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ, [&]() { return launch_quad_mfma<NQ, EC, WPB, MW, KM, OL>(a, 0); });
}

template <int NQ, int EC, int WPB, int MW, int KM, int XG = 0> void hex_mfma_case(const HexArgs &a)
{
    char label[96];
    std::snprintf(label, sizeof label, "hex nq%d MFMA EC%d WPB%d MW%d K%d xg%d", NQ, EC, WPB, MW, KM, XG);
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * a.nelmt * tune::ipow(NQ, 3),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ * NQ, [&]() { return launch_hex_mfma<NQ, EC, WPB, MW, KM, XG>(a, 0); });
}

template <int NQ, int MW, int XG> void hex_mfma2_case(const HexArgs &a)
{
    if constexpr (NQ >= 14 && NQ <= 16)
    {
        char label[96];
        std::snprintf(label, sizeof label, "hex nq%d MFMA two waves per element MW%d xg%d", NQ, MW, XG);
        const double nm = NQ - 1;
        if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * a.nelmt * tune::ipow(NQ, 3),
                        sizeof(double) * (NQ - 1) * NQ))
            return;
        run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
            a.nelmt * (size_t)NQ * NQ * NQ, [&]() { return launch_hex_mfma2<NQ, MW, XG>(a, 0); });
    }
}

template <int NQ, int WPB, int MW, int K, int XG, bool DIRECT = false, bool NTS = true, bool PEEL = true>
void hex_mfma4_case(const HexArgs &a)
{
    if constexpr (NQ >= 9 && NQ <= 16 && hex_mfma4_lds_bytes<NQ, WPB>() <= 160 * 1024)
    {
        char label[96];
        std::snprintf(label, sizeof label, "hex nq%d MFMA 4x4x4 WPB%d MW%d K%d xg%d%s%s", NQ, WPB, MW, K, XG,
                      DIRECT ? (NTS ? " direct" : " direct, cached stores") : "", PEEL ? "" : " nopeel");
        const double nm = NQ - 1;
        if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * a.nelmt * tune::ipow(NQ, 3),
                        sizeof(double) * (NQ - 1) * NQ))
            return;
        run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
            a.nelmt * (size_t)NQ * NQ * NQ, [&]() { return launch_hex_mfma4<NQ, WPB, MW, K, XG, DIRECT, NTS, PEEL>(a, 0); });
    }
}

template <int NQ, int MW, int XG> void hex_mfma4_pair_case(const HexArgs &a)
{
    if constexpr (NQ >= 9 && NQ <= 16)
    {
        char label[96];
        std::snprintf(label, sizeof label, "hex nq%d MFMA 4x4x4 two waves per element MW%d xg%d", NQ, MW, XG);
        const double nm = NQ - 1;
        if (!tune::fits(label, sizeof(double) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * a.nelmt * tune::ipow(NQ, 3),
                        sizeof(double) * (NQ - 1) * NQ))
            return;
        run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
            a.nelmt * (size_t)NQ * NQ * NQ, [&]() { return launch_hex_mfma4_pair<NQ, MW, XG>(a, 0); });
    }
}

// shader-clock time per phase of hex_mfma4_kernel's element loop (STAMP)
template <int NQ, int WPB, int MW, int K, int XG, bool DIRECT = false> void hex_mfma4_phases(const HexArgs &a)
{
    if constexpr (NQ >= 9 && NQ <= 16 && hex_mfma4_lds_bytes<NQ, WPB>() <= 160 * 1024)
    {
        auto kern            = hex_mfma4_kernel<NQ, WPB, MW, K, XG, true, DIRECT>;
        constexpr size_t lds = hex_mfma4_lds_bytes<NQ, WPB, DIRECT>();
        if (lds > 48 * 1024)
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int bpc = 0, cus = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, kWave * WPB, lds));
        CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
        const uint64_t per = (uint64_t)WPB * (K > 0 ? K : 1), need = (a.nelmt + per - 1) / per;
        uint64_t grid      = (uint64_t)bpc * cus;
        if (K != 0 || grid > need)
            grid = need;
        const size_t nslot = (size_t)grid * WPB * 8;
        unsigned long long *dev, host[8] = {};
        CK(hipMalloc((void **)&dev, nslot * sizeof(unsigned long long)));
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep)
        {
            CK(hipMemset(dev, 0, nslot * sizeof(unsigned long long)));
            CK(hipEventRecord(g_e0, 0));
            kern<<<(unsigned)grid, kWave * WPB, lds>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt, dev);
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
        const double n = (double)host[5];
        std::printf("phases%s hex nq%d 4x4x4 WPB%d MW%d K%d (%d wg/CU, stamped launch %.3f ms): clocks per element: stage+issue %.0f | "
                    "sweeps 1+2 %.0f | sweep 3 %.0f | image+flush %.0f | wait next %.0f | sum %.0f (%llu elements)\n",
                    DIRECT ? " [direct stores]" : "", NQ, WPB, MW, K, bpc, ms, host[0] / n, host[1] / n, host[2] / n, host[3] / n, host[4] / n,
                    (host[0] + host[1] + host[2] + host[3] + host[4]) / n, host[5]);
    }
}

int main(int argc, char **argv)
{
    const int nq       = TUNE_NQ;
    const size_t nelmt = argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)1 << 20;
    g_reps             = argc > 2 ? std::atoi(argv[2]) : 15;
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s, %d CUs, dim %d nq %d nelmt %zu reps %d\n", prop.gcnArchName,
                prop.multiProcessorCount, TUNE_DIM, nq, nelmt, g_reps);

    const size_t nm = nq - 1;
    size_t nin = nelmt, nout = nelmt;
    for (int d = 0; d < TUNE_DIM; ++d)
    {
        nin *= nm;
        nout *= nq;
    }
    double *b0, *b1, *b2, *in, *out;
    CK(hipMalloc((void **)&b0, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&b1, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&b2, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&in, sizeof(double) * nin));
    CK(hipMalloc((void **)&out, sizeof(double) * nout));
    tune::capacity() = {sizeof(double) * nin, sizeof(double) * nout, sizeof(double) * nm * nq};
    fill_basis(b0, nm, nq, 0);
    fill_basis(b1, nm, nq, 0);
    fill_basis(b2, nm, nq, 0);
    fill_random(in, nin, 0x5F3759DF, 0, 0);
    CK(hipMemset(out, 0, sizeof(double) * nout));
    CK(hipDeviceSynchronize());

#if TUNE_DIM == 3
    HexArgs a{b0, b1, b2, in, nullptr, out, nelmt};
#define H(NQ, EC, WPB, BM, MW, KM, OUT) hex_case<NQ, EC, WPB, BM, MW, KM, OUT>(a);
#define HM(NQ, EC, WPB, BM, MW, KM, OUT, MF) hex_case<NQ, EC, WPB, BM, MW, KM, OUT, MF>(a);
#define X(NQ, EC, WPB, MW, KM) hex_mfma_case<NQ, EC, WPB, MW, KM>(a);
#define XX(NQ, EC, WPB, MW, KM, XG) hex_mfma_case<NQ, EC, WPB, MW, KM, XG>(a);
    TUNE_CASES
    hex_mfma4_phases<TUNE_NQ, 1, 1, 1, 64>(a);
    hex_mfma4_phases<TUNE_NQ, 1, 2, 1, 64>(a);
    hex_mfma4_phases<TUNE_NQ, 1, 1, 0, 0>(a);
    hex_mfma4_phases<TUNE_NQ, 1, 1, 4, 64>(a);
    hex_mfma4_phases<TUNE_NQ, 1, 1, 1, 64, true>(a);
    hex_mfma4_phases<TUNE_NQ, 1, 2, 1, 64, true>(a);
    for (int rep = 0; rep < 2; ++rep)
    {
        hex_mfma4_pair_case<TUNE_NQ, 2, 64>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 1, 64, false, true, false>(a); // without the peeled k remainder (nq 14, 15 differ)
        hex_mfma4_case<TUNE_NQ, 1, 1, 1, 64, false, true, false>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 1, 64, true, false>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 1, 64, true, false>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 1, 64, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 1, 64, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 3, 1, 64, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 1, 0, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 0, 0, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 2, 64, true>(a);
        hex_mfma4_case<TUNE_NQ, 2, 2, 1, 64, true>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 0, 0>(a);
        hex_mfma4_case<TUNE_NQ, 1, 2, 0, 0>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 2, 64>(a);
        hex_mfma4_case<TUNE_NQ, 1, 1, 4, 64>(a);
        hex_mfma4_case<TUNE_NQ, 2, 1, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 2, 2, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 4, 1, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 4, 2, 1, 64>(a);
        hex_mfma4_case<TUNE_NQ, 4, 2, 0, 0>(a);
        hex_mfma2_case<TUNE_NQ, 1, 64>(a);
        hex_mfma2_case<TUNE_NQ, 2, 64>(a);
        hex_mfma2_case<TUNE_NQ, 3, 64>(a);
        hex_mfma2_case<TUNE_NQ, 2, 0>(a);
    }
#else
    QuadArgs a{b0, b1, in, nullptr, out, nelmt};
#define Q(NQ, EC, WPB, BM, MW, KM, OUT) quad_case<NQ, EC, WPB, BM, MW, KM, OUT>(a);
#define QM(NQ, EC, WPB, BM, MW, KM, OUT, MF) quad_case<NQ, EC, WPB, BM, MW, KM, OUT, MF>(a);
#define M(NQ, EC, WPB, MW, KM) quad_mfma_case<NQ, EC, WPB, MW, KM>(a); quad_mfma_case<NQ, EC, WPB, MW, KM, true>(a);
    TUNE_CASES
#endif
    return 0;
}
