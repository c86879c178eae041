// sf_tune_table.hip -- re-measure every row of csrc/wave_table.h on the current device, each with the
// XCD-grouping alternatives (MF bits 4+: runs of 16 / 64 neighbouring workgroups per XCD), interleaved A/B/A/B so
// that clock drift shows up as a difference between the two runs of the same variant.
// Usage: sf_tune_table [nelmt] [reps] [hex|quad|all|hexf32|quadf32|quadmfma]
#include "../csrc/sf_dispatch.h"
#include "../csrc/wave_launch.h"
#include "../csrc/wave_table.h"
#include "tune_guard.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

static int g_reps = 20;
static size_t g_nelmt;
static hipEvent_t g_e0, g_e1;
static double *g_b, *g_in, *g_out;

template <class F> static void run(const char *label, double dof, double bytes, size_t nout, F launch)
{
    int rc = launch();
    CK(hipDeviceSynchronize());
    if (rc != 0)
    {
        std::printf("%-34s rc=%d\n", label, rc);
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
    const double tmin = t[0], tmean = sum / t.size();
    double ss = 0;
    sumsq_blocking(g_out, nout, &ss, 0);
    std::printf("%-34s min %8.4f mean %8.4f ms | %7.2f / %7.2f GDOF/s | %7.1f GB/s (mean) | norm %.10g\n",
                label, tmin, tmean, dof / (tmin * 1e-3) * 1e-9, dof / (tmean * 1e-3) * 1e-9,
                bytes / (tmean * 1e-3) * 1e-9, std::sqrt(ss));
    std::fflush(stdout);
}

template <int NQ, int MF> static void hex_one()
{
    using C = HexCfg<NQ>;
    char label[96];
    std::snprintf(label, sizeof label, "hex  nq%-2d EC%-3d WPB%d MW%d K%d MF%d", NQ, C::EC, C::WPB, C::MW,
                  C::KM, MF);
    const double nm = NQ - 1;
    HexArgs a{g_b, g_b, g_b, g_in, nullptr, g_out, g_nelmt};
    if (!tune::fits(label, sizeof(double) * g_nelmt * tune::ipow(NQ - 1, 3), sizeof(double) * g_nelmt * tune::ipow(NQ, 3),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, g_nelmt * nm * nm * nm, g_nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ),
        g_nelmt * (size_t)NQ * NQ * NQ,
        [&]() { return launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, MF>(a, 0); });
}

template <int NQ, int MF> static void quad_one()
{
    using C = QuadCfg<NQ>;
    char label[96];
    std::snprintf(label, sizeof label, "quad nq%-2d EC%-3d WPB%d MW%d K%d MF%d", NQ, C::EC, C::WPB, C::MW,
                  C::KM, MF);
    const double nm = NQ - 1;
    QuadArgs a{g_b, g_b, g_in, nullptr, g_out, g_nelmt};
    if (!tune::fits(label, sizeof(double) * g_nelmt * tune::ipow(NQ - 1, 2), sizeof(double) * g_nelmt * tune::ipow(NQ, 2),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, g_nelmt * nm * nm, g_nelmt * 8.0 * (nm * nm + (double)NQ * NQ), g_nelmt * (size_t)NQ * NQ,
        [&]() { return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, MF>(a, 0); });
}

template <int NQ, bool OL, int MW = 1, int XG = 0> static void quad_mfma_one()
{
    char label[96];
    std::snprintf(label, sizeof label, "quad nq%-2d MFMA EC2 WPB4 MW%d %s xg%d", NQ, MW, OL ? "lds" : "st8", XG);
    const double nm = NQ - 1;
    QuadArgs a{g_b, g_b, g_in, nullptr, g_out, g_nelmt};
    if (!tune::fits(label, sizeof(double) * g_nelmt * tune::ipow(NQ - 1, 2), sizeof(double) * g_nelmt * tune::ipow(NQ, 2),
                    sizeof(double) * (NQ - 1) * NQ))
        return;
    run(label, g_nelmt * nm * nm, g_nelmt * 8.0 * (nm * nm + (double)NQ * NQ), g_nelmt * (size_t)NQ * NQ,
        [&]() { return launch_quad_mfma<NQ, 2, 4, MW, (NQ <= 16 ? 1 : 2), OL, XG>(a, 0); });
}

template <class F> static void run_f32(const char *label, double dof, double bytes, size_t nout, F launch)
{
    int rc = launch();
    CK(hipDeviceSynchronize());
    if (rc != 0)
    {
        std::printf("%-34s rc=%d\n", label, rc);
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
    const double tmin = t[0], tmean = sum / t.size();
    double ss = 0;
    sumsq_f32_blocking((const float *)g_out, nout, &ss, 0);
    std::printf("%-34s min %8.4f mean %8.4f ms | %7.2f / %7.2f GDOF/s | %7.1f GB/s (mean) | norm %.7g\n",
                label, tmin, tmean, dof / (tmin * 1e-3) * 1e-9, dof / (tmean * 1e-3) * 1e-9,
                bytes / (tmean * 1e-3) * 1e-9, std::sqrt(ss));
    std::fflush(stdout);
}

template <int NQ, int MF> static void hex_f32_one()
{
    using C = HexCfgF32<NQ>;
    char label[96];
    std::snprintf(label, sizeof label, "hex  f32 nq%-2d EC%-3d WPB%d MW%d MF%d", NQ, C::EC, C::WPB, C::MW, MF);
    const double nm = NQ - 1;
    const float *b  = (const float *)g_b;
    HexArgsT<float> a{b, b, b, (const float *)g_in, nullptr, (float *)g_out, g_nelmt};
    if (!tune::fits(label, sizeof(float) * g_nelmt * tune::ipow(NQ - 1, 3), sizeof(float) * g_nelmt * tune::ipow(NQ, 3),
                    sizeof(float) * (NQ - 1) * NQ))
        return;
    run_f32(label, g_nelmt * nm * nm * nm, g_nelmt * 4.0 * (nm * nm * nm + (double)NQ * NQ * NQ),
            g_nelmt * (size_t)NQ * NQ * NQ, [&]() {
                return launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, MF, float>(a, 0);
            });
}

template <int NQ, int MF> static void quad_f32_one()
{
    using C = QuadCfgF32<NQ>;
    char label[96];
    std::snprintf(label, sizeof label, "quad f32 nq%-2d EC%-3d WPB%d MW%d MF%d", NQ, C::EC, C::WPB, C::MW, MF);
    const double nm = NQ - 1;
    const float *b  = (const float *)g_b;
    QuadArgsT<float> a{b, b, (const float *)g_in, nullptr, (float *)g_out, g_nelmt};
    if (!tune::fits(label, sizeof(float) * g_nelmt * tune::ipow(NQ - 1, 2), sizeof(float) * g_nelmt * tune::ipow(NQ, 2),
                    sizeof(float) * (NQ - 1) * NQ))
        return;
    run_f32(label, g_nelmt * nm * nm, g_nelmt * 4.0 * (nm * nm + (double)NQ * NQ), g_nelmt * (size_t)NQ * NQ,
            [&]() {
                return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, MF, float>(a, 0);
            });
}

template <int NQ> static void hex_f32_row()
{
    fill_basis_f32((float *)g_b, NQ - 1, NQ, 0);
    CK(hipDeviceSynchronize());
    constexpr int B = HexCfgF32<NQ>::MF & 15;
    hex_f32_one<NQ, B>();
    hex_f32_one<NQ, B | (16 << 4)>();
    hex_f32_one<NQ, B | (64 << 4)>();
    hex_f32_one<NQ, B>();
    hex_f32_one<NQ, B | (16 << 4)>();
    hex_f32_one<NQ, B | (64 << 4)>();
}

template <int NQ> static void quad_f32_row()
{
    fill_basis_f32((float *)g_b, NQ - 1, NQ, 0);
    CK(hipDeviceSynchronize());
    constexpr int B = QuadCfgF32<NQ>::MF & 15;
    quad_f32_one<NQ, B>();
    quad_f32_one<NQ, B | (16 << 4)>();
    quad_f32_one<NQ, B | (64 << 4)>();
    quad_f32_one<NQ, B>();
    quad_f32_one<NQ, B | (16 << 4)>();
    quad_f32_one<NQ, B | (64 << 4)>();
}

template <int NQ> static void quad_mfma_row()
{
    fill_basis(g_b, NQ - 1, NQ, 0);
    CK(hipDeviceSynchronize());
    constexpr int MW = NQ <= 16 ? 1 : 2;
    quad_mfma_one<NQ, false, MW, 0>();
    quad_mfma_one<NQ, false, MW, 64>();
    quad_mfma_one<NQ, true, MW, 0>();
    quad_mfma_one<NQ, true, MW, 64>();
    quad_mfma_one<NQ, false, MW, 0>();
    quad_mfma_one<NQ, false, MW, 64>();
    quad_mfma_one<NQ, true, MW, 0>();
    quad_mfma_one<NQ, true, MW, 64>();
}

template <int NQ> static void hex_row()
{
    fill_basis(g_b, NQ - 1, NQ, 0);
    CK(hipDeviceSynchronize());
    constexpr int M = HexCfg<NQ>::MF;
    constexpr int B = M & 15; // alignment bits of the row; XG (bits 4+) swept below
    hex_one<NQ, B>();
    hex_one<NQ, B | (16 << 4)>();
    hex_one<NQ, B | (64 << 4)>();
    hex_one<NQ, B>();
    hex_one<NQ, B | (16 << 4)>();
    hex_one<NQ, B | (64 << 4)>();
}

template <int NQ> static void quad_row()
{
    fill_basis(g_b, NQ - 1, NQ, 0);
    CK(hipDeviceSynchronize());
    constexpr int M = QuadCfg<NQ>::MF;
    constexpr int B = M & 15; // alignment bits of the row; XG (bits 4+) swept below
    quad_one<NQ, B>();
    quad_one<NQ, B | (16 << 4)>();
    quad_one<NQ, B | (64 << 4)>();
    quad_one<NQ, B>();
    quad_one<NQ, B | (16 << 4)>();
    quad_one<NQ, B | (64 << 4)>();
    if constexpr (NQ >= 11)
    {
        quad_mfma_one<NQ, false>();
        quad_mfma_one<NQ, true>();
        quad_mfma_one<NQ, false>();
        quad_mfma_one<NQ, true>();
    }
}

int main(int argc, char **argv)
{
    g_nelmt           = argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)1 << 20;
    g_reps            = argc > 2 ? std::atoi(argv[2]) : 20;
    const char *which = argc > 3 ? argv[3] : "all";
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s, %d CUs, nelmt %zu reps %d\n", prop.gcnArchName, prop.multiProcessorCount,
                g_nelmt, g_reps);
    // buffers sized for the largest row: hex nq = 11 (1000 in / 1331 out per element; quad nq = 32 needs 961 / 1024)
    const size_t nin = g_nelmt * 1000, nout = g_nelmt * 1331;
    tune::capacity() = {sizeof(double) * nin, sizeof(double) * nout, sizeof(double) * 32 * 32};
    CK(hipMalloc((void **)&g_b, sizeof(double) * 32 * 32));
    CK(hipMalloc((void **)&g_in, sizeof(double) * nin));
    CK(hipMalloc((void **)&g_out, sizeof(double) * nout));
    fill_random(g_in, nin, 0x5F3759DF, 0, 0);
    CK(hipMemset(g_out, 0, sizeof(double) * nout));
    CK(hipDeviceSynchronize());

    if (!std::strcmp(which, "hex") || !std::strcmp(which, "all"))
    {
        hex_row<2>(); hex_row<3>(); hex_row<4>(); hex_row<5>(); hex_row<6>();
        hex_row<7>(); hex_row<8>(); hex_row<9>(); hex_row<10>();
        if (g_nelmt <= (1u << 19))
            hex_row<11>();
    }
    if (!std::strcmp(which, "quad") || !std::strcmp(which, "all"))
    {
        quad_row<2>(); quad_row<3>(); quad_row<4>(); quad_row<5>(); quad_row<6>(); quad_row<7>();
        quad_row<8>(); quad_row<9>(); quad_row<10>(); quad_row<11>(); quad_row<12>(); quad_row<13>();
        quad_row<14>(); quad_row<15>(); quad_row<16>(); quad_row<17>(); quad_row<18>(); quad_row<19>(); quad_row<20>(); quad_row<21>();
        quad_row<22>(); quad_row<23>(); quad_row<24>(); quad_row<32>();
    }
    if (!std::strcmp(which, "quadmfma"))
    {
        quad_mfma_row<11>(); quad_mfma_row<12>(); quad_mfma_row<13>(); quad_mfma_row<14>();
        quad_mfma_row<15>(); quad_mfma_row<16>(); quad_mfma_row<17>(); quad_mfma_row<18>();
        quad_mfma_row<19>(); quad_mfma_row<20>(); quad_mfma_row<21>(); quad_mfma_row<22>();
        quad_mfma_row<23>(); quad_mfma_row<24>(); quad_mfma_row<25>(); quad_mfma_row<26>();
        quad_mfma_row<27>(); quad_mfma_row<28>(); quad_mfma_row<29>(); quad_mfma_row<30>();
        quad_mfma_row<31>(); quad_mfma_row<32>();
    }
    if (!std::strcmp(which, "hexf32") || !std::strcmp(which, "quadf32"))
    {
        fill_random_f32((float *)g_in, nin, 0x5F3759DF, 0, 0);
        CK(hipDeviceSynchronize());
    }
    if (!std::strcmp(which, "hexf32"))
    {
        hex_f32_row<2>(); hex_f32_row<3>(); hex_f32_row<4>(); hex_f32_row<5>(); hex_f32_row<6>();
        hex_f32_row<7>(); hex_f32_row<8>(); hex_f32_row<9>(); hex_f32_row<10>();
    }
    if (!std::strcmp(which, "quadf32"))
    {
        quad_f32_row<2>(); quad_f32_row<3>(); quad_f32_row<4>(); quad_f32_row<5>(); quad_f32_row<6>();
        quad_f32_row<7>(); quad_f32_row<8>(); quad_f32_row<9>(); quad_f32_row<10>(); quad_f32_row<11>();
        quad_f32_row<12>(); quad_f32_row<13>(); quad_f32_row<14>(); quad_f32_row<15>(); quad_f32_row<16>();
    }
    return 0;
}
