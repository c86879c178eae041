// sf_tune_f32.hip -- configuration sweep of the T = float instantiations (development tool).
#include "../csrc/sf_dispatch.h"
#include "../csrc/wave_launch.h"
#include "tune_guard.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

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

static void run(const char *label, double dof, double bytes, float *out, size_t nout,
                const std::function<int()> &launch)
{
    int rc = launch();
    CK(hipDeviceSynchronize());
    if (rc != 0)
    {
        std::printf("%-40s rc=%d\n", label, rc);
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
    double ss = 0;
    sumsq_f32_blocking(out, nout, &ss, 0);
    std::printf("%-40s min %8.4f mean %8.4f ms | %7.2f / %7.2f GDOF/s | %7.1f GB/s | norm %.8g\n", label,
                t[0], sum / t.size(), dof / (t[0] * 1e-3) * 1e-9, dof / (sum / t.size() * 1e-3) * 1e-9,
                bytes / (t[0] * 1e-3) * 1e-9, std::sqrt(ss));
    std::fflush(stdout);
}

template <int NQ, int EC, int WPB, int BM, int MW, int KM, int OUT = OUT_LDS, int MEMF = 0>
void hex_case(const HexArgsT<float> &a)
{
    char label[96];
    std::snprintf(label, sizeof label, "hex f32 nq%d EC%d WPB%d %s MW%d K%d o%d mf%d", NQ, EC, WPB,
                  BM == BASIS_LDS ? "lds " : "smem", MW, KM, OUT, MEMF);
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(float) * a.nelmt * tune::ipow(NQ - 1, 3), sizeof(float) * a.nelmt * tune::ipow(NQ, 3),
                    sizeof(float) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm * nm, a.nelmt * 4.0 * (nm * nm * nm + (double)NQ * NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ * NQ,
        [&]() { return launch_hex_wave<NQ, EC, WPB, BM, MW, KM, OUT, MEMF, float>(a, 0); });
}

template <int NQ, int EC, int WPB, int BM, int MW, int KM, int OUT = OUT_LDS, int MEMF = 0>
void quad_case(const QuadArgsT<float> &a)
{
    char label[96];
    std::snprintf(label, sizeof label, "quad f32 nq%d EC%d WPB%d bm%d MW%d K%d o%d mf%d", NQ, EC, WPB, BM, MW, KM, OUT,
                  MEMF);
    const double nm = NQ - 1;
    if (!tune::fits(label, sizeof(float) * a.nelmt * tune::ipow(NQ - 1, 2), sizeof(float) * a.nelmt * tune::ipow(NQ, 2),
                    sizeof(float) * (NQ - 1) * NQ))
        return;
    run(label, a.nelmt * nm * nm, a.nelmt * 4.0 * (nm * nm + (double)NQ * NQ), a.out,
        a.nelmt * (size_t)NQ * NQ,
        [&]() { return launch_quad_wave<NQ, EC, WPB, BM, MW, KM, OUT, MEMF, float>(a, 0); });
}

int main(int argc, char **argv)
{
    const size_t nelmt = argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)1 << 20;
    g_reps             = argc > 2 ? std::atoi(argv[2]) : 15;
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    constexpr int NQ = 8, NM = 7;
    // buffers hold the LARGEST case below: hex nq 8 (343 in / 512 out per element) and quad nq 24 (529 / 576)
    constexpr size_t kMaxIn = 529, kMaxOut = 576, kMaxBasis = 23 * 24;
    static_assert(kMaxIn >= NM * NM * NM && kMaxOut >= NQ * NQ * NQ, "hex nq 8 must fit");
    float *b, *in, *out;
    CK(hipMalloc((void **)&b, sizeof(float) * NM * NQ));
    CK(hipMalloc((void **)&in, sizeof(float) * nelmt * kMaxIn));
    CK(hipMalloc((void **)&out, sizeof(float) * nelmt * kMaxOut));
    tune::capacity() = {sizeof(float) * nelmt * kMaxIn, sizeof(float) * nelmt * kMaxOut, sizeof(float) * kMaxBasis};
    fill_basis_f32(b, NM, NQ, 0);
    fill_random_f32(in, nelmt * kMaxIn, 0x5F3759DF, 0, 0);
    CK(hipDeviceSynchronize());
    HexArgsT<float> h{b, b, b, in, nullptr, out, nelmt};
    for (int rep = 0; rep < 2; ++rep)
    {
        hex_case<8, 2, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h); // table row (fp64 row x 2 elements), XCD runs of 64
        hex_case<8, 1, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h); // one element per wave: 1372-byte inputs, word grid
        hex_case<8, 1, 8, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
        hex_case<8, 1, 8, BASIS_SMEM, 8, 1, OUT_LDS, 1024>(h);
        hex_case<8, 2, 8, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
        hex_case<8, 3, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
        hex_case<8, 3, 8, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
        hex_case<8, 4, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
        hex_case<8, 8, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(h);
    }
    QuadArgsT<float> q{b, b, in, nullptr, out, nelmt};
    quad_case<8, 16, 4, BASIS_SMEM, 4, 1>(q);
    // nq 15 / 16: the derived rows (16 elements per chunk) leave 8 waves per CU; smaller chunks
    float *b2;
    CK(hipMalloc((void **)&b2, sizeof(float) * kMaxBasis));
    for (int rep = 0; rep < 2; ++rep)
    {
#define ROW(NQ, MF)                                                                                \
    {                                                                                              \
        static_assert((NQ - 1) * NQ <= kMaxBasis, "basis buffer");                                 \
        fill_basis_f32(b2, NQ - 1, NQ, 0);                                                         \
        QuadArgsT<float> qq{b2, b2, in, nullptr, out, nelmt};                                      \
        quad_case<NQ, 16, 4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, MF>(qq);                            \
        quad_case<NQ, 8, 4, BASIS_SMEM_COLS16, 4, 1, OUT_LDS, MF>(qq);                             \
        quad_case<NQ, 4, 4, BASIS_SMEM_COLS16, 4, 1, OUT_LDS, MF>(qq);                             \
        quad_case<NQ, 4, 8, BASIS_SMEM_COLS16, 4, 1, OUT_LDS, MF>(qq);                             \
        quad_case<NQ, 2, 8, BASIS_SMEM_COLS16, 4, 1, OUT_LDS, MF>(qq);                             \
    }
        ROW(12, 1024) ROW(13, 1036) ROW(14, 1024) ROW(15, 1036) ROW(16, 1024) ROW(20, 1024) ROW(24, 1024)
#undef ROW
    }
    // nq 4 / 6 / 8 / 10: the fp64 rows moved to small chunks under XCD runs (round 2); the same question for T = float
    for (int rep = 0; rep < 2; ++rep)
    {
#define LOW(NQ, EC0, MW0)                                                                          \
    {                                                                                              \
        fill_basis_f32(b2, NQ - 1, NQ, 0);                                                         \
        QuadArgsT<float> qq{b2, b2, in, nullptr, out, nelmt};                                      \
        quad_case<NQ, EC0, 4, BASIS_SMEM, MW0, 1, OUT_LDS, 0>(qq); /* the pinned row */            \
        quad_case<NQ, EC0, 4, BASIS_SMEM, MW0, 1, OUT_LDS, 1024>(qq);                              \
        quad_case<NQ, 16, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(qq);                                 \
        quad_case<NQ, 8, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(qq);                                  \
        quad_case<NQ, 8, 8, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(qq);                                  \
        quad_case<NQ, 4, 4, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(qq);                                  \
        quad_case<NQ, 4, 8, BASIS_SMEM, 4, 1, OUT_LDS, 1024>(qq);                                  \
    }
        LOW(4, 32, 4) LOW(6, 20, 4) LOW(8, 16, 4) LOW(10, 24, 2)
#undef LOW
#define ODD(NQ, EC0, BM, MW0, MF0)                                                                 \
    {                                                                                              \
        fill_basis_f32(b2, NQ - 1, NQ, 0);                                                         \
        QuadArgsT<float> qq{b2, b2, in, nullptr, out, nelmt};                                      \
        quad_case<NQ, EC0, 4, BM, MW0, 1, OUT_LDS, MF0>(qq); /* the pinned row */                  \
        quad_case<NQ, EC0, 4, BM, MW0, 1, OUT_LDS, 1024 + 8>(qq);                                  \
        quad_case<NQ, 24, 4, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                     \
        quad_case<NQ, 16, 4, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                     \
        quad_case<NQ, 12, 4, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                     \
        quad_case<NQ, 8, 4, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                      \
        quad_case<NQ, 8, 4, BM, 4, 1, OUT_LDS, 8>(qq);                                             \
        quad_case<NQ, 8, 8, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                      \
        quad_case<NQ, 4, 4, BM, 4, 1, OUT_LDS, 1024 + 8>(qq);                                      \
    }
        ODD(5, 48, BASIS_SMEM, 4, 0) ODD(7, 36, BASIS_SMEM, 4, 8) ODD(9, 28, BASIS_SMEM, 4, 8)
        ODD(11, 20, BASIS_SMEM_COLS, 2, 8)
#undef ODD
    }
    return 0;
}
