// sf_tune.hip -- variant sweep of the wave kernels on the current device (development tool).
// Usage: sf_tune [hex|quad|all] [nq] [nelmt] [reps]
// For every instantiated (NQ, EC, WPB, BMODE, MINW) it prints kernel time (hipEvent, min and mean
// over reps), GDOF/s, algorithmic GB/s (8*(nm^d+nq^d) B/element) and sqrt(sum out^2) as a sanity value.
#include "../csrc/wave_launch.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace sf
{
int sumsq_blocking(const double *x, size_t n, double *result_host, hipStream_t s);
int fill_random(double *x, size_t n, uint64_t seed, uint64_t first, hipStream_t s);
int fill_basis(double *b, size_t nm, size_t nq, hipStream_t s);
int stream_copy(const double *src, double *dst, size_t n, hipStream_t s);
} // namespace sf
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

struct Bench
{
    int reps;
    hipEvent_t e0, e1;
    void run(const char *label, double dof, double bytes, double *out, size_t nout,
             const std::function<int()> &launch)
    {
        int rc = launch(); // warm-up (+ occupancy query)
        CK(hipDeviceSynchronize());
        if (rc != 0)
        {
            std::printf("%-34s rc=%d\n", label, rc);
            return;
        }
        double tmin = 1e30, tsum = 0;
        for (int r = 0; r < reps; ++r)
        {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tmin = std::min(tmin, (double)ms);
            tsum += ms;
        }
        double ss = 0;
        sumsq_blocking(out, nout, &ss, 0);
        std::printf("%-34s min %8.4f ms  mean %8.4f ms  %8.2f GDOF/s  %8.1f GB/s  norm %.10g\n",
                    label, tmin, tsum / reps, dof / (tmin * 1e-3) * 1e-9,
                    bytes / (tmin * 1e-3) * 1e-9, std::sqrt(ss));
        std::fflush(stdout);
    }
};

template <int NQ, int EC, int WPB, int BM, int MW, int KM = 0, bool S16 = false, int MF = 0>
void hex_case(Bench &b, const HexArgs &a, int grid = 0)
{
    char label[96];
    std::snprintf(label, sizeof label, "hex nq%d EC%d WPB%d %s MW%d K%d %s mf%d g%d", NQ, EC, WPB,
                  BM == BASIS_LDS ? "lds " : "smem", MW, KM, S16 ? "st16" : "st8 ", MF, grid);
    const double nm = NQ - 1;
    b.run(label, a.nelmt * nm * nm * nm, a.nelmt * 8.0 * (nm * nm * nm + (double)NQ * NQ * NQ),
          a.out, a.nelmt * (size_t)NQ * NQ * NQ,
          [&]() { return launch_hex_wave<NQ, EC, WPB, BM, MW, KM, S16, MF>(a, 0, grid); });
}

template <int NQ, int EC, int WPB, int BM, int MW, int KM = 0, bool S16 = false>
void quad_case(Bench &b, const QuadArgs &a, int grid = 0)
{
    char label[96];
    std::snprintf(label, sizeof label, "quad nq%d EC%d WPB%d %s MW%d K%d %s g%d", NQ, EC, WPB,
                  BM == BASIS_LDS ? "lds " : "smem", MW, KM, S16 ? "st16" : "st8 ", grid);
    const double nm = NQ - 1;
    b.run(label, a.nelmt * nm * nm, a.nelmt * 8.0 * (nm * nm + (double)NQ * NQ), a.out,
          a.nelmt * (size_t)NQ * NQ, [&]() { return launch_quad_wave<NQ, EC, WPB, BM, MW, KM, S16>(a, 0, grid); });
}

int main(int argc, char **argv)
{
    const std::string what = argc > 1 ? argv[1] : "all";
    const int nq           = argc > 2 ? std::atoi(argv[2]) : 8;
    const size_t nelmt     = argc > 3 ? (size_t)std::atoll(argv[3]) : (size_t)1 << 20;
    Bench b;
    b.reps = argc > 4 ? std::atoi(argv[4]) : 10;
    CK(hipEventCreate(&b.e0));
    CK(hipEventCreate(&b.e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s %s, %d CUs, nelmt %zu, reps %d\n", prop.name, prop.gcnArchName,
                prop.multiProcessorCount, nelmt, b.reps);

    const size_t nm = nq - 1;
    double *b0, *b1, *b2, *in, *out;
    const size_t nin3 = nelmt * nm * nm * nm, nout3 = nelmt * (size_t)nq * nq * nq;
    CK(hipMalloc((void **)&b0, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&b1, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&b2, sizeof(double) * nm * nq));
    CK(hipMalloc((void **)&in, sizeof(double) * nin3));
    CK(hipMalloc((void **)&out, sizeof(double) * nout3));
    fill_basis(b0, nm, nq, 0);
    fill_basis(b1, nm, nq, 0);
    fill_basis(b2, nm, nq, 0);
    fill_random(in, nin3, 0x5F3759DF, 0, 0);
    CK(hipMemset(out, 0, sizeof(double) * nout3));
    CK(hipDeviceSynchronize());

    // HBM calibration: copy of (in+out)/2 doubles each way
    {
        const size_t n = (nout3 / 2) & ~(size_t)1;
        b.run("stream copy (16 B lanes, nt)", 0, 16.0 * n, out + n, n,
              [&]() { return stream_copy(out, out + n, n, 0); });
        CK(hipMemset(out, 0, sizeof(double) * nout3));
    }

    if (what == "hex" || what == "all")
    {
        HexArgs a{b0, b1, b2, in, nullptr, out, nelmt};
        if (nq == 8)
        {
            hex_case<8, 2, 4, BASIS_SMEM, 4, 2, true>(b, a);
            hex_case<8, 2, 4, BASIS_SMEM, 4, 3, true>(b, a);
            hex_case<8, 2, 8, BASIS_SMEM, 4, 2, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 1, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 2, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 3, true>(b, a);
            hex_case<8, 4, 8, BASIS_SMEM, 2, 2, true>(b, a);
            hex_case<8, 4, 2, BASIS_SMEM, 2, 2, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 3, 2, true>(b, a);
            hex_case<8, 4, 4, BASIS_LDS, 2, 2, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 2, true, 1>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 2, true, 2>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 2, true, 3>(b, a);
            hex_case<8, 6, 4, BASIS_SMEM, 1, 1, true>(b, a);
            hex_case<8, 6, 4, BASIS_SMEM, 1, 2, true>(b, a);
            hex_case<8, 8, 4, BASIS_SMEM, 1, 1, true>(b, a);
            hex_case<8, 4, 4, BASIS_SMEM, 2, 2, true>(b, a);
            hex_case<8, 2, 4, BASIS_SMEM, 4, 2, true>(b, a);
        }
        else if (nq == 2) { hex_case<2, 64, 4, BASIS_LDS, 2>(b, a); hex_case<2, 64, 4, BASIS_SMEM, 2>(b, a); hex_case<2, 128, 4, BASIS_SMEM, 2>(b, a); }
        else if (nq == 3) { hex_case<3, 14, 4, BASIS_LDS, 2>(b, a); hex_case<3, 14, 4, BASIS_SMEM, 2>(b, a); hex_case<3, 28, 4, BASIS_SMEM, 2>(b, a); }
        else if (nq == 4) { hex_case<4, 8, 4, BASIS_LDS, 2>(b, a); hex_case<4, 8, 4, BASIS_SMEM, 2>(b, a); hex_case<4, 16, 4, BASIS_SMEM, 2>(b, a); }
        else if (nq == 5) { hex_case<5, 5, 4, BASIS_LDS, 2>(b, a); hex_case<5, 5, 4, BASIS_SMEM, 2>(b, a); hex_case<5, 10, 4, BASIS_SMEM, 2>(b, a); }
        else if (nq == 6) { hex_case<6, 6, 4, BASIS_LDS, 2>(b, a); hex_case<6, 6, 4, BASIS_SMEM, 2>(b, a); hex_case<6, 2, 4, BASIS_SMEM, 4>(b, a); hex_case<6, 4, 4, BASIS_SMEM, 3>(b, a); }
        else if (nq == 7) { hex_case<7, 5, 4, BASIS_LDS, 2>(b, a); hex_case<7, 5, 4, BASIS_SMEM, 2>(b, a); hex_case<7, 4, 4, BASIS_SMEM, 2>(b, a); hex_case<7, 2, 4, BASIS_SMEM, 3>(b, a); }
        else if (nq == 9) { hex_case<9, 1, 4, BASIS_LDS, 3>(b, a); hex_case<9, 1, 4, BASIS_SMEM, 3>(b, a); hex_case<9, 2, 4, BASIS_SMEM, 2>(b, a); }
        else if (nq == 10) { hex_case<10, 1, 4, BASIS_LDS, 3>(b, a); hex_case<10, 1, 4, BASIS_SMEM, 3>(b, a); hex_case<10, 2, 4, BASIS_SMEM, 2>(b, a); }
    }
    if (what == "quad" || what == "all")
    {
        QuadArgs a{b0, b1, in, nullptr, out, nelmt};
        if (nq == 8)
        {
            quad_case<8, 16, 4, BASIS_SMEM, 2>(b, a);
            quad_case<8, 16, 4, BASIS_SMEM, 2, 0, true>(b, a);
            quad_case<8, 16, 4, BASIS_SMEM, 2, 4, true>(b, a);
            quad_case<8, 32, 4, BASIS_SMEM, 2, 4, true>(b, a);
        }
    }
    return 0;
}
