// sf_membench11.hip -- launch shapes of the nq = 2 stream kernels (3D: 8 B in / 64 B out per element, 2D: 8 / 32): U output
// pairs per thread, with / without XCD runs, under bench.py's sweep protocol (groups of 8 launches replayed from a HIP graph,
// min over 40 groups).  The arithmetic is the product kernels' (bwdtrans_hex.hip, bwdtrans_quad.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../csrc/sf_common.h"

using sf::double2_t;
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

template <int DIM, int U, int XG, int THREADS>
__global__ __launch_bounds__(THREADS) void nq2_stream(const double *__restrict__ b0, const double *__restrict__ b1,
                                                      const double *__restrict__ b2, const double *__restrict__ in,
                                                      double *__restrict__ out, uint64_t nelmt)
{
    constexpr int SH  = DIM == 3 ? 2 : 1; // pairs per element: 4 (3D) / 2 (2D)
    const uint64_t nv = nelmt << SH;
    const double c0 = b0[0], c1 = b0[1];
    double2_t *out2     = reinterpret_cast<double2_t *>(out);
    const uint64_t base = sf::logical_block<XG>() * ((uint64_t)THREADS * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        const uint64_t v = base + (uint64_t)u * THREADS;
        if (v < nv)
        {
            const int p     = (int)(v & ((1 << SH) - 1));
            const double x  = in[v >> SH];
            const double bj = b1[p & 1], bk = DIM == 3 ? b2[p >> 1] : 1.0;
            double2_t r     = {(x * c0) * bj, (x * c1) * bj};
            if (DIM == 3)
                r = double2_t{r.x * bk, r.y * bk};
            __builtin_nontemporal_store(r, out2 + v);
        }
    }
}

// the same U pairs per thread, but every WAVE writes one contiguous U-KB piece (pair index = (wave * U + u) * 64 + lane)
template <int DIM, int U, int XG, int THREADS>
__global__ __launch_bounds__(THREADS) void nq2_stream_wavepiece(const double *__restrict__ b0, const double *__restrict__ b1,
                                                                const double *__restrict__ b2, const double *__restrict__ in,
                                                                double *__restrict__ out, uint64_t nelmt)
{
    constexpr int SH  = DIM == 3 ? 2 : 1;
    const uint64_t nv = nelmt << SH;
    const double c0 = b0[0], c1 = b0[1];
    double2_t *out2     = reinterpret_cast<double2_t *>(out);
    const uint64_t wave = sf::logical_block<XG>() * (THREADS / 64) + (threadIdx.x >> 6);
    const uint64_t base = wave * (64ull * U) + (threadIdx.x & 63);
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        const uint64_t v = base + (uint64_t)u * 64;
        if (v < nv)
        {
            const int p     = (int)(v & ((1 << SH) - 1));
            const double x  = in[v >> SH];
            const double bj = b1[p & 1], bk = DIM == 3 ? b2[p >> 1] : 1.0;
            double2_t r     = {(x * c0) * bj, (x * c1) * bj};
            if (DIM == 3)
                r = double2_t{r.x * bk, r.y * bk};
            __builtin_nontemporal_store(r, out2 + v);
        }
    }
}

__global__ void fill_pattern(double *p, uint64_t n, double scale)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = scale * (1.0 + 1e-9 * (double)(i * 2654435761ull % 1000003ull));
}

static double *g_b, *g_in, *g_out;
static uint64_t kElmt = 1 << 20;

template <int DIM, int U, int XG, int THREADS = 256> static void go()
{
    const uint64_t nv     = kElmt << (DIM == 3 ? 2 : 1);
    const unsigned blocks = (unsigned)((nv + (uint64_t)THREADS * U - 1) / ((uint64_t)THREADS * U));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 8; ++k)
        nq2_stream<DIM, U, XG, THREADS><<<blocks, THREADS, 0, s>>>(g_b, g_b, g_b, g_in, g_out, kElmt);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 40; ++r)
    {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 8);
    }
    const double bytes = 8.0 * kElmt * (1 + (DIM == 3 ? 8 : 4));
    std::printf("%dD nq 2  %2d pairs/thread  %3d threads  xg %2d  grid %6u   %7.2f us   %7.1f GB/s = %.3f of 8 TB/s\n", DIM, U, THREADS,
                XG, blocks, best * 1e3, bytes / best * 1e-6, bytes / best * 1e-6 / 8000.0);
    std::fflush(stdout);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
}

template <int DIM, int U, int XG, int THREADS = 256> static void go_wp()
{
    const uint64_t nv     = kElmt << (DIM == 3 ? 2 : 1);
    const unsigned blocks = (unsigned)((nv + (uint64_t)THREADS * U - 1) / ((uint64_t)THREADS * U));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 8; ++k)
        nq2_stream_wavepiece<DIM, U, XG, THREADS><<<blocks, THREADS, 0, s>>>(g_b, g_b, g_b, g_in, g_out, kElmt);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 40; ++r)
    {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 8);
    }
    const double bytes = 8.0 * kElmt * (1 + (DIM == 3 ? 8 : 4));
    std::printf("%dD nq 2  %2d pairs/thread, a contiguous %2d KB piece per wave  xg %2d  grid %6u   %7.2f us   %7.1f GB/s = %.3f of 8 TB/s\n", DIM, U, U,
                XG, blocks, best * 1e3, bytes / best * 1e-6, bytes / best * 1e-6 / 8000.0);
    std::fflush(stdout);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
}

template <int DIM> static void rows()
{
    for (int rep = 0; rep < 2; ++rep)
    {
        go<DIM, 4, 64>(); // shipped (3D); 2D ships xg 0
        go<DIM, 4, 0>();
        go<DIM, 1, 0>();
        go<DIM, 1, 64>();
        go<DIM, 2, 0>();
        go<DIM, 2, 64>();
        go<DIM, 8, 0>();
        go<DIM, 8, 64>();
        go<DIM, 16, 0>();
        go<DIM, 16, 8>();
        go<DIM, 2, 64, 512>();
        go<DIM, 4, 64, 512>();
        go<DIM, 8, 64, 128>();
        go<DIM, 4, 16, 1024>();
        go_wp<DIM, 2, 64>();
        go_wp<DIM, 4, 64>();
        go_wp<DIM, 8, 64>();
    }
}

int main(int argc, char **argv)
{
    if (argc > 1)
        kElmt = std::strtoull(argv[1], nullptr, 10);
    CK(hipMalloc((void **)&g_b, 64));
    CK(hipMalloc((void **)&g_in, 8 * kElmt));
    CK(hipMalloc((void **)&g_out, 64 * kElmt));
    CK(hipMemset(g_b, 0, 64));
    CK(hipMemset(g_in, 0, 8 * kElmt));
    CK(hipMemset(g_out, 0, 64 * kElmt));
    if (argc > 2) // non-zero data: every output value differs
    {
        fill_pattern<<<4096, 256>>>(g_in, kElmt, 1.0);
        fill_pattern<<<1, 64>>>(g_b, 8, 0.5);
        CK(hipDeviceSynchronize());
        std::printf("inputs: non-zero pattern\n");
    }
    else
        std::printf("inputs: all zero (every store writes zeros)\n");
    rows<3>();
    rows<2>();
    return 0;
}
