// sf_membench9.hip -- read-only reduction shapes for benchmark01's device column (sum of squares over n doubles):
// persistent grid-stride grids of several sizes and unroll depths against the grid-covers-the-array shape the library
// uses (aux_kernels.hip, sumsq_partial_kernel).  Per-thread partials are written, not reduced: the timing is the stream.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>

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

// persistent: thread t of the grid takes vectors t, t + T, ... U of them in flight
template <int U> __global__ __launch_bounds__(256) void stride_sum(const d2 *__restrict__ x, uint64_t nv, double *__restrict__ part)
{
    const uint64_t T = (uint64_t)gridDim.x * 256;
    uint64_t v       = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (; v + (U - 1) * T < nv; v += U * T)
    {
        d2 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            p[u] = __builtin_nontemporal_load(x + v + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (u & 1)
            {
                a2 = __builtin_fma(p[u].x, p[u].x, a2);
                a3 = __builtin_fma(p[u].y, p[u].y, a3);
            }
            else
            {
                a0 = __builtin_fma(p[u].x, p[u].x, a0);
                a1 = __builtin_fma(p[u].y, p[u].y, a1);
            }
        }
    }
    for (; v < nv; v += T)
    {
        const d2 q = x[v];
        a0         = __builtin_fma(q.x, q.x, a0);
        a1         = __builtin_fma(q.y, q.y, a1);
    }
    part[(uint64_t)blockIdx.x * 256 + threadIdx.x] = (a0 + a1) + (a2 + a3);
}

// the library's shape: workgroup b sums `tpb` consecutive tiles of 256*U vectors
template <int U> __global__ __launch_bounds__(256) void run_sum(const d2 *__restrict__ x, uint64_t nv, uint32_t tpb, double *__restrict__ part)
{
    constexpr uint64_t tile = 256ull * U;
    uint64_t v              = (uint64_t)blockIdx.x * tpb * tile + threadIdx.x;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (uint32_t t = 0; t < tpb; ++t, v += tile)
    {
        if (v + (U - 1) * 256 < nv)
        {
            d2 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                p[u] = __builtin_nontemporal_load(x + v + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                if (u & 1)
                {
                    a2 = __builtin_fma(p[u].x, p[u].x, a2);
                    a3 = __builtin_fma(p[u].y, p[u].y, a3);
                }
                else
                {
                    a0 = __builtin_fma(p[u].x, p[u].x, a0);
                    a1 = __builtin_fma(p[u].y, p[u].y, a1);
                }
            }
        }
    }
    part[(uint64_t)blockIdx.x * 256 + threadIdx.x] = (a0 + a1) + (a2 + a3);
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
    std::printf("%-44s %8.4f ms  %8.1f GB/s (min) %8.1f GB/s (mean)\n", label, tmin, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nmax = 1ull << 29;
    d2 *x;
    double *part;
    CK(hipMalloc((void **)&x, 8 * nmax));
    CK(hipMalloc((void **)&part, 8ull * 256 * 262144));
    CK(hipMemset(x, 0, 8 * nmax));
    for (uint64_t n : {1ull << 29, 1ull << 27, 1ull << 25, 1ull << 23})
    {
        const uint64_t nv = n / 2;
        std::printf("n = %llu doubles (%.0f MB)\n", (unsigned long long)n, 8.0 * n * 1e-6);
        char label[96];
        for (int g : {512, 1024, 1536, 2048, 2560, 3072, 4096, 8192})
        {
            std::snprintf(label, sizeof label, "grid-stride grid %5d unroll 1", g);
            run(label, 8.0 * n, 15, [&] { stride_sum<1><<<g, 256>>>(x, nv, part); });
            std::snprintf(label, sizeof label, "grid-stride grid %5d unroll 2", g);
            run(label, 8.0 * n, 15, [&] { stride_sum<2><<<g, 256>>>(x, nv, part); });
            std::snprintf(label, sizeof label, "grid-stride grid %5d unroll 4", g);
            run(label, 8.0 * n, 15, [&] { stride_sum<4><<<g, 256>>>(x, nv, part); });
        }
        for (uint32_t tpb : {1u, 2u, 4u, 8u, 16u})
        {
            const uint64_t tiles = (nv + 2047) / 2048;
            const unsigned grid  = (unsigned)((tiles + tpb - 1) / tpb);
            if (grid > 262144)
                continue;
            std::snprintf(label, sizeof label, "runs of %2u tiles of 32 KB (grid %u)", tpb, grid);
            run(label, 8.0 * n, 15, [&] { run_sum<8><<<grid, 256>>>(x, nv, tpb, part); });
        }
    }
    return 0;
}
