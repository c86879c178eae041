// sf_membench5.hip -- how much bandwidth does a wave-local "load -> long compute -> store" structure leave on
// the table?  Traffic-only copy of the flagship's shape (per wave and chunk: 10976 B in, 16384 B out, EC = 4,
// K = 2 consecutive chunks with the next chunk prefetched, 4 waves per workgroup) with an artificial delay of
// D shader cycles between the arrival of a chunk and its stores, at two occupancies.
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

constexpr int IN_L = 686, OUT_L = 1024; // 16-B lanes per chunk (4 elements, nq = 8)

template <int LDSPAD>
__global__ __launch_bounds__(256) void shape(const d2 *__restrict__ in, d2 *__restrict__ out,
                                             uint64_t nchunk, int delay)
{
    extern __shared__ double pad[]; // LDSPAD bytes per block only to set the occupancy
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint64_t first = ((uint64_t)blockIdx.x * 4 + wib) * 2;
    d2 x[11], y[11];
    auto load = [&](d2(&r)[11], uint64_t c)
    {
#pragma unroll
        for (int k = 0; k < 11; ++k)
        {
            const int v = k * 64 + lane;
            r[k]        = d2{0, 0};
            if (v < IN_L)
                r[k] = __builtin_nontemporal_load(in + c * IN_L + v);
        }
    };
    auto store = [&](const d2(&r)[11], uint64_t c)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            __builtin_nontemporal_store(r[k % 11], out + c * OUT_L + k * 64 + lane);
    };
    auto wait = [&](d2(&r)[11])
    {
        // consume the loads (forces the wait), then burn `delay` cycles
        double s = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k)
            s += r[k].x;
        if (s == 123.456)
            pad[threadIdx.x] = s;
        const uint64_t t0 = __builtin_readcyclecounter();
        while ((int64_t)(__builtin_readcyclecounter() - t0) < delay)
            __builtin_amdgcn_s_sleep(8);
    };
    if (first >= nchunk)
        return;
    load(x, first);
    if (first + 1 < nchunk)
        load(y, first + 1);
    wait(x);
    store(x, first);
    if (first + 1 < nchunk)
    {
        wait(y);
        store(y, first + 1);
    }
}

static hipEvent_t e0, e1;
static double run(double bytes, int reps, const std::function<void()> &f)
{
    f();
    CK(hipDeviceSynchronize());
    double tsum = 0;
    for (int r = 0; r < reps; ++r)
    {
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tsum += ms;
    }
    CK(hipGetLastError());
    return bytes / (tsum / reps) * 1e-6;
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nchunk = 1 << 18; // 1 Mi elements
    d2 *in, *out;
    CK(hipMalloc((void **)&in, 16ull * nchunk * IN_L));
    CK(hipMalloc((void **)&out, 16ull * nchunk * OUT_L));
    CK(hipMemset(in, 0, 16ull * nchunk * IN_L));
    CK(hipMemset(out, 0, 16ull * nchunk * OUT_L));
    const double bytes = 16.0 * nchunk * (IN_L + OUT_L);
    const unsigned grid = (unsigned)(nchunk / 8);
    CK(hipFuncSetAttribute((const void *)shape<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    std::printf("delay(cycles)   2 blocks/CU (8 waves)   4 blocks/CU (16 waves)   8 blocks/CU (32 waves)   [GB/s mean]\n");
    for (int delay : {0, 1000, 2000, 4000, 8000, 16000})
    {
        const double a = run(bytes, 10, [&] { shape<0><<<grid, 256, 80 * 1024>>>(in, out, nchunk, delay); });
        const double b = run(bytes, 10, [&] { shape<0><<<grid, 256, 40 * 1024>>>(in, out, nchunk, delay); });
        const double c = run(bytes, 10, [&] { shape<0><<<grid, 256, 16 * 1024>>>(in, out, nchunk, delay); });
        std::printf("%8d        %10.1f              %10.1f               %10.1f\n", delay, a, b, c);
        std::fflush(stdout);
    }
    return 0;
}
