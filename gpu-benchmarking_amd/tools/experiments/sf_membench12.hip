// sf_membench12.hip -- does the memory system reward reads and writes that arrive in separate PHASES?
// Pure-read streams reach 0.90 and pure-write streams 0.97 of the 8 TB/s peak on this part, a copy (both at once) 0.83.
// Here every wave of a flat 16-byte copy (K vectors per thread, a contiguous piece per workgroup) issues its loads only
// while the chip-wide 100 MHz real-time counter (s_memrealtime) is in a "read slot" and its stores only in a "write
// slot": no communication, every wave derives the phase from the clock.  Period and read share are swept; mode 0 is the
// same kernel without the waits.
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

// true while the clock is inside the read part [0, rd) of the period
__device__ __forceinline__ bool in_read_slot(unsigned period, unsigned rd)
{
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    return (unsigned)(t % period) < rd;
}

template <int K, bool PHASED>
__global__ __launch_bounds__(256) void phased_copy(const d2 *__restrict__ s, d2 *__restrict__ d, uint64_t nv, unsigned period,
                                                   unsigned rd)
{
    const uint64_t base = (uint64_t)blockIdx.x * (256ull * K) + threadIdx.x;
    if (base + (K - 1) * 256ull >= nv)
        return;
    if (PHASED)
        while (!in_read_slot(period, rd)) // the slot flips with the clock: the loop always ends
            __builtin_amdgcn_s_sleep(8);
    d2 x[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        x[k] = __builtin_nontemporal_load(s + base + k * 256ull);
#pragma unroll
    for (int k = 0; k < K; ++k)
        asm volatile("" : "+v"(x[k])); // the loads have landed
    if (PHASED)
        while (in_read_slot(period, rd))
            __builtin_amdgcn_s_sleep(8);
#pragma unroll
    for (int k = 0; k < K; ++k)
        __builtin_nontemporal_store(x[k], d + base + k * 256ull);
}

__global__ void fill_pattern(double *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = 1.0 + 1e-9 * (double)(i * 2654435761ull % 1000003ull);
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
    std::printf("%-62s %8.1f GB/s (min) %8.1f GB/s (mean) = %.3f of 8 TB/s\n", label, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6,
                bytes / (tsum / reps) * 1e-6 / 8000.0);
    std::fflush(stdout);
}

template <int K> static void rows(const d2 *s, d2 *d, uint64_t nv)
{
    const double bytes    = 32.0 * nv;
    const unsigned blocks = (unsigned)(nv / (256ull * K));
    char label[128];
    std::snprintf(label, sizeof label, "copy, %d vectors per thread (%d KB per wave), no phases", K, K);
    run(label, bytes, 10, [&] { phased_copy<K, false><<<blocks, 256>>>(s, d, nv, 0, 0); });
    for (unsigned period : {200u, 400u, 800u, 1600u, 3200u}) // ticks of 10 ns
        for (unsigned share : {40u, 50u, 60u})
        {
            std::snprintf(label, sizeof label, "  phased: period %5.1f us, read slot %2u %%, %d KB per wave", period * 0.01, share, K);
            run(label, bytes, 10, [&] { phased_copy<K, true><<<blocks, 256>>>(s, d, nv, period, period * share / 100); });
        }
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nv = 1ull << 27; // 2 GiB in, 2 GiB out
    d2 *s, *d;
    CK(hipMalloc((void **)&s, 16 * nv));
    CK(hipMalloc((void **)&d, 16 * nv));
    fill_pattern<<<4096, 256>>>((double *)s, 2 * nv);
    CK(hipMemset(d, 0, 16 * nv));
    CK(hipDeviceSynchronize());
    rows<1>(s, d, nv);
    rows<4>(s, d, nv);
    rows<8>(s, d, nv);
    return 0;
}
