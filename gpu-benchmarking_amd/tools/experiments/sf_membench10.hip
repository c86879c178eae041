// sf_membench10.hip -- can a PERSISTENT grid copy as fast as a dispatcher-ordered one once its loop waits exactly?
// 16-byte copy, persistent grid-stride, software-pipelined D vectors deep per thread: the loads of step n+1 are issued
// before the stores of step n, and the wait in front of the stores counts the younger loads (vmcnt(D)) instead of
// draining everything -- the compiler's own schedule of `dst[v] = src[v]` in a loop waits for vmcnt(0), i.e. for the
// thread's previous STORES as well.  Against: one vector per thread with a grid that covers the array.
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

__global__ __launch_bounds__(256) void flat_copy(const d2 *__restrict__ s, d2 *__restrict__ d, uint64_t nv)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv)
        __builtin_nontemporal_store(__builtin_nontemporal_load(s + v), d + v);
}

// plain persistent loop, D independent vectors per step (what the compiler makes of it)
template <int D> __global__ __launch_bounds__(256) void loop_copy(const d2 *__restrict__ s, d2 *__restrict__ d, uint64_t nv)
{
    const uint64_t T = (uint64_t)gridDim.x * 256;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v + (D - 1) * T < nv; v += D * T)
    {
        d2 x[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            x[k] = __builtin_nontemporal_load(s + v + k * T);
#pragma unroll
        for (int k = 0; k < D; ++k)
            __builtin_nontemporal_store(x[k], d + v + k * T);
    }
}

// pipelined: step n+1's loads are in flight while step n is stored; the touch at the end of the iteration puts the wait
// where the number of younger memory operations is known (touch_staged() of bwdtrans_wave.h)
template <int D> __global__ __launch_bounds__(256) void piped_copy(const d2 *__restrict__ s, d2 *__restrict__ d, uint64_t nv)
{
    const uint64_t T = (uint64_t)gridDim.x * 256;
    uint64_t v       = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v + (D - 1) * T >= nv)
        return;
    d2 x[D], y[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        x[k] = __builtin_nontemporal_load(s + v + k * T);
    for (;;)
    {
        const uint64_t vn = v + D * T;
        const bool more   = vn + (D - 1) * T < nv;
        if (more)
        {
#pragma unroll
            for (int k = 0; k < D; ++k)
                y[k] = __builtin_nontemporal_load(s + vn + k * T);
        }
#pragma unroll
        for (int k = 0; k < D; ++k)
            __builtin_nontemporal_store(x[k], d + v + k * T);
        if (!more)
            break;
#pragma unroll
        for (int k = 0; k < D; ++k)
        {
            asm volatile("" : "+v"(y[k]));
            x[k] = y[k];
        }
        v = vn;
    }
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
    std::printf("%-46s %8.1f GB/s (min) %8.1f GB/s (mean)\n", label, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nv = 1ull << 28; // 4 GiB in, 4 GiB out
    d2 *s, *d;
    CK(hipMalloc((void **)&s, 16 * nv));
    CK(hipMalloc((void **)&d, 16 * nv));
    CK(hipMemset(s, 0, 16 * nv));
    CK(hipMemset(d, 0, 16 * nv));
    const double bytes = 32.0 * nv;
    char label[96];
    run("flat copy, grid covers the array", bytes, 10, [&] { flat_copy<<<(unsigned)(nv / 256), 256>>>(s, d, nv); });
    for (int g : {1024, 1536, 2048, 3072, 4096, 8192})
    {
        std::snprintf(label, sizeof label, "persistent loop      grid %5d depth 1", g);
        run(label, bytes, 10, [&] { loop_copy<1><<<g, 256>>>(s, d, nv); });
        std::snprintf(label, sizeof label, "persistent loop      grid %5d depth 4", g);
        run(label, bytes, 10, [&] { loop_copy<4><<<g, 256>>>(s, d, nv); });
        std::snprintf(label, sizeof label, "persistent pipelined grid %5d depth 1", g);
        run(label, bytes, 10, [&] { piped_copy<1><<<g, 256>>>(s, d, nv); });
        std::snprintf(label, sizeof label, "persistent pipelined grid %5d depth 2", g);
        run(label, bytes, 10, [&] { piped_copy<2><<<g, 256>>>(s, d, nv); });
        std::snprintf(label, sizeof label, "persistent pipelined grid %5d depth 4", g);
        run(label, bytes, 10, [&] { piped_copy<4><<<g, 256>>>(s, d, nv); });
    }
    run("flat copy, grid covers the array", bytes, 10, [&] { flat_copy<<<(unsigned)(nv / 256), 256>>>(s, d, nv); });
    return 0;
}
