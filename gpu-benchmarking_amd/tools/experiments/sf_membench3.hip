// sf_membench3.hip -- is the 40 % read / 60 % write mix itself the ceiling?  Flat, dispatcher-ordered kernels
// (one thread = one 16-B lane, grid covers the array) with the byte mix of the nq=8 hex kernel
// (343 doubles in : 512 doubles out), against copy (50/50) and x += y (67/33) in the same style.
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

// every thread writes one 16-B lane of `out`; R of every W threads also read one lane of `in`
// (R/W = 343/512 -> threads with (v * 343) / 512 changing read a new input lane)
__global__ __launch_bounds__(256) void mix_flat(const d2 *__restrict__ in, d2 *__restrict__ out,
                                                uint64_t nv_out, bool nt)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv_out)
        return;
    const uint64_t r0 = v * 343 / 512, r1 = (v + 1) * 343 / 512;
    d2 x = {1.0, 2.0};
    if (r1 != r0)
        x = nt ? __builtin_nontemporal_load(in + r0) : in[r0];
    if (nt)
        __builtin_nontemporal_store(x, out + v);
    else
        out[v] = x;
}

// block-structured: block b reads a contiguous piece of `in` first (all its lanes), then writes its piece
// of `out`: 256 threads, IN_L input lanes and OUT_L output lanes per block, IN_L/OUT_L = 343/512
template <int PIECES>
__global__ __launch_bounds__(256) void mix_block(const d2 *__restrict__ in, d2 *__restrict__ out,
                                                 uint64_t nblk)
{
    constexpr int IN_L = 343 * PIECES, OUT_L = 512 * PIECES; // lanes per block
    const uint64_t b = blockIdx.x;
    if (b >= nblk)
        return;
    const d2 *src = in + b * IN_L;
    d2 *dst       = out + b * OUT_L;
    d2 acc        = {0.0, 0.0};
    for (int i = threadIdx.x; i < IN_L; i += 256)
        acc += __builtin_nontemporal_load(src + i);
    for (int i = threadIdx.x; i < OUT_L; i += 256)
        __builtin_nontemporal_store(acc, dst + i);
}

__global__ __launch_bounds__(256) void copy_flat(const d2 *__restrict__ in, d2 *__restrict__ out,
                                                 uint64_t nv)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + v), out + v);
}

__global__ __launch_bounds__(256) void add_flat(d2 *__restrict__ x, const d2 *__restrict__ y, uint64_t nv)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv)
        __builtin_nontemporal_store(__builtin_nontemporal_load(x + v) + __builtin_nontemporal_load(y + v),
                                    x + v);
}

__global__ __launch_bounds__(256) void write_flat(d2 *__restrict__ out, uint64_t nv)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv)
        __builtin_nontemporal_store(d2{1.0, 2.0}, out + v);
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
    std::printf("%-44s min %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label, tmin,
                bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nelmt = 1 << 20;
    const uint64_t nv_in = nelmt * 343 / 2, nv_out = nelmt * 512 / 2;
    d2 *in, *out;
    CK(hipMalloc((void **)&in, 16 * (nv_in + 1024)));
    CK(hipMalloc((void **)&out, 16 * nv_out));
    CK(hipMemset(in, 0, 16 * (nv_in + 1024)));
    CK(hipMemset(out, 0, 16 * nv_out));
    const double mixb = 16.0 * (nv_in + nv_out);
    run("mix 343:512 flat nt", mixb, reps, [&] { mix_flat<<<(unsigned)((nv_out + 255) / 256), 256>>>(in, out, nv_out, true); });
    run("mix 343:512 flat plain", mixb, reps, [&] { mix_flat<<<(unsigned)((nv_out + 255) / 256), 256>>>(in, out, nv_out, false); });
    run("mix block 1 piece (5.5K in/8K out)", mixb, reps, [&] { mix_block<1><<<(unsigned)(nv_out / 512), 256>>>(in, out, nv_out / 512); });
    run("mix block 4 pieces (22K/32K)", mixb, reps, [&] { mix_block<4><<<(unsigned)(nv_out / 2048), 256>>>(in, out, nv_out / 2048); });
    run("mix block 16 pieces (88K/128K)", mixb, reps, [&] { mix_block<16><<<(unsigned)(nv_out / 8192), 256>>>(in, out, nv_out / 8192); });
    run("copy flat (50/50)", 32.0 * nv_in, reps, [&] { copy_flat<<<(unsigned)((nv_in + 255) / 256), 256>>>(in, out, nv_in); });
    run("x += y flat (67/33)", 48.0 * (nv_out / 2), reps, [&] { add_flat<<<(unsigned)((nv_out / 2 + 255) / 256), 256>>>(out, out + nv_out / 2, nv_out / 2); });
    run("write-only flat", 16.0 * nv_out, reps, [&] { write_flat<<<(unsigned)((nv_out + 255) / 256), 256>>>(out, nv_out); });
    return 0;
}
