// sf_membench.hip -- HBM streaming ceilings of this device for the access shapes the BwdTrans
// kernels use (development tool; numbers quoted in DESIGN.md).
//   copy / read / write with 8- and 16-byte lanes, plain vs non-temporal, several grid sizes;
//   "hexshape": per wave and iteration read 5488 B and write 8192 B (nq=8 chunk of 2 elements),
//   no arithmetic -> the ceiling of the flagship kernel's traffic pattern.
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

template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void copy16(const d2 *__restrict__ s, d2 *__restrict__ d,
                                              uint64_t nv)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t v            = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; v + (UNR - 1) * stride < nv; v += UNR * stride)
    {
        d2 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            x[u] = NTL ? __builtin_nontemporal_load(s + v + u * stride) : s[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
        {
            if (NTS)
                __builtin_nontemporal_store(x[u], d + v + u * stride);
            else
                d[v + u * stride] = x[u];
        }
    }
    for (; v < nv; v += stride)
        d[v] = s[v];
}

template <bool NTS> __global__ __launch_bounds__(256) void write16(d2 *__restrict__ d, uint64_t nv)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const d2 val          = {1.0, 2.0};
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += stride)
    {
        if (NTS)
            __builtin_nontemporal_store(val, d + v);
        else
            d[v] = val;
    }
}

template <bool NTS> __global__ __launch_bounds__(256) void write8(double *__restrict__ d, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += stride)
    {
        if (NTS)
            __builtin_nontemporal_store(1.0, d + v);
        else
            d[v] = 1.0;
    }
}

template <bool NTL>
__global__ __launch_bounds__(256) void read16(const d2 *__restrict__ s, double *__restrict__ sink,
                                              uint64_t nv)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    double a = 0, b = 0;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += stride)
    {
        const d2 x = NTL ? __builtin_nontemporal_load(s + v) : s[v];
        a += x.x;
        b += x.y;
    }
    if (a + b == 123.456)
        sink[0] = a;
}

// per wave: read IN_B bytes (16-B lanes, contiguous), write OUT_B bytes as NST stores per lane
// shape: 0 = 8-B lanes, each store covers 512 contiguous bytes (what the hex kernel does)
//        1 = 16-B lanes, each store covers 1024 contiguous bytes
template <int SHAPE, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void hexshape(const double *__restrict__ in,
                                                double *__restrict__ out, uint64_t nchunk)
{
    constexpr int IN_D = 686, OUT_D = 1024; // doubles per chunk (2 elements, nq=8)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint64_t nwave = (uint64_t)gridDim.x * 4;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + wib; c < nchunk; c += nwave)
    {
        const d2 *src = reinterpret_cast<const d2 *>(in + c * IN_D);
        d2 x[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            const int v = k * 64 + lane;
            x[k]        = d2{0, 0};
            if (v < IN_D / 2)
                x[k] = NTL ? __builtin_nontemporal_load(src + v) : src[v];
        }
        double *dst = out + c * OUT_D;
        if (SHAPE == 0)
        {
#pragma unroll
            for (int k = 0; k < 16; ++k)
            {
                const double val = (k & 1) ? x[(k >> 1) % 6].y : x[(k >> 1) % 6].x;
                if (NTS)
                    __builtin_nontemporal_store(val, dst + k * 64 + lane);
                else
                    dst[k * 64 + lane] = val;
            }
        }
        else
        {
            d2 *dst2 = reinterpret_cast<d2 *>(dst);
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                if (NTS)
                    __builtin_nontemporal_store(x[k % 6], dst2 + k * 64 + lane);
                else
                    dst2[k * 64 + lane] = x[k % 6];
            }
        }
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
    std::printf("%-44s min %8.4f ms  mean %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label,
                tmin, tsum / reps, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 10;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nchunk = 1 << 19;                 // 1 Mi elements
    const uint64_t nin = nchunk * 686, nout = nchunk * 1024;
    double *in, *out;
    CK(hipMalloc((void **)&in, 8 * nin));
    CK(hipMalloc((void **)&out, 8 * nout));
    CK(hipMemset(in, 0, 8 * nin));
    CK(hipMemset(out, 0, 8 * nout));
    const uint64_t nv = nin / 2; // copy in -> out[0:nin]
    char label[128];
    const int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 0};
    for (int g : grids)
    {
        const unsigned grid = g ? g : (unsigned)((nv + 255) / 256);
        std::snprintf(label, sizeof label, "copy16 plain unr1 grid %u", grid);
        run(label, 16.0 * nv * 2, reps, [&] { copy16<false, false, 1><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
        std::snprintf(label, sizeof label, "copy16 nt/nt unr1 grid %u", grid);
        run(label, 16.0 * nv * 2, reps, [&] { copy16<true, true, 1><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
        if (g)
        {
            std::snprintf(label, sizeof label, "copy16 plain unr4 grid %u", grid);
            run(label, 16.0 * nv * 2, reps, [&] { copy16<false, false, 4><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
            std::snprintf(label, sizeof label, "copy16 nt/nt unr4 grid %u", grid);
            run(label, 16.0 * nv * 2, reps, [&] { copy16<true, true, 4><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
            std::snprintf(label, sizeof label, "copy16 ld-nt/st-plain unr4 grid %u", grid);
            run(label, 16.0 * nv * 2, reps, [&] { copy16<true, false, 4><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
            std::snprintf(label, sizeof label, "copy16 ld-plain/st-nt unr4 grid %u", grid);
            run(label, 16.0 * nv * 2, reps, [&] { copy16<false, true, 4><<<grid, 256>>>((d2 *)in, (d2 *)out, nv); });
        }
    }
    for (int g : {256 * 8, 256 * 32})
    {
        std::snprintf(label, sizeof label, "read16 plain grid %d", g);
        run(label, 8.0 * nout, reps, [&] { read16<false><<<g, 256>>>((d2 *)out, in, nout / 2); });
        std::snprintf(label, sizeof label, "read16 nt grid %d", g);
        run(label, 8.0 * nout, reps, [&] { read16<true><<<g, 256>>>((d2 *)out, in, nout / 2); });
        std::snprintf(label, sizeof label, "write16 plain grid %d", g);
        run(label, 8.0 * nout, reps, [&] { write16<false><<<g, 256>>>((d2 *)out, nout / 2); });
        std::snprintf(label, sizeof label, "write16 nt grid %d", g);
        run(label, 8.0 * nout, reps, [&] { write16<true><<<g, 256>>>((d2 *)out, nout / 2); });
        std::snprintf(label, sizeof label, "write8 plain grid %d", g);
        run(label, 8.0 * nout, reps, [&] { write8<false><<<g, 256>>>(out, nout); });
        std::snprintf(label, sizeof label, "write8 nt grid %d", g);
        run(label, 8.0 * nout, reps, [&] { write8<true><<<g, 256>>>(out, nout); });
    }
    const double hb = 8.0 * (nin + nout);
    for (int g : {256 * 3, 256 * 5, 256 * 8})
    {
        std::snprintf(label, sizeof label, "hexshape st8  plain/plain grid %d", g);
        run(label, hb, reps, [&] { hexshape<0, false, false><<<g, 256>>>(in, out, nchunk); });
        std::snprintf(label, sizeof label, "hexshape st8  nt/nt grid %d", g);
        run(label, hb, reps, [&] { hexshape<0, true, true><<<g, 256>>>(in, out, nchunk); });
        std::snprintf(label, sizeof label, "hexshape st8  nt/plain grid %d", g);
        run(label, hb, reps, [&] { hexshape<0, true, false><<<g, 256>>>(in, out, nchunk); });
        std::snprintf(label, sizeof label, "hexshape st16 plain/plain grid %d", g);
        run(label, hb, reps, [&] { hexshape<1, false, false><<<g, 256>>>(in, out, nchunk); });
        std::snprintf(label, sizeof label, "hexshape st16 nt/nt grid %d", g);
        run(label, hb, reps, [&] { hexshape<1, true, true><<<g, 256>>>(in, out, nchunk); });
    }
    return 0;
}
