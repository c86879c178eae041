// sf_membench2.hip -- which chunk->wave mapping streams the nq=8 hex traffic shape fastest?
// Per chunk (2 elements): read 5488 B, write 8192 B, no arithmetic.
//   MAP 0: strided   wave gw handles chunks gw, gw+W, ...        (K iterations)
//   MAP 1: blocked   wave gw handles chunks gw*K .. gw*K+K-1
//   MAP 2: block-blocked: workgroup b handles chunks [b*4K, (b+1)*4K), wave wib takes every 4th
// grid = ceil(nchunk / (WPB*K)); K=0 means persistent grid of `pgrid` blocks (MAP 0 only).
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

constexpr int IN_D = 686, OUT_D = 1024;

template <int ST16, bool NT, bool RD, bool WR>
__device__ __forceinline__ void one_chunk(const double *__restrict__ in, double *__restrict__ out,
                                          uint64_t c, int lane, double &sink)
{
    d2 x[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
        x[k] = d2{1.0 + k, 2.0};
    if (RD)
    {
        const d2 *src = reinterpret_cast<const d2 *>(in + c * IN_D);
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            const int v = k * 64 + lane;
            if (v < IN_D / 2)
                x[k] = NT ? __builtin_nontemporal_load(src + v) : src[v];
        }
    }
    if (WR)
    {
        double *dst = out + c * OUT_D;
        if (!ST16)
        {
#pragma unroll
            for (int k = 0; k < 16; ++k)
            {
                const double val = (k & 1) ? x[(k >> 1) % 6].y : x[(k >> 1) % 6].x;
                if (NT)
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
                if (NT)
                    __builtin_nontemporal_store(x[k % 6], dst2 + k * 64 + lane);
                else
                    dst2[k * 64 + lane] = x[k % 6];
            }
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            sink += x[k].x + x[k].y;
    }
}

template <int MAP, int ST16, bool NT, bool RD, bool WR>
__global__ __launch_bounds__(256) void hexshape(const double *__restrict__ in,
                                                double *__restrict__ out, uint64_t nchunk, int K)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint64_t nwave = (uint64_t)gridDim.x * 4;
    const uint64_t gw    = (uint64_t)blockIdx.x * 4 + wib;
    double sink          = 0;
    if (MAP == 0)
    {
        for (uint64_t c = gw; c < nchunk; c += nwave)
            one_chunk<ST16, NT, RD, WR>(in, out, c, lane, sink);
    }
    else if (MAP == 1)
    {
        for (uint64_t c = gw * K; c < nchunk && c < (gw + 1) * K; ++c)
            one_chunk<ST16, NT, RD, WR>(in, out, c, lane, sink);
    }
    else
    {
        const uint64_t base = (uint64_t)blockIdx.x * 4 * K;
        for (uint64_t c = base + wib; c < nchunk && c < base + 4 * (uint64_t)K; c += 4)
            one_chunk<ST16, NT, RD, WR>(in, out, c, lane, sink);
    }
    if (sink == 123.456)
        out[0] = sink;
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
    std::printf("%-46s min %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label, tmin,
                bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 10;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nchunk = 1 << 19;
    const uint64_t nin = nchunk * IN_D, nout = nchunk * OUT_D;
    double *in, *out;
    CK(hipMalloc((void **)&in, 8 * nin));
    CK(hipMalloc((void **)&out, 8 * nout));
    CK(hipMemset(in, 0, 8 * nin));
    CK(hipMemset(out, 0, 8 * nout));
    const double hb = 8.0 * (nin + nout);
    char label[128];
    auto grid_for = [&](int K) { return (unsigned)((nchunk + 4 * (uint64_t)K - 1) / (4 * (uint64_t)K)); };
    for (int K : {1, 2, 4, 8, 16, 32, 64})
    {
        const unsigned g = grid_for(K);
        std::snprintf(label, sizeof label, "strided  st16 nt K=%d grid %u", K, g);
        run(label, hb, reps, [&] { hexshape<0, 1, true, true, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "blocked  st16 nt K=%d grid %u", K, g);
        run(label, hb, reps, [&] { hexshape<1, 1, true, true, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "blkblk   st16 nt K=%d grid %u", K, g);
        run(label, hb, reps, [&] { hexshape<2, 1, true, true, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "blkblk   st16 plain K=%d grid %u", K, g);
        run(label, hb, reps, [&] { hexshape<2, 1, false, true, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "blkblk   st8  nt K=%d grid %u", K, g);
        run(label, hb, reps, [&] { hexshape<2, 0, true, true, true><<<g, 256>>>(in, out, nchunk, K); });
    }
    for (int K : {1, 8})
    {
        const unsigned g = grid_for(K);
        std::snprintf(label, sizeof label, "read-only  nt blkblk K=%d", K);
        run(label, 8.0 * nin, reps, [&] { hexshape<2, 1, true, true, false><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "write-only st16 nt blkblk K=%d", K);
        run(label, 8.0 * nout, reps, [&] { hexshape<2, 1, true, false, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "write-only st16 plain blkblk K=%d", K);
        run(label, 8.0 * nout, reps, [&] { hexshape<2, 1, false, false, true><<<g, 256>>>(in, out, nchunk, K); });
        std::snprintf(label, sizeof label, "write-only st8 nt blkblk K=%d", K);
        run(label, 8.0 * nout, reps, [&] { hexshape<2, 0, true, false, true><<<g, 256>>>(in, out, nchunk, K); });
    }
    return 0;
}
