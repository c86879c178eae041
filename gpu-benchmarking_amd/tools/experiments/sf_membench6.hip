// sf_membench6.hip -- cache-policy bits of the streaming accesses: the flagship's traffic mix (343 16-B lanes read,
// 512 written per piece, one 256-thread workgroup per piece, no arithmetic) with every combination of the gfx950
// load / store policy bits (none, nt, sc0, sc1, sc0 sc1, nt sc0 sc1 ...) issued through inline asm.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2);} } while (0)

template <int LP> __device__ __forceinline__ d2 ld(const d2 *p)
{
    d2 v;
    if constexpr (LP == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LP == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LP == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LP == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LP == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LP == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SP> __device__ __forceinline__ void st(d2 *p, d2 v)
{
    if constexpr (SP == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (SP == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (SP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (SP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SP == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SP == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int IN_L, int OUT_L, int LP, int SP>
__global__ __launch_bounds__(256) void mix_block(const d2 *__restrict__ in, d2 *__restrict__ out, uint64_t nblk)
{
    const uint64_t b = blockIdx.x;
    if (b >= nblk) return;
    const d2 *src = in + b * IN_L; d2 *dst = out + b * OUT_L;
    d2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
    if (threadIdx.x < IN_L) a0 = ld<LP>(src + threadIdx.x);
    if (threadIdx.x + 256 < IN_L) a1 = ld<LP>(src + threadIdx.x + 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const d2 acc = a0 + a1;
    for (int i = threadIdx.x; i < OUT_L; i += 256) st<SP>(dst + i, acc);
}
static hipEvent_t e0, e1;
static void run(const char *label, double bytes, int reps, const std::function<void()> &f)
{
    f(); CK(hipDeviceSynchronize()); double tmin = 1e30, tsum = 0;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tmin = ms < tmin ? ms : tmin; tsum += ms; }
    CK(hipGetLastError());
    std::printf("%-34s min %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label, tmin, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}
static const char *pname[] = {"-", "nt", "sc0", "sc1", "sc0sc1", "sc0sc1nt"};
template <int LP, int SP> void go(d2 *in, d2 *out, uint64_t nblk, int reps)
{
    char label[64];
    std::snprintf(label, sizeof label, "343:512 load %-8s store %-8s", pname[LP], pname[SP]);
    run(label, 16.0 * nblk * (343 + 512), reps, [&] { mix_block<343, 512, LP, SP><<<(unsigned)nblk, 256>>>(in, out, nblk); });
}
template <int LP> void row(d2 *in, d2 *out, uint64_t nblk, int reps)
{
    go<LP, 0>(in, out, nblk, reps); go<LP, 1>(in, out, nblk, reps); go<LP, 2>(in, out, nblk, reps);
    go<LP, 3>(in, out, nblk, reps); go<LP, 4>(in, out, nblk, reps); go<LP, 5>(in, out, nblk, reps);
}
int main()
{
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t nblk = 1 << 20; d2 *in, *out;
    CK(hipMalloc((void **)&in, 16ull * nblk * 343)); CK(hipMalloc((void **)&out, 16ull * nblk * 512));
    CK(hipMemset(in, 0, 16ull * nblk * 343)); CK(hipMemset(out, 0, 16ull * nblk * 512));
    const int reps = 12;
    row<0>(in, out, nblk, reps); row<1>(in, out, nblk, reps); row<2>(in, out, nblk, reps);
    row<3>(in, out, nblk, reps); row<4>(in, out, nblk, reps); row<5>(in, out, nblk, reps);
    return 0;
}
