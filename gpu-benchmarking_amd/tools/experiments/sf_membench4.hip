// sf_membench4.hip -- read/write-mix ceilings vs ratio and piece size: block b reads IN_L 16-B lanes then writes
// OUT_L lanes of two contiguous streams, no arithmetic (the 1:0 row is dead-code-eliminated: ignore it).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2);} } while (0)
// block b reads IN_L 16-B lanes then writes OUT_L lanes (contiguous pieces of two streams)
template <int IN_L, int OUT_L>
__global__ __launch_bounds__(256) void mix_block(const d2 *__restrict__ in, d2 *__restrict__ out, uint64_t nblk)
{
    const uint64_t b = blockIdx.x;
    if (b >= nblk) return;
    const d2 *src = in + b * IN_L; d2 *dst = out + b * OUT_L; d2 acc = {0.0, 0.0};
    for (int i = threadIdx.x; i < IN_L; i += 256) acc += __builtin_nontemporal_load(src + i);
    for (int i = threadIdx.x; i < OUT_L; i += 256) __builtin_nontemporal_store(acc, dst + i);
}
static hipEvent_t e0, e1;
static void run(const char *label, double bytes, int reps, const std::function<void()> &f)
{
    f(); CK(hipDeviceSynchronize()); double tmin = 1e30, tsum = 0;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tmin = ms < tmin ? ms : tmin; tsum += ms; }
    CK(hipGetLastError());
    std::printf("%-34s min %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label, tmin, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
}
template <int IN_L, int OUT_L> void go(const char *label, d2 *in, d2 *out, uint64_t nblk, int reps)
{ run(label, 16.0 * nblk * (IN_L + OUT_L), reps, [&] { mix_block<IN_L, OUT_L><<<(unsigned)nblk, 256>>>(in, out, nblk); }); }
int main()
{
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t nblk = 1 << 19; d2 *in, *out;
    CK(hipMalloc((void **)&in, 16ull * nblk * 768)); CK(hipMalloc((void **)&out, 16ull * nblk * 768));
    CK(hipMemset(in, 0, 16ull * nblk * 768)); CK(hipMemset(out, 0, 16ull * nblk * 768));
    const int reps = 15;
    go<512, 512>("1:1  512:512 lanes", in, out, nblk, reps);
    go<256, 512>("1:2  256:512", in, out, nblk, reps);
    go<384, 512>("3:4  384:512", in, out, nblk, reps);
    go<352, 512>("11:16 352:512", in, out, nblk, reps);
    go<343, 512>("343:512", in, out, nblk, reps);
    go<344, 512>("344:512 (128-B multiple)", in, out, nblk, reps);
    go<320, 512>("5:8  320:512", in, out, nblk, reps);
    go<512, 343>("512:343 (read-heavy)", in, out, nblk, reps);
    go<512, 256>("2:1  512:256", in, out, nblk, reps);
    go<0, 512>("0:1 write-only", in, out, nblk, reps);
    go<512, 0>("1:0 read-only", in, out, nblk, reps);
    go<686, 1024>("686:1024", in, out, nblk / 2, reps);
    go<171, 256>("171:256 (~343:512 halved)", in, out, nblk, reps);
    return 0;
}
