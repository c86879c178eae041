// sf_membench7.hip -- XCD-aware workgroup numbering on pure streams: copy and x += y with one 16-byte lane per
// thread, workgroups renumbered so that runs of XG neighbouring workgroups (XG * 4 KiB of each stream) execute on
// the same XCD (workgroups are dealt round-robin to the 8 XCDs; logical_block() in csrc/sf_common.h).
#include "../csrc/sf_common.h"
#include <cstdio>
#include <cstdlib>
#include <functional>
using namespace sf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2);} } while (0)

template <int XG> __global__ __launch_bounds__(256) void copy_k(const double2_t *__restrict__ s, double2_t *__restrict__ d, uint64_t nv)
{
    const uint64_t v = logical_block<XG>() * 256 + threadIdx.x;
    if (v < nv) __builtin_nontemporal_store(__builtin_nontemporal_load(s + v), d + v);
}
template <int XG> __global__ __launch_bounds__(256) void add_k(double2_t *__restrict__ x, const double2_t *__restrict__ y, uint64_t nv)
{
    const uint64_t v = logical_block<XG>() * 256 + threadIdx.x;
    if (v < nv) { const double2_t a = __builtin_nontemporal_load(x + v), b = __builtin_nontemporal_load(y + v); __builtin_nontemporal_store(a + b, x + v); }
}
static hipEvent_t e0, e1;
static void run(const char *label, double bytes, int reps, const std::function<void()> &f)
{
    f(); CK(hipDeviceSynchronize()); double tmin = 1e30, tsum = 0;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tmin = ms < tmin ? ms : tmin; tsum += ms; }
    CK(hipGetLastError());
    std::printf("%-28s min %8.4f ms  %8.1f GB/s (min)  %8.1f GB/s (mean)\n", label, tmin, bytes / tmin * 1e-6, bytes / (tsum / reps) * 1e-6);
    std::fflush(stdout);
}
template <int XG> void go(double2_t *a, double2_t *b, uint64_t nv, int reps)
{
    char l[64];
    const unsigned blocks = (unsigned)((nv + 255) / 256);
    std::snprintf(l, sizeof l, "copy   xg%d", XG);
    run(l, 32.0 * nv, reps, [&] { copy_k<XG><<<blocks, 256>>>(a, b, nv); });
    std::snprintf(l, sizeof l, "x += y xg%d", XG);
    run(l, 48.0 * nv, reps, [&] { add_k<XG><<<blocks, 256>>>(a, b, nv); });
}
int main()
{
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t nv = 1ull << 28; // 4 GiB per array
    double2_t *a, *b;
    CK(hipMalloc((void **)&a, 16 * nv)); CK(hipMalloc((void **)&b, 16 * nv));
    CK(hipMemset(a, 0, 16 * nv)); CK(hipMemset(b, 0, 16 * nv));
    const int reps = 10;
    for (int rep = 0; rep < 2; ++rep)
    {
        go<0>(a, b, nv, reps); go<4>(a, b, nv, reps); go<16>(a, b, nv, reps); go<64>(a, b, nv, reps);
        go<256>(a, b, nv, reps); go<1024>(a, b, nv, reps); go<4096>(a, b, nv, reps);
    }
    return 0;
}
