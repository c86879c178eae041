// sf_launch_latency.hip -- host round trip of "launch + synchronise" (the reference's timing protocol,
// benchmark05/benchmark05.cc:1319-1332) for an empty kernel, under the synchronisation methods HIP offers.
// usage: sf_launch_latency [auto|spin|yield|blocking]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

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

__global__ void empty_kernel(int *p)
{
    if (p && threadIdx.x == 12345)
        *p = 1;
}

static double best_us(int reps, const std::function<void()> &f)
{
    double best = 1e30;
    for (int r = 0; r < reps; ++r)
    {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        const auto t1 = std::chrono::steady_clock::now();
        best          = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    return best;
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "auto";
    unsigned flags   = hipDeviceScheduleAuto;
    if (!std::strcmp(mode, "spin"))
        flags = hipDeviceScheduleSpin;
    else if (!std::strcmp(mode, "yield"))
        flags = hipDeviceScheduleYield;
    else if (!std::strcmp(mode, "blocking"))
        flags = hipDeviceScheduleBlockingSync;
    CK(hipSetDeviceFlags(flags));
    CK(hipFree(nullptr));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int i = 0; i < 20; ++i)
        empty_kernel<<<1, 64>>>(nullptr);
    CK(hipDeviceSynchronize());
    const int reps = 2000;
    std::printf("schedule flag %-9s  launch + hipDeviceSynchronize (null stream)   %7.2f us\n", mode,
                best_us(reps, [&] { empty_kernel<<<1, 64>>>(nullptr); (void)hipDeviceSynchronize(); }));
    std::printf("schedule flag %-9s  launch + hipStreamSynchronize (own stream)    %7.2f us\n", mode,
                best_us(reps, [&] { empty_kernel<<<1, 64, 0, s>>>(nullptr); (void)hipStreamSynchronize(s); }));
    std::printf("schedule flag %-9s  launch + event record + hipEventQuery spin     %7.2f us\n", mode,
                best_us(reps, [&] {
                    empty_kernel<<<1, 64, 0, s>>>(nullptr);
                    (void)hipEventRecord(ev, s);
                    while (hipEventQuery(ev) == hipErrorNotReady)
                    {
                    }
                }));
    std::printf("schedule flag %-9s  launch only (no wait)                          %7.2f us\n", mode,
                best_us(reps, [&] { empty_kernel<<<1, 64, 0, s>>>(nullptr); }));
    CK(hipDeviceSynchronize());
    return 0;
}
