// wave_launch.h -- host-side launchers of the wave kernels (shared by the library and tools/sf_tune).
#pragma once

#include "bwdtrans_mfma.h"
#include "bwdtrans_mfma4.h"
#include "bwdtrans_hmfma4.h"
#include "bwdtrans_wave.h"
#include "sf_dispatch.h" // counter_acquire (batch counter of the persistent 2D kernels)

#include <atomic>

namespace sf
{

constexpr int kMaxDev = 64;

// zeroes the eight ticket counters of a launch's 64-byte slot (one device-wide counter, or one per XCD)
static __global__ void counter_reset_kernel(unsigned long long *ctr)
{
    __hip_atomic_store(ctr + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Persistent grid: as many workgroups as the device keeps resident (occupancy query, cached per
// device), never more than there are chunks.  The caches are atomics: two host threads that race on the first launch
// both run the query and store the same answer.
using OccCache = std::atomic<int>[kMaxDev];
template <class K> inline int resident_blocks(K kern, int threads, size_t lds, std::atomic<int> *cache)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev)
        dev = 0;
    int cached = cache[dev].load(std::memory_order_acquire);
    if (cached == 0)
    {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)kern,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int bpc = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, threads, lds) != hipSuccess ||
            bpc < 1)
        {
            (void)hipGetLastError();
            bpc = 1;
        }
        cache[dev].store(bpc, std::memory_order_release);
        cached = bpc;
    }
    return cached * device_info().num_cu;
}

template <int NQ, int EC, int WPB, int BMODE, int MINW, int KMAP = 0, int OUTM = OUT_ST8, int MEMF = 0,
          typename T = double>
inline int launch_hex_wave(const HexArgsT<T> &a, hipStream_t s, int grid_override = 0)
{
    static OccCache cache = {};
    auto kern            = hex_wave_kernel<NQ, EC, WPB, BMODE, MINW, KMAP, OUTM, MEMF, T>;
    constexpr size_t lds = wave_lds_bytes<NQ, EC, 3, WPB, BMODE, OUTM, T>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t nchunk = (a.nelmt + EC - 1) / EC;
    const uint64_t per    = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need   = (nchunk + per - 1) / per;
    uint64_t grid         = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache);
    if (grid_override > 0)
        grid = (uint64_t)grid_override;
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ, int EC, int WPB, int BMODE, int MINW, int KMAP = 0, int OUTM = OUT_ST8, int MEMF = 0,
          typename T = double>
inline int launch_quad_wave(const QuadArgsT<T> &a, hipStream_t s, int grid_override = 0)
{
    static OccCache cache = {};
    auto kern            = quad_wave_kernel<NQ, EC, WPB, BMODE, MINW, KMAP, OUTM, MEMF, T>;
    constexpr size_t lds = wave_lds_bytes<NQ, EC, 2, WPB, BMODE, OUTM, T>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t nchunk = (a.nelmt + EC - 1) / EC;
    const uint64_t per    = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need   = (nchunk + per - 1) / per;
    uint64_t grid         = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache);
    if (grid_override > 0)
        grid = (uint64_t)grid_override;
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ, int EC, int WPB, int MINW, int KMAP, bool OUTL = false, int XG = 0, typename T = double>
inline int launch_quad_mfma(const QuadArgsT<T> &a, hipStream_t s, int grid_override = 0)
{
    static OccCache cache = {};
    auto kern            = quad_mfma_kernel<NQ, EC, WPB, MINW, KMAP, OUTL, XG, T>;
    constexpr size_t lds = mfma_lds_bytes<NQ, EC, WPB, T>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t nchunk = (a.nelmt + EC - 1) / EC;
    const uint64_t per    = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need   = (nchunk + per - 1) / per;
    uint64_t grid         = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache);
    if (grid_override > 0)
        grid = (uint64_t)grid_override;
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ, int EB, int WPB, int MINW, int GJ, int KMAP, int XG, bool SHB, int DYNB, bool PEEL = true, bool SPLIT = false,
          int XR = 0>
inline int launch_quad_mfma4_impl(const QuadArgs &a, hipStream_t s)
{
    static OccCache cache = {};
    auto kern            = quad_mfma4_kernel<NQ, EB, WPB, MINW, GJ, KMAP, XG, SHB, DYNB, PEEL, SPLIT, false, false, XR>;
    constexpr size_t lds = mfma4_lds_bytes<NQ, EB, WPB, SHB>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    static_assert(DYNB == 0 || KMAP == 0, "the batch counter feeds a persistent grid");
    const uint64_t nchunk = (a.nelmt + EB - 1) / EB;
    const uint64_t per    = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need   = (nchunk + per - 1) / per;
    uint64_t grid         = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache); // also raises the LDS limit
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    if constexpr (DYNB > 0)
    {
        // the batch counter comes from the device's counter ring (never evicted: a pointer baked into a captured graph
        // stays valid until sf_shutdown); no counter to be had (ring exhausted, or first use inside a capture) -> the
        // same kernel with a fixed share per wave
        unsigned long long *ctr = nullptr;
        if (counter_acquire(s, &ctr) != SF_OK)
            return launch_quad_mfma4_impl<NQ, EB, WPB, MINW, GJ, KMAP, XG, SHB, 0, PEEL, SPLIT, 0>(a, s);
        // zeroed by a one-thread kernel, not a memset: under stream capture a memset node on a pointer INSIDE an
        // allocation did not zero the counter on ROCm 7.2 (the replayed grid then saw a stale ticket and exited)
        counter_reset_kernel<<<1, 8, 0, s>>>(ctr);
        kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt, ctr, nullptr);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SF_OK : (int)e;
    }
    else
    {
        kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt, nullptr, nullptr);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SF_OK : (int)e;
    }
}

// SHBONLY: the configuration only fits the LDS with one basis copy (b0 == b1)
template <int NQ, int EB, int WPB, int MINW, int GJ, int KMAP, int XG = 0, bool SHBONLY = false, int DYNB = 0,
          bool PEEL = true, bool SPLIT = false, int XR = 0>
inline int launch_quad_mfma4(const QuadArgs &a, hipStream_t s)
{
    if (a.nelmt == 0)
        return SF_OK;
    if (a.b0 == a.b1)
        return launch_quad_mfma4_impl<NQ, EB, WPB, MINW, GJ, KMAP, XG, true, DYNB, PEEL, SPLIT, XR>(a, s);
    if constexpr (SHBONLY)
        return SF_ENOTBUILT;
    else
        return launch_quad_mfma4_impl<NQ, EB, WPB, MINW, GJ, KMAP, XG, false, DYNB, PEEL, SPLIT, XR>(a, s);
}

template <int NQ, int EC, int WPB, int MINW, int KMAP, int XG = 0, typename T = double>
inline int launch_hex_mfma(const HexArgsT<T> &a, hipStream_t s, int grid_override = 0)
{
    static OccCache cache = {};
    auto kern            = hex_mfma_kernel<NQ, EC, WPB, MINW, KMAP, XG, T>;
    constexpr size_t lds = hex_mfma_lds_bytes<NQ, EC, WPB, T>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t nchunk = (a.nelmt + EC - 1) / EC;
    const uint64_t per    = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need   = (nchunk + per - 1) / per;
    uint64_t grid         = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache);
    if (grid_override > 0)
        grid = (uint64_t)grid_override;
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ, int WPB, int MINW, int KMAP, int XG = 0, bool DIRECT = false, bool NTS = true, bool PEEL = true>
inline int launch_hex_mfma4(const HexArgs &a, hipStream_t s, int grid_override = 0)
{
    static OccCache cache = {};
    auto kern            = hex_mfma4_kernel<NQ, WPB, MINW, KMAP, XG, false, DIRECT, NTS, PEEL>;
    constexpr size_t lds = hex_mfma4_lds_bytes<NQ, WPB, DIRECT>();
    static_assert(lds <= 160 * 1024, "LDS slab exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t per  = (uint64_t)WPB * (KMAP > 0 ? KMAP : (KMAP < 0 ? -KMAP : 1));
    const uint64_t need = (a.nelmt + per - 1) / per;
    uint64_t grid       = (uint64_t)resident_blocks(kern, kWave * WPB, lds, cache);
    if (grid_override > 0)
        grid = (uint64_t)grid_override;
    if (grid > need || KMAP != 0)
        grid = need;
    if (grid > 0x7fffffffull)
        return SF_EINVAL;
    kern<<<(unsigned)grid, kWave * WPB, lds, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt, nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

} // namespace sf
