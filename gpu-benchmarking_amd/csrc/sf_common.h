// sf_common.h -- shared host/device helpers for libsumfact (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/sumfact.h"

namespace sf
{

constexpr int kWave = 64; // CDNA4 wavefront

typedef double double2_t __attribute__((ext_vector_type(2)));

// Launch descriptors filled by the per-nq dispatch tables.
template <typename T> struct HexArgsT
{
    const T *b0, *b1, *b2, *in;
    T *wsp, *out;
    uint64_t nelmt;
};

template <typename T> struct QuadArgsT
{
    const T *b0, *b1, *in;
    T *wsp, *out;
    uint64_t nelmt;
};

using HexArgs  = HexArgsT<double>;
using QuadArgs = QuadArgsT<double>;

struct DeviceInfo
{
    int num_cu;
    int device;
};

const DeviceInfo &device_info();

// Intra-wave LDS hand-off: the 64 lanes of ONE wavefront exchange data through LDS without a
// workgroup barrier.  DS operations of a wave execute in issue order, so the only requirement is
// that the compiler keeps the program order of the LDS accesses on both sides of this point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroups are dealt round-robin to the 8 XCDs (one L2 each).  XG > 0 renumbers them so that runs of XG
// consecutive LOGICAL workgroups -- neighbours in memory -- execute on the same XCD, inside a window of 8*XG
// workgroups that keeps the DRAM access front as compact as before (the grid must be a multiple of 8*XG or the
// tail window falls back to the identity).
template <int XG> __device__ __forceinline__ uint64_t logical_block()
{
    const uint64_t b = blockIdx.x;
    if constexpr (XG <= 0)
        return b;
    else
    {
        constexpr uint64_t W = 8ull * XG;
        const uint64_t win = b / W, r = b - win * W;
        if ((win + 1) * W > gridDim.x)
            return b; // partial tail window
        return win * W + (r & 7) * XG + (r >> 3);
    }
}

template <int A, int B> struct CMax
{
    static constexpr int value = A > B ? A : B;
};

constexpr int cdiv(int a, int b)
{
    return (a + b - 1) / b;
}

} // namespace sf
