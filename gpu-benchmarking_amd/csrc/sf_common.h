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

template <int A, int B> struct CMax
{
    static constexpr int value = A > B ? A : B;
};

constexpr int cdiv(int a, int b)
{
    return (a + b - 1) / b;
}

} // namespace sf
