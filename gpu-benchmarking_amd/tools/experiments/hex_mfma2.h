// hex_mfma2.h -- EXPERIMENT (round 3, negative result; not part of libsumfact.so): the 3D matrix-core kernel of
// csrc/bwdtrans_mfma.h with two wavefronts per element.  Measured by the per-order tuner (sf_tune_hex14/15/16), log in
// profiles/r03/tune_hex_two_waves.log: 285 / 285-293 / 297-306 GDOF/s at nq 14 / 15 / 16 against 285-295 / 293-297 /
// 296-312 of the one-wave kernel -- no gain, both sit at 5.2-5.5 TB/s.
#pragma once

#include "../../csrc/wave_launch.h"

namespace sf
{

// ================================================================================================
// The same three chained GEMMs with TWO wavefronts per element (nq 14..16, where one element's LDS image is 22-32 KB and
// leaves five to seven one-wave workgroups per CU: the matrix pipe is busy 0.46-0.55 of the time there,
// profiles/r03/crossover/crossover.md).  A workgroup is two waves and one element: the waves split the staging words, the
// sweep-1 tiles / sweep-2 accumulators by parity of r (q is padded to 16 here, so tile t is exactly r = t), the sweep-3
// column blocks by parity and the output stream, and meet at four workgroup barriers.  W2 rows stay in registers until
// both waves have gathered all of the input image.  Sums and their order are the one-wave kernel's: bit-identical.
// ================================================================================================
template <int NQ, int WV>
__device__ __forceinline__ void hex_mfma2_body(double *img, const double (&opB0)[HexMfmaGeom<NQ, 1>::KS1],
                                               const double (&opB1)[HexMfmaGeom<NQ, 1>::KS2],
                                               const double (&opB2)[HexMfmaGeom<NQ, 1>::KS3], double *__restrict__ oe, int tid)
{
    using G          = HexMfmaGeom<NQ, 1>;
    constexpr int NM = G::NM, NQ2 = G::NQ2;
    static_assert(G::QP == 16 && G::MT1 == NM, "one 16-row tile per r");
    const int lane = tid & (kWave - 1), a = lane & 15, g = lane >> 4;
    constexpr int NR = (NM - WV + 1) / 2; // this wave's r = WV, WV + 2, ...
    // ---- sweeps 1 and 2 on the tiles of this wave's parity ---------------------------------------------------------
    double4_t w2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k)
    {
        const int r = 2 * k + WV;
        int q       = a < NM ? a : NM - 1; // row R = 16 r + a -> (r, q = a), clamped into the element
        const int arow = (r * NM + q) * G::S;
        double4_t w1t  = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < G::KS1; ++ks)
        {
            const int p  = ks * 4 + g;
            const int pc = p < NM ? p : NM - 1;
            w1t = __builtin_amdgcn_mfma_f64_16x16x4f64(img[arow + pc], opB0[ks], w1t, 0, 0, 0);
        }
        w2[k] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int qs = 0; qs < G::KS2; ++qs)
            w2[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(opB1[qs], w1t[qs], w2[k], 0, 0, 0);
    }
    __syncthreads(); // both waves have gathered all of the input image: it becomes the W2 image [r][pos = j*NQ + i]
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
        {
            const int j = g + 4 * r4;
            if (j < NQ && a < NQ)
                img[(2 * k + WV) * G::W2S + j * NQ + a] = w2[k][r4];
        }
    __syncthreads();
    // ---- sweep 3 on the column blocks of this wave's parity ---------------------------------------------------------
    constexpr int NC = (G::CB - WV + 1) / 2;
    double4_t o[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k)
        o[k] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rs = 0; rs < G::KS3; ++rs)
    {
        const int r  = rs * 4 + g;
        const int rc = r < NM ? r : NM - 1;
#pragma unroll
        for (int k = 0; k < NC; ++k)
        {
            const int cb = 2 * k + WV;
            int pos      = cb * 16 + a;
            if ((cb + 1) * 16 > NQ2)
                pos = pos < NQ2 ? pos : NQ2 - 1;
            o[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(opB2[rs], img[rc * G::W2S + pos], o[k], 0, 0, 0);
        }
    }
    __syncthreads(); // all W2 gathers done: the image becomes the output in final layout
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
        {
            const int kk = g + 4 * r4, pos = (2 * k + WV) * 16 + a;
            if (kk < NQ && pos < NQ2)
                img[kk * NQ2 + pos] = o[k][r4];
        }
    __syncthreads();
    // ---- flat stream out: whole 16-byte words on the 128-byte line grid, 128 lanes ------------------------------------
    {
        const int a0        = line_offset<double>(oe);
        double2_t *grid     = reinterpret_cast<double2_t *>(oe - a0);
        constexpr int NST   = cdiv(G::NQT + 15, 2 * 2 * kWave);
#pragma unroll
        for (int k = 0; k < NST; ++k)
        {
            const int gv = k * 2 * kWave + tid;
            const int d0 = 2 * gv - a0;
            if (d0 >= 0 && d0 + 1 < G::NQT)
            {
                const double2_t x = {img[d0], img[d0 + 1]};
                __builtin_nontemporal_store(x, grid + gv);
            }
            else
            {
                if (d0 >= 0 && d0 < G::NQT)
                    oe[d0] = img[d0];
                if (d0 + 1 >= 0 && d0 + 1 < G::NQT)
                    oe[d0 + 1] = img[d0 + 1];
            }
        }
    }
}

template <int NQ> constexpr size_t hex_mfma2_lds_bytes()
{
    return sizeof(double) * (size_t)HexMfmaGeom<NQ, 1>::ESTRIDE;
}

template <int NQ, int MINW, int XG = 0>
__global__ __launch_bounds__(2 * kWave, MINW) void hex_mfma2_kernel(
    const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ b2,
    const double *__restrict__ in, double *__restrict__ out, uint64_t nelmt)
{
    using G          = HexMfmaGeom<NQ, 1>;
    constexpr int NM = G::NM;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_hexmfma2[];
    double *img   = reinterpret_cast<double *>(lds_raw_hexmfma2);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), a = lane & 15, g = lane >> 4;
    const int wv  = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t e = logical_block<XG>(); // one element per workgroup
    if (e >= nelmt)
        return;
    // ---- the element's input on its 16-byte word grid, 128 lanes ------------------------------------------------------
    const double *src = in + e * (uint64_t)G::NMT;
    const int a0      = line_offset<double>(src);
    const double2_t *grid = reinterpret_cast<const double2_t *>(src - a0);
    constexpr int NLD2 = cdiv(G::NMT + 15, 2 * 2 * kWave);
    double2_t st[NLD2];
#pragma unroll
    for (int k = 0; k < NLD2; ++k)
    {
        const int gv = k * 2 * kWave + tid;
        const int d0 = 2 * gv - a0;
        double2_t x  = {0.0, 0.0};
        if (d0 >= 0 && d0 + 1 < G::NMT)
            x = __builtin_nontemporal_load(grid + gv);
        else
        {
            if (d0 >= 0 && d0 < G::NMT)
                x.x = src[d0];
            if (d0 + 1 >= 0 && d0 + 1 < G::NMT)
                x.y = src[d0 + 1];
        }
        st[k] = x;
    }
    // basis operands (zero outside nm x nq), while the loads are in flight
    double opB0[G::KS1], opB1[G::KS2], opB2[G::KS3];
#pragma unroll
    for (int ks = 0; ks < G::KS1; ++ks)
    {
        const int p = ks * 4 + g;
        opB0[ks]    = (p < NM && a < NQ) ? b0[p * NQ + a] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS2; ++ks)
    {
        const int q = ks * 4 + g;
        opB1[ks]    = (q < NM && a < NQ) ? b1[q * NQ + a] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS3; ++ks)
    {
        const int r = ks * 4 + g;
        opB2[ks]    = (r < NM && a < NQ) ? b2[r * NQ + a] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NLD2; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            const int f = 2 * (k * 2 * kWave + tid) - a0 + h;
            if (f >= 0 && f < G::NMT)
            {
                const int row                    = f / NM;
                img[row * G::S + (f - row * NM)] = st[k][h];
            }
        }
    __syncthreads();
    double *oe = out + e * (uint64_t)G::NQT;
    if (wv == 0)
        hex_mfma2_body<NQ, 0>(img, opB0, opB1, opB2, oe, tid);
    else
        hex_mfma2_body<NQ, 1>(img, opB0, opB1, opB2, oe, tid);
}

// two waves per element (hex_mfma2_kernel): one workgroup per element, grid covers the batch
template <int NQ, int MINW, int XG = 0> inline int launch_hex_mfma2(const HexArgs &a, hipStream_t s)
{
    auto kern            = hex_mfma2_kernel<NQ, MINW, XG>;
    constexpr size_t lds = hex_mfma2_lds_bytes<NQ>();
    static_assert(lds <= 160 * 1024, "LDS image exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    if (a.nelmt > 0x7fffffffull)
        return SF_EINVAL;
    static std::atomic<int> attr_set[kMaxDev] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev)
        dev = 0;
    if (lds > 48 * 1024 && attr_set[dev].load(std::memory_order_acquire) == 0)
    {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev].store(1, std::memory_order_release);
    }
    kern<<<(unsigned)a.nelmt, 2 * kWave, lds, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

} // namespace sf
