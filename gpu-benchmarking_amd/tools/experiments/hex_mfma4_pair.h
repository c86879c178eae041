// hex_mfma4_pair.h -- EXPERIMENT (round 3, negative result; not part of libsumfact.so): csrc/bwdtrans_hmfma4.h's 3D kernel
// with two wavefronts per element.  Measured by the per-order tuner (sf_tune_hex12..16, 131 072 elements, mean GDOF/s;
// profiles/r03/tune_hex1[2-6]_mfma_4x4x4.log): 291 / 277 / 223 / 247 / 283 at nq 12 / 13 / 14 / 15 / 16 against 327 / 316 /
// 285 / 278 / 311-319 of the one-wave kernel: five workgroup barriers per element cost more than the second wave hides.
#pragma once

#include "../../csrc/wave_launch.h"

namespace sf
{

// ================================================================================================
// The same kernel with TWO wavefronts per element.  Measured (profiles/r03/hex_mfma4_phase_stamps.log): with one wave per
// element the waves of nq 14..16 -- four or five per CU, the LDS image of an element is 22-32 KB -- spend 20-23 k clocks
// per element in sequence (staging 1.3 k, sweeps 1 + 2 10.7 k, sweep 3 5.2 k, output image and flush 3.5 k, plus launch
// and load latency), nothing overlaps the memory phases of one wave but the other three or four waves, and the CU
// finishes an element every 5.9 k clocks where HBM would allow 5.0 k.  Two waves per element halve every phase and
// double the waves per CU at the same LDS footprint; the 4x4x4 products leave the matrix pipe the room for it (0.59 of
// the memory time; the 16x16x4 kernel with this split had none: tools/experiments/hex_mfma2.h).
// A workgroup = one element = two waves: the slices of sweeps 1 + 2 by parity of r, the pos groups of sweep 3 by parity,
// staging words and the output stream by lane; five workgroup barriers.  Sums and their order are the one-wave kernel's.
// ================================================================================================
template <int NQ, int WV>
__device__ __forceinline__ void hex_mfma4_pair_body(double *img, const double *__restrict__ b0, const double *__restrict__ b1,
                                                    const double *__restrict__ b2, double *__restrict__ oe, int tid)
{
    using G = HexMfma4Geom<NQ>;
    constexpr int NM = G::NM, NQ2 = G::NQ2, TP = G::TP, TI = G::TI, CG = G::CG, S = G::S, W2S = G::W2S;
    const int lane = tid & (kWave - 1);
    const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
    constexpr int NR = (NM - WV + 1) / 2; // this wave's slices r = WV, WV + 2, ...
    constexpr int NC = (CG - WV + 1) / 2; // this wave's pos groups

    double opB0[TP], opB1[TI][TP], opB2[TI][TP];
#pragma unroll
    for (int kp = 0; kp < TP; ++kp)
    {
        const int p = 4 * kp + hi, i = 4 * blk + lo;
        opB0[kp]    = (p < NM && i < NQ) ? b0[p * NQ + i] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int ks = 0; ks < TP; ++ks)
        {
            const int kk = 4 * ks + hi, o = 4 * t + lo;
            opB1[t][ks]  = (kk < NM && o < NQ) ? b1[kk * NQ + o] : 0.0;
            opB2[t][ks]  = (kk < NM && o < NQ) ? b2[kk * NQ + o] : 0.0;
        }
    int aoff[TP][TP];
#pragma unroll
    for (int tq = 0; tq < TP; ++tq)
#pragma unroll
        for (int kp = 0; kp < TP; ++kp)
        {
            const int q = 4 * tq + lo, p = 4 * kp + hi;
            aoff[tq][kp] = (q < NM ? q : NM - 1) * S + (p < NM ? p : NM - 1);
        }
    __syncthreads(); // the element is staged (both waves)

    // ---- sweeps 1 and 2 on the slices of this wave's parity ------------------------------------------------------------
    double w2[NR][TI];
    double av[2][TP][TP];
    auto gather = [&](double (&dst)[TP][TP], int r) {
        const double *slice = img + r * NM * S;
#pragma unroll
        for (int tq = 0; tq < TP; ++tq)
#pragma unroll
            for (int kp = 0; kp < TP; ++kp)
                dst[tq][kp] = slice[aoff[tq][kp]];
    };
    gather(av[0], WV);
#pragma unroll
    for (int k = 0; k < NR; ++k)
    {
        if (k + 1 < NR)
            gather(av[(k + 1) & 1], 2 * (k + 1) + WV);
        double w1[TP];
#pragma unroll
        for (int tq = 0; tq < TP; ++tq)
            w1[tq] = 0.0;
#pragma unroll
        for (int kp = 0; kp < TP; ++kp)
#pragma unroll
            for (int tq = 0; tq < TP; ++tq)
                w1[tq] = mfma4(av[k & 1][tq][kp], opB0[kp], w1[tq]);
#pragma unroll
        for (int tj = 0; tj < TI; ++tj)
            w2[k][tj] = 0.0;
#pragma unroll
        for (int tq = 0; tq < TP; ++tq)
#pragma unroll
            for (int tj = 0; tj < TI; ++tj)
                w2[k][tj] = mfma4(opB1[tj][tq], w1[tq], w2[k][tj]);
    }
    __syncthreads(); // both waves have gathered all of the input image: it becomes the W2 image [r][pos = j*NQ + i]
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int tj = 0; tj < TI; ++tj)
        {
            const int j = 4 * tj + hi, i = 4 * blk + lo;
            if (j < NQ && i < NQ)
                img[(2 * k + WV) * W2S + j * NQ + i] = w2[k][tj];
        }
    __syncthreads();
    // ---- sweep 3 on the pos groups of this wave's parity ----------------------------------------------------------------
    double o[NC][TI];
    double bv[2][TP];
    int rrow[TP];
#pragma unroll
    for (int kr = 0; kr < TP; ++kr)
    {
        const int r = 4 * kr + hi;
        rrow[kr]    = (r < NM ? r : NM - 1) * W2S;
    }
    auto gather3 = [&](double (&dst)[TP], int cg) {
        int pos = 16 * cg + 4 * blk + lo;
        if ((cg + 1) * 16 > NQ2)
            pos = pos < NQ2 ? pos : NQ2 - 1;
#pragma unroll
        for (int kr = 0; kr < TP; ++kr)
            dst[kr] = img[rrow[kr] + pos];
    };
    gather3(bv[0], WV);
#pragma unroll
    for (int k = 0; k < NC; ++k)
    {
        if (k + 1 < NC)
            gather3(bv[(k + 1) & 1], 2 * (k + 1) + WV);
#pragma unroll
        for (int tk = 0; tk < TI; ++tk)
            o[k][tk] = 0.0;
#pragma unroll
        for (int kr = 0; kr < TP; ++kr)
#pragma unroll
            for (int tk = 0; tk < TI; ++tk)
                o[k][tk] = mfma4(opB2[tk][kr], bv[k & 1][kr], o[k][tk]);
    }
    __syncthreads(); // all W2 gathers done: the image becomes the output in final layout
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int tk = 0; tk < TI; ++tk)
        {
            const int kk = 4 * tk + hi, pos = 16 * (2 * k + WV) + 4 * blk + lo;
            if (kk < NQ && pos < NQ2)
                img[kk * NQ2 + pos] = o[k][tk];
        }
    __syncthreads();
    // ---- flat stream out: whole 16-byte words on the 128-byte line grid, 128 lanes ----------------------------------------
    {
        const int a0      = line_offset<double>(oe);
        double2_t *grid   = reinterpret_cast<double2_t *>(oe - a0);
        constexpr int NST = cdiv(G::NQT + 15, 2 * 2 * kWave);
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

template <int NQ> constexpr size_t hex_mfma4_pair_lds_bytes()
{
    return sizeof(double) * (size_t)HexMfma4Geom<NQ>::SLAB;
}

template <int NQ, int MINW, int XG = 0>
__global__ __launch_bounds__(2 * kWave, MINW) void hex_mfma4_pair_kernel(
    const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ b2,
    const double *__restrict__ in, double *__restrict__ out, uint64_t nelmt)
{
    using G = HexMfma4Geom<NQ>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_hexmfma4p[];
    double *img   = reinterpret_cast<double *>(lds_raw_hexmfma4p);
    const int tid = threadIdx.x;
    const int wv  = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t e = logical_block<XG>(); // one element per workgroup
    if (e >= nelmt)
        return;
    // the element's input on its 16-byte word grid, 128 lanes; staged flat
    {
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
#pragma unroll
        for (int k = 0; k < NLD2; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h)
            {
                const int f = 2 * (k * 2 * kWave + tid) - a0 + h;
                if (f >= 0 && f < G::NMT)
                    img[f] = st[k][h];
            }
    }
    double *oe = out + e * (uint64_t)G::NQT;
    if (wv == 0)
        hex_mfma4_pair_body<NQ, 0>(img, b0, b1, b2, oe, tid);
    else
        hex_mfma4_pair_body<NQ, 1>(img, b0, b1, b2, oe, tid);
}

// two waves per element (hex_mfma4_pair_kernel): one workgroup per element, the grid covers the batch
template <int NQ, int MINW, int XG = 0> inline int launch_hex_mfma4_pair(const HexArgs &a, hipStream_t s)
{
    auto kern            = hex_mfma4_pair_kernel<NQ, MINW, XG>;
    constexpr size_t lds = hex_mfma4_pair_lds_bytes<NQ>();
    static_assert(lds <= 160 * 1024, "LDS image exceeds 160 KiB");
    if (a.nelmt == 0)
        return SF_OK;
    if (a.nelmt > 0x7fffffffull)
        return SF_EINVAL;
    static OccCache cache = {};
    (void)resident_blocks(kern, 2 * kWave, lds, cache); // raises the dynamic LDS limit once per device
    kern<<<(unsigned)a.nelmt, 2 * kWave, lds, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

} // namespace sf
