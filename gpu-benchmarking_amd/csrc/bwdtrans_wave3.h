// bwdtrans_wave3.h -- the flagship wave-per-chunk kernel of bwdtrans_wave.h for ANISOTROPIC compile-time extents
// (nq0, nq1, nq2): same structure -- one wavefront per chunk of EC elements, flat 16-byte non-temporal loads / stores,
// lane owns a pencil, the three images rewritten in place in one LDS slab, basis rows as SGPR operands, XCD runs --
// with every extent taken per direction.  The reference kernels take the three extents at run time
// (benchmark05/benchmark05.cc:291-297); the shapes instantiated here are the table in bwdtrans_rt.hip, every other
// anisotropic shape runs the run-time-extent kernel of bwdtrans_rt.h.
#pragma once

#include "bwdtrans_wave.h"

namespace sf
{

// geometry with the member names chunk_load / chunk_stage / chunk_flush expect (NM = the input pencil length nm0)
template <int NQ0, int NQ1, int NQ2, int EC, typename T = double> struct WaveGeom3
{
    using Scalar = T;
    using Vec    = typename VecOf<T>::type;
    static constexpr int VW  = VecOf<T>::W;
    static constexpr int NM0 = NQ0 - 1, NM1 = NQ1 - 1, NM2 = NQ2 - 1;
    static constexpr int NM  = NM0;
    static constexpr int NMT = NM0 * NM1 * NM2, NQT = NQ0 * NQ1 * NQ2;
    static constexpr int IN_STRIDE = NM0 | 1; // pencils (e,r,q) of nm0 values
    static constexpr int S1 = NM1 | 1;        // pencils (e,i,r) of nm1 values
    static constexpr int S2 = NM2 | 1;        // pencils (e,j,i) of nm2 values
    static constexpr int IN_DBL    = EC * NMT;
    static constexpr bool VEC2     = (IN_DBL % VW) == 0;
    static constexpr int P0 = EC * NM2 * NM1, P1 = EC * NQ0 * NM2, P2 = EC * NQ1 * NQ0;
    static constexpr int PASS0 = cdiv(P0, kWave), PASS1 = cdiv(P1, kWave), PASS2 = cdiv(P2, kWave);
    static constexpr int SLAB0 = CMax<CMax<P0 * IN_STRIDE, P1 * S1>::value, P2 * S2>::value;
    static constexpr int OUT_DBL  = EC * NQT;
    static constexpr int SLAB_OUT = (CMax<SLAB0, OUT_DBL>::value + VW - 1) / VW * VW;
    static constexpr int NLD      = VEC2 ? cdiv(IN_DBL / VW, kWave) : cdiv(IN_DBL, kWave);
    static constexpr bool ALIGN_OK = VEC2 && cdiv(IN_DBL / VW + 7, kWave) == NLD;
};

template <int NQ0, int NQ1, int NQ2, int EC, int WPB, typename T = double> constexpr size_t wave3_lds_bytes()
{
    return sizeof(T) * (size_t)WPB * WaveGeom3<NQ0, NQ1, NQ2, EC, T>::SLAB_OUT;
}

template <int NQ0, int NQ1, int NQ2, int EC, int WPB, int BMODE, int MINW, int XG = 64, typename T = double>
__global__ __launch_bounds__(kWave *WPB, MINW) void hex_wave3_kernel(
    const T *__restrict__ b0, const T *__restrict__ b1, const T *__restrict__ b2,
    const T *__restrict__ in, T *__restrict__ out, uint64_t nelmt)
{
    using G = WaveGeom3<NQ0, NQ1, NQ2, EC, T>;
    constexpr int NM0 = G::NM0, NM1 = G::NM1, NM2 = G::NM2, S1 = G::S1, S2 = G::S2;
    constexpr int NM12 = NM1 * NM2, NQ0NM2 = NQ0 * NM2, NQ01 = NQ0 * NQ1;
    static_assert(BMODE != BASIS_LDS, "basis rows are scalar operands");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw3[];
    T *lds = reinterpret_cast<T *>(lds_raw3);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T *slab = lds + wib * G::SLAB_OUT;

    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const uint64_t c      = logical_block<XG>() * WPB + wib; // one chunk per short-lived wave
    if (c >= nchunk)
        return;
    const uint64_t left = nelmt - c * EC;
    const int evalid    = left >= EC ? EC : (int)left;

    typename G::Vec st[G::NLD];
    chunk_fetch<G, EC, true, false>(st, in, c, nelmt, lane);
    chunk_stage<G, false>(st, slab, lane, G::VEC2 ? 0 : line_offset<T>(in + c * G::IN_DBL));
    wave_lds_fence();

    // ---- direction 0: w1[(e,i,r)][q] = sum_p in[(e,r,q)][p] * B0[p][i] --------------------------------------------
    {
        T u[G::PASS0][NM0], acc[G::PASS0][NQ0];
        read_pencils<NM0, G::PASS0, G::P0, G::IN_STRIDE>(u, slab, lane);
        contract<NM0, NQ0, G::PASS0, BMODE>(u, acc, b0);
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < G::PASS0; ++s)
        {
            const int t = s * kWave + lane;
            if ((s + 1) * kWave <= G::P0 || t < G::P0)
            {
                const int e = t / NM12, rq = t - e * NM12, r = rq / NM1, q = rq - r * NM1;
                T *dst = slab + (e * NQ0NM2 + r) * S1 + q;
#pragma unroll
                for (int i = 0; i < NQ0; ++i)
                    dst[i * NM2 * S1] = acc[s][i];
            }
        }
        wave_lds_fence();
    }
    // ---- direction 1: w2[(e,j,i)][r] = sum_q w1[(e,i,r)][q] * B1[q][j] --------------------------------------------
    {
        T u[G::PASS1][NM1], acc[G::PASS1][NQ1];
        read_pencils<NM1, G::PASS1, G::P1, S1>(u, slab, lane);
        contract<NM1, NQ1, G::PASS1, BMODE>(u, acc, b1);
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < G::PASS1; ++s)
        {
            const int t = s * kWave + lane;
            if ((s + 1) * kWave <= G::P1 || t < G::P1)
            {
                const int e = t / NQ0NM2, ir = t - e * NQ0NM2, i = ir / NM2, r = ir - i * NM2;
                T *dst = slab + (e * NQ01 + i) * S2 + r;
#pragma unroll
                for (int j = 0; j < NQ1; ++j)
                    dst[j * NQ0 * S2] = acc[s][j];
            }
        }
        wave_lds_fence();
    }
    // ---- direction 2: out[e][k][(j,i)] = sum_r w2[(e,j,i)][r] * B2[r][k] ------------------------------------------
    {
        T u[G::PASS2][NM2], acc[G::PASS2][NQ2];
        read_pencils<NM2, G::PASS2, G::P2, S2>(u, slab, lane);
        contract<NM2, NQ2, G::PASS2, BMODE>(u, acc, b2);
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < G::PASS2; ++s)
        {
            const int t = s * kWave + lane;
            if ((s + 1) * kWave <= G::P2 || t < G::P2)
            {
                const int e = t / NQ01, pl = t - e * NQ01;
                T *dst = slab + e * G::NQT + pl;
#pragma unroll
                for (int k = 0; k < NQ2; ++k)
                    dst[k * NQ01] = acc[s][k];
            }
        }
        wave_lds_fence();
        chunk_flush<G, true, false>(slab, out + c * (uint64_t)(EC * G::NQT), evalid * G::NQT, lane);
    }
}

} // namespace sf
