// bwdtrans_mfma.h -- 2D quad BwdTrans on the matrix cores (v_mfma_f64_16x16x4_f64) for HIGH order.
//
// Where the crossover is: per element the 2D path moves 8*(nm^2+nq^2) bytes and needs
// 2*(nq*nm^2 + nq^2*nm) flops; at nq = 32 that is 7.9 flop/B, and the lane-owns-a-pencil VALU kernel
// (bwdtrans_wave.h) becomes issue-bound (measured 1.9 TB/s, profiles/r01/tune_quad32.log) while every
// order <= 16 and every 3D order <= 10 stays HBM-bound.  Replaces the same reference kernel,
// BwdTransQuadKernel_QP_1D (benchmark04/benchmark04.cc:353-426); the reference's own best at nq = 32
// is its cuBLAS column (benchmark04/nq32x32.log:46).
//
// One wavefront = one element at a time, two chained GEMMs on 16x16x4 f64 MFMA tiles:
//   step 1   W[q][i]   = sum_p In[q][p]   * B0[p][i]      A = In tile (from LDS), B = B0 (registers)
//   step 2   Out[j][i] = sum_q B1^T[j][q] * W[q][i]       A = B1^T (registers),   B = W
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md s3): A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15]; D: lane l, register r holds D[(l>>4) + 4r][l&15].
// Step 1's D register r of tile (tm, tn) is therefore W[q = 16tm + 4r + g][i = 16tn + a] on lane
// (g = l>>4, a = l&15) -- exactly the B operand of step 2 at k-step ks = 4tm + r.  The intermediate
// never leaves the accumulator registers: no LDS round trip, no lane movement between the sweeps.
// Both bases live in registers for the whole kernel (zero-padded to the tile grid); the only LDS
// traffic is the input image (row stride padded to S = 2 mod 4 doubles: conflict-free ds_read_b64
// for the A-operand gather).  Padding rows/columns are fed clamped (finite) data times a zero basis.
#pragma once

#include "bwdtrans_wave.h"

namespace sf
{

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NQ, int EC> struct MfmaGeom
{
    static constexpr int NM  = NQ - 1;
    static constexpr int NMT = NM * NM, NQT = NQ * NQ;
    static constexpr int MT1 = cdiv(NM, 16); // q tiles of step 1
    static constexpr int NT  = cdiv(NQ, 16); // i tiles
    static constexpr int KS1 = cdiv(NM, 4);  // p steps
    static constexpr int MT2 = cdiv(NQ, 16); // j tiles
    static constexpr int KS2 = 4 * MT1;      // q steps = rows of step 1's D
    static constexpr int S   = NM + ((6 - NM % 4) % 4); // row stride, S % 4 == 2
    static constexpr int IN_DBL = EC * NMT;
    static constexpr bool VEC2  = (IN_DBL % 2) == 0;
    static constexpr int NLD    = VEC2 ? cdiv(IN_DBL / 2, kWave) : cdiv(IN_DBL, kWave);
    // per-element LDS region: the padded input image; with OUTL it is reused for the element's output
    // image once step 1 has consumed the input, so it must also hold nq^2 doubles
    static constexpr int ESTRIDE = ((NM * S > NQT ? NM * S : NQT) + 1) & ~1;
    static constexpr int SLAB    = EC * ESTRIDE; // doubles per wave
    static_assert(S % 4 == 2 && S >= NM, "row stride");
};

template <int NQ, int EC, int WPB> constexpr size_t mfma_lds_bytes()
{
    return sizeof(double) * (size_t)WPB * MfmaGeom<NQ, EC>::SLAB;
}

// staging registers -> LDS, element e row q at e*ESTRIDE + q*S
template <class G>
__device__ __forceinline__ void mfma_stage(const double2_t (&st)[G::NLD], double *slab, int lane)
{
#pragma unroll
    for (int k = 0; k < G::NLD; ++k)
    {
        const int v = k * kWave + lane;
        if constexpr (G::VEC2)
        {
            if ((k + 1) * kWave <= G::IN_DBL / 2 || v < G::IN_DBL / 2)
            {
                const int f0 = 2 * v, f1 = 2 * v + 1;
                const int r0 = f0 / G::NM, r1 = f1 / G::NM; // flat row index (e*NM + q)
                const int e0 = r0 / G::NM, e1 = r1 / G::NM;
                slab[e0 * G::ESTRIDE + (r0 - e0 * G::NM) * G::S + (f0 - r0 * G::NM)] = st[k].x;
                slab[e1 * G::ESTRIDE + (r1 - e1 * G::NM) * G::S + (f1 - r1 * G::NM)] = st[k].y;
            }
        }
        else
        {
            if ((k + 1) * kWave <= G::IN_DBL || v < G::IN_DBL)
            {
                const int r0 = v / G::NM, e0 = r0 / G::NM;
                slab[e0 * G::ESTRIDE + (r0 - e0 * G::NM) * G::S + (v - r0 * G::NM)] = st[k].x;
            }
        }
    }
}

template <int NQ, int EC, int WPB, int MINW, int KMAP, bool OUTL = false>
__global__ __launch_bounds__(kWave *WPB, MINW) void quad_mfma_kernel(
    const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ in,
    double *__restrict__ out, uint64_t nelmt)
{
    using G          = MfmaGeom<NQ, EC>;
    constexpr int NM = G::NM;

    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = lane & 15, g = lane >> 4;
    double *slab = lds + wib * G::SLAB;

    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const ChunkIter it    = chunk_iter<KMAP, WPB>(nchunk, wib);
    if (it.count == 0)
        return;

    // bases as MFMA operands, zero outside nm x nq
    double opB0[G::KS1][G::NT], opB1[G::KS2][G::MT2];
#pragma unroll
    for (int ks = 0; ks < G::KS1; ++ks)
#pragma unroll
        for (int tn = 0; tn < G::NT; ++tn)
        {
            const int p = ks * 4 + g, i = tn * 16 + a;
            opB0[ks][tn] = (p < NM && i < NQ) ? b0[p * NQ + i] : 0.0;
        }
#pragma unroll
    for (int ks = 0; ks < G::KS2; ++ks)
#pragma unroll
        for (int tm = 0; tm < G::MT2; ++tm)
        {
            const int q = ks * 4 + g, j = tm * 16 + a;
            opB1[ks][tm] = (q < NM && j < NQ) ? b1[q * NQ + j] : 0.0;
        }

    // A-operand gather offsets (clamped into the element: padding meets a zero basis entry)
    int arow[G::MT1];
#pragma unroll
    for (int tm = 0; tm < G::MT1; ++tm)
    {
        const int q = tm * 16 + a;
        arow[tm]    = (q < NM ? q : NM - 1) * G::S;
    }

    using GW = WaveGeom<NQ, EC, 2>; // chunk_load only needs IN_DBL / NLD / NMT, identical here
    static_assert(GW::NLD == G::NLD && GW::IN_DBL == G::IN_DBL, "geometry mismatch");
    double2_t st[G::NLD];
    chunk_fetch<GW, EC>(st, in, it.first, nelmt, lane);

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        const uint64_t left = nelmt - c * EC;
        const int evalid    = left >= EC ? EC : (int)left;

        mfma_stage<G>(st, slab, lane);
        wave_lds_fence();
        if (n + 1 < it.count)
            chunk_fetch<GW, EC>(st, in, c + it.step, nelmt, lane);

#pragma unroll 1
        for (int e = 0; e < evalid; ++e)
        {
            double *img = slab + e * G::ESTRIDE;
            // ---- step 1: W = In * B0 ------------------------------------------------------------
            double4_t w[G::MT1][G::NT];
#pragma unroll
            for (int tm = 0; tm < G::MT1; ++tm)
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                    w[tm][tn] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < G::KS1; ++ks)
            {
                const int p  = ks * 4 + g;
                const int pc = p < NM ? p : NM - 1;
#pragma unroll
                for (int tm = 0; tm < G::MT1; ++tm)
                {
                    const double aop = img[arow[tm] + pc];
#pragma unroll
                    for (int tn = 0; tn < G::NT; ++tn)
                        w[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, opB0[ks][tn], w[tm][tn],
                                                                         0, 0, 0);
                }
            }
            // ---- step 2: Out = B1^T * W  (W straight from step 1's accumulators) ------------------
            double4_t o[G::MT2][G::NT];
#pragma unroll
            for (int tm = 0; tm < G::MT2; ++tm)
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                    o[tm][tn] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < G::KS2; ++ks)
            {
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                {
                    const double bop = w[ks / 4][tn][ks % 4];
#pragma unroll
                    for (int tm = 0; tm < G::MT2; ++tm)
                        o[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(opB1[ks][tm], bop, o[tm][tn],
                                                                         0, 0, 0);
                }
            }
            // ---- store: register r of tile (tm, tn) is Out[j = 16tm + g + 4r][i = 16tn + a] -------
            double *oe = out + (c * EC + e) * (uint64_t)G::NQT;
            if constexpr (OUTL)
            {
                // the element's input image is dead (step 1 has read it): assemble the output there
                // and emit it as one flat 16-B-per-lane stream
                wave_lds_fence();
#pragma unroll
                for (int tm = 0; tm < G::MT2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < G::NT; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                            const int j = tm * 16 + g + 4 * r, i = tn * 16 + a;
                            if (j < NQ && i < NQ)
                                img[j * NQ + i] = o[tm][tn][r];
                        }
                wave_lds_fence();
                constexpr int NST = cdiv(G::NQT / 2, kWave);
                // element outputs are 16-B aligned when nq^2 is even or the element index is even
                const bool al16 = ((G::NQT & 1) == 0) || (((c * EC + e) & 1) == 0);
                if (al16)
                {
                    double2_t *oe2 = reinterpret_cast<double2_t *>(oe);
#pragma unroll
                    for (int k = 0; k < NST; ++k)
                    {
                        const int v = k * kWave + lane;
                        if (v < G::NQT / 2)
                            __builtin_nontemporal_store(*reinterpret_cast<const double2_t *>(img + 2 * v),
                                                        oe2 + v);
                    }
                    if ((G::NQT & 1) && lane == 0)
                        oe[G::NQT - 1] = img[G::NQT - 1];
                }
                else
                {
                    for (int v = lane; v < G::NQT; v += kWave)
                        oe[v] = img[v];
                }
                wave_lds_fence();
            }
            else
            {
#pragma unroll
                for (int tm = 0; tm < G::MT2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < G::NT; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                            const int j = tm * 16 + g + 4 * r, i = tn * 16 + a;
                            if (j < NQ && i < NQ)
                                __builtin_nontemporal_store(o[tm][tn][r], oe + j * NQ + i);
                        }
            }
        }
        wave_lds_fence(); // slab is rewritten by the next chunk's staging
    }
}

} // namespace sf
