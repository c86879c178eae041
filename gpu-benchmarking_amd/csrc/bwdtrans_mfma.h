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

// T = float runs the same two chained GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32, the fp32 vector rate: 2x the fp64
// instruction).  Its A / B lane maps are the fp64 ones; its D map is the standard one, D[row = 4*(l>>4) + r][col = l&15]
// (fp64: row = (l>>4) + 4r).  Register r of a step-1 tile therefore holds the W rows q = 16tm + 4g + r on lane group
// g -- still directly a B operand of step 2 (lane group g supplies k index g), against an A operand that holds
// B1[16tm + 4g + r][j]: the chaining survives, the k-steps of step 2 just visit q in the order r, r+4, r+8, r+12.
// fp32 sums therefore run in that order inside a 16-row tile (tolerance 2e-5 in the tests; fp64 stays ascending).
template <typename T> struct MfmaOp;
template <> struct MfmaOp<double>
{
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // D register r of lane group g holds tile row ...; step-2 k-step `ks`, lane group g, contracts over q = ...
    static constexpr __device__ __host__ int drow(int g, int r)
    {
        return g + 4 * r;
    }
};
template <> struct MfmaOp<float>
{
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static constexpr __device__ __host__ int drow(int g, int r)
    {
        return 4 * g + r;
    }
};

template <int NQ, int EC, typename T = double> struct MfmaGeom
{
    static constexpr int VW  = 16 / (int)sizeof(T);
    static constexpr int NM  = NQ - 1;
    static constexpr int NMT = NM * NM, NQT = NQ * NQ;
    static constexpr int MT1 = cdiv(NM, 16); // q tiles of step 1
    static constexpr int NT  = cdiv(NQ, 16); // i tiles
    static constexpr int KS1 = cdiv(NM, 4);  // p steps
    static constexpr int MT2 = cdiv(NQ, 16); // j tiles
    // q steps of step 2: D registers of step 1 that hold a real q.  fp64: register r of tile tm holds rows 16tm + 4r + g
    // (consecutive groups of four rows: ceil(nm / 4) steps); fp32: rows 16tm + 4g + r (a register holds a real row iff
    // 16tm + r < nm: every register of a tile with at least four rows)
    static constexpr int REM  = NM - 16 * (MT1 - 1);
    static constexpr int KS2 = sizeof(T) == 8 ? cdiv(NM, 4) : 4 * (MT1 - 1) + (REM < 4 ? REM : 4);
    static constexpr int S   = NM + ((6 - NM % 4) % 4); // row stride, S % 4 == 2
    static constexpr int IN_DBL = EC * NMT;
    static constexpr bool VEC2  = (IN_DBL % VW) == 0;
    static constexpr int NLD    = VEC2 ? cdiv(IN_DBL / VW, kWave) : cdiv(IN_DBL, kWave);
    // per-element LDS region: the padded input image; with OUTL the slab is reused for the chunk's output
    // image (element e at e*nq^2) once step 1 has consumed the input, so a region also holds nq^2 scalars
    static constexpr int ESTRIDE = ((NM * S > NQT ? NM * S : NQT) + VW - 1) / VW * VW;
    static constexpr int SLAB    = EC * ESTRIDE; // scalars per wave
    static_assert(S % 4 == 2 && S >= NM, "row stride");
};

template <int NQ, int EC, int WPB, typename T = double> constexpr size_t mfma_lds_bytes()
{
    return sizeof(T) * (size_t)WPB * MfmaGeom<NQ, EC, T>::SLAB;
}

// staging registers -> LDS, element e row q at e*ESTRIDE + q*S
template <class G, typename T>
__device__ __forceinline__ void mfma_stage(const typename VecOf<T>::type (&st)[G::NLD], T *slab, int lane, int sh)
{
    constexpr int VW = G::VW;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k)
    {
        const int v = k * kWave + lane;
        if constexpr (G::VEC2)
        {
            if ((k + 1) * kWave <= G::IN_DBL / VW || v < G::IN_DBL / VW)
            {
#pragma unroll
                for (int h = 0; h < VW; ++h)
                {
                    const int f = VW * v + h;
                    const int r = f / G::NM, e = r / G::NM; // flat row index (e*NM + q)
                    slab[e * G::ESTRIDE + (r - e * G::NM) * G::S + (f - r * G::NM)] = st[k][h];
                }
            }
        }
        else if (k < word_grid_regs<G::IN_DBL, T>())
        {
            // word-grid registers (chunk_load_any): word v holds scalars VW*v - sh + {0 .. VW-1}
#pragma unroll
            for (int h = 0; h < VW; ++h)
            {
                const int f = VW * v - sh + h;
                if (f >= 0 && f < G::IN_DBL)
                {
                    const int r0 = f / G::NM, e0 = r0 / G::NM;
                    slab[e0 * G::ESTRIDE + (r0 - e0 * G::NM) * G::S + (f - r0 * G::NM)] = st[k][h];
                }
            }
        }
    }
}

template <int NQ, int EC, int WPB, int MINW, int KMAP, bool OUTL = false, int XG = 0, typename T = double>
__global__ __launch_bounds__(kWave *WPB, MINW) void quad_mfma_kernel(
    const T *__restrict__ b0, const T *__restrict__ b1, const T *__restrict__ in,
    T *__restrict__ out, uint64_t nelmt)
{
    using G          = MfmaGeom<NQ, EC, T>;
    using Op         = MfmaOp<T>;
    using acc_t      = typename Op::acc_t;
    constexpr int NM = G::NM;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_mfma[];
    T *lds = reinterpret_cast<T *>(lds_raw_mfma);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = lane & 15, g = lane >> 4;
    T *slab = lds + wib * G::SLAB;

    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const ChunkIter it    = chunk_iter<KMAP, WPB, XG>(nchunk, wib);
    if (it.count == 0)
        return;

    // bases as MFMA operands, zero outside nm x nq
    T opB0[G::KS1][G::NT], opB1[G::KS2][G::MT2];
#pragma unroll
    for (int ks = 0; ks < G::KS1; ++ks)
#pragma unroll
        for (int tn = 0; tn < G::NT; ++tn)
        {
            const int p = ks * 4 + g, i = tn * 16 + a;
            opB0[ks][tn] = (p < NM && i < NQ) ? b0[p * NQ + i] : T(0);
        }
#pragma unroll
    for (int ks = 0; ks < G::KS2; ++ks)
#pragma unroll
        for (int tm = 0; tm < G::MT2; ++tm)
        {
            // the q this lane group contributes to q step ks = the row of step 1's D register ks % 4 of tile ks / 4
            const int q = 16 * (ks / 4) + Op::drow(g, ks % 4), j = tm * 16 + a;
            opB1[ks][tm] = (q < NM && j < NQ) ? b1[q * NQ + j] : T(0);
        }

    // A-operand gather offsets (clamped into the element: padding meets a zero basis entry)
    int arow[G::MT1];
#pragma unroll
    for (int tm = 0; tm < G::MT1; ++tm)
    {
        const int q = tm * 16 + a;
        arow[tm]    = (q < NM ? q : NM - 1) * G::S;
    }

    using GW = WaveGeom<NQ, EC, 2, T>; // chunk_load only needs IN_DBL / NLD / NMT, identical here
    static_assert(GW::NLD == G::NLD && GW::IN_DBL == G::IN_DBL, "geometry mismatch");
    typename GW::Vec st[G::NLD];
    chunk_fetch<GW, EC>(st, in, it.first, nelmt, lane);

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        const uint64_t left = nelmt - c * EC;
        const int evalid    = left >= EC ? EC : (int)left;

        mfma_stage<G, T>(st, slab, lane, G::VEC2 ? 0 : line_offset<T>(in + c * G::IN_DBL));
        wave_lds_fence();
        if (n + 1 < it.count)
            chunk_fetch<GW, EC>(st, in, c + it.step, nelmt, lane);

#pragma unroll 1
        for (int e = 0; e < evalid; ++e)
        {
            T *img = slab + e * G::ESTRIDE;
            // ---- step 1: W = In * B0 ------------------------------------------------------------
            acc_t w[G::MT1][G::NT];
#pragma unroll
            for (int tm = 0; tm < G::MT1; ++tm)
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                    w[tm][tn] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int ks = 0; ks < G::KS1; ++ks)
            {
                const int p  = ks * 4 + g;
                const int pc = p < NM ? p : NM - 1;
#pragma unroll
                for (int tm = 0; tm < G::MT1; ++tm)
                {
                    const T aop = img[arow[tm] + pc];
#pragma unroll
                    for (int tn = 0; tn < G::NT; ++tn)
                        w[tm][tn] = Op::mma(aop, opB0[ks][tn], w[tm][tn]);
                }
            }
            // ---- step 2: Out = B1^T * W  (W straight from step 1's accumulators) ------------------
            acc_t o[G::MT2][G::NT];
#pragma unroll
            for (int tm = 0; tm < G::MT2; ++tm)
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                    o[tm][tn] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int ks = 0; ks < G::KS2; ++ks)
            {
#pragma unroll
                for (int tn = 0; tn < G::NT; ++tn)
                {
                    const T bop = w[ks / 4][tn][ks % 4];
#pragma unroll
                    for (int tm = 0; tm < G::MT2; ++tm)
                        o[tm][tn] = Op::mma(opB1[ks][tm], bop, o[tm][tn]);
                }
            }
            // ---- store: register r of tile (tm, tn) is Out[j = 16tm + drow(g, r)][i = 16tn + a] ----
            T *oe = out + (c * EC + e) * (uint64_t)G::NQT;
            if constexpr (OUTL)
            {
                // the element's input image is dead (step 1 has read it): park the output in the slab at
                // e*nq^2 -- never beyond the start of element e+1's still-live input image, since
                // ESTRIDE >= nq^2 -- and let the whole chunk leave as one flat stream after the loop
                wave_lds_fence();
                T *oimg = slab + e * G::NQT;
#pragma unroll
                for (int tm = 0; tm < G::MT2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < G::NT; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                            const int j = tm * 16 + Op::drow(g, r), i = tn * 16 + a;
                            if (j < NQ && i < NQ)
                                oimg[j * NQ + i] = o[tm][tn][r];
                        }
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
                            const int j = tm * 16 + Op::drow(g, r), i = tn * 16 + a;
                            if (j < NQ && i < NQ)
                                __builtin_nontemporal_store(o[tm][tn][r], oe + j * NQ + i);
                        }
            }
        }
        if constexpr (OUTL)
        {
            // 16 B per lane, every wave-wide store on whole 128-byte lines (chunk_flush, bwdtrans_wave.h)
            wave_lds_fence();
            chunk_flush<GW, true, true>(slab, out + c * (uint64_t)(EC * G::NQT), evalid * G::NQT, lane);
        }
        wave_lds_fence(); // slab is rewritten by the next chunk's staging
        if (n + 1 < it.count)
            touch_staged(st); // counted wait for the next chunk here, not vmcnt(0) at the loop header
    }
}

// ================================================================================================
// 3D hex on the matrix cores (nq <= 16): three chained GEMMs per element, one wavefront per element.
//
//   sweep 1  W1[(r,q)][i] = sum_p In[(r,q)][p] * B0[p][i]     A = input rows gathered from LDS, B = B0 regs
//   sweep 2  W2_r[j][i]   = sum_q B1^T[j][q]   * W1[(r,q)][i] A = B1^T regs, B = sweep-1 accumulators
//   sweep 3  Out[k][pos]  = sum_r B2^T[k][r]   * W2[r][pos]   A = B2^T regs, B = W2 gathered from LDS
//
// Sweep 1 enumerates its rows as R = r*QP + q with the q extent padded to QP = 4*ceil(nm/4): a group of
// four D rows (register r4 of tile t, lanes g = 0..3) is then four consecutive q of ONE r, i.e. exactly
// one k-step of sweep 2's B operand -- W1 goes from sweep 1 to sweep 2 without leaving the registers
// (same trick as the 2D kernel).  Sweep 3 contracts r, which indexes different accumulators, so W2 makes
// one trip through the element's (dead) LDS image, stored [r][pos = j*nq + i] with a row stride = 16 mod 32
// doubles (conflict-free B gathers).  The result tile D3 (k on g + 4*reg, 16 consecutive pos on the lanes)
// is assembled in LDS in final layout and leaves as one flat 16-byte-per-lane stream.
// Padding (p, q, r, i, j, k beyond nm / nq) always meets a zero basis entry and clamped, finite data.
// ================================================================================================
template <int NQ, int EC, typename T = double> struct HexMfmaGeom
{
    static constexpr int VW  = 16 / (int)sizeof(T);
    static constexpr int NM  = NQ - 1;
    static constexpr int NMT = NM * NM * NM, NQ2 = NQ * NQ, NQT = NQ * NQ * NQ;
    // padded q extent.  fp64: a multiple of 4 (a D register holds four CONSECUTIVE rows).  fp32: 16 -- its D registers hold
    // rows 4g + reg, so a register is one k-step of sweep 2 only when a whole 16-row tile belongs to one r
    static constexpr int QP  = sizeof(T) == 8 ? (NM + 3) / 4 * 4 : 16;
    static constexpr int M1  = NM * QP;                 // rows of sweep 1
    static constexpr int MT1 = cdiv(M1, 16);
    static constexpr int KS1 = cdiv(NM, 4);             // p steps
    static constexpr int KS2 = QP / 4;                  // q steps
    static constexpr int KS3 = cdiv(NM, 4);             // r steps
    static constexpr int CB  = cdiv(NQ2, 16);           // column blocks of sweep 3
    static constexpr int S   = NM + ((6 - NM % 4) % 4); // input row stride, S % 4 == 2
    static constexpr int W2S = (NQ2 + 15) / 32 * 32 + 16; // W2 row stride, = 16 mod 32, >= NQ2
    static constexpr int IN_DBL = EC * NMT;
    static constexpr bool VEC2  = (IN_DBL % VW) == 0;
    static constexpr int NLD    = VEC2 ? cdiv(IN_DBL / VW, kWave) : cdiv(IN_DBL, kWave);
    static constexpr int E0      = NM * NM * S > NM * W2S ? NM * NM * S : NM * W2S;
    static constexpr int ESTRIDE = ((E0 > NQT ? E0 : NQT) + VW - 1) / VW * VW; // per-element LDS region
    static constexpr int SLAB    = EC * ESTRIDE;
    static_assert(NQ <= 16, "one 16-wide tile per direction");
    static_assert(S % 4 == 2 && W2S % 32 == 16 && W2S >= NQ2, "LDS strides");
    // The tile of sweep 1 after which W2[r] may be written to its LDS row [r*W2S, r*W2S + NQ2): its own last q group
    // must be in, and the row overlaps the input image (rows (r',q') at (r'*NM + q')*S), whose last overlapped row
    // must have been gathered -- sweep 1 gathers input row (r',q') in tile (r'*QP + q') / 16.
    static constexpr int w2_store_tile(int r)
    {
        const int own  = (r * QP + QP - 1) / 16;
        int fr         = (r * W2S + NQ2 - 1) / S; // last overlapped flat input row
        fr             = fr < NM * NM - 1 ? fr : NM * NM - 1;
        const int need = ((fr / NM) * QP + fr % NM) / 16;
        const int t    = own > need ? own : need;
        return t < MT1 - 1 ? t : MT1 - 1;
    }
};

template <int NQ, int EC, int WPB, typename T = double> constexpr size_t hex_mfma_lds_bytes()
{
    return sizeof(T) * (size_t)WPB * HexMfmaGeom<NQ, EC, T>::SLAB;
}

template <int NQ, int EC, int WPB, int MINW, int KMAP, int XG = 0, typename T = double>
__global__ __launch_bounds__(kWave *WPB, MINW) void hex_mfma_kernel(
    const T *__restrict__ b0, const T *__restrict__ b1, const T *__restrict__ b2,
    const T *__restrict__ in, T *__restrict__ out, uint64_t nelmt)
{
    using G          = HexMfmaGeom<NQ, EC, T>;
    using Op         = MfmaOp<T>;
    using acc_t      = typename Op::acc_t;
    using V          = typename VecOf<T>::type;
    constexpr int VW = G::VW;
    constexpr int NM = G::NM, QP = G::QP, NQ2 = G::NQ2;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_hexmfma[];
    T *lds = reinterpret_cast<T *>(lds_raw_hexmfma);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = lane & 15, g = lane >> 4;
    T *slab = lds + wib * G::SLAB;

    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const ChunkIter it    = chunk_iter<KMAP, WPB, XG>(nchunk, wib);
    if (it.count == 0)
        return;

    // basis operands (zero outside nm x nq)
    T opB0[G::KS1], opB1[G::KS2], opB2[G::KS3];
#pragma unroll
    for (int ks = 0; ks < G::KS1; ++ks)
    {
        const int p = ks * 4 + g;
        opB0[ks]    = (p < NM && a < NQ) ? b0[p * NQ + a] : T(0); // B[k = p][col = i = a]
    }
#pragma unroll
    for (int ks = 0; ks < G::KS2; ++ks)
    {
        const int q = Op::drow(g, ks); // the q lane group g contributes to q step ks: row of D register ks of a W1 tile
        opB1[ks]    = (q < NM && a < NQ) ? b1[q * NQ + a] : T(0); // A[row = j = a][k = q]
    }
#pragma unroll
    for (int ks = 0; ks < G::KS3; ++ks)
    {
        const int r = ks * 4 + g;
        opB2[ks]    = (r < NM && a < NQ) ? b2[r * NQ + a] : T(0); // A[row = k = a][k = r]
    }
    // sweep-1 A gather: row R = 16t + a -> (r, q), clamped into the element
    int arow[G::MT1];
#pragma unroll
    for (int t = 0; t < G::MT1; ++t)
    {
        const int R = t * 16 + a;
        int r = R / QP, q = R - r * QP;
        r       = r < NM ? r : NM - 1;
        q       = q < NM ? q : NM - 1;
        arow[t] = (r * NM + q) * G::S;
    }

    using GW = WaveGeom<NQ, EC, 3, T>;
    static_assert(GW::NLD == G::NLD && GW::IN_DBL == G::IN_DBL, "geometry mismatch");
    V st[G::NLD];
    // odd scalars per chunk (one element of an even order): the chunk base is only 8-byte aligned -> word-grid load
    auto fetch = [&](uint64_t cc) {
        if constexpr (G::VEC2)
            chunk_fetch<GW, EC>(st, in, cc, nelmt, lane);
        else
        {
            const uint64_t lft = nelmt - cc * EC;
            chunk_load_any<G::IN_DBL, G::NLD, T>(st, in + cc * G::IN_DBL, lane,
                                                 lft >= EC ? G::IN_DBL : (int)lft * G::NMT);
        }
    };
    fetch(it.first);

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        const uint64_t left = nelmt - c * EC;
        const int evalid    = left >= EC ? EC : (int)left;

        // staging registers -> LDS: element e, row (r,q) at e*ESTRIDE + (r*NM + q)*S
#pragma unroll
        for (int k = 0; k < G::NLD; ++k)
        {
            const int v = k * kWave + lane;
            if constexpr (G::VEC2)
            {
                if ((k + 1) * kWave <= G::IN_DBL / VW || v < G::IN_DBL / VW)
                {
#pragma unroll
                    for (int h = 0; h < VW; ++h)
                    {
                        const int f = VW * v + h, row = f / NM, e = row / (NM * NM);
                        slab[e * G::ESTRIDE + (row - e * NM * NM) * G::S + (f - row * NM)] = st[k][h];
                    }
                }
            }
            else if (k < word_grid_regs<G::IN_DBL, T>())
            {
                const int a0 = line_offset<T>(in + c * G::IN_DBL);
#pragma unroll
                for (int h = 0; h < VW; ++h)
                {
                    const int f = VW * v - a0 + h;
                    if (f >= 0 && f < G::IN_DBL)
                    {
                        const int row = f / NM, e = row / (NM * NM);
                        slab[e * G::ESTRIDE + (row - e * NM * NM) * G::S + (f - row * NM)] = st[k][h];
                    }
                }
            }
        }
        wave_lds_fence();
        if (n + 1 < it.count)
            fetch(c + it.step);

#pragma unroll 1
        for (int e = 0; e < evalid; ++e)
        {
            T *img = slab + e * G::ESTRIDE;
            // ---- sweeps 1 and 2, fused tile by tile ------------------------------------------------------------
            // Sweep 1 produces one 16-row tile of W1 at a time; its four row groups are consumed at once as k-steps of
            // sweep 2 (tile t, group gr holds rows R0 = 16t + 4gr .. +3 = four consecutive q of r = R0 / QP), so only ONE
            // W1 tile is ever live (the unfused form kept all nm*QP/16 tiles and all nm W2 accumulators: 256 VGPRs and a
            // scratch spill at nq = 16).  W2[r] goes to its LDS row [r][pos = j*NQ + i] as soon as (a) its last q group
            // is in and (b) every input row that LDS row overlaps has been gathered (w2_store_tile): q and r stay in
            // ascending order, so the sums are the unfused kernel's bit for bit.
            acc_t w2[NM];
#pragma unroll
            for (int t = 0; t < G::MT1; ++t)
            {
                acc_t w1t = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int ks = 0; ks < G::KS1; ++ks)
                {
                    const int p  = ks * 4 + g;
                    const int pc = p < NM ? p : NM - 1;
                    w1t = Op::mma(img[arow[t] + pc], opB0[ks], w1t);
                }
#pragma unroll
                for (int gr = 0; gr < 4; ++gr)
                {
                    const int R0 = 16 * t + 4 * gr;
                    if (R0 < G::M1)
                    {
                        const int r = R0 / QP, qs = (R0 % QP) / 4;
                        if (qs == 0)
                            w2[r] = acc_t{T(0), T(0), T(0), T(0)};
                        w2[r] = Op::mma(opB1[qs], w1t[gr], w2[r]);
                    }
                }
                // W2 rows that may leave now; lane (g,a), register r4 holds j = drow(g, r4), i = a
                bool any = false;
#pragma unroll
                for (int r = 0; r < NM; ++r)
                    any = any || G::w2_store_tile(r) == t;
                if (any)
                {
                    wave_lds_fence(); // the gathers of tiles 0..t have completed
#pragma unroll
                    for (int r = 0; r < NM; ++r)
                        if (G::w2_store_tile(r) == t)
                        {
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
                            {
                                const int j = Op::drow(g, r4);
                                if (j < NQ && a < NQ)
                                    img[r * G::W2S + j * NQ + a] = w2[r][r4];
                            }
                        }
                    wave_lds_fence();
                }
            }
            // ---- sweep 3 ------------------------------------------------------------------------
            acc_t o[G::CB];
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb)
                o[cb] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int rs = 0; rs < G::KS3; ++rs)
            {
                const int r  = rs * 4 + g;
                const int rc = r < NM ? r : NM - 1;
#pragma unroll
                for (int cb = 0; cb < G::CB; ++cb)
                {
                    int pos = cb * 16 + a;
                    if ((cb + 1) * 16 > NQ2)
                        pos = pos < NQ2 ? pos : NQ2 - 1;
                    o[cb] = Op::mma(opB2[rs], img[rc * G::W2S + pos], o[cb]);
                }
            }
            // ---- Out image in LDS (final layout), then a flat stream --------------------------------------
            wave_lds_fence(); // all W2 gathers done before the image is overwritten
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                {
                    const int k = Op::drow(g, r4), pos = cb * 16 + a;
                    if (k < NQ && pos < NQ2)
                        img[k * NQ2 + pos] = o[cb][r4];
                }
            wave_lds_fence();
            T *oe = out + (c * EC + e) * (uint64_t)G::NQT;
            flush_any<G::NQT, T>(img, oe, G::NQT, lane);
            wave_lds_fence();
        }
        wave_lds_fence(); // slab is rewritten by the next chunk's staging
    }
}

} // namespace sf
