// bwdtrans_hmfma4.h -- 3D BwdTrans on v_mfma_f64_4x4x4_4b_f64 (fp64, nq 12..16): the three chained GEMMs of
// bwdtrans_mfma.h's hex_mfma_kernel on the instruction that runs at the part's full fp64 matrix rate.
//
// Why a second 3D matrix-core kernel: v_mfma_f64_16x16x4_f64 sustains 46-48 TFLOP/s on this part (93 clocks per 2048-flop
// instruction), v_mfma_f64_4x4x4_4b_f64 67-72 (16 clocks per 512 flops; tools/sf_mfma_probe,
// profiles/r03/mfma_probe_rates_f64_f32.log).  At nq = 16 one element is 184 16x16x4 products = 17 112 matrix-pipe clocks
// against 5 000 clocks of HBM time per element and CU (four SIMDs: 0.85 of the memory time -- the pipe is a co-bound); the
// same sums as 736 4x4x4_4b products take 11 776 clocks (0.59).  4-granular tiles also pad nm = 11 to 12 instead of 16.
//
// Lane maps (sf_mfma_probe): lane = 16*hi + 4*blk + lo; block blk of an instruction multiplies A_blk[row = lo][k = hi] by
// B_blk[k = hi][col = lo] into D_blk[row = hi][col = lo].  One wave owns one element; for every slice r:
//   sweep 1  W1[q][i] = sum_p In[r][q][p] B0[p][i]   blocks = the four i tiles; A = In gathered from LDS (one read per
//            product, the same 16 values for every block), B = B0 in registers; D: q on hi, i on (blk, lo)
//   sweep 2  W2[j][i] = sum_q B1[q][j] W1[q][i]      A = B1^T in registers, B = the sweep-1 accumulators as they are
//            (k = q on hi): W1 never leaves the registers
//   W2[r] -> LDS row [r][pos = j*nq + i] once every input row it overlaps has been gathered
// then sweep 3  Out[k][pos] = sum_r B2[r][k] W2[r][pos]: blocks = four neighbouring pos tiles, A = B2^T in registers, B
// gathered from LDS once per r step and reused for the four k tiles.  The result is assembled in LDS in final layout and
// leaves as one flat 16-byte-per-lane stream.  k runs in ascending order everywhere: the sums are the scalar loops'.
#pragma once

#include "bwdtrans_wave.h"

namespace sf
{

template <int NQ, bool DIRECT = false> struct HexMfma4Geom
{
    static constexpr int NM  = NQ - 1;
    static constexpr int NMT = NM * NM * NM, NQ2 = NQ * NQ, NQT = NQ * NQ * NQ;
    static constexpr int TP  = cdiv(NM, 4);  // p / q / r tiles (k steps of the three sweeps, q row tiles of sweep 1)
    static constexpr int TI  = cdiv(NQ, 4);  // i / j / k tiles
    static constexpr int CG  = cdiv(NQ2, 16); // groups of four pos tiles (sweep 3)
    static constexpr int S   = NM; // input image = the element as it lies in HBM (a sweep-1 gather touches 8 distinct
                                   // addresses per half wave: no stride is needed to keep it conflict-free)
    // W2 row stride: = 16 mod 32 doubles (conflict-free sweep-3 gathers), >= NQ2.  DIRECT: no padding -- a two-way
    // conflict on 16 CG reads per element is cheaper than the workgroup per CU the padding costs at nq = 16
    static constexpr int W2S = DIRECT ? (NQ2 + 15) / 16 * 16 : (NQ2 + 15) / 32 * 32 + 16;
    static constexpr int NLD = word_grid_regs<NMT, double>();
    static constexpr int E0  = NM * NM * S > NM * W2S ? NM * NM * S : NM * W2S;
    static constexpr int SLAB = ((DIRECT || E0 > NQT ? E0 : NQT) + 1) / 2 * 2; // doubles per wave (DIRECT: no output image)
    static_assert(TI <= 4, "the i tiles of one slice are the four blocks of an instruction");
    static_assert((DIRECT || W2S % 32 == 16) && W2S >= NQ2, "LDS strides");
    // the slice after whose gathers W2[r] may be written to its LDS row [r*W2S, r*W2S + NQ2): its own, or the last input
    // slice that row overlaps (input row (r', q') lives at (r'*NM + q')*S)
    static constexpr int w2_store_after(int r)
    {
        int fr = (r * W2S + NQ2 - 1) / S; // last overlapped flat input row
        fr     = fr < NM * NM - 1 ? fr : NM * NM - 1;
        const int need = fr / NM;
        return need > r ? need : r;
    }
};

template <int NQ, int WPB, bool DIRECT = false> constexpr size_t hex_mfma4_lds_bytes()
{
    return sizeof(double) * (size_t)WPB * HexMfma4Geom<NQ, DIRECT>::SLAB;
}

__device__ __forceinline__ double mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// v of lane src (any lane pattern): two 32-bit ds_bpermute
__device__ __forceinline__ double lane_gather(double v, int src)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int l = __builtin_amdgcn_ds_bpermute(4 * src, (int)b);
    const int h = __builtin_amdgcn_ds_bpermute(4 * src, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)h << 32) | (unsigned)l);
}

// STAMP (tools/sf_tune only; no product instantiation): shader-clock time per phase of the element loop, summed per wave
// into stamps[8 * wave + 0..5] (staging + next loads issued / sweeps 1 + 2 / sweep 3 / output image + flush issued / wait
// for the next element / elements)
// DIRECT: the sweep-3 accumulators of a pos group go straight to HBM (lane: 4 k rows x 16 consecutive pos = four 128-byte
// runs per instruction) instead of through an output image in LDS: no 64 live accumulators, no image / flush phase
// PEEL: where nm leaves a remainder of one or two over a multiple of four (nm = 13, 14: nq = 14, 15) the last 4-wide k step
// of all three contractions is all but empty; those one or two p / q / r go through the vector pipe instead (one v_fma_f64
// per accumulator and remaining k), which removes a quarter of the matrix instructions.  nq 14 / 15 issue 8.1 / 7.1
// padded flops per byte of traffic without it (nq 12 / 13 / 16: 5.8 / 6.3 / 6.3) and run at 0.65 of the roofline where
// those reach 0.72-0.75.  The sums stay in ascending order of k (matrix steps first, the peeled remainder last).
template <int NQ, int WPB, int MINW, int KMAP, int XG = 0, bool STAMP = false, bool DIRECT = false, bool NTS = true,
          bool PEEL = true>
__global__ __launch_bounds__(kWave *WPB, MINW) void hex_mfma4_kernel(
    const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ b2,
    const double *__restrict__ in, double *__restrict__ out, uint64_t nelmt, unsigned long long *stamps = nullptr)
{
    unsigned long long tphase[5] = {0, 0, 0, 0, 0}, tlast = 0, nel = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP)
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (k >= 0)
                tphase[k] += now - tlast;
            tlast = now;
        }
    };
    using G = HexMfma4Geom<NQ, DIRECT>;
    constexpr int NM = G::NM, NQ2 = G::NQ2, TP = G::TP, TI = G::TI, CG = G::CG, S = G::S, W2S = G::W2S;
    constexpr int RQ = (PEEL && (NM % 4 == 1 || NM % 4 == 2)) ? NM % 4 : 0; // k values contracted on the vector pipe
    constexpr int KM = RQ ? NM / 4 : TP;                                     // 4-wide k steps on the matrix pipe
    constexpr int RQ1 = RQ ? RQ : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_hexmfma4[];
    double *lds    = reinterpret_cast<double *>(lds_raw_hexmfma4);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
    double *img = lds + wib * G::SLAB;

    const ChunkIter it = chunk_iter<KMAP, WPB, XG>(nelmt, wib);
    if (it.count == 0)
        return;

    double2_t st[G::NLD];
    chunk_load_any<G::NMT, G::NLD, double>(st, in + it.first * G::NMT, lane, G::NMT);

    // basis operands (zero outside nm x nq), loaded under the first element's HBM latency
    double opB0[KM], opB1[TI][KM], opB2[TI][KM];
#pragma unroll
    for (int kp = 0; kp < KM; ++kp)
    {
        const int p = 4 * kp + hi, i = 4 * blk + lo;
        opB0[kp]    = (p < NM && i < NQ) ? b0[p * NQ + i] : 0.0; // B[k = p][col = i]
    }
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int ks = 0; ks < KM; ++ks)
        {
            const int kk = 4 * ks + hi, o = 4 * t + lo; // A[row = o (j or k)][k = kk (q or r)]
            opB1[t][ks]  = (kk < NM && o < NQ) ? b1[kk * NQ + o] : 0.0;
            opB2[t][ks]  = (kk < NM && o < NQ) ? b2[kk * NQ + o] : 0.0;
        }
    // the peeled k values 4 KM + u work on the accumulators where they are (D layout: row on hi, column on (blk, lo)):
    // pb0[u] = B0[k][i = 4 blk + lo], pb1[t][u] = B1[k][j = 4 t + hi], pb2[t][u] = B2[k][k' = 4 t + hi]
    double pb0[RQ1], pb1[TI][RQ1], pb2[TI][RQ1];
#pragma unroll
    for (int u = 0; u < RQ; ++u)
    {
        const int kk = 4 * KM + u, i = 4 * blk + lo;
        pb0[u]       = i < NQ ? b0[kk * NQ + i] : 0.0;
#pragma unroll
        for (int t = 0; t < TI; ++t)
        {
            const int o = 4 * t + hi;
            pb1[t][u]   = o < NQ ? b1[kk * NQ + o] : 0.0;
            pb2[t][u]   = o < NQ ? b2[kk * NQ + o] : 0.0;
        }
    }
    // sweep-1 gather offsets inside a slice: row q = 4 tq + lo, column p = 4 kp + hi, clamped into the element
    int aoff[TP][KM];
#pragma unroll
    for (int tq = 0; tq < TP; ++tq)
#pragma unroll
        for (int kp = 0; kp < KM; ++kp)
        {
            const int q = 4 * tq + lo, p = 4 * kp + hi;
            aoff[tq][kp] = (q < NM ? q : NM - 1) * S + (p < NM ? p : NM - 1);
        }
    int poff[TP]; // peeled p of row q = 4 tq + hi (the accumulator's row)
#pragma unroll
    for (int tq = 0; tq < TP; ++tq)
    {
        const int q = 4 * tq + hi;
        poff[tq]    = (q < NM ? q : NM - 1) * S + 4 * KM;
    }

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        stamp(-1);
        // staging registers -> LDS, flat
        {
            const int a0 = line_offset<double>(in + c * G::NMT);
#pragma unroll
            for (int k = 0; k < G::NLD; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    const int f = 2 * (k * kWave + lane) - a0 + h;
                    if (f >= 0 && f < G::NMT)
                        img[f] = st[k][h];
                }
        }
        wave_lds_fence();
        if (n + 1 < it.count)
            chunk_load_any<G::NMT, G::NLD, double>(st, in + (c + it.step) * G::NMT, lane, G::NMT);

        stamp(0);
        // ---- sweeps 1 and 2, slice by slice ---------------------------------------------------------------------
        // One or two waves per SIMD run this kernel and a product takes 16 clocks, so nothing but the wave itself can
        // cover an LDS round trip: the sixteen sweep-1 operands of slice r + 1 are requested before the products of
        // slice r are issued (sweep 3: the operands of the next pos group).
        double w2[NM][TI];
        double av[2][TP][KM], ap[2][TP][RQ1];
        auto gather = [&](double (&dst)[TP][KM], double (&dp)[TP][RQ1], int r) {
            const double *slice = img + r * NM * S;
#pragma unroll
            for (int tq = 0; tq < TP; ++tq)
#pragma unroll
                for (int kp = 0; kp < KM; ++kp)
                    dst[tq][kp] = slice[aoff[tq][kp]];
#pragma unroll
            for (int tq = 0; tq < TP; ++tq)
#pragma unroll
                for (int u = 0; u < RQ; ++u)
                    dp[tq][u] = slice[poff[tq] + u];
        };
        gather(av[0], ap[0], 0);
#pragma unroll
        for (int r = 0; r < NM; ++r)
        {
            if (r + 1 < NM)
                gather(av[(r + 1) & 1], ap[(r + 1) & 1], r + 1);
            double w1[TP];
#pragma unroll
            for (int tq = 0; tq < TP; ++tq)
                w1[tq] = 0.0;
#pragma unroll
            for (int kp = 0; kp < KM; ++kp)
#pragma unroll
                for (int tq = 0; tq < TP; ++tq)
                    w1[tq] = mfma4(av[r & 1][tq][kp], opB0[kp], w1[tq]);
#pragma unroll
            for (int u = 0; u < RQ; ++u)
#pragma unroll
                for (int tq = 0; tq < TP; ++tq)
                    w1[tq] = __builtin_fma(ap[r & 1][tq][u], pb0[u], w1[tq]);
#pragma unroll
            for (int tj = 0; tj < TI; ++tj)
                w2[r][tj] = 0.0;
#pragma unroll
            for (int tq = 0; tq < KM; ++tq)
#pragma unroll
                for (int tj = 0; tj < TI; ++tj)
                    w2[r][tj] = mfma4(opB1[tj][tq], w1[tq], w2[r][tj]);
#pragma unroll
            for (int u = 0; u < RQ; ++u)
            {
                // W1[q = 4 KM + u][i] sits in w1[KM] on the lanes with hi = u: every lane takes it from lane 16 u + (lane & 15)
                const double wq = lane_gather(w1[KM < TP ? KM : TP - 1], 16 * u + (lane & 15));
#pragma unroll
                for (int tj = 0; tj < TI; ++tj)
                    w2[r][tj] = __builtin_fma(pb1[tj][u], wq, w2[r][tj]);
            }
            // W2 rows that may leave now: lane holds W2[r'][j = 4 tj + hi][i = 4 blk + lo].  The gathers of slice r + 1
            // are already issued (LDS operations of a wave execute in order), and a row never overlaps a later slice
            // than w2_store_after() names.
#pragma unroll
            for (int rr = 0; rr < NM; ++rr)
                if (G::w2_store_after(rr) == r)
                {
#pragma unroll
                    for (int tj = 0; tj < TI; ++tj)
                    {
                        const int j = 4 * tj + hi, i = 4 * blk + lo;
                        if (j < NQ && i < NQ)
                            img[rr * W2S + j * NQ + i] = w2[rr][tj];
                    }
                }
        }
        wave_lds_fence();
        stamp(1);
        // ---- sweep 3 --------------------------------------------------------------------------------------------
        double o[DIRECT ? 1 : CG][TI];
        double *oe = out + c * (uint64_t)G::NQT;
        double bv[2][KM], bp[2][RQ1];
        int rrow[KM];
#pragma unroll
        for (int kr = 0; kr < KM; ++kr)
        {
            const int r = 4 * kr + hi;
            rrow[kr]    = (r < NM ? r : NM - 1) * W2S;
        }
        auto gather3 = [&](double (&dst)[KM], double (&dp)[RQ1], int cg) {
            int pos = 16 * cg + 4 * blk + lo;
            if ((cg + 1) * 16 > NQ2)
                pos = pos < NQ2 ? pos : NQ2 - 1;
#pragma unroll
            for (int kr = 0; kr < KM; ++kr)
                dst[kr] = img[rrow[kr] + pos];
#pragma unroll
            for (int u = 0; u < RQ; ++u)
                dp[u] = img[(4 * KM + u) * W2S + pos];
        };
        gather3(bv[0], bp[0], 0);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg)
        {
            if (cg + 1 < CG)
                gather3(bv[(cg + 1) & 1], bp[(cg + 1) & 1], cg + 1);
            double(&oc)[TI] = o[DIRECT ? 0 : cg];
#pragma unroll
            for (int tk = 0; tk < TI; ++tk)
                oc[tk] = 0.0;
#pragma unroll
            for (int kr = 0; kr < KM; ++kr)
#pragma unroll
                for (int tk = 0; tk < TI; ++tk)
                    oc[tk] = mfma4(opB2[tk][kr], bv[cg & 1][kr], oc[tk]);
#pragma unroll
            for (int u = 0; u < RQ; ++u)
#pragma unroll
                for (int tk = 0; tk < TI; ++tk)
                    oc[tk] = __builtin_fma(pb2[tk][u], bp[cg & 1][u], oc[tk]);
            if constexpr (DIRECT)
            {
#pragma unroll
                for (int tk = 0; tk < TI; ++tk)
                {
                    const int k = 4 * tk + hi, pos = 16 * cg + 4 * blk + lo;
                    if (k < NQ && pos < NQ2)
                    {
                        if constexpr (NTS)
                            __builtin_nontemporal_store(oc[tk], oe + k * NQ2 + pos);
                        else
                            oe[k * NQ2 + pos] = oc[tk]; // partial lines of neighbouring pos groups meet in the L2
                    }
                }
            }
        }
        if constexpr (STAMP && !DIRECT)
        {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int tk = 0; tk < TI; ++tk)
                    asm volatile("" : "+v"(o[cg][tk])); // the products have retired before the stamp
        }
        stamp(2);
        if constexpr (!DIRECT)
        {
            // ---- Out image in LDS (final layout), then a flat stream ---------------------------------------------------
            wave_lds_fence(); // all W2 gathers done before the image is overwritten
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int tk = 0; tk < TI; ++tk)
                {
                    const int k = 4 * tk + hi, pos = 16 * cg + 4 * blk + lo;
                    if (k < NQ && pos < NQ2)
                        img[k * NQ2 + pos] = o[cg][tk];
                }
            wave_lds_fence();
            flush_any<G::NQT, double>(img, oe, G::NQT, lane);
        }
        wave_lds_fence(); // the image is rewritten by the next element's staging
        stamp(3);
        if (n + 1 < it.count)
            touch_staged(st); // counted wait for the next element here, not vmcnt(0) at the loop header
        stamp(4);
        ++nel;
    }
    if constexpr (STAMP)
    {
        if (lane == 0)
        {
            unsigned long long *slot = stamps + 8 * ((uint64_t)blockIdx.x * WPB + wib);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                slot[k] = tphase[k];
            slot[5] = nel;
        }
    }
}

} // namespace sf
