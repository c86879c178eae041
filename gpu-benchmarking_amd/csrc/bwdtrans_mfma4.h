// bwdtrans_mfma4.h -- 2D quad BwdTrans on v_mfma_f64_4x4x4_4b_f64: the fp64 instruction MI355X runs fastest.
//
// Why this instruction (measured on the device, profiles/r02/mfma_fp64_rates_and_4x4x4_lane_map.log): with every SIMD
// busy, v_mfma_f64_16x16x4_f64 sustains 33 / 46 / 48 TFLOP/s at 1 / 2 / 4 waves per SIMD, v_fma_f64 42 / 55 / 55, and
// v_mfma_f64_4x4x4_4b_f64 69 / 72 / 68 -- 1.5x the 16x16x4 form and 1.3x the vector pipe, at full rate from ONE wave per
// SIMD, and with a tile granularity of 4 instead of 16 (nm = 27 pads to 28, not 32).  The 16x16x4 kernel of
// bwdtrans_mfma.h therefore sat at that instruction's own ceiling (45 TF/s issued at 2D nq = 32, "MFMA busy 0.68").
// Replaces the same reference kernel, BwdTransQuadKernel_QP_1D (benchmark04/benchmark04.cc:353-426).
//
// The instruction is FOUR independent 4x4x4 products, D_b = A_b * B_b + C_b, b = 0..3.  Lane maps (found with one-hot
// operands, same log), lane = 16*hi + 4*b + lo:
//     A operand   A_b[row = lo][k = hi]
//     B operand   B_b[k = hi][col = lo]
//     C / D       D_b[row = hi][col = lo]
// so a result register is directly the B operand of a product that contracts over ITS row index -- the chaining the
// 16x16x4 kernel uses, at granularity 4:
//     step 1   W[q][i]   = sum_p In[q][p] * B0[p][i]     A = In tile (LDS gather), B = B0 tile (LDS, zero padded)
//     step 2   Out[j][i] = sum_q B1[q][j] * W[q][i]      A = B1 tile (LDS),        B = W  (step 1's accumulators)
// The intermediate never leaves the registers.  The four blocks of an instruction are EB elements x IB = 4/EB
// neighbouring i tiles (EB = 4: four elements, no tile slot is ever idle; EB = 2: two elements x two i tiles, half
// the LDS footprint per wave; EB = 1: one element x four i tiles).  Operands that are the same for several blocks
// (the In tile across i tiles, the basis tiles across elements) are LDS reads of one address by several lanes, which
// the LDS broadcasts; the basis row stride keeps the basis gathers conflict free.
// Padding rows / columns (p, q >= nm; i, j >= nq) meet zero
// entries of the LDS basis copies; data indices are clamped into the element, so padding lanes read finite values.
// The chunk's output is assembled in the slab (the input image is dead after step 1) and leaves as one flat
// line-aligned 16-byte-per-lane stream (chunk_flush, bwdtrans_wave.h).
#pragma once

#include "bwdtrans_wave.h"

namespace sf
{

// basis row stride (doubles): rows hi and hi+1 of a tile read 4*IB consecutive doubles each in one LDS pass, so the
// stride must keep them 4*IB .. 32-4*IB doubles apart modulo the 32 8-byte banks
constexpr int mfma4_basis_stride(int cols, int ib)
{
    int bs = cols;
    while (bs % 32 < 4 * ib || bs % 32 > 32 - 4 * ib)
        ++bs;
    return bs;
}

template <int NQ, int EB> struct Mfma4Geom
{
    static_assert(EB == 1 || EB == 2 || EB == 4, "blocks = EB elements x 4/EB i tiles");
    static constexpr int NM = NQ - 1, IB = 4 / EB;
    static constexpr int NMT = NM * NM, NQT = NQ * NQ;
    static constexpr int TQ = cdiv(NM, 4);  // q tiles = p steps = q steps
    static constexpr int TI = cdiv(NQ, 4);  // i tiles = j tiles
    static constexpr int TG = cdiv(TI, IB); // groups of IB i tiles (one instruction each)
    static constexpr int NMP = 4 * TQ;      // basis rows held in LDS (zero beyond nm)
    // The input image keeps the HBM layout (row stride nm, element stride nm^2): staging is then a flat 16-byte copy
    // with no address arithmetic and no address registers.  The A gathers pay for it with 2-way (some orders 4-way)
    // bank conflicts on ~100 two-cycle reads per chunk, against ~400 sixteen-cycle products; a padded image measured
    // slower (the staging addresses, 2 per staging register, spilled and every chunk paid ~25 VALU per value).
    static constexpr int S    = NM;
    static constexpr int ESTR = NM * NM;
    static constexpr int BS   = mfma4_basis_stride(4 * IB * TG, IB);
    static constexpr int NBAS = NMP * BS;
    static constexpr int SLAB0 = EB * ESTR > EB * NQT ? EB * ESTR : EB * NQT;
    static constexpr int SLAB  = (SLAB0 + 1) & ~1; // doubles per wave
};

// SHB: both directions use the SAME basis array (b0 == b1, the isotropic case of every benchmark run): one LDS copy
template <int NQ, int EB, int WPB, bool SHB = false> constexpr size_t mfma4_lds_bytes()
{
    using G = Mfma4Geom<NQ, EB>;
    return sizeof(double) * (size_t)((SHB ? 1 : 2) * G::NBAS + WPB * G::SLAB);
}

// GJ: j tiles whose accumulators are live together in step 2 (register budget); KMAP: chunks per wave (chunk_iter,
// bwdtrans_wave.h; 0 = persistent grid) -- a workgroup pays for its LDS basis copies once, so it should live for
// several chunks; XG: XCD runs (sf_common.h)
// DYNB > 0: a persistent grid whose waves take BATCHES of DYNB consecutive chunks from a device-wide counter
// (`next_batch`, zeroed before the launch) instead of a fixed share: a fixed share makes the launch as slow as its
// slowest wave, and waves do not run at equal speed (their CU's neighbours, their XCD's memory channels).  One
// address takes ~80 M atomics/s on this part, hence batches; the counter is read a whole batch ahead.
// PEEL: where nm leaves a remainder of one or two rows over a multiple of four (nm = 25, 26, 29, 30: nq = 26, 27, 30, 31)
// the last 4-wide k step of BOTH contractions is all but empty.  Those one or two p (step 1) / q (step 2) go through
// the vector pipe instead -- one v_fma_f64 per accumulator and remaining k, against a 16-clock product per accumulator
// for the padded step -- which removes 1/TQ (12-14 %) of the matrix instructions; the chip sustains a fixed rate of
// issued fp64 matrix work next to the memory stream, so fewer issued products is what moves these orders.
// The sums stay in ascending order of p and q (matrix steps first, the peeled remainder last).
// SPLIT (EB = 2 with an ODD number of i tiles, nq 25..28 and 17..20): an instruction holds two elements x two i tiles, so
// the last, unpaired i tile used to occupy a whole instruction with two of its four blocks idle -- 12.5 % of all
// products at seven tiles.  Its two spare blocks now take every other q tile: block (e, h) computes W[q tile 2u + h][last
// i tile] in step 1 and the partial sum over the q tiles of parity h in step 2; the two partial sums meet through one
// lane swizzle (lanes l and l ^ 4) per output tile.  The unpaired tile then costs half an instruction slot per product:
// 175 instead of 196 instructions per element at nq = 28.  Its sums run over the even q tiles, then the odd ones.
// MEASURED (profiles/r03/tune_mfma4_2[5-8]_split_unpaired_tile.log): 10.7 % fewer matrix instructions change NOTHING on
// the persistent configurations AUTO runs (nq 28: 0.648 with, 0.656 without) and give +3 % on one-chunk workgroups
// (0.569 against 0.551) -- together with the peeled k remainder (-12..14 % instructions, +0..3 %) and the all-zero-data
// run (+3..5 %, tools/experiments/zero_data_clock.py) this rules the matrix pipe out as what bounds nq 25..31.  Off by
// default: AUTO's results stay bit-identical to the generic kernel's.
template <int NQ, int EB, int WPB, int MINW, int GJ, int KMAP, int XG = 0, bool SHB = false, int DYNB = 0, bool PEEL = true,
          bool SPLIT = false, bool STAMP = false, bool EFL = false, int XR = 0, int ST = 0>
__global__ __launch_bounds__(kWave *WPB, MINW) void quad_mfma4_kernel(
    const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ in,
    double *__restrict__ out, uint64_t nelmt, unsigned long long *next_batch = nullptr,
    unsigned long long *stamps = nullptr)
{
    // STAMP (tools/sf_tune_mfma4 only; no product instantiation): shader-clock time per phase of the chunk loop, summed
    // per wave into stamps[8 * wave + 0..5] (staging + next loads issued / step 1 / step 2 + output image / flush issued /
    // wait for the next chunk / chunks).  The stamps go to a buffer of their own; no output value depends on them.
    unsigned long long tphase[5] = {0, 0, 0, 0, 0}, tlast = 0, nch = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP)
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (k >= 0)
                tphase[k] += now - tlast;
            tlast = now;
        }
    };
    using G  = Mfma4Geom<NQ, EB>;
    using GW = WaveGeom<NQ, EB, 2>; // chunk_load / chunk_flush geometry (IN_DBL, NLD, OUT_DBL)
    constexpr int NM = G::NM, IB = G::IB, TQ = G::TQ, TI = G::TI, TG = G::TG, BS = G::BS;
    constexpr int RQ = (PEEL && (NM % 4 == 1 || NM % 4 == 2)) ? NM % 4 : 0; // p / q values contracted on the vector pipe
    constexpr int KQ = RQ ? NM / 4 : TQ;                                     // 4-wide k steps on the matrix pipe
    constexpr bool ODD = SPLIT && IB == 2 && (TI % 2 == 1); // the last i tile has no partner
    constexpr int TGF  = ODD ? TI / 2 : TG;                 // instructions per (q tile, p step) that hold IB paired i tiles
    constexpr int TS   = TI - 1;                            // the unpaired i tile
    constexpr int TQH = (TQ + 1) / 2, KQH = (KQ + 1) / 2;   // q tiles / matrix q steps per half

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bl0 = lds, *bl1 = SHB ? lds : lds + G::NBAS;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *slab   = lds + (SHB ? 1 : 2) * G::NBAS + wib * G::SLAB;
    const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
    const int e = blk / IB, ib = blk % IB;

    // The wave's first chunk is requested BEFORE the workgroup builds its LDS basis copies: the basis loads, the LDS
    // writes and the barrier then run underneath the chunk's HBM latency instead of in front of it (a one-chunk wave
    // otherwise pays for the prologue in full).
    constexpr uint64_t kNone = ~0ull;
    const uint64_t nchunk = (nelmt + EB - 1) / EB;
    const ChunkIter it    = chunk_iter<KMAP, WPB, XG>(nchunk, wib);
    uint64_t c = kNone, cn = kNone; // this chunk, the next one (its loads are requested while this one is computed)
    // batch counter: issue (lane 0) now, look at the value a batch later
    // XR > 0: one ticket counter per XCD (the eight counters share the launch's 64-byte slot).  Ticket k of XCD x is batch
    // ((k / XR) * 8 + x) * XR + k % XR: every XCD works through runs of XR neighbouring batches, and the runs of the eight
    // XCDs interleave inside a window of 8 XR batches -- what logical_block() does for grids that cover the array
    // (neighbours in memory share an L2, the DRAM front stays bounded).  An XCD only ever takes its own runs.
    // ST > 0: ST ticket counters, 128 bytes apart (a line of their own each: one line takes ~80 M atomics/s, which is
    // what forces batches on a single counter); wave w draws from counter w % ST, whose ticket k is batch k * ST + w % ST.
    // The counters advance at the same average rate, so the batches in flight stay neighbours in memory even at one
    // chunk per batch.
    int xcc = 0;
    if constexpr (XR > 0)
    {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
    }
    static_assert(XR == 0 || ST == 0, "one ticket scheme at a time");
    if constexpr (ST > 0)
        xcc = (int)((blockIdx.x * WPB + wib) % ST);
    auto grab_issue = [&]() -> unsigned long long {
        unsigned long long v = 0;
        if (lane == 0)
            v = __hip_atomic_fetch_add(next_batch + (ST > 0 ? 16 * xcc : xcc), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto grab_finish = [&](unsigned long long v) -> uint64_t { // first chunk of that batch, or kNone
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        uint64_t batch      = ((uint64_t)hi32 << 32) | lo32;
        if constexpr (XR > 0)
            batch = ((batch / XR) * 8 + (uint64_t)xcc) * XR + batch % XR;
        if constexpr (ST > 0)
            batch = batch * ST + (uint64_t)xcc;
        const uint64_t first = batch * (uint64_t)(DYNB > 0 ? DYNB : 1);
        return first < nchunk ? first : kNone;
    };
    unsigned long long pending = 0;
    if constexpr (DYNB > 0)
    {
        c       = grab_finish(grab_issue());
        pending = grab_issue();
        if (c != kNone)
            cn = (DYNB > 1 && c + 1 < nchunk) ? c + 1 : kNone; // kNone here = "ask the counter" (resolved in the loop)
    }
    else if (it.count != 0)
    {
        c  = it.first;
        cn = it.count > 1 ? it.first + it.step : kNone;
    }
    typename GW::Vec st[GW::NLD];
    if (c != kNone)
        chunk_fetch<GW, EB>(st, in, c, nelmt, lane);

    // zero-padded LDS copies of the two bases (once per workgroup); every load is requested before the first is used
    {
        constexpr int NIT = cdiv(G::NBAS, kWave * WPB);
        double v0[NIT], v1[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k)
        {
            const int x = k * kWave * WPB + (int)threadIdx.x;
            const int p = x / BS, i = x - p * BS;
            const bool real = p < NM && i < NQ;
            v0[k] = real ? b0[p * NQ + i] : 0.0;
            v1[k] = (real && !SHB) ? b1[p * NQ + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k)
        {
            const int x = k * kWave * WPB + (int)threadIdx.x;
            if (x < G::NBAS)
            {
                bl0[x] = v0[k];
                if constexpr (!SHB)
                    bl1[x] = v1[k];
            }
        }
    }
    __syncthreads();
    if (c == kNone)
        return;

    for (uint64_t n = 0;; ++n)
    {
        const uint64_t left = nelmt - c * EB;
        const int evalid    = left >= (uint64_t)EB ? EB : (int)left;
        stamp(-1);

        // ---- chunk: staging registers -> LDS image (HBM layout: flat copy); then request the next chunk -----------------
        if constexpr (GW::VEC2)
        {
#pragma unroll
            for (int k = 0; k < GW::NLD; ++k)
            {
                const int v = k * kWave + lane;
                if ((k + 1) * kWave <= GW::IN_DBL / 2 || v < GW::IN_DBL / 2)
                    *reinterpret_cast<double2_t *>(slab + 2 * v) = st[k];
            }
        }
        else
        {
            const int sh = line_offset<double>(in + c * GW::IN_DBL);
#pragma unroll
            for (int k = 0; k < word_grid_regs<GW::IN_DBL, double>(); ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    const int f = 2 * (k * kWave + lane) - sh + h;
                    if (f >= 0 && f < GW::IN_DBL)
                        slab[f] = st[k][h];
                }
        }
        wave_lds_fence();
        uint64_t cnn = kNone; // the chunk after the next
        if constexpr (DYNB > 0)
        {
            // c was the last chunk of its batch (or of the batch's valid part): the next chunk opens the batch requested
            // one batch ago
            if (cn == kNone && ((c + 1) % DYNB == 0 || c + 1 >= nchunk))
            {
                cn      = grab_finish(pending);
                pending = grab_issue();
            }
            if (cn != kNone)
            {
                chunk_fetch<GW, EB>(st, in, cn, nelmt, lane);
                cnn = ((cn + 1) % DYNB != 0 && cn + 1 < nchunk) ? cn + 1 : kNone;
            }
        }
        else if (cn != kNone)
        {
            cnn = n + 2 < it.count ? cn + it.step : kNone;
            chunk_fetch<GW, EB>(st, in, cn, nelmt, lane);
        }

        stamp(0);
        // ---- step 1: W[q][i] = sum_p In[q][p] B0[p][i]; D = W[q on hi][i on lo] of block (e, ib) ---------------------
        double w[TQ][TG];
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
            for (int ig = 0; ig < TG; ++ig)
                w[tq][ig] = 0.0;
        double ws[ODD ? TQH : 1]; // W[q tile 2u + ib][unpaired i tile]
#pragma unroll
        for (int u = 0; u < (ODD ? TQH : 1); ++u)
            ws[u] = 0.0;
        {
            int arow[TQ];
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq)
            {
                const int q = 4 * tq + lo;
                arow[tq]    = e * G::ESTR + (q < NM ? q : NM - 1) * G::S;
            }
            const double *btile = bl0 + hi * BS + 4 * ib + lo;
            // operands of p step tp+1 are requested before the products of step tp are issued: an LDS read takes longer
            // than the few 16-cycle products that consume it, and hipcc does not move it up by itself
            double a1[2][TQ], bt[2][TG];
            // the unpaired i tile: block (e, h = ib) takes the q tiles 2u + h
            int arows[ODD ? TQH : 1];
            double a1s[2][ODD ? TQH : 1], bts[2];
            if constexpr (ODD)
            {
#pragma unroll
                for (int u = 0; u < TQH; ++u)
                {
                    const int q = 4 * (2 * u + ib) + lo;
                    arows[u]    = e * G::ESTR + (q < NM ? q : NM - 1) * G::S;
                }
            }
            auto request = [&](int buf, int tp) {
                const int p  = 4 * tp + hi;
                const int pc = p < NM ? p : NM - 1;
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq)
                    a1[buf][tq] = slab[arow[tq] + pc];
#pragma unroll
                for (int ig = 0; ig < TGF; ++ig)
                    bt[buf][ig] = btile[4 * tp * BS + 4 * IB * ig];
                if constexpr (ODD)
                {
#pragma unroll
                    for (int u = 0; u < TQH; ++u)
                        a1s[buf][u] = slab[arows[u] + pc];
                    bts[buf] = bl0[(4 * tp + hi) * BS + 4 * TS + lo];
                }
            };
            request(0, 0);
#pragma unroll
            for (int tp = 0; tp < KQ; ++tp)
            {
                if (tp + 1 < KQ)
                    request((tp + 1) & 1, tp + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ig = 0; ig < TGF; ++ig)
#pragma unroll
                    for (int tq = 0; tq < TQ; ++tq)
                        w[tq][ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[tp & 1][tq], bt[tp & 1][ig], w[tq][ig], 0, 0, 0);
                if constexpr (ODD)
                {
#pragma unroll
                    for (int u = 0; u < TQH; ++u)
                        ws[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1s[tp & 1][u], bts[tp & 1], ws[u], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (RQ > 0)
            {
                // peeled p: W[q = 4tq + hi][i] += In[q][p] * B0[p][i] on the accumulators' own lanes (row on hi, column
                // on lo); rows beyond nm are padding and read a clamped (finite) row
                double ain[RQ][TQ], bin[RQ][TG];
#pragma unroll
                for (int d = 0; d < RQ; ++d)
                {
#pragma unroll
                    for (int tq = 0; tq < TQ; ++tq)
                    {
                        const int q = 4 * tq + hi;
                        ain[d][tq]  = slab[e * G::ESTR + (q < NM ? q : NM - 1) * G::S + 4 * KQ + d];
                    }
#pragma unroll
                    for (int ig = 0; ig < TGF; ++ig)
                        bin[d][ig] = bl0[(4 * KQ + d) * BS + 4 * (IB * ig + ib) + lo];
                }
#pragma unroll
                for (int d = 0; d < RQ; ++d)
#pragma unroll
                    for (int ig = 0; ig < TGF; ++ig)
#pragma unroll
                        for (int tq = 0; tq < TQ; ++tq)
                            w[tq][ig] = __builtin_fma(ain[d][tq], bin[d][ig], w[tq][ig]);
                if constexpr (ODD)
                {
#pragma unroll
                    for (int d = 0; d < RQ; ++d)
                    {
                        const double bs1 = bl0[(4 * KQ + d) * BS + 4 * TS + lo];
#pragma unroll
                        for (int u = 0; u < TQH; ++u)
                        {
                            const int q = 4 * (2 * u + ib) + hi;
                            ws[u] = __builtin_fma(slab[e * G::ESTR + (q < NM ? q : NM - 1) * G::S + 4 * KQ + d], bs1, ws[u]);
                        }
                    }
                }
            }
        }
        // peeled q: row q = 4 KQ + d of W sits in w[TQ - 1] on the lanes with hi == d; every lane needs it for its own
        // column, i.e. from lane 16 d + (lane & 15)
        double wf[RQ > 0 ? RQ : 1][TG];
        if constexpr (RQ > 0)
        {
#pragma unroll
            for (int d = 0; d < RQ; ++d)
#pragma unroll
                for (int ig = 0; ig < TGF; ++ig)
                {
                    const int src = 4 * (16 * d + (lane & 15));
                    const double v = w[TQ - 1][ig];
                    const int vlo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
                    const int vhi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
                    wf[d][ig]     = __hiloint2double(vhi, vlo);
                }
        }
        // the same for the unpaired tile: q tile TQ - 1 sits in ws[(TQ - 1) / 2] of the blocks with ib == (TQ - 1) % 2
        double wfs[RQ > 0 ? RQ : 1];
        if constexpr (RQ > 0 && ODD)
        {
#pragma unroll
            for (int d = 0; d < RQ; ++d)
            {
                const int src = 4 * (16 * d + 4 * (2 * e + (TQ - 1) % 2) + lo);
                const double v = ws[(TQ - 1) / 2];
                const int vlo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
                const int vhi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
                wfs[d]        = __hiloint2double(vhi, vlo);
            }
        }
        wave_lds_fence(); // every gather of the input image has completed: the slab becomes the output image
        stamp(1);

        // ---- step 2: Out[j][i] = sum_q B1[q][j] W[q][i]; D = Out[j on hi][i on lo] ------------------------------------
        {
            const double *atile = bl1 + hi * BS + lo;
            double *oimg        = slab + e * G::NQT + hi * NQ + 4 * ib + lo;
#pragma unroll
            for (int j0 = 0; j0 < TI; j0 += GJ)
            {
                double o[GJ][TG];
#pragma unroll
                for (int t = 0; t < GJ; ++t)
#pragma unroll
                    for (int ig = 0; ig < TG; ++ig)
                        o[t][ig] = 0.0;
                double a2[2][GJ];
                // unpaired i tile: os[t] = partial sum over the q tiles 2u + ib of Out[j tile j0 + t][i tile TS]
                double os[ODD ? GJ : 1], a2s[2][ODD ? GJ : 1];
#pragma unroll
                for (int t = 0; t < (ODD ? GJ : 1); ++t)
                    os[t] = 0.0;
                auto request = [&](int buf, int tq) {
#pragma unroll
                    for (int t = 0; t < GJ; ++t)
                        if (j0 + t < TI)
                            a2[buf][t] = atile[4 * tq * BS + 4 * (j0 + t)];
                };
                auto request_s = [&](int buf, int u) {
                    const int qt = 2 * u + ib; // this block's q tile; beyond the matrix steps it contributes nothing
#pragma unroll
                    for (int t = 0; t < GJ; ++t)
                        if (j0 + t < TI)
                        {
                            const double v = atile[4 * (qt < KQ ? qt : KQ - 1) * BS + 4 * (j0 + t)];
                            a2s[buf][t]    = qt < KQ ? v : 0.0;
                        }
                };
                request(0, 0);
#pragma unroll
                for (int tq = 0; tq < KQ; ++tq)
                {
                    if (tq + 1 < KQ)
                        request((tq + 1) & 1, tq + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < GJ; ++t)
                    {
                        if (j0 + t >= TI)
                            continue;
#pragma unroll
                        for (int ig = 0; ig < TGF; ++ig)
                            o[t][ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[tq & 1][t], w[tq][ig], o[t][ig], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (ODD)
                {
                    request_s(0, 0);
#pragma unroll
                    for (int u = 0; u < KQH; ++u)
                    {
                        if (u + 1 < KQH)
                            request_s((u + 1) & 1, u + 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < GJ; ++t)
                            if (j0 + t < TI)
                                os[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2s[u & 1][t], ws[u], os[t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (RQ > 0)
                {
                    // peeled q: Out[j = 4tj + hi][i] += B1[q][j] * W[q][i]
#pragma unroll
                    for (int d = 0; d < RQ; ++d)
#pragma unroll
                        for (int t = 0; t < GJ; ++t)
                        {
                            if (j0 + t >= TI)
                                continue;
                            const double bq = bl1[(4 * KQ + d) * BS + 4 * (j0 + t) + hi];
#pragma unroll
                            for (int ig = 0; ig < TGF; ++ig)
                                o[t][ig] = __builtin_fma(bq, wf[d][ig], o[t][ig]);
                            if constexpr (ODD) // once: on the half that holds the even q tiles
                                os[t] = __builtin_fma(ib == 0 ? bq : 0.0, wfs[d], os[t]);
                        }
                }
#pragma unroll
                for (int t = 0; t < GJ; ++t)
#pragma unroll
                    for (int ig = 0; ig < TGF; ++ig)
                    {
                        const int tj = j0 + t;
                        if (tj < TI && 4 * tj + hi < NQ && 4 * (IB * ig + ib) + lo < NQ)
                            oimg[4 * tj * NQ + 4 * IB * ig] = o[t][ig];
                    }
                if constexpr (ODD)
                {
                    // the two halves of a block pair (lanes l and l ^ 4) hold the even- and the odd-q-tile partial sums
#pragma unroll
                    for (int t = 0; t < GJ; ++t)
                    {
                        const int tj = j0 + t;
                        if (tj >= TI)
                            continue;
                        const double mine = os[t];
                        const int plo = __builtin_amdgcn_ds_swizzle(__double2loint(mine), 0x101F); // lane ^ 4
                        const int phi = __builtin_amdgcn_ds_swizzle(__double2hiint(mine), 0x101F);
                        const double sum = mine + __hiloint2double(phi, plo);
                        if (ib == 0 && 4 * tj + hi < NQ && 4 * TS + lo < NQ)
                            slab[e * G::NQT + (4 * tj + hi) * NQ + 4 * TS + lo] = sum;
                    }
                }
                if constexpr (EFL)
                {
                    // EFL: the rows of this j group are final: stream them out now, so that a chunk's stores are spread
                    // over its matrix phase instead of leaving as one burst of 13 at the end (the burst holds the wave
                    // for 5 256 of its 17 828 clocks at nq 28, profiles/r03/mfma4_phase_stamps_nq28.log)
                    wave_lds_fence();
                    constexpr int ROWS = 4 * GJ;
                    const int nrows    = NQ - 4 * j0 < ROWS ? NQ - 4 * j0 : ROWS;
#pragma unroll
                    for (int e2 = 0; e2 < EB; ++e2)
                        if (e2 < evalid)
                            flush_any<ROWS * NQ, double>(slab + e2 * G::NQT + 4 * j0 * NQ,
                                                         out + (c * EB + e2) * (uint64_t)G::NQT + 4 * j0 * NQ, nrows * NQ, lane);
                }
            }
        }
        wave_lds_fence();
        stamp(2);
        // 16 B per lane, every wave-wide store on whole 128-byte lines (word-grid store when nq^2 is odd and EB = 1)
        if constexpr (!EFL)
            chunk_flush<GW, true, true>(slab, out + c * (uint64_t)GW::OUT_DBL, evalid * G::NQT, lane);
        wave_lds_fence(); // the slab is rewritten by the next chunk's staging
        stamp(3);
        ++nch;
        if (cn == kNone)
            break;
        touch_staged(st); // counted wait for the next chunk here, not vmcnt(0) at the loop header
        stamp(4);
        c  = cn;
        cn = cnn;
    }
    if constexpr (STAMP)
    {
        if (lane == 0 && stamps) // a slot per wave (atomics on one address would serialise half a million waves)
        {
            unsigned long long *mine = stamps + ((uint64_t)blockIdx.x * WPB + wib) * 8;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                mine[k] = tphase[k];
            mine[5] = nch;
        }
    }
}

} // namespace sf
