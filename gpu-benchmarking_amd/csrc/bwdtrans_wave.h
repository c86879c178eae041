// bwdtrans_wave.h -- flagship BwdTrans kernels for gfx950: one WAVEFRONT streams chunks of elements.
//
// Replaces the reference's block-per-element kernels BwdTransHexKernel_QP (shared)
// (benchmark05/benchmark05.cc:291-429) and BwdTransQuadKernel_QP_1D (shared)
// (benchmark04/benchmark04.cc:353-426).  Same maths and the same sweep order (p, then q, then r,
// ascending summation inside each dot product); everything else is designed for CDNA4:
//
//  * Work unit = a CHUNK of EC consecutive elements owned by ONE 64-lane wavefront.  No workgroup
//    barrier in the element loop (the reference pays 4 __syncthreads per element); lanes of the wave
//    hand data to each other through a private LDS slab, ordered by wave_lds_fence().
//  * Chunk -> wave mapping (template KMAP): K > 0 (shipped) = short-lived waves, wave w takes the K
//    consecutive chunks [wK, wK+K) and the grid covers the batch -- the hardware dispatcher keeps the DRAM
//    access front tight, 10-15 % faster on this read/write mix than KMAP = 0, a persistent grid where wave
//    w takes chunks w, w+W, w+2W ... (profiles/r01/tune_nq8_b_kmap_st16.log).
//  * HBM -> registers with flat, coalesced 16-B-per-lane non-temporal loads of the whole chunk (EC*nm^d
//    scalars are contiguous), issued one chunk AHEAD and parked in VGPRs while the current chunk is
//    computed (software pipelining without LDS double buffers) -> LDS slab.
//  * A sweep is "lane owns a pencil": lane t reads its NIN-long pencil from LDS into registers,
//    multiplies by the NIN x NOUT basis and scatters the NOUT results so that the next sweep's pencils
//    are contiguous again.  Pencil stride in LDS is padded to an odd number of scalars -> conflict-free
//    reads.  Pencils of all EC elements are flattened over the lanes, so low orders fill the wave.
//    The input image and both intermediates live one after another in the SAME slab.
//  * The basis is wave-uniform: BASIS_SMEM fetches it row by row with scalar loads (s_load -> SGPR
//    operand of the FMA: no LDS traffic, no VGPRs, no barrier to stage it); BASIS_SMEM_COLS{,16} do the same
//    in column blocks of 8 / 16 scalars for rows that no longer fit the SGPR file (2D nq 11, 17..24);
//    BASIS_LDS keeps an LDS copy per workgroup (reference point: one ds_read per two FMAs).  contract() pins
//    the row-by-row software pipeline the compiler would otherwise flatten into one spill-heavy block.
//  * Line alignment (template MEMF bits 2 / 3): the lane -> 16-byte-vector mapping of the chunk load and of the
//    OUT_LDS output stream is shifted so that every wave-wide instruction covers whole 128-byte lines
//    (align_shift()); matters for chunks whose byte size is not a multiple of 128 (odd orders: +2-7 %).
//  * Output (template OUTM).  OUT_ST16 (even nq, fp64): last sweep has lane <-> (j,i), registers <-> k;
//    neighbouring lanes swap one value through DPP so the even lane owns out[k][j][i..i+1] and the odd lane
//    out[k+1][j][i-1..i]: every store instruction writes two whole nq^2 planes, 16 B per lane, straight
//    from registers.  OUT_LDS (any nq, any T): the chunk's output image is assembled in the slab and
//    leaves as one flat 16-B-per-lane stream -- what odd orders and the 2D kernel need (their direct
//    stores are fragments).  OUT_ST8: plain per-lane scalar stores (reference point).
//  * Scalar type: every kernel is a template on T (double = the reference's only instantiation; float =
//    the T the reference's templates allow but never instantiate).  "16-byte lane" = double2 or float4.
//
// Algorithmic HBM traffic per element: sizeof(T)*(nm^d + nq^d) bytes (in read once, out written once);
// measured traffic 1.003x that (profiles/hbm_traffic.json).
#pragma once

#include "sf_common.h"

namespace sf
{

// How the last sweep's results reach HBM.
enum OutMode
{
    OUT_ST8  = 0, // straight from registers, one scalar per lane (any nq)
    OUT_ST16 = 1, // straight from registers, lane pairs swap through DPP -> 16 B per lane (even nq, fp64)
    OUT_LDS  = 2  // through the LDS slab in final layout, then one flat 16-B-per-lane stream (any nq)
};

// How the wave-uniform basis operand is delivered to the FMAs.
enum BasisMode
{
    BASIS_LDS = 0, // broadcast ds_read from the workgroup's LDS copy
    BASIS_SMEM = 1, // scalar loads (s_load) from global memory -> SGPR operand
    // the same, but the row is consumed in column blocks of 8 / 16 scalars so that the operand ring
    // stays within the SGPR file at high order (a whole row of nq = 32 doubles would need 128 SGPRs)
    BASIS_SMEM_COLS = 2, // blocks of 8 scalars (one s_load_dwordx16 per row of an fp64 block)
    BASIS_SMEM_COLS16 = 3 // blocks of 16 scalars
};

// the 16-byte lane of a scalar type
template <typename T> struct VecOf
{
    static constexpr int W = 16 / (int)sizeof(T);
    typedef T type __attribute__((ext_vector_type(W)));
};

// fused multiply-add in the scalar type (calling the double builtin on floats converts both ways)
__device__ __forceinline__ double fma_t(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}
__device__ __forceinline__ float fma_t(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}

// exchange a double with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double swap_adjacent(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi     = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int NQ, int EC, int DIM, typename T = double> struct WaveGeom
{
    using Scalar = T;
    using Vec    = typename VecOf<T>::type;
    static constexpr int VW  = VecOf<T>::W; // scalars per 16-byte lane
    static constexpr int NM  = NQ - 1;
    static constexpr int NMP = NM | 1; // padded pencil stride (odd number of scalars)
    static constexpr int NMT = (DIM == 3) ? NM * NM * NM : NM * NM; // modes per element
    static constexpr int NQT = (DIM == 3) ? NQ * NQ * NQ : NQ * NQ; // points per element
    // input pencils keep the global layout when NM is odd (already conflict-free)
    static constexpr int IN_STRIDE = (NM % 2 == 0) ? NM + 1 : NM;
    static constexpr int IN_DBL    = EC * NMT; // scalars per chunk in HBM
    static constexpr bool VEC2     = (IN_DBL % VW) == 0; // chunk is a whole number of 16-B lanes
    // pencils per chunk in each sweep
    static constexpr int P0 = (DIM == 3) ? EC * NM * NM : EC * NM; // (e,r,q) | (e,q)
    static constexpr int P1 = (DIM == 3) ? EC * NQ * NM : EC * NQ; // (e,i,r) | (e,i)
    static constexpr int P2 = EC * NQ * NQ;                       // (e,j,i)   (3D only)
    static constexpr int PASS0 = cdiv(P0, kWave);
    static constexpr int PASS1 = cdiv(P1, kWave);
    static constexpr int PASS2 = cdiv(P2, kWave);
    // LDS slab per wave (scalars): max over the three images that live in it, one after another
    static constexpr int SLAB_IN = P0 * IN_STRIDE;
    static constexpr int SLAB_W1 = P1 * NMP;
    static constexpr int SLAB_W2 = (DIM == 3) ? P2 * NMP : 0;
    static constexpr int SLAB0   = CMax<CMax<SLAB_IN, SLAB_W1>::value, SLAB_W2>::value;
    static constexpr int OUT_DBL = EC * NQT; // scalars per chunk written to HBM
    // slab without / with room for the output image (OUT_LDS), kept 16-B aligned
    static constexpr int SLAB_NOOUT = (SLAB0 + VW - 1) / VW * VW;
    static constexpr int SLAB_OUT   = (CMax<SLAB0, OUT_DBL>::value + VW - 1) / VW * VW;
    static constexpr int NBAS = (NM * NQ + VW - 1) / VW * VW;
    static constexpr int NLD  = VEC2 ? cdiv(IN_DBL / VW, kWave) : cdiv(IN_DBL, kWave);
    // the chunk's 16-B lanes can be shifted by up to 7 so that every wave-wide load covers whole 128-B
    // lines; free when the shifted span needs no extra staging register
    static constexpr bool ALIGN_OK = VEC2 && cdiv(IN_DBL / VW + 7, kWave) == NLD;
};

// lanes to skip so that lane 0 of every load instruction sits on a 128-byte line
__device__ __forceinline__ int align_shift(const void *p)
{
    return __builtin_amdgcn_readfirstlane((int)(((uintptr_t)p >> 4) & 7));
}

template <class G, int OUTM> constexpr int slab_doubles()
{
    return OUTM == OUT_LDS ? G::SLAB_OUT : G::SLAB_NOOUT;
}

template <int NQ, int EC, int DIM, int WPB, int BMODE, int OUTM, typename T = double>
constexpr size_t wave_lds_bytes()
{
    using G = WaveGeom<NQ, EC, DIM, T>;
    return sizeof(T) *
           (size_t)((BMODE == BASIS_LDS ? DIM * G::NBAS : 0) + WPB * slab_doubles<G, OUTM>());
}

// chunk iteration space of one wave
struct ChunkIter
{
    uint64_t first, step, count;
};

template <int KMAP, int WPB, int XG = 0>
__device__ __forceinline__ ChunkIter chunk_iter(uint64_t nchunk, int wib)
{
    const uint64_t gw = logical_block<XG>() * WPB + wib;
    ChunkIter it;
    if constexpr (KMAP == 0)
    {
        const uint64_t nwave = (uint64_t)gridDim.x * WPB;
        it.first = gw;
        it.step  = nwave;
        it.count = gw < nchunk ? (nchunk - gw + nwave - 1) / nwave : 0;
    }
    else if constexpr (KMAP > 0)
    {
        it.first = gw * KMAP;
        it.step  = 1;
        it.count = it.first < nchunk ? (nchunk - it.first < KMAP ? nchunk - it.first : KMAP) : 0;
    }
    else
    {
        // KMAP < 0: the workgroup owns WPB*|KMAP| consecutive chunks and its waves interleave over them
        // (wave w takes chunks w, w+WPB, ...): at every step the workgroup touches one contiguous span
        constexpr int K     = -KMAP;
        const uint64_t base = logical_block<XG>() * WPB * K;
        it.first            = base + wib;
        it.step             = WPB;
        const uint64_t end  = base + (uint64_t)WPB * K < nchunk ? base + (uint64_t)WPB * K : nchunk;
        it.count            = it.first < end ? (end - it.first + WPB - 1) / WPB : 0;
    }
    return it;
}

// Word-grid accesses for a chunk whose base is only scalar-aligned (a number of scalars per chunk that is not a
// multiple of the 16-byte lane: one element of an even order in fp64, most single elements in fp32).  Lane
// k*64 + l owns the 16-byte word number k*64 + l of the 128-byte-line grid the chunk starts in: word w holds the
// scalars VW*w - a + {0 .. VW-1}, a = scalars between the line start and the chunk base.  Interior words are whole
// aligned 16-byte accesses and every wave-wide instruction covers whole lines; the words that straddle the two
// ends are completed with scalar accesses.
template <typename T> __device__ __forceinline__ int line_offset(const void *p)
{
    return __builtin_amdgcn_readfirstlane((int)(((uintptr_t)p / sizeof(T)) & (128 / sizeof(T) - 1)));
}
__device__ __forceinline__ int line_offset_f64(const void *p)
{
    return line_offset<double>(p);
}

template <int NMAX, typename T> constexpr int word_grid_regs()
{
    return cdiv(NMAX + (int)(128 / sizeof(T)) - 1, (int)(16 / sizeof(T)) * kWave);
}

template <int NMAX, int NREG, typename T>
__device__ __forceinline__ void chunk_load_any(typename VecOf<T>::type (&st)[NREG], const T *__restrict__ src,
                                               int lane, int nvalid)
{
    using V          = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    constexpr int NLDA = word_grid_regs<NMAX, T>();
    static_assert(NLDA <= NREG, "staging registers");
    const int a   = line_offset<T>(src);
    const V *grid = reinterpret_cast<const V *>(src - a);
#pragma unroll
    for (int k = 0; k < NLDA; ++k)
    {
        const int gv = k * kWave + lane;
        const int d0 = VW * gv - a;
        V x          = {};
        if (d0 >= 0 && d0 + VW - 1 < nvalid)
            x = __builtin_nontemporal_load(grid + gv);
        else
        {
#pragma unroll
            for (int h = 0; h < VW; ++h)
                if (d0 + h >= 0 && d0 + h < nvalid)
                    x[h] = src[d0 + h];
        }
        st[k] = x;
    }
}

template <int NMAX, int NREG>
__device__ __forceinline__ void chunk_load_any_f64(double2_t (&st)[NREG], const double *__restrict__ src,
                                                   int lane, int nvalid)
{
    chunk_load_any<NMAX, NREG, double>(st, src, lane, nvalid);
}

// A chunk loop requests chunk n+1 (loads into the staging registers), computes chunk n and stores it, then loops.
// Left alone, the wait for the staging registers lands at the loop header, where hipcc no longer knows how many
// stores were issued after the loads and emits s_waitcnt vmcnt(0): every wave then also waits for its own stores to
// reach memory before it may touch the next chunk.  Touching the staging registers at the END of the iteration, in
// the block that issued the stores, lets the counter be exact (vmcnt(number of stores): memory operations of a wave
// retire in order) and the header needs no wait at all.
template <typename V, int N> __device__ __forceinline__ void touch_staged(V (&st)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k)
        asm volatile("" : "+v"(st[k]));
}

// ------------------------------------------------------------------------------------------------
// chunk load: global -> staging registers (issued one chunk ahead of its use)
// ------------------------------------------------------------------------------------------------
template <class G, bool FULL, bool NTL = true, bool AL = false>
__device__ __forceinline__ void chunk_load(typename G::Vec (&st)[G::NLD],
                                           const typename G::Scalar *__restrict__ src, int lane,
                                           int nvalid /*scalars, only if !FULL*/)
{
    using T = typename G::Scalar;
    using V = typename G::Vec;
    constexpr int VW = G::VW;
    if constexpr (G::VEC2)
    {
        const V *srcv = reinterpret_cast<const V *>(src);
        const int sh  = AL ? align_shift(src) : 0;
#pragma unroll
        for (int k = 0; k < G::NLD; ++k)
        {
            const int v = k * kWave + lane - sh;
            if constexpr (FULL)
            {
                if (AL ? (v >= 0 && v < G::IN_DBL / VW)
                       : ((k + 1) * kWave <= G::IN_DBL / VW || v < G::IN_DBL / VW))
                    st[k] = NTL ? __builtin_nontemporal_load(srcv + v) : srcv[v];
            }
            else
            {
                V x = {};
                if (AL && v < 0)
                    ;
                else if (VW * v + VW - 1 < nvalid)
                    x = NTL ? __builtin_nontemporal_load(srcv + v) : srcv[v];
                else
                {
#pragma unroll
                    for (int j = 0; j < VW - 1; ++j) // ragged tail of the last partial chunk
                        if (VW * v + j < nvalid)
                            x[j] = src[VW * v + j];
                }
                st[k] = x;
            }
        }
    }
    else
    {
        // scalars per chunk not a multiple of the 16-byte lane (chunk bases only scalar-aligned): word-grid load
        chunk_load_any<G::IN_DBL, G::NLD, T>(st, src, lane, FULL ? G::IN_DBL : nvalid);
    }
}

// staging registers -> LDS slab, pencil stride IN_STRIDE
template <class G, bool AL = false>
__device__ __forceinline__ void chunk_stage(const typename G::Vec (&st)[G::NLD],
                                            typename G::Scalar *slab, int lane, int sh = 0)
{
    using V = typename G::Vec;
    constexpr int VW = G::VW;
    if constexpr (G::VEC2)
    {
#pragma unroll
        for (int k = 0; k < G::NLD; ++k)
        {
            const int v = k * kWave + lane - (AL ? sh : 0);
            if (AL ? (v >= 0 && v < G::IN_DBL / VW)
                   : ((k + 1) * kWave <= G::IN_DBL / VW || v < G::IN_DBL / VW))
            {
                if constexpr (G::IN_STRIDE == G::NM)
                {
                    *reinterpret_cast<V *>(slab + VW * v) = st[k];
                }
                else
                {
#pragma unroll
                    for (int j = 0; j < VW; ++j)
                    {
                        const int f          = VW * v + j;
                        slab[f + f / G::NM] = st[k][j];
                    }
                }
            }
        }
    }
    else
    {
        // word-grid registers (chunk_load_any): word k*64 + lane holds scalars VW*(k*64 + lane) - sh + {0..VW-1},
        // sh = line_offset(chunk base)
#pragma unroll
        for (int k = 0; k < word_grid_regs<G::IN_DBL, typename G::Scalar>(); ++k)
#pragma unroll
            for (int h = 0; h < VW; ++h)
            {
                const int f = VW * (k * kWave + lane) - sh + h;
                if (f >= 0 && f < G::IN_DBL)
                {
                    if constexpr (G::IN_STRIDE == G::NM)
                        slab[f] = st[k][h];
                    else
                        slab[f + f / G::NM] = st[k][h];
                }
            }
    }
}

// One basis row of a contraction: acc[pass][N0 + n] (+)= u[pass][m] * b[n], n < NB (m == 0 starts the sums).
// T = double: one v_fma_f64 per value.  T = float: neighbouring outputs n, n + 1 share one v_pk_fma_f32 -- the pencil
// value is broadcast to both halves (op_sel), the two basis entries are an SGPR (or VGPR) pair -- so the fp32 sweeps
// issue half the vector instructions; from 2D nq = 17 / 3D nq = 11 they are issue-bound, not memory-bound
// (profiles/r02/sweep_auto_f32.log).  Same products, same ascending order of m: results do not change.
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int NB, int N0, int NOUT, int NPASS>
__device__ __forceinline__ void fma_row(bool first, const double (&um)[NPASS], const double (&b)[NB],
                                        double (&acc)[NPASS][NOUT])
{
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int s = 0; s < NPASS; ++s)
            acc[s][N0 + n] = first ? um[s] * b[n] : fma_t(um[s], b[n], acc[s][N0 + n]);
}
template <int NB, int N0, int NOUT, int NPASS>
__device__ __forceinline__ void fma_row(bool first, const float (&um)[NPASS], const float (&b)[NB],
                                        float (&acc)[NPASS][NOUT])
{
    // measured per order (profiles/r03/sweep_auto_f32_packed.log against r02's): +7..14 % where the sweeps are issue-bound
    // (2D nq 17..24, 3D nq 9..16), neutral below, -3..8 % at 2D nq 25..31 (two-element chunks at full occupancy)
    constexpr int NPAIR = (NOUT >= 9 && NOUT <= 24) ? NB / 2 : 0;
#pragma unroll
    for (int n = 0; n < 2 * NPAIR; n += 2)
#pragma unroll
        for (int s = 0; s < NPASS; ++s)
        {
            const float2_t uu = {um[s], um[s]}, bb = {b[n], b[n + 1]};
            const float2_t a  = {acc[s][N0 + n], acc[s][N0 + n + 1]};
            const float2_t r  = first ? uu * bb : __builtin_elementwise_fma(uu, bb, a);
            acc[s][N0 + n]     = r.x;
            acc[s][N0 + n + 1] = r.y;
        }
#pragma unroll
    for (int n = 2 * NPAIR; n < NB; ++n)
#pragma unroll
        for (int s = 0; s < NPASS; ++s)
            acc[s][N0 + n] = first ? um[s] * b[n] : fma_t(um[s], b[n], acc[s][N0 + n]);
}

// ------------------------------------------------------------------------------------------------
// BASIS_SMEM_COLS: columns [N0, N0+NB) of every basis row through a two-deep SGPR ring, then the next
// column block (same touch / request / FMA / fence order per row as contract() below)
// ------------------------------------------------------------------------------------------------
template <int NIN, int NOUT, int NPASS, int N0, int kColBlock, typename T>
__device__ __forceinline__ void contract_cols(const T (&u)[NPASS][NIN], T (&acc)[NPASS][NOUT],
                                              const T *__restrict__ bas, int &zero)
{
    constexpr int NB = (NOUT - N0) < kColBlock ? (NOUT - N0) : kColBlock;
    T b[2][NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
        b[0][n] = bas[zero + N0 + n];
#pragma unroll
    for (int m = 0; m < NIN; ++m)
    {
#pragma unroll
        for (int n = 0; n < NB; ++n)
            asm volatile("" : "+s"(zero) : "s"(b[m % 2][n]));
        if (m + 1 < NIN)
        {
#pragma unroll
            for (int n = 0; n < NB; ++n)
                b[(m + 1) % 2][n] = bas[zero + (m + 1) * NOUT + N0 + n];
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            T um[NPASS];
#pragma unroll
            for (int s = 0; s < NPASS; ++s)
                um[s] = u[s][m];
            fma_row<NB, N0>(m == 0, um, b[m % 2], acc);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int s = 0; s < NPASS; ++s)
                asm volatile("" : "+s"(zero) : "v"(acc[s][N0 + n]));
    }
    if constexpr (N0 + NB < NOUT)
        contract_cols<NIN, NOUT, NPASS, N0 + NB, kColBlock, T>(u, acc, bas, zero);
}

// ------------------------------------------------------------------------------------------------
// one contraction: acc[pass][n] = sum_m u[pass][m] * B[m*NOUT + n], ascending m, start at 0
// ------------------------------------------------------------------------------------------------
template <int NIN, int NOUT, int NPASS, int BMODE, typename T>
__device__ __forceinline__ void contract(const T (&u)[NPASS][NIN], T (&acc)[NPASS][NOUT],
                                         const T *__restrict__ bas)
{
    if constexpr (BMODE == BASIS_SMEM_COLS || BMODE == BASIS_SMEM_COLS16)
    {
        int z = 0;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        contract_cols<NIN, NOUT, NPASS, 0, (BMODE == BASIS_SMEM_COLS ? 8 : 16), T>(u, acc, bas, z);
        return;
    }
    // The basis is consumed one ROW (fixed m, all n) at a time, with the next row's operands
    // requested before the current row's FMAs and an order fence after them: left alone, hipcc
    // hoists every basis load of the sweep (and of later sweeps) to the top and then spills.
    // The offset is opaque (not the pointer: the pointer must stay a provably global, unclobbered
    // kernel argument to get s_load) so the loads also stay inside the chunk loop.
    int zero = 0;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    // Ring of operand rows, one row ahead.  Scalar loads return OUT OF ORDER, so the wait for row m is
    // always lgkmcnt(0): it would also wait for a row requested just before it.  Hence the order per
    // row: (1) touch row m (the wait lands here, covering only loads issued a whole row of FMAs ago),
    // (2) request row m+1, (3) the FMAs of row m (the new request is in flight underneath them).
    constexpr int RING = 2;
    T b[RING][NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
        b[0][n] = bas[zero + n];
#pragma unroll
    for (int m = 0; m < NIN; ++m)
    {
        if constexpr (BMODE == BASIS_SMEM)
        {
#pragma unroll
            for (int n = 0; n < NOUT; ++n)
                asm volatile("" : "+s"(zero) : "s"(b[m % RING][n]));
        }
        if (m + 1 < NIN)
        {
#pragma unroll
            for (int n = 0; n < NOUT; ++n)
                b[(m + 1) % RING][n] = bas[zero + (m + 1) * NOUT + n];
            // keep the request ABOVE this row's FMAs (the machine scheduler otherwise sinks it to just
            // before the next wait and the latency is exposed again)
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            T um[NPASS];
#pragma unroll
            for (int s = 0; s < NPASS; ++s)
                um[s] = u[s][m];
            fma_row<NOUT, 0>(m == 0, um, b[m % RING], acc);
        }
        // order fence: the next row's operand loads (addressed through `zero`) may not be issued
        // before this row's FMAs, and this row's FMAs may not sink below them
#pragma unroll
        for (int n = 0; n < NOUT; ++n)
#pragma unroll
            for (int s = 0; s < NPASS; ++s)
                asm volatile("" : "+s"(zero) : "v"(acc[s][n]));
    }
}

// read the pencils of this lane: u[pass][m] = slab[t*STRIDE + m], t = pass*64 + lane < NP
template <int NIN, int NPASS, int NP, int STRIDE, typename T>
__device__ __forceinline__ void read_pencils(T (&u)[NPASS][NIN], const T *slab, int lane)
{
#pragma unroll
    for (int s = 0; s < NPASS; ++s)
    {
        int t = s * kWave + lane;
        if ((s + 1) * kWave > NP) // partial pass: idle lanes re-read the last pencil (no branch)
            t = t < NP ? t : NP - 1;
#pragma unroll
        for (int m = 0; m < NIN; ++m)
            u[s][m] = slab[t * STRIDE + m];
    }
}

// ------------------------------------------------------------------------------------------------
// shared prologue: basis pointers (LDS copy or the global arrays themselves) and the wave's slab
// ------------------------------------------------------------------------------------------------
template <class G, int DIM, int WPB, int BMODE, int SLAB, typename T>
__device__ __forceinline__ T *wave_setup(T *lds, const T *const (&gb)[3], const T *(&bs)[3], int wib)
{
    if constexpr (BMODE == BASIS_LDS)
    {
        for (int x = threadIdx.x; x < G::NM * (G::NM + 1); x += kWave * WPB)
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                lds[d * G::NBAS + x] = gb[d][x];
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            bs[d] = lds + d * G::NBAS;
        return lds + DIM * G::NBAS + wib * SLAB;
    }
    else
    {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            bs[d] = gb[d];
        return lds + wib * SLAB;
    }
}

template <class G, int EC, bool NTL = true, bool AL = false>
__device__ __forceinline__ void chunk_fetch(typename G::Vec (&st)[G::NLD],
                                            const typename G::Scalar *__restrict__ in, uint64_t c,
                                            uint64_t nelmt, int lane)
{
    const uint64_t left = nelmt - c * EC;
    if (left >= EC)
        chunk_load<G, true, NTL, AL>(st, in + c * G::IN_DBL, lane, 0);
    else
        chunk_load<G, false, NTL, AL>(st, in + c * G::IN_DBL, lane, (int)left * G::NMT);
}

// Final-sweep store of one pass: lane t owns NOUT values acc[n] destined for dst[n*NSTRIDE]
// (consecutive lanes -> consecutive scalars).  ST16 (fp64): lane pairs exchange so each lane stores 16 B.
template <int NOUT, int NSTRIDE, bool ST16, bool NTS = true, typename T>
__device__ __forceinline__ void store_column(const T (&acc)[NOUT], T *dst, int lane)
{
    if constexpr (ST16)
    {
        static_assert(sizeof(T) == 8, "paired 16-byte stores are the fp64 path");
        static_assert(NOUT % 2 == 0 && NSTRIDE % 2 == 0, "16-byte stores need even extents");
        const bool odd = lane & 1;
        double *d2     = dst - (odd ? 1 : 0) + (odd ? NSTRIDE : 0);
#pragma unroll
        for (int n = 0; n < NOUT; n += 2)
        {
            const double give = odd ? acc[n] : acc[n + 1];
            const double got  = swap_adjacent(give);
            double2_t v;
            v.x = odd ? got : acc[n];
            v.y = odd ? acc[n + 1] : got;
            if (NTS)
                __builtin_nontemporal_store(v, reinterpret_cast<double2_t *>(d2 + n * NSTRIDE));
            else
                *reinterpret_cast<double2_t *>(d2 + n * NSTRIDE) = v;
        }
    }
    else
    {
#pragma unroll
        for (int n = 0; n < NOUT; ++n)
        {
            if (NTS)
                __builtin_nontemporal_store(acc[n], dst + n * NSTRIDE);
            else
                dst[n * NSTRIDE] = acc[n];
        }
    }
}

// Flat stream LDS -> HBM for a destination that is only scalar-aligned (see the word-grid note above):
// whole aligned 16-byte stores on the 128-byte-line grid, scalar stores for the words straddling the two ends.
// LDS side: VW scalar reads per lane (any scalar alignment).
template <int NMAX, typename T>
__device__ __forceinline__ void flush_any(const T *img, T *__restrict__ dst, int nout, int lane)
{
    using V          = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    const int a      = line_offset<T>(dst);
    V *grid          = reinterpret_cast<V *>(dst - a);
    constexpr int NST = word_grid_regs<NMAX, T>();
#pragma unroll
    for (int k = 0; k < NST; ++k)
    {
        const int gv = k * kWave + lane;
        const int d0 = VW * gv - a;
        if (d0 >= 0 && d0 + VW - 1 < nout)
        {
            V x;
#pragma unroll
            for (int h = 0; h < VW; ++h)
                x[h] = img[d0 + h];
            __builtin_nontemporal_store(x, grid + gv);
        }
        else
        {
#pragma unroll
            for (int h = 0; h < VW; ++h)
                if (d0 + h >= 0 && d0 + h < nout)
                    dst[d0 + h] = img[d0 + h];
        }
    }
}

template <int NMAX>
__device__ __forceinline__ void flush_any_f64(const double *img, double *__restrict__ dst, int nout, int lane)
{
    flush_any<NMAX, double>(img, dst, nout, lane);
}

// OUT_LDS epilogue: the slab holds the chunk's output in final layout; stream `nout` scalars to HBM
// with 16 B per lane (chunk bases are 16-B aligned when OUT_DBL is a multiple of VW; else scalar lanes).
template <class G, bool NTS, bool AL = false>
__device__ __forceinline__ void chunk_flush(const typename G::Scalar *slab,
                                            typename G::Scalar *__restrict__ dst, int nout, int lane)
{
    using V = typename G::Vec;
    constexpr int VW = G::VW;
    if constexpr (G::OUT_DBL % VW == 0)
    {
        // AL: lanes shifted so that every wave-wide store covers whole 128-byte lines (one more
        // instruction at most; matters when the chunk's output is not a multiple of 128 B, i.e. odd nq)
        constexpr int NST = cdiv(G::OUT_DBL / VW + (AL ? 7 : 0), kWave);
        V *dstv           = reinterpret_cast<V *>(dst);
        const int sh      = AL ? align_shift(dst) : 0;
#pragma unroll
        for (int k = 0; k < NST; ++k)
        {
            const int v = k * kWave + lane - sh;
            if (AL && v < 0)
                continue;
            if (VW * v + VW - 1 < nout)
            {
                const V x = *reinterpret_cast<const V *>(slab + VW * v);
                if (NTS)
                    __builtin_nontemporal_store(x, dstv + v);
                else
                    dstv[v] = x;
            }
            else
            {
#pragma unroll
                for (int j = 0; j < VW - 1; ++j)
                    if (VW * v + j < nout)
                        dst[VW * v + j] = slab[VW * v + j];
            }
        }
    }
    else
    {
        // scalars per chunk not a multiple of the 16-byte lane: chunk bases are only scalar-aligned
        flush_any<G::OUT_DBL, typename G::Scalar>(slab, dst, nout, lane);
    }
}

// Workgroup-cooperative input (MEMF bit 16, K = 1 mappings): the WPB chunks of a workgroup are contiguous, so all
// its lanes load them as ONE flat word-grid stream -- every 128-byte line of the workgroup's input is fetched once,
// the words straddling two elements included -- and scatter the words into the owners' slabs; one workgroup
// barrier, then every wave computes its own chunk as usual.
template <class G, int WPB, int SLAB>
__device__ __forceinline__ void block_load_stage(typename G::Scalar *slab0, const typename G::Scalar *__restrict__ src,
                                                 int nvalid)
{
    using T          = typename G::Scalar;
    using V          = typename G::Vec;
    constexpr int VW = G::VW;
    constexpr int NB = WPB * G::IN_DBL;
    constexpr int NT = WPB * kWave;
    constexpr int NW = cdiv(cdiv(NB + (int)(128 / sizeof(T)) - 1, VW), NT);
    const int a      = line_offset<T>(src);
    const V *grid    = reinterpret_cast<const V *>(src - a);
    V x[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k)
    {
        const int gv = k * NT + (int)threadIdx.x;
        const int d0 = VW * gv - a;
        V v          = {};
        if (d0 >= 0 && d0 + VW - 1 < nvalid)
            v = __builtin_nontemporal_load(grid + gv);
        else
        {
#pragma unroll
            for (int h = 0; h < VW; ++h)
                if (d0 + h >= 0 && d0 + h < nvalid)
                    v[h] = src[d0 + h];
        }
        x[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int h = 0; h < VW; ++h)
        {
            const int f = VW * (k * NT + (int)threadIdx.x) - a + h;
            if (f >= 0 && f < NB)
            {
                const int w = f / G::IN_DBL, r = f - w * G::IN_DBL;
                T *dst      = slab0 + w * SLAB;
                if constexpr (G::IN_STRIDE == G::NM)
                    dst[r] = x[k][h];
                else
                    dst[r + r / G::NM] = x[k][h];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// 3D hex
// ------------------------------------------------------------------------------------------------
template <int NQ, int EC, int WPB, int BMODE, int MINW, int KMAP, int OUTM, int MEMF = 0,
          typename T = double>
__global__ __launch_bounds__(kWave *WPB, MINW) void hex_wave_kernel(
    const T *__restrict__ b0, const T *__restrict__ b1, const T *__restrict__ b2,
    const T *__restrict__ in, T *__restrict__ out, uint64_t nelmt)
{
    using G          = WaveGeom<NQ, EC, 3, T>;
    constexpr int NM = G::NM, NMP = G::NMP, NM2 = NM * NM, NQ2 = NQ * NQ;
    static_assert(OUTM != OUT_ST16 || (NQ % 2 == 0 && sizeof(T) == 8),
                  "paired 16-byte stores need even nq and fp64");
    constexpr int SLAB = slab_doubles<G, OUTM>();

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T *const gb[3] = {b0, b1, b2};
    const T *bs[3];
    T *slab = wave_setup<G, 3, WPB, BMODE, SLAB>(lds, gb, bs, wib);

    constexpr int XG  = (MEMF >> 4) & 0xfff;
    constexpr bool BL = ((MEMF >> 16) & 1) != 0; // workgroup-cooperative input
    static_assert(!BL || (KMAP == 1 && BMODE != BASIS_LDS), "cooperative input: one chunk per wave, basis in SGPRs");
    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const ChunkIter it    = chunk_iter<KMAP, WPB, XG>(nchunk, wib);
    if constexpr (BL)
    {
        const uint64_t first = logical_block<XG>() * (uint64_t)(WPB * EC); // first element of the workgroup
        if (first < nelmt)
        {
            const uint64_t left = nelmt - first;
            block_load_stage<G, WPB, SLAB>(lds, in + first * G::NMT,
                                           left >= (uint64_t)(WPB * EC) ? WPB * G::IN_DBL : (int)left * G::NMT);
        }
        __syncthreads();
    }
    if (it.count == 0)
        return;

    constexpr bool AL = (MEMF & 4) && G::ALIGN_OK;
    typename G::Vec st[BL ? 1 : G::NLD];
    if constexpr (!BL)
        chunk_fetch<G, EC, !(MEMF & 1), AL>(st, in, it.first, nelmt, lane);

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        const uint64_t left = nelmt - c * EC;
        const int evalid    = left >= EC ? EC : (int)left;

        if constexpr (!BL)
        {
            chunk_stage<G, AL>(st, slab, lane,
                               G::VEC2 ? (AL ? align_shift(in + c * G::IN_DBL) : 0)
                                       : line_offset<T>(in + c * G::IN_DBL));
            wave_lds_fence();

            // request the next chunk of this wave now; it lands in the staging registers while this
            // chunk is being computed
            if (n + 1 < it.count)
                chunk_fetch<G, EC, !(MEMF & 1), AL>(st, in, c + it.step, nelmt, lane);
        }

        // ---- direction 0: w1[(e,i,r)][q] = sum_p in[(e,r,q)][p] * B0[p][i] ----------------------
        {
            T u[G::PASS0][NM], acc[G::PASS0][NQ];
            read_pencils<NM, G::PASS0, G::P0, G::IN_STRIDE>(u, slab, lane);
            contract<NM, NQ, G::PASS0, BMODE>(u, acc, bs[0]);
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < G::PASS0; ++s)
            {
                const int t = s * kWave + lane;
                if ((s + 1) * kWave <= G::P0 || t < G::P0)
                {
                    const int e = t / NM2, rq = t - e * NM2, r = rq / NM, q = rq - r * NM;
                    T *dst = slab + (e * NQ * NM + r) * NMP + q;
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        dst[i * NM * NMP] = acc[s][i];
                }
            }
            wave_lds_fence();
        }
        // ---- direction 1: w2[(e,j,i)][r] = sum_q w1[(e,i,r)][q] * B1[q][j] ----------------------
        {
            T u[G::PASS1][NM], acc[G::PASS1][NQ];
            read_pencils<NM, G::PASS1, G::P1, NMP>(u, slab, lane);
            contract<NM, NQ, G::PASS1, BMODE>(u, acc, bs[1]);
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < G::PASS1; ++s)
            {
                const int t = s * kWave + lane;
                if ((s + 1) * kWave <= G::P1 || t < G::P1)
                {
                    const int e = t / (NQ * NM), ir = t - e * (NQ * NM), i = ir / NM,
                              r = ir - i * NM;
                    T *dst = slab + (e * NQ2 + i) * NMP + r;
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
                        dst[j * NQ * NMP] = acc[s][j];
                }
            }
            wave_lds_fence();
        }
        // ---- direction 2: out[e][k][(j,i)] = sum_r w2[(e,j,i)][r] * B2[r][k] --------------------
        {
            T u[G::PASS2][NM], acc[G::PASS2][NQ];
            read_pencils<NM, G::PASS2, G::P2, NMP>(u, slab, lane);
            contract<NM, NQ, G::PASS2, BMODE>(u, acc, bs[2]);
            T *oc = out + c * (uint64_t)(EC * G::NQT);
            if constexpr (OUTM == OUT_LDS)
            {
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < G::PASS2; ++s)
                {
                    const int t = s * kWave + lane;
                    if ((s + 1) * kWave <= G::P2 || t < G::P2)
                    {
                        const int e = t / NQ2, pl = t - e * NQ2;
                        T *dst = slab + e * G::NQT + pl;
#pragma unroll
                        for (int k = 0; k < NQ; ++k)
                            dst[k * NQ2] = acc[s][k];
                    }
                }
                wave_lds_fence();
                chunk_flush<G, !(MEMF & 2), (MEMF & 8) != 0>(slab, oc, evalid * G::NQT, lane);
            }
            else
            {
#pragma unroll
                for (int s = 0; s < G::PASS2; ++s)
                {
                    const int t = s * kWave + lane;
                    const int e = t / NQ2, pl = t - e * NQ2;
                    if (((s + 1) * kWave <= G::P2 || t < G::P2) && e < evalid)
                        store_column<NQ, NQ2, OUTM == OUT_ST16, !(MEMF & 2)>(
                            acc[s], oc + e * G::NQT + pl, lane);
                }
            }
            wave_lds_fence(); // slab is rewritten by the next chunk's staging
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2D quad
// ------------------------------------------------------------------------------------------------
template <int NQ, int EC, int WPB, int BMODE, int MINW, int KMAP, int OUTM, int MEMF = 0,
          typename T = double>
__global__ __launch_bounds__(kWave *WPB, MINW) void quad_wave_kernel(
    const T *__restrict__ b0, const T *__restrict__ b1, const T *__restrict__ in,
    T *__restrict__ out, uint64_t nelmt)
{
    using G          = WaveGeom<NQ, EC, 2, T>;
    constexpr int NM = G::NM, NMP = G::NMP;
    static_assert(OUTM != OUT_ST16 || (NQ % 2 == 0 && sizeof(T) == 8),
                  "paired 16-byte stores need even nq and fp64");
    constexpr int SLAB = slab_doubles<G, OUTM>();

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T *const gb[3] = {b0, b1, nullptr};
    const T *bs[3];
    T *slab = wave_setup<G, 2, WPB, BMODE, SLAB>(lds, gb, bs, wib);

    const uint64_t nchunk = (nelmt + EC - 1) / EC;
    const ChunkIter it    = chunk_iter<KMAP, WPB, ((MEMF >> 4) & 0xfff)>(nchunk, wib);
    if (it.count == 0)
        return;

    constexpr bool AL = (MEMF & 4) && G::ALIGN_OK;
    typename G::Vec st[G::NLD];
    chunk_fetch<G, EC, !(MEMF & 1), AL>(st, in, it.first, nelmt, lane);

    uint64_t c = it.first;
    for (uint64_t n = 0; n < it.count; ++n, c += it.step)
    {
        const uint64_t left = nelmt - c * EC;
        const int evalid    = left >= EC ? EC : (int)left;

        chunk_stage<G, AL>(st, slab, lane,
                           G::VEC2 ? (AL ? align_shift(in + c * G::IN_DBL) : 0)
                                   : line_offset<T>(in + c * G::IN_DBL));
        wave_lds_fence();
        if (n + 1 < it.count)
            chunk_fetch<G, EC, !(MEMF & 1), AL>(st, in, c + it.step, nelmt, lane);

        // ---- direction 0: w[(e,i)][q] = sum_p in[(e,q)][p] * B0[p][i] ---------------------------
        {
            T u[G::PASS0][NM], acc[G::PASS0][NQ];
            read_pencils<NM, G::PASS0, G::P0, G::IN_STRIDE>(u, slab, lane);
            contract<NM, NQ, G::PASS0, BMODE>(u, acc, bs[0]);
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < G::PASS0; ++s)
            {
                const int t = s * kWave + lane;
                if ((s + 1) * kWave <= G::P0 || t < G::P0)
                {
                    const int e = t / NM, q = t - e * NM;
                    T *dst = slab + e * NQ * NMP + q;
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        dst[i * NMP] = acc[s][i];
                }
            }
            wave_lds_fence();
        }
        // ---- direction 1: out[e][j][i] = sum_q w[(e,i)][q] * B1[q][j] ---------------------------
        {
            T u[G::PASS1][NM], acc[G::PASS1][NQ];
            read_pencils<NM, G::PASS1, G::P1, NMP>(u, slab, lane);
            contract<NM, NQ, G::PASS1, BMODE>(u, acc, bs[1]);
            T *oc = out + c * (uint64_t)(EC * G::NQT);
            if constexpr (OUTM == OUT_LDS)
            {
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < G::PASS1; ++s)
                {
                    const int t = s * kWave + lane;
                    if ((s + 1) * kWave <= G::P1 || t < G::P1)
                    {
                        const int e = t / NQ, i = t - e * NQ;
                        T *dst = slab + e * G::NQT + i;
#pragma unroll
                        for (int j = 0; j < NQ; ++j)
                            dst[j * NQ] = acc[s][j];
                    }
                }
                wave_lds_fence();
                chunk_flush<G, !(MEMF & 2), (MEMF & 8) != 0>(slab, oc, evalid * G::NQT, lane);
            }
            else
            {
#pragma unroll
                for (int s = 0; s < G::PASS1; ++s)
                {
                    const int t = s * kWave + lane;
                    const int e = t / NQ, i = t - e * NQ;
                    if (((s + 1) * kWave <= G::P1 || t < G::P1) && e < evalid)
                        store_column<NQ, NQ, OUTM == OUT_ST16, !(MEMF & 2)>(
                            acc[s], oc + e * G::NQT + i, lane);
                }
            }
            wave_lds_fence();
        }
    }
}

} // namespace sf
