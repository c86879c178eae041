// bwdtrans_rt.h -- 3D hex BwdTrans with RUN-TIME extents nq0 != nq1 != nq2 (each <= 16): the wave-per-chunk design of
// bwdtrans_wave.h for the shapes its compile-time tables do not hold.
//
// The reference kernels take nq0, nq1, nq2 at run time (benchmark05/benchmark05.cc:291-297) and its harness sizes the
// shared memory for the isotropic case only (the `ssize` expressions, :1316-1318) -- anisotropic extents are where that
// bug class lives.  Before this kernel every anisotropic call took the barrier-per-sweep block kernel
// (bwdtrans_generic.hip, 0.34 of the HBM roofline at 8x8x8); this one keeps what makes the isotropic flagship fast:
//   * one WAVEFRONT owns a chunk of `ec` consecutive elements, no workgroup barrier in the element path;
//   * 16-byte non-temporal loads / stores on the 16-byte word grid of the chunk (chunks of an odd number of doubles
//     start 8-byte aligned: the two end words are completed with 8-byte accesses);
//   * "lane owns a pencil": three sweeps (p, q, r -- the reference's order, ascending sums from the first product),
//     pencils of all `ec` elements flattened over the lanes, padded odd pencil strides (conflict-free LDS reads);
//   * workgroups renumbered in XCD runs (sf_common.h).
//   * the wave-uniform basis rows are SGPR operands of the FMAs (scalar BUFFER loads from the caller's arrays: range
//     checked, so a row of NB >= nq columns can be requested without ever reading outside the array; a first version
//     with zero-padded LDS copies and broadcast ds_reads was LDS-bound at 0.30-0.46 of the roofline).
// What run-time extents cost: the loops are unrolled to the compile-time bound NB >= max(nq) (even buckets 2..16) with
// wave-uniform guards per basis row (FMAs on the columns between nq and NB are wasted), and the images ping-pong
// between two LDS regions instead of being rewritten in place (a sweep's pencils cannot all be parked in registers
// when their count is a run-time number).
#pragma once

#include "bwdtrans_wave.h"

namespace sf
{

// everything the kernel needs about the shape, computed once on the host (wave-uniform: lives in SGPRs)
struct RtShape
{
    int nq0, nq1, nq2, nm0, nm1, nm2;
    int ec;             // elements per chunk (one chunk per wave)
    int s0, s1, s2;     // padded pencil strides of the three images (odd)
    int nmt, nqt;       // modes / points per element
    int off_b;          // image B starts here in the wave's slab (doubles); image A at 0
    int slab;           // doubles per wave
    // t / d for t < 65536 as (t * magic) >> 32
    unsigned m_nm0, m_nm1, m_nm12, m_nm2, m_nq0nm2, m_nq01;
};

// magic = ceil(2^32 / d) for d >= 2 (exact for t < 65536, d <= 4096); 0 stands for d = 1
__device__ __forceinline__ int div_magic(int t, unsigned magic)
{
    return magic ? (int)__umulhi((unsigned)t, magic) : t;
}

// ---- wave-uniform basis rows as SGPR operands, read through a BUFFER descriptor ---------------------------------------
// A row of NB columns is read from the caller's nm x nq array starting at column n0 of row m -- with nq < NB that runs
// into the next row (finite values that only reach accumulators nobody stores) and, for the last row, past the end of
// the array.  A scalar BUFFER load range-checks every dword against the descriptor's byte count and returns 0 beyond
// it, so no read ever leaves the array and no guard is needed.  hipcc has no builtin for s_buffer_load: inline asm, and
// therefore explicit waits (a wait for lgkmcnt(0) also covers the compiler's own LDS reads: conservative, never early).
typedef double d2v_t __attribute__((ext_vector_type(2)));
typedef double d4v_t __attribute__((ext_vector_type(4)));
typedef double d8v_t __attribute__((ext_vector_type(8)));
constexpr unsigned kRsrcWord3 = 0x00020000u; // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)

template <int D> struct SRow; // D doubles of one basis row in SGPRs, D in {2, 4, 6, 8}
template <> struct SRow<2>
{
    d2v_t a;
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, int off)
    {
        asm volatile("s_buffer_load_dwordx4 %0, %1, %2" : "=&s"(a) : "s"(r), "s"(off));
    }
    __device__ __forceinline__ void wait()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    }
    __device__ __forceinline__ double get(int n) const
    {
        return a[n];
    }
};
template <> struct SRow<4>
{
    d4v_t a;
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, int off)
    {
        asm volatile("s_buffer_load_dwordx8 %0, %1, %2" : "=&s"(a) : "s"(r), "s"(off));
    }
    __device__ __forceinline__ void wait()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    }
    __device__ __forceinline__ double get(int n) const
    {
        return a[n];
    }
};
template <> struct SRow<8>
{
    d8v_t a;
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, int off)
    {
        asm volatile("s_buffer_load_dwordx16 %0, %1, %2" : "=&s"(a) : "s"(r), "s"(off));
    }
    __device__ __forceinline__ void wait()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    }
    __device__ __forceinline__ double get(int n) const
    {
        return a[n];
    }
};
template <> struct SRow<6>
{
    SRow<4> lo;
    SRow<2> hi;
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, int off)
    {
        lo.issue(r, off);
        hi.issue(r, off + 32);
    }
    __device__ __forceinline__ void wait()
    {
        lo.wait();
        hi.wait();
    }
    __device__ __forceinline__ double get(int n) const
    {
        return n < 4 ? lo.get(n) : hi.get(n - 4);
    }
};

// columns [N0, N0 + D) of acc[g][n] = sum_{m < nm} u[g][m] * B[m][n] for the G pencils a lane owns in this group of
// passes, ascending m, the first product starts the sum.  Rows go through a two-deep ring: wait for row m, request row
// m + 1, run row m's FMAs (all G pencils: a row is fetched once per group, not once per pass) on top of the request.
template <int NB, int G, int N0, int D>
__device__ __forceinline__ void contract_rt_cols(const double (&u)[G][NB - 1], double (&acc)[G][NB],
                                                 __amdgpu_buffer_rsrc_t rsrc, int nm, int nq, int ng)
{
    SRow<D> b[2];
    b[0].issue(rsrc, 8 * N0);
#pragma unroll
    for (int m = 0; m < NB - 1; ++m)
    {
        if (m < nm) // wave-uniform
        {
            b[m % 2].wait();
            if (m + 1 < nm)
                b[(m + 1) % 2].issue(rsrc, 8 * ((m + 1) * nq + N0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (g < ng) // wave-uniform: passes of this group that hold pencils
                {
#pragma unroll
                    for (int n = 0; n < D; ++n)
                        acc[g][N0 + n] = (m == 0) ? u[g][0] * b[0].get(n) : fma_t(u[g][m], b[m % 2].get(n), acc[g][N0 + n]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NB, int G>
__device__ __forceinline__ void contract_rt(const double (&u)[G][NB - 1], double (&acc)[G][NB], __amdgpu_buffer_rsrc_t rsrc,
                                            int nm, int nq, int ng)
{
    if constexpr (NB <= 8)
        contract_rt_cols<NB, G, 0, NB>(u, acc, rsrc, nm, nq, ng);
    else
    {
        contract_rt_cols<NB, G, 0, 8>(u, acc, rsrc, nm, nq, ng);
        contract_rt_cols<NB, G, 8, NB - 8>(u, acc, rsrc, nm, nq, ng);
    }
}

// pencils a lane handles per group of passes (a basis row is fetched once per group).  Measured: groups of 2-4 passes
// with the chunk sized to fill them are SLOWER than one pass per group with one-pass chunks at every shape AUTO sends
// here (8x8x8: 0.27-0.32 of the roofline against 0.52; the larger LDS images and the per-pass guards cost more than the
// shared rows give back), and help only above nq = 12, where the block kernel is ahead anyway
// (profiles/r03/anisotropic_shapes.log).
constexpr int rt_group(int nb)
{
    return 1;
}

template <int NB, typename T>
__device__ __forceinline__ void read_pencil_rt(T (&u)[NB - 1], const T *pencil, int nm)
{
#pragma unroll
    for (int m = 0; m < NB - 1; ++m)
        u[m] = (m < nm) ? pencil[m] : T(0);
}

template <int NB, int XG = 64>
__global__ __launch_bounds__(256) void hex_wave_rt_kernel(const RtShape sh, const double *__restrict__ b0,
                                                          const double *__restrict__ b1,
                                                          const double *__restrict__ b2,
                                                          const double *__restrict__ in, double *__restrict__ out,
                                                          uint64_t nelmt)
{
    using T = double;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb  = blockDim.x >> 6;
    T *imga = lds + wib * sh.slab, *imgb = imga + sh.off_b;

    const uint64_t nchunk = (nelmt + sh.ec - 1) / sh.ec;
    const uint64_t c      = logical_block<XG>() * (uint64_t)wpb + wib;
    const bool mine       = c < nchunk;
    int evalid            = 0;
    if (mine)
    {
        const uint64_t left = nelmt - c * sh.ec;
        evalid              = left >= (uint64_t)sh.ec ? sh.ec : (int)left;
        // ---- chunk -> image A (pencils (e,r,q), stride s0) on the chunk's 16-byte word grid ---------------------------
        const T *src = in + c * (uint64_t)sh.ec * sh.nmt;
        const int n_in = evalid * sh.nmt;
        const int a    = __builtin_amdgcn_readfirstlane((int)(((uintptr_t)src >> 3) & 1));
        const double2_t *grid = reinterpret_cast<const double2_t *>(src - a);
        const int pad = sh.s0 - sh.nm0;
        constexpr int U = 4;
        for (int v0 = 0; 2 * v0 - a < n_in; v0 += U * kWave)
        {
            double2_t x[U];
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int v = v0 + k * kWave + lane, d0 = 2 * v - a;
                x[k] = double2_t{0.0, 0.0};
                if (d0 >= 0 && d0 + 1 < n_in)
                    x[k] = __builtin_nontemporal_load(grid + v);
                else
                {
                    if (d0 >= 0 && d0 < n_in)
                        x[k].x = src[d0];
                    if (d0 + 1 >= 0 && d0 + 1 < n_in)
                        x[k].y = src[d0 + 1];
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int v = v0 + k * kWave + lane, d0 = 2 * v - a;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    const int f = d0 + h;
                    if (f >= 0 && f < n_in)
                        imga[f + (pad ? div_magic(f, sh.m_nm0) : 0)] = x[k][h];
                }
            }
        }
    }
    if (!mine)
        return;
    wave_lds_fence(); // the wave's own staging writes precede its pencil reads

    const int nm12 = sh.nm1 * sh.nm2, nq0nm2 = sh.nq0 * sh.nm2, nq01 = sh.nq0 * sh.nq1;
    // buffer descriptors of the three bases: byte counts = the arrays' sizes, reads beyond them return 0
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void *)b0, 0, 8 * sh.nm0 * sh.nq0, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)b1, 0, 8 * sh.nm1 * sh.nq1, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)b2, 0, 8 * sh.nm2 * sh.nq2, kRsrcWord3);
    constexpr int G = rt_group(NB);
    // ---- direction 0: w1[(e,i,r)][q] = sum_p in[(e,r,q)][p] * B0[p][i]          image A -> image B -----------------
    {
        const int np = evalid * nm12;
        for (int t0 = 0; t0 < np; t0 += G * kWave)
        {
            T u[G][NB - 1], acc[G][NB];
            int tc[G];
            const int ng = (np - t0 + kWave - 1) / kWave; // passes of this group that hold pencils (wave-uniform)
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                tc[g]       = t < np ? t : np - 1;
                read_pencil_rt<NB>(u[g], imga + tc[g] * sh.s0, sh.nm0);
            }
            contract_rt<NB, G>(u, acc, r0, sh.nm0, sh.nq0, ng);
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                const int e = div_magic(tc[g], sh.m_nm12), rq = tc[g] - e * nm12, r = div_magic(rq, sh.m_nm1),
                          q = rq - r * sh.nm1;
                T *dst = imgb + ((e * sh.nq0) * sh.nm2 + r) * sh.s1 + q;
                if (t < np)
                {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k < sh.nq0)
                            dst[k * sh.nm2 * sh.s1] = acc[g][k];
                }
            }
        }
        wave_lds_fence();
    }
    // ---- direction 1: w2[(e,j,i)][r] = sum_q w1[(e,i,r)][q] * B1[q][j]          image B -> image A -----------------
    {
        const int np = evalid * nq0nm2;
        for (int t0 = 0; t0 < np; t0 += G * kWave)
        {
            T u[G][NB - 1], acc[G][NB];
            int tc[G];
            const int ng = (np - t0 + kWave - 1) / kWave; // passes of this group that hold pencils (wave-uniform)
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                tc[g]       = t < np ? t : np - 1;
                read_pencil_rt<NB>(u[g], imgb + tc[g] * sh.s1, sh.nm1);
            }
            contract_rt<NB, G>(u, acc, r1, sh.nm1, sh.nq1, ng);
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                const int e = div_magic(tc[g], sh.m_nq0nm2), ir = tc[g] - e * nq0nm2, i = div_magic(ir, sh.m_nm2),
                          r = ir - i * sh.nm2;
                T *dst = imga + ((e * sh.nq1) * sh.nq0 + i) * sh.s2 + r;
                if (t < np)
                {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k < sh.nq1)
                            dst[k * sh.nq0 * sh.s2] = acc[g][k];
                }
            }
        }
        wave_lds_fence();
    }
    // ---- direction 2: out[e][k][(j,i)] = sum_r w2[(e,j,i)][r] * B2[r][k]        image A -> image B (final layout) --
    {
        const int np = evalid * nq01;
        for (int t0 = 0; t0 < np; t0 += G * kWave)
        {
            T u[G][NB - 1], acc[G][NB];
            int tc[G];
            const int ng = (np - t0 + kWave - 1) / kWave; // passes of this group that hold pencils (wave-uniform)
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                tc[g]       = t < np ? t : np - 1;
                read_pencil_rt<NB>(u[g], imga + tc[g] * sh.s2, sh.nm2);
            }
            contract_rt<NB, G>(u, acc, r2, sh.nm2, sh.nq2, ng);
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (g >= ng)
                    continue;
                const int t = t0 + g * kWave + lane;
                const int e = div_magic(tc[g], sh.m_nq01), pl = tc[g] - e * nq01;
                T *dst = imgb + e * sh.nqt + pl;
                if (t < np)
                {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k < sh.nq2)
                            dst[k * nq01] = acc[g][k];
                }
            }
        }
        wave_lds_fence();
    }
    // ---- image B -> HBM: one flat stream on the 16-byte word grid of the chunk's output -------------------------------
    {
        T *dst        = out + c * (uint64_t)sh.ec * sh.nqt;
        const int n_o = evalid * sh.nqt;
        const int a   = __builtin_amdgcn_readfirstlane((int)(((uintptr_t)dst >> 3) & 1));
        double2_t *grid = reinterpret_cast<double2_t *>(dst - a);
        for (int v = lane; 2 * v - a < n_o; v += kWave)
        {
            const int d0 = 2 * v - a;
            if (d0 >= 0 && d0 + 1 < n_o)
            {
                const double2_t x = {imgb[d0], imgb[d0 + 1]};
                __builtin_nontemporal_store(x, grid + v);
            }
            else
            {
                if (d0 >= 0 && d0 < n_o)
                    dst[d0] = imgb[d0];
                if (d0 + 1 >= 0 && d0 + 1 < n_o)
                    dst[d0 + 1] = imgb[d0 + 1];
            }
        }
    }
}

} // namespace sf
