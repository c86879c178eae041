/*
 * oracle/bwdtrans_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, 64-bit indices, OpenMP over elements) of the arithmetic of the
 * BwdTrans sum-factorisation path of CFD-Xing/gpu-benchmarking.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file; the
 * product (gpu-benchmarking_amd/) never does.
 *
 * The reference has no CPU execution path (every variant is a GPU kernel), and its sources do not
 * build here (Kokkos / CUDA / cuBLAS / Thrust are absent), so this file restates the *loop nests* of
 * the reference kernels, which are plain C++ inside:
 *
 *   hex fused nest      benchmark05/benchmark05.cc:57-101   (thread-per-element kernel body)
 *   hex 3-sweep form    benchmark05/benchmark05.cc:361-423  (QP kernels; intermediate layouts)
 *   quad fused nest     benchmark04/benchmark04.cc:49-72
 *   quad 2-sweep form   benchmark04/benchmark04.cc:393-420
 *   input / basis init  benchmark05/benchmark05.cc:1195-1236, benchmark04/benchmark04.cc:859-889
 *   result reduction    benchmark05/benchmark05.cc:1273-1276 (sum of squares; sqrt at print :1397)
 *   bm01 data / norm    benchmark01/benchmark01.cc:178, :49-52
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against all 216 `norm:` values the
 * reference's committed logs hold (tests/golden/reference_norms.json).
 *
 * Summation order inside every dot product is ascending p / q / r starting from 0.0, as in the
 * reference.  Build with -ffp-contract=off for the parity oracle (no FMA contraction), see Makefile.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------- initialisers ---------------- */

/* in[e][f] = sin(f + 1), identical for every element (benchmark05.cc:1206-1207). */
void oracle_fill_sincos(double *in, size_t nelmt, size_t nm_tot)
{
#pragma omp parallel for schedule(static)
    for (size_t e = 0; e < nelmt; ++e)
        for (size_t f = 0; f < nm_tot; ++f)
            in[e * nm_tot + f] = sin((double)(f + 1));
}

/* basis[x] = cos(x), x = p*nq + i (benchmark05.cc:1220). */
void oracle_fill_basis(double *basis, size_t nm, size_t nq)
{
    for (size_t x = 0; x < nm * nq; ++x)
        basis[x] = cos((double)x);
}

/*
 * Per-value-distinct seeded data (not in the reference: its identical-per-element data hides
 * element-offset bugs).  Counter-based: value(idx) depends only on (seed, idx), so host and
 * device generate bit-identical arrays.  splitmix64 finaliser -> 53-bit mantissa -> U[-1, 1).
 */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double oracle_random_value(uint64_t seed, uint64_t idx)
{
    uint64_t h = mix64(seed ^ mix64(idx));
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

void oracle_fill_random(double *x, size_t n, uint64_t seed, uint64_t first_idx)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        x[i] = oracle_random_value(seed, first_idx + i);
}

/* x[i] = i % 13 + (0.2 + 0.00001 * (i % 100191))  (benchmark01.cc:178; i is 32-bit unsigned). */
void oracle_fill_l2norm(double *x, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
    {
        unsigned int u = (unsigned int)i;
        x[i] = u % 13u + (0.2 + 0.00001 * (u % 100191u));
    }
}

/* benchmark02: data1[i] = i%13 + (0.2 + 1e-5*(i%100191)), data2[i] = i%8 + (0.4 + 3e-5*(i%100721))
 * (benchmark02/benchmark02.cc:84-85); the timed operation is data1 += data2, 40 times (:88-97). */
void oracle_fill_vecadd(double *x, double *y, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
    {
        unsigned int u = (unsigned int)i;
        x[i] = u % 13u + (0.2 + 0.00001 * (u % 100191u));
        y[i] = u % 8u + (0.4 + 0.00003 * (u % 100721u));
    }
}

void oracle_vector_add(double *x, const double *y, size_t n, int times)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
    {
        double v = x[i];
        for (int t = 0; t < times; ++t)
            v += y[i];
        x[i] = v;
    }
}

/* benchmark03: A[i*N + j] = sin(i*N + j + 1), x[j] = j (benchmark03/benchmark03.cc:160-167);
 * y[i] = sum_j A[i][j] * x[j]  (:80-104). */
void oracle_fill_matvec(double *A, double *x, size_t M, size_t N)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < M; ++i)
        for (size_t j = 0; j < N; ++j)
            A[i * N + j] = sin((double)(i * N + j + 1));
    for (size_t j = 0; j < N; ++j)
        x[j] = (double)j;
}

static double dot_pairwise(const double *a, const double *b, size_t n)
{
    if (n <= 64)
    {
        double s = 0.0;
        for (size_t i = 0; i < n; ++i)
            s += a[i] * b[i];
        return s;
    }
    size_t h = n / 2;
    return dot_pairwise(a, b, h) + dot_pairwise(a + h, b + h, n - h);
}

void oracle_matvec(size_t M, size_t N, const double *A, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < M; ++i)
        y[i] = dot_pairwise(A + i * N, x, N);
}

/* ---------------------------------------------------------------- reductions ------------------ */

/* Pairwise (cascade) sum of squares: error O(log n * eps), independent of thread count. */
static double sumsq_pairwise(const double *x, size_t n)
{
    if (n <= 256)
    {
        double s = 0.0;
        for (size_t i = 0; i < n; ++i)
            s += x[i] * x[i];
        return s;
    }
    size_t h = n / 2;
    return sumsq_pairwise(x, h) + sumsq_pairwise(x + h, n - h);
}

double oracle_sumsq(const double *x, size_t n)
{
    enum { NB = 1024 };
    if (n < (size_t)NB * 1024)
        return sumsq_pairwise(x, n);
    double part[NB];
#pragma omp parallel for schedule(static)
    for (int b = 0; b < NB; ++b)
    {
        size_t lo = (size_t)((unsigned __int128)n * b / NB);
        size_t hi = (size_t)((unsigned __int128)n * (b + 1) / NB);
        part[b] = sumsq_pairwise(x + lo, hi - lo);
    }
    int m = NB;
    while (m > 1)
    {
        for (int b = 0; b < m / 2; ++b)
            part[b] = part[2 * b] + part[2 * b + 1];
        m /= 2;
    }
    return part[0];
}

/* ---------------------------------------------------------------- 3D hex ---------------------- */

/*
 * Fused nest, one element at a time (benchmark05.cc:57-101).  Scratch per element:
 * wsp0[nm1*nm2], wsp1[nm2] (the reference keeps them in global memory per thread).
 * in[e][r][q][p] (p fastest), out[e][k][j][i] (i fastest), basis[p*nq + i].
 */
int oracle_bwdtrans_hex_fused(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                              const double *basis0, const double *basis1, const double *basis2,
                              const double *in, double *out)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
        return -1;
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const size_t nm_tot = nm0 * nm1 * nm2, nq_tot = (size_t)nq0 * nq1 * nq2;
    int fail = 0;
#pragma omp parallel
    {
        double *wsp0 = (double *)malloc(sizeof(double) * nm1 * nm2);
        double *wsp1 = (double *)malloc(sizeof(double) * nm2);
        if (!wsp0 || !wsp1)
        {
#pragma omp atomic write
            fail = 1;
        }
        else
        {
#pragma omp for schedule(static)
            for (size_t e = 0; e < nelmt; ++e)
            {
                const double *ine = in + nm_tot * e;
                double *oute      = out + nq_tot * e;
                for (size_t i = 0; i < nq0; ++i)
                {
                    size_t cnt_rqp = 0, cnt_rq = 0;
                    for (size_t r = 0; r < nm2; ++r)
                        for (size_t q = 0; q < nm1; ++q, ++cnt_rq)
                        {
                            double tmp = 0.0;
                            for (size_t p = 0; p < nm0; ++p, ++cnt_rqp)
                                tmp += ine[cnt_rqp] * basis0[p * nq0 + i];
                            wsp0[cnt_rq] = tmp;
                        }
                    for (size_t j = 0; j < nq1; ++j)
                    {
                        cnt_rq = 0;
                        for (size_t r = 0; r < nm2; ++r)
                        {
                            double tmp = 0.0;
                            for (size_t q = 0; q < nm1; ++q, ++cnt_rq)
                                tmp += wsp0[cnt_rq] * basis1[q * nq1 + j];
                            wsp1[r] = tmp;
                        }
                        for (size_t k = 0; k < nq2; ++k)
                        {
                            double tmp = 0.0;
                            for (size_t r = 0; r < nm2; ++r)
                                tmp += wsp1[r] * basis2[r * nq2 + k];
                            oute[k * nq1 * nq0 + j * nq0 + i] = tmp;
                        }
                    }
                }
            }
        }
        free(wsp0);
        free(wsp1);
    }
    return fail ? -2 : 0;
}

/*
 * Three directional sweeps with the reference's intermediate layouts (benchmark05.cc:361-423):
 *   dir 0: wsp1[i][r][q] = sum_p in[r][q][p]   * B0[p*nq0+i]   (cnt_irq = nm1*nm2*i + nm1*r + q)
 *   dir 1: wsp2[j][i][r] = sum_q wsp1[i][r][q] * B1[q*nq1+j]   (cnt_jir = nq0*nm2*j + nm2*i + r)
 *   dir 2: out [k][j][i] = sum_r wsp2[j][i][r] * B2[r*nq2+k]   (cnt_kji = nq0*nq1*k + nq0*j + i)
 */
int oracle_bwdtrans_hex_sweeps(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                               const double *basis0, const double *basis1, const double *basis2,
                               const double *in, double *out)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
        return -1;
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const size_t nm_tot = nm0 * nm1 * nm2, nq_tot = (size_t)nq0 * nq1 * nq2;
    int fail = 0;
#pragma omp parallel
    {
        double *w1 = (double *)malloc(sizeof(double) * nq0 * nm1 * nm2);
        double *w2 = (double *)malloc(sizeof(double) * nq0 * nq1 * nm2);
        if (!w1 || !w2)
        {
#pragma omp atomic write
            fail = 1;
        }
        else
        {
#pragma omp for schedule(static)
            for (size_t e = 0; e < nelmt; ++e)
            {
                const double *ine = in + nm_tot * e;
                double *oute      = out + nq_tot * e;
                for (size_t i = 0; i < nq0; ++i)
                    for (size_t r = 0; r < nm2; ++r)
                        for (size_t q = 0; q < nm1; ++q)
                        {
                            size_t cnt_rqp = nm1 * nm0 * r + nm0 * q;
                            double tmp     = 0.0;
                            for (size_t p = 0; p < nm0; ++p, ++cnt_rqp)
                                tmp += ine[cnt_rqp] * basis0[p * nq0 + i];
                            w1[nm1 * nm2 * i + nm1 * r + q] = tmp;
                        }
                for (size_t j = 0; j < nq1; ++j)
                    for (size_t i = 0; i < nq0; ++i)
                        for (size_t r = 0; r < nm2; ++r)
                        {
                            size_t cnt_irq = nm1 * nm2 * i + nm1 * r;
                            double tmp     = 0.0;
                            for (size_t q = 0; q < nm1; ++q, ++cnt_irq)
                                tmp += w1[cnt_irq] * basis1[q * nq1 + j];
                            w2[nq0 * nm2 * j + nm2 * i + r] = tmp;
                        }
                for (size_t k = 0; k < nq2; ++k)
                    for (size_t j = 0; j < nq1; ++j)
                        for (size_t i = 0; i < nq0; ++i)
                        {
                            size_t cnt_jir = nq0 * nm2 * j + nm2 * i;
                            double tmp     = 0.0;
                            for (size_t r = 0; r < nm2; ++r, ++cnt_jir)
                                tmp += w2[cnt_jir] * basis2[r * nq2 + k];
                            oute[nq0 * nq1 * k + nq0 * j + i] = tmp;
                        }
            }
        }
        free(w1);
        free(w2);
    }
    return fail ? -2 : 0;
}

/*
 * Same three sweeps, loop order chosen for the CPU (contiguous i innermost so the compiler
 * vectorises; intermediates w1[r][q][i], w2[r][j][i]).  Every output is still the ascending-p /
 * ascending-q / ascending-r sum starting from 0.0 of benchmark05.cc:361-423, so without FMA
 * contraction the result is bit-identical to oracle_bwdtrans_hex_sweeps.  This is the form timed as
 * the CPU baseline (bench.py cpu_baseline, kind "port").
 */
int oracle_bwdtrans_hex_vector(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                               const double *basis0, const double *basis1, const double *basis2,
                               const double *in, double *out)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
        return -1;
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const size_t nm_tot = nm0 * nm1 * nm2, nq_tot = (size_t)nq0 * nq1 * nq2;
    int fail = 0;
#pragma omp parallel
    {
        double *w1 = (double *)malloc(sizeof(double) * nm2 * nm1 * nq0);
        double *w2 = (double *)malloc(sizeof(double) * nm2 * nq1 * nq0);
        if (!w1 || !w2)
        {
#pragma omp atomic write
            fail = 1;
        }
        else
        {
#pragma omp for schedule(static)
            for (size_t e = 0; e < nelmt; ++e)
            {
                const double *ine = in + nm_tot * e;
                double *oute      = out + nq_tot * e;
                for (size_t rq = 0; rq < nm2 * nm1; ++rq)
                {
                    double *dst = w1 + rq * nq0;
                    for (size_t i = 0; i < nq0; ++i)
                        dst[i] = 0.0;
                    for (size_t p = 0; p < nm0; ++p)
                    {
                        const double u = ine[rq * nm0 + p];
                        for (size_t i = 0; i < nq0; ++i)
                            dst[i] += u * basis0[p * nq0 + i];
                    }
                }
                for (size_t r = 0; r < nm2; ++r)
                    for (size_t j = 0; j < nq1; ++j)
                    {
                        double *dst = w2 + (r * nq1 + j) * nq0;
                        for (size_t i = 0; i < nq0; ++i)
                            dst[i] = 0.0;
                        for (size_t q = 0; q < nm1; ++q)
                        {
                            const double b   = basis1[q * nq1 + j];
                            const double *src = w1 + (r * nm1 + q) * nq0;
                            for (size_t i = 0; i < nq0; ++i)
                                dst[i] += src[i] * b;
                        }
                    }
                for (size_t k = 0; k < nq2; ++k)
                    for (size_t j = 0; j < nq1; ++j)
                    {
                        double *dst = oute + (k * nq1 + j) * nq0;
                        for (size_t i = 0; i < nq0; ++i)
                            dst[i] = 0.0;
                        for (size_t r = 0; r < nm2; ++r)
                        {
                            const double b   = basis2[r * nq2 + k];
                            const double *src = w2 + (r * nq1 + j) * nq0;
                            for (size_t i = 0; i < nq0; ++i)
                                dst[i] += src[i] * b;
                        }
                    }
            }
        }
        free(w1);
        free(w2);
    }
    return fail ? -2 : 0;
}

/*
 * Register-blocked form of the same three sweeps for isotropic nq = 2..10 -- the timed CPU baseline
 * (bench.py cpu_baseline, kind "port").  The i direction is held in ORACLE_VW-wide vectors (4: AVX2 ymm,
 * 8: AVX-512 zmm; rows padded to a multiple of the width against zero basis columns), 4 / 8 (r,q) pencils, j or
 * k share every basis or intermediate vector they load, and all extents are compile-time constants,
 * so the loops unroll into straight FMA chains.  Every output is still the ascending-p / ascending-q /
 * ascending-r sum starting from 0.0 of benchmark05.cc:361-423: without FMA contraction (liboracle.so)
 * the result is bit-identical to oracle_bwdtrans_hex_sweeps (tests/test_oracle_golden.py).
 */
#ifndef ORACLE_VW
#define ORACLE_VW 4 /* doubles per vector: 4 = AVX2 ymm (liboracle_fast.so), 8 = AVX-512 zmm (liboracle_avx512.so) */
#endif
#define HB_VW ORACLE_VW
#define HB_RB (HB_VW == 8 ? 8 : 4) /* rows / columns that share every vector they load */
typedef double hbv __attribute__((vector_size(8 * HB_VW), aligned(8)));
#define HB_MAXQ 10
#define HB_MAXP 16
#define HB_INLINE static inline __attribute__((always_inline))

/* dir 0, NB consecutive (r,q) rows: w1[rq][i] = sum_p in[rq][p] * B0[p][i] */
HB_INLINE void hb_dir0(const int NM, const int NV, const int NB, const double *restrict b0p,
                       const double *restrict src, double *restrict dst)
{
    const int NQP = HB_VW * NV;
    hbv acc[HB_RB][(HB_MAXP / HB_VW)];
    for (int t = 0; t < NB; ++t)
        for (int v = 0; v < NV; ++v)
            acc[t][v] = (hbv){0.0};
    for (int p = 0; p < NM; ++p)
        for (int v = 0; v < NV; ++v)
        {
            const hbv b = *(const hbv *)(b0p + p * NQP + HB_VW * v);
            for (int t = 0; t < NB; ++t)
                acc[t][v] += src[t * NM + p] * b;
        }
    for (int t = 0; t < NB; ++t)
        for (int v = 0; v < NV; ++v)
            *(hbv *)(dst + t * NQP + HB_VW * v) = acc[t][v];
}

/* dirs 1 and 2, NB consecutive output columns c: dst_c[i] = sum_m src[m*SS + i] * B[m*NQ + c]; PADDED: rows of
 * NQP doubles (intermediate) or NQ doubles (final output, tail written scalar by scalar) DS apart */
HB_INLINE void hb_dir12(const int NM, const int NQ, const int NV, const int NB, const int SS, const int DS,
                        const int PADDED, const double *restrict bas, const double *restrict src,
                        double *restrict dst)
{
    hbv acc[HB_RB][(HB_MAXP / HB_VW)];
    for (int t = 0; t < NB; ++t)
        for (int v = 0; v < NV; ++v)
            acc[t][v] = (hbv){0.0};
    for (int m = 0; m < NM; ++m)
        for (int v = 0; v < NV; ++v)
        {
            const hbv s = *(const hbv *)(src + m * SS + HB_VW * v);
            for (int t = 0; t < NB; ++t)
                acc[t][v] += s * bas[m * NQ + t];
        }
    for (int t = 0; t < NB; ++t)
        for (int v = 0; v < NV; ++v)
        {
            if (PADDED || HB_VW * v + HB_VW <= NQ)
                *(hbv *)(dst + t * DS + HB_VW * v) = acc[t][v];
            else
                for (int x = 0; HB_VW * v + x < NQ; ++x)
                    dst[t * DS + HB_VW * v + x] = acc[t][v][x];
        }
}

HB_INLINE void hex_blocked_element(const int NQ, const double *restrict b0p, const double *restrict b1,
                                   const double *restrict b2, const double *restrict ine,
                                   double *restrict oute)
{
    const int NM = NQ - 1, NV = (NQ + HB_VW - 1) / HB_VW, NQP = HB_VW * NV;
    double w1[(HB_MAXQ - 1) * (HB_MAXQ - 1) * HB_MAXP] __attribute__((aligned(64)));
    double w2[(HB_MAXQ - 1) * HB_MAXQ * HB_MAXP] __attribute__((aligned(64)));
    /* dir 0: w1[r][q][i], HB_RB (r,q) rows per pass */
    {
        const int NR = NM * NM, full = NR / HB_RB * HB_RB;
        for (int rq = 0; rq < full; rq += HB_RB)
            hb_dir0(NM, NV, HB_RB, b0p, ine + rq * NM, w1 + rq * NQP);
        if (NR - full)
            hb_dir0(NM, NV, NR - full, b0p, ine + full * NM, w1 + full * NQP);
    }
    /* dir 1: w2[r][j][i] = sum_q w1[r][q][i] * B1[q][j], HB_RB j per pass */
    for (int r = 0; r < NM; ++r)
    {
        const int full = NQ / HB_RB * HB_RB;
        for (int j = 0; j < full; j += HB_RB)
            hb_dir12(NM, NQ, NV, HB_RB, NQP, NQP, 1, b1 + j, w1 + r * NM * NQP, w2 + (r * NQ + j) * NQP);
        if (NQ - full)
            hb_dir12(NM, NQ, NV, NQ - full, NQP, NQP, 1, b1 + full, w1 + r * NM * NQP,
                     w2 + (r * NQ + full) * NQP);
    }
    /* dir 2: out[k][j][i] = sum_r w2[r][j][i] * B2[r][k], HB_RB k per pass */
    for (int j = 0; j < NQ; ++j)
    {
        const int full = NQ / HB_RB * HB_RB;
        for (int k = 0; k < full; k += HB_RB)
            hb_dir12(NM, NQ, NV, HB_RB, NQ * NQP, NQ * NQ, 0, b2 + k, w2 + j * NQP, oute + (k * NQ + j) * NQ);
        if (NQ - full)
            hb_dir12(NM, NQ, NV, NQ - full, NQ * NQP, NQ * NQ, 0, b2 + full, w2 + j * NQP,
                     oute + (full * NQ + j) * NQ);
    }
}

#define HB_INSTANCE(N)                                                                                        \
    static void hex_blocked_##N(size_t nelmt, const double *b0p, const double *b1, const double *b2,          \
                                const double *in, double *out)                                                \
    {                                                                                                         \
        const size_t nmt = (size_t)(N - 1) * (N - 1) * (N - 1), nqt = (size_t)N * N * N;                      \
        _Pragma("omp parallel for schedule(static)") for (size_t e = 0; e < nelmt; ++e)                       \
            hex_blocked_element(N, b0p, b1, b2, in + nmt * e, out + nqt * e);                                 \
    }
HB_INSTANCE(2)
HB_INSTANCE(3)
HB_INSTANCE(4)
HB_INSTANCE(5)
HB_INSTANCE(6)
HB_INSTANCE(7)
HB_INSTANCE(8)
HB_INSTANCE(9)
HB_INSTANCE(10)
#undef HB_INSTANCE

/* 1 when oracle_bwdtrans_hex_blocked is instantiated for these extents (isotropic nq 2..10) */
int oracle_has_blocked(unsigned nq0, unsigned nq1, unsigned nq2)
{
    return nq0 == nq1 && nq1 == nq2 && nq0 >= 2 && nq0 <= HB_MAXQ;
}

int oracle_bwdtrans_hex_blocked(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                                const double *basis0, const double *basis1, const double *basis2,
                                const double *in, double *out)
{
    if (!oracle_has_blocked(nq0, nq1, nq2))
        return -1;
    const unsigned nq = nq0, nm = nq - 1, nqp = (nq + HB_VW - 1) / HB_VW * HB_VW;
    double b0p[(HB_MAXQ - 1) * HB_MAXP] __attribute__((aligned(64)));
    for (unsigned p = 0; p < nm; ++p)
        for (unsigned i = 0; i < nqp; ++i)
            b0p[p * nqp + i] = i < nq ? basis0[p * nq + i] : 0.0;
    switch (nq)
    {
    case 2: hex_blocked_2(nelmt, b0p, basis1, basis2, in, out); break;
    case 3: hex_blocked_3(nelmt, b0p, basis1, basis2, in, out); break;
    case 4: hex_blocked_4(nelmt, b0p, basis1, basis2, in, out); break;
    case 5: hex_blocked_5(nelmt, b0p, basis1, basis2, in, out); break;
    case 6: hex_blocked_6(nelmt, b0p, basis1, basis2, in, out); break;
    case 7: hex_blocked_7(nelmt, b0p, basis1, basis2, in, out); break;
    case 8: hex_blocked_8(nelmt, b0p, basis1, basis2, in, out); break;
    case 9: hex_blocked_9(nelmt, b0p, basis1, basis2, in, out); break;
    default: hex_blocked_10(nelmt, b0p, basis1, basis2, in, out); break;
    }
    return 0;
}

/* ---------------------------------------------------------------- 2D quad --------------------- */

/* Fused nest (benchmark04.cc:49-72): in[e][q][p] -> out[e][j][i]; scratch wsp[nm1]. */
int oracle_bwdtrans_quad_fused(unsigned nq0, unsigned nq1, size_t nelmt, const double *basis0,
                               const double *basis1, const double *in, double *out)
{
    if (nq0 < 2 || nq1 < 2)
        return -1;
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1;
    const size_t nm_tot = nm0 * nm1, nq_tot = (size_t)nq0 * nq1;
    int fail = 0;
#pragma omp parallel
    {
        double *wsp = (double *)malloc(sizeof(double) * nm1);
        if (!wsp)
        {
#pragma omp atomic write
            fail = 1;
        }
        else
        {
#pragma omp for schedule(static)
            for (size_t e = 0; e < nelmt; ++e)
            {
                const double *ine = in + nm_tot * e;
                double *oute      = out + nq_tot * e;
                for (size_t i = 0; i < nq0; ++i)
                {
                    size_t cnt_qp = 0;
                    for (size_t q = 0; q < nm1; ++q)
                    {
                        double tmp = 0.0;
                        for (size_t p = 0; p < nm0; ++p, ++cnt_qp)
                            tmp += ine[cnt_qp] * basis0[p * nq0 + i];
                        wsp[q] = tmp;
                    }
                    for (size_t j = 0; j < nq1; ++j)
                    {
                        double tmp = 0.0;
                        for (size_t q = 0; q < nm1; ++q)
                            tmp += wsp[q] * basis1[q * nq1 + j];
                        oute[nq0 * j + i] = tmp;
                    }
                }
            }
        }
        free(wsp);
    }
    return fail ? -2 : 0;
}

/*
 * Two sweeps (benchmark04.cc:393-420):
 *   dir 0: wsp[i][q]  = sum_p in[q][p]  * B0[p*nq0+i]   (index nm1*i + q)
 *   dir 1: out[j][i]  = sum_q wsp[i][q] * B1[q*nq1+j]   (index nq0*j + i)
 */
int oracle_bwdtrans_quad_sweeps(unsigned nq0, unsigned nq1, size_t nelmt, const double *basis0,
                                const double *basis1, const double *in, double *out)
{
    if (nq0 < 2 || nq1 < 2)
        return -1;
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1;
    const size_t nm_tot = nm0 * nm1, nq_tot = (size_t)nq0 * nq1;
    int fail = 0;
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * nq0 * nm1);
        if (!w)
        {
#pragma omp atomic write
            fail = 1;
        }
        else
        {
#pragma omp for schedule(static)
            for (size_t e = 0; e < nelmt; ++e)
            {
                const double *ine = in + nm_tot * e;
                double *oute      = out + nq_tot * e;
                for (size_t i = 0; i < nq0; ++i)
                    for (size_t q = 0; q < nm1; ++q)
                    {
                        size_t cnt_qp = nm0 * q;
                        double tmp    = 0.0;
                        for (size_t p = 0; p < nm0; ++p, ++cnt_qp)
                            tmp += ine[cnt_qp] * basis0[p * nq0 + i];
                        w[nm1 * i + q] = tmp;
                    }
                for (size_t j = 0; j < nq1; ++j)
                    for (size_t i = 0; i < nq0; ++i)
                    {
                        size_t cnt_iq = nm1 * i;
                        double tmp    = 0.0;
                        for (size_t q = 0; q < nm1; ++q, ++cnt_iq)
                            tmp += w[cnt_iq] * basis1[q * nq1 + j];
                        oute[nq0 * j + i] = tmp;
                    }
            }
        }
        free(w);
    }
    return fail ? -2 : 0;
}
