/*
 * include/sumfact.h -- C ABI of libsumfact.so, the MI355X (gfx950) drop-in for the BwdTrans
 * sum-factorisation hot path of CFD-Xing/gpu-benchmarking.
 *
 * The reference has no FFI: its "interface" for this path is the kernel call a maintainer makes from
 * run_test<T> (benchmark05/benchmark05.cc:1322-1328, benchmark04/benchmark04.cc:1001-1004): raw
 * device pointers + unsigned extents, caller owns every buffer, the kernel allocates nothing.  Each
 * entry point below replaces one such call site (cited per function).  INTEGRATION.md shows the
 * exact edit in the reference's run_test.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - layouts are the reference's: in[e][r][q][p] (p fastest), out[e][k][j][i] (i fastest),
 *     basis[p*nq + i] row-major nm x nq, nm = nq - 1;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream, as in the reference);
 *     launches are asynchronous, like a <<<>>> launch; the caller synchronises.  One call may enqueue SEVERAL kernels on
 *     `stream` (a 3D nq = 7 / 8 batch above 1 048 576 elements goes out as launches of 524 288 elements each: elements are
 *     independent and the pieces re-align the chip's eight XCDs, +2 % from 2.5 M elements); results do not depend on it;
 *   - element counts are size_t (the reference's 32-bit `unsigned` overflows at
 *     nelmt*nq^3 > 2^32, i.e. above 8 388 608 elements at nq = 8);
 *   - return value: 0 on success, a positive hipError_t, or a negative SF_E* code.  The reference
 *     reports no errors at all; nothing here aborts the process;
 *   - thread / stream safety: every entry point may be called concurrently from several host threads and on several
 *     streams of a device (sf_set_launch_hint is per calling thread).  The library's internal device memory:
 *     (i) reduction partials of sf_sumsq_* and the intermediates of the any-extent fallback (at most 1 GiB, the grid
 *     is cut down to fit) live in a scratch buffer per (device, stream; per thread for hipStreamPerThread), allocated
 *     on the first call that needs it on that stream -- that FIRST call may not be inside a stream capture;
 *     (ii) the batch counters of the persistent 2D kernels (AUTO 2D nq 25..32) live in one 512 KiB buffer per device
 *     that is allocated once and freed only by sf_shutdown(): BwdTrans launches are capture-safe from the first call
 *     (kernel and memset nodes only; inside a capture that precedes the buffer's allocation the same kernel runs
 *     with a fixed share per wave instead), and a captured graph stays replayable until sf_shutdown().
 */
#ifndef SUMFACT_H
#define SUMFACT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_VERSION 100

enum
{
    SF_OK        = 0,
    SF_EINVAL    = -1, /* nq < 2, null pointer with nelmt > 0, unknown variant */
    SF_EALIGN    = -2, /* in/out not 8-byte aligned */
    SF_ENOTBUILT = -3, /* requested variant has no instantiation for this nq */
    SF_ENOMEM    = -4  /* internal workspace allocation failed */
};

/* Kernel strategies (benchmark columns / tuning).  SF_VARIANT_AUTO picks the fastest measured: 3D isotropic nq 2..11
 * WAVE, 12 / 14 / 16 MFMA4, 13 / 15 MFMA; 2D isotropic nq 2..20 WAVE, 21..31 MFMA4, 32 MFMA; 3D anisotropic extents: WAVE where the shape
 * is one of the compile-time triples of csrc/bwdtrans_rt.hip, else WAVE_RT up to nq = 8 per direction (also taken for
 * isotropic shapes in buffers that are only 8-byte aligned); anything else (2D anisotropic, any higher order) GENERIC,
 * which runs every extent: LDS-resident while one element's images fit the
 * 160 KiB of LDS, then through a bounded internal scratch (one intermediate pair per workgroup; first use on a
 * stream allocates, so it is not stream-capture safe; SF_ENOMEM if that allocation fails). */
enum
{
    SF_VARIANT_AUTO       = 0,
    SF_VARIANT_WAVE       = 1, /* flagship: one wavefront per chunk of elements (3D nq 2..11, 2D nq 2..24, 32) */
    SF_VARIANT_THREAD     = 2, /* one thread per element, fused nest  (cf. benchmark05.cc:15-102)  */
    SF_VARIANT_BLOCK_LDS  = 3, /* one workgroup per element, 3 sweeps in LDS (cf. :291-429)       */
    SF_VARIANT_BLOCK_GLB  = 4, /* one workgroup per element, global workspace (cf. :203-289)       */
    SF_VARIANT_GENERIC    = 5, /* runtime-nq fallback (anisotropic nq0 != nq1 != nq2)              */
    SF_VARIANT_MFMA       = 6, /* v_mfma_f64_16x16x4 chained GEMMs: 2D quad nq 11..32, 3D hex nq 4..16 */
    SF_VARIANT_MFMA4      = 7, /* v_mfma_f64_4x4x4_4b chained products (tile granularity 4): 2D quad nq 8..32, 3D hex nq 12..16 */
    SF_VARIANT_WAVE_RT    = 8, /* one wavefront per chunk with RUN-TIME extents: 3D, any extents up to 16 per direction */
    SF_NUM_VARIANTS       = 9
};

int sf_version(void);
const char *sf_error_string(int rc);
const char *sf_variant_name(int variant);

/*
 * 3D hex BwdTrans: out[e][k][j][i] = sum_r sum_q sum_p in[e][r][q][p] B0[p][i] B1[q][j] B2[r][k].
 * Replaces BwdTransHexKernel_QP<<<blocks, dim3(...), smem>>>(nm0,nm1,nm2,nmTot,nq0,nq1,nq2,nelmt,
 * d_basis0,d_basis1,d_basis2,d_in,d_out)  -- benchmark05/benchmark05.cc:1322-1328 (and the other
 * five launches :1265, :1283, :1300, :1342, :1362, which compute the same result).
 */
int sf_bwdtrans_hex_f64(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                        const double *basis0, const double *basis1, const double *basis2,
                        const double *in, double *out, void *stream);

/* Same, with an explicit strategy; wsp (may be NULL) is only used by SF_VARIANT_BLOCK_GLB and
 * SF_VARIANT_THREAD and must then hold nelmt*(nq0*nm1*nm2 + nq0*nq1*nm2) doubles (benchmark05.cc:1243-1244). */
int sf_bwdtrans_hex_f64_variant(int variant, unsigned nq0, unsigned nq1, unsigned nq2,
                                size_t nelmt, const double *basis0, const double *basis1,
                                const double *basis2, const double *in, double *wsp, double *out,
                                void *stream);

/*
 * 2D quad BwdTrans: out[e][j][i] = sum_q sum_p in[e][q][p] B0[p][i] B1[q][j].
 * Replaces BwdTransQuadKernel_QP_1D<<<blocks, threads, smem>>>(nm0,nm1,nmTot,nq0,nq1,nelmt,
 * d_basis0,d_basis1,d_in,d_out) -- benchmark04/benchmark04.cc:1001-1004 (and :912, :930, :947,
 * :966, :983).
 */
int sf_bwdtrans_quad_f64(unsigned nq0, unsigned nq1, size_t nelmt, const double *basis0,
                         const double *basis1, const double *in, double *out, void *stream);

int sf_bwdtrans_quad_f64_variant(int variant, unsigned nq0, unsigned nq1, size_t nelmt,
                                 const double *basis0, const double *basis1, const double *in,
                                 double *wsp, double *out, void *stream);

/*
 * sum_i x[i]^2 -> *result_host (blocking: synchronises `stream`).  Deterministic (fixed-shape
 * two-pass tree).  Replaces thrust::transform_reduce(d_out, d_out + n, x*x, 0, plus)
 * -- benchmark05/benchmark05.cc:1273-1276 -- and bm01's l2norm kernels
 * (benchmark01/benchmark01.cc:245-253).
 */
int sf_sumsq_f64(const double *x, size_t n, double *result_host, void *stream);

/* Same reduction, result left on the device (result_dev[0]); asynchronous. */
int sf_sumsq_f64_async(const double *x, size_t n, double *result_dev, void *stream);

/*
 * Device-side initialisers (the reference fills on the host and copies:
 * benchmark05/benchmark05.cc:1195-1258).
 *   sincos : in[e*nm_tot + f] = sin(f + 1)                      (:1206-1207)
 *   basis  : basis[x] = cos(x), x < nm*nq                        (:1220)
 *   random : x[i] = U[-1,1) from splitmix64(seed, first_idx + i) (not in the reference; identical
 *            to oracle_fill_random so host and device arrays agree bit for bit)
 *   l2norm : x[i] = i%13 + (0.2 + 1e-5*(i%100191))              (benchmark01/benchmark01.cc:178)
 * sin/cos run on the device's libm: values may differ from glibc's in the last ulp.
 */
int sf_fill_sincos_f64(double *in, size_t nelmt, size_t nm_tot, void *stream);
int sf_fill_basis_f64(double *basis, size_t nm, size_t nq, void *stream);
int sf_fill_random_f64(double *x, size_t n, uint64_t seed, uint64_t first_idx, void *stream);
int sf_fill_l2norm_f64(double *x, size_t n, void *stream);

/*
 * HBM calibrators (benchmark02/benchmark02.cc:16-58 is the reference's analogue):
 * dst[i] = src[i] with 16-byte lanes, n doubles.  Used to report the *measured* stream rate next
 * to the 8 TB/s datasheet roofline.
 */
int sf_stream_copy_f64(const double *src, double *dst, size_t n, void *stream);

/*
 * Wave-64 interleaved element layout: data[(e/64)][f][e%64], padded to whole groups of
 * 64 elements.  One thread per element with fully coalesced accesses: the decomposition of the
 * reference's BwdTransHexKernel_Coa (benchmark05/benchmark05.cc:104-201, launch :1283-1286) for a
 * 64-wide wavefront and without its output-index bug (:193-194).  Sizes in doubles, G = ceil(nelmt/64)*64:
 *   in_il G*nm0*nm1*nm2, out_il G*nq0*nq1*nq2, wsp_il G*(nm1*nm2 + nm2).
 * sf_interleave64_f64 converts element-major [e][n] -> interleaved (inverse = 0) or back (inverse = 1).
 */
int sf_bwdtrans_hex_f64_interleaved(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                                    const double *basis0, const double *basis1, const double *basis2,
                                    const double *in_il, double *wsp_il, double *out_il, void *stream);
int sf_interleave64_f64(const double *src, double *dst, size_t nelmt, size_t n, int inverse,
                        void *stream);

/*
 * fp32: the reference's kernels are templates on T but only T = double is ever
 * instantiated (benchmark05/benchmark05.cc:15, 1439); these are the T = float instantiations of the
 * same kernels (float4 lanes).  sumsq accumulates in double.  Same layouts and error codes.
 */
int sf_bwdtrans_hex_f32(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt, const float *basis0,
                        const float *basis1, const float *basis2, const float *in, float *out,
                        void *stream);
int sf_bwdtrans_quad_f32(unsigned nq0, unsigned nq1, size_t nelmt, const float *basis0,
                         const float *basis1, const float *in, float *out, void *stream);
int sf_sumsq_f32(const float *x, size_t n, double *result_host, void *stream);
int sf_fill_sincos_f32(float *in, size_t nelmt, size_t nm_tot, void *stream);
int sf_fill_basis_f32(float *basis, size_t nm, size_t nq, void *stream);
int sf_fill_random_f32(float *x, size_t n, uint64_t seed, uint64_t first_idx, void *stream);

/*
 * benchmark02: x[i] += y[i]  -- replaces add_vector<T,vl><<<>>>
 * (benchmark02/benchmark02.cc:16-58); 24 bytes of HBM traffic per element (:255), so its GB/s is the
 * measured stream rate used as the second roofline denominator.  fill: data1/data2 of :84-85.
 */
int sf_vector_add_f64(double *x, const double *y, size_t n, void *stream);
int sf_fill_vecadd_f64(double *x, double *y, size_t n, void *stream);

/*
 * benchmark03: y = A x, A row-major m x n -- replaces compute_matvec<T,vl><<<>>>
 * (benchmark03/benchmark03.cc:80-104).  fill: A[i*n+j] = sin(i*n+j+1), x[j] = j (:160-167).
 */
int sf_matvec_f64(unsigned m, unsigned n, const double *A, const double *x, double *y, void *stream);
int sf_fill_matvec_f64(double *A, double *x, unsigned m, unsigned n, void *stream);

/*
 * The reference drivers' `threads` / `elblocks` arguments (benchmark05/benchmark05.cc:1428-1429): block
 * size of the thread-per-element / flat-tid kernels and elements per workgroup (`blocks = nelmt/elblocks`,
 * :1188).  They shape only the reference-style decompositions (SF_VARIANT_THREAD / BLOCK_LDS / BLOCK_GLB);
 * the wave and matrix-core kernels choose their own launch shapes.  0 = automatic (default).  The hint belongs to the
 * CALLING host thread: it shapes that thread's later launches only.
 */
int sf_set_launch_hint(unsigned threads, unsigned elblocks);

/* Number of compute units / device name of the current device (for logs). */
int sf_device_info(int *num_cu, int *wave_size, char *name, size_t name_len);

/* Free internal workspaces of the current device (optional; called at exit otherwise never). */
int sf_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* SUMFACT_H */
