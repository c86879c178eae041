// capi.hip -- the extern "C" boundary of libsumfact.so (declared in include/sumfact.h).
// Validation + dispatch only; kernels live in bwdtrans_hex.hip / bwdtrans_quad.hip /
// bwdtrans_generic.hip / aux_kernels.hip.
#include "sf_dispatch.h"

#include <cstdio>
#include <cstring>


using namespace sf;

static inline bool aligned(const void *p, size_t a)
{
    return ((uintptr_t)p & (a - 1)) == 0;
}

extern "C" {

int sf_version(void)
{
    return SF_VERSION;
}

const char *sf_error_string(int rc)
{
    switch (rc)
    {
    case SF_OK: return "success";
    case SF_EINVAL: return "invalid argument (nq < 2, null pointer, unknown variant or missing workspace)";
    case SF_EALIGN: return "pointer not sufficiently aligned";
    case SF_ENOTBUILT: return "variant not instantiated for these extents";
    case SF_ENOMEM: return "internal workspace allocation failed";
    default: return rc > 0 ? hipGetErrorString((hipError_t)rc) : "unknown sumfact error";
    }
}

const char *sf_variant_name(int variant)
{
    static const char *names[SF_NUM_VARIANTS] = {"auto",      "wave",    "thread", "block-lds",
                                                 "block-glb", "generic", "mfma",   "mfma4",
                                                 "wave-rt"};
    return (variant >= 0 && variant < SF_NUM_VARIANTS) ? names[variant] : "?";
}

int sf_bwdtrans_hex_f64_variant(int variant, unsigned nq0, unsigned nq1, unsigned nq2,
                                size_t nelmt, const double *basis0, const double *basis1,
                                const double *basis2, const double *in, double *wsp, double *out,
                                void *stream)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2 || variant < 0 || variant >= SF_NUM_VARIANTS)
        return SF_EINVAL;
    if (nelmt == 0)
        return SF_OK;
    if (!basis0 || !basis1 || !basis2 || !in || !out)
        return SF_EINVAL;
    if (!aligned(in, 8) || !aligned(out, 8) || !aligned(basis0, 8) || !aligned(basis1, 8) ||
        !aligned(basis2, 8))
        return SF_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    HexArgs a{basis0, basis1, basis2, in, wsp, out, (uint64_t)nelmt};
    const bool iso = (nq0 == nq1 && nq1 == nq2);
    // the wave kernels read `in` and write `out` with 16-byte lanes
    const bool vec_ok = aligned(in, 16) && aligned(out, 16);
    switch (variant)
    {
    case SF_VARIANT_AUTO:
    {
        if (iso && vec_ok)
        {
            int rc = launch_hex_wave_nq(nq0, a, s);
            if (rc == SF_ENOTBUILT) // above the wave kernel's table: the measured best matrix-core kernel (nq 12..16)
                rc = hex_auto_kernel(nq0) == SF_VARIANT_MFMA4 ? launch_hex_mfma4_nq(nq0, a, s) : launch_hex_mfma_nq(nq0, a, s);
            if (rc != SF_ENOTBUILT)
                return rc;
        }
        // anisotropic extents, or buffers that are only 8-byte aligned: the compile-time triples of bwdtrans_rt.hip, then
        // the run-time-extent wave kernel (bwdtrans_rt.h), then the barrier-per-sweep block kernel
        int rc = (!iso && vec_ok) ? launch_hex_wave3(nq0, nq1, nq2, a, s) : SF_ENOTBUILT; // compile-time triples
        // the run-time-extent kernel is ahead of the block kernel up to nq = 8 per direction (0.39-0.52 of the roofline
        // against 0.28-0.34; above that its unrolled-to-the-bound loops lose: profiles/r03/anisotropic_shapes.log)
        if (rc == SF_ENOTBUILT && nq0 <= 8 && nq1 <= 8 && nq2 <= 8)
            rc = launch_hex_rt(nq0, nq1, nq2, a, s);
        if (rc != SF_ENOTBUILT)
            return rc;
        return launch_hex_generic(SF_VARIANT_GENERIC, nq0, nq1, nq2, a, s);
    }
    case SF_VARIANT_WAVE_RT:
        return launch_hex_rt(nq0, nq1, nq2, a, s);
    case SF_VARIANT_WAVE:
        if (!vec_ok)
            return SF_EALIGN;
        return iso ? launch_hex_wave_nq(nq0, a, s) : launch_hex_wave3(nq0, nq1, nq2, a, s);
    case SF_VARIANT_MFMA:
        if (!iso)
            return SF_ENOTBUILT;
        if (!vec_ok)
            return SF_EALIGN;
        return launch_hex_mfma_nq(nq0, a, s);
    case SF_VARIANT_MFMA4:
        if (!iso)
            return SF_ENOTBUILT;
        if (!vec_ok)
            return SF_EALIGN;
        return launch_hex_mfma4_nq(nq0, a, s);
    case SF_VARIANT_GENERIC:
        return launch_hex_generic(SF_VARIANT_GENERIC, nq0, nq1, nq2, a, s);
    case SF_VARIANT_THREAD:
    case SF_VARIANT_BLOCK_LDS:
    case SF_VARIANT_BLOCK_GLB:
        return launch_hex_generic(variant, nq0, nq1, nq2, a, s);
    default:
        return SF_ENOTBUILT;
    }
}

int sf_bwdtrans_hex_f64(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                        const double *basis0, const double *basis1, const double *basis2,
                        const double *in, double *out, void *stream)
{
    return sf_bwdtrans_hex_f64_variant(SF_VARIANT_AUTO, nq0, nq1, nq2, nelmt, basis0, basis1, basis2,
                                       in, nullptr, out, stream);
}

int sf_bwdtrans_quad_f64_variant(int variant, unsigned nq0, unsigned nq1, size_t nelmt,
                                 const double *basis0, const double *basis1, const double *in,
                                 double *wsp, double *out, void *stream)
{
    if (nq0 < 2 || nq1 < 2 || variant < 0 || variant >= SF_NUM_VARIANTS)
        return SF_EINVAL;
    if (nelmt == 0)
        return SF_OK;
    if (!basis0 || !basis1 || !in || !out)
        return SF_EINVAL;
    if (!aligned(in, 8) || !aligned(out, 8) || !aligned(basis0, 8) || !aligned(basis1, 8))
        return SF_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    QuadArgs a{basis0, basis1, in, wsp, out, (uint64_t)nelmt};
    const bool iso    = (nq0 == nq1);
    const bool vec_ok = aligned(in, 16) && aligned(out, 16);
    switch (variant)
    {
    case SF_VARIANT_AUTO:
    {
        if (iso && vec_ok)
        {
            // the measured best kernel of the order first (bwdtrans_quad.hip), then whatever else is built for it
            const int first = quad_auto_kernel(nq0);
            int rc = first == SF_VARIANT_MFMA4 ? launch_quad_mfma4_nq(nq0, a, s)
                                               : (first == SF_VARIANT_MFMA ? launch_quad_mfma_nq(nq0, a, s) : SF_ENOTBUILT);
            if (rc == SF_ENOTBUILT)
                rc = launch_quad_wave_nq(nq0, a, s);
            if (rc == SF_ENOTBUILT)
                rc = launch_quad_mfma_nq(nq0, a, s);
            if (rc != SF_ENOTBUILT)
                return rc;
        }
        return launch_quad_generic(SF_VARIANT_GENERIC, nq0, nq1, a, s);
    }
    case SF_VARIANT_WAVE:
        if (!iso)
            return SF_ENOTBUILT;
        if (!vec_ok)
            return SF_EALIGN;
        return launch_quad_wave_nq(nq0, a, s);
    case SF_VARIANT_MFMA:
        if (!iso)
            return SF_ENOTBUILT;
        if (!vec_ok)
            return SF_EALIGN;
        return launch_quad_mfma_nq(nq0, a, s);
    case SF_VARIANT_MFMA4:
        if (!iso)
            return SF_ENOTBUILT;
        if (!vec_ok)
            return SF_EALIGN;
        return launch_quad_mfma4_nq(nq0, a, s);
    case SF_VARIANT_GENERIC:
        return launch_quad_generic(SF_VARIANT_GENERIC, nq0, nq1, a, s);
    case SF_VARIANT_THREAD:
    case SF_VARIANT_BLOCK_LDS:
    case SF_VARIANT_BLOCK_GLB:
        return launch_quad_generic(variant, nq0, nq1, a, s);
    default:
        return SF_ENOTBUILT;
    }
}

int sf_bwdtrans_quad_f64(unsigned nq0, unsigned nq1, size_t nelmt, const double *basis0,
                         const double *basis1, const double *in, double *out, void *stream)
{
    return sf_bwdtrans_quad_f64_variant(SF_VARIANT_AUTO, nq0, nq1, nelmt, basis0, basis1, in,
                                        nullptr, out, stream);
}

int sf_bwdtrans_hex_f64_interleaved(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt,
                                    const double *basis0, const double *basis1, const double *basis2,
                                    const double *in_il, double *wsp_il, double *out_il, void *stream)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
        return SF_EINVAL;
    if (nelmt == 0)
        return SF_OK;
    if (!basis0 || !basis1 || !basis2 || !in_il || !wsp_il || !out_il)
        return SF_EINVAL;
    if (!aligned(in_il, 8) || !aligned(out_il, 8) || !aligned(wsp_il, 8))
        return SF_EALIGN;
    HexArgs a{basis0, basis1, basis2, in_il, wsp_il, out_il, (uint64_t)nelmt};
    return launch_hex_interleaved(nq0, nq1, nq2, a, (hipStream_t)stream);
}

int sf_interleave64_f64(const double *src, double *dst, size_t nelmt, size_t n, int inverse,
                        void *stream)
{
    if ((!src || !dst) && nelmt * n)
        return SF_EINVAL;
    return launch_interleave64(src, dst, nelmt, n, inverse, (hipStream_t)stream);
}

// ---- fp32 (T = float) ------------------------------------------------------------------------------
int sf_bwdtrans_hex_f32(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt, const float *basis0,
                        const float *basis1, const float *basis2, const float *in, float *out,
                        void *stream)
{
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
        return SF_EINVAL;
    if (nelmt == 0)
        return SF_OK;
    if (!basis0 || !basis1 || !basis2 || !in || !out)
        return SF_EINVAL;
    if (!aligned(in, 4) || !aligned(out, 4) || !aligned(basis0, 4) || !aligned(basis1, 4) ||
        !aligned(basis2, 4))
        return SF_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    HexArgsT<float> a{basis0, basis1, basis2, in, nullptr, out, (uint64_t)nelmt};
    if (nq0 == nq1 && nq1 == nq2 && aligned(in, 16) && aligned(out, 16))
    {
        int rc = launch_hex_wave_f32_nq(nq0, a, s);
        if (rc != SF_ENOTBUILT)
            return rc;
    }
    return launch_hex_generic_f32(SF_VARIANT_GENERIC, nq0, nq1, nq2, a, s);
}

int sf_bwdtrans_quad_f32(unsigned nq0, unsigned nq1, size_t nelmt, const float *basis0,
                         const float *basis1, const float *in, float *out, void *stream)
{
    if (nq0 < 2 || nq1 < 2)
        return SF_EINVAL;
    if (nelmt == 0)
        return SF_OK;
    if (!basis0 || !basis1 || !in || !out)
        return SF_EINVAL;
    if (!aligned(in, 4) || !aligned(out, 4) || !aligned(basis0, 4) || !aligned(basis1, 4))
        return SF_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    QuadArgsT<float> a{basis0, basis1, in, nullptr, out, (uint64_t)nelmt};
    if (nq0 == nq1 && aligned(in, 16) && aligned(out, 16))
    {
        int rc = launch_quad_wave_f32_nq(nq0, a, s);
        if (rc != SF_ENOTBUILT)
            return rc;
    }
    return launch_quad_generic_f32(SF_VARIANT_GENERIC, nq0, nq1, a, s);
}

int sf_sumsq_f32(const float *x, size_t n, double *result_host, void *stream)
{
    if (!result_host || (!x && n))
        return SF_EINVAL;
    if (n == 0)
    {
        *result_host = 0.0;
        return SF_OK;
    }
    return sumsq_f32_blocking(x, n, result_host, (hipStream_t)stream);
}

int sf_fill_sincos_f32(float *in, size_t nelmt, size_t nm_tot, void *stream)
{
    if ((!in && nelmt * nm_tot) || nm_tot > 0xffffffffull)
        return SF_EINVAL;
    return fill_sincos_f32(in, nelmt, nm_tot, (hipStream_t)stream);
}

int sf_fill_basis_f32(float *basis, size_t nm, size_t nq, void *stream)
{
    if ((!basis && nm * nq) || nm * nq > 0xffffffffull)
        return SF_EINVAL;
    return fill_basis_f32(basis, nm, nq, (hipStream_t)stream);
}

int sf_fill_random_f32(float *x, size_t n, uint64_t seed, uint64_t first_idx, void *stream)
{
    if (!x && n)
        return SF_EINVAL;
    return fill_random_f32(x, n, seed, first_idx, (hipStream_t)stream);
}

int sf_sumsq_f64(const double *x, size_t n, double *result_host, void *stream)
{
    if (!result_host || (!x && n))
        return SF_EINVAL;
    if (n == 0)
    {
        *result_host = 0.0;
        return SF_OK;
    }
    if (!aligned(x, 8))
        return SF_EALIGN;
    return sumsq_blocking(x, n, result_host, (hipStream_t)stream);
}

int sf_sumsq_f64_async(const double *x, size_t n, double *result_dev, void *stream)
{
    if (!result_dev || (!x && n))
        return SF_EINVAL;
    if (!aligned(x, 8))
        return SF_EALIGN;
    return sumsq_async(x, n, result_dev, (hipStream_t)stream);
}

int sf_fill_sincos_f64(double *in, size_t nelmt, size_t nm_tot, void *stream)
{
    if ((!in && nelmt * nm_tot) || nm_tot > 0xffffffffull)
        return SF_EINVAL;
    return fill_sincos(in, nelmt, nm_tot, (hipStream_t)stream);
}

int sf_fill_basis_f64(double *basis, size_t nm, size_t nq, void *stream)
{
    if ((!basis && nm * nq) || nm * nq > 0xffffffffull)
        return SF_EINVAL;
    return fill_basis(basis, nm, nq, (hipStream_t)stream);
}

int sf_fill_random_f64(double *x, size_t n, uint64_t seed, uint64_t first_idx, void *stream)
{
    if (!x && n)
        return SF_EINVAL;
    return fill_random(x, n, seed, first_idx, (hipStream_t)stream);
}

int sf_fill_l2norm_f64(double *x, size_t n, void *stream)
{
    if (!x && n)
        return SF_EINVAL;
    return fill_l2norm(x, n, (hipStream_t)stream);
}

int sf_stream_copy_f64(const double *src, double *dst, size_t n, void *stream)
{
    if ((!src || !dst) && n)
        return SF_EINVAL;
    return stream_copy(src, dst, n, (hipStream_t)stream);
}

int sf_vector_add_f64(double *x, const double *y, size_t n, void *stream)
{
    if ((!x || !y) && n)
        return SF_EINVAL;
    if (!aligned(x, 8) || !aligned(y, 8))
        return SF_EALIGN;
    return vector_add(x, y, n, (hipStream_t)stream);
}

int sf_fill_vecadd_f64(double *x, double *y, size_t n, void *stream)
{
    if ((!x || !y) && n)
        return SF_EINVAL;
    return fill_vecadd(x, y, n, (hipStream_t)stream);
}

int sf_matvec_f64(unsigned m, unsigned n, const double *A, const double *x, double *y, void *stream)
{
    if ((!A || !x || !y) && m && n)
        return SF_EINVAL;
    if (!aligned(A, 8) || !aligned(x, 8) || !aligned(y, 8))
        return SF_EALIGN;
    return matvec(m, n, A, x, y, (hipStream_t)stream);
}

int sf_fill_matvec_f64(double *A, double *x, unsigned m, unsigned n, void *stream)
{
    if ((!A || !x) && m && n)
        return SF_EINVAL;
    return fill_matvec(A, x, m, n, (hipStream_t)stream);
}

int sf_set_launch_hint(unsigned threads, unsigned elblocks)
{
    return set_launch_hint(threads, elblocks);
}

int sf_device_info(int *num_cu, int *wave_size, char *name, size_t name_len)
{
    int dev      = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess)
        return (int)e;
    if (num_cu)
        *num_cu = p.multiProcessorCount;
    if (wave_size)
        *wave_size = p.warpSize;
    if (name && name_len)
    {
        std::snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return SF_OK;
}

int sf_shutdown(void)
{
    (void)release_counters();
    return release_workspaces();
}

} // extern "C"
