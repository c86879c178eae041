// sf_dispatch.h -- internal launch / helper entry points shared between the C ABI (capi.hip) and the
// translation units that define them (bwdtrans_hex.hip, bwdtrans_quad.hip, bwdtrans_generic.hip,
// aux_kernels.hip).  Every function returns SF_OK, a negative SF_E* code or a positive hipError_t.
#pragma once

#include "sf_common.h"

#include <mutex>

namespace sf
{
int launch_hex_wave_nq(unsigned nq, const HexArgs &a, hipStream_t s);
int launch_hex_mfma_nq(unsigned nq, const HexArgs &a, hipStream_t s);
int launch_hex_mfma4_nq(unsigned nq, const HexArgs &a, hipStream_t s);
int hex_auto_kernel(unsigned nq); // SF_VARIANT_MFMA / SF_VARIANT_MFMA4 above the wave kernel's table
int launch_quad_wave_nq(unsigned nq, const QuadArgs &a, hipStream_t s);
int launch_quad_mfma_nq(unsigned nq, const QuadArgs &a, hipStream_t s);
int launch_quad_mfma4_nq(unsigned nq, const QuadArgs &a, hipStream_t s);
bool quad_prefers_mfma(unsigned nq);
int quad_auto_kernel(unsigned nq);
int launch_hex_generic(int variant, unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a,
                       hipStream_t s);
int launch_hex_wave3(unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a, hipStream_t s);
int launch_hex_rt(unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a, hipStream_t s);
int launch_quad_generic(int variant, unsigned nq0, unsigned nq1, const QuadArgs &a, hipStream_t s);
int sumsq_async(const double *x, size_t n, double *result_dev, hipStream_t s);
int sumsq_blocking(const double *x, size_t n, double *result_host, hipStream_t s);
int fill_sincos(double *in, size_t nelmt, size_t nm_tot, hipStream_t s);
int fill_basis(double *b, size_t nm, size_t nq, hipStream_t s);
int fill_random(double *x, size_t n, uint64_t seed, uint64_t first, hipStream_t s);
int fill_l2norm(double *x, size_t n, hipStream_t s);
int stream_copy(const double *src, double *dst, size_t n, hipStream_t s);
int vector_add(double *x, const double *y, size_t n, hipStream_t s);
int fill_vecadd(double *x, double *y, size_t n, hipStream_t s);
int matvec(unsigned M, unsigned N, const double *A, const double *x, double *y, hipStream_t s);
int fill_matvec(double *A, double *x, unsigned M, unsigned N, hipStream_t s);
int release_workspaces();
// internal scratch buffer of (current device, stream, kind), at least `bytes` long; kinds: 0 reductions, 1 BwdTrans
// intermediates.  Hold scratch_mutex() from the acquire to the end of the enqueue sequence that uses the buffer.
int scratch_acquire(hipStream_t s, int kind, size_t bytes, void **out);
std::recursive_mutex &scratch_mutex();
// 8-byte batch counter for one launch of a persistent kernel on `s` (zero it with a memset node in front of the launch).
// Counters live in one buffer per device that is allocated at first use and freed only by sf_shutdown().  Eager
// launches on a stream share that stream's counter (their work is ordered); a launch enqueued while `s` is capturing
// gets a counter of its own (the graph may replay on any stream).  SF_ENOMEM: none to be had -- fall back.
int counter_acquire(hipStream_t s, unsigned long long **out);
int release_counters();
int set_launch_hint(unsigned threads, unsigned elblocks);
int launch_hex_interleaved(unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a, hipStream_t s);
int launch_interleave64(const double *src, double *dst, size_t nelmt, size_t n, int inverse,
                        hipStream_t s);
int launch_hex_wave_f32_nq(unsigned nq, const HexArgsT<float> &a, hipStream_t s);
int launch_quad_wave_f32_nq(unsigned nq, const QuadArgsT<float> &a, hipStream_t s);
int launch_hex_generic_f32(int variant, unsigned nq0, unsigned nq1, unsigned nq2,
                           const HexArgsT<float> &a, hipStream_t s);
int launch_quad_generic_f32(int variant, unsigned nq0, unsigned nq1, const QuadArgsT<float> &a,
                            hipStream_t s);
int sumsq_f32_blocking(const float *x, size_t n, double *result_host, hipStream_t s);
int fill_sincos_f32(float *in, size_t nelmt, size_t nm_tot, hipStream_t s);
int fill_basis_f32(float *b, size_t nm, size_t nq, hipStream_t s);
int fill_random_f32(float *x, size_t n, uint64_t seed, uint64_t first, hipStream_t s);
} // namespace sf
