/* examples/consumer.c -- a plain C99 consumer of include/sumfact.h, compiled with gcc (not hipcc): what a maintainer of
 * the reference does at benchmark05/benchmark05.cc:1239-1276 (cudaMalloc, fill, kernel, thrust reduce) through the C ABI.
 * Prints sqrt(sum out^2) for nq = 8, 1 048 576 elements of sin/cos data: the reference logs 17134.76235
 * (benchmark05/nq8x8x8.log:45).   make -C examples   (gcc + libsumfact.so + libamdhip64) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include "sumfact.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, sf_error_string(rc_)); return 1; } } while (0)

int main(void)
{
    const unsigned nq = 8, nm = nq - 1;
    const size_t nelmt = 1048576, nin = nelmt * nm * nm * nm, nout = nelmt * nq * nq * nq;
    double *in, *out, *basis, sumsq = 0.0;
    CHECK(hipMalloc((void **)&in, nin * sizeof(double)));
    CHECK(hipMalloc((void **)&out, nout * sizeof(double)));
    CHECK(hipMalloc((void **)&basis, nm * nq * sizeof(double)));
    CHECK(sf_fill_sincos_f64(in, nelmt, nm * nm * nm, NULL));          /* in[e][f] = sin(f + 1)   (:1206-1207) */
    CHECK(sf_fill_basis_f64(basis, nm, nq, NULL));                     /* basis[x] = cos(x)       (:1220)      */
    CHECK(sf_bwdtrans_hex_f64(nq, nq, nq, nelmt, basis, basis, basis, in, out, NULL)); /* the kernel (:1322-1328) */
    CHECK(sf_sumsq_f64(out, nout, &sumsq, NULL));                      /* thrust::transform_reduce (:1273-1276) */
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(basis));
    printf("nelmt %zu norm: %.10g\n", nelmt, sqrt(sumsq));
    return fabs(sqrt(sumsq) - 17134.76235) <= 5.5e-10 * 17134.76235 ? 0 : 2;
}
