// bwdtrans_generic.hip -- runtime-extent BwdTrans kernels.
//
// Two jobs:
//  (1) fallback for extents the compile-time wave kernels are not instantiated for (anisotropic
//      nq0 != nq1 != nq2, or nq beyond the tables);
//  (2) the reference's own work decompositions, written natively for gfx950, as same-hardware baseline
//      columns of the benchmark drivers:
//        SF_VARIANT_BLOCK_LDS  one workgroup per element, all operands in LDS, flat-tid sweeps
//                              (decomposition of BwdTransHexKernel_QP_1D shared,
//                               benchmark05/benchmark05.cc:510-617; 2D benchmark04.cc:353-426)
//        SF_VARIANT_BLOCK_GLB  same with the intermediates in a global workspace (:431-508)
//        SF_VARIANT_THREAD     one thread per element, fused nest, global per-thread scratch (:15-102)
// Intermediate layouts follow the reference: w1[i][r][q], w2[j][i][r].
// All element indexing is 64-bit.
#include "sf_dispatch.h"

namespace sf
{

// ------------------------------------------------------------------------------------------------
// BASIS_LDS = false: the bases are read from global memory (cached) -- only for extents whose bases alone exceed the
// LDS.  wsp_by_block: the global workspace holds one (w1, w2) pair per WORKGROUP (the library's own bounded scratch)
// instead of one per element (the reference's caller-owned layout, benchmark05.cc:1243-1244).
// LDS mode: w2 is written where the element's input image was -- the image is dead once sweep 1 has run -- so the
// LDS need is bases + max(nm^3, nq0 nq1 nm2) + nq0 nm1 nm2 scalars (the reference keeps all three, :299-305).
template <bool GLOBAL_WSP, bool BASIS_LDS, typename T>
__global__ __launch_bounds__(256) void hex_block_kernel(unsigned nq0, unsigned nq1, unsigned nq2,
                                                        HexArgsT<T> a, int wsp_by_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const unsigned nmt = nm0 * nm1 * nm2, nqt = nq0 * nq1 * nq2;
    const unsigned n1 = nq0 * nm1 * nm2, n2 = nq0 * nq1 * nm2;
    const unsigned nb = BASIS_LDS ? nm0 * nq0 + nm1 * nq1 + nm2 * nq2 : 0;
    const T *sb0 = a.b0, *sb1 = a.b1, *sb2 = a.b2;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if (BASIS_LDS)
    {
        T *l0 = lds, *l1 = l0 + nm0 * nq0, *l2 = l1 + nm1 * nq1;
        for (unsigned x = tid; x < nm0 * nq0; x += nt)
            l0[x] = a.b0[x];
        for (unsigned x = tid; x < nm1 * nq1; x += nt)
            l1[x] = a.b1[x];
        for (unsigned x = tid; x < nm2 * nq2; x += nt)
            l2[x] = a.b2[x];
        sb0 = l0, sb1 = l1, sb2 = l2;
    }
    T *sin_ = lds + nb;                       // input image, later w2
    T *sw2  = sin_;
    T *sw1  = sin_ + (nmt > n2 ? nmt : n2);

    for (uint64_t e = blockIdx.x; e < a.nelmt; e += gridDim.x)
    {
        const T *ine = a.in + e * nmt;
        T *oute      = a.out + e * nqt;
        T *w1        = GLOBAL_WSP ? a.wsp + (wsp_by_block ? (uint64_t)blockIdx.x : e) * (uint64_t)(n1 + n2) : sw1;
        T *w2        = GLOBAL_WSP ? w1 + n1 : sw2;
        const T *src = ine;
        if (!GLOBAL_WSP)
        {
            for (unsigned x = tid; x < nmt; x += nt)
                sin_[x] = ine[x];
            src = sin_;
        }
        __syncthreads();
        for (unsigned x = tid; x < n1; x += nt) // x = (i, r, q)
        {
            const unsigned q = x % nm1, ir = x / nm1, r = ir % nm2, i = ir / nm2;
            const T *u = src + (r * nm1 + q) * nm0;
            T t        = 0;
            for (unsigned p = 0; p < nm0; ++p)
                t += u[p] * sb0[p * nq0 + i];
            w1[x] = t;
        }
        __syncthreads();
        for (unsigned x = tid; x < n2; x += nt) // x = (j, i, r)
        {
            const unsigned r = x % nm2, ji = x / nm2, i = ji % nq0, j = ji / nq0;
            const T *u = w1 + (i * nm2 + r) * nm1;
            T t        = 0;
            for (unsigned q = 0; q < nm1; ++q)
                t += u[q] * sb1[q * nq1 + j];
            w2[x] = t;
        }
        __syncthreads();
        for (unsigned x = tid; x < nqt; x += nt) // x = (k, j, i)
        {
            const unsigned ji = x % (nq0 * nq1), k = x / (nq0 * nq1);
            const T *u = w2 + ji * nm2;
            T t        = 0;
            for (unsigned r = 0; r < nm2; ++r)
                t += u[r] * sb2[r * nq2 + k];
            oute[x] = t;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void hex_thread_kernel(unsigned nq0, unsigned nq1, unsigned nq2,
                                                         HexArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const unsigned nmt = nm0 * nm1 * nm2, nqt = nq0 * nq1 * nq2;
    T *sb0 = lds, *sb1 = sb0 + nm0 * nq0, *sb2 = sb1 + nm1 * nq1;
    for (unsigned x = threadIdx.x; x < nm0 * nq0; x += blockDim.x)
        sb0[x] = a.b0[x];
    for (unsigned x = threadIdx.x; x < nm1 * nq1; x += blockDim.x)
        sb1[x] = a.b1[x];
    for (unsigned x = threadIdx.x; x < nm2 * nq2; x += blockDim.x)
        sb2[x] = a.b2[x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nelmt; e += stride)
    {
        const T *ine = a.in + e * nmt;
        T *oute      = a.out + e * nqt;
        T *s0        = a.wsp + e * (uint64_t)(nm1 * nm2 + nm2); // per-thread scratch
        T *s1        = s0 + nm1 * nm2;
        for (unsigned i = 0; i < nq0; ++i)
        {
            for (unsigned rq = 0; rq < nm1 * nm2; ++rq)
            {
                T t = 0;
                for (unsigned p = 0; p < nm0; ++p)
                    t += ine[rq * nm0 + p] * sb0[p * nq0 + i];
                s0[rq] = t;
            }
            for (unsigned j = 0; j < nq1; ++j)
            {
                for (unsigned r = 0; r < nm2; ++r)
                {
                    T t = 0;
                    for (unsigned q = 0; q < nm1; ++q)
                        t += s0[r * nm1 + q] * sb1[q * nq1 + j];
                    s1[r] = t;
                }
                for (unsigned k = 0; k < nq2; ++k)
                {
                    T t = 0;
                    for (unsigned r = 0; r < nm2; ++r)
                        t += s1[r] * sb2[r * nq2 + k];
                    oute[(k * nq1 + j) * nq0 + i] = t;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <bool GLOBAL_WSP, bool BASIS_LDS, typename T>
__global__ __launch_bounds__(256) void quad_block_kernel(unsigned nq0, unsigned nq1, QuadArgsT<T> a,
                                                         int wsp_by_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1;
    const unsigned nmt = nm0 * nm1, nqt = nq0 * nq1, n1 = nq0 * nm1;
    const unsigned nb = BASIS_LDS ? nm0 * nq0 + nm1 * nq1 : 0;
    const T *sb0 = a.b0, *sb1 = a.b1;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if (BASIS_LDS)
    {
        T *l0 = lds, *l1 = l0 + nm0 * nq0;
        for (unsigned x = tid; x < nm0 * nq0; x += nt)
            l0[x] = a.b0[x];
        for (unsigned x = tid; x < nm1 * nq1; x += nt)
            l1[x] = a.b1[x];
        sb0 = l0, sb1 = l1;
    }
    T *sin_ = lds + nb;
    T *sw   = sin_ + nmt;
    for (uint64_t e = blockIdx.x; e < a.nelmt; e += gridDim.x)
    {
        const T *ine = a.in + e * nmt;
        T *oute      = a.out + e * nqt;
        T *w         = GLOBAL_WSP ? a.wsp + (wsp_by_block ? (uint64_t)blockIdx.x : e) * (uint64_t)n1 : sw;
        const T *src = ine;
        if (!GLOBAL_WSP)
        {
            for (unsigned x = tid; x < nmt; x += nt)
                sin_[x] = ine[x];
            src = sin_;
        }
        __syncthreads();
        for (unsigned x = tid; x < n1; x += nt) // x = (i, q)
        {
            const unsigned q = x % nm1, i = x / nm1;
            T t         = 0;
            for (unsigned p = 0; p < nm0; ++p)
                t += src[q * nm0 + p] * sb0[p * nq0 + i];
            w[x] = t;
        }
        __syncthreads();
        for (unsigned x = tid; x < nqt; x += nt) // x = (j, i)
        {
            const unsigned i = x % nq0, j = x / nq0;
            T t         = 0;
            for (unsigned q = 0; q < nm1; ++q)
                t += w[i * nm1 + q] * sb1[q * nq1 + j];
            oute[x] = t;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void quad_thread_kernel(unsigned nq0, unsigned nq1, QuadArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1;
    const unsigned nmt = nm0 * nm1, nqt = nq0 * nq1;
    T *sb0 = lds, *sb1 = sb0 + nm0 * nq0;
    for (unsigned x = threadIdx.x; x < nm0 * nq0; x += blockDim.x)
        sb0[x] = a.b0[x];
    for (unsigned x = threadIdx.x; x < nm1 * nq1; x += blockDim.x)
        sb1[x] = a.b1[x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nelmt; e += stride)
    {
        const T *ine = a.in + e * nmt;
        T *oute      = a.out + e * nqt;
        T *s         = a.wsp + e * (uint64_t)nm1;
        for (unsigned i = 0; i < nq0; ++i)
        {
            for (unsigned q = 0; q < nm1; ++q)
            {
                T t = 0;
                for (unsigned p = 0; p < nm0; ++p)
                    t += ine[q * nm0 + p] * sb0[p * nq0 + i];
                s[q] = t;
            }
            for (unsigned j = 0; j < nq1; ++j)
            {
                T t = 0;
                for (unsigned q = 0; q < nm1; ++q)
                    t += s[q] * sb1[q * nq1 + j];
                oute[j * nq0 + i] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-64 interleaved element layout: data[(e/64)][f][e%64].  One thread per element,
// fused nest, every global access of a wavefront is 64 consecutive doubles.  This is the decomposition of
// the reference's BwdTransHexKernel_Coa (benchmark05/benchmark05.cc:104-201) with the wavefront width of
// CDNA (64, the reference hard-codes 32) and WITHOUT its output-index bug (its output base omits the
// factor nq2, :193-194, so blocks overlap and the published column-7 norm is wrong).
__global__ __launch_bounds__(256) void hex_interleaved_kernel(unsigned nq0, unsigned nq1, unsigned nq2,
                                                              HexArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *lds = reinterpret_cast<double *>(lds_raw);
    const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const uint64_t nmt = (uint64_t)nm0 * nm1 * nm2, nqt = (uint64_t)nq0 * nq1 * nq2;
    const unsigned nrq = nm1 * nm2;
    double *sb0 = lds, *sb1 = sb0 + nm0 * nq0, *sb2 = sb1 + nm1 * nq1;
    for (unsigned x = threadIdx.x; x < nm0 * nq0; x += blockDim.x)
        sb0[x] = a.b0[x];
    for (unsigned x = threadIdx.x; x < nm1 * nq1; x += blockDim.x)
        sb1[x] = a.b1[x];
    for (unsigned x = threadIdx.x; x < nm2 * nq2; x += blockDim.x)
        sb2[x] = a.b2[x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nelmt; e += stride)
    {
        const uint64_t grp = e / kWave, lane = e % kWave;
        const double *ine = a.in + grp * kWave * nmt + lane;          // + 64*f
        double *oute      = a.out + grp * kWave * nqt + lane;         // + 64*g
        double *s0        = a.wsp + grp * kWave * (nrq + nm2) + lane; // + 64*x
        double *s1        = s0 + (uint64_t)kWave * nrq;
        for (unsigned i = 0; i < nq0; ++i)
        {
            for (unsigned rq = 0; rq < nrq; ++rq)
            {
                double t = 0.0;
                for (unsigned p = 0; p < nm0; ++p)
                    t += ine[(uint64_t)kWave * (rq * nm0 + p)] * sb0[p * nq0 + i];
                s0[(uint64_t)kWave * rq] = t;
            }
            for (unsigned j = 0; j < nq1; ++j)
            {
                for (unsigned r = 0; r < nm2; ++r)
                {
                    double t = 0.0;
                    for (unsigned q = 0; q < nm1; ++q)
                        t += s0[(uint64_t)kWave * (r * nm1 + q)] * sb1[q * nq1 + j];
                    s1[(uint64_t)kWave * r] = t;
                }
                for (unsigned k = 0; k < nq2; ++k)
                {
                    double t = 0.0;
                    for (unsigned r = 0; r < nm2; ++r)
                        t += s1[(uint64_t)kWave * r] * sb2[r * nq2 + k];
                    oute[(uint64_t)kWave * ((k * nq1 + j) * nq0 + i)] = t;
                }
            }
        }
    }
}

// element-major [e][n] <-> wave-64 interleaved [(e/64)][n][e%64]
__global__ __launch_bounds__(256) void interleave64_kernel(const double *__restrict__ src,
                                                           double *__restrict__ dst, uint64_t nelmt,
                                                           uint64_t n, int inverse)
{
    const uint64_t total  = (nelmt + kWave - 1) / kWave * kWave * n; // padded to whole groups
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride)
    {
        // x enumerates the interleaved side contiguously inside full groups: (grp, f, lane)
        const uint64_t grp = x / (kWave * n), rem = x - grp * kWave * n;
        const uint64_t f = rem / kWave, lane = rem - f * kWave;
        const uint64_t e = grp * kWave + lane;
        if (e >= nelmt)
            continue;
        const uint64_t il = grp * kWave * n + f * kWave + lane, em = e * n + f;
        if (inverse)
            dst[em] = src[il];
        else
            dst[il] = src[em];
    }
}

// ------------------------------------------------------------------------------------------------
static inline int launch_rc()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

constexpr size_t kMaxDynLds = 160 * 1024;

// Launch hints = the reference drivers' `threads` / `elblocks` CLI arguments
// (benchmark05/benchmark05.cc:1428-1429): block size of the thread-per-element and flat-tid kernels
// (:1265, :1343) and elements per workgroup of the block-per-element kernels (`blocks = nelmt / elblocks`,
// :1188).  0 = automatic.  They only shape these reference-style decompositions.  Per host thread: a thread that sets a
// hint shapes its own later launches, never another thread's.
static thread_local unsigned g_hint_threads = 0, g_hint_elblocks = 0;

// The any-extent fallback keeps one intermediate pair per WORKGROUP in the library's scratch; the grid is cut down so that
// the scratch stays within this budget whatever the order (one workgroup at least).
constexpr uint64_t kScratchBudgetBytes = 1ull << 30;
static inline unsigned by_block_grid(uint64_t nelmt, unsigned cu, uint64_t pair_bytes)
{
    uint64_t grid = nelmt < (uint64_t)cu * 8 ? nelmt : (uint64_t)cu * 8;
    const uint64_t fit = pair_bytes ? kScratchBudgetBytes / pair_bytes : grid;
    if (grid > fit)
        grid = fit;
    return (unsigned)(grid < 1 ? 1 : grid);
}

int set_launch_hint(unsigned threads, unsigned elblocks)
{
    g_hint_threads  = threads;
    g_hint_elblocks = elblocks;
    return SF_OK;
}

static inline unsigned hinted_block(unsigned automatic, unsigned work_items)
{
    if (g_hint_threads == 0)
        return automatic;
    unsigned t = g_hint_threads < work_items ? g_hint_threads : work_items; // min(nq^d, threads), :1343
    t          = (t + kWave - 1) / kWave * kWave;                           // whole wavefronts
    return t < 64 ? 64 : (t > 256 ? 256 : t);
}

static inline unsigned hinted_grid(uint64_t nelmt, uint64_t cap)
{
    if (g_hint_elblocks == 0)
        return (unsigned)(nelmt < cap ? nelmt : cap);
    const uint64_t g = (nelmt + g_hint_elblocks - 1) / g_hint_elblocks;
    return (unsigned)(g < 1 ? 1 : (g > 0x7fffffffull ? 0x7fffffffull : g));
}

template <typename T>
int launch_hex_generic_t(int variant, unsigned nq0, unsigned nq1, unsigned nq2, const HexArgsT<T> &a,
                         hipStream_t s)
{
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const size_t nbas = nm0 * nq0 + nm1 * nq1 + nm2 * nq2;
    const unsigned cu = (unsigned)device_info().num_cu;
    if (variant == SF_VARIANT_THREAD)
    {
        if (!a.wsp)
            return SF_EINVAL;
        const size_t lds = sizeof(T) * nbas;
        if (lds > kMaxDynLds)
            return SF_ENOTBUILT;
        const unsigned bs     = hinted_block(128, 0xffffffffu);
        const uint64_t blocks = (a.nelmt + bs - 1) / bs;
        hex_thread_kernel<T><<<(unsigned)(blocks > cu * 16 ? cu * 16 : blocks), bs, lds, s>>>(
            nq0, nq1, nq2, a);
        return launch_rc();
    }
    // LDS-resident sweeps: bases + input image / w2 (aliased) + w1
    const size_t nmt = nm0 * nm1 * nm2, n1 = nq0 * nm1 * nm2, n2 = (size_t)nq0 * nq1 * nm2;
    const size_t lds_full = sizeof(T) * (nbas + (nmt > n2 ? nmt : n2) + n1);
    const bool bases_fit  = sizeof(T) * nbas <= kMaxDynLds;
    bool glb = (variant == SF_VARIANT_BLOCK_GLB), by_block = false;
    if (variant == SF_VARIANT_BLOCK_LDS && lds_full > kMaxDynLds)
        return SF_ENOTBUILT;
    if (variant == SF_VARIANT_GENERIC && lds_full > kMaxDynLds)
        glb = by_block = true; // the element's images exceed the 160 KiB of LDS: intermediates in the library's scratch
    if (glb && !by_block && !a.wsp)
        return SF_EINVAL;
    const unsigned nqt = nq0 * nq1 * nq2;
    const unsigned thr = hinted_block(nqt <= 64 ? 64 : (nqt <= 128 ? 128 : 256), nqt);
    const uint64_t cap = (uint64_t)cu * 32;
    unsigned grid      = hinted_grid(a.nelmt, cap);
    HexArgsT<T> args   = a;
    // scratch and launch are one unit (see scratch_acquire)
    std::unique_lock<std::recursive_mutex> lock(scratch_mutex(), std::defer_lock);
    if (by_block)
    {
        lock.lock();
        if (n1 + n2 > (~(size_t)0) / sizeof(T) / ((size_t)cu * 8))
            return SF_EINVAL; // the scratch size would overflow
        grid = by_block_grid(a.nelmt, cu, sizeof(T) * (uint64_t)(n1 + n2));
        void *p = nullptr;
        int rc  = scratch_acquire(s, 1, sizeof(T) * (n1 + n2) * grid, &p);
        if (rc != SF_OK)
            return rc;
        args.wsp = static_cast<T *>(p);
    }
    const size_t lds = glb ? (bases_fit ? sizeof(T) * nbas : 0) : lds_full;
    auto go          = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        kern<<<grid, thr, lds, s>>>(nq0, nq1, nq2, args, by_block ? 1 : 0);
    };
    if (!glb)
        go(hex_block_kernel<false, true, T>);
    else if (bases_fit)
        go(hex_block_kernel<true, true, T>);
    else
        go(hex_block_kernel<true, false, T>);
    return launch_rc();
}

template <typename T>
int launch_quad_generic_t(int variant, unsigned nq0, unsigned nq1, const QuadArgsT<T> &a, hipStream_t s)
{
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1;
    const size_t nbas = nm0 * nq0 + nm1 * nq1;
    const unsigned cu = (unsigned)device_info().num_cu;
    if (variant == SF_VARIANT_THREAD)
    {
        if (!a.wsp)
            return SF_EINVAL;
        const size_t lds = sizeof(T) * nbas;
        if (lds > kMaxDynLds)
            return SF_ENOTBUILT;
        const unsigned bs     = hinted_block(128, 0xffffffffu);
        const uint64_t blocks = (a.nelmt + bs - 1) / bs;
        quad_thread_kernel<T><<<(unsigned)(blocks > cu * 16 ? cu * 16 : blocks), bs, lds, s>>>(
            nq0, nq1, a);
        return launch_rc();
    }
    const size_t nmt = nm0 * nm1, n1 = nq0 * nm1;
    const size_t lds_full = sizeof(T) * (nbas + nmt + n1);
    const bool bases_fit  = sizeof(T) * nbas <= kMaxDynLds;
    bool glb = (variant == SF_VARIANT_BLOCK_GLB), by_block = false;
    if (variant == SF_VARIANT_BLOCK_LDS && lds_full > kMaxDynLds)
        return SF_ENOTBUILT;
    if (variant == SF_VARIANT_GENERIC && lds_full > kMaxDynLds)
        glb = by_block = true;
    if (glb && !by_block && !a.wsp)
        return SF_EINVAL;
    const unsigned nqt = nq0 * nq1;
    const unsigned thr = hinted_block(nqt <= 64 ? 64 : (nqt <= 128 ? 128 : 256), nqt);
    const uint64_t cap = (uint64_t)cu * 32;
    unsigned grid      = hinted_grid(a.nelmt, cap);
    QuadArgsT<T> args  = a;
    std::unique_lock<std::recursive_mutex> lock(scratch_mutex(), std::defer_lock);
    if (by_block)
    {
        lock.lock();
        if (n1 > (~(size_t)0) / sizeof(T) / ((size_t)cu * 8))
            return SF_EINVAL;
        grid = by_block_grid(a.nelmt, cu, sizeof(T) * (uint64_t)n1);
        void *p = nullptr;
        int rc  = scratch_acquire(s, 1, sizeof(T) * n1 * grid, &p);
        if (rc != SF_OK)
            return rc;
        args.wsp = static_cast<T *>(p);
    }
    const size_t lds = glb ? (bases_fit ? sizeof(T) * nbas : 0) : lds_full;
    auto go          = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        kern<<<grid, thr, lds, s>>>(nq0, nq1, args, by_block ? 1 : 0);
    };
    if (!glb)
        go(quad_block_kernel<false, true, T>);
    else if (bases_fit)
        go(quad_block_kernel<true, true, T>);
    else
        go(quad_block_kernel<true, false, T>);
    return launch_rc();
}

int launch_hex_generic(int variant, unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a,
                       hipStream_t s)
{
    return launch_hex_generic_t<double>(variant, nq0, nq1, nq2, a, s);
}

int launch_hex_interleaved(unsigned nq0, unsigned nq1, unsigned nq2, const HexArgs &a, hipStream_t s)
{
    const size_t nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
    const size_t lds = sizeof(double) * (nm0 * nq0 + nm1 * nq1 + nm2 * nq2);
    if (lds > kMaxDynLds)
        return SF_ENOTBUILT;
    if (!a.wsp)
        return SF_EINVAL;
    const unsigned cu     = (unsigned)device_info().num_cu;
    const uint64_t blocks = (a.nelmt + 255) / 256;
    hex_interleaved_kernel<<<(unsigned)(blocks > cu * 16 ? cu * 16 : blocks), 256, lds, s>>>(nq0, nq1,
                                                                                            nq2, a);
    return launch_rc();
}

int launch_interleave64(const double *src, double *dst, size_t nelmt, size_t n, int inverse,
                        hipStream_t s)
{
    // the interleaved side is padded to whole groups of 64 elements
    const uint64_t padded = (nelmt + kWave - 1) / kWave * kWave;
    const uint64_t total  = padded * n;
    if (total == 0)
        return SF_OK;
    const uint64_t want = (total + 255) / 256, cap = (uint64_t)device_info().num_cu * 32;
    interleave64_kernel<<<(unsigned)(want > cap ? cap : want), 256, 0, s>>>(src, dst, nelmt, n,
                                                                            inverse);
    return launch_rc();
}
int launch_hex_generic_f32(int variant, unsigned nq0, unsigned nq1, unsigned nq2,
                           const HexArgsT<float> &a, hipStream_t s)
{
    return launch_hex_generic_t<float>(variant, nq0, nq1, nq2, a, s);
}
int launch_quad_generic(int variant, unsigned nq0, unsigned nq1, const QuadArgs &a, hipStream_t s)
{
    return launch_quad_generic_t<double>(variant, nq0, nq1, a, s);
}
int launch_quad_generic_f32(int variant, unsigned nq0, unsigned nq1, const QuadArgsT<float> &a,
                            hipStream_t s)
{
    return launch_quad_generic_t<float>(variant, nq0, nq1, a, s);
}

} // namespace sf
