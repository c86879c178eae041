// aux_kernels.hip -- initialisers, sum-of-squares reduction and the HBM stream calibrator.
//
//   fill_sincos / fill_basis : device-side version of the host init loops
//                              benchmark05/benchmark05.cc:1195-1236, benchmark04/benchmark04.cc:859-889
//   fill_l2norm              : benchmark01/benchmark01.cc:171-181 (set_data)
//   fill_random              : per-value-distinct seeded data (not in the reference; same generator as
//                              oracle_fill_random so host and device arrays agree bit for bit)
//   sumsq                    : thrust::transform_reduce(x*x, plus) of benchmark05.cc:1273-1276 and the
//                              l2norm kernels of benchmark01/benchmark01.cc:15-77, 112-169
//   stream_copy              : bandwidth calibrator
//   vector_add / fill_vecadd : benchmark02 (x += y; benchmark02/benchmark02.cc:16-58, data :84-85)
//   matvec / fill_matvec     : benchmark03 (y = A x; benchmark03/benchmark03.cc:15-104, data :160-167)
#include "sf_dispatch.h"

#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace sf
{

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_sincos_kernel(double *__restrict__ in, uint64_t total,
                                                          uint32_t nm_tot)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride)
    {
        const uint32_t f = (uint32_t)(x % nm_tot);
        in[x]            = sin((double)(f + 1u));
    }
}

__global__ __launch_bounds__(256) void fill_basis_kernel(double *__restrict__ b, uint32_t n)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n)
        b[x] = cos((double)x);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_random_kernel(double *__restrict__ x, uint64_t n,
                                                          uint64_t seed, uint64_t first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const uint64_t h = mix64(seed ^ mix64(first + i));
        x[i]             = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

__global__ __launch_bounds__(256) void fill_l2norm_kernel(double *__restrict__ x, uint64_t n)
{
#pragma clang fp contract(off) // same roundings as the host statement (no FMA): bit-exact
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const uint32_t u = (uint32_t)i;
        const double t   = 0.00001 * (double)(u % 100191u);
        x[i]             = (double)(u % 13u) + (0.2 + t);
    }
}

// ---------------------------------------------------------------------------------------------
// sum of squares: pass 1 = per-block partials (fixed grid for a given n -> deterministic),
// pass 2 = one block folds the partials.  16-B lanes, 4 independent accumulators per lane,
// wave-64 shuffle tree, one LDS slot per wave.
// ---------------------------------------------------------------------------------------------
constexpr int kRedThreads  = 256;
constexpr int kRedMaxBlock = 65536; // partial sums per reduction (workspace: 512 KB)
constexpr int kRedUnroll   = 8;     // 16-byte vectors per thread and tile (tile = 32 KB)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, kWave);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    if (lane == 0)
        red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
    {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i)
            s += red[i];
    }
    return s; // valid in thread 0
}

// Workgroup b sums the contiguous run of `tpb` 32-KB tiles starting at tile b*tpb and writes one partial; the grid
// covers the array (no grid-stride loop): the dispatcher-ordered front reads at 6.5-7 TB/s where the persistent
// grid-stride version of this kernel read at 5.4 TB/s.  Shape and order depend on n only: deterministic.
__global__ __launch_bounds__(kRedThreads) void sumsq_partial_kernel(const double *__restrict__ x,
                                                                    uint64_t n, uint32_t tpb,
                                                                    double *__restrict__ part)
{
    __shared__ double red[kRedThreads / kWave];
    const uint64_t nv   = n / 2; // double2 units (x is 16-B aligned, checked on the host)
    const double2_t *x2 = reinterpret_cast<const double2_t *>(x);
    constexpr uint64_t tile = (uint64_t)kRedThreads * kRedUnroll;
    uint64_t v = (uint64_t)blockIdx.x * tpb * tile + threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (uint32_t t = 0; t < tpb; ++t, v += tile)
    {
        if (v + (kRedUnroll - 1) * kRedThreads < nv)
        {
            double2_t p[kRedUnroll];
#pragma unroll
            for (int u = 0; u < kRedUnroll; ++u)
                p[u] = __builtin_nontemporal_load(x2 + v + u * kRedThreads);
#pragma unroll
            for (int u = 0; u < kRedUnroll; u += 2)
            {
                a0 = __builtin_fma(p[u].x, p[u].x, a0);
                a1 = __builtin_fma(p[u].y, p[u].y, a1);
                a2 = __builtin_fma(p[u + 1].x, p[u + 1].x, a2);
                a3 = __builtin_fma(p[u + 1].y, p[u + 1].y, a3);
            }
        }
        else
        {
            for (int u = 0; u < kRedUnroll; ++u)
            {
                const uint64_t w = v + (uint64_t)u * kRedThreads;
                if (w < nv)
                {
                    const double2_t q = x2[w];
                    if (u & 1)
                    {
                        a2 = __builtin_fma(q.x, q.x, a2);
                        a3 = __builtin_fma(q.y, q.y, a3);
                    }
                    else
                    {
                        a0 = __builtin_fma(q.x, q.x, a0);
                        a1 = __builtin_fma(q.y, q.y, a1);
                    }
                }
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        a2 = __builtin_fma(x[n - 1], x[n - 1], a2);
    const double s = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0)
        part[blockIdx.x] = s;
}

// Large arrays: a persistent grid of 6 workgroups per CU, thread t takes the 16-byte vectors t, t + T, t + 2T ... with ONE
// non-temporal load in flight.  A read-only stream behaves unlike the read/write mixes of the BwdTrans kernels: this
// shape reads at 7.1-7.2 TB/s from 1 GB up (6.8 at 268 MB) where the run-of-tiles shape above reaches 6.4-6.75 and
// deeper unrolling or larger grids lose 5-20 % (tools/sf_membench9, profiles/r02/membench9_read_only_reduction.log).
// Shape and order depend on n and the CU count only: deterministic on a given device.
__global__ __launch_bounds__(kRedThreads) void sumsq_stride_kernel(const double *__restrict__ x, uint64_t n,
                                                                   double *__restrict__ part)
{
    __shared__ double red[kRedThreads / kWave];
    const uint64_t nv   = n / 2;
    const double2_t *x2 = reinterpret_cast<const double2_t *>(x);
    const uint64_t T    = (uint64_t)gridDim.x * kRedThreads;
    double a0 = 0.0, a1 = 0.0;
    for (uint64_t v = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x; v < nv; v += T)
    {
        const double2_t p = __builtin_nontemporal_load(x2 + v);
        a0                = __builtin_fma(p.x, p.x, a0);
        a1                = __builtin_fma(p.y, p.y, a1);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        a0 = __builtin_fma(x[n - 1], x[n - 1], a0);
    const double s = block_sum(a0 + a1, red);
    if (threadIdx.x == 0)
        part[blockIdx.x] = s;
}
constexpr uint64_t kStrideMinDoubles = 1ull << 24; // 134 MB: below, the run-of-tiles shape is as fast or faster

__global__ __launch_bounds__(kRedThreads) void sumsq_final_kernel(const double *__restrict__ part,
                                                                  int npart,
                                                                  double *__restrict__ result)
{
    __shared__ double red[kRedThreads / kWave];
    double a = 0.0;
    for (int i = threadIdx.x; i < npart; i += kRedThreads)
        a += part[i];
    const double s = block_sum(a, red);
    if (threadIdx.x == 0)
        result[0] = s;
}

// scalar (8-byte lanes) variant for unaligned x
__global__ __launch_bounds__(kRedThreads) void sumsq_partial_scalar_kernel(
    const double *__restrict__ x, uint64_t n, double *__restrict__ part)
{
    __shared__ double red[kRedThreads / kWave];
    const uint64_t stride = (uint64_t)gridDim.x * kRedThreads;
    double a              = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += stride)
        a = __builtin_fma(x[i], x[i], a);
    const double s = block_sum(a, red);
    if (threadIdx.x == 0)
        part[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// One 16-byte lane per thread and a grid that covers the whole array (no grid-stride loop): on this
// part a dispatcher-ordered huge grid streams ~6.6 TB/s where a persistent grid-stride copy tops out
// at ~5.7 TB/s (profiles/r01/membench1.log).
__global__ __launch_bounds__(256) void stream_copy_flat_kernel(const double2_t *__restrict__ src,
                                                               double2_t *__restrict__ dst,
                                                               uint64_t nv)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + v), dst + v);
}

// x += y (benchmark02's kernel): 16 B of x and y in, 16 B of x out per thread
__global__ __launch_bounds__(256) void vector_add_kernel(double *__restrict__ x,
                                                         const double *__restrict__ y, uint64_t n)
{
    // runs of 256 neighbouring workgroups (1 MiB of each stream) per XCD: 6.45 -> 6.65 TB/s
    // (tools/sf_membench7, profiles/r01/membench7_xcd_runs_on_streams.log)
    const uint64_t v  = logical_block<256>() * 256 + threadIdx.x;
    const uint64_t nv = n / 2;
    if (v < nv)
    {
        double2_t *x2       = reinterpret_cast<double2_t *>(x) + v;
        const double2_t a   = __builtin_nontemporal_load(x2);
        const double2_t b   = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(y) + v);
        __builtin_nontemporal_store(a + b, x2);
    }
    else if (v == nv && (n & 1))
        x[n - 1] += y[n - 1];
}

__global__ __launch_bounds__(256) void vector_add_scalar_kernel(double *__restrict__ x,
                                                                const double *__restrict__ y,
                                                                uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        x[i] += y[i];
}

__global__ __launch_bounds__(256) void fill_vecadd_kernel(double *__restrict__ x,
                                                          double *__restrict__ y, uint64_t n)
{
#pragma clang fp contract(off) // bit-identical to the host statement
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const uint32_t u = (uint32_t)i;
        const double t1 = 0.00001 * (double)(u % 100191u), t2 = 0.00003 * (double)(u % 100721u);
        x[i] = (double)(u % 13u) + (0.2 + t1);
        y[i] = (double)(u % 8u) + (0.4 + t2);
    }
}

// y = A x, A row-major M x N: one wavefront per row, 16-byte lanes over the row, x from cache,
// wave-64 shuffle tree; summation order fixed by (N, lane) -> deterministic.
__global__ __launch_bounds__(256) void matvec_kernel(uint32_t M, uint32_t N,
                                                     const double *__restrict__ A,
                                                     const double *__restrict__ x,
                                                     double *__restrict__ y)
{
    const int lane     = threadIdx.x & (kWave - 1);
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M)
        return;
    const double *a = A + (uint64_t)row * N;
    double s0 = 0.0, s1 = 0.0;
    if ((N & 1u) == 0 && (((uintptr_t)A | (uintptr_t)x) & 15u) == 0)
    {
        const double2_t *a2 = reinterpret_cast<const double2_t *>(a);
        const double2_t *x2 = reinterpret_cast<const double2_t *>(x);
        for (uint32_t v = lane; v < N / 2; v += kWave)
        {
            const double2_t p = __builtin_nontemporal_load(a2 + v), q = x2[v];
            s0 = __builtin_fma(p.x, q.x, s0);
            s1 = __builtin_fma(p.y, q.y, s1);
        }
    }
    else
    {
        for (uint32_t j = lane; j < N; j += kWave)
            s0 = __builtin_fma(a[j], x[j], s0);
    }
    const double s = wave_sum(s0 + s1);
    if (lane == 0)
        y[row] = s;
}

__global__ __launch_bounds__(256) void fill_matvec_kernel(double *__restrict__ A,
                                                          double *__restrict__ x, uint64_t total,
                                                          uint32_t N)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        A[i] = sin((double)(i + 1));
        if (i < N)
            x[i] = (double)i;
    }
}

// ---- fp32 counterparts (T = float; the reference's templates allow it, benchmark05.cc:15, 1129-1141) -----
__global__ __launch_bounds__(256) void fill_sincos_f32_kernel(float *__restrict__ in, uint64_t total,
                                                              uint32_t nm_tot)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride)
        in[x] = sinf((float)((uint32_t)(x % nm_tot) + 1u)); // sin((T)(f+1)) with T = float
}

__global__ __launch_bounds__(256) void fill_basis_f32_kernel(float *__restrict__ b, uint32_t n)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n)
        b[x] = cosf((float)x);
}

// the fp64 generator's value rounded to float: host and device still agree bit for bit
__global__ __launch_bounds__(256) void fill_random_f32_kernel(float *__restrict__ x, uint64_t n,
                                                              uint64_t seed, uint64_t first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const uint64_t h = mix64(seed ^ mix64(first + i));
        x[i]             = (float)((double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0);
    }
}

// sum of squares of floats, accumulated in double (float accumulation loses the check's digits)
__global__ __launch_bounds__(kRedThreads) void sumsq_partial_f32_kernel(const float *__restrict__ x,
                                                                        uint64_t n,
                                                                        double *__restrict__ part)
{
    __shared__ double red[kRedThreads / kWave];
    const uint64_t stride = (uint64_t)gridDim.x * kRedThreads;
    double a0 = 0.0, a1 = 0.0;
    uint64_t i = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x;
    for (; i + stride < n; i += 2 * stride)
    {
        const double p = x[i], q = x[i + stride];
        a0 = __builtin_fma(p, p, a0);
        a1 = __builtin_fma(q, q, a1);
    }
    if (i < n)
    {
        const double p = x[i];
        a0             = __builtin_fma(p, p, a0);
    }
    const double s = block_sum(a0 + a1, red);
    if (threadIdx.x == 0)
        part[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
const DeviceInfo &device_info()
{
    static DeviceInfo info[64];
    static bool have[64] = {};
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64)
        dev = 0;
    if (!have[dev])
    {
        int cu = 256;
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        info[dev].num_cu = cu > 0 ? cu : 256;
        info[dev].device = dev;
        have[dev]        = true;
    }
    return info[dev];
}

// Internal scratch: one buffer per (device, stream, kind), grown on demand.  Work enqueued on ONE stream is ordered, so
// a buffer keyed by its stream is never used by two kernels at once; different streams (or devices) get different
// buffers, which is what makes the entry points below safe to call concurrently on several streams / host threads.
// g_scratch_mu covers the table AND is held by the callers across their whole enqueue sequence (kernel(s) + async
// copy), so two host threads sharing a stream cannot interleave halves of two reductions.
struct ScratchSlot
{
    int dev;
    hipStream_t stream;
    uint64_t owner; // hipStreamPerThread names a different stream in every host thread: those slots are per thread
    int kind;
    void *ptr;
    size_t bytes;
    uint64_t tick;
};
static std::vector<ScratchSlot> g_scratch;
static std::recursive_mutex g_scratch_mu;
static uint64_t g_scratch_tick = 0;
constexpr size_t kMaxScratchSlots = 128;

std::recursive_mutex &scratch_mutex()
{
    return g_scratch_mu;
}

int scratch_acquire(hipStream_t s, int kind, size_t bytes, void **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return SF_EINVAL;
    std::lock_guard<std::recursive_mutex> lock(g_scratch_mu);
    const uint64_t owner =
        s == hipStreamPerThread ? (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1u : 0;
    ScratchSlot *slot = nullptr;
    for (auto &c : g_scratch)
        if (c.dev == dev && c.stream == s && c.owner == owner && c.kind == kind)
            slot = &c;
    if (!slot)
    {
        if (g_scratch.size() >= kMaxScratchSlots)
        {
            // evict the least recently used slot of this device (hipFree waits for work that still uses it)
            size_t victim = g_scratch.size();
            for (size_t i = 0; i < g_scratch.size(); ++i)
                if (g_scratch[i].dev == dev && (victim == g_scratch.size() || g_scratch[i].tick < g_scratch[victim].tick))
                    victim = i;
            if (victim == g_scratch.size())
                return SF_ENOMEM;
            (void)hipFree(g_scratch[victim].ptr);
            g_scratch.erase(g_scratch.begin() + (long)victim);
        }
        g_scratch.push_back(ScratchSlot{dev, s, owner, kind, nullptr, 0, 0});
        slot = &g_scratch.back();
    }
    if (slot->bytes < bytes)
    {
        if (slot->ptr)
            (void)hipFree(slot->ptr); // synchronises: nothing in flight still reads the old buffer
        slot->ptr = nullptr;
        slot->bytes = 0;
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess)
        {
            (void)hipGetLastError();
            return SF_ENOMEM;
        }
        slot->ptr   = p;
        slot->bytes = bytes;
    }
    slot->tick = ++g_scratch_tick;
    *out       = slot->ptr;
    return SF_OK;
}

// ---- batch counters of the persistent 2D kernels (sf_dispatch.h: counter_acquire) -----------------------------------
constexpr unsigned kCounterSlots  = 8192; // per device
constexpr unsigned kCounterStride = 64;   // bytes: one line per counter
struct CounterRing
{
    char *base    = nullptr;
    unsigned next = 0;
    std::vector<std::pair<hipStream_t, unsigned>> eager; // stream -> slot (per-thread streams: one slot per launch)
};
static CounterRing g_counters[64];
static std::mutex g_counter_mu;

int counter_acquire(hipStream_t s, unsigned long long **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return SF_EINVAL;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess)
    {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_counter_mu);
    CounterRing &r = g_counters[dev];
    if (!r.base)
    {
        if (capturing) // an allocation would invalidate the capture
            return SF_ENOMEM;
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)kCounterSlots * kCounterStride) != hipSuccess)
        {
            (void)hipGetLastError();
            return SF_ENOMEM;
        }
        (void)hipMemset(p, 0, (size_t)kCounterSlots * kCounterStride);
        r.base = static_cast<char *>(p);
    }
    unsigned slot = kCounterSlots;
    if (!capturing && s != hipStreamPerThread)
        for (auto &e : r.eager)
            if (e.first == s)
                slot = e.second;
    if (slot == kCounterSlots)
    {
        if (r.next >= kCounterSlots)
            return SF_ENOMEM;
        slot = r.next++;
        if (!capturing && s != hipStreamPerThread)
            r.eager.emplace_back(s, slot);
    }
    *out = reinterpret_cast<unsigned long long *>(r.base + (size_t)slot * kCounterStride);
    return SF_OK;
}

// sf_shutdown(): graphs captured earlier must not be replayed afterwards
int release_counters()
{
    std::lock_guard<std::mutex> lock(g_counter_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && g_counters[dev].base)
    {
        (void)hipFree(g_counters[dev].base);
        g_counters[dev] = CounterRing{};
    }
    return SF_OK;
}

// pinned 8-byte landing slot of the calling host thread: the result of a blocking reduction is copied there
// asynchronously, so no lock is held while the device works (a copy into pageable memory would block inside the lock)
struct PinnedSlot
{
    double *p = nullptr;
    ~PinnedSlot()
    {
        if (p)
            (void)hipHostFree(p);
    }
};
static thread_local PinnedSlot t_landing;
static double *landing_slot()
{
    if (!t_landing.p)
    {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable) != hipSuccess)
        {
            (void)hipGetLastError();
            return nullptr;
        }
        t_landing.p = static_cast<double *>(p);
    }
    return t_landing.p;
}

struct Workspace
{
    double *part   = nullptr; // kRedMaxBlock partials + 1 result
    double *result = nullptr;
};

// reduction scratch of the calling stream (kind 0)
static int workspace(hipStream_t s, Workspace *ws)
{
    void *p = nullptr;
    int rc  = scratch_acquire(s, 0, sizeof(double) * (kRedMaxBlock + 8), &p);
    if (rc != SF_OK)
        return rc;
    ws->part   = static_cast<double *>(p);
    ws->result = ws->part + kRedMaxBlock;
    return SF_OK;
}

int release_workspaces()
{
    std::lock_guard<std::recursive_mutex> lock(g_scratch_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (size_t i = 0; i < g_scratch.size();)
    {
        if (g_scratch[i].dev == dev)
        {
            (void)hipFree(g_scratch[i].ptr);
            g_scratch.erase(g_scratch.begin() + (long)i);
        }
        else
            ++i;
    }
    return SF_OK;
}

static inline int launch_rc()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

static inline unsigned fill_grid(uint64_t n)
{
    const uint64_t want = (n + 255) / 256;
    const uint64_t cap  = (uint64_t)device_info().num_cu * 16;
    return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

int sumsq_async(const double *x, size_t n, double *result_dev, hipStream_t s)
{
    std::lock_guard<std::recursive_mutex> lock(g_scratch_mu);
    Workspace w, *ws = &w;
    int rc = workspace(s, ws);
    if (rc != SF_OK)
        return rc;
    // fixed shape for a given n: deterministic result
    const uint64_t tile  = (uint64_t)kRedThreads * kRedUnroll;
    const uint64_t tiles = (n / 2 + tile - 1) / tile;
    uint64_t blocks;
    if (((uintptr_t)x & 15u) == 0 && n >= kStrideMinDoubles)
    {
        blocks = 6ull * (uint64_t)device_info().num_cu;
        sumsq_stride_kernel<<<(unsigned)blocks, kRedThreads, 0, s>>>(x, n, ws->part);
    }
    else if (((uintptr_t)x & 15u) == 0)
    {
        const uint64_t tpb = tiles <= (uint64_t)kRedMaxBlock ? 1 : (tiles + kRedMaxBlock - 1) / kRedMaxBlock;
        if (tpb > 0xffffffffull)
            return SF_EINVAL;
        blocks = tiles < 1 ? 1 : (tiles + tpb - 1) / tpb;
        sumsq_partial_kernel<<<(unsigned)blocks, kRedThreads, 0, s>>>(x, n, (uint32_t)tpb, ws->part);
    }
    else
    {
        blocks = tiles < 1 ? 1 : (tiles > 2048 ? 2048 : tiles);
        sumsq_partial_scalar_kernel<<<(unsigned)blocks, kRedThreads, 0, s>>>(x, n, ws->part);
    }
    sumsq_final_kernel<<<1, kRedThreads, 0, s>>>(ws->part, (int)blocks,
                                                 result_dev ? result_dev : ws->result);
    return launch_rc();
}

int sumsq_blocking(const double *x, size_t n, double *result_host, hipStream_t s)
{
    hipError_t e;
    double *land = landing_slot();
    if (!land)
        return SF_ENOMEM;
    {
        // kernels + copy are enqueued as one unit; the copy lands in this thread's pinned slot, so it is asynchronous
        // and the wait happens outside the lock
        std::lock_guard<std::recursive_mutex> lock(g_scratch_mu);
        Workspace w, *ws = &w;
        int rc = workspace(s, ws);
        if (rc != SF_OK)
            return rc;
        rc = sumsq_async(x, n, ws->result, s);
        if (rc != SF_OK)
            return rc;
        e = hipMemcpyAsync(land, ws->result, sizeof(double), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        *result_host = *land;
    return e == hipSuccess ? SF_OK : (int)e;
}

int sumsq_f32_blocking(const float *x, size_t n, double *result_host, hipStream_t s)
{
    hipError_t e;
    double *land = landing_slot();
    if (!land)
        return SF_ENOMEM;
    {
        std::lock_guard<std::recursive_mutex> lock(g_scratch_mu);
        Workspace w, *ws = &w;
        int rc = workspace(s, ws);
        if (rc != SF_OK)
            return rc;
        uint64_t blocks = (n + (uint64_t)kRedThreads * 16 - 1) / ((uint64_t)kRedThreads * 16);
        if (blocks < 1)
            blocks = 1;
        if (blocks > kRedMaxBlock)
            blocks = kRedMaxBlock;
        sumsq_partial_f32_kernel<<<(unsigned)blocks, kRedThreads, 0, s>>>(x, n, ws->part);
        sumsq_final_kernel<<<1, kRedThreads, 0, s>>>(ws->part, (int)blocks, ws->result);
        rc = launch_rc();
        if (rc != SF_OK)
            return rc;
        e = hipMemcpyAsync(land, ws->result, sizeof(double), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        *result_host = *land;
    return e == hipSuccess ? SF_OK : (int)e;
}

int fill_sincos_f32(float *in, size_t nelmt, size_t nm_tot, hipStream_t s)
{
    const uint64_t total = (uint64_t)nelmt * nm_tot;
    if (total == 0)
        return SF_OK;
    fill_sincos_f32_kernel<<<fill_grid(total), 256, 0, s>>>(in, total, (uint32_t)nm_tot);
    return launch_rc();
}

int fill_basis_f32(float *b, size_t nm, size_t nq, hipStream_t s)
{
    const uint32_t n = (uint32_t)(nm * nq);
    if (n == 0)
        return SF_OK;
    fill_basis_f32_kernel<<<(n + 255) / 256, 256, 0, s>>>(b, n);
    return launch_rc();
}

int fill_random_f32(float *x, size_t n, uint64_t seed, uint64_t first, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    fill_random_f32_kernel<<<fill_grid(n), 256, 0, s>>>(x, n, seed, first);
    return launch_rc();
}

int fill_sincos(double *in, size_t nelmt, size_t nm_tot, hipStream_t s)
{
    const uint64_t total = (uint64_t)nelmt * nm_tot;
    if (total == 0)
        return SF_OK;
    fill_sincos_kernel<<<fill_grid(total), 256, 0, s>>>(in, total, (uint32_t)nm_tot);
    return launch_rc();
}

int fill_basis(double *b, size_t nm, size_t nq, hipStream_t s)
{
    const uint32_t n = (uint32_t)(nm * nq);
    if (n == 0)
        return SF_OK;
    fill_basis_kernel<<<(n + 255) / 256, 256, 0, s>>>(b, n);
    return launch_rc();
}

int fill_random(double *x, size_t n, uint64_t seed, uint64_t first, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    fill_random_kernel<<<fill_grid(n), 256, 0, s>>>(x, n, seed, first);
    return launch_rc();
}

int fill_l2norm(double *x, size_t n, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    fill_l2norm_kernel<<<fill_grid(n), 256, 0, s>>>(x, n);
    return launch_rc();
}

int stream_copy(const double *src, double *dst, size_t n, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) != 0 || (n & 1))
        return SF_EALIGN;
    const uint64_t nv     = n / 2;
    const uint64_t blocks = (nv + 255) / 256;
    if (blocks > 0x7fffffffull)
        return SF_EINVAL;
    stream_copy_flat_kernel<<<(unsigned)blocks, 256, 0, s>>>(
        reinterpret_cast<const double2_t *>(src), reinterpret_cast<double2_t *>(dst), nv);
    return launch_rc();
}

int vector_add(double *x, const double *y, size_t n, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) == 0)
    {
        const uint64_t blocks = (n / 2 + 1 + 255) / 256;
        if (blocks > 0x7fffffffull)
            return SF_EINVAL;
        vector_add_kernel<<<(unsigned)blocks, 256, 0, s>>>(x, y, n);
    }
    else
    {
        const uint64_t blocks = (n + 255) / 256;
        if (blocks > 0x7fffffffull)
            return SF_EINVAL;
        vector_add_scalar_kernel<<<(unsigned)blocks, 256, 0, s>>>(x, y, n);
    }
    return launch_rc();
}

int fill_vecadd(double *x, double *y, size_t n, hipStream_t s)
{
    if (n == 0)
        return SF_OK;
    fill_vecadd_kernel<<<fill_grid(n), 256, 0, s>>>(x, y, n);
    return launch_rc();
}

int matvec(unsigned M, unsigned N, const double *A, const double *x, double *y, hipStream_t s)
{
    if (M == 0)
        return SF_OK;
    matvec_kernel<<<(M + 3) / 4, 256, 0, s>>>(M, N, A, x, y);
    return launch_rc();
}

int fill_matvec(double *A, double *x, unsigned M, unsigned N, hipStream_t s)
{
    const uint64_t total = (uint64_t)M * N;
    if (total == 0)
        return SF_OK;
    fill_matvec_kernel<<<fill_grid(total), 256, 0, s>>>(A, x, total, N);
    return launch_rc();
}

} // namespace sf
