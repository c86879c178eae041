// multigpu.h -- the aggregate row of benchmark05: `--ngpus N`, ONE process, N devices.
//
// The reference is single-GPU (CUDA_VISIBLE_DEVICES=1, benchmark05/run.sh:7); the element batch is embarrassingly
// parallel, so it shards as contiguous element ranges [g*nelmt/N, (g+1)*nelmt/N), one per device, each with its own
// stream and its own buffers.  No element data crosses GPUs.  RCCL (ncclCommInitAll over the N devices)
// carries two scalars per size: MAX over devices of the best kernel time and SUM of the per-device sum of squares.
// Timing protocol of the reference (benchmark05.cc:1319-1332) per repetition -- host wall clock around "launch on every
// device, synchronise every device"; the row reports total DOF / the best such wall time, so it is comparable with the
// single-GPU rows (same clock, launch loop included).  Per-device HIP events go to the --json side file only: MAX over
// devices inside a repetition then min over repetitions, and the RCCL MAX of the per-device minima.
#pragma once

#include <rccl/rccl.h>

#include "harness.h"

#define NCCL_CHECK(expr)                                                                           \
    do                                                                                             \
    {                                                                                              \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
        {                                                                                          \
            std::cerr << "RCCL error: " << ncclGetErrorString(r_) << " at " << __FILE__ << ":"     \
                      << __LINE__ << " (" #expr ")" << std::endl;                                  \
            std::exit(6);                                                                          \
        }                                                                                          \
    } while (0)

namespace harness
{

// contiguous element range of device g (sizes differ by at most one element; same rule as shard.py)
inline void shard_range(size_t total, int ngpus, int g, size_t *lo, size_t *hi)
{
    *lo = total * (size_t)g / (size_t)ngpus;
    *hi = total * (size_t)(g + 1) / (size_t)ngpus;
}

struct MultiGpuResult
{
    double t_max_event_s = 0.0; // RCCL MAX over devices of each device's best (min over repetitions) kernel time, HIP events
    double t_event_rep_s = 0.0; // min over repetitions of (MAX over devices of that repetition's kernel time)
    double t_wall_s      = 0.0; // best host wall time of one "launch everywhere + synchronise everywhere": the row's time
    double t_solo_wall_s = 0.0; // device 0 alone over the WHOLE batch, same wall protocol (0: does not fit / not run)
    double sumsq         = 0.0; // SUM over devices
    std::vector<double> per_device_s;
};

class MultiGpu
{
public:
    explicit MultiGpu(int n) : m_n(n), m_comms(n), m_streams(n), m_red(n)
    {
        std::vector<int> devs(n);
        for (int g = 0; g < n; ++g)
            devs[g] = g;
        NCCL_CHECK(ncclCommInitAll(m_comms.data(), n, devs.data()));
        for (int g = 0; g < n; ++g)
        {
            HIP_CHECK(hipSetDevice(g));
            HIP_CHECK(hipStreamCreate(&m_streams[g]));
            HIP_CHECK(hipMalloc((void **)&m_red[g], 2 * sizeof(double)));
        }
    }
    ~MultiGpu()
    {
        for (int g = 0; g < m_n; ++g)
        {
            (void)hipSetDevice(g);
            (void)hipFree(m_red[g]);
            (void)hipStreamDestroy(m_streams[g]);
            (void)ncclCommDestroy(m_comms[g]);
        }
        (void)hipSetDevice(0);
    }
    int size() const
    {
        return m_n;
    }
    hipStream_t stream(int g) const
    {
        return m_streams[g];
    }
    // every device contributes (time, sumsq); returns (MAX time, SUM sumsq) as seen by device 0
    void reduce(const std::vector<double> &t, const std::vector<double> &ss, double *tmax, double *sum)
    {
        for (int g = 0; g < m_n; ++g)
        {
            HIP_CHECK(hipSetDevice(g));
            const double v[2] = {t[g], ss[g]};
            HIP_CHECK(hipMemcpyAsync(m_red[g], v, sizeof v, hipMemcpyHostToDevice, m_streams[g]));
            HIP_CHECK(hipStreamSynchronize(m_streams[g])); // `v` is a stack temporary
        }
        NCCL_CHECK(ncclGroupStart());
        for (int g = 0; g < m_n; ++g)
            NCCL_CHECK(ncclAllReduce(m_red[g], m_red[g], 1, ncclDouble, ncclMax, m_comms[g], m_streams[g]));
        NCCL_CHECK(ncclGroupEnd());
        NCCL_CHECK(ncclGroupStart());
        for (int g = 0; g < m_n; ++g)
            NCCL_CHECK(ncclAllReduce(m_red[g] + 1, m_red[g] + 1, 1, ncclDouble, ncclSum, m_comms[g], m_streams[g]));
        NCCL_CHECK(ncclGroupEnd());
        double out[2];
        HIP_CHECK(hipSetDevice(0));
        HIP_CHECK(hipMemcpyAsync(out, m_red[0], sizeof out, hipMemcpyDeviceToHost, m_streams[0]));
        HIP_CHECK(hipStreamSynchronize(m_streams[0]));
        *tmax = out[0];
        *sum  = out[1];
    }

private:
    int m_n;
    std::vector<ncclComm_t> m_comms;
    std::vector<hipStream_t> m_streams;
    std::vector<double *> m_red;
};

// The 3D flagship on N devices.  `random`: seeded per-value-distinct data generated from the GLOBAL element index, so
// the N shards are exactly the slices of the one-GPU array; otherwise the reference's sin/cos data.
inline MultiGpuResult run_hex_multi(MultiGpu &mg, size_t nelmt, unsigned nq0, unsigned nq1, unsigned nq2, int variant,
                                    bool random, unsigned seed, bool solo = true)
{
    const int n        = mg.size();
    const size_t nmTot = (size_t)(nq0 - 1) * (nq1 - 1) * (nq2 - 1), nqTot = (size_t)nq0 * nq1 * nq2;
    std::vector<double *> in(n), out(n), b0(n), b1(n), b2(n);
    std::vector<size_t> cnt(n);
    std::vector<hipEvent_t> e0(n), e1(n);
    for (int g = 0; g < n; ++g)
    {
        size_t lo, hi;
        shard_range(nelmt, n, g, &lo, &hi);
        cnt[g] = hi - lo;
        HIP_CHECK(hipSetDevice(g));
        HIP_CHECK(hipMalloc((void **)&in[g], std::max<size_t>(1, cnt[g] * nmTot) * sizeof(double)));
        HIP_CHECK(hipMalloc((void **)&out[g], std::max<size_t>(1, cnt[g] * nqTot) * sizeof(double)));
        HIP_CHECK(hipMalloc((void **)&b0[g], (nq0 - 1) * nq0 * sizeof(double)));
        HIP_CHECK(hipMalloc((void **)&b1[g], (nq1 - 1) * nq1 * sizeof(double)));
        HIP_CHECK(hipMalloc((void **)&b2[g], (nq2 - 1) * nq2 * sizeof(double)));
        HIP_CHECK(hipEventCreate(&e0[g]));
        HIP_CHECK(hipEventCreate(&e1[g]));
        hipStream_t s = mg.stream(g);
        if (random)
            SF_CHECK(sf_fill_random_f64(in[g], cnt[g] * nmTot, seed, lo * nmTot, s));
        else
            SF_CHECK(sf_fill_sincos_f64(in[g], cnt[g], nmTot, s));
        SF_CHECK(sf_fill_basis_f64(b0[g], nq0 - 1, nq0, s));
        SF_CHECK(sf_fill_basis_f64(b1[g], nq1 - 1, nq1, s));
        SF_CHECK(sf_fill_basis_f64(b2[g], nq2 - 1, nq2, s));
        HIP_CHECK(hipMemsetAsync(out[g], 0, cnt[g] * nqTot * sizeof(double), s));
    }
    auto launch_all = [&](bool timed)
    {
        for (int g = 0; g < n; ++g)
        {
            HIP_CHECK(hipSetDevice(g));
            hipStream_t s = mg.stream(g);
            if (timed)
                HIP_CHECK(hipEventRecord(e0[g], s));
            SF_CHECK(sf_bwdtrans_hex_f64_variant(variant, nq0, nq1, nq2, cnt[g], b0[g], b1[g], b2[g], in[g], nullptr,
                                                 out[g], s));
            if (timed)
                HIP_CHECK(hipEventRecord(e1[g], s));
        }
        for (int g = 0; g < n; ++g)
        {
            HIP_CHECK(hipSetDevice(g));
            HIP_CHECK(hipStreamSynchronize(mg.stream(g)));
        }
    };
    launch_all(false); // first touch outside the timed loop
    MultiGpuResult r;
    r.per_device_s.assign(n, std::numeric_limits<double>::max());
    r.t_wall_s      = std::numeric_limits<double>::max();
    r.t_event_rep_s = std::numeric_limits<double>::max();
    Timer time;
    for (unsigned t = 0; t < kTests; ++t)
    {
        time.start();
        launch_all(true);
        time.stop();
        r.t_wall_s     = std::min(r.t_wall_s, time.elapsedSeconds());
        double rep_max = 0.0;
        for (int g = 0; g < n; ++g)
        {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0[g], e1[g]));
            r.per_device_s[g] = std::min(r.per_device_s[g], 1e-3 * (double)ms);
            rep_max           = std::max(rep_max, 1e-3 * (double)ms);
        }
        r.t_event_rep_s = std::min(r.t_event_rep_s, rep_max);
    }
    std::vector<double> ss(n, 0.0);
    for (int g = 0; g < n; ++g)
    {
        HIP_CHECK(hipSetDevice(g));
        SF_CHECK(sf_sumsq_f64(out[g], cnt[g] * nqTot, &ss[g], mg.stream(g)));
    }
    mg.reduce(r.per_device_s, ss, &r.t_max_event_s, &r.sumsq);
    for (int g = 0; g < n; ++g)
    {
        HIP_CHECK(hipSetDevice(g));
        HIP_CHECK(hipEventDestroy(e0[g]));
        HIP_CHECK(hipEventDestroy(e1[g]));
        HIP_CHECK(hipFree(in[g]));
        HIP_CHECK(hipFree(out[g]));
        HIP_CHECK(hipFree(b0[g]));
        HIP_CHECK(hipFree(b1[g]));
        HIP_CHECK(hipFree(b2[g]));
    }
    HIP_CHECK(hipSetDevice(0));
    // the same batch on device 0 alone (strong-scaling reference), same wall-clock protocol, fewer repetitions
    if (solo && n > 0)
    {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const double need = (double)sizeof(double) * (double)nelmt * (double)(nmTot + nqTot);
        if (need < 0.92 * (double)free_b)
        {
            DeviceBuffer<double> sin(nelmt * nmTot), sout(nelmt * nqTot), sb0((nq0 - 1) * nq0), sb1((nq1 - 1) * nq1),
                sb2((nq2 - 1) * nq2);
            hipStream_t s = mg.stream(0);
            if (random)
                SF_CHECK(sf_fill_random_f64(sin.get(), nelmt * nmTot, seed, 0, s));
            else
                SF_CHECK(sf_fill_sincos_f64(sin.get(), nelmt, nmTot, s));
            SF_CHECK(sf_fill_basis_f64(sb0.get(), nq0 - 1, nq0, s));
            SF_CHECK(sf_fill_basis_f64(sb1.get(), nq1 - 1, nq1, s));
            SF_CHECK(sf_fill_basis_f64(sb2.get(), nq2 - 1, nq2, s));
            auto one = [&]()
            {
                SF_CHECK(sf_bwdtrans_hex_f64_variant(variant, nq0, nq1, nq2, nelmt, sb0.get(), sb1.get(), sb2.get(),
                                                     sin.get(), nullptr, sout.get(), s));
                HIP_CHECK(hipStreamSynchronize(s));
            };
            one();
            r.t_solo_wall_s = std::numeric_limits<double>::max();
            for (unsigned t = 0; t < 10; ++t)
            {
                time.start();
                one();
                time.stop();
                r.t_solo_wall_s = std::min(r.t_solo_wall_s, time.elapsedSeconds());
            }
        }
    }
    return r;
}

} // namespace harness
