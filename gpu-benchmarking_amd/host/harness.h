// harness.h -- what the three drivers share: HIP error checks, device buffers, the min-of-N timing
// protocol of the reference (host wall clock around launch + device sync, n_tests = 40:
// benchmark05/benchmark05.cc:632, 1319-1332) and the optional machine-readable side file.
// The stdout grammar stays the reference's (3 lines per size) so postprocess.py parses it unchanged;
// everything extra (roofline fractions, device name) goes to --json FILE.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "sumfact.h"
#include "timer.h"

#ifdef SF_WITH_ROCBLAS
#include <rocblas/rocblas.h>
#endif

#define HIP_CHECK(expr)                                                                            \
    do                                                                                             \
    {                                                                                              \
        hipError_t err_ = (expr);                                                                  \
        if (err_ != hipSuccess)                                                                    \
        {                                                                                          \
            std::cerr << "HIP error: " << hipGetErrorString(err_) << " at " << __FILE__ << ":"     \
                      << __LINE__ << " (" #expr ")" << std::endl;                                  \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

#define SF_CHECK(expr)                                                                             \
    do                                                                                             \
    {                                                                                              \
        int rc_ = (expr);                                                                          \
        if (rc_ != SF_OK)                                                                          \
        {                                                                                          \
            std::cerr << "sumfact error " << rc_ << " (" << sf_error_string(rc_) << ") at "        \
                      << __FILE__ << ":" << __LINE__ << " (" #expr ")" << std::endl;               \
            std::exit(3);                                                                          \
        }                                                                                          \
    } while (0)

// Inside a column's launch lambda: a variant that is not built for these extents (SF_ENOTBUILT, e.g. the LDS-resident
// baseline at an order whose images exceed the LDS) marks the column as missing -- it prints 0 -- instead of ending
// the run; every other error still exits.  Needs a `bool col_missing` in scope.
#define SF_COLUMN(expr)                                                                            \
    do                                                                                             \
    {                                                                                              \
        int rc_ = (expr);                                                                          \
        if (rc_ == SF_ENOTBUILT)                                                                   \
        {                                                                                          \
            col_missing = true;                                                                    \
            return;                                                                                \
        }                                                                                          \
        if (rc_ != SF_OK)                                                                          \
        {                                                                                          \
            std::cerr << "sumfact error " << rc_ << " (" << sf_error_string(rc_) << ") at "        \
                      << __FILE__ << ":" << __LINE__ << " (" #expr ")" << std::endl;               \
            std::exit(3);                                                                          \
        }                                                                                          \
    } while (0)

namespace harness
{

constexpr unsigned kTests      = 40;    // n_tests of the reference
constexpr double kHbmPeakGBs   = 8000.0; // MI355X datasheet
constexpr double kSlowBudgetS  = 3.0;   // baseline variants stop repeating after this much time

template <typename T> class DeviceBuffer
{
public:
    explicit DeviceBuffer(size_t n = 0)
    {
        resize(n);
    }
    ~DeviceBuffer()
    {
        if (m_p)
            (void)hipFree(m_p);
    }
    DeviceBuffer(const DeviceBuffer &)            = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void resize(size_t n)
    {
        if (m_p)
            HIP_CHECK(hipFree(m_p));
        m_p = nullptr;
        m_n = n;
        if (n)
            HIP_CHECK(hipMalloc((void **)&m_p, n * sizeof(T)));
    }
    T *get() const
    {
        return m_p;
    }
    size_t size() const
    {
        return m_n;
    }

private:
    T *m_p     = nullptr;
    size_t m_n = 0;
};

inline bool have_gpu()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return false;
    }
    return n > 0;
}

// Options that follow the reference's positional arguments (which stay untouched).
struct Options
{
    std::vector<unsigned> positional;
    long long nelmt   = 0;       // --nelmt N : run this single size instead of the doubling sweep
    long long maxsize = 0;       // --max-size N : stop the sweep above N
    std::string data  = "sincos"; // --data sincos|random
    std::string json;            // --json FILE
    bool baselines    = true;    // --no-baselines : only the flagship column is timed
    int variant       = SF_VARIANT_AUTO; // --variant auto|wave|mfma|... : kernel of the flagship column
    unsigned seed     = 0x5F3759DFu;
    std::string precision = "f64"; // --precision f64|f32
    int ngpus = 1;               // --ngpus N : benchmark05's aggregate row over N devices of this node (multigpu.h)
};

inline Options parse(int argc, char **argv)
{
    Options o;
    for (int a = 1; a < argc; ++a)
    {
        const std::string s = argv[a];
        auto next           = [&](const char *name) -> std::string
        {
            if (a + 1 >= argc)
            {
                std::cerr << name << " needs a value" << std::endl;
                std::exit(1);
            }
            return argv[++a];
        };
        if (s == "--nelmt")
            o.nelmt = std::atoll(next("--nelmt").c_str());
        else if (s == "--max-size")
            o.maxsize = std::atoll(next("--max-size").c_str());
        else if (s == "--data")
            o.data = next("--data");
        else if (s == "--json")
            o.json = next("--json");
        else if (s == "--seed")
            o.seed = (unsigned)std::strtoul(next("--seed").c_str(), nullptr, 0);
        else if (s == "--no-baselines")
            o.baselines = false;
        else if (s == "--precision")
            o.precision = next("--precision");
        else if (s == "--ngpus")
        {
            o.ngpus = std::atoi(next("--ngpus").c_str());
            if (o.ngpus < 1 || o.ngpus > 64)
            {
                std::cerr << "--ngpus needs a device count between 1 and 64" << std::endl;
                std::exit(1);
            }
        }
        else if (s == "--variant")
        {
            const std::string v = next("--variant");
            o.variant           = -1;
            for (int k = 0; k < SF_NUM_VARIANTS; ++k)
                if (v == sf_variant_name(k))
                    o.variant = k;
            if (o.variant < 0)
            {
                std::cerr << "unknown --variant " << v << std::endl;
                std::exit(1);
            }
        }
        else if (s.rfind("--", 0) == 0)
            ; // unknown --flags are tolerated (the reference let Kokkos::initialize eat them)
        else
            o.positional.push_back((unsigned)std::atoi(s.c_str()));
    }
    return o;
}

inline unsigned positional(const Options &o, size_t idx, unsigned dflt)
{
    return idx < o.positional.size() ? o.positional[idx] : dflt;
}

// min over n_tests of the host wall time of fn() + device synchronisation
template <class F> double time_min(F &&fn, double budget_s = 1e30)
{
    Timer time;
    double best = std::numeric_limits<double>::max(), spent = 0.0;
    for (unsigned t = 0; t < kTests; ++t)
    {
        time.start();
        fn();
        HIP_CHECK(hipDeviceSynchronize());
        time.stop();
        const double el = time.elapsedSeconds();
        best            = std::min(best, el);
        spent += el;
        if (spent > budget_s && t >= 2)
            break;
    }
    return best;
}

// min over n_tests of the DEVICE time of fn() between two HIP events on the stream fn() launches on (a second loop,
// after time_min's: the reference's wall-clock protocol stays undisturbed).  What the kernels themselves take, without
// the ~10 us launch + synchronise floor that dominates the wall clock at the low orders; goes to the --json side file.
template <class F> double event_min(F &&fn, double budget_s = 1e30, hipStream_t stream = nullptr)
{
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    double best = std::numeric_limits<double>::max(), spent = 0.0;
    for (unsigned t = 0; t < kTests; ++t)
    {
        HIP_CHECK(hipEventRecord(e0, stream));
        fn();
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, 1e-3 * (double)ms);
        spent += 1e-3 * (double)ms;
        if (spent > budget_s && t >= 2)
            break;
    }
    HIP_CHECK(hipEventDestroy(e0));
    HIP_CHECK(hipEventDestroy(e1));
    return best;
}

// "[a, b, c]" for the side file; columns that were not run carry 0
inline std::string json_array(const double *v, int n, double scale = 1.0)
{
    std::ostringstream o;
    o << std::setprecision(10) << "[";
    for (int i = 0; i < n; ++i)
        o << (i ? ", " : "") << ((v[i] < 1e300 && v[i] > 0.0) ? scale * v[i] : 0.0);
    o << "]";
    return o.str();
}
inline std::string json_rate_array(const double *t, int n, double work)
{
    std::ostringstream o;
    o << std::setprecision(10) << "[";
    for (int i = 0; i < n; ++i)
        o << (i ? ", " : "") << ((t[i] < 1e300 && t[i] > 0.0) ? work / t[i] : 0.0);
    o << "]";
    return o.str();
}

struct JsonLog
{
    std::ostringstream body;
    bool first = true;
    void row(const std::string &txt)
    {
        body << (first ? "" : ",\n") << "  " << txt;
        first = false;
    }
    void write(const std::string &path, const std::string &header)
    {
        if (path.empty())
            return;
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f)
        {
            std::cerr << "cannot write " << path << std::endl;
            return;
        }
        std::fprintf(f, "{%s,\n \"rows\": [\n%s\n ]}\n", header.c_str(), body.str().c_str());
        std::fclose(f);
    }
};

#ifdef SF_WITH_ROCBLAS
// Vendor-library column, counterpart of the reference's cuBLAS column
// (benchmark05/benchmark05.cc:1062-1171, benchmark04/benchmark04.cc:750-836): the sweeps as library
// GEMMs with the intermediates in a global workspace.  Formulated for THIS build's layouts
// (in[e][r][q][p], out[e][k][j][i]); column-major BLAS view, bases row-major nm x nq = col-major nq x nm:
//   dir 0  W1[nq0 x (nm1 nm2 nelmt)]      = B0cm (nq0 x nm0) * IN (nm0 x nm1 nm2 nelmt)      one DGEMM
//   dir 1  W2_(e,r)[nq0 x nq1]            = W1_(e,r) (nq0 x nm1) * B1cm^T                    batched over (e,r)
//   dir 2  OUT_e[(nq0 nq1) x nq2]         = W2_e ((nq0 nq1) x nm2) * B2cm^T                  batched over e
struct RocblasColumn
{
    rocblas_handle handle = nullptr;
    RocblasColumn()
    {
        if (rocblas_create_handle(&handle) != rocblas_status_success)
            handle = nullptr;
    }
    ~RocblasColumn()
    {
        if (handle)
            rocblas_destroy_handle(handle);
    }
    bool ok() const
    {
        return handle != nullptr;
    }
    static void check(rocblas_status st, const char *what)
    {
        if (st != rocblas_status_success)
        {
            std::cerr << "rocBLAS error " << (int)st << " in " << what << std::endl;
            std::exit(5);
        }
    }
    void hex(unsigned nq0, unsigned nq1, unsigned nq2, size_t nelmt, const double *b0,
             const double *b1, const double *b2, const double *in, double *wsp, double *out)
    {
        const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1, nm2 = nq2 - 1;
        const double one = 1.0, zero = 0.0;
        double *w1 = wsp, *w2 = wsp + nelmt * (size_t)nq0 * nm1 * nm2;
        check(rocblas_dgemm(handle, rocblas_operation_none, rocblas_operation_none, nq0,
                            (rocblas_int)(nelmt * nm1 * nm2), nm0, &one, b0, nq0, in, nm0, &zero, w1,
                            nq0),
              "dgemm dir0");
        check(rocblas_dgemm_strided_batched(handle, rocblas_operation_none, rocblas_operation_transpose,
                                            nq0, nq1, nm1, &one, w1, nq0, (rocblas_stride)nq0 * nm1, b1,
                                            nq1, 0, &zero, w2, nq0, (rocblas_stride)nq0 * nq1,
                                            (rocblas_int)(nelmt * nm2)),
              "dgemm_strided_batched dir1");
        check(rocblas_dgemm_strided_batched(handle, rocblas_operation_none, rocblas_operation_transpose,
                                            nq0 * nq1, nq2, nm2, &one, w2, nq0 * nq1,
                                            (rocblas_stride)nq0 * nq1 * nm2, b2, nq2, 0, &zero, out,
                                            nq0 * nq1, (rocblas_stride)nq0 * nq1 * nq2,
                                            (rocblas_int)nelmt),
              "dgemm_strided_batched dir2");
    }
    void quad(unsigned nq0, unsigned nq1, size_t nelmt, const double *b0, const double *b1,
              const double *in, double *wsp, double *out)
    {
        const unsigned nm0 = nq0 - 1, nm1 = nq1 - 1;
        const double one = 1.0, zero = 0.0;
        check(rocblas_dgemm(handle, rocblas_operation_none, rocblas_operation_none, nq0,
                            (rocblas_int)(nelmt * nm1), nm0, &one, b0, nq0, in, nm0, &zero, wsp, nq0),
              "dgemm dir0");
        check(rocblas_dgemm_strided_batched(handle, rocblas_operation_none, rocblas_operation_transpose,
                                            nq0, nq1, nm1, &one, wsp, nq0, (rocblas_stride)nq0 * nm1, b1,
                                            nq1, 0, &zero, out, nq0, (rocblas_stride)nq0 * nq1,
                                            (rocblas_int)nelmt),
              "dgemm_strided_batched dir1");
    }
};
#endif

inline std::string device_header()
{
    int cu = 0, wave = 0;
    char name[256] = "none";
    if (have_gpu())
        (void)sf_device_info(&cu, &wave, name, sizeof name);
    std::ostringstream h;
    h << "\"device\": \"" << name << "\", \"num_cu\": " << cu << ", \"hbm_peak_gb_s\": " << kHbmPeakGBs
      << ", \"n_tests\": " << kTests;
    return h.str();
}

} // namespace harness
