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
