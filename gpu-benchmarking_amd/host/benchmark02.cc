// benchmark02 -- vector addition driver (the in-binary HBM bandwidth calibrator).
//
// Keeps the reference driver's contract (benchmark02/benchmark02.cc:73, 262-273):
//   ./benchmark02                     no arguments used
//   run_test<T>(size) for size = 1024 .. 536 870 912 (doubling, 20 sizes)
//   stdout: banner, then per size   Size N Case: ... / Size N norm: ... / Size N GB/s: ...
//           GB/s = 3 * 8e-9 * size / t_min (:255); norm = sqrt(sum data1^2) after the 40 timed additions
// Columns:  1 Host (OpenMP)   data1[i] += data2[i] on the host cores
//           2 HIP (vl)        sf_vector_add_f64 (16-byte lanes, one lane per thread, dispatcher-ordered grid)
// Extra options: --max-size N, --json FILE.
#include "harness.h"

#include <omp.h>

using namespace harness;

static Options g_opt;
static JsonLog g_json;
static bool g_gpu = false;

static double host_sumsq(const std::vector<double> &x)
{
    const size_t leaf = 4096, nleaf = (x.size() + leaf - 1) / leaf;
    std::vector<double> part(nleaf);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < (long long)nleaf; ++b)
    {
        const size_t lo = (size_t)b * leaf, hi = std::min(x.size(), lo + leaf);
        double s = 0.0;
        for (size_t i = lo; i < hi; ++i)
            s += x[i] * x[i];
        part[b] = s;
    }
    for (size_t m = nleaf; m > 1;)
    {
        const size_t h = m / 2;
        for (size_t b = 0; b < h; ++b)
            part[b] = part[2 * b] + part[2 * b + 1];
        if (m & 1)
            part[h] = part[m - 1];
        m = h + (m & 1);
    }
    return part[0];
}

template <typename T> void run_test(const unsigned int size)
{
    static_assert(sizeof(T) == sizeof(double), "only T = double is instantiated (as in the reference)");
    Timer time;
    const unsigned int n_tests = kTests;

    double time_host = std::numeric_limits<double>::max();
    T result_host    = 0;
    {
        std::vector<T> data1(size), data2(size);
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)size; ++i)
        {
            const unsigned int u = (unsigned int)i;
            data1[i]             = u % 13u + (0.2 + 0.00001 * (u % 100191u));
            data2[i]             = u % 8u + (0.4 + 0.00003 * (u % 100721u));
        }
        for (unsigned int t = 0; t < n_tests; ++t) // all 40: the published norm is after 40 additions
        {
            time.start();
#pragma omp parallel for schedule(static)
            for (long long i = 0; i < (long long)size; ++i)
                data1[i] += data2[i];
            time.stop();
            time_host = std::min(time_host, time.elapsedSeconds());
        }
        result_host = host_sumsq(data1);
    }

    double time_hip = std::numeric_limits<double>::max();
    T result_hip    = 0;
    if (g_gpu)
    {
        DeviceBuffer<T> d1(size), d2(size);
        SF_CHECK(sf_fill_vecadd_f64(d1.get(), d2.get(), size, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        for (unsigned int t = 0; t < n_tests; ++t)
        {
            time.start();
            SF_CHECK(sf_vector_add_f64(d1.get(), d2.get(), size, nullptr));
            HIP_CHECK(hipDeviceSynchronize());
            time.stop();
            time_hip = std::min(time_hip, time.elapsedSeconds());
        }
        SF_CHECK(sf_sumsq_f64(d1.get(), size, &result_hip, nullptr));
    }

    std::cout << std::setprecision(10);
    std::cout << "Size " << size << " Case:     Host (OpenMP)      HIP (vl)" << std::endl;
    std::cout << "Size " << size << " norm: " << std::sqrt(result_host) << " " << std::sqrt(result_hip)
              << std::endl;
    std::cout << "Size " << size << " GB/s: " << sizeof(T) * 3e-9 * size / time_host << " "
              << (g_gpu ? sizeof(T) * 3e-9 * size / time_hip : 0.0) << std::endl;
    std::ostringstream r;
    r << std::setprecision(10) << "{\"size\": " << size << ", \"host_gb_s\": "
      << sizeof(T) * 3e-9 * size / time_host << ", \"hip_gb_s\": "
      << (g_gpu ? sizeof(T) * 3e-9 * size / time_hip : 0.0) << "}";
    g_json.row(r.str());
}

int main(int argc, char **argv)
{
    g_opt = parse(argc, argv);
    g_gpu = have_gpu();
    std::cout << "--------------------------------" << std::endl;
    std::cout << "Benchmark02 : Vector Addition   " << std::endl;
    std::cout << "--------------------------------" << std::endl;
    if (!g_gpu)
        std::cerr << "benchmark02: no HIP device visible, host column only (device column printed as 0)"
                  << std::endl;
    for (unsigned int size = 1024; size < 1000000000u; size *= 2)
    {
        if (g_opt.maxsize > 0 && size > g_opt.maxsize)
            break;
        run_test<double>(size);
    }
    std::ostringstream h;
    h << device_header() << ", \"benchmark\": \"benchmark02\", \"host_threads\": " << omp_get_max_threads();
    g_json.write(g_opt.json, h.str());
    if (g_gpu)
        (void)sf_shutdown();
    return 0;
}
