// benchmark03 -- dense matrix-vector product driver.
//
// Keeps the reference driver's contract (benchmark03/benchmark03.cc:106-110, 338-350):
//   ./benchmark03                     no arguments used
//   run_test<T>(size) for size = 128 .. 16384 (doubling, 8 sizes), M = N = size
//   stdout: banner, then per size   Size N Case: ... / Size N norm: ... / Size N GB/s: ...
//           GB/s = 8e-9 * M * N / t_min (:332), 5-space separators (:326-336); norm = sqrt(sum y^2)
// data: A[i*N+j] = sin(i*N+j+1), x[j] = j (:160-167).
// Columns:  1 Host (OpenMP)   row dot products on the host cores
//           2 HIP (vl)        sf_matvec_f64 (one wavefront per row, 16-byte lanes, shuffle tree)
// Extra options: --max-size N, --json FILE.
#include "harness.h"

#include <omp.h>

using namespace harness;

static Options g_opt;
static JsonLog g_json;
static bool g_gpu = false;

template <typename T> void run_test(const unsigned int size)
{
    static_assert(sizeof(T) == sizeof(double), "only T = double is instantiated (as in the reference)");
    Timer time;
    const unsigned int M = size, N = size;
    const unsigned int n_tests = kTests;

    double time_host = std::numeric_limits<double>::max();
    T result_host    = 0;
    {
        std::vector<T> h_A((size_t)M * N), h_x(N), h_y(M);
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)M; ++i)
            for (unsigned int j = 0; j < N; ++j)
                h_A[(size_t)i * N + j] = std::sin((T)((size_t)i * N + j + 1));
        for (unsigned int j = 0; j < N; ++j)
            h_x[j] = j;
        double spent = 0.0;
        for (unsigned int t = 0; t < n_tests; ++t)
        {
            time.start();
#pragma omp parallel for schedule(static)
            for (long long i = 0; i < (long long)M; ++i)
            {
                const T *a = &h_A[(size_t)i * N];
                T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                unsigned int j = 0;
                for (; j + 4 <= N; j += 4)
                {
                    s0 += a[j] * h_x[j];
                    s1 += a[j + 1] * h_x[j + 1];
                    s2 += a[j + 2] * h_x[j + 2];
                    s3 += a[j + 3] * h_x[j + 3];
                }
                for (; j < N; ++j)
                    s0 += a[j] * h_x[j];
                h_y[i] = (s0 + s1) + (s2 + s3);
            }
            time.stop();
            time_host = std::min(time_host, time.elapsedSeconds());
            spent += time.elapsedSeconds();
            if (spent > 4.0 * kSlowBudgetS && t >= 2)
                break;
        }
        for (unsigned int i = 0; i < M; ++i)
            result_host += h_y[i] * h_y[i];
    }

    double time_hip = std::numeric_limits<double>::max();
    T result_hip    = 0;
    if (g_gpu)
    {
        DeviceBuffer<T> d_A((size_t)M * N), d_x(N), d_y(M);
        SF_CHECK(sf_fill_matvec_f64(d_A.get(), d_x.get(), M, N, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        for (unsigned int t = 0; t < n_tests; ++t)
        {
            time.start();
            SF_CHECK(sf_matvec_f64(M, N, d_A.get(), d_x.get(), d_y.get(), nullptr));
            HIP_CHECK(hipDeviceSynchronize());
            time.stop();
            time_hip = std::min(time_hip, time.elapsedSeconds());
        }
        SF_CHECK(sf_sumsq_f64(d_y.get(), M, &result_hip, nullptr));
    }

    std::cout << std::setprecision(10);
    std::cout << "Size " << size << " Case:     Host (OpenMP)      HIP (vl)" << std::endl;
    std::cout << "Size " << size << " norm: " << std::sqrt(result_host) << "     "
              << std::sqrt(result_hip) << std::endl;
    std::cout << "Size " << size << " GB/s: " << sizeof(T) * 1.0e-9 * M * N / time_host << "     "
              << (g_gpu ? sizeof(T) * 1.0e-9 * M * N / time_hip : 0.0) << std::endl;
    std::ostringstream r;
    r << std::setprecision(10) << "{\"size\": " << size << ", \"host_gb_s\": "
      << sizeof(T) * 1.0e-9 * M * N / time_host << ", \"hip_gb_s\": "
      << (g_gpu ? sizeof(T) * 1.0e-9 * M * N / time_hip : 0.0) << "}";
    g_json.row(r.str());
}

int main(int argc, char **argv)
{
    g_opt = parse(argc, argv);
    g_gpu = have_gpu();
    std::cout << "--------------------------------" << std::endl;
    std::cout << "Benchmark03 : Matrix-Vector Mult" << std::endl;
    std::cout << "--------------------------------" << std::endl;
    if (!g_gpu)
        std::cerr << "benchmark03: no HIP device visible, host column only (device column printed as 0)"
                  << std::endl;
    for (unsigned int size = 2 << 6; size < 2 << 14; size *= 2)
    {
        if (g_opt.maxsize > 0 && size > g_opt.maxsize)
            break;
        run_test<double>(size);
    }
    std::ostringstream h;
    h << device_header() << ", \"benchmark\": \"benchmark03\", \"host_threads\": " << omp_get_max_threads();
    g_json.write(g_opt.json, h.str());
    if (g_gpu)
        (void)sf_shutdown();
    return 0;
}
