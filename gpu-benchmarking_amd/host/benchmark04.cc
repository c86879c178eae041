// benchmark04 -- BwdTrans (2D quad) driver for MI355X.
//
// Keeps the reference driver's contract (benchmark04/benchmark04.cc:428-431, 1058-1075):
//   ./benchmark04 [nq0 nq1 threads elblocks]          defaults 8 8 128 1
//   run_test<T>(size, nq0, nq1, threads, elblocks) for size = 128 .. 1 048 576 (doubling)
//   stdout: banner, "BwdTrans (NQ = a, b)", then per size
//           nelmt N Case: ... / nelmt N norm: ... / nelmt N DOF/s: ...   (setprecision(10), 5 spaces)
// Columns (all behind the C ABI of libsumfact.so):
//   1 HIP (thread/elmt)     decomposition of BwdTransQuadKernel        (:15-76)
//   2 HIP (block/elmt glb)  BwdTransQuadKernel_QP_1D, global wsp       (:302-351)
//   3 HIP (block/elmt LDS)  BwdTransQuadKernel_QP_1D, shared           (:353-426)
//   4 HIP (wave/chunk)      flagship (sf_bwdtrans_quad_f64; matrix cores from nq = 12)
//   5 rocBLAS               1 DGEMM + 1 strided-batched DGEMM, global wsp (cuBLAS column :750-836)
// Extra options AFTER the positional ones: --nelmt N, --data sincos|random, --json FILE,
// --no-baselines, --seed S, --variant auto|wave|mfma (kernel behind column 4), --precision f64|f32
// (f32 = the T = float instantiation the reference's templates allow: flagship column only).
#include "harness.h"

#include <type_traits>

using namespace harness;

static Options g_opt;
static JsonLog g_json;

template <typename T>
void run_test(const unsigned int size, const unsigned int _nq0, const unsigned int _nq1,
              const unsigned int _threads, const unsigned int _elblocks)
{
    constexpr bool kF32 = std::is_same<T, float>::value; // --precision f32: flagship column only
    // threads / elblocks shape the reference-style baseline columns (1-3), as in the reference
    SF_CHECK(sf_set_launch_hint(_threads, _elblocks));
    const size_t nelmt = size;
    const unsigned nq0 = _nq0, nq1 = _nq1;
    const unsigned nm0 = nq0 - 1u, nm1 = nq1 - 1u;
    const size_t nmTot = (size_t)nm0 * nm1, nqTot = (size_t)nq0 * nq1;

    DeviceBuffer<T> d_in(nelmt * nmTot), d_out(nelmt * nqTot);
    DeviceBuffer<T> d_basis0(nm0 * nq0), d_basis1(nm1 * nq1);
    DeviceBuffer<T> d_wsp((g_opt.baselines && !kF32) ? nelmt * (size_t)nq0 * nm1 : 0);

    // in[e][f] = sin(f+1), basis[x] = cos(x)  (benchmark04.cc:859-889), generated on the device
    if constexpr (kF32)
    {
        if (g_opt.data == "random")
            SF_CHECK(sf_fill_random_f32(d_in.get(), nelmt * nmTot, g_opt.seed, 0, nullptr));
        else
            SF_CHECK(sf_fill_sincos_f32(d_in.get(), nelmt, nmTot, nullptr));
        SF_CHECK(sf_fill_basis_f32(d_basis0.get(), nm0, nq0, nullptr));
        SF_CHECK(sf_fill_basis_f32(d_basis1.get(), nm1, nq1, nullptr));
    }
    else
    {
        if (g_opt.data == "random")
            SF_CHECK(sf_fill_random_f64(d_in.get(), nelmt * nmTot, g_opt.seed, 0, nullptr));
        else
            SF_CHECK(sf_fill_sincos_f64(d_in.get(), nelmt, nmTot, nullptr));
        SF_CHECK(sf_fill_basis_f64(d_basis0.get(), nm0, nq0, nullptr));
        SF_CHECK(sf_fill_basis_f64(d_basis1.get(), nm1, nq1, nullptr));
    }
    HIP_CHECK(hipDeviceSynchronize());

    constexpr int NCOL       = 5;
    const int variants[NCOL] = {SF_VARIANT_THREAD, SF_VARIANT_BLOCK_GLB, SF_VARIANT_BLOCK_LDS,
                                g_opt.variant, -1 /* rocBLAS */};
    const char *names[NCOL]  = {"HIP (thread/elmt)", "HIP (block/elmt glb)", "HIP (block/elmt LDS)",
                                "HIP (wave/chunk)", "rocBLAS"};
    double times[NCOL], etimes[NCOL], results[NCOL];
#ifdef SF_WITH_ROCBLAS
    static RocblasColumn blas;
#endif
    for (int v = 0; v < NCOL; ++v)
    {
        times[v]   = std::numeric_limits<double>::max();
        etimes[v]  = std::numeric_limits<double>::max();
        results[v] = 0.0;
        if ((!g_opt.baselines || kF32) && v != 3)
            continue;
        HIP_CHECK(hipMemsetAsync(d_out.get(), 0, nelmt * nqTot * sizeof(T), nullptr));
        bool col_missing = false;
        auto launch = [&]()
        {
            if constexpr (kF32)
                SF_COLUMN(sf_bwdtrans_quad_f32(nq0, nq1, nelmt, d_basis0.get(), d_basis1.get(),
                                              d_in.get(), d_out.get(), nullptr));
            else
            {
                if (variants[v] >= 0)
                    SF_COLUMN(sf_bwdtrans_quad_f64_variant(variants[v], nq0, nq1, nelmt,
                                                          d_basis0.get(), d_basis1.get(), d_in.get(),
                                                          d_wsp.get(), d_out.get(), nullptr));
#ifdef SF_WITH_ROCBLAS
                else
                    blas.quad(nq0, nq1, nelmt, d_basis0.get(), d_basis1.get(), d_in.get(),
                              d_wsp.get(), d_out.get());
#endif
            }
        };
#ifdef SF_WITH_ROCBLAS
        if (variants[v] < 0 && !blas.ok())
            continue;
#else
        if (variants[v] < 0)
            continue;
#endif
        launch();
        HIP_CHECK(hipDeviceSynchronize());
        if (col_missing) // not built for these extents: the column prints 0
            continue;
        times[v] = time_min(launch, v >= 3 ? 1e30 : kSlowBudgetS);
        // the same launches between HIP events (side file only; the wall clock above is the reference's protocol)
        etimes[v] = event_min(launch, v >= 3 ? 1e30 : kSlowBudgetS);
        if constexpr (kF32)
            SF_CHECK(sf_sumsq_f32(d_out.get(), nelmt * nqTot, &results[v], nullptr));
        else
            SF_CHECK(sf_sumsq_f64(d_out.get(), nelmt * nqTot, &results[v], nullptr));
    }

    // Display results (grammar of benchmark04.cc:1022-1055)
    std::cout << std::setprecision(10);
    std::cout << "nelmt " << nelmt << " Case:";
    for (int v = 0; v < NCOL; ++v)
        std::cout << " " << names[v];
    std::cout << std::endl;
    std::cout << "nelmt " << nelmt << " norm: ";
    for (int v = 0; v < NCOL; ++v)
        std::cout << (v ? "     " : "") << std::sqrt(results[v]);
    std::cout << std::endl;
    std::cout << "nelmt " << nelmt << " DOF/s: ";
    for (int v = 0; v < NCOL; ++v)
    {
        const double dofs = times[v] < 1e300 ? 1.0e-9 * nelmt * (double)nmTot / times[v] : 0.0;
        std::cout << (v ? "     " : "") << dofs;
    }
    std::cout << std::endl;
    std::cout << std::flush;

    const double bytes = (double)sizeof(T) * nelmt * (double)(nmTot + nqTot);
    std::ostringstream r;
    r << std::setprecision(10) << "{\"nelmt\": " << nelmt << ", \"nq\": [" << nq0 << "," << nq1
      << "], \"wave_gdof_s\": " << 1.0e-9 * nelmt * (double)nmTot / times[3]
      << ", \"wave_gb_s\": " << 1.0e-9 * bytes / times[3]
      << ", \"wave_frac_hbm_roofline\": " << 1.0e-9 * bytes / times[3] / kHbmPeakGBs
      << ", \"wave_gdof_s_event\": " << 1.0e-9 * nelmt * (double)nmTot / etimes[3]
      << ", \"wave_frac_hbm_roofline_event\": " << 1.0e-9 * bytes / etimes[3] / kHbmPeakGBs
      << ", \"t_wall_min\": " << json_array(times, NCOL) << ", \"t_event_min\": " << json_array(etimes, NCOL)
      << ", \"gdof_s_event\": " << json_rate_array(etimes, NCOL, 1.0e-9 * nelmt * (double)nmTot)
      << ", \"norm\": " << std::sqrt(results[3]) << "}";
    g_json.row(r.str());
}

int main(int argc, char **argv)
{
    g_opt                 = parse(argc, argv);
    unsigned int nq0      = positional(g_opt, 0, 8u);
    unsigned int nq1      = positional(g_opt, 1, 8u);
    unsigned int threads  = positional(g_opt, 2, 128u);
    unsigned int elblocks = positional(g_opt, 3, 1u);

    std::cout << "--------------------------------" << std::endl;
    std::cout << "Benchmark04 : BwdTrans (2D)     " << std::endl;
    std::cout << "--------------------------------" << std::endl;
    std::cout << "BwdTrans (NQ = " << nq0 << ", " << nq1 << ")" << std::endl;
    if (nq0 < 2 || nq1 < 2)
    {
        std::cerr << "nq must be >= 2 in every direction" << std::endl;
        return 1;
    }
    if (!have_gpu())
    {
        std::cerr << "benchmark04: no HIP device visible; the kernels have no CPU fallback" << std::endl;
        return 4;
    }
    const bool f32 = (g_opt.precision == "f32");
    if (g_opt.nelmt > 0)
    {
        if (f32)
            run_test<float>((unsigned)g_opt.nelmt, nq0, nq1, threads, elblocks);
        else
            run_test<double>((unsigned)g_opt.nelmt, nq0, nq1, threads, elblocks);
    }
    else
        for (unsigned int size = 2 << 6; size < 2 << 20; size <<= 1)
        {
            if (g_opt.maxsize > 0 && size > g_opt.maxsize)
                break;
            if (f32)
                run_test<float>(size, nq0, nq1, threads, elblocks);
            else
                run_test<double>(size, nq0, nq1, threads, elblocks);
        }
    g_json.write(g_opt.json, device_header() + ", \"benchmark\": \"benchmark04\"");
    (void)sf_shutdown();
    return 0;
}
