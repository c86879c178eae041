// benchmark05 -- BwdTrans (3D hex) driver for MI355X.
//
// Keeps the reference driver's contract (benchmark05/benchmark05.cc:619-622, 1423-1442):
//   ./benchmark05 [nq0 nq1 nq2 threads elblocks]     defaults 8 8 8 128 1
//   run_test<T>(size, nq0, nq1, nq2, threads, elblocks) for size = 128 .. 1 048 576 (doubling)
//   stdout: banner, "BwdTrans (NQ = a, b, c)", then per size the three lines
//           nelmt N Case: ... / nelmt N norm: ... / nelmt N DOF/s: ...   (setprecision(10), 5 spaces)
// so the reference's postprocess.py parses the log unchanged.  The columns are this build's kernels,
// all behind the C ABI of libsumfact.so:
//   1 HIP (thread/elmt)     one thread per element, fused nest        (decomposition of :15-102)
//   2 HIP (block/elmt glb)  one workgroup per element, global wsp     (:431-508)
//   3 HIP (block/elmt LDS)  one workgroup per element, all in LDS     (:510-617)
//   4 HIP (wave/chunk)      flagship: one wavefront streams chunks    (sf_bwdtrans_hex_f64)
//   5 rocBLAS               1 DGEMM + 2 strided-batched DGEMMs, global wsp (cuBLAS column :1062-1171)
//   6 HIP (thread/elmt il64) one thread per element on the wave-64 interleaved layout: the `_Coa`
//                           decomposition (:104-201) without its output-index bug (:193-194)
// `threads` / `elblocks` shape the reference-style baseline columns (block size, elements per workgroup);
// the flagship kernels pick their own launch shapes.
// Extra options go AFTER the positional ones: --nelmt N, --data sincos|random, --json FILE,
// --no-baselines, --seed S, --variant auto|wave|mfma (kernel behind column 4), --precision f64|f32
// (f32 = the T = float instantiation the reference's templates allow: flagship column only),
// --ngpus N (aggregate row: the batch sharded over N devices of this node, one process, RCCL for MAX(time) /
// SUM(sum of squares), host/multigpu.h; only the flagship column is run, the others print 0).
#include "harness.h"
#include "multigpu.h"

#include <memory>
#include <type_traits>

using namespace harness;

static Options g_opt;
static JsonLog g_json;
static std::unique_ptr<MultiGpu> g_multi;

// --ngpus N: same three lines, flagship column = total DOF / best host wall time of "launch on every device + synchronise
// every device" (the single-GPU rows' clock); the HIP-event figures and the speed-up over device 0 alone on the same
// batch go to the --json side file
static void run_test_multi(const unsigned int size, const unsigned nq0, const unsigned nq1, const unsigned nq2)
{
    const size_t nelmt = size;
    const size_t nmTot = (size_t)(nq0 - 1) * (nq1 - 1) * (nq2 - 1), nqTot = (size_t)nq0 * nq1 * nq2;
    const MultiGpuResult r =
        run_hex_multi(*g_multi, nelmt, nq0, nq1, nq2, g_opt.variant, g_opt.data == "random", g_opt.seed);
    const char *names[6] = {"HIP (thread/elmt)", "HIP (block/elmt glb)", "HIP (block/elmt LDS)",
                            "HIP (wave/chunk)", "rocBLAS", "HIP (thread/elmt il64)"};
    const double dofs = 1.0e-9 * nelmt * (double)nmTot / r.t_wall_s;
    std::cout << std::setprecision(10);
    std::cout << "nelmt " << nelmt << " Case:";
    for (int v = 0; v < 6; ++v)
        std::cout << " " << names[v];
    std::cout << std::endl;
    std::cout << "nelmt " << nelmt << " norm: ";
    for (int v = 0; v < 6; ++v)
        std::cout << (v ? "     " : "") << (v == 3 ? std::sqrt(r.sumsq) : 0.0);
    std::cout << std::endl;
    std::cout << "nelmt " << nelmt << " DOF/s: ";
    for (int v = 0; v < 6; ++v)
        std::cout << (v ? "     " : "") << (v == 3 ? dofs : 0.0);
    std::cout << std::endl << std::flush;
    const double bytes = 8.0 * nelmt * (double)(nmTot + nqTot);
    std::ostringstream j;
    j << std::setprecision(10) << "{\"nelmt\": " << nelmt << ", \"nq\": [" << nq0 << "," << nq1 << "," << nq2
      << "], \"ngpus\": " << g_multi->size() << ", \"wave_gdof_s\": " << dofs
      << ", \"wave_gdof_s_host_wall\": " << dofs
      << ", \"wave_gdof_s_event\": " << 1.0e-9 * nelmt * (double)nmTot / r.t_event_rep_s
      << ", \"wave_gdof_s_event_max_of_device_minima\": " << 1.0e-9 * nelmt * (double)nmTot / r.t_max_event_s
      << ", \"wave_gb_s\": " << 1.0e-9 * bytes / r.t_wall_s
      << ", \"wave_frac_hbm_roofline_per_gpu\": " << 1.0e-9 * bytes / r.t_wall_s / kHbmPeakGBs / g_multi->size()
      << ", \"wave_frac_hbm_roofline_per_gpu_event\": " << 1.0e-9 * bytes / r.t_event_rep_s / kHbmPeakGBs / g_multi->size();
    if (r.t_solo_wall_s > 0.0)
        j << ", \"single_gpu_same_batch_gdof_s\": " << 1.0e-9 * nelmt * (double)nmTot / r.t_solo_wall_s
          << ", \"speedup_vs_1gpu_same_batch\": " << r.t_solo_wall_s / r.t_wall_s;
    j << ", \"scaling_verified_on_hardware\": " << (g_multi->size() > 1 ? "true" : "false")
      << ", \"per_device_ms\": [";
    for (size_t g = 0; g < r.per_device_s.size(); ++g)
        j << (g ? ", " : "") << 1e3 * r.per_device_s[g];
    j << "], \"norm\": " << std::sqrt(r.sumsq) << "}";
    g_json.row(j.str());
}

template <typename T>
void run_test(const unsigned int size, const unsigned int _nq0, const unsigned int _nq1,
              const unsigned int _nq2, const unsigned int _threads, const unsigned int _elblocks)
{
    constexpr bool kF32 = std::is_same<T, float>::value; // --precision f32: flagship column only
    // threads / elblocks shape the reference-style baseline columns (1-3), as in the reference
    SF_CHECK(sf_set_launch_hint(_threads, _elblocks));
    const size_t nelmt = size;
    const unsigned nq0 = _nq0, nq1 = _nq1, nq2 = _nq2;
    const unsigned nm0 = nq0 - 1u, nm1 = nq1 - 1u, nm2 = nq2 - 1u;
    const size_t nmTot = (size_t)nm0 * nm1 * nm2, nqTot = (size_t)nq0 * nq1 * nq2;

    DeviceBuffer<T> d_in(nelmt * nmTot), d_out(nelmt * nqTot);
    DeviceBuffer<T> d_basis0(nm0 * nq0), d_basis1(nm1 * nq1), d_basis2(nm2 * nq2);
    DeviceBuffer<T> d_wsp((g_opt.baselines && !kF32) ? nelmt * ((size_t)nq0 * nm1 * nm2 + (size_t)nq0 * nq1 * nm2) : 0);

    // in[e][f] = sin(f+1), basis[x] = cos(x)  (benchmark05.cc:1195-1236), generated on the device
    if constexpr (kF32)
    {
        if (g_opt.data == "random")
            SF_CHECK(sf_fill_random_f32(d_in.get(), nelmt * nmTot, g_opt.seed, 0, nullptr));
        else
            SF_CHECK(sf_fill_sincos_f32(d_in.get(), nelmt, nmTot, nullptr));
        SF_CHECK(sf_fill_basis_f32(d_basis0.get(), nm0, nq0, nullptr));
        SF_CHECK(sf_fill_basis_f32(d_basis1.get(), nm1, nq1, nullptr));
        SF_CHECK(sf_fill_basis_f32(d_basis2.get(), nm2, nq2, nullptr));
    }
    else
    {
        if (g_opt.data == "random")
            SF_CHECK(sf_fill_random_f64(d_in.get(), nelmt * nmTot, g_opt.seed, 0, nullptr));
        else
            SF_CHECK(sf_fill_sincos_f64(d_in.get(), nelmt, nmTot, nullptr));
        SF_CHECK(sf_fill_basis_f64(d_basis0.get(), nm0, nq0, nullptr));
        SF_CHECK(sf_fill_basis_f64(d_basis1.get(), nm1, nq1, nullptr));
        SF_CHECK(sf_fill_basis_f64(d_basis2.get(), nm2, nq2, nullptr));
    }
    HIP_CHECK(hipDeviceSynchronize());

    constexpr int NCOL        = 6;
    const int variants[NCOL]  = {SF_VARIANT_THREAD, SF_VARIANT_BLOCK_GLB, SF_VARIANT_BLOCK_LDS,
                                 g_opt.variant, -1 /* rocBLAS */, -2 /* interleaved layout */};
    const char *names[NCOL]   = {"HIP (thread/elmt)", "HIP (block/elmt glb)", "HIP (block/elmt LDS)",
                                 "HIP (wave/chunk)", "rocBLAS", "HIP (thread/elmt il64)"};
    // column 6 works on its own copies of in/out in the interleaved layout (as the reference keeps
    // d_in_coa next to d_in, benchmark05.cc:1240); padded to whole groups of 64 elements
    const size_t padded = (nelmt + 63) / 64 * 64;
    const bool il_on    = g_opt.baselines && !kF32;
    DeviceBuffer<double> d_in_il(il_on ? padded * nmTot : 0), d_out_il(il_on ? padded * nqTot : 0);
    DeviceBuffer<double> d_wsp_il(il_on ? padded * ((size_t)nm1 * nm2 + nm2) : 0);
    if constexpr (!kF32)
    {
        if (il_on)
        {
            SF_CHECK(sf_interleave64_f64(d_in.get(), d_in_il.get(), nelmt, nmTot, 0, nullptr));
            HIP_CHECK(hipMemsetAsync(d_out_il.get(), 0, padded * nqTot * sizeof(double), nullptr));
        }
    }
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
                SF_COLUMN(sf_bwdtrans_hex_f32(nq0, nq1, nq2, nelmt, d_basis0.get(), d_basis1.get(),
                                             d_basis2.get(), d_in.get(), d_out.get(), nullptr));
            else
            {
                if (variants[v] == -2)
                    SF_COLUMN(sf_bwdtrans_hex_f64_interleaved(nq0, nq1, nq2, nelmt, d_basis0.get(),
                                                             d_basis1.get(), d_basis2.get(),
                                                             d_in_il.get(), d_wsp_il.get(),
                                                             d_out_il.get(), nullptr));
                else if (variants[v] >= 0)
                    SF_COLUMN(sf_bwdtrans_hex_f64_variant(variants[v], nq0, nq1, nq2, nelmt,
                                                         d_basis0.get(), d_basis1.get(),
                                                         d_basis2.get(), d_in.get(), d_wsp.get(),
                                                         d_out.get(), nullptr));
#ifdef SF_WITH_ROCBLAS
                else if (variants[v] == -1)
                    blas.hex(nq0, nq1, nq2, nelmt, d_basis0.get(), d_basis1.get(), d_basis2.get(),
                             d_in.get(), d_wsp.get(), d_out.get());
#endif
            }
        };
#ifdef SF_WITH_ROCBLAS
        if (variants[v] == -1 && !blas.ok())
            continue;
#else
        if (variants[v] == -1)
            continue;
#endif
        launch(); // first touch outside the timed loop
        HIP_CHECK(hipDeviceSynchronize());
        if (col_missing) // not built for these extents: the column prints 0
            continue;
        times[v] = time_min(launch, (v == 3 || v == 4) ? 1e30 : kSlowBudgetS);
        // the same launches between HIP events (side file only; the wall clock above is the reference's protocol)
        etimes[v] = event_min(launch, (v == 3 || v == 4) ? 1e30 : kSlowBudgetS);
        if constexpr (kF32)
            SF_CHECK(sf_sumsq_f32(d_out.get(), nelmt * nqTot, &results[v], nullptr));
        else if (variants[v] == -2) // padded lanes were zeroed and are never written
            SF_CHECK(sf_sumsq_f64(d_out_il.get(), padded * nqTot, &results[v], nullptr));
        else
            SF_CHECK(sf_sumsq_f64(d_out.get(), nelmt * nqTot, &results[v], nullptr));
    }

    // Display results (grammar of benchmark05.cc:1387-1420)
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
    r << std::setprecision(10) << "{\"nelmt\": " << nelmt << ", \"nq\": [" << nq0 << "," << nq1 << ","
      << nq2 << "], \"wave_gdof_s\": " << 1.0e-9 * nelmt * (double)nmTot / times[3]
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
    unsigned int nq2      = positional(g_opt, 2, 8u);
    unsigned int threads  = positional(g_opt, 3, 128u);
    unsigned int elblocks = positional(g_opt, 4, 1u);

    std::cout << "--------------------------------" << std::endl;
    std::cout << "Benchmark05 : BwdTrans (3D)     " << std::endl;
    std::cout << "--------------------------------" << std::endl;
    std::cout << "BwdTrans (NQ = " << nq0 << ", " << nq1 << ", " << nq2 << ")" << std::endl;
    if (nq0 < 2 || nq1 < 2 || nq2 < 2)
    {
        std::cerr << "nq must be >= 2 in every direction" << std::endl;
        return 1;
    }
    if (!have_gpu())
    {
        std::cerr << "benchmark05: no HIP device visible; the kernels have no CPU fallback" << std::endl;
        return 4;
    }
    const bool f32 = (g_opt.precision == "f32");
    if (g_opt.ngpus > 1 || getenv("SF_FORCE_MULTIGPU_PATH"))
    {
        int ndev = 0;
        HIP_CHECK(hipGetDeviceCount(&ndev));
        if (ndev < g_opt.ngpus)
        {
            std::cerr << "benchmark05: --ngpus " << g_opt.ngpus << " but this node shows " << ndev
                      << " device(s); refusing to print an aggregate row for fewer GPUs" << std::endl;
            return 5;
        }
        if (f32)
        {
            std::cerr << "benchmark05: --ngpus runs the fp64 flagship only" << std::endl;
            return 1;
        }
        g_multi.reset(new MultiGpu(g_opt.ngpus));
        std::cout << "BwdTrans on " << g_opt.ngpus << " GPU(s): element ranges, RCCL MAX(time) / SUM(norm^2)" << std::endl;
        if (g_opt.nelmt > 0)
            run_test_multi((unsigned)g_opt.nelmt, nq0, nq1, nq2);
        else
            for (unsigned int size = 2 << 6; size < 2 << 20; size <<= 1)
            {
                if (g_opt.maxsize > 0 && size > g_opt.maxsize)
                    break;
                run_test_multi(size, nq0, nq1, nq2);
            }
        g_json.write(g_opt.json, device_header() + ", \"benchmark\": \"benchmark05\", \"ngpus\": " +
                                     std::to_string(g_opt.ngpus));
        g_multi.reset();
        (void)sf_shutdown();
        return 0;
    }
    if (g_opt.nelmt > 0)
    {
        if (f32)
            run_test<float>((unsigned)g_opt.nelmt, nq0, nq1, nq2, threads, elblocks);
        else
            run_test<double>((unsigned)g_opt.nelmt, nq0, nq1, nq2, threads, elblocks);
    }
    else
        for (unsigned int size = 2 << 6; size < 2 << 20; size <<= 1)
        {
            if (g_opt.maxsize > 0 && size > g_opt.maxsize)
                break;
            if (f32)
                run_test<float>(size, nq0, nq1, nq2, threads, elblocks);
            else
                run_test<double>(size, nq0, nq1, nq2, threads, elblocks);
        }
    g_json.write(g_opt.json, device_header() + ", \"benchmark\": \"benchmark05\"");
    (void)sf_shutdown();
    return 0;
}
