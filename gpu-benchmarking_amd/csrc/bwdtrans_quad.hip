// bwdtrans_quad.hip -- compile-time instantiations of the 2D quad wave kernel + nq dispatch.
// (EC, WPB, BMODE, MINW) per nq: tuned configuration, see tools/sf_tune and DESIGN.md.
#include "sf_dispatch.h"
#include "wave_launch.h"
#include "wave_table.h"

#include <cstdlib>

namespace sf
{

// small batches: see bwdtrans_hex.hip (HexSmall)
template <int NQ> struct QuadSmall
{
    static constexpr int EC = (QuadCfg<NQ>::EC / 4 + 1) / 2 * 2 < 2 ? 2 : (QuadCfg<NQ>::EC / 4 + 1) / 2 * 2;
};

// nq = 2 (one mode per element): out[e][j][i] = (in[e] * B0[i]) * B1[j]: a pure stream, two lanes per element
// (see hex_nq2_stream_kernel in bwdtrans_hex.hip)
__global__ __launch_bounds__(256) void quad_nq2_stream_kernel(const double *__restrict__ b0,
                                                              const double *__restrict__ b1,
                                                              const double *__restrict__ in,
                                                              double *__restrict__ out, uint64_t nelmt)
{
    constexpr int U   = 1; // one output pair per thread: 0.76-0.77 of the roofline against 0.72 with four (tools/sf_membench11)
    const uint64_t nv = nelmt * 2;
    const double c0 = b0[0], c1 = b0[1];
    double2_t *out2 = reinterpret_cast<double2_t *>(out);
    const uint64_t base = logical_block<64>() * (256ull * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        const uint64_t v = base + (uint64_t)u * 256;
        if (v < nv)
        {
            const double x  = in[v >> 1];
            const double bj = b1[v & 1];
            const double2_t r = {(x * c0) * bj, (x * c1) * bj};
            __builtin_nontemporal_store(r, out2 + v);
        }
    }
}

static int launch_quad_nq2(const QuadArgs &a, hipStream_t s)
{
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t blocks = (a.nelmt * 2 + 255) / 256;
    if (blocks > 0x7fffffffull)
        return SF_EINVAL;
    quad_nq2_stream_kernel<<<(unsigned)blocks, 256, 0, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ> static int go(const QuadArgs &a, hipStream_t s)
{
    if constexpr (NQ == 2)
        return launch_quad_nq2(a, s);
    using C = QuadCfg<NQ>;
    constexpr uint64_t per_block = (uint64_t)C::EC * C::WPB * (C::KM > 0 ? C::KM : 1);
    if (a.nelmt < 2 * per_block * (uint64_t)device_info().num_cu)
        return launch_quad_wave<NQ, QuadSmall<NQ>::EC, 1, C::BM, C::MW, 1, C::OUT>(a, s);
    return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF>(a, s);
}

// nq = 2, T = float: one 16-byte vector per thread = the four outputs of one element
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void quad_nq2_stream_f32_kernel(const float *__restrict__ b0, const float *__restrict__ b1,
                                                                  const float *__restrict__ in, float *__restrict__ out,
                                                                  uint64_t nelmt)
{
    const uint64_t v = logical_block<64>() * 256ull + threadIdx.x;
    if (v < nelmt)
    {
        const float x  = in[v];
        const float x0 = x * b0[0], x1 = x * b0[1];
        const float4_t r = {x0 * b1[0], x1 * b1[0], x0 * b1[1], x1 * b1[1]};
        __builtin_nontemporal_store(r, reinterpret_cast<float4_t *>(out) + v);
    }
}

// fp32 orders from here on run the matrix-core kernel: its 16-wide tiles are (nearly) full there, and the vector kernel
// is issue-bound (0.41-0.52 of the roofline at nq 25..32, profiles/r02/sweep_auto_f32.log)
constexpr int kQuadF32MfmaFrom = 25; // below, the vector kernel is ahead at every configuration (0.55-0.73 against 0.35-0.55)

template <int NQ> static int go_f32(const QuadArgsT<float> &a, hipStream_t s)
{
    if constexpr (NQ == 2)
    {
        if (a.nelmt == 0)
            return SF_OK;
        const uint64_t blocks = (a.nelmt + 255) / 256;
        if (blocks > 0x7fffffffull)
            return SF_EINVAL;
        quad_nq2_stream_f32_kernel<<<(unsigned)blocks, 256, 0, s>>>(a.b0, a.b1, a.in, a.out, a.nelmt);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SF_OK : (int)e;
    }
    if constexpr (NQ >= kQuadF32MfmaFrom)
    {
        // v_mfma_f32_16x16x4_f32 (bwdtrans_mfma.h, T = float): two-element chunks, LDS-staged line-aligned output where the
        // rows are not whole lines, two waves per SIMD, XCD runs
        // per order the best of tools/experiments/f32_mfma_cfg.sh over two boxes (profiles/r03/f32_mfma_configurations*.log;
        // fraction of the fp32 HBM roofline at 1 Mi elements, vector kernel in brackets): 25 0.54-0.57 (0.51)
        // 26 0.53-0.54 (0.51)  27 0.56-0.57 (0.50)  28 0.57 (0.52)  29 0.60-0.61 (0.52)  30 0.57-0.58 (0.52)
        // 31 0.61-0.62 (0.51)  32 0.64 (0.42); at nq 17..24 the vector kernel stays ahead (second log)
        constexpr int best = NQ == 25 ? 3 : ((NQ == 28 || NQ == 32) ? 2 : ((NQ == 26 || NQ == 30 || NQ == 31) ? 4 : (NQ == 27 ? 0 : 1)));
        static const int cfg = getenv("SF_F32_MFMA_CFG") ? atoi(getenv("SF_F32_MFMA_CFG")) : best; // development knob
        switch (cfg)
        {
        case 1: return launch_quad_mfma<NQ, 2, 4, 4, 2, (NQ != 32), 64, float>(a, s);
        case 2: return launch_quad_mfma<NQ, 4, 4, 2, 1, (NQ != 32), 64, float>(a, s);
        case 3: return launch_quad_mfma<NQ, 4, 4, 4, 1, (NQ != 32), 64, float>(a, s);
        case 4: return launch_quad_mfma<NQ, 2, 4, 4, 1, (NQ != 32), 64, float>(a, s);
        case 5:
        {
            using C = QuadCfgF32<NQ>;
            return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF, float>(a, s);
        }
        default: return launch_quad_mfma<NQ, 2, 4, 2, 2, (NQ != 32), 64, float>(a, s);
        }
    }
    else
    {
        using C = QuadCfgF32<NQ>;
        return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF, float>(a, s);
    }
}

// fp32 (T = float): the vector-ALU kernel up to nq = kQuadF32MfmaFrom - 1, the fp32 matrix-core kernel above
int launch_quad_wave_f32_nq(unsigned nq, const QuadArgsT<float> &a, hipStream_t s)
{
    switch (nq)
    {
#define SF_CASE(N) case N: return go_f32<N>(a, s);
        SF_CASE(2) SF_CASE(3) SF_CASE(4) SF_CASE(5) SF_CASE(6) SF_CASE(7) SF_CASE(8) SF_CASE(9)
        SF_CASE(10) SF_CASE(11) SF_CASE(12) SF_CASE(13) SF_CASE(14) SF_CASE(15) SF_CASE(16)
        SF_CASE(17) SF_CASE(18) SF_CASE(19) SF_CASE(20) SF_CASE(21) SF_CASE(22) SF_CASE(23) SF_CASE(24)
        SF_CASE(25) SF_CASE(26) SF_CASE(27) SF_CASE(28) SF_CASE(29) SF_CASE(30) SF_CASE(31)
        SF_CASE(32)
#undef SF_CASE
    default: return SF_ENOTBUILT;
    }
}

// matrix-core kernel (bwdtrans_mfma.h): every order 11..32; chunks of 2 elements
// Measured per order (profiles/r01/tune_table_quadmfma.log): two waves per SIMD above nq = 16 (the
// compiler otherwise spends > 256 registers and halves the occupancy); the chunk leaves through LDS as a
// flat line-aligned stream where direct tile stores would write fragments (rows that are not multiples of
// 128 B), and straight from the accumulators where rows are whole lines (nq = 16, 24, 32) or short.
constexpr bool quad_mfma_lds_out(int nq)
{
    return nq == 13 || nq == 14 || nq == 15 || nq == 17 || (nq >= 20 && nq <= 23) || (nq >= 25 && nq <= 31);
}
template <int NQ> static int go_mfma(const QuadArgs &a, hipStream_t s)
{
    constexpr int EC = (NQ >= 13 && NQ <= 15) ? 4 : 2; // +3 % at nq 13..15 (tune_quad1[3-5]_scol2.log)
    // nq 25..31: two waves per workgroup, one chunk per wave (+3-7 %, profiles/r01/tune_quad2[6-8]_mfma3.log)
    constexpr bool mid = NQ >= 25 && NQ <= 31;
    // runs of 64 neighbouring workgroups per XCD: +1-3.5 % where the kernel is not compute-bound
    return launch_quad_mfma<NQ, EC, (mid ? 2 : 4), (NQ <= 16 ? 1 : 2), ((NQ <= 16 || mid) ? 1 : 2),
                            quad_mfma_lds_out(NQ), 64>(a, s);
}

int launch_quad_mfma_nq(unsigned nq, const QuadArgs &a, hipStream_t s)
{
    switch (nq)
    {
#define SF_CASE(N) case N: return go_mfma<N>(a, s);
        SF_CASE(11) SF_CASE(12) SF_CASE(13) SF_CASE(14) SF_CASE(15) SF_CASE(16) SF_CASE(17)
        SF_CASE(18) SF_CASE(19) SF_CASE(20) SF_CASE(21) SF_CASE(22) SF_CASE(23) SF_CASE(24)
        SF_CASE(25) SF_CASE(26) SF_CASE(27) SF_CASE(28) SF_CASE(29) SF_CASE(30) SF_CASE(31)
        SF_CASE(32)
#undef SF_CASE
    default: return SF_ENOTBUILT;
    }
}

// 4x4x4_4b matrix-core kernel (bwdtrans_mfma4.h), every order 8..32.  Configuration per order from
// profiles/r02/tune_mfma4_<nq>.log (1 Mi elements, mean of 10 launches, fraction of the 8 TB/s HBM roofline; in brackets
// the better of the wave kernel and the 16x16x4 kernel in the same run):
//   21 0.718 (0.698)  22 0.694 (0.635)  23 0.697 (0.615)  24 0.707 (0.583)  25 0.655 (0.584)  26 0.647 (0.567)
//   27 0.660 (0.577)  28 0.669 (0.605)  29 0.641 (0.578)  30 0.625 (0.587)  31 0.619 (0.613)  32 0.654 (0.702)
// Up to nq = 24 short-lived workgroups (one or two chunks per wave) are ahead -- the hardware dispatcher balances them
// and three waves per SIMD fit the LDS; from 25 only two fit, a one-chunk wave can no longer hide its matrix phase
// (halving the products lifts it from 0.56 to 0.77 at nq = 28), and a persistent grid with register
// prefetch wins -- fed in batches from a device-wide counter, +2-5 % over fixed shares
// (profiles/r02/tune_mfma4_28_dynamic_batches.log).
template <int NQ> static int go_mfma4(const QuadArgs &a, hipStream_t s)
{
    if constexpr (NQ <= 22 || NQ == 24)
        return launch_quad_mfma4<NQ, 2, 4, 2, 4, 1, 64>(a, s); // two elements x two i tiles, one chunk per wave
    else if constexpr (NQ == 23)
        return launch_quad_mfma4<NQ, 2, 4, 2, 4, 2, 64>(a, s);
    else if constexpr (NQ <= 27)
        // four elements per instruction, persistent grid taking batches of 8 chunks from a device-wide counter
        // (a fixed share per wave is 2-5 % slower: the launch then waits for its slowest wave)
        return launch_quad_mfma4<NQ, 4, 4, 1, 4, 0, 0, false, 8>(a, s);
    else
        return launch_quad_mfma4<NQ, 2, 4, 2, 4, 0, 0, false, 4>(a, s); // 28 .. 32: two elements, batches of 4
}

int launch_quad_mfma4_nq(unsigned nq, const QuadArgs &a, hipStream_t s)
{
    switch (nq)
    {
#define SF_CASE(N) case N: return go_mfma4<N>(a, s);
        SF_CASE(8) SF_CASE(9) SF_CASE(10) SF_CASE(11) SF_CASE(12) SF_CASE(13) SF_CASE(14) SF_CASE(15) SF_CASE(16)
        SF_CASE(17) SF_CASE(18) SF_CASE(19) SF_CASE(20) SF_CASE(21) SF_CASE(22) SF_CASE(23) SF_CASE(24)
        SF_CASE(25) SF_CASE(26) SF_CASE(27) SF_CASE(28) SF_CASE(29) SF_CASE(30) SF_CASE(31) SF_CASE(32)
#undef SF_CASE
    default: return SF_ENOTBUILT;
    }
}

// What SF_VARIANT_AUTO runs in 2D: wave kernel up to nq = 20, the 4x4x4_4b matrix-core kernel for 21..31, the 16x16x4
// matrix-core kernel at 32 (exact 16-wide tiles there), wave kernel again for whatever else is in its table.
int quad_auto_kernel(unsigned nq)
{
    return nq >= 21 && nq <= 31 ? SF_VARIANT_MFMA4 : (nq == 32 ? SF_VARIANT_MFMA : SF_VARIANT_WAVE);
}

// Orders for which the 16x16x4 kernel is ahead of the wave kernel (kept for the record: AUTO no longer asks): nq >= 25.  On MI355X the fp64 matrix and vector
// pipes have the same peak, so the exact-size FMAs of the wave kernel win wherever their operands can be fed; with
// scalar-register basis blocks, four-element chunks and XCD runs that is every order up to 24 (349 / 355 / 358 / 359 /
// 366 GDOF/s at nq = 12 .. 16 against 344 / 341 / 341 / 336 / 347 on the matrix cores,
// profiles/r01/tune_quad1[2-6]_xcd_runs.log; 290-351 against 207-282 at nq = 17 .. 24).  From nq = 25 one pencil pass
// per wave can no longer hide the scalar-load latency and the (padded) 16x16x4 tiles are ahead.
bool quad_prefers_mfma(unsigned nq)
{
    return nq >= 25;
}

int launch_quad_wave_nq(unsigned nq, const QuadArgs &a, hipStream_t s)
{
    switch (nq)
    {
#define SF_CASE(N) case N: return go<N>(a, s);
        SF_CASE(2) SF_CASE(3) SF_CASE(4) SF_CASE(5) SF_CASE(6) SF_CASE(7) SF_CASE(8) SF_CASE(9)
        SF_CASE(10) SF_CASE(11) SF_CASE(12) SF_CASE(13) SF_CASE(14) SF_CASE(15) SF_CASE(16)
        SF_CASE(17) SF_CASE(18) SF_CASE(19) SF_CASE(20) SF_CASE(21) SF_CASE(22) SF_CASE(23) SF_CASE(24)
        SF_CASE(32)
#undef SF_CASE
    default: return SF_ENOTBUILT;
    }
}

} // namespace sf
