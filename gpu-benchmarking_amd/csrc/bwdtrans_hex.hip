// bwdtrans_hex.hip -- compile-time instantiations of the 3D hex wave kernel + nq dispatch.
// One row per isotropic nq; the tuple (EC, WPB, BMODE, MINW) is the tuned configuration
// (tools/sf_tune prints the sweep these were picked from; DESIGN.md records the numbers).
#include "sf_dispatch.h"
#include "wave_launch.h"
#include "wave_table.h"

#include <cstdlib>

namespace sf
{

// Small batches: with the tuned (fat) chunks a batch of a few thousand elements occupies only a fraction
// of the 256 CUs, so below ~2 workgroups per CU a fine-grained instantiation is launched instead:
// quarter-size chunks (kept even: 16-byte alignment of the chunk), one wave per workgroup, one chunk per wave.
template <int NQ> struct HexSmall
{
    static constexpr int EC  = HexCfg<NQ>::EC == 1 ? 1
                               : ((HexCfg<NQ>::EC / 4 + 1) / 2 * 2 < 2 ? 2 : (HexCfg<NQ>::EC / 4 + 1) / 2 * 2);
    static constexpr int OUT = HexCfg<NQ>::OUT;
};

// nq = 2 (one mode per element): out[e][k][j][i] = ((in[e] * B0[i]) * B1[j]) * B2[k] -- no sweep has anything to
// sum, so the element is a pure stream: one 16-byte output pair (i = 0, 1) per lane, four lanes per element, the
// input value re-read through the cache.  Same multiplication order as the sweeps, hence bit-identical results.
__global__ __launch_bounds__(256) void hex_nq2_stream_kernel(const double *__restrict__ b0,
                                                             const double *__restrict__ b1,
                                                             const double *__restrict__ b2,
                                                             const double *__restrict__ in,
                                                             double *__restrict__ out, uint64_t nelmt)
{
    // ONE output pair per thread: under bench.py's sweep protocol 0.81-0.82 of the roofline against 0.75-0.77 with four pairs
    // per thread and 0.72 with sixteen (tools/sf_membench11, profiles/r02/membench11_nq2_stream_shapes.log) -- the shorter a
    // wave lives between its first and last store, the better this 89 %-writes stream runs
    constexpr int U   = 1; // output pairs per thread
    const uint64_t nv = nelmt * 4;
    const double c0 = b0[0], c1 = b0[1];
    double2_t *out2 = reinterpret_cast<double2_t *>(out);
    const uint64_t base = logical_block<64>() * (256ull * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        const uint64_t v = base + (uint64_t)u * 256;
        if (v < nv)
        {
            const int p    = (int)(v & 3); // pair index inside the element: j = p & 1, k = p >> 1
            const double x = in[v >> 2];
            const double bj = b1[p & 1], bk = b2[p >> 1];
            const double2_t r = {((x * c0) * bj) * bk, ((x * c1) * bj) * bk};
            __builtin_nontemporal_store(r, out2 + v);
        }
    }
}

static int launch_hex_nq2(const HexArgs &a, hipStream_t s)
{
    if (a.nelmt == 0)
        return SF_OK;
    const uint64_t blocks = (a.nelmt * 4 + 255) / 256;
    if (blocks > 0x7fffffffull)
        return SF_EINVAL;
    hex_nq2_stream_kernel<<<(unsigned)blocks, 256, 0, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SF_OK : (int)e;
}

template <int NQ> static int go(const HexArgs &a, hipStream_t s)
{
    if constexpr (NQ == 2)
        return launch_hex_nq2(a, s);
    using C = HexCfg<NQ>;
    constexpr uint64_t per_block = (uint64_t)C::EC * C::WPB * (C::KM > 0 ? C::KM : 1);
    if (a.nelmt < 2 * per_block * (uint64_t)device_info().num_cu)
        return launch_hex_wave<NQ, HexSmall<NQ>::EC, 1, C::BM, C::MW, 1, HexSmall<NQ>::OUT>(a, s);
    // Very large batches (BASELINE configs[4]: 10 M elements = 68 GB): elements are independent, so the batch is
    // enqueued as back-to-back launches of hex_piece(NQ) elements each.  Over a launch of many milliseconds the eight
    // XCDs drift apart and the DRAM access front widens; a launch boundary every ~0.6 ms re-aligns them: +2 % at
    // nq = 7 / 8 from 2.5 M elements up, nothing at 1 Mi, nothing at nq 9 / 10, a loss at the low orders whose
    // pieces would be short launches (profiles/r02/large_batch_split.log) -- hence per order.
    constexpr uint64_t piece = hex_piece(NQ);
    if (piece > 0 && a.nelmt > 2 * piece)
    {
        static_assert(piece == 0 || piece % (8 * 64 * per_block) == 0, "pieces hold whole XCD windows");
        constexpr uint64_t NMT = (uint64_t)(NQ - 1) * (NQ - 1) * (NQ - 1), NQT = (uint64_t)NQ * NQ * NQ;
        for (uint64_t lo = 0; lo < a.nelmt; lo += piece)
        {
            HexArgs part = a;
            part.in      = a.in + lo * NMT;
            part.out     = a.out + lo * NQT;
            part.nelmt   = a.nelmt - lo < piece ? a.nelmt - lo : piece;
            const int rc = launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF>(part, s);
            if (rc != SF_OK)
                return rc;
        }
        return SF_OK;
    }
    return launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF>(a, s);
}

// returns SF_ENOTBUILT when nq has no instantiation
int launch_hex_wave_nq(unsigned nq, const HexArgs &a, hipStream_t s)
{
    switch (nq)
    {
    case 2: return go<2>(a, s);
    case 3: return go<3>(a, s);
    case 4: return go<4>(a, s);
    case 5: return go<5>(a, s);
    case 6: return go<6>(a, s);
    case 7: return go<7>(a, s);
    case 8: return go<8>(a, s);
    case 9: return go<9>(a, s);
    case 10: return go<10>(a, s);
    case 11: return go<11>(a, s);
    default: return SF_ENOTBUILT;
    }
}

// matrix-core kernel (bwdtrans_mfma.h, hex_mfma_kernel): orders 4..16.
// SF_VARIANT_AUTO uses it above the wave kernel's table (nq 11..16); below, the wave kernel is faster.
template <int NQ> static int go_mfma(const HexArgs &a, hipStream_t s)
{
    if constexpr (NQ <= 10)
        // best of the sweep at nq 8..10 (profiles/r01/tune_hex*_mfma2.log): chunks of 2 elements
        return launch_hex_mfma<NQ, 2, 2, 2, 1>(a, s);
    else
    {
        // One element's LDS image is 11-32 KB here, so LDS -- not registers -- bounds the residency: one
        // element per wave and one-wave workgroups keep 5-12 waves per CU in flight (chunks of 2 left 2-3).
        // Round 3, sweeps 1 and 2 fused (profiles/r03/tune_hex1[2-6]_fused_sweeps.log, 131 072 elements, mean GDOF/s and
        // fraction of the HBM roofline): 12: 323 = 0.74, 13: 335 = 0.76, 14: 292 = 0.66, 15: 293 = 0.65, 16: 309 = 0.68
        // (unfused: 0.71 / 0.75 / 0.68 / 0.64 / 0.66, with 256 VGPRs and a scratch spill at nq = 16).  Several elements
        // per wave with the next one's loads in flight (K = 2, 4, persistent) change nothing.  At nq = 14 the kernel issues
        // 156 tile products of 2048 flops per element for 199 kflop of useful work (13 and 14 pad to 16 in every
        // direction), i.e. 42 TFLOP/s issued at the measured rate -- the fp64 matrix rate this chip holds next to its
        // memory stream; nq = 16 (tiles 92 % full) is bounded by its 32 KB LDS image: five waves per CU.
        constexpr int WPB = NQ == 11 ? 4 : 1;
        constexpr int MW  = (NQ == 14 || NQ == 15) ? 2 : 1;
        return launch_hex_mfma<NQ, 1, WPB, MW, 1, 64>(a, s);
    }
}

int launch_hex_mfma_nq(unsigned nq, const HexArgs &a, hipStream_t s)
{
    switch (nq)
    {
    case 4: return go_mfma<4>(a, s);
    case 5: return go_mfma<5>(a, s);
    case 6: return go_mfma<6>(a, s);
    case 7: return go_mfma<7>(a, s);
    case 8: return go_mfma<8>(a, s);
    case 9: return go_mfma<9>(a, s);
    case 10: return go_mfma<10>(a, s);
    case 11: return go_mfma<11>(a, s);
    case 12: return go_mfma<12>(a, s);
    case 13: return go_mfma<13>(a, s);
    case 14: return go_mfma<14>(a, s);
    case 15: return go_mfma<15>(a, s);
    case 16: return go_mfma<16>(a, s);
    default: return SF_ENOTBUILT;
    }
}

// 4x4x4_4b matrix-core kernel (bwdtrans_hmfma4.h, hex_mfma4_kernel): orders 12..16, one element per wave.
// profiles/r03/tune_hex1[2-6]_mfma_4x4x4.log (131 072 elements, mean GDOF/s = fraction of the HBM roofline; in brackets
// the 16x16x4 kernel in the same run): 12: 327 = 0.75 (300-315)   13: 316 (323-329)   14: 293 = 0.675 with the k remainder
// peeled onto the vector pipe, 279 without (283-287)   15: 271 peeled, 274 not (290-292)   16: 319 = 0.72 with the
// accumulators stored directly and unpadded W2 rows (five workgroups per CU), 310 through the LDS output image (290-307).
// AUTO runs it at nq 12, 14 and 16 (hex_auto_kernel()).
template <int NQ> static int go_mfma4(const HexArgs &a, hipStream_t s)
{
    // SF_HEX_MFMA4_CFG is a development knob, read at every call (1: output through an LDS image, 3: accumulators stored
    // directly -- whole 128-byte lines only where nq is a multiple of 4 and `out` is 128-byte aligned)
    const char *env = getenv("SF_HEX_MFMA4_CFG");
    const int cfg   = env ? atoi(env) : (NQ == 16 ? 3 : 1);
    if (cfg == 3)
        return launch_hex_mfma4<NQ, 1, 2, 1, 64, true>(a, s);
    return launch_hex_mfma4<NQ, 1, (NQ == 12 ? 1 : 2), 1, 64>(a, s);
}

int launch_hex_mfma4_nq(unsigned nq, const HexArgs &a, hipStream_t s)
{
    switch (nq)
    {
    case 12: return go_mfma4<12>(a, s);
    case 13: return go_mfma4<13>(a, s);
    case 14: return go_mfma4<14>(a, s);
    case 15: return go_mfma4<15>(a, s);
    case 16: return go_mfma4<16>(a, s);
    default: return SF_ENOTBUILT;
    }
}

// the measured best matrix-core kernel above the wave kernel's table
int hex_auto_kernel(unsigned nq)
{
    return (nq == 12 || nq == 14 || nq == 16) ? SF_VARIANT_MFMA4 : SF_VARIANT_MFMA;
}

// fp32 (T = float): same kernels with float4 lanes.  Chunks hold twice the fp64 element count (same
// bytes), always the LDS-staged flat output (the DPP pair store is the fp64 path).
// nq = 2, T = float: the stream form of hex_nq2_stream_kernel with float4 lanes -- one 16-byte vector per thread holds the
// four (j, i) outputs of plane k, two threads per element; same multiplication order as the sweeps
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hex_nq2_stream_f32_kernel(const float *__restrict__ b0, const float *__restrict__ b1,
                                                                 const float *__restrict__ b2, const float *__restrict__ in,
                                                                 float *__restrict__ out, uint64_t nelmt)
{
    const uint64_t nv = nelmt * 2;
    const uint64_t v  = logical_block<64>() * 256ull + threadIdx.x;
    if (v < nv)
    {
        const float x  = in[v >> 1];
        const float bk = b2[v & 1];
        const float x0 = x * b0[0], x1 = x * b0[1];
        const float4_t r = {(x0 * b1[0]) * bk, (x1 * b1[0]) * bk, (x0 * b1[1]) * bk, (x1 * b1[1]) * bk};
        __builtin_nontemporal_store(r, reinterpret_cast<float4_t *>(out) + v);
    }
}

template <int NQ> static int go_f32(const HexArgsT<float> &a, hipStream_t s)
{
    if constexpr (NQ == 2)
    {
        if (a.nelmt == 0)
            return SF_OK;
        const uint64_t blocks = (a.nelmt * 2 + 255) / 256;
        if (blocks > 0x7fffffffull)
            return SF_EINVAL;
        hex_nq2_stream_f32_kernel<<<(unsigned)blocks, 256, 0, s>>>(a.b0, a.b1, a.b2, a.in, a.out, a.nelmt);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? SF_OK : (int)e;
    }
    if constexpr (NQ >= 12)
    {
        // fp32 matrix-core kernel (hex_mfma_kernel, T = float: v_mfma_f32_16x16x4_f32) from nq = 13, where it is ahead of
        // the vector kernel (tools/experiments/f32_hex_cfg.py, profiles/r03/f32_hex_mfma_configurations.log; 131 072
        // elements, fraction of the fp32 HBM roofline, vector kernel in brackets): 13 0.594 (0.518)  14 0.590 (0.552)
        // 15 0.676 (0.524)  16 0.705 (0.549); nq = 12 stays on the vector kernel (0.605 against 0.508).
        // SF_F32_HEX_CFG is a development knob (0: vector kernel).
        constexpr int best = NQ == 12 ? 0 : (NQ <= 14 ? 2 : 1);
        static const int cfg = getenv("SF_F32_HEX_CFG") ? atoi(getenv("SF_F32_HEX_CFG")) : best;
        switch (cfg)
        {
        case 1: return launch_hex_mfma<NQ, 1, 1, 1, 1, 64, float>(a, s);
        case 2: return launch_hex_mfma<NQ, 1, 1, 2, 1, 64, float>(a, s);
        case 3: return launch_hex_mfma<NQ, 1, 2, 2, 1, 64, float>(a, s);
        case 4: return launch_hex_mfma<NQ, 2, 1, 1, 1, 64, float>(a, s);
        default: break;
        }
    }
    using C = HexCfgF32<NQ>;
    return launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::OUT, C::MF, float>(a, s);
}

int launch_hex_wave_f32_nq(unsigned nq, const HexArgsT<float> &a, hipStream_t s)
{
    switch (nq)
    {
    case 2: return go_f32<2>(a, s);
    case 3: return go_f32<3>(a, s);
    case 4: return go_f32<4>(a, s);
    case 5: return go_f32<5>(a, s);
    case 6: return go_f32<6>(a, s);
    case 7: return go_f32<7>(a, s);
    case 8: return go_f32<8>(a, s);
    case 9: return go_f32<9>(a, s);
    case 10: return go_f32<10>(a, s);
    case 11: return go_f32<11>(a, s);
    case 12: return go_f32<12>(a, s);
    case 13: return go_f32<13>(a, s);
    case 14: return go_f32<14>(a, s);
    case 15: return go_f32<15>(a, s);
    case 16: return go_f32<16>(a, s);
    default: return SF_ENOTBUILT;
    }
}

} // namespace sf
