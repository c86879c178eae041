// bwdtrans_hex.hip -- compile-time instantiations of the 3D hex wave kernel + nq dispatch.
// One row per isotropic nq; the tuple (EC, WPB, BMODE, MINW) is the tuned configuration
// (tools/sf_tune prints the sweep these were picked from; DESIGN.md records the numbers).
#include "sf_dispatch.h"
#include "wave_launch.h"
#include "wave_table.h"

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

template <int NQ> static int go(const HexArgs &a, hipStream_t s)
{
    using C = HexCfg<NQ>;
    constexpr uint64_t per_block = (uint64_t)C::EC * C::WPB * (C::KM > 0 ? C::KM : 1);
    if (a.nelmt < 2 * per_block * (uint64_t)device_info().num_cu)
        return launch_hex_wave<NQ, HexSmall<NQ>::EC, 1, C::BM, C::MW, 1, HexSmall<NQ>::OUT>(a, s);
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
        // element per wave and small workgroups keep 5-12 waves per CU in flight (chunks of 2 left 2-3).
        // profiles/r01/tune_hex1[1-6]_mfma.log: 265 / 313 / 266 / 283 / 252 / 286 GDOF/s at nq = 11..16
        // (chunks of 2: 212 / 211 / 255 / 132 / 138 / 162).
        constexpr int WPB = NQ == 11 ? 4 : (NQ <= 12 ? 2 : 1);
        return launch_hex_mfma<NQ, 1, WPB, 2, 1, 64>(a, s);
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

// fp32 (T = float): same kernels with float4 lanes.  Chunks hold twice the fp64 element count (same
// bytes), always the LDS-staged flat output (the DPP pair store is the fp64 path).
template <int NQ> static int go_f32(const HexArgsT<float> &a, hipStream_t s)
{
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
    default: return SF_ENOTBUILT;
    }
}

} // namespace sf
