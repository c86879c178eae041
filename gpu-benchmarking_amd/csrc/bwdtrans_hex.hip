// bwdtrans_hex.hip -- compile-time instantiations of the 3D hex wave kernel + nq dispatch.
// One row per isotropic nq; the tuple (EC, WPB, BMODE, MINW) is the tuned configuration
// (tools/sf_tune prints the sweep these were picked from; DESIGN.md records the numbers).
#include "wave_launch.h"

namespace sf
{

// NQ -> elements per chunk, waves per block, basis delivery, min waves/SIMD, chunk mapping
// (0 = persistent), 16-byte stores
template <int NQ> struct HexCfg;
template <> struct HexCfg<2>  { static constexpr int EC = 64, WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<3>  { static constexpr int EC = 14, WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<4>  { static constexpr int EC = 8,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<5>  { static constexpr int EC = 5,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<6>  { static constexpr int EC = 6,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<7>  { static constexpr int EC = 5,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<8>  { static constexpr int EC = 2,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<9>  { static constexpr int EC = 3,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };
template <> struct HexCfg<10> { static constexpr int EC = 2,  WPB = 4, BM = BASIS_SMEM, MW = 2, KM = 0; static constexpr bool S16 = false; };

template <int NQ> static int go(const HexArgs &a, hipStream_t s)
{
    using C = HexCfg<NQ>;
    return launch_hex_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::S16 && (NQ % 2 == 0)>(a, s);
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
    default: return SF_ENOTBUILT;
    }
}

} // namespace sf
