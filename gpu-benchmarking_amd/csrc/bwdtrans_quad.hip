// bwdtrans_quad.hip -- compile-time instantiations of the 2D quad wave kernel + nq dispatch.
// (EC, WPB, BMODE, MINW) per nq: tuned configuration, see tools/sf_tune and DESIGN.md.
#include "wave_launch.h"

namespace sf
{

template <int NQ> struct QuadCfg
{
    // default: fill one wave pass with elements, two passes per lane
    static constexpr int PER = (kWave / NQ) < 1 ? 1 : (kWave / NQ);
    static constexpr int EC0 = (NQ <= 10) ? 2 * PER : PER;
    // chunk must hold an even number of doubles for the 16-byte loads
    static constexpr int EC  = ((EC0 * (NQ - 1) * (NQ - 1)) % 2 == 0) ? EC0 : EC0 + 1;
    // scalar-operand rows need 2*NQ SGPRs each (ring of 3): beyond nq ~ 10 they spill -> LDS copy
    static constexpr int WPB = 4, BM = (NQ <= 10) ? BASIS_SMEM : BASIS_LDS, MW = (NQ <= 16) ? 2 : 1,
                         KM = 2;
    static constexpr bool S16 = (NQ % 2 == 0);
};

template <int NQ> static int go(const QuadArgs &a, hipStream_t s)
{
    using C = QuadCfg<NQ>;
    return launch_quad_wave<NQ, C::EC, C::WPB, C::BM, C::MW, C::KM, C::S16>(a, s);
}

int launch_quad_wave_nq(unsigned nq, const QuadArgs &a, hipStream_t s)
{
    switch (nq)
    {
#define SF_CASE(N) case N: return go<N>(a, s);
        SF_CASE(2) SF_CASE(3) SF_CASE(4) SF_CASE(5) SF_CASE(6) SF_CASE(7) SF_CASE(8) SF_CASE(9)
        SF_CASE(10) SF_CASE(11) SF_CASE(12) SF_CASE(13) SF_CASE(14) SF_CASE(15) SF_CASE(16)
        SF_CASE(32)
#undef SF_CASE
    default: return SF_ENOTBUILT;
    }
}

} // namespace sf
