// wave_table.h -- the tuned configuration of the wave-per-chunk kernels, one row per isotropic nq.
// Shared by the dispatchers (bwdtrans_hex.hip, bwdtrans_quad.hip) and tools/sf_tune_table.hip, which
// re-measures every row (and its memory-flag alternatives) on the device.
#pragma once
#include "bwdtrans_wave.h"

namespace sf
{

// MF: memory flags of the kernel (bit 0: plain instead of non-temporal loads, bit 1: plain stores,
// bit 2: shift the chunk's 16-byte lanes so every wave-wide load covers whole 128-byte lines,
// bit 3: the same for the OUT_LDS output stream, bits 4-15: XG, runs of XG neighbouring workgroups per XCD, bit 16: workgroup-cooperative input -- measured, not used)
// NQ -> elements per chunk, waves per block, basis delivery, min waves/SIMD, chunk mapping
// (0 = persistent), 16-byte stores
// XCD grouping (bwdtrans_wave.h, logical_block()): runs of XG neighbouring workgroups execute on the same XCD
constexpr int XG16 = 16 << 4, XG64 = 64 << 4;

template <int NQ> struct HexCfg;
#define SF_HEX_CFG(NQ_, EC_, WPB_, BM_, MW_, KM_, OUT_, MF_)                                        \
    template <> struct HexCfg<NQ_>                                                                 \
    {                                                                                              \
        static constexpr int EC = EC_, WPB = WPB_, BM = BM_, MW = MW_, KM = KM_;                   \
        static constexpr int OUT = OUT_, MF = MF_;                                                      \
    }
//          nq  EC  WPB  basis       MINW KMAP out        GDOF/s min/mean @1Mi elements (profiles/r01/tune_hex*.log)
SF_HEX_CFG(2,  128, 4, BASIS_SMEM, 2, 1, OUT_ST16, 0); //  73 /  70
SF_HEX_CFG(3,  14,  2, BASIS_SMEM, 2, 1, OUT_LDS, XG64 | 8);  // 167 (four waves per workgroup, no XCD runs: 162; chunks of 4-8 elements: 158)
SF_HEX_CFG(4,  4,   8, BASIS_SMEM, 4, 1, OUT_LDS, XG64);  // 237 (EC 8, 4 waves/block: 229)
SF_HEX_CFG(5,  2,   4, BASIS_SMEM, 2, 1, OUT_LDS, XG64 | 8);  // 267 (profiles/r01/tune_hex5_xcd_runs.log)
SF_HEX_CFG(6,  2,   8, BASIS_SMEM, 2, 1, OUT_LDS, XG64);  // 284
SF_HEX_CFG(7,  1,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64 | 8);  // 305 (one element per wave; output via the word-grid store)
SF_HEX_CFG(8,  1,   8, BASIS_SMEM, 4, 1, OUT_LDS, XG64);  // 322 / 316-320 (without XCD runs, 4 waves/block: 312; four-element chunks: 285-303)
SF_HEX_CFG(9,  1,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64 | 8);  // 317
SF_HEX_CFG(10, 1,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64 | 8);  // 313 / 309 (two-element chunks: 305 / 300)
SF_HEX_CFG(11, 1,   4, BASIS_SMEM, 1, 1, OUT_LDS, XG64);  // 314 (matrix-core kernel: 281); 131 072 elements
#undef SF_HEX_CFG

// elements per launch when a batch is large enough to be enqueued in pieces (bwdtrans_hex.hip, go<NQ>()); 0 = never
constexpr uint64_t hex_piece(int nq)
{
    return (nq == 7 || nq == 8) ? (1ull << 19) : 0;
}

template <int NQ> struct QuadCfg;
#define SF_QUAD_CFG(NQ_, EC_, WPB_, BM_, MW_, KM_, OUT_, MF_)                                       \
    template <> struct QuadCfg<NQ_>                                                                \
    {                                                                                              \
        static constexpr int EC = EC_, WPB = WPB_, BM = BM_, MW = MW_, KM = KM_, OUT = OUT_;       \
        static constexpr int MF = MF_;                                                             \
    }
// scalar-operand rows need 2*nq SGPRs each (ring of 3): beyond nq ~ 10 they spill -> LDS copy of the basis
//           nq  EC  WPB  basis      MINW KMAP out        GDOF/s min/mean @1Mi (profiles/r01/tune_quad*.log)
SF_QUAD_CFG(2,  128, 4, BASIS_SMEM, 2, 1, OUT_LDS, 0);  // 115 / 112 (9 us kernel: launch-bound)
SF_QUAD_CFG(3,  42,  4, BASIS_SMEM, 2, 1, OUT_LDS, 8);
SF_QUAD_CFG(4,  16,  4, BASIS_SMEM, 2, 1, OUT_LDS, XG64);  // 260 (without XCD runs: 253) -- profiles/r02/tune_quad_low_orders_xcd_runs.log
SF_QUAD_CFG(5,  8,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64 | 8);  // 294 (24 elements, no XCD runs: 278)
SF_QUAD_CFG(6,  8,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64);  // 319 (ten elements without XCD runs: 305-308)
SF_QUAD_CFG(7,  6,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64 | 12); // 333 (18 elements: 308)
SF_QUAD_CFG(8,  4,   4, BASIS_SMEM, 4, 1, OUT_LDS, XG64);  // 343-346 (eight elements, paired register stores: 315-317)
SF_QUAD_CFG(9,  4,   4, BASIS_SMEM, 2, 1, OUT_LDS, 8);  // 351 (14 elements: 328; XCD runs: 345-349)
SF_QUAD_CFG(10, 4,   8, BASIS_SMEM, 2, 1, OUT_LDS, XG64);  // 349 (twelve elements: 334)
SF_QUAD_CFG(11, 4,   4, BASIS_SMEM_COLS, 2, 1, OUT_LDS, XG64 | 8); // 353 (ten elements: 311-320; LDS copy of the basis: 297)
SF_QUAD_CFG(12, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, XG64);      // 352-355 (four elements: 348; matrix-core kernel with XCD runs: 344)
SF_QUAD_CFG(13, 4,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, XG64 | 12); // 355 (341)
SF_QUAD_CFG(14, 4,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, XG64);      // 358 (341)
SF_QUAD_CFG(15, 4,   8, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, XG64 | 12); // 359 (336)
SF_QUAD_CFG(16, 4,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS, XG64);      // 366 (347)
// nq 17..24: vector-ALU kernel with column-blocked scalar operands (16 columns per SGPR ring): the padded
// 16x16x4 matrix-core tiles need more pipe cycles here than the exact-size FMAs (profiles/r01/tune_quad*_scol2.log)
SF_QUAD_CFG(17, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS,  XG64); // 328 (matrix-core kernel: 282); 3 elements fill 48-51 of the 64 lanes
SF_QUAD_CFG(18, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS,  XG64); // 332 (207)
SF_QUAD_CFG(19, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS,  XG64); // 334 (233)
SF_QUAD_CFG(20, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS,  XG64); // 341 (236); VALU 0.89 busy with 2 elements (38-40 lanes)
SF_QUAD_CFG(21, 3,   4, BASIS_SMEM_COLS16, 2, 1, OUT_ST8,  XG64); // 329 (257)
SF_QUAD_CFG(22, 2,   4, BASIS_SMEM_COLS16, 2, 1, OUT_LDS,  XG64); // 308 (247)
SF_QUAD_CFG(23, 2,   4, BASIS_SMEM_COLS16, 2, 1, OUT_ST8,  XG64); // 306 (260)
SF_QUAD_CFG(24, 2,   4, BASIS_SMEM_COLS16, 2, 1, OUT_ST16, XG64); // 282 (270)
SF_QUAD_CFG(32, 2,   4, BASIS_SMEM_COLS16, 1, 2, OUT_ST16, 0); // 135 (LDS copy of the basis: 118): VALU-issue-bound; AUTO uses the matrix cores (340)
#undef SF_QUAD_CFG

// fp32 rows follow from the fp64 ones: twice the elements per chunk (the same bytes), twice the waves per
// SIMD (half the register footprint, capped at 4), one chunk per wave, LDS-staged 16-byte output stream.
// The memory flags are tuned separately (tools/sf_tune_table hexf32 / quadf32).
constexpr int hex_f32_mf(int nq)
{
    return ((nq % 2 && nq >= 3) ? 8 : 0) | (nq >= 7 ? XG64 : 0); // odd orders: line-aligned output; XCD runs from nq = 7 (+3-6 %)
}
constexpr int quad_f32_mf(int nq)
{
    return (nq == 15 ? 12 : ((nq == 7 || nq == 9 || nq == 11) ? 8 : 0)) | (nq >= 12 ? XG64 : 0);
}
template <int NQ> struct HexCfgF32
{
    static constexpr int EC = 2 * HexCfg<NQ>::EC, WPB = HexCfg<NQ>::WPB, BM = HexCfg<NQ>::BM;
    static constexpr int MW = HexCfg<NQ>::MW >= 2 ? 4 : 2, KM = 1, OUT = OUT_LDS, MF = hex_f32_mf(NQ);
};
// nq = 7, 9: the fp64 rows moved to one element per chunk (their odd nq^3 output then leaves through the 8-byte
// word-grid store, which is fp64-only); fp32 keeps the two-element-based rule (566 / 585 GDOF/s against 492 / 520)
template <> struct HexCfgF32<3> // pinned: the fp64 row moved to two-wave workgroups under XCD runs (measured for fp64 only)
{
    static constexpr int EC = 28, WPB = 4, BM = BASIS_SMEM, MW = 4, KM = 1, OUT = OUT_LDS, MF = 8;
};
template <> struct HexCfgF32<6>
{
    static constexpr int EC = 4, WPB = 4, BM = BASIS_SMEM, MW = 4, KM = 1, OUT = OUT_LDS, MF = 0; // 552 (8 waves/block: 527)
};
template <> struct HexCfgF32<7>
{
    static constexpr int EC = 4, WPB = 4, BM = BASIS_SMEM, MW = 4, KM = 1, OUT = OUT_LDS, MF = XG64 | 8;
};
template <> struct HexCfgF32<9>
{
    static constexpr int EC = 4, WPB = 2, BM = BASIS_SMEM, MW = 2, KM = 1, OUT = OUT_LDS, MF = XG64 | 8;
};
template <int NQ> struct QuadCfgF32
{
    static constexpr int EC = 2 * QuadCfg<NQ>::EC, WPB = QuadCfg<NQ>::WPB, BM = QuadCfg<NQ>::BM;
    static constexpr int MW = QuadCfg<NQ>::MW >= 2 ? 4 : 2, KM = 1, OUT = OUT_LDS, MF = quad_f32_mf(NQ);
};

// fp32, 2D nq = 4 / 6 / 8 / 10: pinned and measured on their own (profiles/r02/tune_quad_low_orders_xcd_runs.log): the fp64
// rows of these orders moved to four-element chunks under XCD runs; T = float wants eight (the same bytes) -- but not the
// eight-wave workgroups of the fp64 nq = 10 row, and 32 / 16 elements at nq = 4 / 6
#define SF_QUAD_F32_LOW(NQ_, EC_, MW_)                                                             \
    template <> struct QuadCfgF32<NQ_>                                                             \
    {                                                                                              \
        static constexpr int EC = EC_, WPB = 4, BM = BASIS_SMEM, MW = MW_, KM = 1, OUT = OUT_LDS, MF = XG64; \
    }
SF_QUAD_F32_LOW(4, 32, 4);  // 465 (without XCD runs: 440)
SF_QUAD_F32_LOW(6, 16, 4);  // 598-610 (20 elements, no XCD runs: 593-600)
SF_QUAD_F32_LOW(8, 8, 4);   // 688-691 (16 elements: 662-667)
SF_QUAD_F32_LOW(10, 8, 4);  // 699-703 (24 elements at two waves per SIMD: 669-671)
#undef SF_QUAD_F32_LOW

// fp32, 2D nq = 5 / 7 / 9 / 11: likewise pinned (BM_: scalar-operand rows, column blocks at nq = 11)
#define SF_QUAD_F32_ODD(NQ_, EC_, BM_, MW_, MF_)                                                   \
    template <> struct QuadCfgF32<NQ_>                                                             \
    {                                                                                              \
        static constexpr int EC = EC_, WPB = 4, BM = BM_, MW = MW_, KM = 1, OUT = OUT_LDS, MF = MF_; \
    }
SF_QUAD_F32_ODD(5, 16, BASIS_SMEM, 4, XG64 | 8);       // 552 (48 elements, no XCD runs: 503)
SF_QUAD_F32_ODD(7, 16, BASIS_SMEM, 4, XG64 | 8);       // 628 (36 elements: 596)
SF_QUAD_F32_ODD(9, 8, BASIS_SMEM, 4, 8);               // 683 (28 elements: 644; with XCD runs: 665)
SF_QUAD_F32_ODD(11, 8, BASIS_SMEM_COLS, 4, XG64 | 8);  // 686 (20 elements at two waves per SIMD: 656)
#undef SF_QUAD_F32_ODD

// fp32, 2D nq = 12 .. 16, pinned row by row (not derived from the fp64 rows, which are re-tuned independently): the doubled
// chunks (16-20 elements) leave 8 waves per CU; four-element chunks measure 721-728 / 727-738 / 727-733 / 770 GDOF/s at
// nq = 12 / 14 / 15 / 16 against 694-701 / 699-712 / 553-558 / 655-659; nq = 13 keeps 16 elements at two waves per SIMD
// (709-716 / 701 mean against 680 / 660) -- profiles/r02/tune_f32_chunk_size_2d.log
#define SF_QUAD_F32(NQ_, EC_, MW_, MF_)                                                            \
    template <> struct QuadCfgF32<NQ_>                                                             \
    {                                                                                              \
        static constexpr int EC = EC_, WPB = 4, BM = BASIS_SMEM_COLS16, MW = MW_, KM = 1, OUT = OUT_LDS, MF = MF_; \
    }
SF_QUAD_F32(12, 4, 4, XG64);
SF_QUAD_F32(13, 16, 2, XG64 | 12);
SF_QUAD_F32(14, 4, 4, XG64);
SF_QUAD_F32(15, 4, 4, XG64 | 12);
SF_QUAD_F32(16, 4, 4, XG64);
SF_QUAD_F32(20, 8, 4, XG64); // 563 / 520 (six elements: 2 x the fp64 row; four: 490 / 478)
// nq 25..31: fp64 runs these orders on the matrix cores (no fp32 MFMA kernel here); in fp32 the vector-ALU kernel's
// registers fit (two elements: one pencil pass, 61 VGPRs of operands), so the T = float rows extend the wave table
SF_QUAD_F32(25, 2, 4, XG64 | 8);
SF_QUAD_F32(26, 2, 4, XG64);
SF_QUAD_F32(27, 2, 4, XG64 | 8);
SF_QUAD_F32(28, 2, 4, XG64);
SF_QUAD_F32(29, 2, 4, XG64 | 8);
SF_QUAD_F32(30, 2, 4, XG64);
SF_QUAD_F32(31, 2, 4, XG64 | 8);
#undef SF_QUAD_F32

// fp32, 3D nq = 12 .. 16: fp64 uses the matrix-core kernel there (the fp64 wave kernel would need > 256 VGPRs); with
// 4-byte scalars one element per wave fits (4 pencil passes: 124 VGPRs of operands at nq = 16, 16 KB of LDS per wave)
#define SF_HEX_F32(NQ_, WPB_, MW_, MF_)                                                            \
    template <> struct HexCfgF32<NQ_>                                                              \
    {                                                                                              \
        static constexpr int EC = 1, WPB = WPB_, BM = BASIS_SMEM, MW = MW_, KM = 1, OUT = OUT_LDS, MF = MF_; \
    }
SF_HEX_F32(12, 4, 2, XG64);
SF_HEX_F32(13, 4, 2, XG64 | 8);
SF_HEX_F32(14, 4, 2, XG64);
SF_HEX_F32(15, 2, 2, XG64 | 8);
SF_HEX_F32(16, 2, 2, XG64);
#undef SF_HEX_F32

} // namespace sf
