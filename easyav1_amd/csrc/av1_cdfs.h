// The adaptive contexts of the AV1 symbol decoder (AV1 specification 8.3: one CDF array per syntax element and context)
// and the caller-supplied table blob they start from.  Part of the block-level parser (dav1d
// src/cdf.c, src/scan.c, src/dequant_tables.c: /root/reference/cmake/dav1d.cmake:971,990,974).
//
// TABLE-AGNOSTIC BY DESIGN: the default CDFs (about 15 000 numbers), the coefficient scan orders and the quantiser
// look-ups are DATA of the AV1 specification that cannot be written down from memory to the last digit.  The parser
// owns none of them: ea1_av1_tables_load() takes one blob (layout below, described at run time by
// ea1_av1_tables_section()) from whoever has the normative tables -- exactly as Gaussian_Sequence reaches the film
// grain kernel.  The tests run the parser with ARBITRARY tables (random CDFs, generated scans): the syntax, the
// context derivation and the adaptation are what is under test, and they do not depend on the numbers.
//
// A CDF is kept as the specification prints it: N increasing 15-bit cumulative values, the last 32768, then the
// adaptation counter (N + 1 entries of uint16_t).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace ea1 {
namespace av1 {

// X(name, dimensions, symbols): uint16_t name dimensions [symbols + 1]
// Names follow the specification's Default_*_Cdf tables; dimensions are the context indices of 8.3.2.
#define EA1_AV1_MODE_CDFS(X)                                                                                          \
    X(IntraFrameYMode, [5][5], 13)                                                                                    \
    X(YMode, [4], 13)                                                                                                 \
    X(UVModeCflNotAllowed, [13], 13)                                                                                  \
    X(UVModeCflAllowed, [13], 14)                                                                                     \
    X(AngleDelta, [8], 7)                                                                                             \
    X(FilterIntraMode, , 5)                                                                                           \
    X(FilterIntra, [22], 2)                                                                                           \
    X(CflSign, , 8)                                                                                                   \
    X(CflAlpha, [6], 16)                                                                                              \
    X(Partition8, [4], 4)                                                                                             \
    X(Partition16, [4], 10)                                                                                           \
    X(Partition32, [4], 10)                                                                                           \
    X(Partition64, [4], 10)                                                                                           \
    X(Partition128, [4], 8)                                                                                           \
    X(Skip, [3], 2)                                                                                                   \
    X(SkipMode, [3], 2)                                                                                               \
    X(IsInter, [4], 2)                                                                                                \
    X(SegmentId, [3], 8)                                                                                              \
    X(SegmentIdPredicted, [3], 2)                                                                                     \
    X(DeltaQ, , 4)                                                                                                    \
    X(DeltaLF, , 4)                                                                                                   \
    X(DeltaLFMulti, [4], 4)                                                                                           \
    X(Tx8x8, [3], 2)                                                                                                  \
    X(Tx16x16, [3], 3)                                                                                                \
    X(Tx32x32, [3], 3)                                                                                                \
    X(Tx64x64, [3], 3)                                                                                                \
    X(TxfmSplit, [21], 2)                                                                                             \
    X(IntraTxTypeSet1, [2][13], 7)                                                                                    \
    X(IntraTxTypeSet2, [3][13], 5)                                                                                    \
    X(InterTxTypeSet1, [2], 16)                                                                                       \
    X(InterTxTypeSet2, , 12)                                                                                          \
    X(InterTxTypeSet3, [4], 2)                                                                                        \
    X(CompMode, [5], 2)                                                                                               \
    X(CompRefType, [5], 2)                                                                                            \
    X(UniCompRef, [3][3], 2)                                                                                          \
    X(CompRef, [3][3], 2)                                                                                             \
    X(CompBwdRef, [3][2], 2)                                                                                          \
    X(SingleRef, [3][6], 2)                                                                                           \
    X(NewMv, [6], 2)                                                                                                  \
    X(ZeroMv, [2], 2)                                                                                                 \
    X(RefMv, [6], 2)                                                                                                  \
    X(CompoundMode, [8], 8)                                                                                           \
    X(DrlMode, [3], 2)                                                                                                \
    X(InterpFilter, [16], 3)                                                                                          \
    X(MotionMode, [22], 3)                                                                                            \
    X(UseObmc, [22], 2)                                                                                               \
    X(InterIntra, [3], 2)                                                                                             \
    X(InterIntraMode, [3], 4)                                                                                         \
    X(WedgeInterIntra, [22], 2)                                                                                       \
    X(WedgeIndex, [22], 16)                                                                                           \
    X(CompoundType, [22], 2)                                                                                          \
    X(CompGroupIdx, [6], 2)                                                                                           \
    X(CompoundIdx, [6], 2)                                                                                            \
    X(MvJoint, [2], 4)                                                                                                \
    X(MvSign, [2][2], 2)                                                                                              \
    X(MvClass, [2][2], 11)                                                                                            \
    X(MvClass0Bit, [2][2], 2)                                                                                         \
    X(MvClass0Fr, [2][2][2], 4)                                                                                       \
    X(MvClass0Hp, [2][2], 2)                                                                                          \
    X(MvBit, [2][2][10], 2)                                                                                           \
    X(MvFr, [2][2], 4)                                                                                                \
    X(MvHp, [2][2], 2)                                                                                                \
    X(Intrabc, , 2)                                                                                                   \
    X(PaletteYMode, [7][3], 2)                                                                                        \
    X(PaletteUVMode, [2], 2)                                                                                          \
    X(PaletteYSize, [7], 7)                                                                                           \
    X(PaletteUVSize, [7], 7)                                                                                          \
    X(PaletteSize2YColor, [5], 2)                                                                                     \
    X(PaletteSize3YColor, [5], 3)                                                                                     \
    X(PaletteSize4YColor, [5], 4)                                                                                     \
    X(PaletteSize5YColor, [5], 5)                                                                                     \
    X(PaletteSize6YColor, [5], 6)                                                                                     \
    X(PaletteSize7YColor, [5], 7)                                                                                     \
    X(PaletteSize8YColor, [5], 8)                                                                                     \
    X(PaletteSize2UVColor, [5], 2)                                                                                    \
    X(PaletteSize3UVColor, [5], 3)                                                                                    \
    X(PaletteSize4UVColor, [5], 4)                                                                                    \
    X(PaletteSize5UVColor, [5], 5)                                                                                    \
    X(PaletteSize6UVColor, [5], 6)                                                                                    \
    X(PaletteSize7UVColor, [5], 7)                                                                                    \
    X(PaletteSize8UVColor, [5], 8)                                                                                    \
    X(UseWiener, , 2)                                                                                                 \
    X(UseSgrproj, , 2)                                                                                                \
    X(RestorationType, , 3)

// the coefficient CDFs: the blob holds four copies, one per quantiser class of 8.3.? init_coeff_cdfs (base_q_idx <= 20, <= 60,
// <= 120, above), a tile context holds the one its frame selected
#define EA1_AV1_COEF_CDFS(X)                                                                                          \
    X(TxbSkip, [5][13], 2)                                                                                            \
    X(EobPt16, [2][2], 5)                                                                                             \
    X(EobPt32, [2][2], 6)                                                                                             \
    X(EobPt64, [2][2], 7)                                                                                             \
    X(EobPt128, [2][2], 8)                                                                                            \
    X(EobPt256, [2][2], 9)                                                                                            \
    X(EobPt512, [2], 10)                                                                                              \
    X(EobPt1024, [2], 11)                                                                                             \
    X(EobExtra, [5][2][9], 2)                                                                                         \
    X(DcSign, [2][3], 2)                                                                                              \
    X(CoeffBaseEob, [5][2][4], 3)                                                                                     \
    X(CoeffBase, [5][2][42], 4)                                                                                       \
    X(CoeffBr, [4][2][21], 4)

#define EA1_AV1_DECLARE_CDF(name, dims, n) uint16_t name dims[(n) + 1];
struct ModeCdfs {
    EA1_AV1_MODE_CDFS(EA1_AV1_DECLARE_CDF)
};
struct CoefCdfs {
    EA1_AV1_COEF_CDFS(EA1_AV1_DECLARE_CDF)
};
struct Cdfs {
    ModeCdfs m;
    CoefCdfs c;
    // the vector symbol decoders (av1_frame_driver.h MsacReader) load and store whole 16-byte vectors at a CDF array's start: behind the last array
    // there must be memory of this object's own (round-4 advisor: it used to hold only because the last member has four symbols)
    uint16_t vector_tail[8] = {};
};

// one section of the blob / one member of the structs above
struct CdfSection {
    const char *name;
    size_t offset;   // in ModeCdfs or CoefCdfs
    int count;       // CDFs in the section (product of the dimensions)
    int symbols;     // N
};
extern const CdfSection kModeSections[];
extern const int kNumModeSections;
extern const CdfSection kCoefSections[];
extern const int kNumCoefSections;

// coefficient scan orders: position (row * width + column) of the c-th coefficient, for the transform sizes that are coded
// (64-sample sides are coded as 32).  Index = the transform size numbering of easyav1_amd_video.h (ea1_tx_size_index).
// kind 0: Default_Scan_*, 1: Mrow_Scan_* (sizes up to 16 x 16), 2: Mcol_Scan_*
struct ScanSection {
    const char *name;
    int kind, wl2, hl2;   // log2 of the coded width / height
    size_t offset;        // entries from the start of Tables::scans
};
const ScanSection *scan_section_table();
int scan_section_count();
constexpr size_t kScanEntries = 4912;   // sum of the sections' sizes (checked at start-up)

struct Tables {
    ModeCdfs mode;
    CoefCdfs coef[4];
    uint16_t scans[kScanEntries];
    const uint16_t *scan[3][19];   // [kind][tx size index], nullptr where the specification has none
    int16_t dc_q[256], ac_q[256];  // Dc_Qlookup[0] / Ac_Qlookup[0] (8 bit)
    // Quantizer_Matrix: levels 0..14 x {luma, chroma} x the 14 coded transform sizes (the order of the scans), row-major per size
    uint8_t qmatrix[15][2][3344];
    const uint8_t *qm[15][2][19];  // [level][plane > 0][tx size index]; 64-sample sides use the matrix of 32
    int16_t upscale[64][8];        // Upscale_Filter (super-resolution, 7.16): the reconstruction's, travels with the same tables
    int16_t gaussian[2048];        // Gaussian_Sequence (film grain synthesis, 7.18.3.2): not the parser's, but it travels with the same tables
};

}  // namespace av1
}  // namespace ea1
struct ea1_av1_tables;
namespace ea1 {
namespace av1 {
const Tables *tables_of(const ea1_av1_tables *t);

// every adaptation counter of a context back to zero (8.? exit process of the symbol decoder / frame end update)
void reset_counters(Cdfs *c);

}  // namespace av1
}  // namespace ea1
