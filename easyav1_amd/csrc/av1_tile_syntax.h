// Block-level syntax of an AV1 tile (AV1 specification 5.11 "tile group OBU syntax" from decode_tile down, with the
// context derivation of 8.3.2, motion vector prediction 7.10, the coefficient reading of 5.11.39 and the loop filter
// level derivation of 7.14.4): everything between the tile's bytes and the hand-over structures of
// easyav1_amd_video.h.  What dav1d does in src/decode.c, src/refmvs.c, src/recon_tmpl.c (read side), src/lf_mask.c
// (/root/reference/cmake/dav1d.cmake:975,985,1014,980), written from the specification's processes, not from dav1d's
// code (absent here).  Host C++ by design (BASELINE.json north_star keeps the symbol parse on the CPU).
//
// The traversal is a template over its symbol source: TileSyntax<MsacReader> parses, TileSyntax<of a writer> walks the
// SAME syntax choosing values and encoding them (tests/: the bit-exact stream generator the round-trip tests need).
//
// Parity: PINNED since round 4 against dav1d 1.5.3 (the decoder the reference pins; its dav1d_* entry points out of Pillow's bundled libavif) on libaom 3.13.2 streams, bit-exact: tests/test_real_av1.py, tests/golden/make_video_fixtures.py, DESIGN.md section 2.
// (Rounds 1-3 had no decoder to compare with: the syntax order, the context rules and the small constant tables below were written from
// knowledge of the specification and marked [recalled]; real streams then found two slips -- the inverse map of the 16-type inter transform
// set and the stop condition of the distance weights.  What libaom does not emit -- super-resolution, scaled references, delta_lf, switch
// frames -- is still covered by the generator round trip only.)  The big tables come as a blob (av1_cdfs.h; data/av1_spec_tables.bin).
//
// Scope of this revision (anything else ends the frame with EA1_ENOTSUP and a message): 8-bit 4:2:0 and 4:0:0.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

#include "av1_cdfs.h"
#include "ea1_av1_tables.inc"
#include "easyav1_amd_obu.h"
#include "easyav1_amd_video.h"

namespace ea1 {
namespace av1 {

// ---- block sizes (specification numbering) ----
enum {
    BLOCK_4X4, BLOCK_4X8, BLOCK_8X4, BLOCK_8X8, BLOCK_8X16, BLOCK_16X8, BLOCK_16X16, BLOCK_16X32, BLOCK_32X16, BLOCK_32X32, BLOCK_32X64,
    BLOCK_64X32, BLOCK_64X64, BLOCK_64X128, BLOCK_128X64, BLOCK_128X128, BLOCK_4X16, BLOCK_16X4, BLOCK_8X32, BLOCK_32X8, BLOCK_16X64,
    BLOCK_64X16, BLOCK_SIZES, BLOCK_INVALID = 255
};
constexpr uint8_t kBlkWl2[BLOCK_SIZES] = {2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 2, 4, 3, 5, 4, 6};
constexpr uint8_t kBlkHl2[BLOCK_SIZES] = {2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6, 7, 6, 7, 4, 2, 5, 3, 6, 4};
constexpr uint8_t kSizeGroup[BLOCK_SIZES] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 0, 0, 1, 1, 2, 2};
constexpr uint8_t kMaxTxDepth[BLOCK_SIZES] = {0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 2, 2, 3, 3, 4, 4};
// Wedge_Bits: 4 for the nine sizes with a wedge codebook (8x8 .. 32x32, 8x32, 32x8), else 0
constexpr uint8_t kWedgeBits[BLOCK_SIZES] = {0, 0, 0, 4, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 0, 0};
inline int block_size_of(int wl2, int hl2)
{
    for (int i = 0; i < BLOCK_SIZES; i++)
        if (kBlkWl2[i] == wl2 && kBlkHl2[i] == hl2) return i;
    return BLOCK_INVALID;
}
inline int bw4_of(int bs) { return 1 << (kBlkWl2[bs] - 2); }
inline int bh4_of(int bs) { return 1 << (kBlkHl2[bs] - 2); }

enum { PARTITION_NONE, PARTITION_HORZ, PARTITION_VERT, PARTITION_SPLIT, PARTITION_HORZ_A, PARTITION_HORZ_B, PARTITION_VERT_A, PARTITION_VERT_B,
       PARTITION_HORZ_4, PARTITION_VERT_4 };

// ---- prediction modes (specification numbering; 0..12 and UV_CFL as in easyav1_amd_video.h) ----
enum { NEARESTMV = 13, NEARMV, GLOBALMV, NEWMV, NEAREST_NEARESTMV, NEAR_NEARMV, NEAREST_NEWMV, NEW_NEARESTMV, NEAR_NEWMV, NEW_NEARMV, GLOBAL_GLOBALMV,
       NEW_NEWMV };
enum { INTRA_FRAME = 0, LAST_FRAME, LAST2_FRAME, LAST3_FRAME, GOLDEN_FRAME, BWDREF_FRAME, ALTREF2_FRAME, ALTREF_FRAME, REF_NONE = 0xFF };
enum { MOTION_SIMPLE = 0, MOTION_OBMC, MOTION_LOCALWARP };
enum { INTERP_SWITCHABLE = 4 };

// transform types (specification numbering)
enum { DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, FLIPADST_DCT, DCT_FLIPADST, FLIPADST_FLIPADST, ADST_FLIPADST, FLIPADST_ADST, IDTX, V_DCT, H_DCT, V_ADST,
       H_ADST, V_FLIPADST, H_FLIPADST, WHT_WHT };
enum { TX_CLASS_2D, TX_CLASS_HORIZ, TX_CLASS_VERT };
enum { TX_SET_DCTONLY, TX_SET_INTRA_1, TX_SET_INTRA_2, TX_SET_INTER_1, TX_SET_INTER_2, TX_SET_INTER_3 };
// (column kind, row kind) of easyav1_amd_video.h for each type: "ADST_DCT" = ADST down the columns, DCT along the rows
constexpr uint8_t kTxTypeToIr[17] = {EA1_TXTP(0, 0), EA1_TXTP(1, 0), EA1_TXTP(0, 1), EA1_TXTP(1, 1), EA1_TXTP(2, 0), EA1_TXTP(0, 2), EA1_TXTP(2, 2),
                                     EA1_TXTP(1, 2), EA1_TXTP(2, 1), EA1_TXTP(3, 3), EA1_TXTP(0, 3), EA1_TXTP(3, 0), EA1_TXTP(1, 3), EA1_TXTP(3, 1),
                                     EA1_TXTP(2, 3), EA1_TXTP(3, 2), EA1_TXTP_WHT};
// [recalled] Tx_Type_Intra_Inv_Set1 / 2, Tx_Type_Inter_Inv_Set1 / 2 / 3: symbol -> transform type
constexpr uint8_t kIntraInvSet1[7] = {IDTX, DCT_DCT, V_DCT, H_DCT, ADST_ADST, ADST_DCT, DCT_ADST};
constexpr uint8_t kIntraInvSet2[5] = {IDTX, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST};
// (pinned against dav1d 1.5.3 on libaom streams, round 4: DCT_DCT is the EIGHTH symbol of the 16-type set, after the identity and the six 1-D types)
constexpr uint8_t kInterInvSet1[16] = {IDTX, V_DCT, H_DCT, V_ADST, H_ADST, V_FLIPADST, H_FLIPADST, DCT_DCT, ADST_DCT, DCT_ADST, FLIPADST_DCT, DCT_FLIPADST,
                                       ADST_ADST, FLIPADST_FLIPADST, ADST_FLIPADST, FLIPADST_ADST};
constexpr uint8_t kInterInvSet2[12] = {IDTX, V_DCT, H_DCT, DCT_DCT, ADST_DCT, DCT_ADST, FLIPADST_DCT, DCT_FLIPADST, ADST_ADST, FLIPADST_FLIPADST,
                                       ADST_FLIPADST, FLIPADST_ADST};
constexpr uint8_t kInterInvSet3[2] = {IDTX, DCT_DCT};
// [recalled] Mode_To_Txfm: default transform type of a chroma block from its intra mode (UV_CFL last)
constexpr uint8_t kModeToTxfm[14] = {DCT_DCT, ADST_DCT, DCT_ADST, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST, DCT_ADST, ADST_DCT, ADST_ADST, ADST_DCT, DCT_ADST,
                                     ADST_ADST, DCT_DCT};
// [recalled] Filter_Intra_Mode_To_Intra_Dir
constexpr uint8_t kFilterIntraToDir[5] = {EA1_DC_PRED, EA1_V_PRED, EA1_H_PRED, EA1_D157_PRED, EA1_DC_PRED};
// [recalled] Intra_Mode_Context
constexpr uint8_t kIntraModeCtx[13] = {0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0};
// [recalled] Compound_Mode_Ctx_Map
constexpr uint8_t kCompoundModeCtxMap[3][5] = {{0, 1, 1, 1, 1}, {1, 2, 3, 4, 4}, {4, 4, 5, 6, 7}};
// [recalled] Quant_Dist_Weight / Quant_Dist_Lookup (7.11.3.15)
constexpr uint8_t kQuantDistWeight[3][2] = {{2, 3}, {2, 5}, {2, 7}};
constexpr uint8_t kQuantDistLookup[4][2] = {{9, 7}, {11, 5}, {12, 4}, {13, 3}};
// [recalled] restoration coefficient coding (5.11.58): ranges, sub-exponential parameters, start-of-tile references
constexpr int kWienerTapsMin[3] = {-5, -23, -17}, kWienerTapsMax[3] = {10, 8, 46}, kWienerTapsK[3] = {1, 2, 3}, kWienerTapsMid[3] = {3, -7, 15};
constexpr int kSgrXqdMin[2] = {-96, -32}, kSgrXqdMax[2] = {31, 95}, kSgrXqdMid[2] = {-32, 31};
// the sixteen self-guided parameter sets {r0, eps0, r1, eps1} (Sgr_Params; the kernels' own copy of ea1_av1_tables.inc): the syntax
// needs the radii -- a set with radius 0 codes no weight for that pass
constexpr int16_t kSgrParams[16][4] = {EA1_SGR_PARAMS_VALUES};

// ---- syntax elements, for the value policy of a stream generator (a reader ignores them) ----
enum Elem {
    E_PARTITION, E_SPLIT_OR, E_SKIP, E_SKIP_MODE, E_IS_INTER, E_Y_MODE, E_UV_MODE, E_ANGLE_DELTA, E_USE_FILTER_INTRA, E_FILTER_INTRA_MODE, E_CFL_SIGNS,
    E_CFL_ALPHA, E_CDEF_IDX, E_DELTA_Q_ABS, E_DELTA_Q_BITS, E_DELTA_SIGN, E_DELTA_LF_ABS, E_DELTA_LF_BITS, E_TX_DEPTH, E_TXFM_SPLIT, E_TX_TYPE, E_ALL_ZERO,
    E_EOB_PT, E_EOB_EXTRA, E_EOB_EXTRA_BIT, E_COEFF_BASE_EOB, E_COEFF_BASE, E_COEFF_BR, E_DC_SIGN, E_SIGN_BIT, E_GOLOMB_LEN_BIT, E_GOLOMB_DATA_BIT,
    E_COMP_MODE, E_COMP_REF_TYPE, E_REF_BIT, E_NEW_MV, E_ZERO_MV, E_REF_MV, E_COMPOUND_MODE, E_DRL, E_MV_JOINT, E_MV_SIGN, E_MV_CLASS, E_MV_CLASS0_BIT,
    E_MV_FR, E_MV_HP, E_MV_BIT, E_INTERINTRA, E_INTERINTRA_MODE, E_WEDGE_INTERINTRA, E_WEDGE_INDEX, E_WEDGE_SIGN, E_MOTION_MODE, E_USE_OBMC,
    E_COMP_GROUP_IDX, E_COMPOUND_IDX, E_COMPOUND_TYPE, E_MASK_TYPE, E_INTERP_FILTER, E_LR_TYPE, E_LR_BITS, E_LR_SGR_SET, E_SEGMENT_ID, E_SEG_ID_PREDICTED, E_HAS_PALETTE, E_PALETTE_SIZE, E_PALETTE_CACHE_BIT, E_PALETTE_COLOR_BITS, E_PALETTE_COLOR_IDX, E_USE_INTRABC, E_COUNT
};

// ---- what is remembered of every 4x4 unit of the frame (the specification's YModes[][], RefFrames[][], ... arrays): ONE record per block, kept by
//      the tile that decoded it (TileSyntax::recs); the frame-wide unit array (FrameInfo::unit) holds, per 4x4 unit, the stamp of the frame that
//      wrote it and the index of its block's record -- 4 bytes a unit instead of a 32-byte record (at 4K the record array was 16.6 MB, rewritten
//      by every frame).  Transform size and type are per unit, in byte arrays of their own (FrameInfo::utx_size / utx_type). ----
struct Mi {
    uint32_t stamp;          // (unused in the record; the unit array carries the stamp)
    uint8_t size;            // MiSizes
    uint8_t y_mode;          // YModes (intra 0..12, inter NEARESTMV..)
    uint8_t ref[2];          // RefFrames (INTRA_FRAME / 1..7 / REF_NONE)
    int16_t mv[2][2];        // Mvs [list][0 = row, 1 = column]
    uint8_t is_inter, skip, skip_mode, tx_size;   // IsInters, Skips, SkipModes, InterTxSizes (ea1_tx_size_index numbering)
    uint8_t interp[2];       // InterpFilters
    uint8_t comp_group_idx, compound_idx;
    uint8_t tx_type;         // TxTypes (luma)
    uint8_t pad[3];
    uint32_t block;          // index of the block in the frame's list (for the flags of its neighbours)
};

// ---- frame-level inputs of a tile ----
struct FrameInfo {
    const ea1_av1_sequence_header *seq;
    const ea1_av1_frame_header *hdr;
    const Tables *tab;
    int mi_rows, mi_cols, mi_stride;
    uint32_t *unit;              // per 4x4 unit: stamp << 24 | index of the block's record in its tile's list
    uint8_t *utx_size, *utx_type;   // per 4x4 unit: InterTxSizes (ea1_tx_size_index numbering), TxTypes (luma)
    uint32_t stamp;              // 1 .. 255: "written for this frame"
    bool intra_only;             // FrameIsIntra
    bool mono;                   // mono_chrome (NumPlanes == 1): no chroma syntax; the hand-over still lists chroma transform blocks without
                                 // coefficients and DC / inter prediction, which keeps the two chroma planes at 128 (they are not shown)
    int ssx = 1, ssy = 1;        // subsampling_x / subsampling_y of the chroma planes (4:2:0 and 4:0:0: 1, 1; 4:2:2: 1, 0; 4:4:4: 0, 0).  Round 5: the SYNTAX
                                 // follows them; the reconstruction path (hand-over, kernels) is 4:2:0 only and such frames are not handed to it
    uint8_t order_hint[8];       // OrderHints[ref] of the seven references (index 1..7)
    uint8_t sign_bias[8];        // RefFrameSignBias
    uint8_t ref_slot[8];         // the reconstruction context's slot that holds reference 1..7
    bool gm_warp_ok[8];          // the reference has the frame's size (not is_scaled(), 7.11.3.3): only then it can be warped, by the global model (7.11.3.1) or a local one (5.11.27)
    int qidx_class;              // which of the four coefficient CDF sets the frame starts from
    bool reduced_tx_set, lossless;   // lossless: CodedLossless (every segment)
    int tx_mode_select;
    // segmentation (5.11.9 / 5.11.19, 7.? get_segment_id): the map being written, the map of the primary reference (or null)
    uint8_t *seg_ids;
    const uint8_t *prev_seg_ids;
    bool seg_lossless[8];            // LosslessArray
    // motion field of use_ref_frame_mvs (7.9) at 8x8 granularity: {row, column} of the saved vector that was carried to the position and its
    // reference's distance (row = kInvalidMv where none); MotionFieldMvs[ref][..] = mv_projection(vector, mf_ref_to_dst[ref], distance)
    const int16_t *mf_src;
    int8_t mf_ref_to_dst[8];
    // motion this frame leaves for later ones (7.19), written as its blocks are decoded: per 8x8 area the reference and vector of the unit at
    // its odd row and column (the only one 7.9.2 ever projects from); null when nothing is to be saved
    uint8_t *save_ref;
    int16_t *save_mv;
    bool save_past[8];          // the reference lies in the past (only those are kept)
    int mf_w8;
};
constexpr int16_t kInvalidMv = -32768;
// 7.9.3: mv scaled by numerator / denominator (frame distances)
inline void mv_projection(const int16_t *mv, int numerator, int denominator, int16_t *out)
{
    static const int div_mult[32] = {0,    16384, 8192, 5461, 4096, 3276, 2730, 2340, 2048, 1820, 1638, 1489, 1365, 1260, 1170, 1092,
                                     1024, 963,   910,  862,  819,  780,  744,  712,  682,  655,  630,  606,  585,  564,  546,  528};
    const int den = std::min(31, denominator), num = std::min(31, std::max(-31, numerator));
    for (int i = 0; i < 2; i++) {
        const int64_t v = int64_t(mv[i]) * num * div_mult[den];
        const int64_t scaled = v < 0 ? -((-v + (1 << 13)) >> 14) : ((v + (1 << 13)) >> 14);
        out[i] = int16_t(std::min<int64_t>((1 << 14) - 1, std::max<int64_t>(-(1 << 14) + 1, scaled)));
    }
}

// the growing arrays of a parsed frame: trivially copyable records, appended hundreds of thousands of times per frame.  What std::vector costs here
// -- a constructor call per element, a copy of everything when it grows, a zero fill on resize -- this does not: growth is realloc (large blocks move
// by remapping pages), append_uninit hands out raw space.  The subset of the vector interface the parser uses.
template <class T>
struct PodVec {
    static_assert(std::is_trivially_copyable<T>::value, "records only");
    T *p = nullptr;
    size_t n = 0, cap = 0;
    PodVec() = default;
    PodVec(const PodVec &) = delete;
    PodVec &operator=(const PodVec &) = delete;
    PodVec(PodVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr, o.n = o.cap = 0; }
    PodVec &operator=(PodVec &&o) noexcept
    {
        if (this != &o) {
            free(p);
            p = o.p, n = o.n, cap = o.cap;
            o.p = nullptr, o.n = o.cap = 0;
        }
        return *this;
    }
    ~PodVec() { free(p); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T *data() { return p; }
    const T *data() const { return p; }
    T *begin() { return p; }
    T *end() { return p + n; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; }
    void clear() { n = 0; }
    void grow(size_t want)
    {
        size_t c = cap ? cap : 1024;
        while (c < want) c *= 2;
        T *q = static_cast<T *>(realloc(p, c * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p = q, cap = c;
    }
    void reserve(size_t want)
    {
        if (want > cap) grow(want);
    }
    T *append_uninit(size_t k)   // room for k more elements, not initialised
    {
        if (n + k > cap) grow(n + k);
        T *at = p + n;
        n += k;
        return at;
    }
    void push_back(const T &v) { *append_uninit(1) = v; }
    void append(const T *first, size_t k)
    {
        if (k) memcpy(append_uninit(k), first, k * sizeof(T));
    }
    void append_fill(size_t k, const T &v)
    {
        T *at = append_uninit(k);
        for (size_t i = 0; i < k; i++) at[i] = v;
    }
};

// ---- the parsed frame under construction ----
struct FrameOut {
    PodVec<ea1_block> blocks;
    PodVec<ea1_txblock> txs;
    PodVec<int16_t> coefs;
    PodVec<uint8_t> aux;
    std::vector<uint8_t> cdef_idx;           // per 64x64, 0xFF = none
    std::vector<ea1_lr_unit> lr[3];
    int lr_rows[3], lr_cols[3];
    int sb64_cols;
    uint64_t temporal_candidates = 0;   // diagnostics: motion field vectors that reached a block's MV stack (tests: the path is exercised)
    // tiles parsed side by side (av1_frame_driver.h run_tiles): every tile appends to record vectors of its OWN FrameOut, while the arrays that
    // are indexed by position in the frame (cdef_idx, lr: a tile writes the entries of its own superblocks, nobody else's) stay the frame's
    FrameOut *frame = nullptr;
    FrameOut &positional() { return frame ? *frame : *this; }
    const FrameOut &positional() const { return frame ? *frame : *this; }
};

#ifdef EA1_AV1_PROFILE
#include <x86intrin.h>
struct ProfCounters { unsigned long long t[8]; };
inline ProfCounters &prof() { static ProfCounters p = {}; return p; }
struct ProfScope { int k; unsigned long long t0; ProfScope(int k_) : k(k_), t0(__rdtsc()) {} ~ProfScope() { prof().t[k] += __rdtsc() - t0; } };
#define EA1_PROF(k) ProfScope prof_scope_##k(k)
#else
#define EA1_PROF(k)
#endif

// tile-local state + the traversal.  IO: int sym(Elem, uint16_t *cdf, int n [, unsigned allowed]), int bit(Elem), unsigned lit(Elem, int nbits)
template <class IO>
struct TileSyntax {
    IO &io;
    const FrameInfo &f;
    FrameOut &out;
    Cdfs cdf;                       // the tile's adaptive copy
    int row_start, row_end, col_start, col_end;   // MiRowStart ...
    // above / left contexts, indexed by the 4-sample unit inside the tile column / superblock row
    std::vector<uint8_t> above_level[3], above_dc[3];
    uint8_t left_level[3][32], left_dc[3][32];
    // BlockDecoded of the current superblock (luma / chroma), with the -1 row and column at index 0
    uint8_t decoded[2][34][34];
    // deltas
    int current_q_index;
    int delta_lf[4];
    bool read_deltas;
    int8_t cdef_cur[4];             // cdef_idx of the (up to four) 64x64 areas of the current superblock: -1 = not read yet
    int ref_lr_wiener[3][2][3], ref_sgr_xqd[3][2];
    int sb_size4;                   // 16 or 32
    int16_t dq_buf[1024] = {};       // the dequantised transform block under construction (kept zero between blocks)
    int32_t quant_buf[1024] = {};    // its levels (Quant[]), likewise
    // the levels again, clipped to 15, one byte each in rows of kLevStride with four zero columns and rows behind the largest block: the context
    // sums of the coefficient loop read their neighbours without testing for the block's edge (kept zero between blocks, like quant_buf)
    static constexpr int kLevStride = 32 + 4;
    uint8_t lev_buf[kLevStride * (32 + 4) + 4] = {};
    uint8_t lev3_buf[kLevStride * (32 + 4) + 4] = {};   // the same clipped to 3 (the coeff_base context sums five of them)
    std::vector<uint8_t> above_seg_pred;
    uint8_t left_seg_pred[32];
    int segment_id;
    bool lossless;                  // Lossless of the current block (its segment's)
    int err = EA1_OK;
    const char *err_text = nullptr;

    // ---- the current block (the specification's global variables of decode_block) ----
    int mi_row, mi_col, mi_size, bw4, bh4;
    bool has_chroma, avail_u, avail_l, avail_u_chroma, avail_l_chroma;
    int skip, skip_mode, is_inter, y_mode, uv_mode, angle_delta_y, angle_delta_uv, use_filter_intra, filter_intra_mode;
    int cfl_alpha_u, cfl_alpha_v;
    int use_intrabc;
    int pal_size[2];                // PaletteSizeY / PaletteSizeUV
    uint8_t pal_colors[3][8];       // palette_colors_y / _u / _v
    std::vector<uint8_t> color_map[2];   // ColorMapY / ColorMapUV of the block (row-major, the plane's block size)
    int ref_frame[2];
    int16_t mv[2][2];
    int interp_filter[2], motion_mode, interintra, interintra_mode, wedge_interintra, wedge_index, wedge_sign, mask_type;
    int comp_group_idx, compound_idx, compound_type;   // compound_type: EA1_COMP_* (wedge / diff-weighted / average / distance)
    int tx_size;                    // TxSize (index)
    bool uniform_tx;                // every unit of the block has tx_size (read_block_tx_size; else read_var_tx_size wrote the units' own)
    int ref_mv_idx;
    // motion vector prediction
    int num_mv_found, new_mv_count, found_match, close_matches, total_matches;
    int16_t ref_stack_mv[10][2][2];
    int weight_stack[10];
    int16_t global_mvs[2][2];
    int new_mv_context, ref_mv_context, zero_mv_context;
    int16_t pred_mv[2][2];
    int num_samples, num_samples_scanned;
    int cand_list[8][4];
    int32_t local_warp[6];
    bool local_valid;

    TileSyntax(IO &io_, const FrameInfo &f_, FrameOut &out_) : io(io_), f(f_), out(out_) {}

    void fail(int code, const char *what)
    {
        if (err == EA1_OK) {
            err = code;
            err_text = what;
        }
    }
    std::vector<Mi> recs;   // the records of this tile's blocks (FrameInfo::unit indexes it)
    // the record of the block that covers unit (r, c) -- of THIS tile, written for this frame; anything else reads as an empty record (nothing in
    // the syntax may look across a tile's border or at what has not been decoded)
    const Mi &mi_at(int r, int c) const
    {
        static const Mi none{};
        if (!is_inside(r, c)) return none;
        const uint32_t u = f.unit[size_t(r) * f.mi_stride + c];
        return (u >> 24) == f.stamp && (u & 0xFFFFFF) < recs.size() ? recs[u & 0xFFFFFF] : none;
    }
    int tx_size_at(int r, int c) const { return f.utx_size[size_t(r) * f.mi_stride + c]; }
    int tx_type_at(int r, int c) const { return f.utx_type[size_t(r) * f.mi_stride + c]; }
    bool is_inside(int r, int c) const { return c >= col_start && c < col_end && r >= row_start && r < row_end; }
    // the chroma block of the current block in 4-sample units: a luma side of 4 samples that is subsampled belongs to the 8-sample area's chroma block
    // (get_plane_residual_size of the block scaled up to 8, libaom scale_chroma_bsize)
    int chroma_w4() const { return f.ssx ? std::max(bw4, 2) >> 1 : bw4; }
    int chroma_h4() const { return f.ssy ? std::max(bh4, 2) >> 1 : bh4; }
    bool written(int r, int c) const { return (f.unit[size_t(r) * f.mi_stride + c] >> 24) == f.stamp; }

    // ================= tile =================
    void init_tile(const Cdfs &frame_cdfs, int r0, int r1, int c0, int c1)
    {
        cdf = frame_cdfs;
        row_start = r0, row_end = r1, col_start = c0, col_end = c1;
        // a tile's records start on the arenas' alignment (warp matrices: 4 bytes; coefficient runs: 2 entries), so that tiles parsed into arenas of
        // their own and appended in tile order give the very layout of a tile-after-tile parse
        while (out.aux.size() & 3) out.aux.push_back(0);
        if (out.coefs.size() & 1) out.coefs.push_back(0);
        sb_size4 = f.seq->use_128x128_superblock ? 32 : 16;
        const int w4 = c1 - c0 + 32;
        for (int p = 0; p < 3; p++) {
            above_level[p].assign(size_t(w4), 0);
            above_dc[p].assign(size_t(w4), 0);
        }
        above_seg_pred.assign(size_t(w4), 0);
        segment_id = 0;
        lossless = f.lossless;
        current_q_index = f.hdr->base_q_idx;
        for (int i = 0; i < 4; i++) delta_lf[i] = 0;
        for (int p = 0; p < 3; p++) {
            for (int pass = 0; pass < 2; pass++)
                for (int j = 0; j < 3; j++) ref_lr_wiener[p][pass][j] = kWienerTapsMid[j];
            for (int i = 0; i < 2; i++) ref_sgr_xqd[p][i] = kSgrXqdMid[i];
        }
    }

    // decode_tile (5.11.2)
    void decode_tile()
    {
        const int sb_bs = f.seq->use_128x128_superblock ? BLOCK_128X128 : BLOCK_64X64;
        for (int r = row_start; r < row_end && err == EA1_OK; r += sb_size4) {
            memset(left_level, 0, sizeof left_level);
            memset(left_dc, 0, sizeof left_dc);
            memset(left_seg_pred, 0, sizeof left_seg_pred);
            for (int c = col_start; c < col_end && err == EA1_OK; c += sb_size4) {
                read_deltas = f.hdr->delta_q_present;
                for (int i = 0; i < 4; i++) cdef_cur[i] = -1;
                clear_block_decoded_flags(r, c);
                read_lr(r, c, sb_bs);
                decode_partition(r, c, sb_bs);
                store_cdef(r, c);
            }
        }
    }

    // 5.11.3
    void clear_block_decoded_flags(int r, int c)
    {
        for (int pl = 0; pl < 2; pl++) {
            const int sx = pl ? f.ssx : 0, sy = pl ? f.ssy : 0;
            const int sb_w4 = (col_end - c) >> sx, sb_h4 = (row_end - r) >> sy;
            const int nx = sb_size4 >> sx, ny = sb_size4 >> sy;
            // row -1: decoded up to (and not including) column sb_w4; column -1: decoded down to row sb_h4; everything else not yet
            memset(&decoded[pl][0][0], 0, sizeof decoded[pl][0] * size_t(ny + 2));
            memset(&decoded[pl][0][0], 1, size_t(std::min(std::max(sb_w4 + 1, 0), nx + 2)));
            for (int y = 0; y <= ny && y < sb_h4; y++) decoded[pl][y + 1][0] = 1;
            decoded[pl][ny + 1][0] = 0;
        }
    }

    void store_cdef(int r, int c)
    {
        // the 64x64 areas of this superblock that start inside the frame
        const int n = sb_size4 / 16;
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                const int rr = r + 16 * y, cc = c + 16 * x;
                if (rr >= f.mi_rows || cc >= f.mi_cols) continue;
                const int8_t v = cdef_cur[y * 2 + x];
                FrameOut &fo = out.positional();
                fo.cdef_idx[size_t(rr / 16) * fo.sb64_cols + cc / 16] = v < 0 ? 0xFF : uint8_t(v);
            }
    }

    // ================= partition (5.11.4) =================
    uint16_t *partition_cdf(int bsl2, int ctx)
    {
        switch (bsl2) {
        case 3: return cdf.m.Partition8[ctx];
        case 4: return cdf.m.Partition16[ctx];
        case 5: return cdf.m.Partition32[ctx];
        case 6: return cdf.m.Partition64[ctx];
        default: return cdf.m.Partition128[ctx];
        }
    }
    // the block sizes a partition of a square of 2^l2 produces must exist: which partitions a generator may choose (a reader takes
    // what the stream says and checks in decode_block)
    unsigned allowed_partitions(int r, int c, int l2) const
    {
        (void)r, (void)c;   // (blocks may reach beyond the coded frame: the hand-over's frame is padded for them)
        unsigned m = 1u << PARTITION_NONE | 1u << PARTITION_HORZ | 1u << PARTITION_VERT | 1u << PARTITION_SPLIT;
        if (l2 > 3) m |= 1u << PARTITION_HORZ_A | 1u << PARTITION_HORZ_B | 1u << PARTITION_VERT_A | 1u << PARTITION_VERT_B;
        if (l2 > 3 && l2 < 7) m |= 1u << PARTITION_HORZ_4 | 1u << PARTITION_VERT_4;   // (a 128x128 square has no four-way partitions)
        // 4:2:2: a block taller than wide has no chroma block size (get_plane_residual_size == BLOCK_INVALID, AV1 spec 5.11.4: conformance forbids it)
        if (f.ssx && !f.ssy) m &= ~(1u << PARTITION_VERT | 1u << PARTITION_VERT_A | 1u << PARTITION_VERT_B | 1u << PARTITION_VERT_4);
        return m;
    }
    void decode_partition(int r, int c, int bs)
    {
        if (r >= f.mi_rows || c >= f.mi_cols || err != EA1_OK) return;
        const bool au = is_inside(r - 1, c), al = is_inside(r, c - 1);
        const int l2 = kBlkWl2[bs];
        const int n4 = 1 << (l2 - 2), half = n4 >> 1, quarter = half >> 1;
        const bool has_rows = (r + half) < f.mi_rows, has_cols = (c + half) < f.mi_cols;
        int partition;
        if (bs < BLOCK_8X8) {
            partition = PARTITION_NONE;
        } else {
            // context: is the neighbour's block narrower / shorter than this square
            const int above = au && (kBlkWl2[mi_at(r - 1, c).size] < l2), left = al && (kBlkHl2[mi_at(r, c - 1).size] < l2);
            uint16_t *pc = partition_cdf(l2, left * 2 + above);
            if (has_rows && has_cols) {
                partition = io.sym(E_PARTITION, pc, l2 == 3 ? 4 : (l2 == 7 ? 8 : 10), allowed_partitions(r, c, l2));
            } else if (has_cols) {
                // split_or_horz: the probability of every partition that splits the columns
                auto p = [&](int i) { return int(pc[i]) - (i ? int(pc[i - 1]) : 0); };
                int psum = p(PARTITION_VERT) + p(PARTITION_SPLIT);
                if (l2 != 3) psum += p(PARTITION_HORZ_A) + p(PARTITION_VERT_A) + p(PARTITION_VERT_B);
                if (l2 != 3 && l2 != 7) psum += p(PARTITION_VERT_4);
                uint16_t tmp[3] = {uint16_t((1 << 15) - psum), 1 << 15, 0};
                partition = io.sym_noadapt(E_SPLIT_OR, tmp, 2, 3u) ? PARTITION_SPLIT : PARTITION_HORZ;
            } else if (has_rows) {
                auto p = [&](int i) { return int(pc[i]) - (i ? int(pc[i - 1]) : 0); };
                int psum = p(PARTITION_HORZ) + p(PARTITION_SPLIT);
                if (l2 != 3) psum += p(PARTITION_HORZ_A) + p(PARTITION_HORZ_B) + p(PARTITION_VERT_A);
                if (l2 != 3 && l2 != 7) psum += p(PARTITION_HORZ_4);
                uint16_t tmp[3] = {uint16_t((1 << 15) - psum), 1 << 15, 0};
                partition = io.sym_noadapt(E_SPLIT_OR, tmp, 2, (f.ssx && !f.ssy) ? 2u : 3u) ? PARTITION_SPLIT : PARTITION_VERT;   // (4:2:2: no tall blocks, see above)
            } else {
                partition = PARTITION_SPLIT;
            }
        }
        // Partition_Subsize
        auto sub = [&](int wl, int hl) { return block_size_of(wl, hl); };
        const int split_size = l2 > 2 ? sub(l2 - 1, l2 - 1) : BLOCK_INVALID;
        switch (partition) {
        case PARTITION_NONE: decode_block(r, c, bs); break;
        case PARTITION_HORZ:
            decode_block(r, c, sub(l2, l2 - 1));
            if (has_rows) decode_block(r + half, c, sub(l2, l2 - 1));
            break;
        case PARTITION_VERT:
            decode_block(r, c, sub(l2 - 1, l2));
            if (has_cols) decode_block(r, c + half, sub(l2 - 1, l2));
            break;
        case PARTITION_SPLIT:
            decode_partition(r, c, split_size);
            decode_partition(r, c + half, split_size);
            decode_partition(r + half, c, split_size);
            decode_partition(r + half, c + half, split_size);
            break;
        case PARTITION_HORZ_A:
            decode_block(r, c, split_size);
            decode_block(r, c + half, split_size);
            decode_block(r + half, c, sub(l2, l2 - 1));
            break;
        case PARTITION_HORZ_B:
            decode_block(r, c, sub(l2, l2 - 1));
            decode_block(r + half, c, split_size);
            decode_block(r + half, c + half, split_size);
            break;
        case PARTITION_VERT_A:
            decode_block(r, c, split_size);
            decode_block(r + half, c, split_size);
            decode_block(r, c + half, sub(l2 - 1, l2));
            break;
        case PARTITION_VERT_B:
            decode_block(r, c, sub(l2 - 1, l2));
            decode_block(r, c + half, split_size);
            decode_block(r + half, c + half, split_size);
            break;
        case PARTITION_HORZ_4:
            for (int i = 0; i < 4; i++)
                if (i < 3 || r + quarter * 3 < f.mi_rows) decode_block(r + quarter * i, c, sub(l2, l2 - 2));
            break;
        case PARTITION_VERT_4:
            for (int i = 0; i < 4; i++)
                if (i < 3 || c + quarter * 3 < f.mi_cols) decode_block(r, c + quarter * i, sub(l2 - 2, l2));
            break;
        }
    }

    // ================= block (5.11.5) =================
    void decode_block(int r, int c, int bs)
    {
        if (err != EA1_OK) return;
        if (bs == BLOCK_INVALID) return fail(EA1_EINVAL, "a partition that yields no block size");
        if (f.ssx && !f.ssy && kBlkWl2[bs] < kBlkHl2[bs]) return fail(EA1_EINVAL, "a block taller than wide in a 4:2:2 stream (no chroma block size: AV1 spec 5.11.4)");
        mi_row = r, mi_col = c, mi_size = bs;
        bw4 = bw4_of(bs), bh4 = bh4_of(bs);
        has_chroma = (!f.ssy || (r & 1) || !(bh4 & 1)) && (!f.ssx || (c & 1) || !(bw4 & 1));
        avail_u = is_inside(r - 1, c), avail_l = is_inside(r, c - 1);
        avail_u_chroma = avail_u, avail_l_chroma = avail_l;
        if (has_chroma) {
            if (f.ssy && bh4 == 1) avail_u_chroma = is_inside(r - 2, c);
            if (f.ssx && bw4 == 1) avail_l_chroma = is_inside(r, c - 2);
        } else {
            avail_u_chroma = avail_l_chroma = false;
        }
        // defaults of the elements a path may not read
        angle_delta_y = angle_delta_uv = 0, use_filter_intra = 0, filter_intra_mode = 0, cfl_alpha_u = cfl_alpha_v = 0;
        uv_mode = EA1_DC_PRED, interintra = 0, interintra_mode = 0, wedge_interintra = 0, wedge_index = 0, wedge_sign = 0, mask_type = 0;
        motion_mode = MOTION_SIMPLE, comp_group_idx = 0, compound_idx = 1, compound_type = EA1_COMP_AVG, ref_mv_idx = 0;
        interp_filter[0] = interp_filter[1] = 0, skip_mode = 0, local_valid = false, num_samples = 0;
        pal_size[0] = pal_size[1] = 0;
        use_intrabc = 0;
        memset(mv, 0, sizeof mv);
        {
            EA1_PROF(0);
            if (f.intra_only) intra_frame_mode_info();
            else inter_frame_mode_info();
        }
        if (err != EA1_OK) return;
        palette_tokens();
        {
            EA1_PROF(2);
            read_block_tx_size();
        }
        if (skip) reset_block_context();
        // what later blocks (and later frames' contexts) see of this one
        const uint32_t block_index = uint32_t(out.blocks.size());
        EA1_PROF(3);
        const int in_h4 = std::min(bh4, f.mi_rows - r), in_w4 = std::min(bw4, f.mi_cols - c);   // the part inside the coded frame
        {
            // one record for the block; its units get the record's index (and, when the block has ONE transform size, that size: read_var_tx_size
            // wrote the units' own otherwise; tx_type is written by the coefficient reading)
            Mi proto{};
            proto.size = uint8_t(bs);
            proto.y_mode = uint8_t(y_mode);
            proto.ref[0] = uint8_t(ref_frame[0]), proto.ref[1] = uint8_t(ref_frame[1]);
            memcpy(proto.mv, mv, sizeof mv);
            proto.is_inter = uint8_t(is_inter), proto.skip = uint8_t(skip), proto.skip_mode = uint8_t(skip_mode);
            proto.interp[0] = uint8_t(interp_filter[0]), proto.interp[1] = uint8_t(interp_filter[1]);
            proto.comp_group_idx = uint8_t(comp_group_idx), proto.compound_idx = uint8_t(compound_idx);
            proto.block = block_index;
            if (f.save_ref && is_inter) {
                int keep_ref = -1, keep_list = 0;
                for (int list = 0; list < 2; list++) {
                    const int rf = ref_frame[list];
                    if (rf == REF_NONE || rf <= INTRA_FRAME || !f.save_past[rf]) continue;
                    if (std::abs(int(mv[list][0])) > 4095 || std::abs(int(mv[list][1])) > 4095) continue;   // REFMVS_LIMIT
                    keep_ref = rf, keep_list = list;
                }
                if (keep_ref >= 0) {
                    const int w8 = f.mi_cols >> 1;
                    for (int y8 = r >> 1; y8 < (r + in_h4) >> 1; y8++)
                        for (int x8 = c >> 1; x8 < (c + in_w4) >> 1; x8++) {
                            const size_t i = size_t(y8) * w8 + x8;
                            f.save_ref[i] = uint8_t(keep_ref);
                            f.save_mv[2 * i] = mv[keep_list][0], f.save_mv[2 * i + 1] = mv[keep_list][1];
                        }
                }
            }
            if (recs.size() >= 0xFFFFFF) return fail(EA1_EINVAL, "too many blocks in a tile");
            const uint32_t word = (f.stamp << 24) | uint32_t(recs.size());
            recs.push_back(proto);
            for (int y = 0; y < in_h4; y++) {
                const size_t at = size_t(r + y) * f.mi_stride + c;
                std::fill_n(f.unit + at, in_w4, word);
                if (uniform_tx) memset(f.utx_size + at, tx_size, size_t(in_w4));
            }
        }
        if (f.hdr->segmentation_enabled)
            for (int y = 0; y < in_h4; y++) memset(f.seg_ids + size_t(r + y) * f.mi_stride + c, segment_id, size_t(in_w4));
        emit_block();
        {
            EA1_PROF(4);
            residual();
        }
        finish_block();
    }

    // ================= mode info =================
    int skip_ctx() const { return (avail_u ? mi_at(mi_row - 1, mi_col).skip : 0) + (avail_l ? mi_at(mi_row, mi_col - 1).skip : 0); }

    // ---- segmentation ----
    bool seg_feature(int feature) const { return f.hdr->segmentation_enabled && f.hdr->feature_enabled[segment_id][feature]; }
    int seg_data(int feature) const { return f.hdr->feature_data[segment_id][feature]; }
    // get_qidx(ignoreDeltaQ, segment_id)
    int qidx_of_block(bool ignore_delta_q) const
    {
        const ea1_av1_frame_header &h = *f.hdr;
        if (seg_feature(0)) {
            int q = h.base_q_idx + seg_data(0);
            if (!ignore_delta_q && h.delta_q_present) q = current_q_index + seg_data(0);
            return std::min(255, std::max(0, q));
        }
        return (!ignore_delta_q && h.delta_q_present) ? current_q_index : h.base_q_idx;
    }
    static int neg_deinterleave(int diff, int ref, int max)
    {
        if (!ref) return diff;
        if (ref >= max - 1) return max - diff - 1;
        if (2 * ref < max) {
            if (diff <= 2 * ref) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1);
            return diff;
        }
        if (diff <= 2 * (max - ref - 1)) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1);
        return max - (diff + 1);
    }
    // 5.11.9
    void read_segment_id()
    {
        const int stride = f.mi_stride;
        const int prev_ul = (avail_u && avail_l) ? f.seg_ids[size_t(mi_row - 1) * stride + mi_col - 1] : -1;
        const int prev_u = avail_u ? f.seg_ids[size_t(mi_row - 1) * stride + mi_col] : -1;
        const int prev_l = avail_l ? f.seg_ids[size_t(mi_row) * stride + mi_col - 1] : -1;
        int pred;
        if (prev_u == -1) pred = prev_l == -1 ? 0 : prev_l;
        else if (prev_l == -1) pred = prev_u;
        else pred = prev_ul == prev_u ? prev_u : prev_l;
        if (skip) {
            segment_id = pred;
        } else {
            int ctx;
            if (prev_ul < 0) ctx = 0;
            else if (prev_ul == prev_u && prev_ul == prev_l) ctx = 2;
            else if (prev_ul == prev_u || prev_ul == prev_l || prev_u == prev_l) ctx = 1;
            else ctx = 0;
            const int max = f.hdr->last_active_seg_id + 1;
            // a generator may only pick symbols whose mapped id stays within the active segments (a reader clips, as the specification does)
            unsigned allowed = 0;
            for (int v = 0; v < 8; v++) {
                const int id = neg_deinterleave(v, pred, max);
                if (id >= 0 && id < max) allowed |= 1u << v;
            }
            const int v = io.sym(E_SEGMENT_ID, cdf.m.SegmentId[ctx], 8, allowed);
            segment_id = std::min(max - 1, std::max(0, neg_deinterleave(v, pred, max)));
        }
    }
    void intra_segment_id()
    {
        if (f.hdr->segmentation_enabled) read_segment_id();
        else segment_id = 0;
        lossless = f.hdr->segmentation_enabled ? f.seg_lossless[segment_id] : f.lossless;
    }
    int predicted_segment_id() const   // get_segment_id
    {
        if (!f.prev_seg_ids) return 0;
        int seg = 7;
        for (int y = 0; y < std::min(bh4, f.mi_rows - mi_row); y++)
            for (int x = 0; x < std::min(bw4, f.mi_cols - mi_col); x++) seg = std::min<int>(seg, f.prev_seg_ids[size_t(mi_row + y) * f.mi_stride + mi_col + x]);
        return seg;
    }
    void set_seg_pred_context(int v)
    {
        const int x0 = mi_col - col_start, y0 = mi_row & (sb_size4 - 1);
        for (int i = 0; i < bw4; i++) above_seg_pred[size_t(x0 + i)] = uint8_t(v);
        for (int i = 0; i < bh4 && y0 + i < 32; i++) left_seg_pred[y0 + i] = uint8_t(v);
    }
    // 5.11.19
    void inter_segment_id(bool pre_skip)
    {
        const ea1_av1_frame_header &h = *f.hdr;
        if (!h.segmentation_enabled) {
            segment_id = 0;
            return;
        }
        const int predicted = predicted_segment_id();
        if (!h.segmentation_update_map) {
            segment_id = predicted;
            return;
        }
        if (pre_skip && !h.seg_id_pre_skip) {
            segment_id = 0;
            return;
        }
        if (!pre_skip && skip) {
            set_seg_pred_context(0);
            read_segment_id();
            return;
        }
        if (h.segmentation_temporal_update) {
            const int ctx = left_seg_pred[mi_row & (sb_size4 - 1)] + above_seg_pred[size_t(mi_col - col_start)];
            const int pred_flag = io.sym(E_SEG_ID_PREDICTED, cdf.m.SegmentIdPredicted[ctx], 2);
            if (pred_flag) segment_id = predicted;
            else read_segment_id();
            set_seg_pred_context(pred_flag);
        } else {
            read_segment_id();
        }
    }
    void read_skip()
    {
        if (f.hdr->seg_id_pre_skip && seg_feature(6)) skip = 1;   // SEG_LVL_SKIP
        else skip = io.sym(E_SKIP, cdf.m.Skip[skip_ctx()], 2);
    }

    // 5.11.7
    void intra_frame_mode_info()
    {
        skip = 0;
        segment_id = 0;
        if (f.hdr->seg_id_pre_skip) intra_segment_id();
        read_skip();
        if (!f.hdr->seg_id_pre_skip) intra_segment_id();
        read_cdef();
        read_delta_qindex();
        read_delta_lf();
        read_deltas = false;
        ref_frame[0] = INTRA_FRAME, ref_frame[1] = REF_NONE;
        is_inter = 0;
        if (f.hdr->allow_intrabc) {
            // (a generator may only choose it where a legal block vector exists: the source must lie in superblock rows above, inside the tile)
            int16_t target[2];
            const bool can = intrabc_target(target);
            use_intrabc = io.sym(E_USE_INTRABC, cdf.m.Intrabc, 2, can ? 3u : 1u);
            if (use_intrabc) {
                is_inter = 1;
                y_mode = EA1_DC_PRED, uv_mode = EA1_DC_PRED;
                interp_filter[0] = interp_filter[1] = EA1_FILTER_BILINEAR;
                find_mv_stack(false);
                // assign_mv (5.11.26) for intra block copy
                memcpy(pred_mv[0], ref_stack_mv[0][0], 4);
                if (!pred_mv[0][0] && !pred_mv[0][1]) memcpy(pred_mv[0], ref_stack_mv[1][0], 4);
                if (!pred_mv[0][0] && !pred_mv[0][1]) {
                    if (mi_row - sb_size4 < row_start) pred_mv[0][0] = 0, pred_mv[0][1] = int16_t(-(sb_size4 * 4 + 256) * 8);
                    else pred_mv[0][0] = int16_t(-(sb_size4 * 4 * 8)), pred_mv[0][1] = 0;
                }
                read_mv(0, can ? target : nullptr);
                return;
            }
        }
        const int above = kIntraModeCtx[avail_u ? intra_mode_of(mi_at(mi_row - 1, mi_col)) : EA1_DC_PRED];
        const int left = kIntraModeCtx[avail_l ? intra_mode_of(mi_at(mi_row, mi_col - 1)) : EA1_DC_PRED];
        y_mode = io.sym(E_Y_MODE, cdf.m.IntraFrameYMode[above][left], 13);
        intra_tail();
    }
    static int intra_mode_of(const Mi &m) { return m.y_mode < 13 ? int(m.y_mode) : int(EA1_DC_PRED); }
    static bool is_directional(int mode) { return mode >= EA1_V_PRED && mode <= EA1_D67_PRED; }

    // intra_angle_info_y, uv_mode, read_cfl_alphas, intra_angle_info_uv, filter_intra_mode_info (palette: not in this revision)
    void intra_tail()
    {
        if (mi_size >= BLOCK_8X8 && is_directional(y_mode)) angle_delta_y = io.sym(E_ANGLE_DELTA, cdf.m.AngleDelta[y_mode - EA1_V_PRED], 7) - 3;
        if (has_chroma && !f.mono) {
            bool cfl_allowed;
            if (lossless) cfl_allowed = chroma_w4() == 1 && chroma_h4() == 1;   // the plane residual size is 4x4 (4:2:0: luma blocks up to 8x8)
            else cfl_allowed = std::max(bw4, bh4) <= 8;
            uv_mode = cfl_allowed ? io.sym(E_UV_MODE, cdf.m.UVModeCflAllowed[y_mode], 14) : io.sym(E_UV_MODE, cdf.m.UVModeCflNotAllowed[y_mode], 13);
            if (uv_mode == EA1_UV_CFL_PRED) {
                const int signs = io.sym(E_CFL_SIGNS, cdf.m.CflSign, 8);
                const int sign_u = (signs + 1) / 3, sign_v = (signs + 1) % 3;   // 0 zero, 1 negative, 2 positive
                if (sign_u) {
                    const int a = 1 + io.sym(E_CFL_ALPHA, cdf.m.CflAlpha[(sign_u - 1) * 3 + sign_v], 16);
                    cfl_alpha_u = sign_u == 1 ? -a : a;
                }
                if (sign_v) {
                    const int a = 1 + io.sym(E_CFL_ALPHA, cdf.m.CflAlpha[(sign_v - 1) * 3 + sign_u], 16);
                    cfl_alpha_v = sign_v == 1 ? -a : a;
                }
            }
            if (mi_size >= BLOCK_8X8 && is_directional(uv_mode)) angle_delta_uv = io.sym(E_ANGLE_DELTA, cdf.m.AngleDelta[uv_mode - EA1_V_PRED], 7) - 3;
        }
        if (mi_size >= BLOCK_8X8 && bw4 <= 16 && bh4 <= 16 && f.hdr->allow_screen_content_tools) palette_mode_info();
        if (f.seq->enable_filter_intra && y_mode == EA1_DC_PRED && pal_size[0] == 0 && std::max(bw4, bh4) <= 8) {
            use_filter_intra = io.sym(E_USE_FILTER_INTRA, cdf.m.FilterIntra[mi_size], 2);
            if (use_filter_intra) filter_intra_mode = io.sym(E_FILTER_INTRA_MODE, cdf.m.FilterIntraMode, 5);
        }
    }

    // ---- palette (5.11.46 palette_mode_info, 7.11.4 palette cache, 5.11.49 palette_tokens, 8.3.2 colour context) ----
    // the palette of a neighbour is in the records already made: sizes in its block, colours at the head of its aux data
    int neighbour_palette(int r, int c, int plane, const uint8_t **colors) const
    {
        const ea1_block &b = out.blocks[mi_at(r, c).block];
        const int n = b.pal_size[plane];
        if (n) *colors = out.aux.data() + b.aux_off + (plane ? b.pal_size[0] : 0);
        return n;
    }
    int palette_cache(int plane, uint8_t cache[16]) const
    {
        const uint8_t *above = nullptr, *left = nullptr;
        int above_n = 0, left_n = 0;
        if (((mi_row * 4) % 64) && avail_u) above_n = neighbour_palette(mi_row - 1, mi_col, plane, &above);
        if (avail_l) left_n = neighbour_palette(mi_row, mi_col - 1, plane, &left);
        int ai = 0, li = 0, n = 0;
        while (ai < above_n && li < left_n) {
            const int ac = above[ai], lc = left[li];
            if (lc < ac) {
                if (n == 0 || lc != cache[n - 1]) cache[n++] = uint8_t(lc);
                li++;
            } else {
                if (n == 0 || ac != cache[n - 1]) cache[n++] = uint8_t(ac);
                ai++;
                if (lc == ac) li++;
            }
        }
        for (; ai < above_n; ai++)
            if (n == 0 || above[ai] != cache[n - 1]) cache[n++] = above[ai];
        for (; li < left_n; li++)
            if (n == 0 || left[li] != cache[n - 1]) cache[n++] = left[li];
        return n;
    }
    static int ceil_log2(int x)
    {
        if (x < 2) return 0;
        int i = 1, p = 2;
        while (p < x) i++, p <<= 1;
        return i;
    }
    // the colours of one plane that come from the cache, the first literal and the delta-coded rest (luma: deltas + 1; U: plain deltas)
    void read_palette_colors(int plane, int n, uint8_t *colors)
    {
        uint8_t cache[16];
        const int cache_n = palette_cache(plane, cache);
        int idx = 0;
        for (int i = 0; i < cache_n && idx < n; i++)
            if (io.bit(E_PALETTE_CACHE_BIT)) colors[idx++] = cache[i];
        if (idx < n) colors[idx++] = uint8_t(io.lit(E_PALETTE_COLOR_BITS, 8));
        int bits = 0;
        if (idx < n) bits = 8 - 3 + int(io.lit(E_PALETTE_COLOR_BITS, 2));
        while (idx < n) {
            int delta = int(io.lit(E_PALETTE_COLOR_BITS, bits));
            if (plane == 0) delta++;
            const int v = std::min(255, colors[idx - 1] + delta);
            colors[idx] = uint8_t(v);
            const int range = 256 - v - (plane == 0 ? 1 : 0);
            bits = std::min(bits, ceil_log2(range));
            idx++;
        }
        std::sort(colors, colors + n);
    }
    void palette_mode_info()
    {
        const int bsize_ctx = (kBlkWl2[mi_size] - 2) + (kBlkHl2[mi_size] - 2) - 2;
        if (y_mode == EA1_DC_PRED) {
            const uint8_t *dummy;
            const int ctx = (avail_u && neighbour_palette(mi_row - 1, mi_col, 0, &dummy) > 0) + (avail_l && neighbour_palette(mi_row, mi_col - 1, 0, &dummy) > 0);
            // (of the blocks with a 4-sample side only 4x16 and 16x4 get here: they follow BLOCK_8X8 in the enumeration)
            if (io.sym(E_HAS_PALETTE, cdf.m.PaletteYMode[bsize_ctx][ctx], 2)) {
                pal_size[0] = 2 + io.sym(E_PALETTE_SIZE, cdf.m.PaletteYSize[bsize_ctx], 7);
                read_palette_colors(0, pal_size[0], pal_colors[0]);
            }
        }
        if (has_chroma && !f.mono && uv_mode == EA1_DC_PRED) {
            if (io.sym(E_HAS_PALETTE, cdf.m.PaletteUVMode[pal_size[0] > 0], 2)) {
                const int n = pal_size[1] = 2 + io.sym(E_PALETTE_SIZE, cdf.m.PaletteUVSize[bsize_ctx], 7);
                read_palette_colors(1, n, pal_colors[1]);
                if (io.bit(E_PALETTE_CACHE_BIT)) {   // delta_encode_palette_colors_v
                    const int bits = 8 - 4 + int(io.lit(E_PALETTE_COLOR_BITS, 2));
                    pal_colors[2][0] = uint8_t(io.lit(E_PALETTE_COLOR_BITS, 8));
                    for (int idx = 1; idx < n; idx++) {
                        int delta = int(io.lit(E_PALETTE_COLOR_BITS, bits));
                        if (delta && io.bit(E_PALETTE_CACHE_BIT)) delta = -delta;
                        int v = pal_colors[2][idx - 1] + delta;
                        if (v < 0) v += 256;
                        if (v >= 256) v -= 256;
                        pal_colors[2][idx] = uint8_t(v);
                    }
                } else {
                    for (int idx = 0; idx < n; idx++) pal_colors[2][idx] = uint8_t(io.lit(E_PALETTE_COLOR_BITS, 8));
                }
            }
        }
    }
    uint16_t *palette_color_cdf(int uv, int n, int ctx)
    {
        switch (n) {
        case 2: return uv ? cdf.m.PaletteSize2UVColor[ctx] : cdf.m.PaletteSize2YColor[ctx];
        case 3: return uv ? cdf.m.PaletteSize3UVColor[ctx] : cdf.m.PaletteSize3YColor[ctx];
        case 4: return uv ? cdf.m.PaletteSize4UVColor[ctx] : cdf.m.PaletteSize4YColor[ctx];
        case 5: return uv ? cdf.m.PaletteSize5UVColor[ctx] : cdf.m.PaletteSize5YColor[ctx];
        case 6: return uv ? cdf.m.PaletteSize6UVColor[ctx] : cdf.m.PaletteSize6YColor[ctx];
        case 7: return uv ? cdf.m.PaletteSize7UVColor[ctx] : cdf.m.PaletteSize7YColor[ctx];
        default: return uv ? cdf.m.PaletteSize8UVColor[ctx] : cdf.m.PaletteSize8YColor[ctx];
        }
    }
    // the colour index map of one plane: the ON-SCREEN part (ow x oh: what lies inside the coded frame) anti-diagonal by anti-diagonal,
    // the rest of the w x h block repeated from its last column / row
    void read_color_map(int uv, int n, int w, int h, int ow, int oh)
    {
        std::vector<uint8_t> &m = color_map[uv];
        m.assign(size_t(w) * h, 0);
        const int bw_ = w, bh_ = h;
        w = ow, h = oh;
        m[0] = uint8_t(lit_ns(E_PALETTE_COLOR_IDX, n));
        static const int8_t ctx_of_hash[9] = {-1, -1, 0, -1, -1, 4, 3, 2, 1};   // [recalled] Palette_Color_Context
        for (int i = 1; i < h + w - 1; i++)
            for (int j = std::min(i, w - 1); j >= std::max(0, i - h + 1); j--) {
                const int r = i - j, c = j;
                int scores[8] = {0}, order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
                if (c > 0) scores[m[size_t(r) * bw_ + c - 1]] += 2;
                if (r > 0 && c > 0) scores[m[size_t(r - 1) * bw_ + c - 1]] += 1;
                if (r > 0) scores[m[size_t(r - 1) * bw_ + c]] += 2;
                for (int a = 0; a < 3; a++) {
                    int max_score = scores[a], max_idx = a;
                    for (int b = a + 1; b < n; b++)
                        if (scores[b] > max_score) max_score = scores[b], max_idx = b;
                    if (max_idx != a) {
                        const int keep = order[max_idx];
                        for (int k = max_idx; k > a; k--) scores[k] = scores[k - 1], order[k] = order[k - 1];
                        scores[a] = max_score, order[a] = keep;
                    }
                }
                const int hash = scores[0] + 2 * scores[1] + 2 * scores[2];   // Palette_Color_Hash_Multipliers {1, 2, 2}
                const int ctx = hash >= 0 && hash < 9 ? ctx_of_hash[hash] : -1;
                if (ctx < 0) return fail(EA1_EINVAL, "palette colour context");
                m[size_t(r) * bw_ + c] = uint8_t(order[io.sym(E_PALETTE_COLOR_IDX, palette_color_cdf(uv, n, ctx), n)]);
            }
        for (int i = 0; i < oh; i++)
            for (int j = ow; j < bw_; j++) m[size_t(i) * bw_ + j] = m[size_t(i) * bw_ + ow - 1];
        for (int i = oh; i < bh_; i++)
            for (int j = 0; j < bw_; j++) m[size_t(i) * bw_ + j] = m[size_t(oh - 1) * bw_ + j];
    }
    void palette_tokens()
    {
        const int ow = std::min(bw4, f.mi_cols - mi_col) * 4, oh = std::min(bh4, f.mi_rows - mi_row) * 4;
        if (pal_size[0]) read_color_map(0, pal_size[0], bw4 * 4, bh4 * 4, ow, oh);
        if (pal_size[1]) {
            // the chroma block of a 4x16 / 16x4 block covers the whole 8-sample-wide / -high luma area (5.11.49: a width / height below 4
            // grows by 2, on screen as well)
            int cw = (bw4 * 4) >> f.ssx, ch = (bh4 * 4) >> f.ssy, cow = ow >> f.ssx, coh = oh >> f.ssy;
            if (cw < 4) cw += 2, cow += 2;
            if (ch < 4) ch += 2, coh += 2;
            read_color_map(1, pal_size[1], cw, ch, cow, coh);
        }
    }

    // 5.11.56
    void read_cdef()
    {
        if (skip || f.hdr->coded_lossless || !f.seq->enable_cdef || f.hdr->allow_intrabc) return;
        const int i = ((mi_row & (sb_size4 - 1)) >= 16 ? 2 : 0) + ((mi_col & (sb_size4 - 1)) >= 16 ? 1 : 0);
        if (cdef_cur[i] == -1) {
            const int v = int(io.lit(E_CDEF_IDX, f.hdr->cdef_bits));
            // a block wider / taller than 64 samples sets the other 64x64 areas it covers too
            cdef_cur[i] = int8_t(v);
            if (bw4 > 16) cdef_cur[i | 1] = int8_t(v);
            if (bh4 > 16) cdef_cur[i | 2] = int8_t(v);
            if (bw4 > 16 && bh4 > 16) cdef_cur[3] = int8_t(v);
        }
    }
    // 5.11.57
    void read_delta_qindex()
    {
        const int sb_bs = f.seq->use_128x128_superblock ? BLOCK_128X128 : BLOCK_64X64;
        if (mi_size == sb_bs && skip) return;
        if (!read_deltas) return;
        int a = io.sym(E_DELTA_Q_ABS, cdf.m.DeltaQ, 4);
        if (a == 3) {
            const int rem = int(io.lit(E_DELTA_Q_BITS, 3)) + 1;
            a = int(io.lit(E_DELTA_Q_BITS, rem)) + (1 << rem) + 1;
        }
        if (a) {
            const int reduced = io.bit(E_DELTA_SIGN) ? -a : a;
            current_q_index = std::min(255, std::max(1, current_q_index + reduced * (1 << f.hdr->delta_q_res)));
        }
    }
    // 5.11.58
    void read_delta_lf()
    {
        const int sb_bs = f.seq->use_128x128_superblock ? BLOCK_128X128 : BLOCK_64X64;
        if (mi_size == sb_bs && skip) return;
        if (!(read_deltas && f.hdr->delta_lf_present)) return;
        const int count = f.hdr->delta_lf_multi ? (f.mono ? 2 : 4) : 1;   // FRAME_LF_COUNT, two less without chroma
        for (int i = 0; i < count; i++) {
            int a = io.sym(E_DELTA_LF_ABS, f.hdr->delta_lf_multi ? cdf.m.DeltaLFMulti[i] : cdf.m.DeltaLF, 4);
            if (a == 3) {
                const int n = int(io.lit(E_DELTA_LF_BITS, 3)) + 1;
                a = int(io.lit(E_DELTA_LF_BITS, n)) + (1 << n) + 1;
            }
            if (a) {
                const int reduced = io.bit(E_DELTA_SIGN) ? -a : a;
                delta_lf[i] = std::min(63, std::max(-63, delta_lf[i] + reduced * (1 << f.hdr->delta_lf_res)));
            }
        }
    }

    // 5.11.18
    void inter_frame_mode_info()
    {
        skip = 0;
        inter_segment_id(true);
        // skip mode needs a block of at least 8x8 and no segment feature that fixes the reference or the motion
        if (f.hdr->skip_mode_present && std::min(bw4, bh4) >= 2 && !seg_feature(6) && !seg_feature(5) && !seg_feature(7)) {
            const int ctx = (avail_u ? mi_at(mi_row - 1, mi_col).skip_mode : 0) + (avail_l ? mi_at(mi_row, mi_col - 1).skip_mode : 0);
            skip_mode = io.sym(E_SKIP_MODE, cdf.m.SkipMode[ctx], 2);
        }
        if (skip_mode) skip = 1;
        else read_skip();
        if (!f.hdr->seg_id_pre_skip) inter_segment_id(false);
        lossless = f.hdr->segmentation_enabled ? f.seg_lossless[segment_id] : f.lossless;
        read_cdef();
        read_delta_qindex();
        read_delta_lf();
        read_deltas = false;
        if (skip_mode) {
            is_inter = 1;
        } else if (seg_feature(5)) {       // SEG_LVL_REF_FRAME
            is_inter = seg_data(5) != INTRA_FRAME;
        } else if (seg_feature(7)) {       // SEG_LVL_GLOBALMV
            is_inter = 1;
        } else {
            const bool ai = avail_u && !mi_at(mi_row - 1, mi_col).is_inter, li = avail_l && !mi_at(mi_row, mi_col - 1).is_inter;   // AboveIntra / LeftIntra
            int ctx;
            if (avail_u && avail_l) ctx = (li && ai) ? 3 : (li || ai);
            else if (avail_u || avail_l) ctx = 2 * (avail_u ? ai : li);
            else ctx = 0;
            is_inter = io.sym(E_IS_INTER, cdf.m.IsInter[ctx], 2);
        }
        if (is_inter) {
            inter_block_mode_info();
        } else {
            ref_frame[0] = INTRA_FRAME, ref_frame[1] = REF_NONE;
            y_mode = io.sym(E_Y_MODE, cdf.m.YMode[kSizeGroup[mi_size]], 13);
            intra_tail();
        }
    }

    // ---- neighbours' references for the contexts of 8.3.2 ----
    struct Neigh {
        bool avail, intra, single;
        int ref0, ref1;
    };
    Neigh neigh(bool avail, int r, int c) const
    {
        Neigh n{avail, false, true, INTRA_FRAME, REF_NONE};
        if (avail) {
            const Mi &m = mi_at(r, c);
            n.ref0 = m.ref[0], n.ref1 = m.ref[1];
            n.intra = n.ref0 <= INTRA_FRAME;
            n.single = n.ref1 == REF_NONE || n.ref1 <= INTRA_FRAME;
        }
        return n;
    }
    static int count_ctx(int a, int b) { return a < b ? 0 : (a == b ? 1 : 2); }
    static bool is_backward(int ref) { return ref >= BWDREF_FRAME && ref <= ALTREF_FRAME; }
    static bool samedir_pair(int r0, int r1)
    {
        if (r0 <= INTRA_FRAME || r1 <= INTRA_FRAME || r0 == REF_NONE || r1 == REF_NONE) return false;
        return (r0 >= BWDREF_FRAME) == (r1 >= BWDREF_FRAME);
    }

    // 5.11.25
    void read_ref_frames()
    {
        if (skip_mode) {
            ref_frame[0] = f.hdr->skip_mode_frame[0], ref_frame[1] = f.hdr->skip_mode_frame[1];
            return;
        }
        if (seg_feature(5)) {
            ref_frame[0] = seg_data(5), ref_frame[1] = REF_NONE;
            return;
        }
        if (seg_feature(6) || seg_feature(7)) {
            ref_frame[0] = LAST_FRAME, ref_frame[1] = REF_NONE;
            return;
        }
        const Neigh a = neigh(avail_u, mi_row - 1, mi_col), l = neigh(avail_l, mi_row, mi_col - 1);
        // how often each reference occurs in the two neighbours
        int cnt[8] = {0};
        for (const Neigh *n : {&a, &l})
            if (n->avail) {
                if (n->ref0 > INTRA_FRAME && n->ref0 <= ALTREF_FRAME) cnt[n->ref0]++;
                if (n->ref1 > INTRA_FRAME && n->ref1 <= ALTREF_FRAME) cnt[n->ref1]++;
            }
        const int fwd = cnt[LAST_FRAME] + cnt[LAST2_FRAME] + cnt[LAST3_FRAME] + cnt[GOLDEN_FRAME];
        const int bwd = cnt[BWDREF_FRAME] + cnt[ALTREF2_FRAME] + cnt[ALTREF_FRAME];
        const int ctx_p1 = count_ctx(fwd, bwd);
        const int ctx_p2 = count_ctx(cnt[BWDREF_FRAME] + cnt[ALTREF2_FRAME], cnt[ALTREF_FRAME]);
        const int ctx_p3 = count_ctx(cnt[LAST_FRAME] + cnt[LAST2_FRAME], cnt[LAST3_FRAME] + cnt[GOLDEN_FRAME]);
        const int ctx_p4 = count_ctx(cnt[LAST_FRAME], cnt[LAST2_FRAME]);
        const int ctx_p5 = count_ctx(cnt[LAST3_FRAME], cnt[GOLDEN_FRAME]);
        const int ctx_p6 = count_ctx(cnt[BWDREF_FRAME], cnt[ALTREF2_FRAME]);
        int comp_mode = 0;
        if (f.hdr->reference_select && std::min(bw4, bh4) >= 2) {
            int ctx;
            if (a.avail && l.avail) {
                if (a.single && l.single) ctx = is_backward(a.ref0) ^ is_backward(l.ref0);
                else if (a.single) ctx = 2 + (is_backward(a.ref0) || a.intra);
                else if (l.single) ctx = 2 + (is_backward(l.ref0) || l.intra);
                else ctx = 4;
            } else if (a.avail) {
                ctx = a.single ? is_backward(a.ref0) : 3;
            } else if (l.avail) {
                ctx = l.single ? is_backward(l.ref0) : 3;
            } else {
                ctx = 1;
            }
            comp_mode = io.sym(E_COMP_MODE, cdf.m.CompMode[ctx], 2);
        }
        if (comp_mode) {
            const bool a_comp = a.avail && !a.intra && !a.single, l_comp = l.avail && !l.intra && !l.single;
            const bool a_uni = a_comp && samedir_pair(a.ref0, a.ref1), l_uni = l_comp && samedir_pair(l.ref0, l.ref1);
            int ctx;
            if (a.avail && !a.intra && l.avail && !l.intra) {
                const int samedir = samedir_pair(a.ref0, l.ref0);
                if (!a_comp && !l_comp) ctx = 1 + 2 * samedir;
                else if (!a_comp) ctx = !l_uni ? 1 : 3 + samedir;
                else if (!l_comp) ctx = !a_uni ? 1 : 3 + samedir;
                else if (!a_uni && !l_uni) ctx = 0;
                else if (!a_uni || !l_uni) ctx = 2;
                else ctx = 3 + ((a.ref0 == BWDREF_FRAME) == (l.ref0 == BWDREF_FRAME));
            } else if (a.avail && l.avail) {
                if (a_comp) ctx = 1 + 2 * a_uni;
                else if (l_comp) ctx = 1 + 2 * l_uni;
                else ctx = 2;
            } else if (a_comp) {
                ctx = 4 * a_uni;
            } else if (l_comp) {
                ctx = 4 * l_uni;
            } else {
                ctx = 2;
            }
            const int comp_ref_type = io.sym(E_COMP_REF_TYPE, cdf.m.CompRefType[ctx], 2);
            if (comp_ref_type == 0) {   // UNIDIR_COMP_REFERENCE
                if (io.sym(E_REF_BIT, cdf.m.UniCompRef[ctx_p1][0], 2)) {
                    ref_frame[0] = BWDREF_FRAME, ref_frame[1] = ALTREF_FRAME;
                } else if (io.sym(E_REF_BIT, cdf.m.UniCompRef[count_ctx(cnt[LAST2_FRAME], cnt[LAST3_FRAME] + cnt[GOLDEN_FRAME])][1], 2)) {
                    ref_frame[0] = LAST_FRAME;
                    ref_frame[1] = io.sym(E_REF_BIT, cdf.m.UniCompRef[ctx_p5][2], 2) ? GOLDEN_FRAME : LAST3_FRAME;
                } else {
                    ref_frame[0] = LAST_FRAME, ref_frame[1] = LAST2_FRAME;
                }
            } else {
                if (io.sym(E_REF_BIT, cdf.m.CompRef[ctx_p3][0], 2) == 0) ref_frame[0] = io.sym(E_REF_BIT, cdf.m.CompRef[ctx_p4][1], 2) ? LAST2_FRAME : LAST_FRAME;
                else ref_frame[0] = io.sym(E_REF_BIT, cdf.m.CompRef[ctx_p5][2], 2) ? GOLDEN_FRAME : LAST3_FRAME;
                if (io.sym(E_REF_BIT, cdf.m.CompBwdRef[ctx_p2][0], 2) == 0) ref_frame[1] = io.sym(E_REF_BIT, cdf.m.CompBwdRef[ctx_p6][1], 2) ? ALTREF2_FRAME : BWDREF_FRAME;
                else ref_frame[1] = ALTREF_FRAME;
            }
        } else {
            if (io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p1][0], 2)) {
                if (io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p2][1], 2) == 0) ref_frame[0] = io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p6][5], 2) ? ALTREF2_FRAME : BWDREF_FRAME;
                else ref_frame[0] = ALTREF_FRAME;
            } else if (io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p3][2], 2)) {
                ref_frame[0] = io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p5][4], 2) ? GOLDEN_FRAME : LAST3_FRAME;
            } else {
                ref_frame[0] = io.sym(E_REF_BIT, cdf.m.SingleRef[ctx_p4][3], 2) ? LAST2_FRAME : LAST_FRAME;
            }
            ref_frame[1] = REF_NONE;
        }
    }

    static bool has_newmv(int m) { return m == NEWMV || m == NEW_NEWMV || m == NEAR_NEWMV || m == NEW_NEARMV || m == NEAREST_NEWMV || m == NEW_NEARESTMV; }
    static bool has_nearmv(int m) { return m == NEARMV || m == NEAR_NEARMV || m == NEAR_NEWMV || m == NEW_NEARMV; }
    int gm_type(int ref) const { return f.hdr->gm_type[ref - LAST_FRAME]; }

    // 5.11.23
    void inter_block_mode_info()
    {
        read_ref_frames();
        const bool is_compound = ref_frame[1] != REF_NONE && ref_frame[1] > INTRA_FRAME;
        find_mv_stack(is_compound);
        if (skip_mode) {
            y_mode = NEAREST_NEARESTMV;
        } else if (seg_feature(6) || seg_feature(7)) {
            y_mode = GLOBALMV;
        } else if (is_compound) {
            const int ctx = kCompoundModeCtxMap[ref_mv_context >> 1][std::min(new_mv_context, 4)];
            y_mode = NEAREST_NEARESTMV + io.sym(E_COMPOUND_MODE, cdf.m.CompoundMode[ctx], 8);
        } else {
            if (io.sym(E_NEW_MV, cdf.m.NewMv[new_mv_context], 2) == 0) y_mode = NEWMV;
            else if (io.sym(E_ZERO_MV, cdf.m.ZeroMv[zero_mv_context], 2) == 0) y_mode = GLOBALMV;
            else y_mode = io.sym(E_REF_MV, cdf.m.RefMv[ref_mv_context], 2) == 0 ? NEARESTMV : NEARMV;
        }
        ref_mv_idx = 0;
        if (y_mode == NEWMV || y_mode == NEW_NEWMV) {
            for (int idx = 0; idx < 2; idx++)
                if (num_mv_found > idx + 1) {
                    if (io.sym(E_DRL, cdf.m.DrlMode[drl_ctx(idx)], 2) == 0) {
                        ref_mv_idx = idx;
                        break;
                    }
                    ref_mv_idx = idx + 1;
                }
        } else if (has_nearmv(y_mode)) {
            ref_mv_idx = 1;
            for (int idx = 1; idx < 3; idx++)
                if (num_mv_found > idx + 1) {
                    if (io.sym(E_DRL, cdf.m.DrlMode[drl_ctx(idx)], 2) == 0) {
                        ref_mv_idx = idx;
                        break;
                    }
                    ref_mv_idx = idx + 1;
                }
        }
        assign_mv(is_compound);
        read_interintra_mode(is_compound);
        read_motion_mode(is_compound);
        read_compound_type(is_compound);
        if (f.hdr->interpolation_filter == INTERP_SWITCHABLE) {
            const int dirs = f.seq->enable_dual_filter ? 2 : 1;
            for (int dir = 0; dir < dirs; dir++) {
                if (needs_interp_filter()) {
                    int ctx = ((dir & 1) * 2 + (is_compound ? 1 : 0)) * 4;
                    int left_type = 3, above_type = 3;
                    if (avail_l) {
                        const Mi &m = mi_at(mi_row, mi_col - 1);
                        if (m.ref[0] == ref_frame[0] || m.ref[1] == ref_frame[0]) left_type = m.interp[dir];
                    }
                    if (avail_u) {
                        const Mi &m = mi_at(mi_row - 1, mi_col);
                        if (m.ref[0] == ref_frame[0] || m.ref[1] == ref_frame[0]) above_type = m.interp[dir];
                    }
                    if (left_type == above_type) ctx += left_type;
                    else if (left_type == 3) ctx += above_type;
                    else if (above_type == 3) ctx += left_type;
                    else ctx += 3;
                    interp_filter[dir] = io.sym(E_INTERP_FILTER, cdf.m.InterpFilter[ctx], 3);
                } else {
                    interp_filter[dir] = 0;
                }
            }
            if (!f.seq->enable_dual_filter) interp_filter[1] = interp_filter[0];
        } else {
            interp_filter[0] = interp_filter[1] = f.hdr->interpolation_filter;
        }
    }
    int drl_ctx(int idx) const
    {
        const int w0 = weight_stack[idx], w1 = weight_stack[idx + 1];
        if (w0 >= 640 && w1 >= 640) return 0;
        if (w0 >= 640) return 1;
        return 2;
    }
    bool needs_interp_filter() const
    {
        const bool large = std::min(bw4, bh4) >= 2;
        if (skip_mode || motion_mode == MOTION_LOCALWARP) return false;
        if (large && y_mode == GLOBALMV) return gm_type(ref_frame[0]) == EA1_GM_TRANSLATION;
        // (either reference: libaom is_nontrans_global_motion loops over both, dav1d ORs both into has_subpel_filter.  Until round 5 only RefFrame[1] was
        // looked at -- invisible to the round trip and to libaom's streams, whose compound global pairs share one type; found beside dav1d on generated streams)
        if (large && y_mode == GLOBAL_GLOBALMV) return gm_type(ref_frame[0]) == EA1_GM_TRANSLATION || gm_type(ref_frame[1]) == EA1_GM_TRANSLATION;
        return true;
    }
    static int get_mode(int y_mode, int list)
    {
        if (list == 0) {
            if (y_mode < NEAREST_NEARESTMV) return y_mode;
            if (y_mode == NEW_NEWMV || y_mode == NEW_NEARESTMV || y_mode == NEW_NEARMV) return NEWMV;
            if (y_mode == NEAREST_NEARESTMV || y_mode == NEAREST_NEWMV) return NEARESTMV;
            if (y_mode == NEAR_NEARMV || y_mode == NEAR_NEWMV) return NEARMV;
            return GLOBALMV;
        }
        if (y_mode == NEW_NEWMV || y_mode == NEAREST_NEWMV || y_mode == NEAR_NEWMV) return NEWMV;
        if (y_mode == NEAREST_NEARESTMV || y_mode == NEW_NEARESTMV) return NEARESTMV;
        if (y_mode == NEAR_NEARMV || y_mode == NEW_NEARMV) return NEARMV;
        return GLOBALMV;
    }
    // 5.11.26
    void assign_mv(bool is_compound)
    {
        for (int i = 0; i < 1 + int(is_compound); i++) {
            const int comp_mode = get_mode(y_mode, i);
            if (comp_mode == GLOBALMV) {
                memcpy(pred_mv[i], global_mvs[i], 4);
            } else {
                int pos = comp_mode == NEARESTMV ? 0 : ref_mv_idx;
                if (comp_mode == NEWMV && num_mv_found <= 1) pos = 0;
                memcpy(pred_mv[i], ref_stack_mv[pos][i], 4);
            }
            if (comp_mode == NEWMV) read_mv(i);
            else memcpy(mv[i], pred_mv[i], 4);
        }
    }
    // A block vector a generator may write for the current block: the source rectangle lies in the superblock rows above this one, inside
    // the tile and the visible frame (rows above are complete, which is all AV1's and the hand-over's rules ask of them).  The choice is
    // drawn from the position only (no state of its own): a reader calls this too and ignores the result.
    bool intrabc_target(int16_t target[2]) const
    {
        // (a block with a 4-sample side: the chroma block it may own belongs to the whole 8-sample area and moves by the same vector)
        const int w = std::max(bw4, 2) * 4, h = std::max(bh4, 2) * 4, x = (bw4 == 1 ? mi_col & ~1 : mi_col) * 4, y = (bh4 == 1 ? mi_row & ~1 : mi_row) * 4;
        const int sb_y0 = (mi_row / sb_size4) * sb_size4 * 4;
        const int x_lo = col_start * 4, x_hi = std::min(col_end * 4, int(f.hdr->frame_width)) - w - 2;
        const int y_lo = row_start * 4, y_hi = std::min(sb_y0, int(f.hdr->frame_height)) - h - 2;
        if (x_hi < x_lo || y_hi < y_lo) return false;
        // a position-dependent pseudo-random pick on the 8-sample grid
        const uint32_t hsh = uint32_t(mi_row) * 2654435761u ^ uint32_t(mi_col) * 40503u ^ f.stamp * 97u;
        const int sx = x_lo + int((hsh >> 3) % uint32_t((x_hi - x_lo) / 8 + 1)) * 8, sy = y_lo + int((hsh >> 17) % uint32_t((y_hi - y_lo) / 8 + 1)) * 8;
        target[0] = int16_t((sy - y) * 8), target[1] = int16_t((sx - x) * 8);
        return true;
    }
    // 5.11.32; `forced`: the vector a generator must arrive at (its symbols are then determined: passed as one-value masks)
    void read_mv(int ref, const int16_t *forced = nullptr)
    {
        int diff[2] = {0, 0};
        const int ctx = use_intrabc ? 1 : 0;   // MvCtx
        int want[2] = {0, 0};
        unsigned jmask = ~0u;
        if (forced) {
            want[0] = forced[0] - pred_mv[ref][0], want[1] = forced[1] - pred_mv[ref][1];
            jmask = 1u << ((want[0] ? 2 : 0) | (want[1] ? 1 : 0));
        }
        const int joint = io.sym(E_MV_JOINT, cdf.m.MvJoint[ctx], 4, jmask);
        if (joint == 2 || joint == 3) diff[0] = read_mv_component(ctx, 0, forced ? &want[0] : nullptr);
        if (joint == 1 || joint == 3) diff[1] = read_mv_component(ctx, 1, forced ? &want[1] : nullptr);
        mv[ref][0] = int16_t(pred_mv[ref][0] + diff[0]);
        mv[ref][1] = int16_t(pred_mv[ref][1] + diff[1]);
    }
    int read_mv_component(int ctx, int comp, const int *want = nullptr)
    {
        // a wanted value (whole samples: a multiple of 8, non-zero) fixes sign, class and integer bits
        unsigned m_sign = ~0u, m_class = ~0u, m_bit0 = ~0u;
        int want_d = 0;
        if (want && std::abs(*want) >= 8) {   // (a reader gets here with a target it does not follow: anything it computes is ignored)
            const int m8 = std::abs(*want) / 8;   // 1 ..
            int cls = 0;
            if (m8 > 2) {
                while ((2 << cls) < m8) cls++;      // m8 in (2^cls, 2^(cls+1)]
                want_d = m8 - (1 << cls) - 1;
            }
            m_sign = 1u << (*want < 0), m_class = 1u << cls, m_bit0 = 1u << (m8 - 1 > 1 ? 0 : m8 - 1);
        }
        const int sign = io.sym(E_MV_SIGN, cdf.m.MvSign[ctx][comp], 2, m_sign);
        const int cls = io.sym(E_MV_CLASS, cdf.m.MvClass[ctx][comp], 11, m_class);
        const bool integer = f.hdr->force_integer_mv, hp_allowed = f.hdr->allow_high_precision_mv;
        int mag;
        if (cls == 0) {
            const int b = io.sym(E_MV_CLASS0_BIT, cdf.m.MvClass0Bit[ctx][comp], 2, m_bit0);
            const int fr = integer ? 3 : io.sym(E_MV_FR, cdf.m.MvClass0Fr[ctx][comp][b], 4);
            const int hp = hp_allowed ? io.sym(E_MV_HP, cdf.m.MvClass0Hp[ctx][comp], 2) : 1;
            mag = ((b << 3) | (fr << 1) | hp) + 1;
        } else {
            int d = 0;
            for (int i = 0; i < cls; i++) d |= io.sym(E_MV_BIT, cdf.m.MvBit[ctx][comp][i], 2, m_class != ~0u ? 1u << ((want_d >> i) & 1) : ~0u) << i;
            mag = 2 << (cls + 2);
            const int fr = integer ? 3 : io.sym(E_MV_FR, cdf.m.MvFr[ctx][comp], 4);
            const int hp = hp_allowed ? io.sym(E_MV_HP, cdf.m.MvHp[ctx][comp], 2) : 1;
            mag += ((d << 3) | (fr << 1) | hp) + 1;
        }
        return sign ? -mag : mag;
    }

    // 5.11.28
    void read_interintra_mode(bool is_compound)
    {
        interintra = 0;
        if (!skip_mode && f.seq->enable_interintra_compound && !is_compound && mi_size >= BLOCK_8X8 && mi_size <= BLOCK_32X32) {
            const int ctx = kSizeGroup[mi_size] - 1;
            interintra = io.sym(E_INTERINTRA, cdf.m.InterIntra[ctx], 2);
            if (interintra) {
                interintra_mode = io.sym(E_INTERINTRA_MODE, cdf.m.InterIntraMode[ctx], 4);
                ref_frame[1] = INTRA_FRAME;
                wedge_interintra = io.sym(E_WEDGE_INTERINTRA, cdf.m.WedgeInterIntra[mi_size], 2);
                if (wedge_interintra) wedge_index = io.sym(E_WEDGE_INDEX, cdf.m.WedgeIndex[mi_size], 16);
            }
        }
    }
    bool has_overlappable_candidates() const
    {
        if (avail_u)
            for (int x4 = mi_col; x4 < std::min(f.mi_cols, mi_col + bw4); x4 += 2) {
                int x5 = x4 | 1;
                if (x5 >= f.mi_cols) x5 = f.mi_cols - 1;
                const Mi &m = mi_at(mi_row - 1, x5);
                if (m.ref[0] > INTRA_FRAME && m.ref[0] != REF_NONE) return true;
            }
        if (avail_l)
            for (int y4 = mi_row; y4 < std::min(f.mi_rows, mi_row + bh4); y4 += 2) {
                int y5 = y4 | 1;
                if (y5 >= f.mi_rows) y5 = f.mi_rows - 1;
                const Mi &m = mi_at(y5, mi_col - 1);
                if (m.ref[0] > INTRA_FRAME && m.ref[0] != REF_NONE) return true;
            }
        return false;
    }
    // 5.11.27
    void read_motion_mode(bool is_compound)
    {
        motion_mode = MOTION_SIMPLE;
        if (skip_mode || !f.hdr->is_motion_mode_switchable || std::min(bw4, bh4) < 2) return;
        if (!f.hdr->force_integer_mv && (y_mode == GLOBALMV || y_mode == GLOBAL_GLOBALMV) && gm_type(ref_frame[0]) > EA1_GM_TRANSLATION) return;
        if (is_compound || ref_frame[1] == INTRA_FRAME || !has_overlappable_candidates()) return;
        find_warp_samples();
        // (a reference of another size -- is_scaled() -- cannot be warped: the three-way symbol is not coded for it)
        if (f.hdr->force_integer_mv || num_samples == 0 || !f.hdr->allow_warped_motion || !f.gm_warp_ok[ref_frame[0]]) {
            motion_mode = io.sym(E_USE_OBMC, cdf.m.UseObmc[mi_size], 2) ? MOTION_OBMC : MOTION_SIMPLE;
        } else {
            motion_mode = io.sym(E_MOTION_MODE, cdf.m.MotionMode[mi_size], 3);
            if (motion_mode == MOTION_LOCALWARP) warp_estimation();
        }
    }
    int relative_dist(int a, int b) const
    {
        if (!f.seq->enable_order_hint) return 0;
        const int bits = f.seq->order_hint_bits;
        int diff = a - b;
        const int m = 1 << (bits - 1);
        diff = (diff & (m - 1)) - (diff & m);
        return diff;
    }
    // 5.11.29
    void read_compound_type(bool is_compound)
    {
        comp_group_idx = 0, compound_idx = 1;
        compound_type = EA1_COMP_AVG;
        if (skip_mode) return;
        if (is_compound) {
            const Neigh a = neigh(avail_u, mi_row - 1, mi_col), l = neigh(avail_l, mi_row, mi_col - 1);
            if (f.seq->enable_masked_compound) {
                int ctx = 0;
                if (a.avail) {
                    if (!a.single) ctx += mi_at(mi_row - 1, mi_col).comp_group_idx;
                    else if (a.ref0 == ALTREF_FRAME) ctx += 3;
                }
                if (l.avail) {
                    if (!l.single) ctx += mi_at(mi_row, mi_col - 1).comp_group_idx;
                    else if (l.ref0 == ALTREF_FRAME) ctx += 3;
                }
                comp_group_idx = io.sym(E_COMP_GROUP_IDX, cdf.m.CompGroupIdx[std::min(5, ctx)], 2);
            }
            if (comp_group_idx == 0) {
                if (f.seq->enable_jnt_comp) {
                    const int fwd = std::abs(relative_dist(f.order_hint[ref_frame[0]], f.hdr->order_hint));
                    const int bck = std::abs(relative_dist(f.order_hint[ref_frame[1]], f.hdr->order_hint));
                    int ctx = fwd == bck ? 3 : 0;
                    if (a.avail) {
                        if (!a.single) ctx += mi_at(mi_row - 1, mi_col).compound_idx;
                        else if (a.ref0 == ALTREF_FRAME) ctx++;
                    }
                    if (l.avail) {
                        if (!l.single) ctx += mi_at(mi_row, mi_col - 1).compound_idx;
                        else if (l.ref0 == ALTREF_FRAME) ctx++;
                    }
                    compound_idx = io.sym(E_COMPOUND_IDX, cdf.m.CompoundIdx[ctx], 2);
                    compound_type = compound_idx ? EA1_COMP_AVG : EA1_COMP_DIST;
                }
            } else {
                if (kWedgeBits[mi_size] == 0) compound_type = EA1_COMP_DIFFWTD;
                else compound_type = io.sym(E_COMPOUND_TYPE, cdf.m.CompoundType[mi_size], 2) ? EA1_COMP_DIFFWTD : EA1_COMP_WEDGE;
            }
            if (compound_type == EA1_COMP_WEDGE) {
                wedge_index = io.sym(E_WEDGE_INDEX, cdf.m.WedgeIndex[mi_size], 16);
                wedge_sign = io.bit(E_WEDGE_SIGN);
            } else if (compound_type == EA1_COMP_DIFFWTD) {
                mask_type = io.bit(E_MASK_TYPE);
            }
        }
    }
    // 7.11.3.15: the weight of the first prediction out of 16
    int distance_weight() const
    {
        int dist[2];
        for (int i = 0; i < 2; i++) dist[i] = std::min(31, std::abs(relative_dist(f.order_hint[ref_frame[i]], f.hdr->order_hint)));
        const int d0 = dist[1], d1 = dist[0];
        const int order = d0 <= d1;
        if (d0 == 0 || d1 == 0) return kQuantDistLookup[3][order];
        int i;
        for (i = 0; i < 3; i++) {
            const int c0 = kQuantDistWeight[i][order], c1 = kQuantDistWeight[i][!order];
            // (round 4, pinned on libaom streams beside dav1d 1.5.3: the nearer reference's product must EXCEED the farther one's to stop)
            if (order) {
                if (d0 * c0 > d1 * c1) break;
            } else if (d0 * c0 < d1 * c1) {
                break;
            }
        }
        return kQuantDistLookup[i][order];
    }

#include "av1_tile_syntax_mvpred.inc"
#include "av1_tile_syntax_residual.inc"
};

}  // namespace av1
}  // namespace ea1
