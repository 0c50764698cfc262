// Internal device-side view of parsed frames and of the decoder context, shared by the AV1
// kernels (av1_*.hip).  Not part of the C ABI.
//
// Batching model: a context owns `nchains` independent decoding chains (closed GOPs are
// independent units); one submit reconstructs ONE frame of EVERY chain, and each
// kernel launch covers all chains (blockIdx.z = chain).  A 1080p frame is only ~10 MB of
// algorithmic traffic (~2 us at HBM speed), so a launch that serves one frame is pure launch
// latency; 16-64 frames per launch give the chip >>256 workgroups per kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "easyav1_amd_video.h"

namespace ea1 {

constexpr int kMaxChains = 256;
// intra wavefront (launch_intra): a level whose widest chain has at least kWideLevel items gets a launch of its own; runs of at least
// kShortRun narrower levels are walked by one workgroup per chain
constexpr int kWideLevel = 96;
constexpr int kShortRun = 3;

// One picture in HBM: three planes in one allocation, row stride a multiple of 128 bytes so that
// every row start is aligned for 16-byte vector access.
struct DevPic {
    uint8_t *p[3];
    int stride[3];
    int w, h;  // luma size
    uint8_t ssx, ssy;   // chroma subsampling of the context (ea1_recon_create_layout): 1, 1 for 4:2:0
    // Round2(size, subsampling): the chroma planes of an odd-sized (visible) picture round up
    __host__ __device__ int pw(int pl) const { return pl ? (w + ssx) >> ssx : w; }
    __host__ __device__ int ph(int pl) const { return pl ? (h + ssy) >> ssy : h; }
};

constexpr int kTxSizes = EA1_NUM_TX_SIZES;
// transform sizes in the order of ea1_tx_size_index(): squares, 2:1 shapes, 4:1 shapes
__host__ __device__ constexpr int tx_size_index(int wl, int hl)
{
    return wl == hl ? wl - 2
                    : (wl + 1 == hl ? 5 + 2 * (wl - 2)
                                    : (hl + 1 == wl ? 6 + 2 * (hl - 2)
                                                    : (wl + 2 == hl ? 13 + 2 * (wl - 2) : (hl + 2 == wl ? 14 + 2 * (hl - 2) : -1))));
}

// one transform block to run; built on the host at job creation
struct TxItem {
    uint32_t coef_off;  // int16 index of its coefficients
    uint16_t x, y;      // position in the plane, pixels
    uint8_t plane;
    uint8_t txtp;
    uint8_t intra;      // 1: residual goes to the residual plane (added by the wavefront kernel)
    uint8_t nz_log2;    // 2..5: all non-zero coefficients lie in the top-left (1 << nz_log2)^2 corner
    uint8_t corner;     // != 0: the arena holds that corner only, its width as the row stride (EA1_COEFS_CORNER);
                        // frames handed over in EA1_COEFS_DIAGONAL layout: 1 + D (last coded diagonal) until
                        // expand_coefs_kernel has written the corner, and src24 = the block's packed entries
    uint8_t src24[3];   // EA1_COEFS_DIAGONAL only: (int16 index into the uploaded packed coefficients) / 2, little endian
};
static_assert(sizeof(TxItem) == 16, "TxItem layout");

// one step of the intra wavefront (a transform block of an intra block, or a row part of one):
// everything the kernel needs in one 16-byte load
struct IntraItem {
    uint16_t x, y;      // position in the plane, pixels
    uint8_t plane;
    uint8_t size;       // log2 width | log2 height << 4
    uint8_t mode;       // EA1_*_PRED (UV_CFL: DC + alpha * luma AC), kIntraFilter + m, kIntraPalette
    uint8_t flags;      // kII* bits
    int8_t angle_delta; // directional modes
    int8_t alpha;       // chroma from luma
    uint8_t part;       // row part (items taller than one wave pass are cut into parts)
    uint8_t ii;         // inter-intra blend: 1 + ii_mode (0 = plain intra)
    uint32_t aux;       // palette: byte offset of {colours, index map}; inter-intra wedge: mask index
};
static_assert(sizeof(IntraItem) == 16, "IntraItem layout");
enum : uint8_t {
    kIIHaveLeft = 1, kIIHaveTop = 2, kIIResid = 4, kIIHaveTopRight = 8, kIIHaveBottomLeft = 16,
    kIISmoothEdge = 32,  // a neighbour uses a SMOOTH mode (selects the edge filter strengths)
    kIIEdgeFilter = 64,  // sequence-level intra edge filter
};
constexpr uint8_t kIntraFilter = 32;   // mode = kIntraFilter + EA1_FILTER_*
constexpr uint8_t kIntraPalette = 48;
constexpr uint8_t kIntraBC = 49;       // intra block copy: aux = block vector (y << 16 | x & 0xFFFF, 1/8 luma sample)

// where a palette item finds its colours and the first index of its transform block (byte
// offsets into the aux arena); IntraItem::aux indexes the job's PalRef table
struct PalRef {
    uint32_t colors_off, map_off;
    uint32_t map_stride;
};

// one inter-prediction tile: a TS x TS luma square (TS = 8, 16 or 32) with ONE motion vector set.
// Blocks are cut into tiles of min(w, h, 32); built on the host at job creation (32 bytes, two
// uniform 16-byte loads instead of a block-map -> block-record chain).
struct McItem {
    uint16_t x8, y8;    // tile origin in 8-pixel units
    uint8_t flags;      // bit1 compound, bit2 / bit3: chroma tile of a block whose chroma is 4 wide / high
    uint8_t filt;       // filter_h | filter_v << 2
    uint8_t ref[2];
    int16_t mv[2][2];   // [ref][y, x]
    // extended compound types only (tiles of the "ext" lists):
    uint32_t mask_off;  // wedge: offset of this tile's first mask byte in the context's wedge table
    uint8_t comp_type;  // EA1_COMP_*
    uint8_t comp_arg;   // DIST: weight of ref[0] (of 16); DIFFWTD: 1 = inverted mask
    uint8_t mask_stride_log2;  // wedge: log2 of the block width (row stride of its mask)
    uint8_t ox8, oy8;   // position of the tile inside its block, 8-sample units (scaled references step from the BLOCK's corner)
    uint8_t pad[7];
};
static_assert(sizeof(McItem) == 32, "McItem layout");

// one overlapped-block-motion-compensation strip (AV1 spec 7.11.3.10): the part of an OBMC block
// next to ONE inter neighbour above (pass 0) or to the left (pass 1), predicted with that
// neighbour's motion and blended into the block's own prediction
struct ObmcItem {
    uint16_t x, y;      // strip origin, luma samples
    uint8_t w, h;       // strip size, luma samples (chroma: half)
    uint8_t flags;      // bit0: also the chroma planes
    uint8_t filt;       // the NEIGHBOUR's filter_h | filter_v << 2
    uint8_t ref;        // the neighbour's first reference slot
    uint8_t pad[3];
    int16_t mv[2];      // the neighbour's first motion vector (y, x)
};
static_assert(sizeof(ObmcItem) == 16, "ObmcItem layout");

// one small inter prediction (blocks with a 4-sample luma side): a w x h rectangle of one plane
// predicted from one reference; luma of such blocks, and their chroma either as one block or --
// when the whole 8x8 luma area is inter -- piecewise with each block's own motion (spec 7.11.3.1)
struct PieceItem {
    uint16_t x, y;      // position in the plane, samples
    uint8_t w, h;       // 2..16, w * h <= 64
    uint8_t plane;
    uint8_t filt;       // filter_h | filter_v << 2
    uint8_t ref;
    uint8_t pad[3];
    int16_t mv[2];      // y, x in 1/8 luma sample
};
static_assert(sizeof(PieceItem) == 16, "PieceItem layout");

// one rectangle (at most 8x8) of one plane of a block with warped motion (AV1 spec 7.11.3.5):
// each of its one or two predictions is either warped -- the host has evaluated the affine map at
// the centre of the 8x8 (ix4/iy4 integer position, sx4/sy4 16-bit fraction) and the shear
// parameters -- or translated by the block's motion vector (matrix failed the shear test, or the
// chroma block is smaller than 8x8)
struct WarpRef {
    int32_t ix4, iy4;
    uint16_t sx4, sy4;
    int16_t alpha, beta, gamma, delta;
    int16_t mv[2];      // translated prediction: y, x in 1/8 luma sample
};
struct WarpItem {
    uint16_t x, y;      // position in the plane, samples
    uint8_t plane, w, h;
    uint8_t flags;      // bit0 compound, bit1 / bit2: prediction 0 / 1 is warped
    uint8_t ref[2];
    uint8_t fset;       // translated predictions: filter set (0..5) horizontal | vertical << 4
    uint8_t comp_type, comp_arg;
    uint8_t mask_stride_log2;   // wedge compound: log2 of the block width (row stride of its mask)
    uint8_t bx, by;             // position of the rectangle inside its block, plane samples (a translated prediction from a SCALED reference steps from the block's corner)
    WarpRef r[2];
    uint32_t mask_off;          // wedge compound: offset in the context's wedge table of the mask byte of this
                                // rectangle's first sample (chroma: of the luma sample (2x, 2y) it averages from)
    uint32_t pad2[3];
};
static_assert(sizeof(WarpRef) == 24 && sizeof(WarpItem) == 80, "WarpItem layout");

// Everything the kernels need about ONE frame of ONE chain; lives in device memory, built once
// per (job, chain) and referenced by pointer from the launch arguments.
//
// Deblocking edge classes (GPU analogue of dav1d's lf_mask.c output), from the transform grid:
//   luma_edge   per 4x4 luma unit:   bits 0-1 left edge, bits 2-3 top edge
//               0 none, 1 -> 4-tap, 2 -> 8-tap, 3 -> 16-wide (14-tap) class
//   chroma_edge per 4x4 chroma unit (4:2:0: 8x8 luma; cw4 x ch4 of them): bits 0-1 left, bits 2-3 top; 0 none, 1 -> 4-tap, 2 -> 6-tap
// unit_flags (per 8x8 luma unit): bit0 = every block in it has the skip flag
struct FrameView {
    ea1_frame_desc desc;
    const int16_t *coefs;
    const McItem *mc_items[3];  // tiles of 8x8, 16x16 and 32x32 luma: single reference / averaged compound
    int mc_count[3];
    const McItem *mc_ext_items[3];  // tiles with distance-weighted / wedge / difference-weighted compound
    int mc_ext_count[3];
    const ObmcItem *obmc_items[2];  // pass 0 (from above), pass 1 (from the left)
    int obmc_count[2];
    const PieceItem *piece_items;   // predictions of blocks with a 4-sample side
    int piece_count;
    const WarpItem *warp_items;     // predictions of blocks with warped motion: luma rectangles first, then chroma
    int warp_count, warp_luma_count;
    const int8_t *warp_filter;  // the context's warp filter table, 193 x 8 (NULL until set)
    const uint8_t *wedge;       // the context's wedge mask table
    const uint8_t *luma_edge;   // [h4][w4]
    const uint8_t *chroma_edge; // [ch4][cw4]
    const uint8_t *unit_flags;  // [h8][w8]
    const uint16_t *lvl_y;      // [h4][w4] luma deblock levels of the unit's block: Y-vertical | Y-horizontal << 8
    const uint16_t *lvl_c;      // [ch4][cw4] chroma levels of the block owning the unit's chroma: U | V << 8
    const uint8_t *aux;         // palette data
    const PalRef *pal_refs;
    const uint8_t *cdef_idx;    // [sb_rows][sb_cols]
    const ea1_lr_unit *lr[3];
    int lr_cols[3], lr_rows[3];
    // transform work lists (inter AND intra blocks), one per transform size
    const TxItem *tx_items[kTxSizes];
    int tx_count[kTxSizes];
    // intra wavefront: items (all planes) sorted by dependency level
    const IntraItem *intra_items;
    const uint32_t *level_start;  // [num_levels + 1]
    int num_levels;
    int w8, h8, sb_cols, sb_rows;
    int cw4, ch4;        // 4x4 units of a chroma plane: (w8, h8) for 4:2:0, (w8, 2 h8) for 4:2:2, (2 w8, 2 h8) for 4:4:4
    int ssx, ssy;        // chroma subsampling (the context's layout)
    int vis_w, vis_h;    // visible luma size (ea1_frame_desc.crop): reference reads and restoration clamp here; the
                         // reference slots below carry it as their w / h, the other pictures the coded size
    // chain buffers this frame works on
    DevPic work;         // reconstruction + in-place deblocking
    DevPic cdef;         // CDEF output
    DevPic dst;          // finished picture (reference slot)
    DevPic slots[EA1_NUM_REF_SLOTS];  // reference pictures of this chain (inline: uniform scalar loads)
    // super-resolution (ea1_frame_desc.upscaled_width, up_vis_w != 0): the deblocked and the CDEF-filtered picture upscaled
    // horizontally (av1_lr.hip upscale_kernel); restoration reads THESE and counts its units on up_vis_w; up_cw: their stored width
    DevPic up_deb, up_cdef;
    int up_cw, up_vis_w;
    const int16_t *upscale_filter;   // [64][8]
    int16_t *resid[3];   // residual planes for intra blocks (same geometry as `work`, stride in int16)
    uint8_t *dir_var;    // CDEF direction / variance class, 2 bytes per 8x8 unit
    // Scaled references (AV1 spec 7.11.3.3): a reference picture of another size than this frame.  ref_scale[k] =
    // {xScale, yScale} of slot k in 1 / 16384 units (16384: same size); scaled_mask: the slots this frame reads that are
    // not 16384 -- all its translated predictions then go through mc_scaled_kernel (av1_mc.hip)
    int ref_scale[EA1_NUM_REF_SLOTS][2];
    uint32_t scaled_mask;
};

// XCD-aware placement of a (gx, gy, gz) tile grid.  Workgroups are dealt round-robin over the 8
// XCDs, each with a private L2, so with the plain 3-D grid every tile's neighbours -- whose halos
// and partial cache lines it shares -- sit behind other L2s and each line is fetched from the
// fabric once per XCD that needs it.  Launched as a 1-D grid of 8 * ceil(tiles / 8) workgroups
// instead, workgroup b takes tile (b % 8) * ceil(tiles / 8) + b / 8 (x fastest, then y, then z): every
// XCD walks one contiguous eighth of the tiles.  Placement only affects speed, never results.
struct TilePos {
    int x, y, z;
    bool valid;
};
// the grid decomposition divides by run-time values; a division costs ~30 vector instructions per
// wave (a few per cent of a filter tile), so the host passes magic multipliers instead and the
// quotients are one scalar multiply-high each: q = (t * m) >> 32 with m = floor(2^32 / d) + 1 is
// exact for t < 2^32 / d (tile counts stay below 2^24, d below 2^8); m = 0 stands for d = 1.
struct TileGrid {
    int gx, gy, gz;
    uint32_t mx, my;
    int x_origin;   // plane x (samples) of tile column 0: 0 for whole frames, the strip's first column in a tile-column split
};
inline uint32_t tile_magic(int d) { return d <= 1 ? 0u : uint32_t((uint64_t(1) << 32) / uint32_t(d)) + 1u; }
inline TileGrid tile_grid(int gx, int gy, int gz, int x_origin = 0) { return TileGrid{gx, gy, gz, tile_magic(gx), tile_magic(gy), x_origin}; }
#ifdef __HIPCC__
__device__ __forceinline__ TilePos xcd_tile(const TileGrid &g)
{
    const uint32_t total = uint32_t(g.gx) * g.gy * g.gz, per = gridDim.x >> 3;
    const uint32_t t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    TilePos r;
    r.valid = t < total;
    const uint32_t row = g.mx ? __umulhi(t, g.mx) : t;
    r.x = int(t - row * g.gx);
    r.z = int(g.my ? __umulhi(row, g.my) : row);
    r.y = int(row - uint32_t(r.z) * g.gy);
    return r;
}
#endif
inline dim3 xcd_grid(const TileGrid &g) { return dim3(unsigned((size_t(g.gx) * g.gy * g.gz + 7) / 8 * 8)); }

// launch argument: one FrameView pointer per chain (NULL = chain idle in this submit)
struct Batch {
    const FrameView *v[kMaxChains];
    int n;
};

// host-side accounting of a job (SURVEY.md 8(d) per-kernel algorithmic bytes)
struct JobStats {
    uint64_t tx_coefs[kTxSizes], tx_pixels[kTxSizes];   // per size: coefficient positions / pixels touched
    uint64_t intra_pixels;                // pixels predicted by the wavefront
    uint64_t intra_resid_pixels;          // of which carry a residual
    uint64_t inter_luma_px, ref_luma_px;  // predicted luma pixels / luma reference pixels fetched
    int tx_count[kTxSizes];
    int mc_count[3];
    int mc_ext_count[3];
    int obmc_count[2];
    int piece_count;
    int interintra_blocks;
    uint64_t piece_px;
    int warp_count, warp_luma_count;
    uint64_t warp_px;
    uint32_t ref_mask;     // reference slots the frame's inter blocks (and OBMC neighbours) read
    uint32_t warp_ref_mask;   // ... the slots a prediction of the warp list really warps from (a scaled reference must not be among them)
    uint64_t obmc_px[2];   // samples (all planes) blended per pass
    int num_levels;
    int max_level_items;
    int level0_items;
    const uint32_t *level_start;      // host copy of the level offsets, num_levels + 1 entries (owned by the job)
    bool lr_has_sgr[3];   // a restoration unit of the plane is self-guided
    bool intra_special;       // an intra item needs more than the edge-register predictors (directional, CfL, ...)
    bool luma_odd_row_edges;  // a luma transform edge lies on an odd 4-row unit (4-sample-high transforms)
    int w8, h8;
    int width, height;
    int strip_x0, strip_x1;   // luma columns whose blocks this job reconstructs ([0, width) unless made by ea1_job_create_strip)
    ea1_frame_desc desc;
};

// what one submit looks like to the launchers
struct SubmitInfo {
    uint32_t *wave_counters;     // device, 2 * kMaxChains words, zeroed at context creation (unused since the grid barrier was removed)
    int up_cw;                   // super-resolution: stored width of the upscaled pictures of this submit's frames (0: none)
    Batch batch;                 // device FrameViews
    const JobStats *stats[kMaxChains];  // host stats per chain (NULL = idle)
    int width, height;
    int ssx, ssy;                // chroma subsampling of the context's pictures
    // Tile-column split: the in-loop filters of this submit cover the luma columns
    // [win_x0, win_x1) only -- the rank's strip plus the margin its restoration needs -- instead of [0, width).
    // Multiples of 8.  The launchers may round outwards to their tile sizes: columns beyond the strip hold
    // scratch values that nothing reads.
    int win_x0, win_x1;          // deblock + CDEF window
    int lr_x0, lr_x1;            // restoration window (the strip itself, multiples of 64)
    bool any_scaled;             // a frame of the submit reads a reference of another size (FrameView::scaled_mask)
};

// Optional per-kernel timing with HIP events on the launch stream (off by default; bench.py's
// roofline figure).  Every launcher brackets its kernels with begin()/end().
struct Prof {
    virtual void begin(const char *kernel, uint64_t algorithmic_bytes, hipStream_t st) = 0;
    virtual void end(hipStream_t st) = 0;
    virtual ~Prof() = default;
};
struct ProfScope {
    Prof *p;
    hipStream_t st;
    ProfScope(Prof *p_, const char *kernel, uint64_t bytes, hipStream_t st_) : p(p_), st(st_)
    {
        if (p) p->begin(kernel, bytes, st);
    }
    ~ProfScope()
    {
        if (p) p->end(st);
    }
};

// wedge mask table (wedge.cpp): host-built, one copy per decoder context
uint32_t wedge_mask_offset(int wl2, int hl2, int sign, int index);
uint32_t wedge_table_bytes();
void build_wedge_table(uint8_t *out);

// kernel launchers (one per translation unit); all asynchronous on `st`
void launch_inter_pred(const SubmitInfo &s, hipStream_t st, Prof *prof);
void launch_itx(const SubmitInfo &s, hipStream_t st, Prof *prof);
void launch_intra(const SubmitInfo &s, hipStream_t st, Prof *prof);
void launch_deblock(const SubmitInfo &s, hipStream_t st, Prof *prof);
void launch_cdef(const SubmitInfo &s, hipStream_t st, Prof *prof, int in_second);
void launch_lr(const SubmitInfo &s, hipStream_t st, Prof *prof, int deb_second, int cdef_second);
void launch_upscale(const SubmitInfo &s, hipStream_t st, Prof *prof, int deb_second, int cdef_second);
// hand-over: deblocking edge classes, levels and CDEF unit flags derived on the GPU (av1_derive.hip)
struct DeriveArgs {
    const ea1_block *blocks;
    const ea1_txblock *txs;
    uint32_t num_blocks, num_tx;
    uint32_t *map;          // [h4][w4] scratch
    uint32_t *txmap[3];     // per plane scratch: [h4][w4], [h8][w8], [h8][w8]
    uint8_t *luma_edge, *chroma_edge, *unit_flags;
    uint16_t *lvl_y, *lvl_c;
    int w4, h4, w8, h8;
    int cw4, ch4, ssx, ssy; // chroma units / subsampling (FrameView)
    int vis_w, vis_h;       // visible luma size: 4x4 units that start at or beyond it are not deblocked (AV1 spec 7.14.2 onScreen)
    uint8_t lf_level[4];
    uint8_t lf_per_block;
    // ---- work lists made on the device (derive_items_kernel): transform items per size, inter tiles per class.
    // The host only COUNTS them (it validates every record anyway): per transform size four buckets by coded corner
    // (bucket b holds the hinted items with nz_log2 = b + 2; items without a parser hint and zero-fill items go to
    // bucket 0), per tile class the host-made items first (blocks with EA1_BLK_WARP whose set-up fell back to
    // translation), then the device-made ones.  Order inside a bucket is whatever the atomics give: items of one
    // list touch disjoint samples.
    uint8_t corner_layout;  // ea1_frame_desc.coef_layout == EA1_COEFS_CORNER
    uint8_t pad0;
    int strip_x0, strip_x1; // luma columns whose blocks get items (tile-column split)
    TxItem *tx_items[kTxSizes];
    uint32_t tx_base[kTxSizes][4];
    McItem *mc_items[3], *mc_ext_items[3];
    uint32_t mc_base[3], mc_ext_base[3];
    uint32_t *cursors;      // kDeriveCursors words, zeroed by derive_block_map_kernel
    int16_t *coefs;         // the job's coefficient region: kZeroCoefs zeros (written by derive_block_map_kernel), then the uploaded range
    uint32_t coef_rebase;   // TxItem::coef_off = coef_rebase + ea1_txblock::coef_off
    // EA1_COEFS_DIAGONAL: the uploaded entries are packed diagonals (`packed`, index = ea1_txblock::coef_off - packed_lo);
    // derive_items_kernel gives every coded block a corner-layout place behind the zeros (cursor kCoefCursor),
    // expand_coefs_kernel fills it
    const int16_t *packed;
    uint32_t packed_lo;
    uint8_t diagonal_layout;
    uint32_t tx_count[kTxSizes];
};
// log2 (width, height) of the transform sizes in tx_size_index() order
constexpr int kTxLog2[19][2] = {{2, 2}, {3, 3}, {4, 4}, {5, 5}, {6, 6}, {2, 3}, {3, 2}, {3, 4}, {4, 3}, {4, 5}, {5, 4}, {5, 6}, {6, 5},
                                {2, 4}, {4, 2}, {3, 5}, {5, 3}, {4, 6}, {6, 4}};
static_assert(kTxSizes == 19 && tx_size_index(kTxLog2[18][0], kTxLog2[18][1]) == 18 && tx_size_index(kTxLog2[11][0], kTxLog2[11][1]) == 11, "transform size table");
constexpr int kDeriveCursors = kTxSizes * 4 + 6 + 1;
constexpr int kCoefCursor = kTxSizes * 4 + 6;
// positions (x, y) of a sw x sh block with x + y < d (EA1_COEFS_DIAGONAL: entries in front of diagonal d)
__host__ __device__ constexpr int diag_before(int sw, int sh, int d)
{
    const int a = sw < sh ? sw : sh, b = sw < sh ? sh : sw;
    if (d <= a) return d * (d + 1) / 2;
    if (d <= b) return a * (a + 1) / 2 + (d - a) * a;
    const int r = sw + sh - 1 - d;
    return r > 0 ? sw * sh - r * (r + 1) / 2 : sw * sh;
}
constexpr uint32_t kZeroCoefs = 1024;   // zero coefficients in front of a job's uploaded coefficients (transform blocks that only clear)
// the jobs of one hand-over batch: their DeriveArgs live in their own arenas
struct DeriveBatch {
    const DeriveArgs *a[kMaxChains];
    int n;
};
// max_*: the largest count over the batch's jobs (grid sizes); max_items: the longest transform item list of a
// job in EA1_COEFS_DIAGONAL layout (0: none, no expansion launch)
void launch_derive(const DeriveBatch &b, uint32_t max_blocks, uint32_t max_tx, int max_units, uint32_t max_items, hipStream_t st);
// hand-over copies done by a kernel (av1_split.hip): dst / src 16-byte aligned, bytes a multiple of 2
struct PullSeg {
    void *dst;
    const void *src;   // pinned host memory (hipHostMalloc), read over the host link
    uint64_t bytes;
    uint32_t first_chunk, pad;
};
constexpr int kPullSegs = 32;
struct PullArgs {
    PullSeg seg[kPullSegs];
    int n;
    uint32_t total_chunks;
};
void launch_pull(PullArgs &a, hipStream_t st);
size_t columns_bytes(int w, int h);
void launch_copy_columns(const DevPic &pic, int x, int w, uint8_t *buf, bool to_picture, hipStream_t st);
int launch_dsp_itx_add(uint8_t *plane, size_t stride, int w, int h, const int16_t *d_coefs, const uint32_t *pos,
                       const uint8_t *txtp, int count, hipStream_t st);

}  // namespace ea1
