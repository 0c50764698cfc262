/*
 * easyav1_amd_video.h -- C ABI of the MI355X AV1 frame-reconstruction path.
 *
 * Drop-in boundary.  easyav1 decodes video through dav1d_send_data / dav1d_get_picture
 * (reference: /root/reference/src/easyav1.c:2390, :2399, :2403; open/close/flush at :1405,
 * :3969, :3059).  BASELINE.json's north_star keeps OBU + entropy/symbol parse on the host, so the
 * GPU-side boundary is "reconstruct one frame from its parsed form".  dav1d (absent from
 * /root/reference: empty submodule, listed only in cmake/dav1d.cmake:970-1015) has exactly this
 * split internally when frame threading is on: pass 1 parses into per-frame buffers
 * (coefficients, per-block modes), pass 2 reconstructs and filters.  The structures below are
 * that hand-over ("ReconJob" in SURVEY.md section 7), re-designed for a GPU consumer:
 *
 *   dav1d unit (cmake/dav1d.cmake line)                     here
 *   ------------------------------------------------------  -----------------------------------
 *   dav1d_open/close            (easyav1.c:1405,:3969)      ea1_recon_create / _destroy
 *   frame_thread cf/b buffers,  recon_tmpl.c (:1014)        ea1_frame_desc + ea1_block[] + coefs
 *   dav1d_send_data past the parse (easyav1.c:2390)         ea1_job_create + ea1_recon_submit
 *   itx_tmpl.c/itx_1d.c         (:1008,:979)                itx kernels (also ea1_dsp_itx_add)
 *   mc_tmpl.c                   (:1013)                     inter-prediction kernel
 *   ipred_tmpl.c                (:1007)                     intra wavefront kernel
 *   loopfilter_tmpl.c/lf_mask.c (:1010,:980)                deblock kernels
 *   cdef_tmpl.c                 (:1003)                     CDEF kernel
 *   looprestoration_tmpl.c      (:1011)                     Wiener / self-guided kernels
 *   dav1d_get_picture           (easyav1.c:2399)            ea1_recon_get_picture / _download
 *   dav1d_flush                 (easyav1.c:3059)            ea1_recon_flush
 *
 * Scope of this IR revision (documented restrictions): 8 bit; 4:2:0 (which carries 4:0:0), 4:4:4 and 4:2:2 -- a context is
 * made for one layout (ea1_recon_create_layout), the text below describes 4:2:0 and ea1_frame_desc.layout says what changes; coded blocks
 * 4x4..128x128 of any AV1 shape (1:1, 1:2, 2:1, 1:4, 4:1), including the blocks with a 4-sample
 * luma side, whose chroma is owned by the last block of each 8x8 luma area and, when all blocks of
 * that area are inter, predicted piecewise with each block's own motion; transform blocks of all 19 AV1 sizes 4x4..64x64 listed per block (uniform grids and
 * variable inter transform trees alike); every AV1 intra predictor (DC, V, H, PAETH, SMOOTH*, the
 * eight directional modes with angle delta, edge filter and upsampling, recursive filter-intra,
 * chroma-from-luma, palette) and intra block copy; inter prediction with the regular / smooth /
 * sharp 8-tap (4-tap for 4-sample sides) and bilinear filters: single reference, averaged,
 * distance-weighted, wedge and difference-weighted masked compound, inter-intra (smooth and
 * wedge), overlapped block motion compensation, warped motion (local warp and global motion, with
 * the filter table of ea1_av1_tables.inc or the caller's own); deblock levels per frame or per block; frames of any size up to the
 * context's, scaled references, super-resolution.  Not in this revision: high bit depth.  Film grain on the output picture: easyav1_amd_grain.h.
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_VIDEO_H
#define EASYAV1_AMD_VIDEO_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_audio.h" /* error codes */

#ifdef __cplusplus
extern "C" {
#endif

/* pictures a reconstruction chain keeps (indices for ea1_frame_desc.dst_slot / ea1_block.ref): AV1's eight reference slots may name eight
 * DIFFERENT pictures while a frame that refreshes none of them is decoded (a leaf of libaom's frame pyramid does), so nine (round 4) */
#define EA1_NUM_REF_SLOTS 9

/* 1-D transform kinds; a 2-D transform type is (col_kind << 2) | row_kind; EA1_TXTP_WHT is the
 * lossless 4x4 Walsh-Hadamard transform */
enum { EA1_TX_DCT = 0, EA1_TX_ADST = 1, EA1_TX_FLIPADST = 2, EA1_TX_IDENTITY = 3 };
#define EA1_TXTP(col_kind, row_kind) (((col_kind) << 2) | (row_kind))
#define EA1_TXTP_WHT 16

/* intra prediction modes (AV1 numbering) */
enum {
    EA1_DC_PRED = 0, EA1_V_PRED = 1, EA1_H_PRED = 2, EA1_D45_PRED = 3, EA1_D135_PRED = 4,
    EA1_D113_PRED = 5, EA1_D157_PRED = 6, EA1_D203_PRED = 7, EA1_D67_PRED = 8, EA1_SMOOTH_PRED = 9,
    EA1_SMOOTH_V_PRED = 10, EA1_SMOOTH_H_PRED = 11, EA1_PAETH_PRED = 12,
    EA1_UV_CFL_PRED = 13,   /* chroma only */
    /* intra block copy (AV1 spec 5.11.x use_intrabc, 7.11.3 with the current frame as reference):
     * mode_y = mode_uv = EA1_INTRABC_PRED, mv[0] = the block vector (whole luma samples, i.e.
     * multiples of 8; chroma may land on a half sample and is then interpolated bilinearly).
     * Intra frames only, frames without in-loop filters (allow_intrabc switches them off), blocks
     * of any size (a block with a 4-sample side copies its luma; the chroma block of its 8x8 area
     * is copied by the area's last block, with that block's vector); every sample the copy reads
     * must belong to a block earlier in the list. */
    EA1_INTRABC_PRED = 14,
    EA1_N_INTRA_MODES = 15
};
/* recursive filter-intra modes (ea1_block.filter_intra = 1 + mode) */
enum { EA1_FILTER_DC = 0, EA1_FILTER_V = 1, EA1_FILTER_H = 2, EA1_FILTER_D157 = 3, EA1_FILTER_PAETH = 4 };

/* interpolation filter kinds (per direction) */
enum { EA1_FILTER_REGULAR = 0, EA1_FILTER_SMOOTH = 1, EA1_FILTER_SHARP = 2, EA1_FILTER_BILINEAR = 3 };

/* compound prediction kinds (ea1_block.comp_type, blocks with EA1_BLK_COMPOUND) */
enum {
    EA1_COMP_AVG = 0,      /* (a + b + 1) >> 1 of the two predictions                        */
    EA1_COMP_DIST = 1,     /* distance weights: comp_arg = weight of ref[0] out of 16        */
    EA1_COMP_WEDGE = 2,    /* comp_arg = wedge index (0..15) | sign << 4                     */
    EA1_COMP_DIFFWTD = 3   /* mask from |a - b|: comp_arg = 0 (38 base) or 1 (inverted)      */
};
/* inter-intra kinds (ea1_block.ii_mode, blocks with EA1_BLK_INTERINTRA): 0..3 = II_DC, II_V,
 * II_H, II_SMOOTH; bit 7 set = wedge mask with ii_arg = wedge index (sign is always 0) */
enum { EA1_II_DC = 0, EA1_II_V = 1, EA1_II_H = 2, EA1_II_SMOOTH = 3, EA1_II_WEDGE = 0x80 };

/* ea1_block.flags */
#define EA1_BLK_INTER 0x0001      /* inter-predicted (else intra)                             */
#define EA1_BLK_COMPOUND 0x0002   /* two references                                           */
#define EA1_BLK_SKIP 0x0004       /* no residual in any plane (CDEF skip, deblock inner edges) */
#define EA1_BLK_HAVE_LEFT 0x0008  /* left neighbour is inside the tile/frame (intra edges)     */
#define EA1_BLK_HAVE_TOP 0x0010   /* top neighbour is inside the tile/frame                   */
#define EA1_BLK_HAVE_TOP_RIGHT 0x0020   /* samples above and right of the block are decoded   */
#define EA1_BLK_HAVE_BOTTOM_LEFT 0x0040 /* samples left of and below the block are decoded    */
#define EA1_BLK_OBMC 0x0080       /* overlapped block motion compensation                     */
#define EA1_BLK_INTERINTRA 0x0100
#define EA1_BLK_PALETTE_Y 0x0200  /* luma palette (pal_size[0] colours, index map in aux)     */
#define EA1_BLK_PALETTE_UV 0x0400
/* blocks with a 4-sample luma side that own chroma: edge availability of their CHROMA block (it
 * starts one 4x4 unit before the luma block where that coordinate is odd); blocks with both sides
 * >= 8 use the four flags above for all planes */
#define EA1_BLK_CHROMA_HAVE_LEFT 0x0800
#define EA1_BLK_CHROMA_HAVE_TOP 0x1000
#define EA1_BLK_CHROMA_HAVE_TOP_RIGHT 0x2000
#define EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT 0x4000
/* warped motion (AV1 spec 7.11.3.5, dav1d warp8x8 / warpmv.c): local warp (motion_mode) or global
 * motion.  ea1_block.warp_refs tells which of the block's predictions are warped; aux_off points
 * to one warp matrix per warped prediction, in reference order: 6 little-endian int32 each
 * (warpParams[0..5] of the specification, 16 fractional bits).  The shear parameters and their
 * validity test (7.11.3.6) are derived here; a prediction whose matrix fails the test is made by
 * plain translation with the block's motion vector, as the specification prescribes.  Needs blocks
 * of at least 8x8 luma samples; chroma is warped when the chroma block is at least 8x8, and
 * translated otherwise.  Every compound kind may have warped predictions.
 * The 193 x 8 filter table: the library's own, or the caller's through ea1_recon_set_warp_filter(). */
#define EA1_BLK_WARP 0x8000

/* One coded block (prediction unit): luma w x h = (1 << bwl) x (1 << bhl) at (4*x4, 4*y4);
 * chroma is the co-located (w/2) x (h/2).  A block with a 4-sample side owns chroma only if it is
 * the LAST block of its 8x8 luma area in that direction (x4 odd for w == 4, y4 odd for h == 4);
 * its chroma block then covers max(w,8)/2 x max(h,8)/2 samples starting at the 8-aligned luma
 * position, and its chroma transform blocks are listed with it.  Such blocks are single-reference,
 * without OBMC / inter-intra; a palette is possible from 64 samples on (4x16 and 16x4: the chroma
 * palette then belongs to the chroma block just described).  Blocks are listed in decoding order.  48 bytes. */
typedef struct ea1_block {
    uint16_t x4, y4;        /* luma position in 4-pixel units                                */
    uint8_t bwl, bhl;       /* log2 of luma width / height in pixels (2..7, ratio <= 4; 7 only as 128x128, 128x64, 64x128) */
    uint16_t flags;         /* EA1_BLK_*                                                     */
    uint8_t mode_y;         /* intra: luma mode; inter: filter_h | filter_v << 2             */
    uint8_t mode_uv;        /* intra: chroma mode                                            */
    uint8_t ref[2];         /* reference slots (0..7) of the two predictions                 */
    int8_t angle_delta[2];  /* directional modes: -3..3 steps of 3 degrees, luma / chroma    */
    int8_t cfl_alpha[2];    /* EA1_UV_CFL_PRED: signed alpha of U, V in 1/8 units (-16..16)  */
    int16_t mv[2][2];       /* [ref][0 = y, 1 = x] in 1/8 luma pixel                         */
    uint8_t comp_type;      /* EA1_COMP_*                                                    */
    uint8_t comp_arg;
    uint8_t ii_mode;        /* EA1_II_*                                                      */
    uint8_t ii_arg;
    uint8_t filter_intra;   /* 0: none, else 1 + EA1_FILTER_* (luma, replaces mode_y)        */
    uint8_t pal_size[2];    /* palette sizes (2..8) of luma / chroma, 0 = no palette         */
    uint8_t warp_refs;      /* EA1_BLK_WARP: bit r set = the prediction from ref[r] is warped  */
    uint32_t first_tx;      /* first entry of this block in the transform-block list         */
    uint16_t n_tx;          /* its number of entries (all planes)                            */
    uint16_t pad1;
    uint32_t aux_off;       /* palette data in the aux arena: pal_size[0] luma colours, then
                               pal_size[1] U and pal_size[1] V colours, then the luma index map
                               (w*h bytes, row-major) and the chroma index map (the chroma block's
                               width x height: w/2 x h/2, max(w,8)/2 x max(h,8)/2 for a 4-sample side);
                               EA1_BLK_WARP blocks: their warp matrices (24 bytes each)       */
    /* OBMC neighbours are found through the block map; their motion is taken from their records */
    uint8_t lf_level[4];    /* deblock levels of this block (Y vertical edges, Y horizontal edges, U,
                               V) after segment / reference / mode deltas; read only when
                               ea1_frame_desc.lf_per_block is set.  0 = take the level of the block
                               on the other side of the edge (AV1 spec 7.14.4)                   */
} ea1_block;

/* One transform block.  The entries of a block follow each other: luma in raster order, then U,
 * then V (the order in which intra prediction + reconstruction proceeds), and tile the block's
 * area in each plane exactly.  Entries without coefficients (coef_off = EA1_NO_COEFS) are still
 * listed: intra prediction runs per transform block and deblocking follows the transform grid.
 * 12 bytes. */
#define EA1_NO_COEFS 0xFFFFFFFFu
typedef struct ea1_txblock {
    uint16_t x4, y4;     /* position IN THE PLANE in 4-sample units                          */
    uint8_t plane;       /* 0..2                                                            */
    uint8_t size;        /* log2 width | log2 height << 4 (2..6 each, ratio <= 4)            */
    uint8_t txtp;        /* EA1_TXTP(col, row) or EA1_TXTP_WHT                               */
    uint8_t nz_log2;     /* optional bound on the coded coefficients, from the parser's end-of-block
                            position: 2..5 = every non-zero coefficient lies in the top-left
                            (1 << nz_log2) x (1 << nz_log2) corner (coefficients beyond it are NOT
                            read); 0 = unknown, the hand-over scans the block (dav1d's itxfm_add gets
                            the same information as `eob`)                                       */
    uint32_t coef_off;   /* int16 index (even) into the coefficient arena: min(w,32) x
                            min(h,32) entries, row-major (the coded corner only with
                            ea1_frame_desc.coef_layout == EA1_COEFS_CORNER); EA1_NO_COEFS = all zero */
} ea1_txblock;

enum { EA1_COEFS_DENSE = 0, EA1_COEFS_CORNER = 1, EA1_COEFS_DIAGONAL = 2 };
/* ea1_txblock.nz_log2 in the EA1_COEFS_DIAGONAL layout: the corner class (2..5) and the last coded anti-diagonal */
#define EA1_NZ_DIAGONAL(corner_log2, last_diagonal) ((uint8_t)(((corner_log2) - 2) | ((last_diagonal) << 2)))

/* loop-restoration unit parameters (8 bytes) */
enum { EA1_LR_NONE = 0, EA1_LR_WIENER = 1, EA1_LR_SGRPROJ = 2 };
typedef struct ea1_lr_unit {
    uint8_t type;        /* EA1_LR_*                                                       */
    uint8_t sgr_set;     /* 0..15                                                          */
    int8_t coef[2][3];   /* Wiener: [0] vertical, [1] horizontal taps c0..c2 (c3 implied);
                            self-guided: coef[0][0..1] = xqd[0..1]                         */
} ea1_lr_unit;

/* Per-frame parameters (host struct, copied at job creation) */
typedef struct ea1_frame_desc {
    uint32_t width, height;         /* luma pixels, multiples of 8                          */
    uint8_t is_inter;               /* 0: key/intra-only frame                              */
    uint8_t dst_slot;               /* reference slot the finished frame is stored in       */
    uint8_t lf_level[4];            /* deblock: Y vertical-edge, Y horizontal-edge, U, V; 0 = off */
    uint8_t lf_sharpness;           /* 0..7                                                 */
    uint8_t cdef_enable;
    uint8_t cdef_damping;           /* 3..6                                                 */
    uint8_t cdef_y_strength[8];     /* pri << 2 | sec                                       */
    uint8_t cdef_uv_strength[8];
    uint8_t lr_type[3];             /* 0: plane not restored, else per-unit types are used  */
    uint8_t lr_unit_log2[3];        /* restoration unit size: 6, 7 or 8                     */
    uint8_t intra_edge_filter;      /* sequence header enable_intra_edge_filter             */
    uint8_t lf_per_block;           /* 1: deblock levels come from ea1_block.lf_level (delta_lf /
                                       segment / ref / mode deltas).  lf_level[] above still gates: both luma
                                       levels 0 = no deblocking at all, lf_level[2] / [3] = 0 = that chroma
                                       plane not filtered (dav1d_calc_lf_values); a single luma level of 0
                                       does NOT switch its pass off then -- the blocks' own levels decide */
    uint8_t crop;                   /* visible size = (width - (crop & 15)) x (height - (crop >> 4)) luma samples,
                                       0..7 each: a frame whose size is not a multiple of 8 is coded in whole
                                       8x8 units, but references are read and restoration runs with the edges
                                       of the VISIBLE picture (AV1 spec 7.11.3.3 lastX / lastY, 7.17 PlaneEndX /
                                       unit counts); chroma: (size + 1) >> 1.  Deblocking skips the 4x4 units that
                                       start at or beyond the visible size (7.14.2 onScreen).  Beyond the visible area the
                                       finished picture holds the CDEF output unchanged (never read again) */
    uint8_t coef_layout;            /* EA1_COEFS_DENSE: a coded transform block owns min(w,32) x min(h,32) entries of the
                                       coefficient arena, row-major.  EA1_COEFS_CORNER: it owns only its coded corner,
                                       min(w, 32, n) x min(h, 32, n) entries with n = 1 << ea1_txblock.nz_log2, row-major
                                       with that corner's width as the row stride; every coded transform block then
                                       states nz_log2 (2..5).  A parser that knows the end-of-block position writes
                                       a quarter of the bytes this way, and they are what crosses PCIe.
                                       EA1_COEFS_DIAGONAL: it owns the positions (x, y) of its min(w,32) x min(h,32)
                                       block with x + y <= D, in the order d = x + y ascending and y ascending inside
                                       a diagonal -- the shape a coding scan up to the end-of-block position fills;
                                       ea1_txblock.nz_log2 = EA1_NZ_DIAGONAL(corner class 2..5, D), D in 0 .. w + h - 2.
                                       Positions beyond the corner or beyond diagonal D are zero.  A third of the bytes
                                       of the corner layout on the graded streams; the hand-over expands it to the
                                       corner layout on the device (av1_derive.hip), the kernels read that.
                                       coef_off stays even in every layout (a block's entries may end on an odd count) */
    uint8_t pad;                    /* blocks that reach beyond the coded frame (AV1: whenever the frame size is not a multiple of the
                                       block sizes in use at its right / bottom edge): width x height above is then the coded size
                                       ROUNDED UP so that every block fits -- the true coded size (MiCols * 4 x MiRows * 4) is
                                       (width - 8 * (pad & 15)) x (height - 8 * (pad >> 4)), 0..15 units each, and `crop` counts from it.
                                       The blocks still tile width x height exactly: the parser closes what its blocks leave open in
                                       the padding with filler blocks (intra DC, skipped) at the end of the list.  What follows the TRUE
                                       coded size: intra edge samples are fetched up to it and repeated beyond (7.11.2 maxX / maxY), CDEF
                                       filters 8x8 blocks inside it and treats samples beyond it as unavailable (7.15), OBMC takes
                                       neighbours inside it (7.11.3.10), chroma-from-luma repeats the last coded luma column / row of
                                       a block whose luma transform blocks end before the block does (7.11.5 MaxLumaW / MaxLumaH:
                                       transform blocks that START beyond the coded size are not listed).  The finished picture beyond
                                       the coded size is undefined */
    uint32_t num_blocks;
    uint32_t num_coefs;             /* int16 entries in the coefficient arena               */
    uint32_t num_tx;                /* entries of the transform-block list                  */
    uint32_t num_aux;               /* bytes of the aux arena (palettes)                    */
    uint32_t upscaled_width;        /* super-resolution (AV1 spec 7.16): 0 = none; else the frame above is the DOWNSCALED one --
                                       reconstructed, deblocked and CDEF-filtered at that size -- and the picture that comes out
                                       (and is kept as a reference) is upscaled_width visible samples wide, inside
                                       (upscaled_width + 7) & ~7 stored ones: both filtered pictures are upscaled horizontally
                                       with the Upscale_Filter table of the context (ea1_recon_set_upscale_filter), restoration
                                       runs on the upscaled pictures (its units are counted on that width) */
    uint8_t layout;                 /* EA1_LAYOUT_*: the chroma layout of the frame = the layout of the context that reconstructs it
                                       (ea1_recon_create_layout).  Everything above that says "chroma: half" describes EA1_LAYOUT_420 (0,
                                       which also carries 4:0:0: grey chroma planes); with EA1_LAYOUT_444 a chroma plane has the luma
                                       plane's geometry, with EA1_LAYOUT_422 half its width and all of its height: block and transform
                                       positions of the chroma planes, the chroma of blocks with a 4-sample side (owned by the last block
                                       of the pair in a subsampled direction only), restoration units and palette maps follow
                                       subsampling_x / subsampling_y of the AV1 specification throughout */
    uint8_t reserved[3];
} ea1_frame_desc;

/* chroma layouts (8 bit): easyav1_pixel_layout of the reference minus 4:0:0 (/root/reference/src/easyav1.h:183-188, set from dav1d's
 * Dav1dPixelLayout at /root/reference/src/easyav1.c:3287-3311) */
#define EA1_LAYOUT_420 0
#define EA1_LAYOUT_444 1
#define EA1_LAYOUT_422 2

/* device view of a finished picture (dav1d's Dav1dPicture data/stride, easyav1.c:3557-3563) */
typedef struct ea1_picture {
    const uint8_t *data[3];  /* device pointers                                            */
    size_t stride[3];        /* bytes                                                      */
    uint32_t width, height;
} ea1_picture;

typedef struct ea1_recon ea1_recon; /* decoder context: reference slots + work buffers     */
typedef struct ea1_job ea1_job;     /* one parsed frame, resident in HBM                   */

int ea1_recon_create(ea1_recon **out, uint32_t width, uint32_t height, int device);
/*
 * A context with `nchains` (1..256) independent decoding chains.  Closed GOPs are independent
 * units: the shim hands GOP k to chain k % nchains and every kernel launch of
 * ea1_recon_submit_batch serves one frame of every chain, so a launch carries nchains frames of
 * work instead of one (a single 1080p frame is ~2 us of HBM traffic: not worth a launch).
 * ea1_recon_create == ea1_recon_create_batch(.., 1, ..); the un-suffixed calls address chain 0.
 */
int ea1_recon_create_batch(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device);
/* ... whose pictures have the chroma layout `layout` (EA1_LAYOUT_*; ea1_recon_create_batch: EA1_LAYOUT_420).  Every frame handed to the
 * context states the same layout (ea1_frame_desc.layout), every picture that goes in or comes out has chroma planes of
 * Round2(width, subsampling_x) x Round2(height, subsampling_y) samples.  A tile-column split (ea1_job_create_strip) is 4:2:0 only. */
int ea1_recon_create_layout(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device, int layout);
int ea1_recon_layout(const ea1_recon *r);
int ea1_recon_num_chains(const ea1_recon *r);
void ea1_recon_destroy(ea1_recon *r);
/* forget all reference slots (seek) */
int ea1_recon_flush(ea1_recon *r);

/*
 * The warp interpolation filters (AV1 spec: Warped_Filters[193][8], dav1d: dav1d_mc_warp_filter):
 * 193 rows of 8 signed taps that sum to 128, row = 64 + position in 1/64 sample (-1 .. 2), taps in
 * specification order (tap k weighs the sample at offset k - 3 from the integer position).  The
 * table is data of the AV1 specification (the reference links it inside dav1d, cmake/dav1d.cmake:994
 * src/warpmv.c + tables.c).  Every context starts with the library's own copy (include/ea1_av1_tables.inc,
 * EA1_WARPED_FILTERS_VALUES: written down from knowledge of the specification and cross-checked -- rows
 * sum to 128, mirror symmetry inside each segment, the third segment is the first moved two taps -- but,
 * like every table of the video path here, not pinned against a dav1d build); this call replaces it
 * (a front end that owns the normative table).  `filter` is host memory, 193 * 8 bytes, copied.
 */
#define EA1_WARP_FILTER_ROWS 193
int ea1_recon_set_warp_filter(ea1_recon *r, const int8_t *filter);

/* Load a reference slot from planar host memory (seeds the synthetic streams' first references
 * and lets tests inject arbitrary references). */
int ea1_recon_set_ref(ea1_recon *r, int slot, const uint8_t *y, size_t y_stride, const uint8_t *u,
                      const uint8_t *v, size_t c_stride);
int ea1_recon_set_ref_chain(ea1_recon *r, int chain, int slot, const uint8_t *y, size_t y_stride, const uint8_t *u,
                            const uint8_t *v, size_t c_stride);
/*
 * Frame sizes and scaled references (AV1 spec 7.11.3.3 / 7.11.3.4; dav1d: the `scaled` variants of mc_tmpl.c,
 * cmake/dav1d.cmake:1013).  A context decodes frames of ANY size up to its own (ea1_frame_desc.width / height,
 * multiples of 8; all frames of one submit alike), every reference slot remembers the size of the picture it holds, and
 * an inter frame may predict from a slot of another size within the specification's limits (the reference at most twice
 * as large and at most 16 times smaller per direction): positions step through the reference in 1/1024 sample units
 * from the block's corner, xStep = Round2(xScale, 4), with the sub-pel filters picked per sample.  Covered on that path:
 * every translated prediction -- blocks of any size, every compound kind, OBMC, the inter half of inter-intra blocks;
 * a frame that combines a scaled reference with warped motion (which a valid stream never does: 7.11.3.1) is refused at
 * submit.  Super-resolution (7.16) is not implemented.
 * ea1_recon_set_ref_sized loads a picture of width x height (<= the context's) into a slot; ea1_recon_slot_size tells
 * the coded size of the picture a slot holds (downloads and ea1_recon_get_picture use it).
 */
int ea1_recon_set_ref_sized(ea1_recon *r, int chain, int slot, uint32_t width, uint32_t height, const uint8_t *y, size_t y_stride,
                            const uint8_t *u, const uint8_t *v, size_t c_stride);
int ea1_recon_slot_size(const ea1_recon *r, int chain, int slot, uint32_t *width, uint32_t *height);
/* The visible size of a picture loaded with ea1_recon_set_ref*: crop as in ea1_frame_desc.crop (pictures a submit stores carry
 * their frame's crop by themselves).  Scale factors, the "is this reference scaled" decision and the 2 : 1 / 1 : 16 limits all
 * use the VISIBLE sizes (7.11.3.3: RefUpscaledWidth / FrameWidth are the pictures' true sizes). */
int ea1_recon_set_ref_crop(ea1_recon *r, int chain, int slot, uint8_t crop);
/* Super-resolution (ea1_frame_desc.upscaled_width): Upscale_Filter[64][8] of the specification (7.16; dav1d: dav1d_resize_filter),
 * host memory, copied.  The library does not own the table: a table blob carries it (easyav1_amd_obu.h, kind 5).  Rows sum to 128;
 * an all-zero table leaves the context without one (frames with upscaled_width are then refused).  Replaces what dav1d's
 * resize / super-resolution stage does between CDEF and loop restoration (/root/reference/cmake/dav1d.cmake:1013, mc_tmpl.c resize). */
int ea1_recon_set_upscale_filter(ea1_recon *r, const int16_t *taps);

/* The same for a picture that carries padding (ea1_frame_desc.pad: its true coded size lies inside the loaded width x height): visible
 * size = size - 8 * pad - crop per direction */
int ea1_recon_set_ref_pad(ea1_recon *r, int chain, int slot, uint8_t pad, uint8_t crop);

/*
 * Hand over one parsed frame: copies the description to the device and derives the schedule
 * (per-size transform lists, the intra wavefront levels, the block and edge maps).  All pointers
 * are host memory.
 *   txs        desc->num_tx transform blocks (ea1_block.first_tx / n_tx index into it)
 *   aux        desc->num_aux bytes of palette data (may be NULL when num_aux == 0)
 *   cdef_idx   one byte per 64x64 superblock, raster order: strength index 0..7, 0xFF = none
 *   lr[p]      restoration units of plane p, raster order, (may be NULL when lr_type[p] == 0)
 */
int ea1_job_create(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks,
                   const ea1_txblock *txs, const int16_t *coefs, const uint8_t *aux,
                   const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3], ea1_job **out);
/* Cost model of the hand-over: validation and the work lists are host work of this call;
 * the records and coefficients travel on the context's own upload stream and the deblock / CDEF side information is
 * derived from them by kernels on that stream, so the call returns while that is in flight.  Submits order
 * themselves behind it on the device.  The input arrays may be reused as soon as the call returns -- unless they
 * are pinned memory (ea1_host_alloc), which is copied by DMA with no staging: keep those until ea1_job_wait.
 * ea1_job_destroy never synchronises: the job's memory returns to a pool once its last submit has finished. */
int ea1_job_wait(ea1_job *job);
void ea1_job_destroy(ea1_job *job);
/* algorithmic bytes of the job as SURVEY.md section 8(d) defines them: P*(1+r) + 2*C + M */
int ea1_job_stats(const ea1_job *job, double *ref_pixels_per_pixel, uint64_t *coef_count,
                  uint64_t *block_bytes, uint64_t *algorithmic_bytes);

/* shape of the derived schedule (diagnostics): wavefront levels, widest level, intra items,
 * inter tiles per size class (8/16/32) and transform blocks per size (EA1_NUM_TX_SIZES entries,
 * index = ea1_tx_size_index(log2 w, log2 h)) */
#define EA1_NUM_TX_SIZES 19
/* 4x4 8x8 16x16 32x32 64x64 | 4x8 8x4 8x16 16x8 16x32 32x16 32x64 64x32 | 4x16 16x4 8x32 32x8 16x64 64x16; -1 = not a size */
int ea1_tx_size_index(int log2w, int log2h);
int ea1_job_schedule_info(const ea1_job *job, int *num_levels, int *max_level_items, int *intra_items,
                          int *mc_tiles, int *tx_blocks);

/* Reconstruct + loop-filter + CDEF + restore one frame into its dst_slot; asynchronous. */
int ea1_recon_submit(ea1_recon *r, const ea1_job *job, void *stream);

/* stage mask for ea1_recon_submit_stages: lets tests observe intermediate pictures */
#define EA1_STAGE_RECON 1
#define EA1_STAGE_DEBLOCK 2
#define EA1_STAGE_CDEF 4
#define EA1_STAGE_LR 8
#define EA1_STAGE_ALL 15
int ea1_recon_submit_stages(ea1_recon *r, const ea1_job *job, unsigned stages, void *stream);
/* jobs: ea1_recon_num_chains() entries; jobs[c] is reconstructed on chain c, NULL = chain idle */
int ea1_recon_submit_batch(ea1_recon *r, ea1_job *const *jobs, unsigned stages, void *stream);

/*
 * Tile-column split of ONE stream over several GPUs (BASELINE.json configs[3]).
 * The reference is one process with dav1d's tile threads inside it (src/easyav1.c:1401-1405); here every
 * rank (one process per GPU) owns a contiguous range of tile columns, [x_begin, x_end) in luma samples,
 * superblock aligned, of every frame:
 *
 *   1. every rank receives the whole parsed frame and makes its job with ea1_job_create_strip: the work
 *      lists (prediction, transforms, intra wavefront) hold the strip's own blocks only and only their
 *      coefficients are uploaded; the deblock / CDEF / restoration side information covers the frame
 *   2. ea1_recon_submit_phase(EA1_PHASE_RECON): reconstruction of the strip into the chain's work picture
 *      (inter prediction reads the reference slots, which every rank holds in full)
 *   3. halo exchange: 16 luma (8 chroma) columns of reconstruction on either side of the strip come from the
 *      neighbouring ranks: ea1_recon_get_columns(EA1_PIC_WORK) of the own edge columns -> collective ->
 *      ea1_recon_put_columns(EA1_PIC_WORK) of the neighbours' columns
 *   4. ea1_recon_submit_phase(EA1_PHASE_FILTER): deblock + CDEF on [x_begin - 16, x_end + 16), restoration on
 *      the strip, into the frame's reference slot.  Why 16 columns suffice: a deblocked sample depends on the
 *      one vertical edge within 6 samples of it and on the 7 reconstructed samples either side of that edge,
 *      so deblocking is exact from x_begin - 8 on; CDEF needs the whole 8x8 block of a sample for its
 *      direction and reaches 2 samples beyond it; restoration reads 3 columns of CDEF output beside the strip
 *   5. all-gather of the finished strips: ea1_recon_get_columns(dst_slot) -> collective ->
 *      ea1_recon_put_columns(dst_slot) of the other ranks' strips: every rank holds the whole reference
 * The collectives themselves are the host side's (torch.distributed: RCCL over xGMI, or gloo staged through
 * host memory in the tests): easyav1_amd/parallel.py.  Buffers are DEVICE memory, ea1_recon_columns_bytes(w)
 * bytes for w columns: Y rows of w bytes, then U rows of w / 2, then V rows of w / 2.
 */
int ea1_job_create_strip(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                         const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx,
                         const ea1_lr_unit *const lr[3], uint32_t x_begin, uint32_t x_end, ea1_job **out);
#define EA1_PHASE_RECON 1
#define EA1_PHASE_FILTER 2
/* jobs: one per chain as for ea1_recon_submit_batch; all of the same strip */
int ea1_recon_submit_phase(ea1_recon *r, ea1_job *const *jobs, unsigned phase, void *stream);
#define EA1_PIC_WORK (-1)   /* the reconstruction before the in-loop filters; 0..7: a reference slot */
size_t ea1_recon_columns_bytes(const ea1_recon *r, uint32_t w);
int ea1_recon_get_columns(ea1_recon *r, int chain, int pic, uint32_t x, uint32_t w, void *d_buf, void *stream);
int ea1_recon_put_columns(ea1_recon *r, int chain, int pic, uint32_t x, uint32_t w, const void *d_buf, void *stream);

int ea1_recon_get_picture(ea1_recon *r, int slot, ea1_picture *out);
int ea1_recon_get_picture_chain(ea1_recon *r, int chain, int slot, ea1_picture *out);
int ea1_recon_download_chain(ea1_recon *r, int chain, int slot, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                             size_t c_stride, void *stream);
/*
 * Delivery path of a player that keeps many GOPs in flight: one picture of every chain to host
 * memory WITHOUT synchronising.  A picture travels in its device layout as ONE copy per chain
 * (three pitched plane copies per chain cost more host time than the reconstruction of the frame):
 * ea1_recon_picture_layout() tells where the planes lie inside the `bytes`-sized block and their
 * strides -- what easyav1_video_frame.data[] / stride[] point at (src/easyav1.c:3557-3568).
 * slots[c] < 0: chain skipped; dst[c]: `bytes` bytes of host memory, pinned (hipHostMalloc) for
 * truly asynchronous copies -- or of DEVICE memory (a player that parks finished pictures of GOPs decoded
 * ahead in HBM and sends them to the host in presentation order: libeasyav1.so's video sink).  The caller orders the copies against the submits (events) and waits
 * for `stream` before reading.  PCIe-inclusive rate: DESIGN.md section 5.
 */
/* pinned host memory for those copies (hipHostMalloc / hipHostFree of the library's own HIP runtime) */
int ea1_host_alloc(size_t bytes, void **out);
void ea1_host_free(void *p);
int ea1_recon_picture_layout(const ea1_recon *r, size_t plane_offset[3], size_t plane_stride[3], size_t *bytes);
int ea1_recon_download_batch_async(ea1_recon *r, const int *slots, uint8_t *const *dst, void *stream);
/* copy a slot to planar host memory (D2H on `stream`, synchronised before return) */
int ea1_recon_download(ea1_recon *r, int slot, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                       size_t c_stride, void *stream);

/*
 * Per-kernel timing for the roofline report: when enabled, every kernel launch of a submit is
 * bracketed by HIP events on the launch stream.  _read() synchronises, aggregates by kernel name
 * (the names rocprofv3 shows, template arguments included) and resets.  `algorithmic_bytes` is
 * the per-launch figure of DESIGN.md summed over the launches.
 */
typedef struct ea1_prof_entry {
    char name[56];
    double ms_total;
    uint64_t launches;
    uint64_t algorithmic_bytes;
} ea1_prof_entry;
int ea1_recon_profile_enable(ea1_recon *r, int on);

/*
 * Warms the reconstruction path on `device`: decodes three tiny generated frames (every coding tool) on a context of its own and throws
 * them away.  The HIP runtime loads a code object and sets every kernel up at its first launch -- ~230 ms on the critical path of a stream's
 * first picture otherwise; a decoder calls this from a spare thread while it parses its first frame (dav1d_open has no counterpart: its
 * code is mapped with the process).  Optional; any thread; idempotent in effect.
 */
int ea1_recon_warm(int device);
int ea1_recon_profile_read(ea1_recon *r, ea1_prof_entry *out, int cap, int *count);

/*
 * DSP-level entry (dav1d's itxfm_add(dst, stride, coeff, eob) in batch form): adds the inverse
 * transform of `count` blocks of one size to a device picture plane.
 *   d_coefs  int16, per block min(w,32)*min(h,32) entries, row-major, blocks back to back
 *   pos      host, count * 2 uint32: x, y of each block in pixels
 *   txtp     host, count bytes, EA1_TXTP(col, row)
 */
int ea1_dsp_itx_add(uint8_t *d_plane, size_t stride, int w, int h, const int16_t *d_coefs,
                    const uint32_t *pos, const uint8_t *txtp, int count, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif
