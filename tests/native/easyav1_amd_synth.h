/*
 * easyav1_amd_synth.h -- seeded synthetic "parsed frame" streams (the ReconJob IR of
 * easyav1_amd_video.h), standing in for the host-side OBU/entropy parse that north_star keeps on
 * the CPU.  There is no AV1 encoder, sample media or default-CDF table in this environment
 * (SURVEY.md section 0 findings 2 and 10), so benchmark and parity inputs are generated at the
 * hand-over point: exactly the data dav1d's parse pass leaves for its reconstruction pass.
 *
 * Distributions follow SURVEY.md section 8(d): GOP of `gop` frames (1 key + inter), random
 * quad-tree partition of 64x64 superblocks down to 8x8 with P(split) = 1/2, inter frames 90 %
 * inter blocks (70 % single reference / 30 % averaged compound), motion vectors uniform in
 * +-64 px at 1/8 pel, regular/smooth/sharp filters, intra DC/V/H/PAETH, transform
 * types from {DCT, ADST, identity} combinations legal for the size, 30 % of transform blocks
 * all-zero, Laplacian(b = 20) coefficients on a geometric number of low-frequency positions,
 * deblock levels Y 20 / UV 12, CDEF damping 5 with eight strength pairs, luma Wiener + chroma
 * self-guided (set 5) restoration.
 *
 * Host-only code; deterministic for a given (params, frame_index) on every platform.
 */
#ifndef EASYAV1_AMD_SYNTH_H
#define EASYAV1_AMD_SYNTH_H

#include "easyav1_amd_video.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ea1_synth_params.features: coding tools drawn in addition to the graded set of SURVEY.md 8(d)
 * ("stress sets").  0 = the graded set: square blocks, one transform per block and plane. */
#define EA1_SYN_RECT 0x001         /* all partition shapes: horz / vert / 4-way / T shapes        */
#define EA1_SYN_TX_SPLIT 0x002     /* transform sizes one or two steps below the largest          */
#define EA1_SYN_ALL_TXTP 0x004     /* all 16 1-D kind pairs legal for the size (FLIPADST, V/H id) */
#define EA1_SYN_SMOOTH 0x008       /* SMOOTH / SMOOTH_V / SMOOTH_H intra modes                    */
#define EA1_SYN_DIRECTIONAL 0x010  /* directional modes with angle delta, intra edge filter on    */
#define EA1_SYN_CFL 0x020          /* chroma from luma                                            */
#define EA1_SYN_FILTER_INTRA 0x040 /* recursive filter intra                                      */
#define EA1_SYN_PALETTE 0x080      /* palette blocks                                              */
#define EA1_SYN_COMPOUND 0x100     /* distance-weighted, wedge and difference-weighted compound   */
#define EA1_SYN_OBMC 0x200         /* overlapped block motion compensation                        */
#define EA1_SYN_INTERINTRA 0x400   /* inter-intra (smooth and wedge)                              */
#define EA1_SYN_BILINEAR 0x800     /* bilinear interpolation filter                               */
#define EA1_SYN_LF_DELTA 0x1000    /* per-block deblocking levels (delta_lf / ref / mode deltas)  */
#define EA1_SYN_SUB8 0x2000        /* blocks with 4-sample luma sides (4x4, 4x8, 8x4, 4x16, 16x4) */
#define EA1_SYN_LR_MIX 0x4000      /* restoration type drawn per unit and plane: none / Wiener (chroma
                                      5-tap) / self-guided with any of the 16 parameter sets       */
#define EA1_SYN_ALL 0x7FFF
#define EA1_SYN_LOSSLESS 0x8000    /* lossless frames: every transform a 4x4 Walsh-Hadamard, no loop filters
                                      (not part of EA1_SYN_ALL: it switches the other transform tools off) */

#define EA1_SYN_WARP 0x10000       /* warped motion on blocks of 8x8 and more, some matrices failing the
                                      shear test (not part of EA1_SYN_ALL: the context needs a filter
                                      table first, ea1_recon_set_warp_filter)                       */

#define EA1_SYN_INTRABC 0x20000    /* intra block copy in the intra frames, which then carry no in-loop
                                      filters (not part of EA1_SYN_ALL)                             */

#define EA1_SYN_NO_NZ_HINT 0x40000  /* leave ea1_txblock.nz_log2 at 0: the hand-over finds the extent of the
                                      coefficients itself (default: the generator states it, as a parser would) */

#define EA1_SYN_COMPACT_COEFS 0x80000  /* coefficient arena in EA1_COEFS_CORNER layout (ea1_synth_default_params sets
                                         it: what a parser hands over; streams made with an explicit feature set
                                         without it use the dense layout) */

#define EA1_SYN_DIAGONAL_COEFS 0x100000  /* coefficient arena in EA1_COEFS_DIAGONAL layout (wins over EA1_SYN_COMPACT_COEFS;
                                           ea1_synth_default_params sets it) */

typedef struct ea1_synth_params {
    uint32_t width, height;   /* multiples of 8                                             */
    uint32_t tile_cols;       /* uniform tile columns (1, 4, 8, ...)                        */
    uint32_t gop;             /* frames per GOP; frame_index % gop == 0 is a key frame      */
    uint64_t seed;
    uint32_t features;        /* EA1_SYN_*                                                  */
    uint8_t lr_unit_log2;     /* luma restoration unit (6..8); chroma uses one less         */
    uint8_t enable_deblock, enable_cdef, enable_lr;
    uint32_t tile_rows;       /* uniform tile rows (1, 2, 4, ...)                           */
    uint8_t crop_right, crop_bottom;   /* 0..7: the visible picture is that much smaller than width x height */
    uint8_t pad[2];
} ea1_synth_params;

typedef struct ea1_synth_frame {
    ea1_frame_desc desc;
    ea1_block *blocks;        /* desc.num_blocks                                            */
    ea1_txblock *txs;         /* desc.num_tx                                                */
    int16_t *coefs;           /* desc.num_coefs                                             */
    uint8_t *aux;             /* desc.num_aux                                               */
    uint8_t *cdef_idx;        /* num_sb                                                     */
    ea1_lr_unit *lr[3];       /* lr_count[p] units per plane                                */
    uint32_t lr_count[3];
    uint32_t num_sb;
} ea1_synth_frame;

/* defaults of SURVEY.md section 8(d) for a picture size (unit 64 up to 1080p, 128 above) */
void ea1_synth_default_params(ea1_synth_params *p, uint32_t width, uint32_t height, uint32_t tile_cols,
                              uint64_t seed);
int ea1_synth_frame_create(const ea1_synth_params *p, uint32_t frame_index, ea1_synth_frame **out);
void ea1_synth_frame_destroy(ea1_synth_frame *f);
/* seeded noise + gradients picture (tight planes): initial reference content */
void ea1_synth_fill_picture(uint32_t width, uint32_t height, uint64_t seed, uint8_t *y, uint8_t *u, uint8_t *v);

/*
 * A STAND-IN for the warp filter table, for tests and benchmarks of the warp kernel: 193 rows of 8
 * taps with the structure of the normative table (rows sum to 128, row = 64 + position in 1/64
 * sample, all taps in int8), made from a 4-tap Catmull-Rom interpolator in integer arithmetic.  It
 * is NOT the AV1 table (that one is specification data which neither this repository nor the
 * reference tree holds); the kernel and the oracle take whatever table they are given.
 */
void ea1_synth_warp_filter(int8_t filter[193 * 8]);
/* likewise a STAND-IN for Gaussian_Sequence[2048] of the film grain synthesis (12-bit, bell shaped) */
void ea1_synth_gaussian_sequence(int16_t sequence[2048]);

#ifdef __cplusplus
}
#endif
#endif
