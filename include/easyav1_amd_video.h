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
 * Scope of this IR revision (documented restrictions): 8-bit 4:2:0; square coded
 * blocks 8x8..64x64, one transform per block and plane (luma tx = block, chroma tx = block/2);
 * intra modes DC/V/H/PAETH/SMOOTH*; single-reference or averaged compound inter prediction with
 * the regular/smooth/sharp 8-tap filters; no OBMC/warp/palette/CfL/intrabc/film grain/super-res.
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

#define EA1_NUM_REF_SLOTS 8

/* 1-D transform kinds; a 2-D transform type is (col_kind << 2) | row_kind */
enum { EA1_TX_DCT = 0, EA1_TX_ADST = 1, EA1_TX_FLIPADST = 2, EA1_TX_IDENTITY = 3 };
#define EA1_TXTP(col_kind, row_kind) (((col_kind) << 2) | (row_kind))

/* intra prediction modes */
enum {
    EA1_DC_PRED = 0, EA1_V_PRED = 1, EA1_H_PRED = 2, EA1_PAETH_PRED = 3,
    EA1_SMOOTH_PRED = 4, EA1_SMOOTH_V_PRED = 5, EA1_SMOOTH_H_PRED = 6,
    EA1_N_INTRA_MODES = 7
};

/* interpolation filter kinds (per direction) */
enum { EA1_FILTER_REGULAR = 0, EA1_FILTER_SMOOTH = 1, EA1_FILTER_SHARP = 2 };

/* ea1_block.flags */
#define EA1_BLK_INTER 0x01      /* inter-predicted (else intra)                              */
#define EA1_BLK_COMPOUND 0x02   /* two references, rounded average                           */
#define EA1_BLK_SKIP 0x04       /* no residual in any plane (drives CDEF's skip condition)   */
#define EA1_BLK_HAVE_LEFT 0x08  /* left neighbour is inside the tile/frame (intra edges)      */
#define EA1_BLK_HAVE_TOP 0x10   /* top neighbour is inside the tile/frame                    */
/* ea1_block.cf_flags: bit p set = plane p has coefficients stored in the arena */

/* One coded block: luma w x h = (1 << bwl) x (1 << bhl) at (4*x4, 4*y4); chroma is the
 * co-located (w/2) x (h/2).  32 bytes. */
typedef struct ea1_block {
    uint16_t x4, y4;     /* luma position in 4-pixel units                                  */
    uint8_t bwl, bhl;    /* log2 of luma width / height in pixels (3..6; bwl == bhl for now) */
    uint8_t flags;       /* EA1_BLK_*                                                       */
    uint8_t mode_y;      /* intra: luma mode; inter: filter_h | filter_v << 2               */
    uint8_t mode_uv;     /* intra: chroma mode (inter: unused)                              */
    uint8_t ref[2];      /* reference slots (0..7) of the two predictions                   */
    uint8_t txtp[3];     /* transform type per plane, EA1_TXTP(col, row)                    */
    uint8_t cf_flags;    /* which planes carry coefficients                                 */
    uint8_t pad0;
    int16_t mv[2][2];    /* [ref][0 = y, 1 = x] in 1/8 luma pixel                           */
    uint32_t coef_off;   /* int16 index into the coefficient arena: planes with coefficients
                            follow each other (Y, U, V), each min(w,32) x min(h,32), row-major */
    uint32_t pad1;
} ea1_block;

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
    uint8_t pad[3];
    uint32_t num_blocks;
    uint32_t num_coefs;             /* int16 entries in the arena                           */
} ea1_frame_desc;

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
 * A context with `nchains` (1..64) independent decoding chains.  Closed GOPs are independent
 * units: the shim hands GOP k to chain k % nchains and every kernel launch of
 * ea1_recon_submit_batch serves one frame of every chain, so a launch carries nchains frames of
 * work instead of one (a single 1080p frame is ~2 us of HBM traffic: not worth a launch).
 * ea1_recon_create == ea1_recon_create_batch(.., 1, ..); the un-suffixed calls address chain 0.
 */
int ea1_recon_create_batch(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device);
int ea1_recon_num_chains(const ea1_recon *r);
void ea1_recon_destroy(ea1_recon *r);
/* forget all reference slots (seek) */
int ea1_recon_flush(ea1_recon *r);

/* Load a reference slot from planar host memory (seeds the synthetic streams' first references
 * and lets tests inject arbitrary references). */
int ea1_recon_set_ref(ea1_recon *r, int slot, const uint8_t *y, size_t y_stride, const uint8_t *u,
                      const uint8_t *v, size_t c_stride);
int ea1_recon_set_ref_chain(ea1_recon *r, int chain, int slot, const uint8_t *y, size_t y_stride, const uint8_t *u,
                            const uint8_t *v, size_t c_stride);

/*
 * Hand over one parsed frame: copies the description to the device and derives the schedule
 * (per-size transform lists, the intra wavefront levels, the 8x8 block map).  All pointers are
 * host memory.
 *   cdef_idx   one byte per 64x64 superblock, raster order: strength index 0..7, 0xFF = none
 *   lr[p]      restoration units of plane p, raster order, (may be NULL when lr_type[p] == 0)
 */
int ea1_job_create(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks,
                   const int16_t *coefs, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3],
                   ea1_job **out);
void ea1_job_destroy(ea1_job *job);
/* algorithmic bytes of the job as SURVEY.md section 8(d) defines them: P*(1+r) + 2*C + M */
int ea1_job_stats(const ea1_job *job, double *ref_pixels_per_pixel, uint64_t *coef_count,
                  uint64_t *block_bytes, uint64_t *algorithmic_bytes);

/* shape of the derived schedule (diagnostics): wavefront levels, widest level, intra items,
 * inter tiles per size class (8/16/32) and transform blocks per size class (4..64) */
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

int ea1_recon_get_picture(ea1_recon *r, int slot, ea1_picture *out);
int ea1_recon_get_picture_chain(ea1_recon *r, int chain, int slot, ea1_picture *out);
int ea1_recon_download_chain(ea1_recon *r, int chain, int slot, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                             size_t c_stride, void *stream);
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
