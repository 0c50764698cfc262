/*
 * easyav1_amd_grain.h -- C ABI of film grain synthesis.
 *
 * dav1d applies film grain to the OUTPUT copy of a picture (the reference pictures stay free of
 * grain): src/fg_apply_tmpl.c + src/filmgrain_tmpl.c (/root/reference/cmake/dav1d.cmake:1004,:1005),
 * default Dav1dSettings.apply_grain = 1, which easyav1 leaves alone (src/easyav1.c:1401-1403); the
 * picture easyav1 hands out (src/easyav1.c:3557-3568) therefore carries the grain.  Here the same
 * step is one GPU pass per picture: reference slot -> the chain's display picture.
 *
 * Split of the work (AV1 specification 7.18.3):
 *   host, per picture (ea1_recon_film_grain_*):  the pseudo-random grain templates (73 x 82 luma,
 *       38 x 44 per chroma plane for 4:2:0) with their auto-regressive filter (7.18.3.3), the three
 *       256-entry scaling look-up tables (7.18.3.4) and one random offset per 32 x 32 luma block
 *       (7.18.3.5) -- a few hundred microseconds of strictly sequential integer work
 *   GPU: per sample, the noise value from the template at the block's offset with the horizontal
 *       and vertical overlap blends, scaled by the look-up table of the (merged) sample value, added
 *       and clipped: picture read once, written once (HBM bound: 3 bytes of traffic per sample).
 *
 * The Gaussian sequence (2048 entries) is specification data that is recall-only here, like the warp
 * filter table: it is handed over once per context, ea1_recon_set_gaussian_sequence(); pictures are
 * refused until then.  8-bit 4:2:0 as the rest of the path.  Parity: pinned against dav1d 1.5.3's apply_grain on libaom streams (tests/test_real_av1.py).
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_GRAIN_H
#define EASYAV1_AMD_GRAIN_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_video.h"

#ifdef __cplusplus
extern "C" {
#endif

/* film_grain_params() of the frame header (AV1 specification 5.9.30 / 6.8.20), values as coded
 * unless noted; dav1d: Dav1dFilmGrainData */
typedef struct ea1_film_grain {
    uint16_t grain_seed;
    uint8_t num_y_points;            /* 0..14 */
    uint8_t point_y[14][2];          /* [value, scaling], values increasing */
    uint8_t chroma_scaling_from_luma;
    uint8_t num_uv_points[2];        /* 0..10 (cb, cr) */
    uint8_t point_uv[2][10][2];
    uint8_t scaling_shift;           /* grain_scaling_minus_8 + 8: 8..11 */
    uint8_t ar_coeff_lag;            /* 0..3 */
    int8_t ar_coeffs_y[24];          /* ar_coeffs_y_plus_128 - 128; 2 * lag * (lag + 1) used */
    int8_t ar_coeffs_uv[2][25];      /* one more than luma: the last one weighs the luma grain */
    uint8_t ar_coeff_shift;          /* ar_coeff_shift_minus_6 + 6: 6..9 */
    uint8_t grain_scale_shift;       /* 0..3 */
    uint8_t uv_mult[2];              /* cb_mult, cr_mult as coded (128 = 0) */
    uint8_t uv_luma_mult[2];
    uint16_t uv_offset[2];           /* as coded, 0..511 (256 = 0) */
    uint8_t overlap_flag;
    uint8_t clip_to_restricted_range;
    uint8_t matrix_is_identity;      /* sequence header: matrix_coefficients == MC_IDENTITY (chroma range = luma range) */
    uint8_t crop;                    /* visible size as ea1_frame_desc.crop: grain is synthesised for the visible
                                        picture (noise stripes over its width / height), samples beyond it are copied */
} ea1_film_grain;

#define EA1_GAUSSIAN_SEQUENCE_SIZE 2048
/* Gaussian_Sequence[2048] of the specification (dav1d: dav1d_gaussian_sequence), host memory, copied */
int ea1_recon_set_gaussian_sequence(ea1_recon *r, const int16_t *sequence);

/*
 * Apply film grain to the picture in `slots[c]` of every chain c (slots[c] < 0 or params[c] == NULL:
 * chain idle) and leave the result in that chain's display picture; one kernel launch for all
 * chains, asynchronous on `stream` (the host-side template generation happens inside the call).
 * params[c]->num_y_points == 0 with no chroma points copies the picture unchanged.
 */
int ea1_recon_film_grain_batch(ea1_recon *r, const int *slots, const ea1_film_grain *const *params, void *stream);
/* one chain */
int ea1_recon_film_grain_chain(ea1_recon *r, int chain, int slot, const ea1_film_grain *params, void *stream);

/* the display picture of a chain: device view / copy to planar host memory (synchronises `stream`) */
int ea1_recon_get_display_chain(ea1_recon *r, int chain, ea1_picture *out);
int ea1_recon_download_display_chain(ea1_recon *r, int chain, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                                     size_t c_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif
