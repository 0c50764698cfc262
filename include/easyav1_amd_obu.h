/*
 * easyav1_amd_obu.h -- C ABI of the table-free part of the AV1 bitstream front end (SURVEY.md
 * section 8(f) row N1, first slice): OBU framing and the sequence header.
 *
 * easyav1 asks dav1d for exactly this before it decodes anything: dav1d_parse_sequence_header
 * (/root/reference/src/easyav1.c:2339, seek_sequence_header: while seeking it scans packets until
 * one carries a sequence header) and reads the header's fields when it hands out frames
 * (src/easyav1.c:3287-3507: layout, hbd, color_range, pri, trc, mtrx, chr).  Both are plain
 * fixed-length-code parsing (AV1 specification 5.3 OBU syntax, 5.5 sequence header OBU, 4.10.5
 * leb128): none of the recall-only entropy-coding tables that block the rest of N1 (default CDFs,
 * scan orders) is involved.  Host code by design (BASELINE.json north_star keeps bitstream parsing
 * on the CPU); not a fallback for anything the GPU path does.
 *
 * Parity: UNPINNED (dav1d is absent); checked against an independent OBU writer in tests/.
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_OBU_H
#define EASYAV1_AMD_OBU_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_audio.h"

#ifdef __cplusplus
extern "C" {
#endif

/* obu_type (AV1 specification 6.2.2) */
enum {
    EA1_OBU_SEQUENCE_HEADER = 1, EA1_OBU_TEMPORAL_DELIMITER = 2, EA1_OBU_FRAME_HEADER = 3, EA1_OBU_TILE_GROUP = 4,
    EA1_OBU_METADATA = 5, EA1_OBU_FRAME = 6, EA1_OBU_REDUNDANT_FRAME_HEADER = 7, EA1_OBU_TILE_LIST = 8, EA1_OBU_PADDING = 15
};

typedef struct ea1_obu {
    uint8_t type;            /* obu_type */
    uint8_t has_extension;
    uint8_t temporal_id, spatial_id;
    uint8_t has_size_field;
    uint8_t pad[3];
    const uint8_t *payload;  /* inside the caller's buffer */
    size_t payload_size;
    size_t total_size;       /* header + size field + payload: offset of the next OBU */
} ea1_obu;

/*
 * The OBU at the start of `data` (low-overhead format: what WebM blocks and CodecPrivate carry).  An
 * OBU without a size field extends to the end of the buffer.  EA1_EINVAL: forbidden bit set,
 * truncated header / size / payload.
 */
int ea1_av1_next_obu(const uint8_t *data, size_t size, ea1_obu *out);

/* pixel layouts and bit depths with dav1d's numbering (src/easyav1.c:3296-3322 switches on them) */
enum { EA1_LAYOUT_I400 = 0, EA1_LAYOUT_I420 = 1, EA1_LAYOUT_I422 = 2, EA1_LAYOUT_I444 = 3 };

#define EA1_MAX_OPERATING_POINTS 32
typedef struct ea1_av1_sequence_header {
    uint8_t profile, still_picture, reduced_still_picture_header;
    uint8_t layout;              /* EA1_LAYOUT_* */
    uint8_t hbd;                 /* 0: 8 bit, 1: 10 bit, 2: 12 bit (Dav1dSequenceHeader.hbd) */
    uint8_t bit_depth;           /* 8 / 10 / 12 */
    uint8_t monochrome, subsampling_x, subsampling_y, chroma_sample_position;
    uint8_t color_description_present, color_range;
    uint8_t color_primaries, transfer_characteristics, matrix_coefficients;   /* 2 = unspecified when absent */
    uint8_t separate_uv_delta_q, film_grain_params_present;
    uint32_t max_width, max_height;                   /* max_frame_{width,height}_minus_1 + 1 */
    uint8_t frame_width_bits, frame_height_bits;      /* ..._bits_minus_1 + 1 */
    uint8_t frame_id_numbers_present, delta_frame_id_length, additional_frame_id_length;
    uint8_t use_128x128_superblock, enable_filter_intra, enable_intra_edge_filter;
    uint8_t enable_interintra_compound, enable_masked_compound, enable_warped_motion, enable_dual_filter;
    uint8_t enable_order_hint, enable_jnt_comp, enable_ref_frame_mvs;
    uint8_t seq_force_screen_content_tools, seq_force_integer_mv;   /* 2 = selected per frame */
    uint8_t order_hint_bits;
    uint8_t enable_superres, enable_cdef, enable_restoration;
    uint8_t timing_info_present, equal_picture_interval, decoder_model_info_present, initial_display_delay_present;
    uint32_t num_units_in_display_tick, time_scale, num_ticks_per_picture;   /* the last: ..._minus_1 + 1 */
    uint8_t buffer_delay_length, buffer_removal_time_length, frame_presentation_time_length;   /* ..._minus_1 + 1 */
    uint32_t num_units_in_decoding_tick;
    uint8_t num_operating_points;                                   /* operating_points_cnt_minus_1 + 1 */
    uint16_t operating_point_idc[EA1_MAX_OPERATING_POINTS];
    uint8_t seq_level_idx[EA1_MAX_OPERATING_POINTS], seq_tier[EA1_MAX_OPERATING_POINTS];
    uint8_t decoder_model_present_for_op[EA1_MAX_OPERATING_POINTS], low_delay_mode[EA1_MAX_OPERATING_POINTS];
    uint32_t decoder_buffer_delay[EA1_MAX_OPERATING_POINTS], encoder_buffer_delay[EA1_MAX_OPERATING_POINTS];
    uint8_t initial_display_delay[EA1_MAX_OPERATING_POINTS];        /* ..._minus_1 + 1, 0 = not signalled */
} ea1_av1_sequence_header;

/*
 * dav1d_parse_sequence_header (src/easyav1.c:2339): walk the OBUs of `data` and parse the first
 * sequence header.  EA1_OK: found and parsed; EA1_ENOENT: the data is well formed but holds no
 * sequence header (dav1d: DAV1D_ERR(ENOENT), which easyav1 tolerates); EA1_EINVAL: malformed OBU
 * or sequence header (reserved profile, truncated, ...).
 */
int ea1_av1_parse_sequence_header(const uint8_t *data, size_t size, ea1_av1_sequence_header *out);

#ifdef __cplusplus
}
#endif
#endif
