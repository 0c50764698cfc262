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
 * Parity: pinned since round 4 -- libaom 3.13.2 streams parse to frames whose pictures equal dav1d 1.5.3's (tests/test_real_av1.py); also checked
 * against an independent OBU writer in tests/.
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_OBU_H
#define EASYAV1_AMD_OBU_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_audio.h"
#include "easyav1_amd_grain.h"
#include "easyav1_amd_video.h"
#include "easyav1_amd_ir_packet.h"

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

/* ---------------------------------------------------------------------------------------------
 * Frame header (AV1 specification 5.9 uncompressed_header): second table-free slice of N1.  What
 * dav1d's src/obu.c (/root/reference/cmake/dav1d.cmake:984) extracts per frame and what the GPU
 * path takes from it: frame size, reference bookkeeping, loop filter / CDEF / loop restoration
 * parameters (-> ea1_frame_desc), global motion matrices (-> EA1_BLK_WARP blocks), film grain
 * parameters (-> ea1_film_grain).  The parser keeps the per-slot state the header syntax depends on
 * (frame sizes, order hints, saved global motion / loop filter deltas / segmentation / film grain).
 * Not covered: large-scale-tile tools (OBU_TILE_LIST).
 * --------------------------------------------------------------------------------------------- */
enum { EA1_KEY_FRAME = 0, EA1_INTER_FRAME = 1, EA1_INTRA_ONLY_FRAME = 2, EA1_SWITCH_FRAME = 3 };
enum { EA1_GM_IDENTITY = 0, EA1_GM_TRANSLATION = 1, EA1_GM_ROTZOOM = 2, EA1_GM_AFFINE = 3 };
#define EA1_PRIMARY_REF_NONE 7
#define EA1_MAX_TILE_COLS 64
#define EA1_MAX_TILE_ROWS 64

typedef struct ea1_av1_film_grain_syntax {   /* film_grain_params() as coded */
    uint8_t apply_grain, update_grain, film_grain_params_ref_idx;
    uint8_t num_y_points, point_y_value[14], point_y_scaling[14];
    uint8_t chroma_scaling_from_luma, num_cb_points, point_cb_value[10], point_cb_scaling[10];
    uint8_t num_cr_points, point_cr_value[10], point_cr_scaling[10];
    uint8_t grain_scaling_minus_8, ar_coeff_lag;
    uint8_t ar_coeffs_y_plus_128[24], ar_coeffs_cb_plus_128[25], ar_coeffs_cr_plus_128[25];
    uint8_t ar_coeff_shift_minus_6, grain_scale_shift;
    uint8_t cb_mult, cb_luma_mult, cr_mult, cr_luma_mult, overlap_flag, clip_to_restricted_range;
    uint16_t grain_seed, cb_offset, cr_offset;
} ea1_av1_film_grain_syntax;

typedef struct ea1_av1_frame_header {
    uint8_t show_existing_frame, frame_to_show_map_idx;
    uint8_t frame_type, show_frame, showable_frame, error_resilient_mode;
    uint8_t disable_cdf_update, allow_screen_content_tools, force_integer_mv;
    uint8_t frame_size_override_flag, primary_ref_frame, refresh_frame_flags;
    uint8_t order_hint, frame_refs_short_signaling, allow_intrabc;
    uint8_t allow_high_precision_mv, is_filter_switchable, interpolation_filter, is_motion_mode_switchable, use_ref_frame_mvs;
    uint8_t disable_frame_end_update_cdf, allow_warped_motion, reduced_tx_set;
    uint8_t tx_mode_select, reference_select, skip_mode_present, skip_mode_frame[2];   /* SkipModeFrame: 1..7, 0 = none */
    int8_t ref_frame_idx[7];
    uint8_t ref_order_hint[8];                /* read only with error_resilient_mode && enable_order_hint */
    uint32_t current_frame_id, display_frame_id, frame_presentation_time;
    uint32_t frame_width, frame_height, upscaled_width, render_width, render_height;
    uint8_t use_superres, superres_denom;     /* SuperresDenom: 8 = none, 9..16 */
    uint32_t mi_cols, mi_rows;
    /* tile_info() */
    uint8_t uniform_tile_spacing_flag, tile_cols, tile_rows, tile_cols_log2, tile_rows_log2;
    uint8_t tile_size_bytes;                  /* TileSizeBytes: 1..4 */
    uint16_t context_update_tile_id;
    uint16_t mi_col_starts[EA1_MAX_TILE_COLS + 1], mi_row_starts[EA1_MAX_TILE_ROWS + 1];
    /* quantization_params() */
    uint8_t base_q_idx, using_qmatrix, qm_y, qm_u, qm_v, diff_uv_delta;
    int8_t delta_q_y_dc, delta_q_u_dc, delta_q_u_ac, delta_q_v_dc, delta_q_v_ac;
    /* segmentation_params() */
    uint8_t segmentation_enabled, segmentation_update_map, segmentation_temporal_update, segmentation_update_data;
    uint8_t feature_enabled[8][8], seg_id_pre_skip, last_active_seg_id;
    int16_t feature_data[8][8];
    /* delta_q_params() / delta_lf_params() */
    uint8_t delta_q_present, delta_q_res, delta_lf_present, delta_lf_res, delta_lf_multi;
    uint8_t coded_lossless, all_lossless;
    /* loop_filter_params() */
    uint8_t loop_filter_level[4], loop_filter_sharpness, loop_filter_delta_enabled, loop_filter_delta_update;
    int8_t loop_filter_ref_deltas[8], loop_filter_mode_deltas[2];
    /* cdef_params() */
    uint8_t cdef_damping, cdef_bits, cdef_y_pri_strength[8], cdef_y_sec_strength[8], cdef_uv_pri_strength[8], cdef_uv_sec_strength[8];
    /* lr_params() */
    uint8_t frame_restoration_type[3];        /* EA1_LR_*: 0 none, 1 Wiener, 2 self-guided, 3 switchable */
    uint8_t lr_unit_shift, lr_uv_shift;
    uint16_t loop_restoration_size[3];
    /* global_motion_params(): gm_params[ref - LAST_FRAME], warpParams of AV1 spec 7.11.3.6 */
    uint8_t gm_type[7];
    int32_t gm_params[7][6];
    /* film_grain_params(): after load_grain_params when update_grain == 0 */
    ea1_av1_film_grain_syntax film_grain;
    /* decoder-model side information */
    uint8_t buffer_removal_time_present;
    uint32_t buffer_removal_time[EA1_MAX_OPERATING_POINTS];
    uint32_t header_bits;                     /* size of uncompressed_header() in bits (frame_header_obu: byte_alignment follows) */
} ea1_av1_frame_header;

typedef struct ea1_av1_parser ea1_av1_parser;   /* sequence header + the 8 reference slots' saved state */
int ea1_av1_parser_create(ea1_av1_parser **out);
void ea1_av1_parser_destroy(ea1_av1_parser *p);
/* activate a sequence header (a new one that differs resets the reference slots) */
int ea1_av1_parser_set_sequence_header(ea1_av1_parser *p, const ea1_av1_sequence_header *seq);
/*
 * Parse the uncompressed_header() at the start of the payload of an OBU_FRAME_HEADER / OBU_FRAME /
 * OBU_REDUNDANT_FRAME_HEADER (temporal_id / spatial_id: from the OBU extension, 0 without one).
 * EA1_ESTATE: no sequence header yet; EA1_EINVAL: malformed, or a reference the header needs is
 * missing (e.g. a stream entered at a non-key frame).
 */
int ea1_av1_parser_frame_header(ea1_av1_parser *p, const uint8_t *payload, size_t size, int temporal_id, int spatial_id,
                                ea1_av1_frame_header *out);
/*
 * The frame of that header has been decoded: run the reference frame update process (AV1 spec
 * 7.20; for show_existing_frame of a key frame 7.21) so that later headers see its state.
 */
int ea1_av1_parser_end_frame(ea1_av1_parser *p, const ea1_av1_frame_header *hdr);

/*
 * Tile group (AV1 specification 5.11.1 tile_group_obu): where the entropy-coded bytes of each tile
 * lie.  `payload` is the payload of an OBU_TILE_GROUP, or of an OBU_FRAME from byte
 * (hdr->header_bits + 7) / 8 on (frame_header_obu, byte_alignment, then the tile group).  Tiles
 * are independently decodable: this table is what a tile-parallel symbol decoder starts from.
 */
typedef struct ea1_av1_tile {
    uint16_t tile_num, tile_row, tile_col, pad;
    uint32_t mi_row_start, mi_row_end, mi_col_start, mi_col_end;
    size_t offset, size;          /* the tile's bytes inside `payload` */
} ea1_av1_tile;
int ea1_av1_parse_tile_group(const ea1_av1_frame_header *hdr, const uint8_t *payload, size_t size, ea1_av1_tile *tiles, int cap,
                             int *count);

/*
 * The OBU layer in one call (AV1 specification 7.5 ordering of OBUs, 5.9.1 frame_header_obu): walk the
 * OBUs at `data` until ONE frame is complete and describe it.  Sequence headers are parsed and
 * activated, temporal delimiters / padding / metadata / tile lists are passed over, OBUs of layers
 * outside operating point 0 are dropped (6.2.2 drop_obu), a repeated frame header (OBU_FRAME_HEADER
 * again or OBU_REDUNDANT_FRAME_HEADER) must be a bit-for-bit copy of the first, and tile groups
 * are collected until the last tile of the frame has been seen.
 *   consumed   bytes of `data` used (call again with data + consumed for the next frame)
 *   hdr        the frame's header (after ea1_av1_parser_frame_header)
 *   tiles      byte ranges of its tiles, offsets relative to `data`; none for show_existing_frame
 * EA1_OK: a frame was assembled -- decode it, then call ea1_av1_parser_end_frame(hdr).
 * EA1_ENOENT: `data` ended before a frame was complete (consumed = what was skipped; a partial frame
 * is discarded).  EA1_EINVAL: malformed.  What dav1d_send_data + dav1d_parse_sequence_header do with
 * a packet up to the start of the symbol decode (src/easyav1.c:2339, :2390).
 */
int ea1_av1_parser_next_frame(ea1_av1_parser *p, const uint8_t *data, size_t size, size_t *consumed, ea1_av1_frame_header *hdr,
                              ea1_av1_tile *tiles, int cap, int *count);
/* what a reference slot holds (saved by ea1_av1_parser_end_frame): the block-level parser needs the order hints and sizes */
typedef struct ea1_av1_slot_info {
    uint8_t valid, frame_type, order_hint, showable_frame;
    uint32_t upscaled_width, frame_width, frame_height;
} ea1_av1_slot_info;
int ea1_av1_parser_slot_info(const ea1_av1_parser *p, int slot, ea1_av1_slot_info *out);
/* the active sequence header (EA1_ESTATE before one has been seen) */
int ea1_av1_parser_sequence_header(const ea1_av1_parser *p, ea1_av1_sequence_header *out);

/* ---------------------------------------------------------------------------------------------
 * Symbol decoder (AV1 specification 8.2; dav1d src/msac.c, /root/reference/cmake/dav1d.cmake:977):
 * the adaptive multi-symbol arithmetic decoder every tile's bytes go through.  Table-free: the
 * engine and the CDF adaptation are arithmetic; what N1 still lacks are the DEFAULT CDF arrays the
 * syntax elements start from (recall-only data).  CDF layout as in the specification: N increasing
 * 15-bit cumulative values, the last one 32768, followed by the adaptation counter (N + 1 entries).
 * dav1d keeps 32768 - cdf[i]; the arithmetic is the same.
 * --------------------------------------------------------------------------------------------- */
typedef struct ea1_msac {
    const uint8_t *data;      /* the tile's bytes (borrowed)                           */
    size_t size, pos_bits;    /* bits consumed from data                                */
    uint32_t value;           /* SymbolValue                                            */
    uint32_t range;           /* SymbolRange: 32768 .. 65535                            */
    int64_t max_bits;         /* SymbolMaxBits                                          */
    int disable_update;       /* frame header disable_cdf_update                        */
} ea1_msac;
/* init_symbol(sz) (8.2.2); sz >= 1 */
int ea1_msac_init(ea1_msac *s, const uint8_t *data, size_t size, int disable_cdf_update);
/* read_symbol(cdf) (8.2.6) with adaptation unless disabled: returns the symbol 0 .. n - 1 */
int ea1_msac_symbol(ea1_msac *s, uint16_t *cdf, int n);
/* read_bool() (8.2.5) and read_literal(n) (8.2.4): equiprobable bits, no adaptation */
int ea1_msac_bool(ea1_msac *s);
uint32_t ea1_msac_literal(ea1_msac *s, int nbits);
/* bits of the tile not yet drawn into the decoder (negative once it runs on padding zeros) */
int64_t ea1_msac_bits_left(const ea1_msac *s);

/*
 * The hand-over from the frame header to the GPU path:
 *   _to_desc      frame-level fields of ea1_frame_desc (size in whole 8x8 units + crop to the frame size,
 *                 is_inter, deblock levels /
 *                 sharpness, lf_per_block, CDEF damping / strengths, restoration types / unit sizes,
 *                 intra_edge_filter); dst_slot and the num_* counts stay for the block parser to fill.
 *                 EA1_EINVAL for what the path does not decode (not 8-bit 4:2:0, super-resolution).
 *   _film_grain   ea1_film_grain of the frame (EA1_ENOENT: no grain to apply)
 * Global motion: gm_params[ref] IS the matrix an EA1_BLK_WARP block carries for a global-motion
 * prediction from reference ref + 1 (LAST_FRAME = 1).
 */
int ea1_av1_frame_header_to_desc(const ea1_av1_frame_header *hdr, const ea1_av1_sequence_header *seq, ea1_frame_desc *desc);
int ea1_av1_frame_header_film_grain(const ea1_av1_frame_header *hdr, const ea1_av1_sequence_header *seq, ea1_film_grain *out);

/* ---------------------------------------------------------------------------------------------
 * Block-level parse (AV1 specification 5.11 from decode_tile down; dav1d src/decode.c, refmvs.c, the read
 * side of recon_tmpl.c, lf_mask.c: /root/reference/cmake/dav1d.cmake:975,985,1014,980; call site
 * dav1d_send_data, /root/reference/src/easyav1.c:2390): tile bytes -> the hand-over structures of
 * easyav1_amd_video.h.  TABLE-AGNOSTIC: the default CDFs, the scan orders and the quantiser look-ups
 * are data of the specification that this library does not own; the caller loads them as ONE blob
 * whose layout the library describes at run time:
 *
 *   bytes 0..15      u32 magic "EA1T", u32 version, u32 size of the blob in uint16 units, u32 0
 *                    (ea1_av1_tables_blob_header writes them)
 *   then sections    of uint16 values, little endian, in the order ea1_av1_tables_section() lists:
 *     kind 0  `count / symbols` CDFs of `symbols` values each, as the specification prints them
 *             (increasing, ending at 32768) WITHOUT the trailing adaptation counter; the arrays'
 *             index order is the specification's (e.g. IntraFrameYMode[5][5][13]); the coefficient
 *             arrays come four times (suffix _q0.._q3: the quantiser classes of init_coeff_cdfs)
 *     kind 1  a scan order: width x height positions (row * width + column)
 *     kind 2  Dc_Qlookup[0] / Ac_Qlookup[0] (8 bit), 256 values
 *     kind 4  one size of one quantiser matrix: Quantizer_Matrix[level 0..14][luma / chroma], split by transform size (the 14
 *             coded sizes, 64-sample sides use the matrix of 32), width x height weights 1..255, row-major
 *     kind 5  Upscale_Filter[64][8], int16 values (two's complement), rows summing to 128: the super-resolution taps (7.16) of the
 *             reconstruction; all zero = none (frames with super-resolution are then refused)
 *     kind 3  Gaussian_Sequence, 2048 int16 values (two's complement): not needed by the parser, but the film grain
 *             synthesis of easyav1_amd_grain.h needs it and a front end keeps it with the same tables; all zero = no grain
 *
 * Scope of this revision: see easyav1_amd/csrc/av1_tile_syntax.h (8-bit 4:2:0 and 4:0:0):
 * what lies outside returns EA1_ENOTSUP with the reason in ea1_last_error().  Parity: pinned since round 4 against dav1d 1.5.3
 * on libaom streams (tests/test_real_av1.py); also tested by round trip against a generator that walks the same syntax
 * (tests/native/av1_streamgen.cpp) with arbitrary tables, and end to end against the oracle.
 * --------------------------------------------------------------------------------------------- */
typedef struct ea1_av1_tables ea1_av1_tables;
typedef struct ea1_av1_tables_section_info {
    const char *name;
    int kind;          /* 0 CDFs, 1 scan, 2 quantiser look-up, 3 Gaussian sequence, 4 quantiser matrix, 5 Upscale_Filter */
    size_t offset;     /* byte offset in the blob */
    uint32_t count;    /* uint16 values in the section */
    int symbols;       /* kind 0: values per CDF */
    int width, height; /* kind 1 */
} ea1_av1_tables_section_info;
size_t ea1_av1_tables_blob_size(void);
int ea1_av1_tables_num_sections(void);
int ea1_av1_tables_section(int index, ea1_av1_tables_section_info *out);
int ea1_av1_tables_blob_header(void *blob, size_t size);
/* checks every section (CDFs non-decreasing and ending at 32768, scans permutations, quantisers positive) and copies */
int ea1_av1_tables_load(const void *blob, size_t size, ea1_av1_tables **out);
void ea1_av1_tables_destroy(ea1_av1_tables *t);
/* the blob's Gaussian_Sequence (2048 values), for ea1_recon_set_gaussian_sequence */
const int16_t *ea1_av1_tables_gaussian_sequence(const ea1_av1_tables *t);
/* the blob's Upscale_Filter (64 phases x 8 taps; all zero = none), for ea1_recon_set_upscale_filter */
const int16_t *ea1_av1_tables_upscale_filter(const ea1_av1_tables *t);

/*
 * One decoder's parse state: the header parser above + the saved CDFs of the eight reference slots + the
 * mapping of AV1's slots onto the reconstruction context's eight pictures.  `tables` must outlive it.
 */
typedef struct ea1_av1_frame_parser ea1_av1_frame_parser;
int ea1_av1_frame_parser_create(const ea1_av1_tables *tables, ea1_av1_frame_parser **out);
void ea1_av1_frame_parser_destroy(ea1_av1_frame_parser *p);
/* forget the reference state (seek): the next frame must be a key frame */
void ea1_av1_frame_parser_reset(ea1_av1_frame_parser *p);
/* threads that parse the TILES of one frame side by side (1 = the calling thread alone, the default; EASYAV1_AMD_TILE_THREADS sets another default):
 * the tiles of a frame are independent after the per-tile copy of the frame's contexts (dav1d parses them as separate tasks, src/thread_task.c,
 * /root/reference/cmake/dav1d.cmake:992); the records come out in tile order, byte for byte as with one thread. */
int ea1_av1_frame_parser_set_threads(ea1_av1_frame_parser *p, int threads);
/* of those threads, how many the following frames may use (cheap, may be called before every parse: a caller that runs several parsers
 * at once gives each what the host has left) */
int ea1_av1_frame_parser_set_thread_limit(ea1_av1_frame_parser *p, int threads);
/* Round 5: 8-bit 4:2:2 and 4:4:4 streams are parsed (chroma block / transform sizes, chroma-from-luma and palette geometry, the per-plane contexts,
 * restoration units follow subsampling_x / subsampling_y) and handed out with ea1_frame_desc.layout = EA1_LAYOUT_422 / _444: the records of such a frame
 * count chroma positions and sizes in the chroma plane's own 4-sample units, and the context that reconstructs them is made for that layout
 * (ea1_recon_create_layout).  `on` = 0: a consumer that is 4:2:0 only gets EA1_ENOTSUP for such streams instead (default: on). */
int ea1_av1_frame_parser_allow_chroma_formats(ea1_av1_frame_parser *p, int on);
typedef struct ea1_av1_parsed {
    ea1_ir_frame frame;              /* arrays owned by the parser, valid until its next call */
    ea1_av1_frame_header header;
    uint8_t show_frame;              /* the picture is to be output */
    uint8_t show_existing;           /* show_existing_frame: no records; the picture to output is context picture frame.desc.dst_slot */
    uint8_t has_film_grain;
    ea1_film_grain film_grain;
    double parse_seconds;            /* host time of the symbol parse of this frame */
    size_t consumed;                 /* bytes of `data` this frame took: a temporal unit may hold more frames (call again from there) */
} ea1_av1_parsed;
/*
 * Parse the FIRST frame in `data` (the OBUs of a container packet, or what is left of one: a temporal unit may hold hidden
 * frames in front of the shown one, or a lone show_existing_frame header).  EA1_OK: out is filled (frame.desc.dst_slot = the
 * context picture to decode into; blocks' ref[] = context pictures; out->consumed = where the next frame starts).
 * EA1_ENOENT: no frame in the data (e.g. only a sequence header).  EA1_ENOTSUP: a coding tool outside this revision's
 * scope.  EA1_EINVAL: malformed.
 */
int ea1_av1_frame_parser_parse(ea1_av1_frame_parser *p, const uint8_t *data, size_t size, ea1_av1_parsed *out);

#ifdef __cplusplus
}
#endif
#endif
