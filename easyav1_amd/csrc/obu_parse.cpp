// AV1 OBU framing + sequence header (AV1 specification 4.10, 5.3, 5.5): the table-free slice of
// the bitstream front end, see include/easyav1_amd_obu.h.  Host C++ by design.
#include <cstring>

#include "ea1_common.h"
#include "easyav1_amd_obu.h"
#include "obu_bits.h"

namespace {

using ea1::BitReader;

// leb128() (4.10.5): at most 8 bytes, value below 2^32
bool leb128(const uint8_t *p, size_t size, uint64_t *value, size_t *used)
{
    uint64_t v = 0;
    for (size_t i = 0; i < 8; i++) {
        if (i >= size) return false;
        v |= uint64_t(p[i] & 0x7F) << (7 * i);
        if (!(p[i] & 0x80)) {
            if (v > 0xFFFFFFFFull) return false;
            *value = v;
            *used = i + 1;
            return true;
        }
    }
    return false;
}

int parse_sequence_header(const uint8_t *p, size_t size, ea1_av1_sequence_header *h)
{
    memset(h, 0, sizeof *h);
    BitReader b{p, size};
    h->profile = uint8_t(b.f(3));
    if (h->profile > 2) return ea1::fail(EA1_EINVAL, "sequence header: reserved seq_profile %u", h->profile);
    h->still_picture = uint8_t(b.f(1));
    h->reduced_still_picture_header = uint8_t(b.f(1));
    if (h->reduced_still_picture_header && !h->still_picture) return ea1::fail(EA1_EINVAL, "sequence header: reduced header without still_picture");
    if (h->reduced_still_picture_header) {
        h->num_operating_points = 1;
        h->seq_level_idx[0] = uint8_t(b.f(5));
    } else {
        h->timing_info_present = uint8_t(b.f(1));
        if (h->timing_info_present) {
            h->num_units_in_display_tick = b.f(32);
            h->time_scale = b.f(32);
            if (!h->num_units_in_display_tick || !h->time_scale) return ea1::fail(EA1_EINVAL, "sequence header: zero timing unit");
            h->equal_picture_interval = uint8_t(b.f(1));
            if (h->equal_picture_interval) {
                const uint32_t n = b.uvlc();
                if (n == 0xFFFFFFFFu) return ea1::fail(EA1_EINVAL, "sequence header: num_ticks_per_picture_minus_1");
                h->num_ticks_per_picture = n + 1;
            }
            h->decoder_model_info_present = uint8_t(b.f(1));
            if (h->decoder_model_info_present) {
                h->buffer_delay_length = uint8_t(b.f(5) + 1);
                h->num_units_in_decoding_tick = b.f(32);
                h->buffer_removal_time_length = uint8_t(b.f(5) + 1);
                h->frame_presentation_time_length = uint8_t(b.f(5) + 1);
            }
        }
        h->initial_display_delay_present = uint8_t(b.f(1));
        h->num_operating_points = uint8_t(b.f(5) + 1);
        for (int i = 0; i < h->num_operating_points; i++) {
            h->operating_point_idc[i] = uint16_t(b.f(12));
            h->seq_level_idx[i] = uint8_t(b.f(5));
            h->seq_tier[i] = h->seq_level_idx[i] > 7 ? uint8_t(b.f(1)) : 0;
            if (h->decoder_model_info_present) {
                h->decoder_model_present_for_op[i] = uint8_t(b.f(1));
                if (h->decoder_model_present_for_op[i]) {
                    h->decoder_buffer_delay[i] = b.f(h->buffer_delay_length);
                    h->encoder_buffer_delay[i] = b.f(h->buffer_delay_length);
                    h->low_delay_mode[i] = uint8_t(b.f(1));
                }
            }
            if (h->initial_display_delay_present && b.f(1)) h->initial_display_delay[i] = uint8_t(b.f(4) + 1);
        }
    }
    h->frame_width_bits = uint8_t(b.f(4) + 1);
    h->frame_height_bits = uint8_t(b.f(4) + 1);
    h->max_width = b.f(h->frame_width_bits) + 1;
    h->max_height = b.f(h->frame_height_bits) + 1;
    if (!h->reduced_still_picture_header) h->frame_id_numbers_present = uint8_t(b.f(1));
    if (h->frame_id_numbers_present) {
        h->delta_frame_id_length = uint8_t(b.f(4) + 2);
        h->additional_frame_id_length = uint8_t(b.f(3) + 1);
    }
    h->use_128x128_superblock = uint8_t(b.f(1));
    h->enable_filter_intra = uint8_t(b.f(1));
    h->enable_intra_edge_filter = uint8_t(b.f(1));
    if (h->reduced_still_picture_header) {
        h->seq_force_screen_content_tools = 2;   // SELECT_SCREEN_CONTENT_TOOLS
        h->seq_force_integer_mv = 2;             // SELECT_INTEGER_MV
    } else {
        h->enable_interintra_compound = uint8_t(b.f(1));
        h->enable_masked_compound = uint8_t(b.f(1));
        h->enable_warped_motion = uint8_t(b.f(1));
        h->enable_dual_filter = uint8_t(b.f(1));
        h->enable_order_hint = uint8_t(b.f(1));
        if (h->enable_order_hint) {
            h->enable_jnt_comp = uint8_t(b.f(1));
            h->enable_ref_frame_mvs = uint8_t(b.f(1));
        }
        h->seq_force_screen_content_tools = b.f(1) ? 2 : uint8_t(b.f(1));   // seq_choose_screen_content_tools
        if (h->seq_force_screen_content_tools > 0) h->seq_force_integer_mv = b.f(1) ? 2 : uint8_t(b.f(1));   // seq_choose_integer_mv
        else h->seq_force_integer_mv = 2;
        if (h->enable_order_hint) h->order_hint_bits = uint8_t(b.f(3) + 1);
    }
    h->enable_superres = uint8_t(b.f(1));
    h->enable_cdef = uint8_t(b.f(1));
    h->enable_restoration = uint8_t(b.f(1));
    // color_config() (5.5.2)
    const uint32_t high_bitdepth = b.f(1);
    h->bit_depth = 8;
    if (h->profile == 2 && high_bitdepth) h->bit_depth = b.f(1) ? 12 : 10;
    else if (high_bitdepth) h->bit_depth = 10;
    h->hbd = uint8_t((h->bit_depth - 8) / 2);
    h->monochrome = h->profile == 1 ? 0 : uint8_t(b.f(1));
    h->color_description_present = uint8_t(b.f(1));
    if (h->color_description_present) {
        h->color_primaries = uint8_t(b.f(8));
        h->transfer_characteristics = uint8_t(b.f(8));
        h->matrix_coefficients = uint8_t(b.f(8));
    } else {
        h->color_primaries = h->transfer_characteristics = h->matrix_coefficients = 2;   // unspecified
    }
    if (h->monochrome) {
        h->color_range = uint8_t(b.f(1));
        h->subsampling_x = h->subsampling_y = 1;
        h->layout = EA1_LAYOUT_I400;
    } else {
        if (h->color_primaries == 1 && h->transfer_characteristics == 13 && h->matrix_coefficients == 0) {   // BT.709 + sRGB + identity
            h->color_range = 1;
            h->subsampling_x = h->subsampling_y = 0;
            if (!(h->profile == 1 || (h->profile == 2 && h->bit_depth == 12))) return ea1::fail(EA1_EINVAL, "sequence header: sRGB identity needs a 4:4:4 profile");
        } else {
            h->color_range = uint8_t(b.f(1));
            if (h->profile == 0) {
                h->subsampling_x = h->subsampling_y = 1;
            } else if (h->profile == 1) {
                h->subsampling_x = h->subsampling_y = 0;
            } else if (h->bit_depth == 12) {
                h->subsampling_x = uint8_t(b.f(1));
                h->subsampling_y = h->subsampling_x ? uint8_t(b.f(1)) : 0;
            } else {
                h->subsampling_x = 1;
                h->subsampling_y = 0;
            }
            if (h->subsampling_x && h->subsampling_y) h->chroma_sample_position = uint8_t(b.f(2));
        }
        h->separate_uv_delta_q = uint8_t(b.f(1));
        h->layout = uint8_t(h->subsampling_x ? (h->subsampling_y ? EA1_LAYOUT_I420 : EA1_LAYOUT_I422) : EA1_LAYOUT_I444);
    }
    h->film_grain_params_present = uint8_t(b.f(1));
    if (b.overrun) return ea1::fail(EA1_EINVAL, "sequence header: truncated");
    // trailing_bits(): a one, then zeros to the end of the OBU (5.3.4)
    if (b.f(1) != 1 || b.overrun) return ea1::fail(EA1_EINVAL, "sequence header: trailing bits");
    while ((b.pos >> 3) < b.size)
        if (b.f(1)) return ea1::fail(EA1_EINVAL, "sequence header: non-zero trailing bits");
    return EA1_OK;
}

}  // namespace

extern "C" int ea1_av1_next_obu(const uint8_t *data, size_t size, ea1_obu *out)
{
    if (!data || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_next_obu: NULL argument");
    memset(out, 0, sizeof *out);
    if (size < 1) return ea1::fail(EA1_EINVAL, "OBU: empty");
    const uint8_t h = data[0];
    if (h & 0x80) return ea1::fail(EA1_EINVAL, "OBU: forbidden bit set");
    out->type = (h >> 3) & 15;
    out->has_extension = (h >> 2) & 1;
    out->has_size_field = (h >> 1) & 1;
    size_t pos = 1;
    if (out->has_extension) {
        if (size < 2) return ea1::fail(EA1_EINVAL, "OBU: truncated extension header");
        out->temporal_id = data[1] >> 5;
        out->spatial_id = (data[1] >> 3) & 3;
        pos = 2;
    }
    uint64_t payload = size - pos;
    if (out->has_size_field) {
        size_t used = 0;
        if (!leb128(data + pos, size - pos, &payload, &used)) return ea1::fail(EA1_EINVAL, "OBU: bad size field");
        pos += used;
        if (payload > size - pos) return ea1::fail(EA1_EINVAL, "OBU: payload of %llu bytes, %zu left", (unsigned long long)payload, size - pos);
    }
    out->payload = data + pos;
    out->payload_size = size_t(payload);
    out->total_size = pos + size_t(payload);
    return EA1_OK;
}

extern "C" int ea1_av1_parse_sequence_header(const uint8_t *data, size_t size, ea1_av1_sequence_header *out)
{
    if (!data || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_parse_sequence_header: NULL argument");
    size_t pos = 0;
    while (pos < size) {
        ea1_obu o;
        const int rc = ea1_av1_next_obu(data + pos, size - pos, &o);
        if (rc != EA1_OK) return rc;
        if (o.type == EA1_OBU_SEQUENCE_HEADER) return parse_sequence_header(o.payload, o.payload_size, out);
        pos += o.total_size;
    }
    return ea1::fail(EA1_ENOENT, "no sequence header in these %zu bytes", size);
}
