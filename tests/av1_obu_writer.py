"""Independent AV1 OBU / sequence header WRITER for the parser tests (there is no AV1 encoder and no
stream in this environment): builds the bits of sequence_header_obu() (AV1 specification 5.5) from a
dict of syntax elements, most significant bit first, and wraps payloads into OBUs (5.3)."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def f(self, n, v):
        assert 0 <= v < (1 << n), (n, v)
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def uvlc(self, v):
        n = (v + 1).bit_length() - 1
        self.f(n, 0) if n else None
        self.f(1, 1)
        if n:
            self.f(n, v + 1 - (1 << n))

    def trailing(self):
        self.bits.append(1)
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return bytes(sum(self.bits[i + k] << (7 - k) for k in range(8)) for i in range(0, len(self.bits), 8))


def leb128(v, pad_to=0):
    out = []
    while True:
        b = v & 0x7F
        v >>= 7
        if v or len(out) + 1 < pad_to:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def obu(obu_type, payload, has_size=True, extension=None, size_pad=0):
    h = (obu_type << 3) | ((extension is not None) << 2) | (has_size << 1)
    out = bytes([h])
    if extension is not None:
        t, s = extension
        out += bytes([(t << 5) | (s << 3)])
    if has_size:
        out += leb128(len(payload), size_pad)
    return out + payload


def random_sequence_header(rng):
    """-> dict of the fields of ea1_av1_sequence_header (expected parse result)"""
    h = {}
    h["profile"] = int(rng.integers(3))
    h["still_picture"] = int(rng.integers(2))
    h["reduced_still_picture_header"] = int(h["still_picture"] and rng.integers(2))
    n_ops = 1
    if not h["reduced_still_picture_header"]:
        h["timing_info_present"] = int(rng.integers(2))
        if h["timing_info_present"]:
            h["num_units_in_display_tick"] = int(rng.integers(1, 1 << 32))
            h["time_scale"] = int(rng.integers(1, 1 << 32))
            h["equal_picture_interval"] = int(rng.integers(2))
            if h["equal_picture_interval"]:
                h["num_ticks_per_picture"] = int(rng.choice([1, 2, 3, 8, 1001, 65536, 1 << 20]))
            h["decoder_model_info_present"] = int(rng.integers(2))
            if h["decoder_model_info_present"]:
                h["buffer_delay_length"] = int(rng.integers(1, 33))
                h["num_units_in_decoding_tick"] = int(rng.integers(1 << 32))
                h["buffer_removal_time_length"] = int(rng.integers(1, 33))
                h["frame_presentation_time_length"] = int(rng.integers(1, 33))
        h["initial_display_delay_present"] = int(rng.integers(2))
        n_ops = int(rng.integers(1, 33)) if rng.random() < 0.3 else int(rng.integers(1, 4))
    h["num_operating_points"] = n_ops
    for k in ("operating_point_idc", "seq_level_idx", "seq_tier", "decoder_model_present_for_op", "low_delay_mode", "decoder_buffer_delay",
              "encoder_buffer_delay", "initial_display_delay"):
        h[k] = [0] * 32
    for i in range(n_ops):
        h["seq_level_idx"][i] = int(rng.integers(32))
        if h["reduced_still_picture_header"]:
            continue
        h["operating_point_idc"][i] = int(rng.integers(1 << 12))
        if h["seq_level_idx"][i] > 7:
            h["seq_tier"][i] = int(rng.integers(2))
        if h.get("decoder_model_info_present"):
            h["decoder_model_present_for_op"][i] = int(rng.integers(2))
            if h["decoder_model_present_for_op"][i]:
                n = h["buffer_delay_length"]
                h["decoder_buffer_delay"][i] = int(rng.integers(1 << n)) if n < 32 else int(rng.integers(1 << 32))
                h["encoder_buffer_delay"][i] = int(rng.integers(1 << n)) if n < 32 else int(rng.integers(1 << 32))
                h["low_delay_mode"][i] = int(rng.integers(2))
        if h.get("initial_display_delay_present") and rng.integers(2):
            h["initial_display_delay"][i] = int(rng.integers(1, 17))
    h["frame_width_bits"] = int(rng.integers(1, 17))
    h["frame_height_bits"] = int(rng.integers(1, 17))
    h["max_width"] = int(rng.integers(1, (1 << h["frame_width_bits"]) + 1))
    h["max_height"] = int(rng.integers(1, (1 << h["frame_height_bits"]) + 1))
    if not h["reduced_still_picture_header"]:
        h["frame_id_numbers_present"] = int(rng.integers(2))
    if h.get("frame_id_numbers_present"):
        h["delta_frame_id_length"] = int(rng.integers(2, 18))
        h["additional_frame_id_length"] = int(rng.integers(1, 9))
    for k in ("use_128x128_superblock", "enable_filter_intra", "enable_intra_edge_filter"):
        h[k] = int(rng.integers(2))
    if h["reduced_still_picture_header"]:
        h["seq_force_screen_content_tools"] = 2
        h["seq_force_integer_mv"] = 2
    else:
        for k in ("enable_interintra_compound", "enable_masked_compound", "enable_warped_motion", "enable_dual_filter", "enable_order_hint"):
            h[k] = int(rng.integers(2))
        if h["enable_order_hint"]:
            h["enable_jnt_comp"] = int(rng.integers(2))
            h["enable_ref_frame_mvs"] = int(rng.integers(2))
        h["seq_force_screen_content_tools"] = int(rng.integers(3))
        h["seq_force_integer_mv"] = int(rng.integers(3)) if h["seq_force_screen_content_tools"] > 0 else 2
        if h["enable_order_hint"]:
            h["order_hint_bits"] = int(rng.integers(1, 9))
    for k in ("enable_superres", "enable_cdef", "enable_restoration"):
        h[k] = int(rng.integers(2))
    # colour configuration
    if h["profile"] == 2:
        h["bit_depth"] = int(rng.choice([8, 10, 12]))
    else:
        h["bit_depth"] = int(rng.choice([8, 10]))
    h["hbd"] = (h["bit_depth"] - 8) // 2
    h["monochrome"] = 0 if h["profile"] == 1 else int(rng.integers(2))
    h["color_description_present"] = int(rng.integers(2))
    if h["color_description_present"]:
        h["color_primaries"] = int(rng.integers(256))
        h["transfer_characteristics"] = int(rng.integers(256))
        h["matrix_coefficients"] = int(rng.integers(256))
        if (h["color_primaries"], h["transfer_characteristics"], h["matrix_coefficients"]) == (1, 13, 0):
            h["matrix_coefficients"] = 1
    else:
        h["color_primaries"] = h["transfer_characteristics"] = h["matrix_coefficients"] = 2
    if h["monochrome"]:
        h["color_range"] = int(rng.integers(2))
        h["subsampling_x"] = h["subsampling_y"] = 1
        h["layout"] = 0
    else:
        h["color_range"] = int(rng.integers(2))
        if h["profile"] == 0:
            h["subsampling_x"] = h["subsampling_y"] = 1
        elif h["profile"] == 1:
            h["subsampling_x"] = h["subsampling_y"] = 0
        elif h["bit_depth"] == 12:
            h["subsampling_x"] = int(rng.integers(2))
            h["subsampling_y"] = int(rng.integers(2)) if h["subsampling_x"] else 0
        else:
            h["subsampling_x"], h["subsampling_y"] = 1, 0
        if h["subsampling_x"] and h["subsampling_y"]:
            h["chroma_sample_position"] = int(rng.integers(4))
        h["separate_uv_delta_q"] = int(rng.integers(2))
        h["layout"] = (1 if h["subsampling_y"] else 2) if h["subsampling_x"] else 3
    h["film_grain_params_present"] = int(rng.integers(2))
    return h


def write_sequence_header(h):
    w = BitWriter()
    g = lambda k: h.get(k, 0)
    w.f(3, h["profile"])
    w.f(1, h["still_picture"])
    w.f(1, h["reduced_still_picture_header"])
    if h["reduced_still_picture_header"]:
        w.f(5, h["seq_level_idx"][0])
    else:
        w.f(1, g("timing_info_present"))
        if g("timing_info_present"):
            w.f(32, h["num_units_in_display_tick"])
            w.f(32, h["time_scale"])
            w.f(1, g("equal_picture_interval"))
            if g("equal_picture_interval"):
                w.uvlc(h["num_ticks_per_picture"] - 1)
            w.f(1, g("decoder_model_info_present"))
            if g("decoder_model_info_present"):
                w.f(5, h["buffer_delay_length"] - 1)
                w.f(32, h["num_units_in_decoding_tick"])
                w.f(5, h["buffer_removal_time_length"] - 1)
                w.f(5, h["frame_presentation_time_length"] - 1)
        w.f(1, g("initial_display_delay_present"))
        w.f(5, h["num_operating_points"] - 1)
        for i in range(h["num_operating_points"]):
            w.f(12, h["operating_point_idc"][i])
            w.f(5, h["seq_level_idx"][i])
            if h["seq_level_idx"][i] > 7:
                w.f(1, h["seq_tier"][i])
            if g("decoder_model_info_present"):
                w.f(1, h["decoder_model_present_for_op"][i])
                if h["decoder_model_present_for_op"][i]:
                    w.f(h["buffer_delay_length"], h["decoder_buffer_delay"][i])
                    w.f(h["buffer_delay_length"], h["encoder_buffer_delay"][i])
                    w.f(1, h["low_delay_mode"][i])
            if g("initial_display_delay_present"):
                w.f(1, int(h["initial_display_delay"][i] > 0))
                if h["initial_display_delay"][i]:
                    w.f(4, h["initial_display_delay"][i] - 1)
    w.f(4, h["frame_width_bits"] - 1)
    w.f(4, h["frame_height_bits"] - 1)
    w.f(h["frame_width_bits"], h["max_width"] - 1)
    w.f(h["frame_height_bits"], h["max_height"] - 1)
    if not h["reduced_still_picture_header"]:
        w.f(1, g("frame_id_numbers_present"))
    if g("frame_id_numbers_present"):
        w.f(4, h["delta_frame_id_length"] - 2)
        w.f(3, h["additional_frame_id_length"] - 1)
    w.f(1, h["use_128x128_superblock"])
    w.f(1, h["enable_filter_intra"])
    w.f(1, h["enable_intra_edge_filter"])
    if not h["reduced_still_picture_header"]:
        for k in ("enable_interintra_compound", "enable_masked_compound", "enable_warped_motion", "enable_dual_filter", "enable_order_hint"):
            w.f(1, h[k])
        if h["enable_order_hint"]:
            w.f(1, h["enable_jnt_comp"])
            w.f(1, h["enable_ref_frame_mvs"])
        if h["seq_force_screen_content_tools"] == 2:
            w.f(1, 1)
        else:
            w.f(1, 0)
            w.f(1, h["seq_force_screen_content_tools"])
        if h["seq_force_screen_content_tools"] > 0:
            if h["seq_force_integer_mv"] == 2:
                w.f(1, 1)
            else:
                w.f(1, 0)
                w.f(1, h["seq_force_integer_mv"])
        if h["enable_order_hint"]:
            w.f(3, h["order_hint_bits"] - 1)
    w.f(1, h["enable_superres"])
    w.f(1, h["enable_cdef"])
    w.f(1, h["enable_restoration"])
    w.f(1, int(h["bit_depth"] > 8))
    if h["profile"] == 2 and h["bit_depth"] > 8:
        w.f(1, int(h["bit_depth"] == 12))
    if h["profile"] != 1:
        w.f(1, h["monochrome"])
    w.f(1, h["color_description_present"])
    if h["color_description_present"]:
        w.f(8, h["color_primaries"])
        w.f(8, h["transfer_characteristics"])
        w.f(8, h["matrix_coefficients"])
    if h["monochrome"]:
        w.f(1, h["color_range"])
    else:
        w.f(1, h["color_range"])
        if h["profile"] == 2 and h["bit_depth"] == 12:
            w.f(1, h["subsampling_x"])
            if h["subsampling_x"]:
                w.f(1, h["subsampling_y"])
        if h["subsampling_x"] and h["subsampling_y"]:
            w.f(2, g("chroma_sample_position"))
        w.f(1, h["separate_uv_delta_q"])
    w.f(1, h["film_grain_params_present"])
    w.trailing()
    return w.bytes()
