"""OBU framing + sequence header through the C ABI (include/easyav1_amd_obu.h): harness-side wrapper."""
import ctypes

from . import _lib


class Obu(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("has_extension", ctypes.c_uint8), ("temporal_id", ctypes.c_uint8), ("spatial_id", ctypes.c_uint8),
                ("has_size_field", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 3), ("payload", ctypes.c_void_p), ("payload_size", ctypes.c_size_t),
                ("total_size", ctypes.c_size_t)]


class SequenceHeader(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_uint8) for n in ("profile", "still_picture", "reduced_still_picture_header", "layout", "hbd", "bit_depth", "monochrome",
                                               "subsampling_x", "subsampling_y", "chroma_sample_position", "color_description_present", "color_range",
                                               "color_primaries", "transfer_characteristics", "matrix_coefficients", "separate_uv_delta_q",
                                               "film_grain_params_present")] +
                [("max_width", ctypes.c_uint32), ("max_height", ctypes.c_uint32)] +
                [(n, ctypes.c_uint8) for n in ("frame_width_bits", "frame_height_bits", "frame_id_numbers_present", "delta_frame_id_length",
                                               "additional_frame_id_length", "use_128x128_superblock", "enable_filter_intra", "enable_intra_edge_filter",
                                               "enable_interintra_compound", "enable_masked_compound", "enable_warped_motion", "enable_dual_filter",
                                               "enable_order_hint", "enable_jnt_comp", "enable_ref_frame_mvs", "seq_force_screen_content_tools",
                                               "seq_force_integer_mv", "order_hint_bits", "enable_superres", "enable_cdef", "enable_restoration",
                                               "timing_info_present", "equal_picture_interval", "decoder_model_info_present",
                                               "initial_display_delay_present")] +
                [("num_units_in_display_tick", ctypes.c_uint32), ("time_scale", ctypes.c_uint32), ("num_ticks_per_picture", ctypes.c_uint32)] +
                [(n, ctypes.c_uint8) for n in ("buffer_delay_length", "buffer_removal_time_length", "frame_presentation_time_length")] +
                [("num_units_in_decoding_tick", ctypes.c_uint32), ("num_operating_points", ctypes.c_uint8),
                 ("operating_point_idc", ctypes.c_uint16 * 32), ("seq_level_idx", ctypes.c_uint8 * 32), ("seq_tier", ctypes.c_uint8 * 32),
                 ("decoder_model_present_for_op", ctypes.c_uint8 * 32), ("low_delay_mode", ctypes.c_uint8 * 32),
                 ("decoder_buffer_delay", ctypes.c_uint32 * 32), ("encoder_buffer_delay", ctypes.c_uint32 * 32),
                 ("initial_display_delay", ctypes.c_uint8 * 32)])

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else int(v)
        return out


def next_obu(data):
    """-> (rc, Obu, payload bytes)"""
    L = _lib.lib()
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
    o = Obu()
    L.ea1_av1_next_obu.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    rc = L.ea1_av1_next_obu(buf, len(data), ctypes.byref(o))
    payload = b""
    if rc == 0:
        off = o.payload - ctypes.addressof(buf)
        payload = bytes(data[off: off + o.payload_size])
    return rc, o, payload


def parse_sequence_header(data):
    """-> (rc, dict or None)"""
    L = _lib.lib()
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
    h = SequenceHeader()
    L.ea1_av1_parse_sequence_header.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    rc = L.ea1_av1_parse_sequence_header(buf, len(data), ctypes.byref(h))
    return rc, (h.as_dict() if rc == 0 else None)
