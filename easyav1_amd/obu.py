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


U8, U16, U32, I8, I16, I32 = ctypes.c_uint8, ctypes.c_uint16, ctypes.c_uint32, ctypes.c_int8, ctypes.c_int16, ctypes.c_int32


class FilmGrainSyntax(ctypes.Structure):
    _fields_ = [("apply_grain", U8), ("update_grain", U8), ("film_grain_params_ref_idx", U8), ("num_y_points", U8), ("point_y_value", U8 * 14),
                ("point_y_scaling", U8 * 14), ("chroma_scaling_from_luma", U8), ("num_cb_points", U8), ("point_cb_value", U8 * 10),
                ("point_cb_scaling", U8 * 10), ("num_cr_points", U8), ("point_cr_value", U8 * 10), ("point_cr_scaling", U8 * 10),
                ("grain_scaling_minus_8", U8), ("ar_coeff_lag", U8), ("ar_coeffs_y_plus_128", U8 * 24), ("ar_coeffs_cb_plus_128", U8 * 25),
                ("ar_coeffs_cr_plus_128", U8 * 25), ("ar_coeff_shift_minus_6", U8), ("grain_scale_shift", U8), ("cb_mult", U8), ("cb_luma_mult", U8),
                ("cr_mult", U8), ("cr_luma_mult", U8), ("overlap_flag", U8), ("clip_to_restricted_range", U8), ("grain_seed", U16),
                ("cb_offset", U16), ("cr_offset", U16)]


def _to_py(v):
    if isinstance(v, ctypes.Structure):
        return {n: _to_py(getattr(v, n)) for n, _ in v._fields_}
    if hasattr(v, "__len__"):
        return [_to_py(x) for x in v]
    return int(v)


class FrameHeader(ctypes.Structure):
    _fields_ = ([(n, U8) for n in ("show_existing_frame", "frame_to_show_map_idx", "frame_type", "show_frame", "showable_frame", "error_resilient_mode",
                                   "disable_cdf_update", "allow_screen_content_tools", "force_integer_mv", "frame_size_override_flag",
                                   "primary_ref_frame", "refresh_frame_flags", "order_hint", "frame_refs_short_signaling", "allow_intrabc",
                                   "allow_high_precision_mv", "is_filter_switchable", "interpolation_filter", "is_motion_mode_switchable",
                                   "use_ref_frame_mvs", "disable_frame_end_update_cdf", "allow_warped_motion", "reduced_tx_set", "tx_mode_select",
                                   "reference_select", "skip_mode_present")] +
                [("skip_mode_frame", U8 * 2), ("ref_frame_idx", I8 * 7), ("ref_order_hint", U8 * 8),
                 ("current_frame_id", U32), ("display_frame_id", U32), ("frame_presentation_time", U32),
                 ("frame_width", U32), ("frame_height", U32), ("upscaled_width", U32), ("render_width", U32), ("render_height", U32),
                 ("use_superres", U8), ("superres_denom", U8), ("mi_cols", U32), ("mi_rows", U32),
                 ("uniform_tile_spacing_flag", U8), ("tile_cols", U8), ("tile_rows", U8), ("tile_cols_log2", U8), ("tile_rows_log2", U8),
                 ("tile_size_bytes", U8), ("context_update_tile_id", U16), ("mi_col_starts", U16 * 65), ("mi_row_starts", U16 * 65),
                 ("base_q_idx", U8), ("using_qmatrix", U8), ("qm_y", U8), ("qm_u", U8), ("qm_v", U8), ("diff_uv_delta", U8),
                 ("delta_q_y_dc", I8), ("delta_q_u_dc", I8), ("delta_q_u_ac", I8), ("delta_q_v_dc", I8), ("delta_q_v_ac", I8),
                 ("segmentation_enabled", U8), ("segmentation_update_map", U8), ("segmentation_temporal_update", U8), ("segmentation_update_data", U8),
                 ("feature_enabled", (U8 * 8) * 8), ("seg_id_pre_skip", U8), ("last_active_seg_id", U8), ("feature_data", (I16 * 8) * 8),
                 ("delta_q_present", U8), ("delta_q_res", U8), ("delta_lf_present", U8), ("delta_lf_res", U8), ("delta_lf_multi", U8),
                 ("coded_lossless", U8), ("all_lossless", U8),
                 ("loop_filter_level", U8 * 4), ("loop_filter_sharpness", U8), ("loop_filter_delta_enabled", U8), ("loop_filter_delta_update", U8),
                 ("loop_filter_ref_deltas", I8 * 8), ("loop_filter_mode_deltas", I8 * 2),
                 ("cdef_damping", U8), ("cdef_bits", U8), ("cdef_y_pri_strength", U8 * 8), ("cdef_y_sec_strength", U8 * 8),
                 ("cdef_uv_pri_strength", U8 * 8), ("cdef_uv_sec_strength", U8 * 8),
                 ("frame_restoration_type", U8 * 3), ("lr_unit_shift", U8), ("lr_uv_shift", U8), ("loop_restoration_size", U16 * 3),
                 ("gm_type", U8 * 7), ("gm_params", (I32 * 6) * 7), ("film_grain", FilmGrainSyntax),
                 ("buffer_removal_time_present", U8), ("buffer_removal_time", U32 * 32), ("header_bits", U32)])

    def as_dict(self):
        return _to_py(self)


def sequence_header_struct(d):
    """dict (as from parse_sequence_header or the test writer) -> SequenceHeader"""
    h = SequenceHeader()
    for name, _ in h._fields_:
        v = d.get(name, 0)
        if isinstance(v, list):
            arr = getattr(h, name)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(h, name, v)
    return h


class Parser:
    """ea1_av1_parser"""

    def __init__(self):
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        L.ea1_av1_parser_create.argtypes = [ctypes.c_void_p]
        _lib.check(L.ea1_av1_parser_create(ctypes.byref(self._h)), "ea1_av1_parser_create")

    def set_sequence_header(self, seq_struct):
        L = _lib.lib()
        L.ea1_av1_parser_set_sequence_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(L.ea1_av1_parser_set_sequence_header(self._h, ctypes.byref(seq_struct)), "ea1_av1_parser_set_sequence_header")

    def frame_header(self, payload, temporal_id=0, spatial_id=0):
        """-> (rc, FrameHeader)"""
        L = _lib.lib()
        buf = (ctypes.c_uint8 * max(len(payload), 1)).from_buffer_copy(bytes(payload) or b"\0")
        h = FrameHeader()
        L.ea1_av1_parser_frame_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        rc = L.ea1_av1_parser_frame_header(self._h, buf, len(payload), temporal_id, spatial_id, ctypes.byref(h))
        return rc, h

    def end_frame(self, h):
        L = _lib.lib()
        L.ea1_av1_parser_end_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(L.ea1_av1_parser_end_frame(self._h, ctypes.byref(h)), "ea1_av1_parser_end_frame")

    def close(self):
        if self._h:
            L = _lib.lib()
            L.ea1_av1_parser_destroy.argtypes = [ctypes.c_void_p]
            L.ea1_av1_parser_destroy.restype = None
            L.ea1_av1_parser_destroy(self._h)
            self._h = ctypes.c_void_p()


class Tile(ctypes.Structure):
    _fields_ = [("tile_num", U16), ("tile_row", U16), ("tile_col", U16), ("pad", U16), ("mi_row_start", U32), ("mi_row_end", U32),
                ("mi_col_start", U32), ("mi_col_end", U32), ("offset", ctypes.c_size_t), ("size", ctypes.c_size_t)]


def parse_tile_group(hdr_struct, payload, cap=4096):
    """-> (rc, [dict per tile])"""
    L = _lib.lib()
    buf = (ctypes.c_uint8 * max(len(payload), 1)).from_buffer_copy(bytes(payload) or b"\0")
    tiles = (Tile * cap)()
    n = ctypes.c_int()
    L.ea1_av1_parse_tile_group.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rc = L.ea1_av1_parse_tile_group(ctypes.byref(hdr_struct), buf, len(payload), tiles, cap, ctypes.byref(n))
    return rc, [_to_py(tiles[i]) for i in range(min(n.value, cap))]


def next_frame(parser, data, cap=4096):
    """-> (rc, consumed, FrameHeader, [tile dicts]) for the first complete frame in `data`"""
    L = _lib.lib()
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
    h = FrameHeader()
    tiles = (Tile * cap)()
    n = ctypes.c_int()
    used = ctypes.c_size_t()
    L.ea1_av1_parser_next_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p]
    rc = L.ea1_av1_parser_next_frame(parser._h, buf, len(data), ctypes.byref(used), ctypes.byref(h), tiles, cap, ctypes.byref(n))
    return rc, used.value, h, [_to_py(tiles[i]) for i in range(min(n.value, cap))]


class Msac(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos_bits", ctypes.c_size_t), ("value", U32), ("range", U32),
                ("max_bits", ctypes.c_int64), ("disable_update", ctypes.c_int)]


class SymbolDecoder:
    """ea1_msac over a bytes object"""

    def __init__(self, data, disable_cdf_update=False):
        L = _lib.lib()
        self._buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        self._s = Msac()
        L.ea1_msac_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        L.ea1_msac_symbol.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ea1_msac_bool.argtypes = [ctypes.c_void_p]
        L.ea1_msac_literal.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ea1_msac_literal.restype = ctypes.c_uint32
        L.ea1_msac_bits_left.argtypes = [ctypes.c_void_p]
        L.ea1_msac_bits_left.restype = ctypes.c_int64
        _lib.check(L.ea1_msac_init(ctypes.byref(self._s), self._buf, len(data), int(disable_cdf_update)), "ea1_msac_init")
        self._L = L

    def symbol(self, cdf):
        """cdf: numpy uint16 array of N + 1 entries (adapted in place) -> symbol"""
        return self._L.ea1_msac_symbol(ctypes.byref(self._s), cdf.ctypes.data, len(cdf) - 1)

    def bool(self):
        return self._L.ea1_msac_bool(ctypes.byref(self._s))

    def literal(self, n):
        return self._L.ea1_msac_literal(ctypes.byref(self._s), n)

    def bits_left(self):
        return self._L.ea1_msac_bits_left(ctypes.byref(self._s))
