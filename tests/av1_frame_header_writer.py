"""Independent WRITER of AV1 uncompressed_header() (AV1 specification 5.9) for the parser tests: emits
the bits of a frame header from a dict of syntax elements and keeps its own mirror of the reference
slots (what the header syntax depends on).  No AV1 encoder or stream exists in this environment."""
import numpy as np

from tests.av1_obu_writer import BitWriter

KEY, INTER, INTRA_ONLY, SWITCH = 0, 1, 2, 3
SEG_BITS = [8, 6, 6, 6, 6, 3, 0, 0]
SEG_SIGNED = [1, 1, 1, 1, 1, 0, 0, 0]
SEG_MAX = [255, 63, 63, 63, 63, 7, 0, 0]
DEFAULT_REF_DELTAS = [1, 0, 0, 0, -1, 0, -1, -1]


def default_gm():
    return [[(1 << 16) if i % 3 == 2 else 0 for i in range(6)] for _ in range(7)]


class Bits(BitWriter):
    def su(self, n, v):
        self.f(n, v & ((1 << n) - 1))

    def ns(self, n, v):
        w = n.bit_length()
        m = (1 << w) - n
        if v < m:
            self.f(w - 1, v) if w > 1 else None
        else:
            self.f(w - 1, (v + m) >> 1) if w > 1 else None
            self.f(1, (v + m) & 1)

    def subexp(self, num_syms, v):
        i, mk, k = 0, 0, 3
        while True:
            b2 = k + i - 1 if i else k
            a = 1 << b2
            if num_syms <= mk + 3 * a:
                self.ns(num_syms - mk, v - mk)
                return
            if v < mk + a:
                self.f(1, 0)
                self.f(b2, v - mk)
                return
            self.f(1, 1)
            i += 1
            mk += a

    def unsigned_subexp_with_ref(self, mx, r, value):
        def recenter(r, x):
            if x > 2 * r:
                return x
            return (x - r) << 1 if x >= r else ((r - x) << 1) - 1
        self.subexp(mx, recenter(r, value) if (r << 1) <= mx else recenter(mx - 1 - r, mx - 1 - value))


def tile_log2(blk, target):
    k = 0
    while (blk << k) < target:
        k += 1
    return k


class Slots:
    """mirror of the decoder's reference slots"""

    def __init__(self):
        self.s = [None] * 8

    def refresh(self, hdr):
        entry = dict(order_hint=hdr["order_hint"], upscaled_width=hdr["upscaled_width"], frame_width=hdr["frame_width"],
                     frame_height=hdr["frame_height"], render_width=hdr["render_width"], render_height=hdr["render_height"],
                     frame_type=hdr["frame_type"], gm=[list(r) for r in hdr["gm_params"]], ref_deltas=list(hdr["loop_filter_ref_deltas"]),
                     mode_deltas=list(hdr["loop_filter_mode_deltas"]), feature_enabled=[list(r) for r in hdr["feature_enabled"]],
                     feature_data=[list(r) for r in hdr["feature_data"]], grain=dict(hdr["film_grain"]), frame_id=hdr.get("current_frame_id", 0),
                     showable_frame=hdr["showable_frame"])
        for i in range(8):
            if hdr["refresh_frame_flags"] & (1 << i):
                self.s[i] = entry


def write_film_grain(w, seq, hdr, slots):
    g = hdr["film_grain"]
    if not seq["film_grain_params_present"] or (not hdr["show_frame"] and not hdr["showable_frame"]):
        return
    w.f(1, g["apply_grain"])
    if not g["apply_grain"]:
        return
    w.f(16, g["grain_seed"])
    if hdr["frame_type"] == INTER:
        w.f(1, g["update_grain"])
    if not g["update_grain"]:
        w.f(3, g["film_grain_params_ref_idx"])
        return
    w.f(4, g["num_y_points"])
    for i in range(g["num_y_points"]):
        w.f(8, g["point_y_value"][i])
        w.f(8, g["point_y_scaling"][i])
    if not seq["monochrome"]:
        w.f(1, g["chroma_scaling_from_luma"])
    if not (seq["monochrome"] or g["chroma_scaling_from_luma"] or (seq["subsampling_x"] == 1 and seq["subsampling_y"] == 1 and g["num_y_points"] == 0)):
        for c in ("cb", "cr"):
            w.f(4, g[f"num_{c}_points"])
            for i in range(g[f"num_{c}_points"]):
                w.f(8, g[f"point_{c}_value"][i])
                w.f(8, g[f"point_{c}_scaling"][i])
    w.f(2, g["grain_scaling_minus_8"])
    w.f(2, g["ar_coeff_lag"])
    npl = 2 * g["ar_coeff_lag"] * (g["ar_coeff_lag"] + 1)
    npc = npl
    if g["num_y_points"]:
        npc = npl + 1
        for i in range(npl):
            w.f(8, g["ar_coeffs_y_plus_128"][i])
    for c in ("cb", "cr"):
        if g["chroma_scaling_from_luma"] or g[f"num_{c}_points"]:
            for i in range(npc):
                w.f(8, g[f"ar_coeffs_{c}_plus_128"][i])
    w.f(2, g["ar_coeff_shift_minus_6"])
    w.f(2, g["grain_scale_shift"])
    for c in ("cb", "cr"):
        if g[f"num_{c}_points"]:
            w.f(8, g[f"{c}_mult"])
            w.f(8, g[f"{c}_luma_mult"])
            w.f(9, g[f"{c}_offset"])
    w.f(1, g["overlap_flag"])
    w.f(1, g["clip_to_restricted_range"])


def write_frame_header(seq, hdr, slots, temporal_id=0, spatial_id=0):
    """hdr: syntax elements + the derived values the syntax needs (sizes, lossless flags); -> bytes of the
    header (padded with zero bits to a byte)"""
    w = Bits()
    id_len = seq.get("additional_frame_id_length", 0) + seq.get("delta_frame_id_length", 0) if seq.get("frame_id_numbers_present") else 0
    ft = hdr["frame_type"]
    intra = ft in (KEY, INTRA_ONLY)
    num_planes = 1 if seq["monochrome"] else 3
    if not seq["reduced_still_picture_header"]:
        w.f(1, hdr.get("show_existing_frame", 0))
        if hdr.get("show_existing_frame"):
            w.f(3, hdr["frame_to_show_map_idx"])
            if seq.get("decoder_model_info_present") and not seq.get("equal_picture_interval"):
                w.f(seq["frame_presentation_time_length"], hdr["frame_presentation_time"])
            if id_len:
                w.f(id_len, hdr["display_frame_id"])
            nbits = len(w.bits)
            while len(w.bits) % 8:
                w.bits.append(0)
            return w.bytes(), nbits
        w.f(2, ft)
        w.f(1, hdr["show_frame"])
        if hdr["show_frame"] and seq.get("decoder_model_info_present") and not seq.get("equal_picture_interval"):
            w.f(seq["frame_presentation_time_length"], hdr["frame_presentation_time"])
        if not hdr["show_frame"]:
            w.f(1, hdr["showable_frame"])
        if not (ft == SWITCH or (ft == KEY and hdr["show_frame"])):
            w.f(1, hdr["error_resilient_mode"])
    w.f(1, hdr["disable_cdf_update"])
    if seq["seq_force_screen_content_tools"] == 2:
        w.f(1, hdr["allow_screen_content_tools"])
    if hdr["allow_screen_content_tools"] and seq["seq_force_integer_mv"] == 2:
        w.f(1, hdr["coded_force_integer_mv"])
    if id_len:
        w.f(id_len, hdr["current_frame_id"])
    if ft != SWITCH and not seq["reduced_still_picture_header"]:
        w.f(1, hdr["frame_size_override_flag"])
    if seq.get("order_hint_bits", 0):
        w.f(seq["order_hint_bits"], hdr["order_hint"])
    if not (intra or hdr["error_resilient_mode"]):
        w.f(3, hdr["primary_ref_frame"])
    if seq.get("decoder_model_info_present"):
        w.f(1, hdr["buffer_removal_time_present"])
        if hdr["buffer_removal_time_present"]:
            for op in range(seq["num_operating_points"]):
                if seq["decoder_model_present_for_op"][op]:
                    idc = seq["operating_point_idc"][op]
                    if idc == 0 or (((idc >> temporal_id) & 1) and ((idc >> (spatial_id + 8)) & 1)):
                        w.f(seq["buffer_removal_time_length"], hdr["buffer_removal_time"][op])
    if not (ft == SWITCH or (ft == KEY and hdr["show_frame"])):
        w.f(8, hdr["refresh_frame_flags"])
    if (not intra or hdr["refresh_frame_flags"] != 0xFF) and hdr["error_resilient_mode"] and seq.get("enable_order_hint"):
        for i in range(8):
            w.f(seq["order_hint_bits"], hdr["ref_order_hint"][i])

    def superres():
        if seq["enable_superres"]:
            w.f(1, hdr["use_superres"])
            if hdr["use_superres"]:
                w.f(3, hdr["superres_denom"] - 9)

    def frame_size():
        if hdr["frame_size_override_flag"]:
            w.f(seq["frame_width_bits"], hdr["upscaled_width"] - 1)
            w.f(seq["frame_height_bits"], hdr["frame_height"] - 1)
        superres()

    def render_size():
        diff = hdr["render_width"] != hdr["upscaled_width"] or hdr["render_height"] != hdr["frame_height"]
        w.f(1, int(diff))
        if diff:
            w.f(16, hdr["render_width"] - 1)
            w.f(16, hdr["render_height"] - 1)

    if intra:
        frame_size()
        render_size()
        if hdr["allow_screen_content_tools"] and hdr["upscaled_width"] == hdr["frame_width"]:
            w.f(1, hdr["allow_intrabc"])
    else:
        if seq.get("enable_order_hint"):
            w.f(1, hdr["frame_refs_short_signaling"])
            if hdr["frame_refs_short_signaling"]:
                w.f(3, hdr["last_frame_idx"])
                w.f(3, hdr["gold_frame_idx"])
        for i in range(7):
            if not hdr["frame_refs_short_signaling"]:
                w.f(3, hdr["ref_frame_idx"][i])
            if id_len:
                w.f(seq["delta_frame_id_length"], hdr["delta_frame_id_minus_1"][i])
        if hdr["frame_size_override_flag"] and not hdr["error_resilient_mode"]:
            found = hdr.get("found_ref", -1)
            for i in range(7):
                w.f(1, int(i == found))
                if i == found:
                    break
            if found < 0:
                frame_size()
                render_size()
            else:
                superres()
        else:
            frame_size()
            render_size()
        if not hdr["force_integer_mv"]:
            w.f(1, hdr["allow_high_precision_mv"])
        w.f(1, hdr["is_filter_switchable"])
        if not hdr["is_filter_switchable"]:
            w.f(2, hdr["interpolation_filter"])
        w.f(1, hdr["is_motion_mode_switchable"])
        if not (hdr["error_resilient_mode"] or not seq.get("enable_ref_frame_mvs")):
            w.f(1, hdr["use_ref_frame_mvs"])
    if not (seq["reduced_still_picture_header"] or hdr["disable_cdf_update"]):
        w.f(1, hdr["disable_frame_end_update_cdf"])
    # tile_info
    sb_shift = 5 if seq["use_128x128_superblock"] else 4
    sb_cols = (hdr["mi_cols"] + (1 << sb_shift) - 1) >> sb_shift
    sb_rows = (hdr["mi_rows"] + (1 << sb_shift) - 1) >> sb_shift
    sb_size = sb_shift + 2
    max_w_sb = 4096 >> sb_size
    max_area_sb = (4096 * 2304) >> (2 * sb_size)
    min_l2c = tile_log2(max_w_sb, sb_cols)
    max_l2c = tile_log2(1, min(sb_cols, 64))
    max_l2r = tile_log2(1, min(sb_rows, 64))
    min_l2t = max(min_l2c, tile_log2(max_area_sb, sb_rows * sb_cols))
    w.f(1, hdr["uniform_tile_spacing_flag"])
    if hdr["uniform_tile_spacing_flag"]:
        for l2 in range(min_l2c, max_l2c):
            more = int(hdr["tile_cols_log2"] > l2)
            w.f(1, more)
            if not more:
                break
        min_l2r = max(min_l2t - hdr["tile_cols_log2"], 0)
        for l2 in range(min_l2r, max_l2r):
            more = int(hdr["tile_rows_log2"] > l2)
            w.f(1, more)
            if not more:
                break
    else:
        start, widest = 0, 0
        for size in hdr["tile_widths_sb"]:
            w.ns(min(sb_cols - start, max_w_sb), size - 1)
            widest = max(widest, size)
            start += size
        assert start == sb_cols
        area = (sb_rows * sb_cols) >> (min_l2t + 1) if min_l2t > 0 else sb_rows * sb_cols
        max_h_sb = max(area // widest, 1)
        start = 0
        for size in hdr["tile_heights_sb"]:
            w.ns(min(sb_rows - start, max_h_sb), size - 1)
            start += size
        assert start == sb_rows
    if hdr["tile_cols_log2"] or hdr["tile_rows_log2"]:
        w.f(hdr["tile_cols_log2"] + hdr["tile_rows_log2"], hdr["context_update_tile_id"])
        w.f(2, hdr["tile_size_bytes"] - 1)

    def delta_q(v):
        w.f(1, int(v != 0))
        if v:
            w.su(7, v)
    w.f(8, hdr["base_q_idx"])
    delta_q(hdr["delta_q_y_dc"])
    if num_planes > 1:
        if seq["separate_uv_delta_q"]:
            w.f(1, hdr["diff_uv_delta"])
        delta_q(hdr["delta_q_u_dc"])
        delta_q(hdr["delta_q_u_ac"])
        if hdr["diff_uv_delta"]:
            delta_q(hdr["delta_q_v_dc"])
            delta_q(hdr["delta_q_v_ac"])
    w.f(1, hdr["using_qmatrix"])
    if hdr["using_qmatrix"]:
        w.f(4, hdr["qm_y"])
        w.f(4, hdr["qm_u"])
        if seq["separate_uv_delta_q"]:
            w.f(4, hdr["qm_v"])
    # segmentation
    w.f(1, hdr["segmentation_enabled"])
    if hdr["segmentation_enabled"]:
        if hdr["primary_ref_frame"] != 7:
            w.f(1, hdr["segmentation_update_map"])
            if hdr["segmentation_update_map"]:
                w.f(1, hdr["segmentation_temporal_update"])
            w.f(1, hdr["segmentation_update_data"])
        if hdr["segmentation_update_data"]:
            for i in range(8):
                for j in range(8):
                    w.f(1, hdr["feature_enabled"][i][j])
                    if hdr["feature_enabled"][i][j]:
                        if SEG_SIGNED[j]:
                            w.su(1 + SEG_BITS[j], hdr["coded_feature_value"][i][j])
                        elif SEG_BITS[j]:
                            w.f(SEG_BITS[j], hdr["coded_feature_value"][i][j])
    if hdr["base_q_idx"] > 0:
        w.f(1, hdr["delta_q_present"])
    if hdr["delta_q_present"]:
        w.f(2, hdr["delta_q_res"])
        if not hdr["allow_intrabc"]:
            w.f(1, hdr["delta_lf_present"])
        if hdr["delta_lf_present"]:
            w.f(2, hdr["delta_lf_res"])
            w.f(1, hdr["delta_lf_multi"])
    # loop filter
    if not (hdr["coded_lossless"] or hdr["allow_intrabc"]):
        w.f(6, hdr["loop_filter_level"][0])
        w.f(6, hdr["loop_filter_level"][1])
        if num_planes > 1 and (hdr["loop_filter_level"][0] or hdr["loop_filter_level"][1]):
            w.f(6, hdr["loop_filter_level"][2])
            w.f(6, hdr["loop_filter_level"][3])
        w.f(3, hdr["loop_filter_sharpness"])
        w.f(1, hdr["loop_filter_delta_enabled"])
        if hdr["loop_filter_delta_enabled"]:
            w.f(1, hdr["loop_filter_delta_update"])
            if hdr["loop_filter_delta_update"]:
                for i in range(8):
                    upd = hdr["update_ref_delta"][i]
                    w.f(1, upd)
                    if upd:
                        w.su(7, hdr["loop_filter_ref_deltas"][i])
                for i in range(2):
                    upd = hdr["update_mode_delta"][i]
                    w.f(1, upd)
                    if upd:
                        w.su(7, hdr["loop_filter_mode_deltas"][i])
    # cdef
    if not (hdr["coded_lossless"] or hdr["allow_intrabc"] or not seq["enable_cdef"]):
        w.f(2, hdr["cdef_damping"] - 3)
        w.f(2, hdr["cdef_bits"])
        for i in range(1 << hdr["cdef_bits"]):
            w.f(4, hdr["cdef_y_pri_strength"][i])
            w.f(2, min(hdr["cdef_y_sec_strength"][i], 3))
            if num_planes > 1:
                w.f(4, hdr["cdef_uv_pri_strength"][i])
                w.f(2, min(hdr["cdef_uv_sec_strength"][i], 3))
    # loop restoration
    if not (hdr["all_lossless"] or hdr["allow_intrabc"] or not seq["enable_restoration"]):
        inv = {0: 0, 3: 1, 1: 2, 2: 3}
        uses, uses_c = False, False
        for i in range(num_planes):
            w.f(2, inv[hdr["frame_restoration_type"][i]])
            if hdr["frame_restoration_type"][i]:
                uses = True
                uses_c |= i > 0
        if uses:
            if seq["use_128x128_superblock"]:
                w.f(1, hdr["lr_unit_shift"] - 1)
            else:
                w.f(1, int(hdr["lr_unit_shift"] > 0))
                if hdr["lr_unit_shift"]:
                    w.f(1, hdr["lr_unit_shift"] - 1)
            if seq["subsampling_x"] and seq["subsampling_y"] and uses_c:
                w.f(1, hdr["lr_uv_shift"])
    if not hdr["coded_lossless"]:
        w.f(1, hdr["tx_mode_select"])
    if not intra:
        w.f(1, hdr["reference_select"])
    if hdr.get("skip_mode_allowed"):
        w.f(1, hdr["skip_mode_present"])
    if not (intra or hdr["error_resilient_mode"] or not seq.get("enable_warped_motion")):
        w.f(1, hdr["allow_warped_motion"])
    w.f(1, hdr["reduced_tx_set"])
    # global motion
    if not intra:
        prev = default_gm()
        if hdr["primary_ref_frame"] != 7:
            prev = slots.s[hdr["ref_frame_idx"][hdr["primary_ref_frame"]]]["gm"]
        for ref in range(7):
            t = hdr["gm_type"][ref]
            w.f(1, int(t != 0))
            if t:
                w.f(1, int(t == 2))
                if t != 2:
                    w.f(1, int(t == 1))

            def param(idx):
                abs_bits, prec_bits = 12, 15
                if idx < 2:
                    if t == 1:
                        abs_bits = 9 - (not hdr["allow_high_precision_mv"])
                        prec_bits = 3 - (not hdr["allow_high_precision_mv"])
                    else:
                        abs_bits, prec_bits = 12, 6
                prec_diff = 16 - prec_bits
                rnd = (1 << 16) if idx % 3 == 2 else 0
                sub = (1 << prec_bits) if idx % 3 == 2 else 0
                mx = 1 << abs_bits
                r = (prev[ref][idx] >> prec_diff) - sub
                v = (hdr["gm_params"][ref][idx] - rnd) >> prec_diff
                w.unsigned_subexp_with_ref(2 * mx + 1, r + mx, v + mx)
            if t >= 2:
                param(2)
                param(3)
                if t == 3:
                    param(4)
                    param(5)
            if t >= 1:
                param(0)
                param(1)
    write_film_grain(w, seq, hdr, slots)
    nbits = len(w.bits)
    while len(w.bits) % 8:
        w.bits.append(0)
    return w.bytes(), nbits
