"""AV1 frame header parser (second table-free slice) against the independent writer
of tests/av1_frame_header_writer.py: random sequences of key / inter / intra-only / switch / shown-existing
frames whose headers depend on the reference-slot state the parser has to keep.  Parity with dav1d is
UNPINNED (dav1d absent); both sides follow AV1 specification 5.9 / 7.8 / 7.20 / 7.21."""
import numpy as np
import pytest

from tests import av1_frame_header_writer as fw
from tests import av1_obu_writer as wr

KEY, INTER, INTRA_ONLY, SWITCH = 0, 1, 2, 3


@pytest.fixture(scope="module")
def obu():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import obu
    return obu


def make_seq(rng, **fixed):
    while True:
        s = wr.random_sequence_header(rng)
        if s["reduced_still_picture_header"] or s["still_picture"]:
            continue
        break
    s["frame_width_bits"] = s["frame_height_bits"] = 12
    s["max_width"] = int(rng.integers(64, 2049))
    s["max_height"] = int(rng.integers(64, 1201))
    if s.get("frame_id_numbers_present"):
        s["delta_frame_id_length"] = int(rng.integers(6, 11))
        s["additional_frame_id_length"] = int(rng.integers(1, 7))
    if s.get("decoder_model_info_present"):
        s["frame_presentation_time_length"] = min(s["frame_presentation_time_length"], 32)
    s.update(fixed)
    for k in ("enable_order_hint", "enable_ref_frame_mvs", "enable_jnt_comp", "order_hint_bits", "frame_id_numbers_present", "decoder_model_info_present",
              "equal_picture_interval", "enable_warped_motion", "separate_uv_delta_q", "additional_frame_id_length", "delta_frame_id_length",
              "frame_presentation_time_length", "buffer_removal_time_length", "timing_info_present"):
        s.setdefault(k, 0)
    if not s["enable_order_hint"]:
        s["order_hint_bits"] = 0
    return s


def rel_dist(seq, a, b):
    if not seq["enable_order_hint"]:
        return 0
    diff = a - b
    m = 1 << (seq["order_hint_bits"] - 1)
    return (diff & (m - 1)) - (diff & m)


def skip_mode(seq, hdr, slots):
    """-> (allowed, [f0, f1])"""
    if hdr["frame_type"] in (KEY, INTRA_ONLY) or not hdr["reference_select"] or not seq["enable_order_hint"]:
        return False, [0, 0]
    hints = [slots.s[i]["order_hint"] for i in hdr["ref_frame_idx"]]
    fwd = bwd = -1
    for i, hint in enumerate(hints):
        d = rel_dist(seq, hint, hdr["order_hint"])
        if d < 0 and (fwd < 0 or rel_dist(seq, hint, hints[fwd]) > 0):
            fwd = i
        elif d > 0 and (bwd < 0 or rel_dist(seq, hint, hints[bwd]) < 0):
            bwd = i
    if fwd < 0:
        return False, [0, 0]
    if bwd >= 0:
        return True, [1 + min(fwd, bwd), 1 + max(fwd, bwd)]
    second = -1
    for i, hint in enumerate(hints):
        if rel_dist(seq, hint, hints[fwd]) < 0 and (second < 0 or rel_dist(seq, hint, hints[second]) > 0):
            second = i
    if second < 0:
        return False, [0, 0]
    return True, [1 + min(fwd, second), 1 + max(fwd, second)]


def random_grain(rng, seq, hdr, slots):
    g = dict(apply_grain=0, update_grain=0, film_grain_params_ref_idx=0, num_y_points=0, point_y_value=[0] * 14, point_y_scaling=[0] * 14,
             chroma_scaling_from_luma=0, num_cb_points=0, point_cb_value=[0] * 10, point_cb_scaling=[0] * 10, num_cr_points=0,
             point_cr_value=[0] * 10, point_cr_scaling=[0] * 10, grain_scaling_minus_8=0, ar_coeff_lag=0, ar_coeffs_y_plus_128=[0] * 24,
             ar_coeffs_cb_plus_128=[0] * 25, ar_coeffs_cr_plus_128=[0] * 25, ar_coeff_shift_minus_6=0, grain_scale_shift=0, cb_mult=0,
             cb_luma_mult=0, cr_mult=0, cr_luma_mult=0, overlap_flag=0, clip_to_restricted_range=0, grain_seed=0, cb_offset=0, cr_offset=0)
    if not seq["film_grain_params_present"] or (not hdr["show_frame"] and not hdr["showable_frame"]):
        return g
    g["apply_grain"] = int(rng.integers(4) > 0)
    if not g["apply_grain"]:
        return g
    g["grain_seed"] = int(rng.integers(1 << 16))
    g["update_grain"] = 1
    if hdr["frame_type"] == INTER:
        cands = [i for i in set(hdr["ref_frame_idx"]) if slots.s[i]["grain"]["apply_grain"]]
        if cands and rng.integers(2):
            idx = int(rng.choice(cands))
            seed = g["grain_seed"]
            g = dict(slots.s[idx]["grain"])
            g.update(grain_seed=seed, apply_grain=1, update_grain=0, film_grain_params_ref_idx=idx)
            return g
    g["num_y_points"] = int(rng.integers(0, 15))
    for i, v in enumerate(np.sort(rng.choice(256, g["num_y_points"], replace=False))):
        g["point_y_value"][i] = int(v)
        g["point_y_scaling"][i] = int(rng.integers(256))
    g["chroma_scaling_from_luma"] = 0 if seq["monochrome"] else int(rng.integers(2))
    if not (seq["monochrome"] or g["chroma_scaling_from_luma"] or (seq["subsampling_x"] == 1 and seq["subsampling_y"] == 1 and g["num_y_points"] == 0)):
        for c in ("cb", "cr"):
            g[f"num_{c}_points"] = int(rng.integers(0, 11))
            for i, v in enumerate(np.sort(rng.choice(256, g[f"num_{c}_points"], replace=False))):
                g[f"point_{c}_value"][i] = int(v)
                g[f"point_{c}_scaling"][i] = int(rng.integers(256))
        # AV1 spec 6.8.20: with 4:2:0, num_cb_points and num_cr_points are both zero or both not (dav1d refuses a header that breaks it, and so does this
        # repository's parser since round 5): the plane without points takes the other plane's
        if seq["subsampling_x"] == 1 and seq["subsampling_y"] == 1 and bool(g["num_cb_points"]) != bool(g["num_cr_points"]):
            src, dst = ("cb", "cr") if g["num_cb_points"] else ("cr", "cb")
            g[f"num_{dst}_points"] = g[f"num_{src}_points"]
            g[f"point_{dst}_value"], g[f"point_{dst}_scaling"] = list(g[f"point_{src}_value"]), list(g[f"point_{src}_scaling"])
    g["grain_scaling_minus_8"] = int(rng.integers(4))
    g["ar_coeff_lag"] = int(rng.integers(4))
    npl = 2 * g["ar_coeff_lag"] * (g["ar_coeff_lag"] + 1)
    npc = npl + (1 if g["num_y_points"] else 0)
    if g["num_y_points"]:
        for i in range(npl):
            g["ar_coeffs_y_plus_128"][i] = int(rng.integers(256))
    for c in ("cb", "cr"):
        if g["chroma_scaling_from_luma"] or g[f"num_{c}_points"]:
            for i in range(npc):
                g[f"ar_coeffs_{c}_plus_128"][i] = int(rng.integers(256))
    g["ar_coeff_shift_minus_6"] = int(rng.integers(4))
    g["grain_scale_shift"] = int(rng.integers(4))
    for c in ("cb", "cr"):
        if g[f"num_{c}_points"]:
            g[f"{c}_mult"], g[f"{c}_luma_mult"], g[f"{c}_offset"] = int(rng.integers(256)), int(rng.integers(256)), int(rng.integers(512))
    g["overlap_flag"], g["clip_to_restricted_range"] = int(rng.integers(2)), int(rng.integers(2))
    return g


def random_header(rng, seq, slots, ft, frame_no, fixed=None):
    fixed = fixed or {}
    h = {}
    intra = ft in (KEY, INTRA_ONLY)
    planes = 1 if seq["monochrome"] else 3
    h["frame_type"] = ft
    h["show_frame"] = fixed.get("show_frame", int(rng.integers(4) > 0))
    h["showable_frame"] = int(ft != KEY) if h["show_frame"] else int(rng.integers(2))
    h["error_resilient_mode"] = 1 if (ft == SWITCH or (ft == KEY and h["show_frame"])) else fixed.get("error_resilient_mode", int(rng.integers(4) == 0))
    if seq["decoder_model_info_present"] and not seq["equal_picture_interval"]:
        h["frame_presentation_time"] = int(rng.integers(1 << min(seq["frame_presentation_time_length"], 31))) if h["show_frame"] else 0
    h["disable_cdf_update"] = int(rng.integers(2))
    h["allow_screen_content_tools"] = int(rng.integers(2)) if seq["seq_force_screen_content_tools"] == 2 else seq["seq_force_screen_content_tools"]
    h["coded_force_integer_mv"] = int(rng.integers(2))
    fim = 0
    if h["allow_screen_content_tools"]:
        fim = h["coded_force_integer_mv"] if seq["seq_force_integer_mv"] == 2 else seq["seq_force_integer_mv"]
    h["force_integer_mv"] = 1 if intra else fim
    id_len = seq["additional_frame_id_length"] + seq["delta_frame_id_length"] if seq["frame_id_numbers_present"] else 0
    if id_len:
        h["current_frame_id"] = (1000 + frame_no) % (1 << id_len)
    h["frame_size_override_flag"] = 1 if ft == SWITCH else fixed.get("frame_size_override_flag", int(rng.integers(2)))
    h["order_hint"] = (frame_no * 2 + int(rng.integers(2))) % (1 << seq["order_hint_bits"]) if seq["order_hint_bits"] else 0
    h["order_hint"] = fixed.get("order_hint", h["order_hint"])
    reset = ft == KEY and h["show_frame"]
    valid = [i for i in range(8) if slots.s[i] is not None and not reset]
    h["primary_ref_frame"] = 7 if (intra or h["error_resilient_mode"]) else fixed.get("primary_ref_frame", int(rng.choice([7, 7] + list(range(7)))))
    if seq["decoder_model_info_present"]:
        h["buffer_removal_time_present"] = int(rng.integers(2))
        h["buffer_removal_time"] = [0] * 32
        if h["buffer_removal_time_present"]:
            for op in range(seq["num_operating_points"]):
                idc = seq["operating_point_idc"][op]
                if seq["decoder_model_present_for_op"][op] and (idc == 0 or ((idc & 1) and ((idc >> 8) & 1))):
                    h["buffer_removal_time"][op] = int(rng.integers(1 << min(seq["buffer_removal_time_length"], 31)))
    if ft == SWITCH or (ft == KEY and h["show_frame"]):
        h["refresh_frame_flags"] = 0xFF
    else:
        h["refresh_frame_flags"] = fixed.get("refresh_frame_flags", int(rng.integers(0, 255)) if ft == INTRA_ONLY else int(rng.integers(256)))
    h["ref_order_hint"] = [0] * 8
    if (not intra or h["refresh_frame_flags"] != 0xFF) and h["error_resilient_mode"] and seq["enable_order_hint"]:
        h["ref_order_hint"] = [slots.s[i]["order_hint"] if (slots.s[i] is not None and not reset) else 0 for i in range(8)]
    # sizes
    h["use_superres"], h["superres_denom"] = 0, 8
    h["found_ref"] = -1
    h["ref_frame_idx"] = [0] * 7
    h["frame_refs_short_signaling"] = 0
    if not intra:
        h["ref_frame_idx"] = fixed.get("ref_frame_idx", [int(rng.choice(valid)) for _ in range(7)])
        for k in ("frame_refs_short_signaling", "last_frame_idx", "gold_frame_idx"):
            if k in fixed:
                h[k] = fixed[k]
        if id_len:
            h["delta_frame_id_minus_1"] = [((h["current_frame_id"] - slots.s[i]["frame_id"]) % (1 << id_len)) - 1 for i in h["ref_frame_idx"]]
    if not intra and h["frame_size_override_flag"] and not h["error_resilient_mode"] and rng.integers(2):
        h["found_ref"] = int(rng.integers(7))
        src = slots.s[h["ref_frame_idx"][h["found_ref"]]]
        h["upscaled_width"], h["frame_height"] = src["upscaled_width"], src["frame_height"]
        h["render_width"], h["render_height"] = src["render_width"], src["render_height"]
    else:
        if h["frame_size_override_flag"]:
            lo_w, lo_h = fixed.get("frame_size_range", (16, 16))    # (a test that reconstructs keeps the references within the 2 : 1 limit)
            h["upscaled_width"] = int(rng.integers(lo_w, seq["max_width"] + 1))
            h["frame_height"] = int(rng.integers(lo_h, seq["max_height"] + 1))
        else:
            h["upscaled_width"], h["frame_height"] = seq["max_width"], seq["max_height"]
        if rng.integers(3) == 0:
            h["render_width"], h["render_height"] = int(rng.integers(1, 65537)), int(rng.integers(1, 65537))
        else:
            h["render_width"], h["render_height"] = h["upscaled_width"], h["frame_height"]
    if seq["enable_superres"] and rng.integers(2):
        h["use_superres"], h["superres_denom"] = 1, int(rng.integers(9, 17))
    h["frame_width"] = (h["upscaled_width"] * 8 + h["superres_denom"] // 2) // h["superres_denom"]
    h["mi_cols"] = 2 * ((h["frame_width"] + 7) >> 3)
    h["mi_rows"] = 2 * ((h["frame_height"] + 7) >> 3)
    h["allow_intrabc"] = int(rng.integers(2)) if (intra and h["allow_screen_content_tools"] and h["upscaled_width"] == h["frame_width"]) else 0
    if not intra:
        h["allow_high_precision_mv"] = 0 if h["force_integer_mv"] else int(rng.integers(2))
        h["is_filter_switchable"] = int(rng.integers(2))
        h["interpolation_filter"] = 4 if h["is_filter_switchable"] else int(rng.integers(4))
        h["is_motion_mode_switchable"] = int(rng.integers(2))
        h["use_ref_frame_mvs"] = 0 if (h["error_resilient_mode"] or not seq["enable_ref_frame_mvs"]) else int(rng.integers(2))
    else:
        h.update(allow_high_precision_mv=0, is_filter_switchable=0, interpolation_filter=0, is_motion_mode_switchable=0, use_ref_frame_mvs=0)
    h["disable_frame_end_update_cdf"] = 1 if h["disable_cdf_update"] else int(rng.integers(2))
    # tiles
    sb_shift = 5 if seq["use_128x128_superblock"] else 4
    sb_cols = (h["mi_cols"] + (1 << sb_shift) - 1) >> sb_shift
    sb_rows = (h["mi_rows"] + (1 << sb_shift) - 1) >> sb_shift
    sb_size = sb_shift + 2
    max_w_sb, max_area_sb = 4096 >> sb_size, (4096 * 2304) >> (2 * sb_size)
    min_l2c = fw.tile_log2(max_w_sb, sb_cols)
    max_l2c, max_l2r = fw.tile_log2(1, min(sb_cols, 64)), fw.tile_log2(1, min(sb_rows, 64))
    min_l2t = max(min_l2c, fw.tile_log2(max_area_sb, sb_rows * sb_cols))
    h["uniform_tile_spacing_flag"] = int(rng.integers(3) > 0)
    if h["uniform_tile_spacing_flag"]:
        h["tile_cols_log2"] = int(rng.integers(min_l2c, max_l2c + 1))
        tw = (sb_cols + (1 << h["tile_cols_log2"]) - 1) >> h["tile_cols_log2"]
        col_starts = list(range(0, sb_cols, tw))
        min_l2r = max(min_l2t - h["tile_cols_log2"], 0)
        h["tile_rows_log2"] = int(rng.integers(min_l2r, max(max_l2r, min_l2r) + 1))
        th = (sb_rows + (1 << h["tile_rows_log2"]) - 1) >> h["tile_rows_log2"]
        row_starts = list(range(0, sb_rows, th))
    else:
        widths, start = [], 0
        while start < sb_cols:
            size = int(rng.integers(1, min(sb_cols - start, max_w_sb, 6) + 1))
            if len(widths) == 63:
                size = min(sb_cols - start, max_w_sb)
            widths.append(size)
            start += size
        area = (sb_rows * sb_cols) >> (min_l2t + 1) if min_l2t > 0 else sb_rows * sb_cols
        max_h_sb = max(area // max(widths), 1)
        heights, start = [], 0
        while start < sb_rows:
            size = int(rng.integers(1, min(sb_rows - start, max_h_sb, 6) + 1))
            if len(heights) == 63:
                size = min(sb_rows - start, max_h_sb)
            heights.append(size)
            start += size
        if sum(widths) != sb_cols or sum(heights) != sb_rows:   # the 64-tile cap did not fit: fall back to one tile per maximum
            widths = [min(max_w_sb, sb_cols - s0) for s0 in range(0, sb_cols, max_w_sb)]
            heights = [min(max_h_sb, sb_rows - s0) for s0 in range(0, sb_rows, max_h_sb)]
        h["tile_widths_sb"], h["tile_heights_sb"] = widths, heights
        col_starts = list(np.cumsum([0] + widths[:-1]))
        row_starts = list(np.cumsum([0] + heights[:-1]))
        h["tile_cols_log2"], h["tile_rows_log2"] = fw.tile_log2(1, len(widths)), fw.tile_log2(1, len(heights))
    h["tile_cols"], h["tile_rows"] = len(col_starts), len(row_starts)
    h["mi_col_starts"] = [int(s0) << sb_shift for s0 in col_starts] + [h["mi_cols"]]
    h["mi_row_starts"] = [int(s0) << sb_shift for s0 in row_starts] + [h["mi_rows"]]
    h["tile_size_bytes"], h["context_update_tile_id"] = 4, 0
    if h["tile_cols_log2"] or h["tile_rows_log2"]:
        h["context_update_tile_id"] = int(rng.integers(h["tile_cols"] * h["tile_rows"]))
        h["tile_size_bytes"] = int(rng.integers(1, 5))
    # quantisation
    dq = lambda: int(rng.integers(-64, 64)) if rng.integers(3) == 0 else 0
    h["base_q_idx"] = int(rng.choice([0, 0, int(rng.integers(1, 256))]))
    lossless_bias = rng.integers(3) == 0
    h["delta_q_y_dc"] = 0 if lossless_bias else dq()
    h["diff_uv_delta"] = int(rng.integers(2)) if (planes > 1 and seq["separate_uv_delta_q"]) else 0
    if planes > 1:
        h["delta_q_u_dc"], h["delta_q_u_ac"] = (0, 0) if lossless_bias else (dq(), dq())
        if h["diff_uv_delta"]:
            h["delta_q_v_dc"], h["delta_q_v_ac"] = (0, 0) if lossless_bias else (dq(), dq())
        else:
            h["delta_q_v_dc"], h["delta_q_v_ac"] = h["delta_q_u_dc"], h["delta_q_u_ac"]
    else:
        h.update(delta_q_u_dc=0, delta_q_u_ac=0, delta_q_v_dc=0, delta_q_v_ac=0)
    h["using_qmatrix"] = int(rng.integers(2))
    h["qm_y"] = h["qm_u"] = h["qm_v"] = 0
    if h["using_qmatrix"]:
        h["qm_y"], h["qm_u"] = int(rng.integers(16)), int(rng.integers(16))
        h["qm_v"] = int(rng.integers(16)) if seq["separate_uv_delta_q"] else h["qm_u"]
    # segmentation: starts from the primary reference frame's parameters
    prev = slots.s[h["ref_frame_idx"][h["primary_ref_frame"]]] if h["primary_ref_frame"] != 7 else None
    fe = [list(r) for r in prev["feature_enabled"]] if prev else [[0] * 8 for _ in range(8)]
    fd = [list(r) for r in prev["feature_data"]] if prev else [[0] * 8 for _ in range(8)]
    h["segmentation_enabled"] = int(rng.integers(2))
    h["segmentation_update_map"] = h["segmentation_temporal_update"] = h["segmentation_update_data"] = 0
    h["coded_feature_value"] = [[0] * 8 for _ in range(8)]
    if h["segmentation_enabled"]:
        if h["primary_ref_frame"] == 7:
            h["segmentation_update_map"], h["segmentation_update_data"] = 1, 1
        else:
            h["segmentation_update_map"] = int(rng.integers(2))
            h["segmentation_temporal_update"] = int(rng.integers(2)) if h["segmentation_update_map"] else 0
            h["segmentation_update_data"] = int(rng.integers(2))
        if h["segmentation_update_data"]:
            for i in range(8):
                for j in range(8):
                    fe[i][j] = int(rng.integers(4) == 0)
                    fd[i][j] = 0
                    if fe[i][j]:
                        if fw.SEG_SIGNED[j]:
                            v = int(rng.integers(-(1 << fw.SEG_BITS[j]), 1 << fw.SEG_BITS[j]))
                            fd[i][j] = max(-fw.SEG_MAX[j], min(fw.SEG_MAX[j], v))
                        else:
                            v = int(rng.integers(1 << fw.SEG_BITS[j])) if fw.SEG_BITS[j] else 0
                            fd[i][j] = min(fw.SEG_MAX[j], v)
                        h["coded_feature_value"][i][j] = v
    else:
        fe, fd = [[0] * 8 for _ in range(8)], [[0] * 8 for _ in range(8)]
    h["feature_enabled"], h["feature_data"] = fe, fd
    h["last_active_seg_id"] = max([i for i in range(8) if any(fe[i])], default=0)
    h["seg_id_pre_skip"] = int(any(fe[i][j] for i in range(8) for j in range(5, 8)))
    h["delta_q_present"] = int(rng.integers(2)) if h["base_q_idx"] > 0 else 0
    h["delta_q_res"] = int(rng.integers(4)) if h["delta_q_present"] else 0
    h["delta_lf_present"] = int(rng.integers(2)) if (h["delta_q_present"] and not h["allow_intrabc"]) else 0
    h["delta_lf_res"] = int(rng.integers(4)) if h["delta_lf_present"] else 0
    h["delta_lf_multi"] = int(rng.integers(2)) if h["delta_lf_present"] else 0
    lossless = True
    for seg in range(8):
        q = h["base_q_idx"]
        if h["segmentation_enabled"] and fe[seg][0]:
            q = max(0, min(255, q + fd[seg][0]))
        if q or any(h[k] for k in ("delta_q_y_dc", "delta_q_u_dc", "delta_q_u_ac", "delta_q_v_dc", "delta_q_v_ac")):
            lossless = False
    h["coded_lossless"] = int(lossless)
    h["all_lossless"] = int(lossless and h["frame_width"] == h["upscaled_width"])
    # loop filter
    ref_d = list(prev["ref_deltas"]) if prev else list(fw.DEFAULT_REF_DELTAS)
    mode_d = list(prev["mode_deltas"]) if prev else [0, 0]
    h.update(loop_filter_level=[0] * 4, loop_filter_sharpness=0, loop_filter_delta_enabled=0, loop_filter_delta_update=0,
             update_ref_delta=[0] * 8, update_mode_delta=[0] * 2)
    if h["coded_lossless"] or h["allow_intrabc"]:
        ref_d, mode_d = list(fw.DEFAULT_REF_DELTAS), [0, 0]
    else:
        h["loop_filter_level"][0], h["loop_filter_level"][1] = int(rng.integers(64)), int(rng.choice([0, int(rng.integers(64))]))
        if planes > 1 and (h["loop_filter_level"][0] or h["loop_filter_level"][1]):
            h["loop_filter_level"][2], h["loop_filter_level"][3] = int(rng.integers(64)), int(rng.integers(64))
        h["loop_filter_sharpness"] = int(rng.integers(8))
        h["loop_filter_delta_enabled"] = int(rng.integers(2))
        if h["loop_filter_delta_enabled"]:
            h["loop_filter_delta_update"] = int(rng.integers(2))
            if h["loop_filter_delta_update"]:
                for i in range(8):
                    h["update_ref_delta"][i] = int(rng.integers(2))
                    if h["update_ref_delta"][i]:
                        ref_d[i] = int(rng.integers(-64, 64))
                for i in range(2):
                    h["update_mode_delta"][i] = int(rng.integers(2))
                    if h["update_mode_delta"][i]:
                        mode_d[i] = int(rng.integers(-64, 64))
    h["loop_filter_ref_deltas"], h["loop_filter_mode_deltas"] = ref_d, mode_d
    # CDEF
    h.update(cdef_damping=3, cdef_bits=0, cdef_y_pri_strength=[0] * 8, cdef_y_sec_strength=[0] * 8, cdef_uv_pri_strength=[0] * 8, cdef_uv_sec_strength=[0] * 8)
    if not (h["coded_lossless"] or h["allow_intrabc"] or not seq["enable_cdef"]):
        h["cdef_damping"], h["cdef_bits"] = int(rng.integers(3, 7)), int(rng.integers(4))
        for i in range(1 << h["cdef_bits"]):
            h["cdef_y_pri_strength"][i] = int(rng.integers(16))
            h["cdef_y_sec_strength"][i] = int(rng.choice([0, 1, 2, 4]))
            if planes > 1:
                h["cdef_uv_pri_strength"][i] = int(rng.integers(16))
                h["cdef_uv_sec_strength"][i] = int(rng.choice([0, 1, 2, 4]))
    # loop restoration
    h.update(frame_restoration_type=[0] * 3, lr_unit_shift=0, lr_uv_shift=0, loop_restoration_size=[0] * 3)
    if not (h["all_lossless"] or h["allow_intrabc"] or not seq["enable_restoration"]):
        for i in range(planes):
            h["frame_restoration_type"][i] = int(rng.integers(4))
        if any(h["frame_restoration_type"]):
            h["lr_unit_shift"] = int(rng.integers(1, 3)) if seq["use_128x128_superblock"] else int(rng.integers(3))
            if seq["subsampling_x"] and seq["subsampling_y"] and any(h["frame_restoration_type"][1:]):
                h["lr_uv_shift"] = int(rng.integers(2))
            h["loop_restoration_size"][0] = 256 >> (2 - h["lr_unit_shift"])
            h["loop_restoration_size"][1] = h["loop_restoration_size"][2] = h["loop_restoration_size"][0] >> h["lr_uv_shift"]
    h["tx_mode_select"] = 0 if h["coded_lossless"] else int(rng.integers(2))
    h["reference_select"] = 0 if intra else int(rng.integers(2))
    allowed, frames = skip_mode(seq, h, slots) if not intra else (False, [0, 0])
    h["skip_mode_allowed"] = allowed
    h["skip_mode_present"] = int(rng.integers(2)) if allowed else 0
    h["skip_mode_frame"] = frames
    h["allow_warped_motion"] = 0 if (intra or h["error_resilient_mode"] or not seq["enable_warped_motion"]) else int(rng.integers(2))
    h["reduced_tx_set"] = int(rng.integers(2))
    # global motion: values on the grid of their precision, within range of the predicted value's coder
    h["gm_type"] = [0] * 7
    h["gm_params"] = fw.default_gm()
    if not intra:
        for ref in range(7):
            t = int(rng.choice([0, 0, 1, 2, 3]))
            h["gm_type"][ref] = t
            g = h["gm_params"][ref]

            def val(idx):
                abs_bits, prec_bits = 12, 15
                if idx < 2:
                    if t == 1:
                        abs_bits, prec_bits = 9 - (not h["allow_high_precision_mv"]), 3 - (not h["allow_high_precision_mv"])
                    else:
                        abs_bits, prec_bits = 12, 6
                mx = 1 << abs_bits
                v = int(rng.integers(-mx, mx + 1)) if rng.integers(2) else int(rng.integers(-8, 9))
                return (v << (16 - prec_bits)) + ((1 << 16) if idx % 3 == 2 else 0)
            if t >= 2:
                g[2], g[3] = val(2), val(3)
                if t == 3:
                    g[4], g[5] = val(4), val(5)
                else:
                    g[4], g[5] = -g[3], g[2]
            if t >= 1:
                g[0], g[1] = val(0), val(1)
    h["film_grain"] = random_grain(rng, seq, h, slots)
    return h


COMPARE_SKIP = {"header_bits", "display_frame_id", "frame_to_show_map_idx", "show_existing_frame"}


def check(obu, parser, seq, hdr, slots, payload, nbits):
    rc, got_struct = parser.frame_header(payload + b"\xAA\x55")   # trailing bytes: tile data in an OBU_FRAME
    assert rc == 0, hdr
    got = got_struct.as_dict()
    assert got["header_bits"] == nbits
    for k, v in got.items():
        if k in COMPARE_SKIP or k not in hdr:
            continue
        want = hdr[k]
        if k in ("mi_col_starts", "mi_row_starts"):
            v = v[:len(want)]
        if k == "buffer_removal_time" or k == "ref_order_hint":
            want = list(want) + [0] * (len(v) - len(want))
        assert v == want, (k, v, want)
    return got_struct


def test_random_frame_sequences(obu):
    rng = np.random.default_rng(2024)
    kinds = dict(key=0, inter=0, intra_only=0, switch=0, existing=0, superres=0, found_ref=0, nonuniform=0, lossless=0, gm=0, grain_ref=0, seg_prev=0)
    for stream in range(60):
        seq = make_seq(rng)
        parser = obu.Parser()
        parser.set_sequence_header(obu.sequence_header_struct(seq))
        slots = fw.Slots()
        for frame_no in range(int(rng.integers(3, 12))):
            have_refs = all(s is not None for s in slots.s)
            if not have_refs:
                ft = KEY
            else:
                ft = int(rng.choice([KEY, INTER, INTER, INTER, INTER, INTRA_ONLY, SWITCH]))
            if have_refs and rng.integers(8) == 0:   # show an existing frame
                idx = int(rng.integers(8))
                src = slots.s[idx]
                hdr = dict(show_existing_frame=1, frame_to_show_map_idx=idx, display_frame_id=src["frame_id"], frame_presentation_time=0)
                if seq["decoder_model_info_present"] and not seq["equal_picture_interval"]:
                    hdr["frame_presentation_time"] = int(rng.integers(1 << min(seq["frame_presentation_time_length"], 31)))
                payload, nbits = fw.write_frame_header(seq, dict(hdr, frame_type=src["frame_type"]), slots)
                rc, got = parser.frame_header(payload)
                assert rc == 0
                d = got.as_dict()
                assert d["show_existing_frame"] == 1 and d["frame_type"] == src["frame_type"] and d["header_bits"] == nbits
                assert d["refresh_frame_flags"] == (0xFF if src["frame_type"] == KEY else 0)
                assert (d["upscaled_width"], d["frame_height"], d["gm_params"]) == (src["upscaled_width"], src["frame_height"], src["gm"])
                parser.end_frame(got)
                if src["frame_type"] == KEY:
                    slots.s = [src] * 8
                kinds["existing"] += 1
                continue
            hdr = random_header(rng, seq, slots, ft, frame_no)
            payload, nbits = fw.write_frame_header(seq, hdr, slots)
            got = check(obu, parser, seq, hdr, slots, payload, nbits)
            parser.end_frame(got)
            slots.refresh(hdr)
            kinds[{KEY: "key", INTER: "inter", INTRA_ONLY: "intra_only", SWITCH: "switch"}[ft]] += 1
            kinds["superres"] += hdr["use_superres"]
            kinds["found_ref"] += hdr["found_ref"] >= 0
            kinds["nonuniform"] += not hdr["uniform_tile_spacing_flag"]
            kinds["lossless"] += hdr["coded_lossless"]
            kinds["gm"] += any(hdr["gm_type"])
            kinds["grain_ref"] += hdr["film_grain"]["apply_grain"] and not hdr["film_grain"]["update_grain"]
            kinds["seg_prev"] += hdr["segmentation_enabled"] and not hdr["segmentation_update_data"]
        parser.close()
    assert all(v > 0 for v in kinds.values()), kinds


def simple_seq(**kw):
    s = dict(profile=0, still_picture=0, reduced_still_picture_header=0, num_operating_points=1, frame_width_bits=12, frame_height_bits=12,
             max_width=640, max_height=360, use_128x128_superblock=0, enable_filter_intra=1, enable_intra_edge_filter=1, enable_interintra_compound=1,
             enable_masked_compound=1, enable_warped_motion=1, enable_dual_filter=1, enable_order_hint=1, enable_jnt_comp=1, enable_ref_frame_mvs=1,
             seq_force_screen_content_tools=0, seq_force_integer_mv=2, order_hint_bits=7, enable_superres=0, enable_cdef=1, enable_restoration=1,
             bit_depth=8, hbd=0, monochrome=0, color_description_present=0, color_primaries=2, transfer_characteristics=2, matrix_coefficients=2,
             color_range=0, subsampling_x=1, subsampling_y=1, layout=1, chroma_sample_position=0, separate_uv_delta_q=0, film_grain_params_present=1)
    for k in ("operating_point_idc", "seq_level_idx", "seq_tier", "decoder_model_present_for_op", "low_delay_mode", "decoder_buffer_delay",
              "encoder_buffer_delay", "initial_display_delay"):
        s[k] = [0] * 32
    for k in ("frame_id_numbers_present", "decoder_model_info_present", "equal_picture_interval", "additional_frame_id_length", "delta_frame_id_length",
              "frame_presentation_time_length", "buffer_removal_time_length", "timing_info_present"):
        s[k] = 0
    s.update(kw)
    return s


def plain_header(rng, seq, slots, ft, frame_no, **fixed):
    """a random header without the size override (its expectations depend on the references' sizes)"""
    fixed.setdefault("frame_size_override_flag", 0)
    return random_header(rng, seq, slots, ft, frame_no, fixed)


def test_short_signaling_of_references(obu):
    """frame_refs_short_signaling (AV1 spec 7.8): with slots holding order hints 10, 12, .., 24 (slot i: 10 + 2 i)
    and the current frame at 19, LAST = slot 4 (18) and GOLDEN = slot 0 (10): backward references are
    24 (latest -> ALTREF), 20 (earliest -> BWDREF), 22 (-> ALTREF2); the remaining forward ones in
    descending order 16, 14 -> LAST2, LAST3"""
    rng = np.random.default_rng(5)
    seq = simple_seq()
    parser = obu.Parser()
    parser.set_sequence_header(obu.sequence_header_struct(seq))
    slots = fw.Slots()

    def push(h):
        payload, nbits = fw.write_frame_header(seq, h, slots)
        got = check(obu, parser, seq, h, slots, payload, nbits)
        parser.end_frame(got)
        slots.refresh(h)
        return got.as_dict()
    push(plain_header(rng, seq, slots, KEY, 0, show_frame=1, order_hint=0))
    for i in range(8):   # one inter frame per slot, order hint 10 + 2 i
        push(plain_header(rng, seq, slots, INTER, 1 + i, order_hint=10 + 2 * i, refresh_frame_flags=1 << i, error_resilient_mode=0))
    assert [s0["order_hint"] for s0 in slots.s] == [10 + 2 * i for i in range(8)]
    got = push(plain_header(rng, seq, slots, INTER, 20, order_hint=19, refresh_frame_flags=0, error_resilient_mode=0, frame_refs_short_signaling=1,
                            last_frame_idx=4, gold_frame_idx=0, ref_frame_idx=[4, 3, 2, 0, 5, 6, 7]))
    # LAST 18, LAST2 16, LAST3 14, GOLDEN 10, BWDREF 20, ALTREF2 22, ALTREF 24
    assert got["ref_frame_idx"] == [4, 3, 2, 0, 5, 6, 7] and got["frame_refs_short_signaling"] == 1
    if got["reference_select"]:
        assert got["skip_mode_frame"] == [1, 5]     # nearest past (LAST, 18) and nearest future (BWDREF, 20)
    parser.close()


def test_frame_header_errors_and_truncation(obu):
    rng = np.random.default_rng(9)
    seq = simple_seq()
    parser = obu.Parser()
    rc, _ = parser.frame_header(b"\x00\x00")
    assert rc == -6                                   # EA1_ESTATE: no sequence header yet
    parser.set_sequence_header(obu.sequence_header_struct(seq))
    slots = fw.Slots()
    # an inter frame before any key frame: its references are empty
    inter = bytes([0b00110000, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00])
    rc, _ = parser.frame_header(inter)
    assert rc == -1
    key = plain_header(rng, seq, slots, KEY, 0, show_frame=1)
    payload, nbits = fw.write_frame_header(seq, key, slots)
    for n in range(len(payload)):                     # every truncation is refused, never read past the end
        rc, _ = parser.frame_header(payload[:n])
        assert rc == -1 or n * 8 >= nbits
    rc, got = parser.frame_header(payload)
    assert rc == 0
    parser.end_frame(got)
    for _ in range(500):                              # garbage never crashes the parser
        junk = bytes(rng.integers(0, 256, int(rng.integers(1, 120)), dtype=np.uint8))
        assert parser.frame_header(junk)[0] in (0, -1)
    parser.close()


def test_hand_over_to_the_gpu_path(obu):
    """frame header -> ea1_frame_desc fields and ea1_film_grain (what the reconstruction and the film
    grain pass take)"""
    import ctypes
    import easyav1_amd
    from easyav1_amd import grain, ir
    L = easyav1_amd.load()
    rng = np.random.default_rng(31)
    seq = simple_seq()
    seq_struct = obu.sequence_header_struct(seq)
    parser = obu.Parser()
    parser.set_sequence_header(seq_struct)
    slots = fw.Slots()
    done = 0
    for trial in range(40):
        h = plain_header(rng, seq, slots, KEY, 0, show_frame=1)
        payload, nbits = fw.write_frame_header(seq, h, slots)
        rc, got = parser.frame_header(payload)
        assert rc == 0
        desc = ir.FrameDesc()
        L.ea1_av1_frame_header_to_desc.argtypes = [ctypes.c_void_p] * 3
        assert L.ea1_av1_frame_header_to_desc(ctypes.byref(got), ctypes.byref(seq_struct), ctypes.byref(desc)) == 0
        assert (desc.width, desc.height, desc.is_inter) == (h["mi_cols"] * 4, h["mi_rows"] * 4, 0)
        assert list(desc.lf_level) == h["loop_filter_level"] and desc.lf_sharpness == h["loop_filter_sharpness"]
        assert desc.cdef_enable == int(not (h["coded_lossless"] or h["allow_intrabc"])) and desc.cdef_damping == h["cdef_damping"]
        for i in range(1 << h["cdef_bits"]):
            assert desc.cdef_y_strength[i] == (h["cdef_y_pri_strength"][i] << 2) | min(h["cdef_y_sec_strength"][i], 3)
        assert [int(v) for v in desc.lr_type] == [int(t != 0) for t in h["frame_restoration_type"]]
        if h["frame_restoration_type"][0]:
            assert 1 << desc.lr_unit_log2[0] == h["loop_restoration_size"][0]
        fg = grain.FilmGrain()
        L.ea1_av1_frame_header_film_grain.argtypes = [ctypes.c_void_p] * 3
        rc = L.ea1_av1_frame_header_film_grain(ctypes.byref(got), ctypes.byref(seq_struct), ctypes.byref(fg))
        g = h["film_grain"]
        assert rc == (0 if g["apply_grain"] else -7)
        if rc == 0:
            assert fg.grain_seed == g["grain_seed"] and fg.num_y_points == g["num_y_points"] and fg.scaling_shift == g["grain_scaling_minus_8"] + 8
            assert [fg.point_y[i][0] for i in range(g["num_y_points"])] == g["point_y_value"][:g["num_y_points"]]
            npl = 2 * g["ar_coeff_lag"] * (g["ar_coeff_lag"] + 1)
            if g["num_y_points"]:
                assert [fg.ar_coeffs_y[i] for i in range(npl)] == [v - 128 for v in g["ar_coeffs_y_plus_128"][:npl]]
            assert fg.ar_coeff_shift == g["ar_coeff_shift_minus_6"] + 6 and fg.overlap_flag == g["overlap_flag"]
            if g["num_cb_points"]:
                assert (fg.uv_mult[0], fg.uv_luma_mult[0], fg.uv_offset[0]) == (g["cb_mult"], g["cb_luma_mult"], g["cb_offset"])
            # the converted parameters are legal input of the film grain oracle
            from tests import oracle_lib as ol
            ol.oracle_set_gaussian_sequence(grain.synth_gaussian_sequence())
            pic = ir.synth_picture(64, 32, 3)
            out = ol.oracle_film_grain(fg, pic)
            assert out[0].shape == (32, 64)
            done += 1
    assert done >= 10
    # what the path does not decode is refused, not mis-decoded
    seq10 = simple_seq(bit_depth=10, hbd=1)
    desc = ir.FrameDesc()
    assert L.ea1_av1_frame_header_to_desc(ctypes.byref(got), ctypes.byref(obu.sequence_header_struct(seq10)), ctypes.byref(desc)) == -1
    parser.close()


def test_tile_groups(obu):
    """tile_group_obu: whole frame in one group / split over several groups, every TileSizeBytes, as the
    tail of an OBU_FRAME; truncations are refused"""
    rng = np.random.default_rng(77)
    seq = simple_seq(max_width=1920, max_height=1080)
    parser = obu.Parser()
    parser.set_sequence_header(obu.sequence_header_struct(seq))
    slots = fw.Slots()
    multi = 0
    for trial in range(60):
        h = plain_header(rng, seq, slots, KEY, 0, show_frame=1)
        hdr_bytes, nbits = fw.write_frame_header(seq, h, slots)
        rc, got = parser.frame_header(hdr_bytes)
        assert rc == 0
        ntiles = h["tile_cols"] * h["tile_rows"]
        bodies = [bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)) for _ in range(ntiles)]
        # split the tiles into one or more groups
        cuts = sorted(set([0, ntiles] + [int(c) for c in rng.integers(0, ntiles + 1, int(rng.integers(0, 3)))]))
        groups = [(a, b - 1) for a, b in zip(cuts[:-1], cuts[1:])]
        for gi, (t0, t1) in enumerate(groups):
            w = fw.Bits()
            present = ntiles > 1 and (len(groups) > 1 or rng.integers(2))
            if ntiles > 1:
                w.f(1, int(present))
            if present:
                bits = h["tile_cols_log2"] + h["tile_rows_log2"]
                w.f(bits, t0)
                w.f(bits, t1)
            while len(w.bits) % 8:
                w.bits.append(0)
            data = w.bytes()
            want = []
            for t in range(t0, t1 + 1):
                if t != t1:
                    data += (len(bodies[t]) - 1).to_bytes(h["tile_size_bytes"], "little")
                want.append((t, len(data), len(bodies[t])))
                data += bodies[t]
            if len(groups) == 1 and not present and gi == 0 and trial % 2:
                payload = hdr_bytes + data     # OBU_FRAME: header, byte alignment, tile group
                rc, tiles = obu.parse_tile_group(got, payload[(nbits + 7) // 8:])
            else:
                rc, tiles = obu.parse_tile_group(got, data)
            assert rc == 0
            assert [(t["tile_num"], t["offset"], t["size"]) for t in tiles] == want
            for t in tiles:
                assert (t["tile_row"], t["tile_col"]) == divmod(t["tile_num"], h["tile_cols"])
                assert t["mi_col_start"] == h["mi_col_starts"][t["tile_col"]] and t["mi_col_end"] == h["mi_col_starts"][t["tile_col"] + 1]
                assert t["mi_row_start"] == h["mi_row_starts"][t["tile_row"]] and t["mi_row_end"] == h["mi_row_starts"][t["tile_row"] + 1]
            if t1 > t0:
                multi += 1
                for n in range(0, len(data) - len(bodies[t1]), 5):   # cut inside the sized tiles: refused
                    rc, _ = obu.parse_tile_group(got, data[:n])
                    assert rc == -1, n
    assert multi > 10
    parser.close()


def test_temporal_units_to_frames(obu):
    """the whole OBU layer: temporal units (delimiter, sequence header, metadata / padding, frame header
    + tile groups or OBU_FRAME, repeated headers, shown-existing frames) -> one call per frame"""
    rng = np.random.default_rng(404)
    seq = simple_seq(max_width=1280, max_height=720)
    seq_obu = wr.obu(1, wr.write_sequence_header(seq))
    parser = obu.Parser()
    slots = fw.Slots()
    stream = b""
    expect = []   # per frame: (header dict, [(tile_num, bytes)])
    for frame_no in range(14):
        have_refs = all(s0 is not None for s0 in slots.s)
        ft = KEY if not have_refs else int(rng.choice([KEY, INTER, INTER, INTER, INTRA_ONLY]))
        h = plain_header(rng, seq, slots, ft, frame_no)
        hb, nbits = fw.write_frame_header(seq, h, slots)
        ntiles = h["tile_cols"] * h["tile_rows"]
        bodies = [bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8)) for _ in range(ntiles)]
        tu = wr.obu(2, b"")
        if frame_no == 0 or rng.integers(3) == 0:
            tu += seq_obu
        if rng.integers(3) == 0:
            tu += wr.obu(5, bytes(rng.integers(0, 256, 12, dtype=np.uint8)))      # metadata: passed over
        cuts = sorted(set([0, ntiles] + [int(c) for c in rng.integers(0, ntiles + 1, int(rng.integers(0, 3)))]))
        groups = [(a, b - 1) for a, b in zip(cuts[:-1], cuts[1:])]

        def tile_group(t0, t1, force_flag):
            w = fw.Bits()
            present = ntiles > 1 and (force_flag or rng.integers(2))
            if ntiles > 1:
                w.f(1, int(present))
            if present:
                bits = h["tile_cols_log2"] + h["tile_rows_log2"]
                w.f(bits, t0)
                w.f(bits, t1)
            while len(w.bits) % 8:
                w.bits.append(0)
            data = w.bytes()
            for t in range(t0, t1 + 1):
                if t != t1:
                    data += (len(bodies[t]) - 1).to_bytes(h["tile_size_bytes"], "little")
                data += bodies[t]
            return data
        if len(groups) == 1 and rng.integers(2):
            tu += wr.obu(6, hb + tile_group(0, ntiles - 1, False))                  # OBU_FRAME
        else:
            trail = fw.Bits()
            trail.bits = [int(c) for byte in hb for c in f"{byte:08b}"][:nbits]
            trail.trailing()
            header_obu = trail.bytes()
            tu += wr.obu(3, header_obu)
            for gi, (t0, t1) in enumerate(groups):
                if gi and rng.integers(2):
                    tu += wr.obu(7, header_obu)                                   # redundant copy between tile groups
                tu += wr.obu(4, tile_group(t0, t1, len(groups) > 1))
        if rng.integers(4) == 0:
            tu += wr.obu(15, b"\\0" * 5)                                            # padding
        stream += tu
        expect.append((h, bodies))
        slots.refresh(h)
    # walk the stream frame by frame
    pos, frames = 0, 0
    while frames < len(expect):
        rc, used, got, tiles = obu.next_frame(parser, stream[pos:])
        assert rc == 0, (frames, rc)
        h, bodies = expect[frames]
        d = got.as_dict()
        for k in ("frame_type", "show_frame", "order_hint", "refresh_frame_flags", "frame_width", "frame_height", "tile_cols", "tile_rows", "base_q_idx"):
            assert d[k] == h[k], (frames, k)
        assert [t["tile_num"] for t in tiles] == list(range(len(bodies)))
        for t in tiles:
            assert stream[pos + t["offset"]: pos + t["offset"] + t["size"]] == bodies[t["tile_num"]]
        parser.end_frame(got)
        pos += used
        frames += 1
    rc, used, _, _ = obu.next_frame(parser, stream[pos:])
    assert rc == -7 and pos + used == len(stream)      # only padding / nothing left: EA1_ENOENT
    # a stream cut in the middle of a frame: ENOENT (incomplete) or EINVAL (cut inside an OBU), never a frame
    p2 = obu.Parser()
    first_tu_end = stream.index(wr.obu(2, b""), 2)
    for n in range(3, first_tu_end - 1, 3):
        rc, used, _, tiles = obu.next_frame(p2, stream[:n])
        assert rc in (-7, -1) and tiles == []
    p2.close()
    parser.close()
