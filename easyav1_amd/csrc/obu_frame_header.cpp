// AV1 frame header: uncompressed_header() and the state it depends on (AV1 specification 5.9,
// 7.8 set frame refs, 7.20 reference frame update, 7.21 reference frame loading).  Second
// table-free slice of the bitstream front end, see include/easyav1_amd_obu.h.  Host C++ by design.
#include <algorithm>
#include <cstring>
#include <new>

#include "ea1_common.h"
#include "easyav1_amd_obu.h"
#include "obu_bits.h"

namespace {

using ea1::BitReader;

constexpr int kRefs = 8, kRefsPerFrame = 7;
constexpr int kSegFeatureBits[8] = {8, 6, 6, 6, 6, 3, 0, 0};
constexpr int kSegFeatureSigned[8] = {1, 1, 1, 1, 1, 0, 0, 0};
constexpr int kSegFeatureMax[8] = {255, 63, 63, 63, 63, 7, 0, 0};
constexpr int8_t kDefaultRefDeltas[8] = {1, 0, 0, 0, -1, 0, -1, -1};

struct Slot {   // what the reference frame update process saves (7.20)
    bool valid = false;
    uint32_t frame_id = 0, upscaled_width = 0, frame_width = 0, frame_height = 0, render_width = 0, render_height = 0;
    uint8_t frame_type = 0, order_hint = 0, showable_frame = 0;
    int32_t gm_params[7][6];
    int8_t lf_ref_deltas[8], lf_mode_deltas[2];
    uint8_t feature_enabled[8][8];
    int16_t feature_data[8][8];
    ea1_av1_film_grain_syntax grain;
};

void default_gm(int32_t gm[7][6])
{
    for (int r = 0; r < 7; r++)
        for (int i = 0; i < 6; i++) gm[r][i] = (i % 3 == 2) ? 1 << 16 : 0;
}

}  // namespace

struct ea1_av1_parser {
    bool have_seq = false;
    ea1_av1_sequence_header seq;
    Slot slot[kRefs];
    uint32_t current_frame_id = 0;   // of the previous frame header (PrevFrameID)
    // state of the header being parsed that end_frame saves into the refreshed slots
    int8_t cur_lf_ref_deltas[8], cur_lf_mode_deltas[2];
};

namespace {

int tile_log2(int blk, int target)
{
    int k = 0;
    while ((blk << k) < target) k++;
    return k;
}

struct Parse {
    ea1_av1_parser &ps;
    const ea1_av1_sequence_header &seq;
    BitReader b;
    ea1_av1_frame_header &h;
    int temporal_id, spatial_id;
    bool frame_is_intra = false;
    int num_planes;
    uint8_t order_hints[8] = {0};   // OrderHints[refFrame], index 1..7
    int32_t prev_gm[7][6];

    int rel_dist(int a, int c) const
    {
        if (!seq.enable_order_hint) return 0;
        int diff = a - c;
        const int m = 1 << (seq.order_hint_bits - 1);
        return (diff & (m - 1)) - (diff & m);
    }

    void superres_and_size()
    {
        h.use_superres = seq.enable_superres ? uint8_t(b.f(1)) : 0;
        h.superres_denom = h.use_superres ? uint8_t(b.f(3) + 9) : 8;
        h.upscaled_width = h.frame_width;
        h.frame_width = (h.upscaled_width * 8 + h.superres_denom / 2) / h.superres_denom;
        h.mi_cols = 2 * ((h.frame_width + 7) >> 3);
        h.mi_rows = 2 * ((h.frame_height + 7) >> 3);
    }
    void frame_size()
    {
        if (h.frame_size_override_flag) {
            h.frame_width = b.f(seq.frame_width_bits) + 1;
            h.frame_height = b.f(seq.frame_height_bits) + 1;
        } else {
            h.frame_width = seq.max_width;
            h.frame_height = seq.max_height;
        }
        superres_and_size();
    }
    void render_size()
    {
        if (b.f(1)) {
            h.render_width = b.f(16) + 1;
            h.render_height = b.f(16) + 1;
        } else {
            h.render_width = h.upscaled_width;
            h.render_height = h.frame_height;
        }
    }
    void frame_size_with_refs()
    {
        for (int i = 0; i < kRefsPerFrame; i++) {
            if (b.f(1)) {   // found_ref
                const Slot &s = ps.slot[h.ref_frame_idx[i]];
                h.upscaled_width = s.upscaled_width;
                h.frame_width = h.upscaled_width;
                h.frame_height = s.frame_height;
                h.render_width = s.render_width;
                h.render_height = s.render_height;
                superres_and_size();
                return;
            }
        }
        frame_size();
        render_size();
    }

    // 7.8
    int set_frame_refs(int last_idx, int gold_idx)
    {
        int8_t *idx = h.ref_frame_idx;
        for (int i = 0; i < kRefsPerFrame; i++) idx[i] = -1;
        idx[0] = int8_t(last_idx);
        idx[3] = int8_t(gold_idx);   // GOLDEN_FRAME - LAST_FRAME
        bool used[kRefs] = {false};
        used[last_idx] = used[gold_idx] = true;
        const int cur = 1 << (seq.order_hint_bits - 1);
        int shifted[kRefs];
        for (int i = 0; i < kRefs; i++) shifted[i] = cur + rel_dist(ps.slot[i].order_hint, h.order_hint);
        if (shifted[last_idx] >= cur || shifted[gold_idx] >= cur) return ea1::fail(EA1_EINVAL, "frame header: short signaling with LAST / GOLDEN not in the past");
        auto find = [&](bool backward, bool latest) {
            int ref = -1, best = 0;
            for (int i = 0; i < kRefs; i++) {
                const int hint = shifted[i];
                if (used[i] || (backward ? hint < cur : hint >= cur)) continue;
                if (ref < 0 || (latest ? hint >= best : hint < best)) {
                    ref = i;
                    best = hint;
                }
            }
            return ref;
        };
        int ref = find(true, true);   // ALTREF: latest backward
        if (ref >= 0) {
            idx[6] = int8_t(ref);
            used[ref] = true;
        }
        ref = find(true, false);      // BWDREF: earliest backward
        if (ref >= 0) {
            idx[4] = int8_t(ref);
            used[ref] = true;
        }
        ref = find(true, false);      // ALTREF2
        if (ref >= 0) {
            idx[5] = int8_t(ref);
            used[ref] = true;
        }
        static const int list[5] = {1, 2, 4, 5, 6};   // LAST2, LAST3, BWDREF, ALTREF2, ALTREF (minus LAST_FRAME)
        for (int k = 0; k < 5; k++) {
            if (idx[list[k]] >= 0) continue;
            ref = find(false, true);  // latest forward
            if (ref >= 0) {
                idx[list[k]] = int8_t(ref);
                used[ref] = true;
            }
        }
        ref = -1;
        int earliest = 0;
        for (int i = 0; i < kRefs; i++)
            if (ref < 0 || shifted[i] < earliest) {
                ref = i;
                earliest = shifted[i];
            }
        for (int i = 0; i < kRefsPerFrame; i++)
            if (idx[i] < 0) idx[i] = int8_t(ref);
        return EA1_OK;
    }

    int tile_info()
    {
        const int sb_shift = seq.use_128x128_superblock ? 5 : 4, sb_size = sb_shift + 2;
        const int sb_cols = int((h.mi_cols + (1u << sb_shift) - 1) >> sb_shift), sb_rows = int((h.mi_rows + (1u << sb_shift) - 1) >> sb_shift);
        const int max_tile_width_sb = 4096 >> sb_size;
        int max_tile_area_sb = (4096 * 2304) >> (2 * sb_size);
        const int min_log2_cols = tile_log2(max_tile_width_sb, sb_cols);
        const int max_log2_cols = tile_log2(1, std::min(sb_cols, EA1_MAX_TILE_COLS)), max_log2_rows = tile_log2(1, std::min(sb_rows, EA1_MAX_TILE_ROWS));
        const int min_log2_tiles = std::max(min_log2_cols, tile_log2(max_tile_area_sb, sb_rows * sb_cols));
        h.uniform_tile_spacing_flag = uint8_t(b.f(1));
        int cols = 0, rows = 0;
        if (h.uniform_tile_spacing_flag) {
            int l2 = min_log2_cols;
            while (l2 < max_log2_cols && b.f(1)) l2++;
            h.tile_cols_log2 = uint8_t(l2);
            const int tw = (sb_cols + (1 << l2) - 1) >> l2;
            for (int start = 0; start < sb_cols; start += tw) h.mi_col_starts[cols++] = uint16_t(start << sb_shift);
            l2 = std::max(min_log2_tiles - int(h.tile_cols_log2), 0);
            while (l2 < max_log2_rows && b.f(1)) l2++;
            h.tile_rows_log2 = uint8_t(l2);
            const int th = (sb_rows + (1 << l2) - 1) >> l2;
            for (int start = 0; start < sb_rows; start += th) h.mi_row_starts[rows++] = uint16_t(start << sb_shift);
        } else {
            int widest = 0;
            for (int start = 0; start < sb_cols;) {
                if (cols >= EA1_MAX_TILE_COLS) return ea1::fail(EA1_EINVAL, "frame header: more than 64 tile columns");
                h.mi_col_starts[cols++] = uint16_t(start << sb_shift);
                const int size = int(b.ns(uint32_t(std::min(sb_cols - start, max_tile_width_sb)))) + 1;
                widest = std::max(widest, size);
                start += size;
            }
            h.tile_cols_log2 = uint8_t(tile_log2(1, cols));
            max_tile_area_sb = min_log2_tiles > 0 ? (sb_rows * sb_cols) >> (min_log2_tiles + 1) : sb_rows * sb_cols;
            const int max_tile_height_sb = std::max(max_tile_area_sb / widest, 1);
            for (int start = 0; start < sb_rows;) {
                if (rows >= EA1_MAX_TILE_ROWS) return ea1::fail(EA1_EINVAL, "frame header: more than 64 tile rows");
                h.mi_row_starts[rows++] = uint16_t(start << sb_shift);
                start += int(b.ns(uint32_t(std::min(sb_rows - start, max_tile_height_sb)))) + 1;
            }
            h.tile_rows_log2 = uint8_t(tile_log2(1, rows));
        }
        h.mi_col_starts[cols] = uint16_t(h.mi_cols);
        h.mi_row_starts[rows] = uint16_t(h.mi_rows);
        h.tile_cols = uint8_t(cols);
        h.tile_rows = uint8_t(rows);
        h.tile_size_bytes = 4;
        if (h.tile_cols_log2 || h.tile_rows_log2) {
            h.context_update_tile_id = uint16_t(b.f(h.tile_cols_log2 + h.tile_rows_log2));
            h.tile_size_bytes = uint8_t(b.f(2) + 1);
            if (h.context_update_tile_id >= cols * rows) return ea1::fail(EA1_EINVAL, "frame header: context_update_tile_id %u of %d tiles", h.context_update_tile_id, cols * rows);
        }
        return EA1_OK;
    }

    int read_delta_q() { return b.f(1) ? b.su(7) : 0; }

    void quantization_params()
    {
        h.base_q_idx = uint8_t(b.f(8));
        h.delta_q_y_dc = int8_t(read_delta_q());
        if (num_planes > 1) {
            h.diff_uv_delta = seq.separate_uv_delta_q ? uint8_t(b.f(1)) : 0;
            h.delta_q_u_dc = int8_t(read_delta_q());
            h.delta_q_u_ac = int8_t(read_delta_q());
            if (h.diff_uv_delta) {
                h.delta_q_v_dc = int8_t(read_delta_q());
                h.delta_q_v_ac = int8_t(read_delta_q());
            } else {
                h.delta_q_v_dc = h.delta_q_u_dc;
                h.delta_q_v_ac = h.delta_q_u_ac;
            }
        }
        h.using_qmatrix = uint8_t(b.f(1));
        if (h.using_qmatrix) {
            h.qm_y = uint8_t(b.f(4));
            h.qm_u = uint8_t(b.f(4));
            h.qm_v = seq.separate_uv_delta_q ? uint8_t(b.f(4)) : h.qm_u;
        }
    }

    void segmentation_params()
    {
        h.segmentation_enabled = uint8_t(b.f(1));
        if (h.segmentation_enabled) {
            if (h.primary_ref_frame == EA1_PRIMARY_REF_NONE) {
                h.segmentation_update_map = 1;
                h.segmentation_update_data = 1;
            } else {
                h.segmentation_update_map = uint8_t(b.f(1));
                if (h.segmentation_update_map) h.segmentation_temporal_update = uint8_t(b.f(1));
                h.segmentation_update_data = uint8_t(b.f(1));
            }
            if (h.segmentation_update_data) {
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j < 8; j++) {
                        h.feature_enabled[i][j] = uint8_t(b.f(1));
                        int v = 0;
                        if (h.feature_enabled[i][j]) {
                            const int bits = kSegFeatureBits[j], lim = kSegFeatureMax[j];
                            if (kSegFeatureSigned[j]) v = std::min(std::max(b.su(1 + bits), -lim), lim);
                            else v = std::min(std::max(int(bits ? b.f(bits) : 0), 0), lim);
                        }
                        h.feature_data[i][j] = int16_t(v);
                    }
            }
            // else: the parameters loaded from the primary reference frame stay (load_previous)
        } else {
            memset(h.feature_enabled, 0, sizeof h.feature_enabled);
            memset(h.feature_data, 0, sizeof h.feature_data);
        }
        h.seg_id_pre_skip = h.last_active_seg_id = 0;
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++)
                if (h.feature_enabled[i][j]) {
                    h.last_active_seg_id = uint8_t(i);
                    if (j >= 5) h.seg_id_pre_skip = 1;
                }
    }

    void loop_filter_params()
    {
        if (h.coded_lossless || h.allow_intrabc) {
            memcpy(h.loop_filter_ref_deltas, kDefaultRefDeltas, 8);
            h.loop_filter_mode_deltas[0] = h.loop_filter_mode_deltas[1] = 0;
            return;
        }
        h.loop_filter_level[0] = uint8_t(b.f(6));
        h.loop_filter_level[1] = uint8_t(b.f(6));
        if (num_planes > 1 && (h.loop_filter_level[0] || h.loop_filter_level[1])) {
            h.loop_filter_level[2] = uint8_t(b.f(6));
            h.loop_filter_level[3] = uint8_t(b.f(6));
        }
        h.loop_filter_sharpness = uint8_t(b.f(3));
        h.loop_filter_delta_enabled = uint8_t(b.f(1));
        if (h.loop_filter_delta_enabled) {
            h.loop_filter_delta_update = uint8_t(b.f(1));
            if (h.loop_filter_delta_update) {
                for (int i = 0; i < 8; i++)
                    if (b.f(1)) h.loop_filter_ref_deltas[i] = int8_t(b.su(7));
                for (int i = 0; i < 2; i++)
                    if (b.f(1)) h.loop_filter_mode_deltas[i] = int8_t(b.su(7));
            }
        }
    }

    void cdef_params()
    {
        h.cdef_damping = 3;
        if (h.coded_lossless || h.allow_intrabc || !seq.enable_cdef) return;
        h.cdef_damping = uint8_t(b.f(2) + 3);
        h.cdef_bits = uint8_t(b.f(2));
        for (int i = 0; i < (1 << h.cdef_bits); i++) {
            h.cdef_y_pri_strength[i] = uint8_t(b.f(4));
            h.cdef_y_sec_strength[i] = uint8_t(b.f(2));
            if (h.cdef_y_sec_strength[i] == 3) h.cdef_y_sec_strength[i] = 4;
            if (num_planes > 1) {
                h.cdef_uv_pri_strength[i] = uint8_t(b.f(4));
                h.cdef_uv_sec_strength[i] = uint8_t(b.f(2));
                if (h.cdef_uv_sec_strength[i] == 3) h.cdef_uv_sec_strength[i] = 4;
            }
        }
    }

    void lr_params()
    {
        if (h.all_lossless || h.allow_intrabc || !seq.enable_restoration) return;
        static const uint8_t remap[4] = {0, 3, 1, 2};   // lr_type -> NONE, SWITCHABLE, WIENER, SGRPROJ (EA1_LR_* numbering)
        bool uses_lr = false, uses_chroma_lr = false;
        for (int i = 0; i < num_planes; i++) {
            h.frame_restoration_type[i] = remap[b.f(2)];
            if (h.frame_restoration_type[i]) {
                uses_lr = true;
                if (i) uses_chroma_lr = true;
            }
        }
        if (!uses_lr) return;
        if (seq.use_128x128_superblock) {
            h.lr_unit_shift = uint8_t(b.f(1) + 1);
        } else {
            h.lr_unit_shift = uint8_t(b.f(1));
            if (h.lr_unit_shift) h.lr_unit_shift = uint8_t(h.lr_unit_shift + b.f(1));
        }
        if (seq.subsampling_x && seq.subsampling_y && uses_chroma_lr) h.lr_uv_shift = uint8_t(b.f(1));
        h.loop_restoration_size[0] = uint16_t(256 >> (2 - h.lr_unit_shift));
        h.loop_restoration_size[1] = h.loop_restoration_size[2] = uint16_t(h.loop_restoration_size[0] >> h.lr_uv_shift);
    }

    void skip_mode_params()
    {
        bool allowed = false;
        int f0 = 0, f1 = 0;
        if (!frame_is_intra && h.reference_select && seq.enable_order_hint) {
            int fwd = -1, bwd = -1, fwd_hint = 0, bwd_hint = 0;
            for (int i = 0; i < kRefsPerFrame; i++) {
                const int hint = ps.slot[h.ref_frame_idx[i]].order_hint;
                if (rel_dist(hint, h.order_hint) < 0) {
                    if (fwd < 0 || rel_dist(hint, fwd_hint) > 0) {
                        fwd = i;
                        fwd_hint = hint;
                    }
                } else if (rel_dist(hint, h.order_hint) > 0) {
                    if (bwd < 0 || rel_dist(hint, bwd_hint) < 0) {
                        bwd = i;
                        bwd_hint = hint;
                    }
                }
            }
            if (fwd >= 0 && bwd >= 0) {
                allowed = true;
                f0 = 1 + std::min(fwd, bwd);
                f1 = 1 + std::max(fwd, bwd);
            } else if (fwd >= 0) {
                int second = -1, second_hint = 0;
                for (int i = 0; i < kRefsPerFrame; i++) {
                    const int hint = ps.slot[h.ref_frame_idx[i]].order_hint;
                    if (rel_dist(hint, fwd_hint) < 0 && (second < 0 || rel_dist(hint, second_hint) > 0)) {
                        second = i;
                        second_hint = hint;
                    }
                }
                if (second >= 0) {
                    allowed = true;
                    f0 = 1 + std::min(fwd, second);
                    f1 = 1 + std::max(fwd, second);
                }
            }
        }
        h.skip_mode_present = allowed ? uint8_t(b.f(1)) : 0;
        h.skip_mode_frame[0] = allowed ? uint8_t(f0) : 0;
        h.skip_mode_frame[1] = allowed ? uint8_t(f1) : 0;
    }

    // 5.9.25 .. 5.9.28
    int decode_subexp(int num_syms)
    {
        int i = 0, mk = 0;
        const int k = 3;
        for (;;) {
            const int b2 = i ? k + i - 1 : k, a = 1 << b2;
            if (num_syms <= mk + 3 * a) return int(b.ns(uint32_t(num_syms - mk))) + mk;
            if (!b.f(1)) return int(b.f(b2)) + mk;
            i++;
            mk += a;
            if (b.overrun) return 0;
        }
    }
    static int inverse_recenter(int r, int v)
    {
        if (v > 2 * r) return v;
        return (v & 1) ? r - ((v + 1) >> 1) : r + (v >> 1);
    }
    int decode_unsigned_subexp_with_ref(int mx, int r)
    {
        const int v = decode_subexp(mx);
        return (r << 1) <= mx ? inverse_recenter(r, v) : mx - 1 - inverse_recenter(mx - 1 - r, v);
    }
    void read_global_param(int type, int ref, int idx)
    {
        int abs_bits = 12, prec_bits = 15;
        if (idx < 2) {
            if (type == EA1_GM_TRANSLATION) {
                abs_bits = 9 - !h.allow_high_precision_mv;
                prec_bits = 3 - !h.allow_high_precision_mv;
            } else {
                abs_bits = 12;
                prec_bits = 6;
            }
        }
        const int prec_diff = 16 - prec_bits, round = (idx % 3) == 2 ? 1 << 16 : 0, sub = (idx % 3) == 2 ? 1 << prec_bits : 0, mx = 1 << abs_bits;
        const int r = (prev_gm[ref][idx] >> prec_diff) - sub;
        const int v = decode_unsigned_subexp_with_ref(2 * mx + 1, r + mx) - mx;
        h.gm_params[ref][idx] = v * (1 << prec_diff) + round;
    }
    void global_motion_params()
    {
        default_gm(h.gm_params);
        if (frame_is_intra) return;
        for (int ref = 0; ref < 7; ref++) {
            int type = EA1_GM_IDENTITY;
            if (b.f(1)) type = b.f(1) ? EA1_GM_ROTZOOM : (b.f(1) ? EA1_GM_TRANSLATION : EA1_GM_AFFINE);
            h.gm_type[ref] = uint8_t(type);
            if (type >= EA1_GM_ROTZOOM) {
                read_global_param(type, ref, 2);
                read_global_param(type, ref, 3);
                if (type == EA1_GM_AFFINE) {
                    read_global_param(type, ref, 4);
                    read_global_param(type, ref, 5);
                } else {
                    h.gm_params[ref][4] = -h.gm_params[ref][3];
                    h.gm_params[ref][5] = h.gm_params[ref][2];
                }
            }
            if (type >= EA1_GM_TRANSLATION) {
                read_global_param(type, ref, 0);
                read_global_param(type, ref, 1);
            }
        }
    }

    int film_grain_params()
    {
        ea1_av1_film_grain_syntax &g = h.film_grain;
        memset(&g, 0, sizeof g);
        if (!seq.film_grain_params_present || (!h.show_frame && !h.showable_frame)) return EA1_OK;
        g.apply_grain = uint8_t(b.f(1));
        if (!g.apply_grain) return EA1_OK;
        g.grain_seed = uint16_t(b.f(16));
        g.update_grain = h.frame_type == EA1_INTER_FRAME ? uint8_t(b.f(1)) : 1;
        if (!g.update_grain) {
            const int idx = int(b.f(3));
            bool is_ref = false;
            for (int i = 0; i < kRefsPerFrame; i++) is_ref |= h.ref_frame_idx[i] == idx;
            if (!is_ref || !ps.slot[idx].valid) return ea1::fail(EA1_EINVAL, "frame header: film_grain_params_ref_idx %d is not one of the frame's references", idx);
            const uint16_t seed = g.grain_seed;
            g = ps.slot[idx].grain;
            g.grain_seed = seed;
            g.apply_grain = 1;
            g.update_grain = 0;
            g.film_grain_params_ref_idx = uint8_t(idx);
            return EA1_OK;
        }
        g.num_y_points = uint8_t(b.f(4));
        if (g.num_y_points > 14) return ea1::fail(EA1_EINVAL, "frame header: num_y_points %u", g.num_y_points);
        for (int i = 0; i < g.num_y_points; i++) {
            g.point_y_value[i] = uint8_t(b.f(8));
            g.point_y_scaling[i] = uint8_t(b.f(8));
        }
        g.chroma_scaling_from_luma = seq.monochrome ? 0 : uint8_t(b.f(1));
        if (!(seq.monochrome || g.chroma_scaling_from_luma || (seq.subsampling_x == 1 && seq.subsampling_y == 1 && g.num_y_points == 0))) {
            g.num_cb_points = uint8_t(b.f(4));
            if (g.num_cb_points > 10) return ea1::fail(EA1_EINVAL, "frame header: num_cb_points %u", g.num_cb_points);
            for (int i = 0; i < g.num_cb_points; i++) {
                g.point_cb_value[i] = uint8_t(b.f(8));
                g.point_cb_scaling[i] = uint8_t(b.f(8));
            }
            g.num_cr_points = uint8_t(b.f(4));
            if (g.num_cr_points > 10) return ea1::fail(EA1_EINVAL, "frame header: num_cr_points %u", g.num_cr_points);
            for (int i = 0; i < g.num_cr_points; i++) {
                g.point_cr_value[i] = uint8_t(b.f(8));
                g.point_cr_scaling[i] = uint8_t(b.f(8));
            }
            // AV1 spec 6.8.20: with 4:2:0 the two chroma planes have scaling points both or neither (dav1d refuses the header otherwise: "Error parsing frame header")
            if (seq.subsampling_x == 1 && seq.subsampling_y == 1 && !g.num_cb_points != !g.num_cr_points)
                return ea1::fail(EA1_EINVAL, "frame header: film grain with scaling points for one chroma plane only (4:2:0: num_cb_points %u, num_cr_points %u)", g.num_cb_points, g.num_cr_points);
        }
        g.grain_scaling_minus_8 = uint8_t(b.f(2));
        g.ar_coeff_lag = uint8_t(b.f(2));
        const int num_pos_luma = 2 * g.ar_coeff_lag * (g.ar_coeff_lag + 1);
        int num_pos_chroma = num_pos_luma;
        if (g.num_y_points) {
            num_pos_chroma = num_pos_luma + 1;
            for (int i = 0; i < num_pos_luma; i++) g.ar_coeffs_y_plus_128[i] = uint8_t(b.f(8));
        }
        if (g.chroma_scaling_from_luma || g.num_cb_points)
            for (int i = 0; i < num_pos_chroma; i++) g.ar_coeffs_cb_plus_128[i] = uint8_t(b.f(8));
        if (g.chroma_scaling_from_luma || g.num_cr_points)
            for (int i = 0; i < num_pos_chroma; i++) g.ar_coeffs_cr_plus_128[i] = uint8_t(b.f(8));
        g.ar_coeff_shift_minus_6 = uint8_t(b.f(2));
        g.grain_scale_shift = uint8_t(b.f(2));
        if (g.num_cb_points) {
            g.cb_mult = uint8_t(b.f(8));
            g.cb_luma_mult = uint8_t(b.f(8));
            g.cb_offset = uint16_t(b.f(9));
        }
        if (g.num_cr_points) {
            g.cr_mult = uint8_t(b.f(8));
            g.cr_luma_mult = uint8_t(b.f(8));
            g.cr_offset = uint16_t(b.f(9));
        }
        g.overlap_flag = uint8_t(b.f(1));
        g.clip_to_restricted_range = uint8_t(b.f(1));
        return EA1_OK;
    }

    int run()
    {
        const int id_len = seq.frame_id_numbers_present ? seq.additional_frame_id_length + seq.delta_frame_id_length : 0;
        const int all_frames = 0xFF;
        num_planes = seq.monochrome ? 1 : 3;
        memset(&h, 0, sizeof h);
        h.primary_ref_frame = EA1_PRIMARY_REF_NONE;
        h.superres_denom = 8;
        h.tile_size_bytes = 4;
        if (seq.reduced_still_picture_header) {
            h.frame_type = EA1_KEY_FRAME;
            frame_is_intra = true;
            h.show_frame = 1;
            h.error_resilient_mode = 1;
        } else {
            h.show_existing_frame = uint8_t(b.f(1));
            if (h.show_existing_frame) {
                h.frame_to_show_map_idx = uint8_t(b.f(3));
                if (seq.decoder_model_info_present && !seq.equal_picture_interval) h.frame_presentation_time = b.f(seq.frame_presentation_time_length);
                if (seq.frame_id_numbers_present) h.display_frame_id = b.f(id_len);
                const Slot &s = ps.slot[h.frame_to_show_map_idx];
                if (!s.valid) return ea1::fail(EA1_EINVAL, "frame header: show_existing_frame of empty slot %u", h.frame_to_show_map_idx);
                if (seq.frame_id_numbers_present && h.display_frame_id != s.frame_id) return ea1::fail(EA1_EINVAL, "frame header: display_frame_id mismatch");
                h.frame_type = s.frame_type;
                h.show_frame = 1;
                h.showable_frame = s.showable_frame;
                h.refresh_frame_flags = h.frame_type == EA1_KEY_FRAME ? uint8_t(all_frames) : 0;
                h.upscaled_width = s.upscaled_width;
                h.frame_width = s.frame_width;
                h.frame_height = s.frame_height;
                h.render_width = s.render_width;
                h.render_height = s.render_height;
                h.order_hint = s.order_hint;
                memcpy(h.gm_params, s.gm_params, sizeof h.gm_params);
                if (seq.film_grain_params_present) h.film_grain = s.grain;
                h.header_bits = uint32_t(b.pos);
                return b.overrun ? ea1::fail(EA1_EINVAL, "frame header: truncated") : EA1_OK;
            }
            h.frame_type = uint8_t(b.f(2));
            frame_is_intra = h.frame_type == EA1_INTRA_ONLY_FRAME || h.frame_type == EA1_KEY_FRAME;
            h.show_frame = uint8_t(b.f(1));
            if (h.show_frame && seq.decoder_model_info_present && !seq.equal_picture_interval) h.frame_presentation_time = b.f(seq.frame_presentation_time_length);
            h.showable_frame = h.show_frame ? uint8_t(h.frame_type != EA1_KEY_FRAME) : uint8_t(b.f(1));
            if (h.frame_type == EA1_SWITCH_FRAME || (h.frame_type == EA1_KEY_FRAME && h.show_frame)) h.error_resilient_mode = 1;
            else h.error_resilient_mode = uint8_t(b.f(1));
        }
        // a shown key frame invalidates every slot (state as seen by THIS header; end_frame overwrites them anyway)
        const bool reset_refs = h.frame_type == EA1_KEY_FRAME && h.show_frame;
        auto slot_valid = [&](int i) { return !reset_refs && ps.slot[i].valid; };
        auto slot_hint = [&](int i) { return reset_refs ? 0 : int(ps.slot[i].order_hint); };
        h.disable_cdf_update = uint8_t(b.f(1));
        h.allow_screen_content_tools = seq.seq_force_screen_content_tools == 2 ? uint8_t(b.f(1)) : seq.seq_force_screen_content_tools;
        if (h.allow_screen_content_tools) h.force_integer_mv = seq.seq_force_integer_mv == 2 ? uint8_t(b.f(1)) : seq.seq_force_integer_mv;
        if (frame_is_intra) h.force_integer_mv = 1;
        if (seq.frame_id_numbers_present) h.current_frame_id = b.f(id_len);
        if (h.frame_type == EA1_SWITCH_FRAME) h.frame_size_override_flag = 1;
        else if (!seq.reduced_still_picture_header) h.frame_size_override_flag = uint8_t(b.f(1));
        h.order_hint = uint8_t(b.f(seq.order_hint_bits));
        if (!(frame_is_intra || h.error_resilient_mode)) h.primary_ref_frame = uint8_t(b.f(3));
        if (seq.decoder_model_info_present) {
            h.buffer_removal_time_present = uint8_t(b.f(1));
            if (h.buffer_removal_time_present)
                for (int op = 0; op < seq.num_operating_points; op++) {
                    if (!seq.decoder_model_present_for_op[op]) continue;
                    const int idc = seq.operating_point_idc[op];
                    const bool in_t = (idc >> temporal_id) & 1, in_s = (idc >> (spatial_id + 8)) & 1;
                    if (idc == 0 || (in_t && in_s)) h.buffer_removal_time[op] = b.f(seq.buffer_removal_time_length);
                }
        }
        if (h.frame_type == EA1_SWITCH_FRAME || (h.frame_type == EA1_KEY_FRAME && h.show_frame)) h.refresh_frame_flags = uint8_t(all_frames);
        else h.refresh_frame_flags = uint8_t(b.f(8));
        if (h.frame_type == EA1_INTRA_ONLY_FRAME && h.refresh_frame_flags == all_frames) return ea1::fail(EA1_EINVAL, "frame header: intra-only frame that refreshes every slot");
        if ((!frame_is_intra || h.refresh_frame_flags != all_frames) && h.error_resilient_mode && seq.enable_order_hint)
            for (int i = 0; i < kRefs; i++) {
                h.ref_order_hint[i] = uint8_t(b.f(seq.order_hint_bits));
                if (h.ref_order_hint[i] != slot_hint(i) || !slot_valid(i)) {   // 7.20: the slot becomes an empty frame with that hint
                    Slot &s = ps.slot[i];
                    s = Slot();
                    default_gm(s.gm_params);
                    memcpy(s.lf_ref_deltas, kDefaultRefDeltas, 8);
                    s.order_hint = h.ref_order_hint[i];
                    // stays invalid for prediction: a header that references it is refused below
                }
            }
        if (frame_is_intra) {
            frame_size();
            render_size();
            if (h.allow_screen_content_tools && h.upscaled_width == h.frame_width) h.allow_intrabc = uint8_t(b.f(1));
        } else {
            if (seq.enable_order_hint) {
                h.frame_refs_short_signaling = uint8_t(b.f(1));
                if (h.frame_refs_short_signaling) {
                    const int last_idx = int(b.f(3)), gold_idx = int(b.f(3));
                    const int rc = set_frame_refs(last_idx, gold_idx);
                    if (rc != EA1_OK) return rc;
                }
            }
            for (int i = 0; i < kRefsPerFrame; i++) {
                if (!h.frame_refs_short_signaling) h.ref_frame_idx[i] = int8_t(b.f(3));
                if (seq.frame_id_numbers_present) {
                    const uint32_t delta = b.f(seq.delta_frame_id_length) + 1;
                    const uint32_t expected = (h.current_frame_id + (1u << id_len) - delta) % (1u << id_len);
                    if (ps.slot[h.ref_frame_idx[i]].valid && ps.slot[h.ref_frame_idx[i]].frame_id != expected)
                        return ea1::fail(EA1_EINVAL, "frame header: reference %d has frame id %u, expected %u", i, ps.slot[h.ref_frame_idx[i]].frame_id, expected);
                }
                if (!ps.slot[h.ref_frame_idx[i]].valid) return ea1::fail(EA1_EINVAL, "frame header: reference slot %d is empty", h.ref_frame_idx[i]);
            }
            if (h.frame_size_override_flag && !h.error_resilient_mode) {
                frame_size_with_refs();
            } else {
                frame_size();
                render_size();
            }
            h.allow_high_precision_mv = h.force_integer_mv ? 0 : uint8_t(b.f(1));
            h.is_filter_switchable = uint8_t(b.f(1));
            h.interpolation_filter = h.is_filter_switchable ? 4 : uint8_t(b.f(2));
            h.is_motion_mode_switchable = uint8_t(b.f(1));
            h.use_ref_frame_mvs = (h.error_resilient_mode || !seq.enable_ref_frame_mvs) ? 0 : uint8_t(b.f(1));
            for (int i = 0; i < kRefsPerFrame; i++) order_hints[1 + i] = ps.slot[h.ref_frame_idx[i]].order_hint;
        }
        if (h.frame_width > seq.max_width || h.frame_height > seq.max_height || h.upscaled_width > seq.max_width)
            return ea1::fail(EA1_EINVAL, "frame header: frame %ux%u larger than the sequence maximum %ux%u", h.upscaled_width, h.frame_height, seq.max_width, seq.max_height);
        h.disable_frame_end_update_cdf = (seq.reduced_still_picture_header || h.disable_cdf_update) ? 1 : uint8_t(b.f(1));
        // load_previous() / setup_past_independence(): what the delta-coded parts start from
        default_gm(prev_gm);
        memcpy(h.loop_filter_ref_deltas, kDefaultRefDeltas, 8);
        h.loop_filter_delta_enabled = 0;
        if (h.primary_ref_frame != EA1_PRIMARY_REF_NONE) {
            const Slot &s = ps.slot[h.ref_frame_idx[h.primary_ref_frame]];
            memcpy(prev_gm, s.gm_params, sizeof prev_gm);
            memcpy(h.loop_filter_ref_deltas, s.lf_ref_deltas, 8);
            memcpy(h.loop_filter_mode_deltas, s.lf_mode_deltas, 2);
            memcpy(h.feature_enabled, s.feature_enabled, sizeof h.feature_enabled);
            memcpy(h.feature_data, s.feature_data, sizeof h.feature_data);
        }
        int rc = tile_info();
        if (rc != EA1_OK) return rc;
        quantization_params();
        segmentation_params();
        if (h.base_q_idx > 0) h.delta_q_present = uint8_t(b.f(1));
        if (h.delta_q_present) h.delta_q_res = uint8_t(b.f(2));
        if (h.delta_q_present) {
            if (!h.allow_intrabc) h.delta_lf_present = uint8_t(b.f(1));
            if (h.delta_lf_present) {
                h.delta_lf_res = uint8_t(b.f(2));
                h.delta_lf_multi = uint8_t(b.f(1));
            }
        }
        h.coded_lossless = 1;
        for (int seg = 0; seg < 8; seg++) {
            int q = h.base_q_idx;
            if (h.segmentation_enabled && h.feature_enabled[seg][0]) q = std::min(std::max(q + h.feature_data[seg][0], 0), 255);
            if (!(q == 0 && !h.delta_q_y_dc && !h.delta_q_u_ac && !h.delta_q_u_dc && !h.delta_q_v_ac && !h.delta_q_v_dc)) h.coded_lossless = 0;
        }
        h.all_lossless = h.coded_lossless && h.frame_width == h.upscaled_width;
        loop_filter_params();
        cdef_params();
        lr_params();
        h.tx_mode_select = h.coded_lossless ? 0 : uint8_t(b.f(1));
        h.reference_select = frame_is_intra ? 0 : uint8_t(b.f(1));
        skip_mode_params();
        h.allow_warped_motion = (frame_is_intra || h.error_resilient_mode || !seq.enable_warped_motion) ? 0 : uint8_t(b.f(1));
        h.reduced_tx_set = uint8_t(b.f(1));
        global_motion_params();
        rc = film_grain_params();
        if (rc != EA1_OK) return rc;
        h.header_bits = uint32_t(b.pos);
        if (b.overrun) return ea1::fail(EA1_EINVAL, "frame header: truncated");
        return EA1_OK;
    }
};

}  // namespace

extern "C" int ea1_av1_parser_create(ea1_av1_parser **out)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_create: out is NULL");
    *out = new (std::nothrow) ea1_av1_parser();
    return *out ? EA1_OK : ea1::fail(EA1_ENOMEM, "out of memory");
}

extern "C" void ea1_av1_parser_destroy(ea1_av1_parser *p) { delete p; }

extern "C" int ea1_av1_parser_set_sequence_header(ea1_av1_parser *p, const ea1_av1_sequence_header *seq)
{
    if (!p || !seq) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_set_sequence_header: NULL argument");
    if (p->have_seq && memcmp(&p->seq, seq, sizeof *seq) == 0) return EA1_OK;
    p->seq = *seq;
    p->have_seq = true;
    for (Slot &s : p->slot) s = Slot();
    return EA1_OK;
}

extern "C" int ea1_av1_parser_frame_header(ea1_av1_parser *p, const uint8_t *payload, size_t size, int temporal_id, int spatial_id,
                                           ea1_av1_frame_header *out)
{
    if (!p || !payload || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_frame_header: NULL argument");
    if (!p->have_seq) return ea1::fail(EA1_ESTATE, "ea1_av1_parser_frame_header: no sequence header yet");
    if (temporal_id < 0 || temporal_id > 7 || spatial_id < 0 || spatial_id > 3) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_frame_header: layer ids");
    Parse ps{*p, p->seq, BitReader{payload, size}, *out, temporal_id, spatial_id};
    return ps.run();
}

extern "C" int ea1_av1_parser_end_frame(ea1_av1_parser *p, const ea1_av1_frame_header *h)
{
    if (!p || !h) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_end_frame: NULL argument");
    if (!p->have_seq) return ea1::fail(EA1_ESTATE, "ea1_av1_parser_end_frame: no sequence header yet");
    if (h->show_existing_frame && h->frame_type != EA1_KEY_FRAME) return EA1_OK;   // nothing is refreshed
    Slot s;
    if (h->show_existing_frame) {
        s = p->slot[h->frame_to_show_map_idx];   // 7.21: the shown key frame is loaded, then refreshes every slot
        if (!s.valid) return ea1::fail(EA1_ESTATE, "ea1_av1_parser_end_frame: slot %u is empty", h->frame_to_show_map_idx);
    } else {
        s.valid = true;
        s.frame_id = h->current_frame_id;
        s.upscaled_width = h->upscaled_width;
        s.frame_width = h->frame_width;
        s.frame_height = h->frame_height;
        s.render_width = h->render_width;
        s.render_height = h->render_height;
        s.frame_type = h->frame_type;
        s.order_hint = h->order_hint;
        s.showable_frame = h->showable_frame;
        memcpy(s.gm_params, h->gm_params, sizeof s.gm_params);
        memcpy(s.lf_ref_deltas, h->loop_filter_ref_deltas, 8);
        memcpy(s.lf_mode_deltas, h->loop_filter_mode_deltas, 2);
        memcpy(s.feature_enabled, h->feature_enabled, sizeof s.feature_enabled);
        memcpy(s.feature_data, h->feature_data, sizeof s.feature_data);
        s.grain = h->film_grain;
    }
    for (int i = 0; i < kRefs; i++)
        if (h->refresh_frame_flags & (1 << i)) p->slot[i] = s;
    p->current_frame_id = h->current_frame_id;
    return EA1_OK;
}

extern "C" int ea1_av1_frame_header_to_desc(const ea1_av1_frame_header *h, const ea1_av1_sequence_header *seq, ea1_frame_desc *d)
{
    if (!h || !seq || !d) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_header_to_desc: NULL argument");
    // (4:0:0 goes through the 4:2:0 path with nothing coded in the chroma planes: they stay at 128 and are not shown)
    if (seq->hbd) return ea1::fail(EA1_EINVAL, "the reconstruction path decodes 8-bit streams only");
    memset(d, 0, sizeof *d);
    // 4:4:4 / 4:2:2: the records of such a frame count chroma positions and sizes in the chroma plane's own samples; the context that takes them has the
    // same layout (ea1_recon_create_layout)
    d->layout = uint8_t(seq->layout == EA1_LAYOUT_I444 ? EA1_LAYOUT_444 : (seq->layout == EA1_LAYOUT_I422 ? EA1_LAYOUT_422 : EA1_LAYOUT_420));
    // super-resolution: the frame is coded (and reconstructed up to CDEF) at frame_width, the picture that comes out is upscaled_width wide
    d->upscaled_width = (h->use_superres && h->upscaled_width != h->frame_width) ? h->upscaled_width : 0;
    d->width = h->mi_cols * 4;     // the coded area: whole 8x8 luma units (the picture is cropped to frame_width x frame_height on output)
    d->height = h->mi_rows * 4;
    d->crop = uint8_t((d->width - h->frame_width) | ((d->height - h->frame_height) << 4));   // 0..7 each: mi units are 4 samples, pairs of them are coded
    d->is_inter = !(h->frame_type == EA1_KEY_FRAME || h->frame_type == EA1_INTRA_ONLY_FRAME);
    memcpy(d->lf_level, h->loop_filter_level, 4);
    d->lf_sharpness = h->loop_filter_sharpness;
    bool deltas = h->delta_lf_present;
    if (h->loop_filter_delta_enabled) {
        for (int i = 0; i < 8; i++) deltas |= h->loop_filter_ref_deltas[i] != 0;
        deltas |= h->loop_filter_mode_deltas[0] || h->loop_filter_mode_deltas[1];
    }
    if (h->segmentation_enabled)
        for (int s = 0; s < 8; s++)
            for (int f = 1; f <= 4; f++) deltas |= h->feature_enabled[s][f] != 0;   // SEG_LVL_ALT_LF_*
    d->lf_per_block = deltas && (h->loop_filter_level[0] || h->loop_filter_level[1]);
    d->cdef_enable = seq->enable_cdef && !h->coded_lossless && !h->allow_intrabc;
    d->cdef_damping = h->cdef_damping;
    for (int i = 0; i < (1 << h->cdef_bits); i++) {
        d->cdef_y_strength[i] = uint8_t((h->cdef_y_pri_strength[i] << 2) | std::min<int>(h->cdef_y_sec_strength[i], 3));
        d->cdef_uv_strength[i] = uint8_t((h->cdef_uv_pri_strength[i] << 2) | std::min<int>(h->cdef_uv_sec_strength[i], 3));
    }
    for (int p = 0; p < 3; p++) {
        d->lr_type[p] = h->frame_restoration_type[p] ? 1 : 0;
        int l2 = 5;   // (chroma units of 32 samples: lr_unit_shift = 0 with lr_uv_shift = 1 -- read as 64 until round 5, found beside dav1d on a generated stream)
        while ((1 << l2) < h->loop_restoration_size[p]) l2++;
        d->lr_unit_log2[p] = uint8_t(h->loop_restoration_size[p] ? l2 : (p ? 5 : 6));
    }
    d->intra_edge_filter = seq->enable_intra_edge_filter;
    return EA1_OK;
}

extern "C" int ea1_av1_frame_header_film_grain(const ea1_av1_frame_header *h, const ea1_av1_sequence_header *seq, ea1_film_grain *out)
{
    if (!h || !seq || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_header_film_grain: NULL argument");
    const ea1_av1_film_grain_syntax &g = h->film_grain;
    if (!g.apply_grain) return ea1::fail(EA1_ENOENT, "the frame carries no film grain");
    memset(out, 0, sizeof *out);
    out->grain_seed = g.grain_seed;
    out->num_y_points = g.num_y_points;
    for (int i = 0; i < g.num_y_points; i++) {
        out->point_y[i][0] = g.point_y_value[i];
        out->point_y[i][1] = g.point_y_scaling[i];
    }
    out->chroma_scaling_from_luma = g.chroma_scaling_from_luma;
    out->num_uv_points[0] = g.num_cb_points;
    out->num_uv_points[1] = g.num_cr_points;
    for (int i = 0; i < g.num_cb_points; i++) {
        out->point_uv[0][i][0] = g.point_cb_value[i];
        out->point_uv[0][i][1] = g.point_cb_scaling[i];
    }
    for (int i = 0; i < g.num_cr_points; i++) {
        out->point_uv[1][i][0] = g.point_cr_value[i];
        out->point_uv[1][i][1] = g.point_cr_scaling[i];
    }
    out->scaling_shift = uint8_t(g.grain_scaling_minus_8 + 8);
    out->ar_coeff_lag = g.ar_coeff_lag;
    const int npl = 2 * g.ar_coeff_lag * (g.ar_coeff_lag + 1);
    for (int i = 0; i < npl && g.num_y_points; i++) out->ar_coeffs_y[i] = int8_t(int(g.ar_coeffs_y_plus_128[i]) - 128);
    // the chroma list holds one more coefficient (the luma weight) only when luma grain exists
    const int npc = npl + (g.num_y_points ? 1 : 0);
    for (int i = 0; i < npc; i++) {
        if (g.chroma_scaling_from_luma || g.num_cb_points) out->ar_coeffs_uv[0][i] = int8_t(int(g.ar_coeffs_cb_plus_128[i]) - 128);
        if (g.chroma_scaling_from_luma || g.num_cr_points) out->ar_coeffs_uv[1][i] = int8_t(int(g.ar_coeffs_cr_plus_128[i]) - 128);
    }
    out->ar_coeff_shift = uint8_t(g.ar_coeff_shift_minus_6 + 6);
    out->grain_scale_shift = g.grain_scale_shift;
    out->uv_mult[0] = g.cb_mult;
    out->uv_mult[1] = g.cr_mult;
    out->uv_luma_mult[0] = g.cb_luma_mult;
    out->uv_luma_mult[1] = g.cr_luma_mult;
    out->uv_offset[0] = g.cb_offset;
    out->uv_offset[1] = g.cr_offset;
    if (!g.num_cb_points) {   // not coded: neutral values (unused by the synthesis without points)
        out->uv_mult[0] = out->uv_luma_mult[0] = 128;
        out->uv_offset[0] = 256;
    }
    if (!g.num_cr_points) {
        out->uv_mult[1] = out->uv_luma_mult[1] = 128;
        out->uv_offset[1] = 256;
    }
    out->overlap_flag = g.overlap_flag;
    out->clip_to_restricted_range = g.clip_to_restricted_range;
    out->matrix_is_identity = seq->matrix_coefficients == 0;
    return EA1_OK;
}

extern "C" int ea1_av1_parse_tile_group(const ea1_av1_frame_header *h, const uint8_t *payload, size_t size, ea1_av1_tile *tiles, int cap, int *count)
{
    if (!h || !payload || !count || (cap > 0 && !tiles)) return ea1::fail(EA1_EINVAL, "ea1_av1_parse_tile_group: NULL argument");
    *count = 0;
    if (h->show_existing_frame || !h->tile_cols || !h->tile_rows) return ea1::fail(EA1_EINVAL, "ea1_av1_parse_tile_group: the header describes no coded frame");
    const int num_tiles = h->tile_cols * h->tile_rows;
    BitReader b{payload, size};
    int tg_start = 0, tg_end = num_tiles - 1;
    if (num_tiles > 1 && b.f(1)) {   // tile_start_and_end_present_flag
        const int bits = h->tile_cols_log2 + h->tile_rows_log2;
        tg_start = int(b.f(bits));
        tg_end = int(b.f(bits));
    }
    if (b.overrun || tg_start > tg_end || tg_end >= num_tiles) return ea1::fail(EA1_EINVAL, "tile group: tiles %d..%d of %d", tg_start, tg_end, num_tiles);
    size_t pos = (b.pos + 7) >> 3;   // byte_alignment()
    for (int t = tg_start; t <= tg_end; t++) {
        size_t tile_size;
        if (t == tg_end) {
            if (pos > size) return ea1::fail(EA1_EINVAL, "tile group: truncated");
            tile_size = size - pos;
        } else {
            if (pos + h->tile_size_bytes > size) return ea1::fail(EA1_EINVAL, "tile group: truncated tile size");
            uint64_t v = 0;   // le(TileSizeBytes)
            for (int k = 0; k < h->tile_size_bytes; k++) v |= uint64_t(payload[pos + size_t(k)]) << (8 * k);
            pos += h->tile_size_bytes;
            tile_size = size_t(v) + 1;
            if (tile_size > size - pos) return ea1::fail(EA1_EINVAL, "tile group: tile %d of %zu bytes, %zu left", t, tile_size, size - pos);
        }
        if (!tile_size) return ea1::fail(EA1_EINVAL, "tile group: empty tile %d", t);
        if (*count < cap) {
            ea1_av1_tile &o = tiles[*count];
            memset(&o, 0, sizeof o);
            o.tile_num = uint16_t(t);
            o.tile_row = uint16_t(t / h->tile_cols);
            o.tile_col = uint16_t(t % h->tile_cols);
            o.mi_row_start = h->mi_row_starts[o.tile_row];
            o.mi_row_end = h->mi_row_starts[o.tile_row + 1];
            o.mi_col_start = h->mi_col_starts[o.tile_col];
            o.mi_col_end = h->mi_col_starts[o.tile_col + 1];
            o.offset = pos;
            o.size = tile_size;
        }
        (*count)++;
        pos += tile_size;
    }
    if (*count > cap) return ea1::fail(EA1_ECAPACITY, "tile group: %d tiles, room for %d", *count, cap);
    return EA1_OK;
}

extern "C" int ea1_av1_parser_slot_info(const ea1_av1_parser *p, int slot, ea1_av1_slot_info *out)
{
    if (!p || !out || slot < 0 || slot >= kRefs) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_slot_info: bad argument");
    const Slot &s = p->slot[slot];
    out->valid = s.valid;
    out->frame_type = s.frame_type;
    out->order_hint = s.order_hint;
    out->showable_frame = s.showable_frame;
    out->upscaled_width = s.upscaled_width;
    out->frame_width = s.frame_width;
    out->frame_height = s.frame_height;
    return EA1_OK;
}

extern "C" int ea1_av1_parser_sequence_header(const ea1_av1_parser *p, ea1_av1_sequence_header *out)
{
    if (!p || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_sequence_header: NULL argument");
    if (!p->have_seq) return ea1::fail(EA1_ESTATE, "no sequence header yet");
    *out = p->seq;
    return EA1_OK;
}

static int next_frame_impl(ea1_av1_parser *p, const uint8_t *data, size_t size, size_t *consumed, ea1_av1_frame_header *hdr,
                           ea1_av1_tile *tiles, int cap, int *count);

extern "C" int ea1_av1_parser_next_frame(ea1_av1_parser *p, const uint8_t *data, size_t size, size_t *consumed, ea1_av1_frame_header *hdr,
                                         ea1_av1_tile *tiles, int cap, int *count)
{
    if (!p || !data || !consumed || !hdr || !count || (cap > 0 && !tiles)) return ea1::fail(EA1_EINVAL, "ea1_av1_parser_next_frame: NULL argument");
    const int rc = next_frame_impl(p, data, size, consumed, hdr, tiles, cap, count);
    if (rc != EA1_OK) *count = 0;   // no partial frame is ever reported
    return rc;
}

static int next_frame_impl(ea1_av1_parser *p, const uint8_t *data, size_t size, size_t *consumed, ea1_av1_frame_header *hdr,
                           ea1_av1_tile *tiles, int cap, int *count)
{
    *consumed = 0;
    *count = 0;
    bool have_header = false;
    const uint8_t *first_header = nullptr;   // uncompressed_header() bytes of the frame, for the copy check
    size_t first_header_bytes = 0;
    int next_tile = 0, num_tiles = 0;
    size_t pos = 0;
    while (pos < size) {
        ea1_obu o;
        int rc = ea1_av1_next_obu(data + pos, size - pos, &o);
        if (rc != EA1_OK) return rc;
        const size_t obu_end = pos + o.total_size;
        // drop_obu: with an extension, only layers that operating point 0 contains are decoded
        bool drop = false;
        if (o.type != EA1_OBU_SEQUENCE_HEADER && o.type != EA1_OBU_TEMPORAL_DELIMITER && o.has_extension && p->have_seq) {
            const int idc = p->seq.operating_point_idc[0];
            if (idc && !(((idc >> o.temporal_id) & 1) && ((idc >> (o.spatial_id + 8)) & 1))) drop = true;
        }
        if (!drop) switch (o.type) {
        case EA1_OBU_SEQUENCE_HEADER: {
            if (have_header) return ea1::fail(EA1_EINVAL, "sequence header inside a frame");
            ea1_av1_sequence_header sh;
            rc = ea1_av1_parse_sequence_header(data + pos, o.total_size, &sh);
            if (rc != EA1_OK) return rc;
            rc = ea1_av1_parser_set_sequence_header(p, &sh);
            if (rc != EA1_OK) return rc;
            break;
        }
        case EA1_OBU_TEMPORAL_DELIMITER:
            if (have_header) return ea1::fail(EA1_EINVAL, "temporal delimiter inside a frame");
            break;
        case EA1_OBU_FRAME_HEADER:
        case EA1_OBU_REDUNDANT_FRAME_HEADER:
        case EA1_OBU_FRAME: {
            if (!p->have_seq) return ea1::fail(EA1_EINVAL, "frame before any sequence header");
            if (have_header) {
                if (o.type == EA1_OBU_FRAME) return ea1::fail(EA1_EINVAL, "OBU_FRAME inside a frame");
                // frame_header_copy(): must repeat the first header bit for bit
                if (o.payload_size < first_header_bytes || memcmp(o.payload, first_header, (hdr->header_bits) >> 3) != 0)
                    return ea1::fail(EA1_EINVAL, "repeated frame header differs from the first");
                break;
            }
            if (o.type == EA1_OBU_REDUNDANT_FRAME_HEADER) return ea1::fail(EA1_EINVAL, "redundant frame header without a frame header");
            rc = ea1_av1_parser_frame_header(p, o.payload, o.payload_size, o.temporal_id, o.spatial_id, hdr);
            if (rc != EA1_OK) return rc;
            if (hdr->show_existing_frame) {
                if (o.type == EA1_OBU_FRAME) return ea1::fail(EA1_EINVAL, "OBU_FRAME with show_existing_frame");
                *consumed = obu_end;
                return EA1_OK;
            }
            have_header = true;
            first_header = o.payload;
            first_header_bytes = (hdr->header_bits + 7) >> 3;
            num_tiles = hdr->tile_cols * hdr->tile_rows;
            if (o.type != EA1_OBU_FRAME) break;
            // OBU_FRAME: byte_alignment, then the tile group in the same OBU
            if (first_header_bytes > o.payload_size) return ea1::fail(EA1_EINVAL, "OBU_FRAME: truncated");
            o.payload += first_header_bytes;
            o.payload_size -= first_header_bytes;
        }
        // fall through: the tile group of an OBU_FRAME
        case EA1_OBU_TILE_GROUP: {
            if (!have_header) return ea1::fail(EA1_EINVAL, "tile group without a frame header");
            int n = 0;
            rc = ea1_av1_parse_tile_group(hdr, o.payload, o.payload_size, tiles + *count, cap - *count, &n);
            if (rc != EA1_OK) return rc;
            for (int k = 0; k < n; k++) {
                ea1_av1_tile &t = tiles[*count + k];
                if (t.tile_num != next_tile) return ea1::fail(EA1_EINVAL, "tile %d where tile %d was expected", t.tile_num, next_tile);
                next_tile++;
                t.offset += size_t(o.payload - data);
            }
            *count += n;
            if (next_tile == num_tiles) {
                *consumed = obu_end;
                return EA1_OK;
            }
            break;
        }
        default: break;   // metadata, tile list, padding, reserved types: passed over
        }
        pos = obu_end;
        if (!have_header) *consumed = pos;
    }
    *count = 0;
    return ea1::fail(EA1_ENOENT, "the data ends before a frame is complete");
}
