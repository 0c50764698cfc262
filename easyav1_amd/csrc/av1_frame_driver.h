// Frame-level driver of the block-level AV1 parser: CDF bookkeeping across frames (AV1 specification 7.20 reference
// frame update / 7.21, the load_cdfs / save_cdfs / init_symbol / exit_symbol processes of 8.2), the mapping of AV1's
// reference slots onto the reconstruction context's eight pictures, and the per-frame set-up of TileSyntax.  Shared by
// the parser (av1_frame_parse.cpp, IO = the symbol decoder) and the stream generator of the tests (IO = an encoder).
#pragma once
#include <cstdio>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <memory>
#include <string>
#include <vector>

#include "av1_tile_syntax.h"
#include "ea1_common.h"

namespace ea1 {
namespace av1 {

struct FrameDriver {
    const Tables *tab = nullptr;
    ea1_av1_parser *hp = nullptr;            // headers + what the header syntax needs of the slots
    std::vector<uint32_t> unit;              // FrameInfo::unit
    std::vector<uint8_t> utx_size, utx_type; // FrameInfo::utx_size / utx_type
    int mi_stride = 0;
    uint32_t stamp = 0;                      // 1 .. 255
    std::unique_ptr<Cdfs> saved[8];          // CDFs saved with each reference slot (save_cdfs)
    Cdfs cur, tile_saved;                    // the frame's CDFs; those of context_update_tile_id after its tile
    uint8_t phys[8];                         // AV1 slot -> picture of the reconstruction context, 0xFF = none
    std::vector<uint8_t> seg_cur, seg_prev;  // SegmentIds of the frame / PrevSegmentIds (load_previous_segment_ids)
    std::vector<uint8_t> seg_saved[8];       // SavedSegmentIds with their size in mi units
    int seg_saved_cols[8] = {0}, seg_saved_rows[8] = {0};
    // motion vectors saved with the slots (7.19) and the motion field projected from them (7.9)
    struct Motion {
        std::vector<uint8_t> ref;     // MfRefFrames per 4x4 unit (REF_NONE: none)
        std::vector<int16_t> mv;      // MfMvs, 2 per unit
        int rows = 0, cols = 0;
        uint8_t order_hints[8] = {0}; // SavedOrderHints[ref]
        uint8_t frame_type = 0;
    };
    std::shared_ptr<Motion> motion[8];
    std::shared_ptr<Motion> cur_motion;   // what this frame will leave in the slots it refreshes: filled by the tiles as they decode (FrameInfo::save_ref)
    std::vector<int16_t> mf;             // the motion field as {saved row, saved column, reference offset} per 8x8 (kInvalidMv: none): MotionFieldMvs[dst] on demand
    FrameOut out;
    FrameInfo info;
    ea1_av1_sequence_header seq;
    ea1_frame_desc desc;

    bool allow_chroma_formats = true;    // hand out frames of 4:2:2 / 4:4:4 streams (ea1_frame_desc.layout says which; a consumer that is 4:2:0 only switches it off)

    FrameDriver() { reset_slots(); }
    void reset_slots()
    {
        for (int i = 0; i < 8; i++) {
            phys[i] = 0xFF;
            saved[i].reset();
            motion[i].reset();
        }
    }

    // what of `hdr` this revision does not decode
    static const char *unsupported(const ea1_av1_sequence_header &s, const ea1_av1_frame_header &h)
    {
        if (s.hbd) return "only 8-bit 4:2:0 (and 4:0:0) is decoded: a stream of 10 / 12 bits";
        (void)h;
        return nullptr;
    }

    // Set-up for the frame of `hdr` (already parsed by hp): CDFs, references, output arrays.  EA1_OK / error code.
    int begin_frame(const ea1_av1_frame_header &hdr)
    {
        int rc = ea1_av1_parser_sequence_header(hp, &seq);
        if (rc != EA1_OK) return rc;
        if (const char *why = unsupported(seq, hdr)) return ea1::fail(EA1_ENOTSUP, "AV1 block parser: %s is not supported", why);
        // 4:2:2 and 4:4:4 (round 5; checked against dav1d through the oracle and through the kernels, tests/test_real_av1_chroma.py): ea1_frame_desc.layout
        // names the layout, the context that reconstructs the frames is made for it; a 4:2:0-only consumer may switch them off
        if (!allow_chroma_formats && seq.layout != EA1_LAYOUT_I420 && seq.layout != EA1_LAYOUT_I400)
            return ea1::fail(EA1_ENOTSUP, "AV1 block parser: the caller takes 8-bit 4:2:0 (and 4:0:0) only; this stream is %s", seq.layout == EA1_LAYOUT_I444 ? "4:4:4" : "4:2:2");
        rc = ea1_av1_frame_header_to_desc(&hdr, &seq, &desc);
        if (rc != EA1_OK) return rc;
        // blocks may reach beyond the coded frame (whenever its size is not a multiple of the block sizes in use at its right / bottom
        // edge: 1080, 720, 2160 rows ...): the hand-over's frame is the coded one rounded up to whole 64x64 areas, ea1_frame_desc.pad
        // (128x128 with superblocks of that size), says by how much, and what the blocks leave open out there is closed with filler
        // blocks in end_frame
        {
            const uint32_t sbm = seq.use_128x128_superblock ? 127u : 63u;
            const uint32_t pw = (desc.width + sbm) & ~sbm, ph = (desc.height + sbm) & ~sbm;
            desc.pad = uint8_t(((pw - desc.width) / 8) | (((ph - desc.height) / 8) << 4));
            desc.width = pw, desc.height = ph;
        }
        const bool intra_only = hdr.frame_type == EA1_KEY_FRAME || hdr.frame_type == EA1_INTRA_ONLY_FRAME;
        info = FrameInfo{};
        info.seq = &seq;
        info.hdr = &hdr;
        info.tab = tab;
        info.mi_rows = int(hdr.mi_rows), info.mi_cols = int(hdr.mi_cols);
        info.mi_stride = mi_stride = info.mi_cols;
        const size_t units = size_t(info.mi_rows) * info.mi_cols;
        if (unit.size() < units) unit.assign(units, 0), utx_size.assign(units, 0), utx_type.assign(units, 0);
        if (++stamp > 255) {   // wrapped: no stale stamp may match
            std::fill(unit.begin(), unit.end(), 0u);
            stamp = 1;
        }
        info.unit = unit.data(), info.utx_size = utx_size.data(), info.utx_type = utx_type.data();
        info.stamp = stamp;
        info.intra_only = intra_only;
        cur_motion.reset();
        info.save_ref = nullptr, info.save_mv = nullptr;
        info.mono = seq.monochrome != 0;
        info.ssx = seq.monochrome ? 1 : int(seq.subsampling_x), info.ssy = seq.monochrome ? 1 : int(seq.subsampling_y);
        info.reduced_tx_set = hdr.reduced_tx_set;
        info.lossless = hdr.coded_lossless;
        info.tx_mode_select = hdr.tx_mode_select;
        info.qidx_class = hdr.base_q_idx <= 20 ? 0 : (hdr.base_q_idx <= 60 ? 1 : (hdr.base_q_idx <= 120 ? 2 : 3));
        // references
        info.order_hint[0] = 0, info.sign_bias[0] = 0, info.ref_slot[0] = 0, info.gm_warp_ok[0] = false;
        for (int r = 1; r <= 7; r++) {
            info.order_hint[r] = 0, info.sign_bias[r] = 0, info.ref_slot[r] = 0, info.gm_warp_ok[r] = false;
            if (intra_only) continue;
            const int slot = hdr.ref_frame_idx[r - 1];
            ea1_av1_slot_info si;
            rc = ea1_av1_parser_slot_info(hp, slot, &si);
            if (rc != EA1_OK) return rc;
            if (!si.valid || phys[slot] == 0xFF) return ea1::fail(EA1_EINVAL, "AV1 block parser: reference %d (slot %d) has never been decoded", r, slot);
            info.order_hint[r] = si.order_hint;
            info.ref_slot[r] = phys[slot];
            if (seq.enable_order_hint) {
                const int bits = seq.order_hint_bits, m = 1 << (bits - 1);
                int diff = int(si.order_hint) - int(hdr.order_hint);
                diff = (diff & (m - 1)) - (diff & m);
                info.sign_bias[r] = diff > 0;
            }
            // is_scaled() (7.11.3.3) compares the reference's UPSCALED width with this frame's (downscaled) width
            info.gm_warp_ok[r] = si.upscaled_width == hdr.frame_width && si.frame_height == hdr.frame_height;
        }
        if (hdr.refresh_frame_flags && seq.enable_ref_frame_mvs) {
            cur_motion = std::make_shared<Motion>();
            Motion *m = cur_motion.get();
            m->rows = info.mi_rows, m->cols = info.mi_cols;
            m->frame_type = hdr.frame_type;
            for (int r = 0; r < 8; r++) m->order_hints[r] = info.order_hint[r];
            const size_t n = size_t(m->cols >> 1) * size_t(m->rows >> 1);
            m->ref.assign(n, REF_NONE);
            m->mv.assign(2 * n, 0);
            if (!intra_only) {
                info.save_ref = m->ref.data(), info.save_mv = m->mv.data();
                for (int r = 0; r < 8; r++) info.save_past[r] = r > INTRA_FRAME && rel_dist(info.order_hint[r], hdr.order_hint) < 0;
            }
        }
        // where the picture goes: a context picture no AV1 slot will point at once this frame's refresh is done
        // (reading a reference and writing the result into the same picture is fine: the last filter stage writes
        // the picture after every prediction has been made)
        int dst = -1;
        for (int pic = 0; pic < EA1_NUM_REF_SLOTS && dst < 0; pic++) {   // (nine pictures: eight slots may hold eight different ones while a ninth is decoded)
            bool used = false;
            for (int s = 0; s < 8; s++)
                if (phys[s] == pic && !(hdr.refresh_frame_flags & (1 << s))) used = true;
            if (!used) dst = pic;
        }
        if (dst < 0) return ea1::fail(EA1_ENOTSUP, "AV1 block parser: all pictures of the reconstruction context stay referenced");
        desc.dst_slot = uint8_t(dst);
        // CDFs
        if (hdr.primary_ref_frame == EA1_PRIMARY_REF_NONE) {
            cur.m = tab->mode;
            cur.c = tab->coef[info.qidx_class];
        } else {
            const int slot = hdr.ref_frame_idx[hdr.primary_ref_frame];
            if (!saved[slot]) return ea1::fail(EA1_EINVAL, "AV1 block parser: primary_ref_frame names slot %d, which holds no contexts", slot);
            cur = *saved[slot];
        }
        tile_saved = cur;
        // segmentation maps
        seg_cur.assign(size_t(info.mi_rows) * info.mi_cols, 0);
        info.seg_ids = seg_cur.data();
        info.prev_seg_ids = nullptr;
        if (hdr.segmentation_enabled && hdr.primary_ref_frame != EA1_PRIMARY_REF_NONE) {
            const int slot = hdr.ref_frame_idx[hdr.primary_ref_frame];
            if (seg_saved_cols[slot] == info.mi_cols && seg_saved_rows[slot] == info.mi_rows && !seg_saved[slot].empty()) {
                seg_prev = seg_saved[slot];
                info.prev_seg_ids = seg_prev.data();
            }
        }
        if (hdr.segmentation_enabled && !hdr.segmentation_update_map && info.prev_seg_ids) seg_cur = seg_prev, info.seg_ids = seg_cur.data();
        for (int sg = 0; sg < 8; sg++) {
            int q = hdr.base_q_idx;
            if (hdr.segmentation_enabled && hdr.feature_enabled[sg][0]) q = std::min(255, std::max(0, q + hdr.feature_data[sg][0]));
            info.seg_lossless[sg] = q == 0 && !hdr.delta_q_y_dc && !hdr.delta_q_u_dc && !hdr.delta_q_u_ac && !hdr.delta_q_v_dc && !hdr.delta_q_v_ac;
        }
        info.mf_src = nullptr;
        info.mf_w8 = info.mi_cols >> 1;
        if (hdr.use_ref_frame_mvs) motion_field_estimation(hdr);
        // output
        out.blocks.clear(), out.txs.clear(), out.coefs.clear(), out.aux.clear();
        out.sb64_cols = int(desc.width) / 64;   // of the padded frame (whole superblocks): the hand-over reads one entry per 64x64 area of it
        out.cdef_idx.assign(size_t(out.sb64_cols) * (desc.height / 64), 0xFF);
        for (int p = 0; p < 3; p++) {
            out.lr[p].clear();
            out.lr_rows[p] = out.lr_cols[p] = 0;
            if (!hdr.frame_restoration_type[p]) continue;
            const int sx = p ? int(seq.subsampling_x) : 0, sy = p ? int(seq.subsampling_y) : 0, usz = hdr.loop_restoration_size[p];
            auto units = [&](int size) { return std::max((size + (usz >> 1)) / usz, 1); };
            out.lr_rows[p] = units((int(hdr.frame_height) + sy) >> sy);
            out.lr_cols[p] = units((int(hdr.upscaled_width) + sx) >> sx);
            ea1_lr_unit none;
            memset(&none, 0, sizeof none);
            out.lr[p].assign(size_t(out.lr_rows[p]) * out.lr_cols[p], none);
        }
        return EA1_OK;
    }

    int rel_dist(int a, int b) const
    {
        if (!seq.enable_order_hint) return 0;
        const int m = 1 << (seq.order_hint_bits - 1);
        const int diff = a - b;
        return (diff & (m - 1)) - (diff & m);
    }
    // 7.9.3 (mv_projection lives in av1_tile_syntax.h: the tiles scale the saved vectors when they use them)
    static int project(int v8, int delta, int dst_sign, int max8, int max_off8, bool *valid)
    {
        const int base8 = (v8 >> 3) << 3;
        const int offset8 = delta >= 0 ? (delta >> 6) : -((-delta) >> 6);
        v8 += dst_sign * offset8;
        if (v8 < 0 || v8 >= max8 || v8 < base8 - max_off8 || v8 >= base8 + 8 + max_off8) *valid = false;
        return v8;
    }
    // 7.9.2: the saved vectors of reference `src` carried along their trajectories into the current frame
    bool motion_projection(const ea1_av1_frame_header &hdr, int src, int dst_sign)
    {
        const int slot = hdr.ref_frame_idx[src - LAST_FRAME];
        const Motion *m = motion[slot].get();
        // (an intra frame -- key or intra-only -- holds no motion: it is not projected from and does not use up a place of the stack;
        // libaom and dav1d skip both kinds)
        if (!m || m->rows != info.mi_rows || m->cols != info.mi_cols || m->frame_type == EA1_INTRA_ONLY_FRAME || m->frame_type == EA1_KEY_FRAME) return false;
        const int w8 = info.mi_cols >> 1, h8 = info.mi_rows >> 1;
        const int ref_to_cur = rel_dist(info.order_hint[src], hdr.order_hint);
        if (std::abs(ref_to_cur) > 31) return true;   // (nothing of it can be projected; it still takes its place in the stack)
        int offset_of[8];   // the saved frame's distance to each of ITS references, 0 = not usable
        for (int r = 0; r < 8; r++) {
            const int o = r > INTRA_FRAME ? rel_dist(info.order_hint[src], m->order_hints[r]) : 0;
            offset_of[r] = (o > 0 && o <= 31) ? o : 0;
        }
        for (int y = 0; y < h8; y++)
            for (int x = 0; x < w8; x++) {
                const size_t at = size_t(y) * w8 + x;     // (saved at 8x8 granularity: the unit at the odd row and column, 7.19)
                const int src_ref = m->ref[at];
                if (src_ref == REF_NONE || src_ref <= INTRA_FRAME) continue;
                const int ref_offset = offset_of[src_ref & 7];
                if (!ref_offset) continue;
                const int16_t *mv = &m->mv[2 * at];
                int16_t proj[2];
                mv_projection(mv, ref_to_cur * dst_sign, ref_offset, proj);
                bool valid = true;
                const int py = project(y, proj[0], dst_sign, h8, 0, &valid), px = project(x, proj[1], dst_sign, w8, 8, &valid);
                if (!valid) continue;
                // MotionFieldMvs[dst][py][px] of 7.9.2 for ALL seven dst is mv_projection(mv, ref_to_dst, ref_offset): the vector and the offset are
                // kept, the tile that looks a position up scales them for ITS reference (add_tpl_ref_mv) -- a later projection onto the same
                // position overwrites an earlier one exactly as the seven stores would
                int16_t *dstp = &mf[3 * (size_t(py) * w8 + px)];
                dstp[0] = mv[0], dstp[1] = mv[1], dstp[2] = int16_t(ref_offset);
            }
        return true;
    }
    // 7.9.1
    void motion_field_estimation(const ea1_av1_frame_header &hdr)
    {
        const int w8 = info.mi_cols >> 1, h8 = info.mi_rows >> 1;
        mf.assign(size_t(3) * h8 * w8, kInvalidMv);
        info.mf_src = mf.data();
        for (int r = 0; r < 8; r++) info.mf_ref_to_dst[r] = int8_t(r >= LAST_FRAME ? rel_dist(hdr.order_hint, info.order_hint[r]) : 0);
        const Motion *last = motion[hdr.ref_frame_idx[0]].get();
        const int last_alt = last ? last->order_hints[ALTREF_FRAME] : -1;
        if (last_alt != info.order_hint[GOLDEN_FRAME]) motion_projection(hdr, LAST_FRAME, -1);
        int ref_stamp = 1;   // MFMV_STACK_SIZE - 2
        if (rel_dist(info.order_hint[BWDREF_FRAME], hdr.order_hint) > 0 && motion_projection(hdr, BWDREF_FRAME, 1)) ref_stamp--;
        if (rel_dist(info.order_hint[ALTREF2_FRAME], hdr.order_hint) > 0 && motion_projection(hdr, ALTREF2_FRAME, 1)) ref_stamp--;
        if (rel_dist(info.order_hint[ALTREF_FRAME], hdr.order_hint) > 0 && ref_stamp >= 0 && motion_projection(hdr, ALTREF_FRAME, 1)) ref_stamp--;
        if (ref_stamp >= 0) motion_projection(hdr, LAST2_FRAME, -1);
    }
    // show_existing_frame (7.21): no tile data; the picture of slot frame_to_show_map_idx is output, and when it is a key frame the
    // reference update process runs with it (every slot then holds that frame: its picture, its contexts, its segment map)
    int show_existing(const ea1_av1_frame_header &hdr, ea1_ir_frame *frame)
    {
        int rc = ea1_av1_parser_sequence_header(hp, &seq);
        if (rc != EA1_OK) return rc;
        const int slot = hdr.frame_to_show_map_idx;
        if (phys[slot] == 0xFF) return ea1::fail(EA1_EINVAL, "AV1 block parser: show_existing_frame of slot %d, which holds no picture", slot);
        if (hdr.frame_type == EA1_KEY_FRAME) {
            for (int s2 = 0; s2 < 8; s2++) {
                if (s2 == slot) continue;
                phys[s2] = phys[slot];
                if (saved[slot]) {
                    if (!saved[s2]) saved[s2].reset(new Cdfs);
                    *saved[s2] = *saved[slot];
                } else {
                    saved[s2].reset();
                }
                seg_saved[s2] = seg_saved[slot];
                seg_saved_cols[s2] = seg_saved_cols[slot], seg_saved_rows[s2] = seg_saved_rows[slot];
                motion[s2] = motion[slot];
            }
        }
        rc = ea1_av1_parser_end_frame(hp, &hdr);
        if (rc != EA1_OK) return rc;
        memset(frame, 0, sizeof *frame);
        const uint32_t cw = hdr.mi_cols ? hdr.mi_cols * 4 : ((hdr.frame_width + 7) & ~7u), chh = hdr.mi_rows ? hdr.mi_rows * 4 : ((hdr.frame_height + 7) & ~7u);
        const uint32_t sbm = seq.use_128x128_superblock ? 127u : 63u;
        frame->desc.width = (cw + sbm) & ~sbm, frame->desc.height = (chh + sbm) & ~sbm;
        frame->desc.pad = uint8_t(((frame->desc.width - cw) / 8) | (((frame->desc.height - chh) / 8) << 4));
        frame->desc.crop = uint8_t((cw - hdr.frame_width) | ((chh - hdr.frame_height) << 4));
        if (hdr.upscaled_width && hdr.upscaled_width != hdr.frame_width) {
            // a frame that was decoded with super-resolution is kept upscaled: upscaled_width visible samples in whole 8-sample units
            frame->desc.width = (hdr.upscaled_width + 7u) & ~7u;
            frame->desc.pad &= 0xF0;
            frame->desc.crop = uint8_t((frame->desc.crop & 0xF0) | (frame->desc.width - hdr.upscaled_width));
        }
        frame->desc.dst_slot = phys[slot];
        return EA1_OK;
    }

    // one tile: IO is bound to the tile's bytes by the caller
    template <class IO>
    int run_tile(IO &io, const ea1_av1_tile &t, bool is_context_tile)
    {
        TileSyntax<IO> ts(io, info, out);
        ts.init_tile(cur, int(t.mi_row_start), int(t.mi_row_end), int(t.mi_col_start), int(t.mi_col_end));
        ts.decode_tile();
        if (ts.err != EA1_OK) return ea1::fail(ts.err, "AV1 block parser, tile %u: %s", t.tile_num, ts.err_text);
        if (is_context_tile) tile_saved = ts.cdf;
        return EA1_OK;
    }

    // ---- tiles side by side (dav1d's per-tile tasks, src/thread_task.c: /root/reference/cmake/dav1d.cmake:992) ----
    // After the per-tile copy of the frame's CDFs the tiles of a frame share nothing they write: contexts, above / left state and the record
    // vectors are the tile's own; the per-position arrays of the frame (mi, segment ids, cdef_idx, restoration units) are written at the tile's
    // own positions only, and every neighbour a tile reads lies inside it (is_inside(), AvailU / AvailL).  Each tile therefore parses into a
    // FrameOut of its own on whatever thread takes it; merge_tiles() then appends them in TILE ORDER and rebases the three kinds of index a
    // record holds (block -> its transform blocks, transform block -> its coefficients, block -> its palette / warp bytes): the result is
    // byte for byte what the tile-after-tile loop above produces (tests/test_av1_block_parse.py, tests/test_real_av1.py).
    std::vector<FrameOut> tile_out;
    std::vector<int> tile_rc;
    std::vector<std::string> tile_msg;
    void begin_tiles(int count)
    {
        if (int(tile_out.size()) < count) tile_out.resize(size_t(count));
        tile_rc.assign(size_t(count), EA1_OK);
        tile_msg.assign(size_t(count), std::string());
        for (int k = 0; k < count; k++) {
            FrameOut &o = tile_out[size_t(k)];
            o.blocks.clear(), o.txs.clear(), o.coefs.clear(), o.aux.clear();
            o.temporal_candidates = 0;
            o.frame = &out;
        }
    }
    // one tile into tile_out[k]; any thread
    template <class IO>
    void run_tile_into(IO &io, const ea1_av1_tile &t, bool is_context_tile, int k)
    {
        TileSyntax<IO> ts(io, info, tile_out[size_t(k)]);
        ts.init_tile(cur, int(t.mi_row_start), int(t.mi_row_end), int(t.mi_col_start), int(t.mi_col_end));
        ts.decode_tile();
        if (ts.err != EA1_OK) {
            tile_rc[size_t(k)] = ts.err;
            char buf[256];
            snprintf(buf, sizeof buf, "AV1 block parser, tile %u: %s", t.tile_num, ts.err_text ? ts.err_text : "");
            tile_msg[size_t(k)] = buf;
            return;
        }
        if (is_context_tile) tile_saved = ts.cdf;   // (one tile of the frame: a single writer)
    }
    int merge_tiles(int count)
    {
        for (int k = 0; k < count; k++)
            if (tile_rc[size_t(k)] != EA1_OK) return ea1::fail(tile_rc[size_t(k)], "%s", tile_msg[size_t(k)].c_str());
        size_t nb = out.blocks.size(), nt = out.txs.size(), nc = out.coefs.size(), na = out.aux.size();
        for (int k = 0; k < count; k++) {
            const FrameOut &o = tile_out[size_t(k)];
            na = (na + 3) & ~size_t(3), nc = (nc + 1) & ~size_t(1);
            nb += o.blocks.size(), nt += o.txs.size(), nc += o.coefs.size(), na += o.aux.size();
        }
        out.blocks.reserve(nb), out.txs.reserve(nt), out.coefs.reserve(nc), out.aux.reserve(na);
        for (int k = 0; k < count; k++) {
            const FrameOut &o = tile_out[size_t(k)];
            while (out.aux.size() & 3) out.aux.push_back(0);        // what init_tile() does to the frame's arenas in the tile-after-tile parse
            if (out.coefs.size() & 1) out.coefs.push_back(0);
            const uint32_t tx_base = uint32_t(out.txs.size()), coef_base = uint32_t(out.coefs.size()), aux_base = uint32_t(out.aux.size());
            for (ea1_block b : o.blocks) {
                b.first_tx += tx_base;
                if ((b.flags & EA1_BLK_WARP) || b.pal_size[0] || b.pal_size[1]) b.aux_off += aux_base;
                out.blocks.push_back(b);
            }
            for (ea1_txblock t : o.txs) {
                if (t.coef_off != EA1_NO_COEFS) t.coef_off += coef_base;
                out.txs.push_back(t);
            }
            out.coefs.append(o.coefs.data(), o.coefs.size());
            out.aux.append(o.aux.data(), o.aux.size());
            out.temporal_candidates += o.temporal_candidates;
        }
        return EA1_OK;
    }

    // the tiles of a frame in order (mi ranges from the header): what a writer iterates over
    static void tile_table(const ea1_av1_frame_header &h, std::vector<ea1_av1_tile> *tiles)
    {
        tiles->clear();
        for (int r = 0; r < h.tile_rows; r++)
            for (int c = 0; c < h.tile_cols; c++) {
                ea1_av1_tile t;
                memset(&t, 0, sizeof t);
                t.tile_num = uint16_t(r * h.tile_cols + c), t.tile_row = uint16_t(r), t.tile_col = uint16_t(c);
                t.mi_row_start = h.mi_row_starts[r], t.mi_row_end = h.mi_row_starts[r + 1];
                t.mi_col_start = h.mi_col_starts[c], t.mi_col_end = h.mi_col_starts[c + 1];
                tiles->push_back(t);
            }
    }

    // 8x8 areas of the padded frame that no block covers (blocks that reach beyond the coded frame are at least 32 samples on that side
    // and aligned: they cover whole 8x8 areas): one filler block each -- intra DC, skipped, no coefficients -- so that the hand-over's
    // blocks tile its frame.  Nothing inside the coded frame ever reads what they predict.
    void add_filler_blocks()
    {
        const int w8 = int(desc.width) / 8, h8 = int(desc.height) / 8;
        // the coded frame is covered by its blocks (a partition tree leaves nothing out): only blocks that reach past it matter, and only
        // the padding has to be searched for open areas
        const int cw8 = (info.mi_cols + 1) >> 1, ch8 = (info.mi_rows + 1) >> 1;
        std::vector<uint8_t> covered(size_t(w8) * h8, 0);
        for (int y8 = 0; y8 < std::min(ch8, h8); y8++) memset(&covered[size_t(y8) * w8], 1, size_t(std::min(cw8, w8)));
        for (const ea1_block &b : out.blocks) {
            const int x8 = b.x4 / 2, y8 = b.y4 / 2, n8w = std::max(1, (1 << b.bwl) / 8), n8h = std::max(1, (1 << b.bhl) / 8);
            if (x8 + n8w <= cw8 && y8 + n8h <= ch8) continue;
            for (int j = 0; j < n8h && y8 + j < h8; j++)
                for (int i = 0; i < n8w && x8 + i < w8; i++) covered[size_t(y8 + j) * w8 + x8 + i] = 1;
        }
        for (int y8 = 0; y8 < h8; y8++)
            for (int x8 = (y8 < ch8 ? std::min(cw8, w8) : 0); x8 < w8; x8++) {
                if (covered[size_t(y8) * w8 + x8]) continue;
                // the largest aligned square (64, 32, 16, 8 samples) that is open as a whole
                int n8 = 8;
                for (; n8 > 1; n8 >>= 1) {
                    if ((x8 & (n8 - 1)) || (y8 & (n8 - 1)) || x8 + n8 > w8 || y8 + n8 > h8) continue;
                    bool open = true;
                    for (int j = 0; j < n8 && open; j++)
                        for (int i = 0; i < n8 && open; i++) open = !covered[size_t(y8 + j) * w8 + x8 + i];
                    if (open) break;
                }
                for (int j = 0; j < n8; j++)
                    for (int i = 0; i < n8; i++) covered[size_t(y8 + j) * w8 + x8 + i] = 1;
                int l2 = 3;
                while ((8 << (l2 - 3)) < 8 * n8) l2++;
                ea1_block b;
                memset(&b, 0, sizeof b);
                b.x4 = uint16_t(2 * x8), b.y4 = uint16_t(2 * y8);
                b.bwl = b.bhl = uint8_t(l2);
                b.flags = uint16_t(EA1_BLK_SKIP | (x8 ? EA1_BLK_HAVE_LEFT : 0) | (y8 ? EA1_BLK_HAVE_TOP : 0));
                b.mode_y = b.mode_uv = EA1_DC_PRED;
                b.first_tx = uint32_t(out.txs.size());
                b.n_tx = 3;
                for (int pl = 0; pl < 3; pl++) {
                    ea1_txblock t;
                    memset(&t, 0, sizeof t);
                    const int sx = pl ? info.ssx : 0, sy = pl ? info.ssy : 0;
                    t.x4 = uint16_t((2 * x8) >> sx), t.y4 = uint16_t((2 * y8) >> sy);
                    t.plane = uint8_t(pl);
                    t.size = uint8_t((l2 - sx) | ((l2 - sy) << 4));
                    t.coef_off = EA1_NO_COEFS;
                    out.txs.push_back(t);
                }
                out.blocks.push_back(b);
            }
    }

    // after the last tile: frame end CDF update, reference update, the finished ea1_ir_frame
    int end_frame(const ea1_av1_frame_header &hdr, ea1_ir_frame *frame)
    {
        if (!hdr.disable_frame_end_update_cdf) {
            cur = tile_saved;
            reset_counters(&cur);
        }
        // AV1 spec 7.4 (decode frame wrap-up): a frame that keeps the previous segmentation map hands exactly that map on -- not the per-block minima its
        // blocks read from it and stored while decoding (found in round 5 beside dav1d: two such frames in a row, generated streams only)
        if (hdr.segmentation_enabled && !hdr.segmentation_update_map && info.prev_seg_ids) seg_cur = seg_prev;
        std::shared_ptr<Motion> saved_motion = std::move(cur_motion);   // (7.19; null when the sequence keeps no motion or no slot is refreshed)
        for (int s = 0; s < 8; s++)
            if (hdr.refresh_frame_flags & (1 << s)) {
                motion[s] = saved_motion;
                if (!saved[s]) saved[s].reset(new Cdfs);
                *saved[s] = cur;
                phys[s] = desc.dst_slot;
                seg_saved[s] = seg_cur;
                seg_saved_cols[s] = info.mi_cols, seg_saved_rows[s] = info.mi_rows;
            }
        int rc = ea1_av1_parser_end_frame(hp, &hdr);
        if (rc != EA1_OK) return rc;
        if (desc.pad) add_filler_blocks();
        desc.num_blocks = uint32_t(out.blocks.size());
        desc.num_tx = uint32_t(out.txs.size());
        desc.num_coefs = uint32_t(out.coefs.size());
        desc.num_aux = uint32_t(out.aux.size());
        desc.coef_layout = EA1_COEFS_DIAGONAL;
        memset(frame, 0, sizeof *frame);
        frame->desc = desc;
        frame->blocks = out.blocks.data();
        frame->txs = out.txs.data();
        frame->coefs = out.coefs.data();
        frame->aux = out.aux.data();
        frame->cdef_idx = out.cdef_idx.data();
        frame->num_sb = uint32_t(out.cdef_idx.size());
        for (int p = 0; p < 3; p++) {
            frame->lr[p] = out.lr[p].empty() ? nullptr : out.lr[p].data();
            frame->lr_count[p] = uint32_t(out.lr[p].size());
        }
        return EA1_OK;
    }
};

// ---- IO of the parser ----
// The specification-literal symbol decoder of msac.cpp (one bit at a time): kept as the reference the fast one is checked against
// (EA1_AV1_SLOW_MSAC=1 makes the parser use it; tests parse every stream both ways).
struct SlowMsacReader {
    ea1_msac s;
    int init(const uint8_t *data, size_t size, int disable_update) { return ea1_msac_init(&s, data, size, disable_update); }
    int sym(Elem, uint16_t *cdf, int n, unsigned = ~0u) { return ea1_msac_symbol(&s, cdf, n); }
    int sym_noadapt(Elem, uint16_t *cdf, int n, unsigned = ~0u)
    {
        const int keep = s.disable_update;
        s.disable_update = 1;
        const int v = ea1_msac_symbol(&s, cdf, n);
        s.disable_update = keep;
        return v;
    }
    int bit(Elem, bool = false) { return ea1_msac_bool(&s); }
    unsigned lit(Elem, int nbits) { return nbits ? ea1_msac_literal(&s, nbits) : 0u; }
};

// The same arithmetic (AV1 specification 8.2.6) on a 64-bit window: the top 16 bits are SymbolValue, below them the bits of the tile
// already fetched -- complemented, as the specification's value register holds them -- and past the end of the tile the stream reads as
// zeros (ones here).  Eight bytes at most per refill instead of a bit at a time.
#if defined(__SSE2__) && !defined(EA1_MSAC_SCALAR) && !defined(__HIP_DEVICE_COMPILE__)
#define EA1_MSAC_SSE2 1
// 4 * (n - symbol - 1) per lane for n symbols, 0 from lane n - 1 on
struct MsacStepTable {
    alignas(16) uint16_t v[17][16];
    constexpr MsacStepTable() : v{}
    {
        for (int n = 0; n <= 16; n++)
            for (int s = 0; s < 16; s++) v[n][s] = uint16_t(s < n - 1 ? 4 * (n - s - 1) : 0);
    }
};
static constexpr MsacStepTable kMsacStepTable{};
#define kMsacStep kMsacStepTable.v
#else
#define EA1_MSAC_SSE2 0
#endif

#define EA1_MSAC_INLINE inline __attribute__((always_inline))

// Round 5: every call site names its alphabet size as a literal, so sym() is forced inline and dispatches at compile time to a decoder written for
// that size -- 2: no loop; 3 / 4 (the coefficient levels: most symbols of a stream): four lanes in the low half of a vector, boundaries picked out
// of a general register; 5 - 8 / 9 - 16: one / two vectors -- with the adaptation FUSED in (the CDF vector is in a register already).  The window
// refills eight bytes at a time.
struct MsacReader {
    const uint8_t *p = nullptr, *end = nullptr;
    uint64_t dif = 0;
    uint32_t rng = 1u << 15;
    int cnt = -15;   // fetched bits below the value field; negative while the value field itself is incomplete
    bool disable_update = false;

    int init(const uint8_t *data, size_t size, int disable)
    {
        if (!data || !size) return ea1::fail(EA1_EINVAL, "AV1 block parser: empty tile");
        p = data, end = data + size, dif = 0, rng = 1u << 15, cnt = -15, disable_update = disable != 0;
        refill();
        return EA1_OK;
    }
    void refill()
    {
        if (end - p >= 8) {
            // as many whole bytes as fit below the fetched bits, in one load
            uint64_t w;
            memcpy(&w, p, 8);
            w = ~__builtin_bswap64(w);
            const int k = (48 - cnt) >> 3;   // 1..7 (cnt <= 40)
            dif |= (w >> (64 - 8 * k)) << (48 - cnt - 8 * k);
            p += k, cnt += 8 * k;
            return;
        }
        while (cnt <= 40) {
            const uint32_t b = p < end ? *p++ : 0u;
            dif |= uint64_t(~b & 0xFFu) << (40 - cnt);
            cnt += 8;
        }
    }
    EA1_MSAC_INLINE void renorm(uint32_t new_rng, uint64_t new_dif)
    {
        const int bits = __builtin_clz(new_rng) - 16;
        rng = new_rng << bits;
        dif = new_dif << bits;
        cnt -= bits;
        if (__builtin_expect(cnt < 0, 0)) refill();
    }
    // the generic decoder (8.2.6, any n): what the specialised ones below are checked against, and the path of a run-time n
    int decode(const uint16_t *cdf, int n)
    {
        const uint32_t v = uint32_t(dif >> 48);
        uint32_t cur = rng, prev;
        int symbol = -1;
        do {
            symbol++;
            prev = cur;
            const uint32_t f = (1u << 15) - cdf[symbol];
            cur = (((rng >> 8) * (f >> 6)) >> 1) + 4u * uint32_t(n - symbol - 1);
        } while (v < cur);
        renorm(prev - cur, dif - (uint64_t(cur) << 48));
        return symbol;
    }
    static void adapt(uint16_t *cdf, int n, int symbol)
    {
        const int fl = 31 - __builtin_clz(uint32_t(n));
        const int rate = 3 + (cdf[n] > 15) + (cdf[n] > 31) + (fl < 2 ? fl : 2);
        uint32_t tmp = 0;
        for (int i = 0; i < n - 1; i++) {
            if (i == symbol) tmp = 1u << 15;
            if (tmp < cdf[i]) cdf[i] = uint16_t(cdf[i] - ((cdf[i] - tmp) >> rate));
            else cdf[i] = uint16_t(cdf[i] + ((tmp - cdf[i]) >> rate));
        }
        cdf[n] = uint16_t(cdf[n] + (cdf[n] < 32));
    }
#if EA1_MSAC_SSE2
    // N = 3, 4: cdf[0 .. 3] in the low half of a vector (for N = 3 the fourth lane is the counter: masked out of the search, rewritten after the store)
    template <int N, bool ADAPT>
    EA1_MSAC_INLINE int sym_small(uint16_t *cdf)
    {
        static_assert(N == 3 || N == 4, "four lanes");
        const uint32_t v = uint32_t(dif >> 48);
        const __m128i k32768 = _mm_set1_epi16(short(0x8000));
        const __m128i c = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(cdf));
        const __m128i f = _mm_slli_epi16(_mm_srli_epi16(_mm_sub_epi16(k32768, c), 6), 7);
        const __m128i b = _mm_add_epi16(_mm_mulhi_epu16(f, _mm_set1_epi16(short(rng & 0xFF00))), _mm_loadl_epi64(reinterpret_cast<const __m128i *>(kMsacStep[N])));
        unsigned mask = unsigned(_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_subs_epu16(b, _mm_set1_epi16(short(v))), _mm_setzero_si128())));   // lanes with boundary <= v
        mask &= (1u << (2 * N)) - 1u;   // (lane N - 1 holds 32768: boundary 0, always reached)
        const int symbol = __builtin_ctz(mask) >> 1;
        const uint64_t bb = uint64_t(_mm_cvtsi128_si64(b));
        const uint32_t cur = uint32_t(bb >> (16 * symbol)) & 0xFFFFu;
        const uint32_t prev = uint32_t(((bb << 16) | rng) >> (16 * symbol)) & 0xFFFFu;   // (rng < 65536)
        if (ADAPT && !disable_update) {
            const unsigned count = cdf[N];
            const int rate = (N == 3 ? 4 : 5) + (count > 15) + (count > 31);
            const __m128i sh = _mm_cvtsi32_si128(rate);
            const __m128i below = _mm_cmpgt_epi16(_mm_set1_epi16(short(symbol)), _mm_set_epi16(7, 6, 5, 4, 3, 2, 1, 0));   // lane < symbol
            const __m128i up = _mm_srl_epi16(_mm_sub_epi16(k32768, c), sh), down = _mm_srl_epi16(c, sh);
            _mm_storel_epi64(reinterpret_cast<__m128i *>(cdf), _mm_sub_epi16(_mm_add_epi16(c, _mm_andnot_si128(below, up)), _mm_and_si128(below, down)));
            cdf[N] = uint16_t(count + (count < 32));   // (N = 3: lane 3 of the store was the counter)
        }
        renorm(prev - cur, dif - (uint64_t(cur) << 48));
        return symbol;
    }
    // N = 5 .. 16: one or two whole vectors; lanes from N on belong to the counter and to whatever follows the array and are put back as they were
    template <int N, bool ADAPT>
    EA1_MSAC_INLINE int sym_wide(uint16_t *cdf)
    {
        static_assert(N >= 5 && N <= 16, "one or two vectors");
        const uint32_t v = uint32_t(dif >> 48);
        const __m128i k32768 = _mm_set1_epi16(short(0x8000)), r8 = _mm_set1_epi16(short(rng & 0xFF00)), vv = _mm_set1_epi16(short(v)), zero = _mm_setzero_si128();
        const __m128i c0 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(cdf));
        const __m128i f0 = _mm_slli_epi16(_mm_srli_epi16(_mm_sub_epi16(k32768, c0), 6), 7);
        const __m128i b0 = _mm_add_epi16(_mm_mulhi_epu16(f0, r8), _mm_load_si128(reinterpret_cast<const __m128i *>(kMsacStep[N])));
        unsigned mask = unsigned(_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_subs_epu16(b0, vv), zero)));
        alignas(16) uint16_t bnd[16];
        _mm_store_si128(reinterpret_cast<__m128i *>(bnd), b0);
        __m128i c1 = zero;
        if (N > 8) {
            c1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(cdf + 8));
            const __m128i f1 = _mm_slli_epi16(_mm_srli_epi16(_mm_sub_epi16(k32768, c1), 6), 7);
            const __m128i b1 = _mm_add_epi16(_mm_mulhi_epu16(f1, r8), _mm_load_si128(reinterpret_cast<const __m128i *>(kMsacStep[N] + 8)));
            mask |= unsigned(_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_subs_epu16(b1, vv), zero))) << 16;
            _mm_store_si128(reinterpret_cast<__m128i *>(bnd + 8), b1);
        }
        if (N < 16) mask &= (1u << (2 * N)) - 1u;
        const int symbol = __builtin_ctz(mask) >> 1;
        const uint32_t cur = bnd[symbol];
        const uint32_t prev = symbol ? bnd[symbol - 1] : rng;
        if (ADAPT && !disable_update) {
            const unsigned count = cdf[N];
            const int rate = 5 + (count > 15) + (count > 31);   // (floor(log2 N) >= 2)
            const __m128i sh = _mm_cvtsi32_si128(rate), sym = _mm_set1_epi16(short(symbol)), lanes = _mm_set_epi16(7, 6, 5, 4, 3, 2, 1, 0);
            {
                const __m128i below = _mm_cmpgt_epi16(sym, lanes);
                const __m128i up = _mm_srl_epi16(_mm_sub_epi16(k32768, c0), sh), down = _mm_srl_epi16(c0, sh);
                __m128i moved = _mm_sub_epi16(_mm_add_epi16(c0, _mm_andnot_si128(below, up)), _mm_and_si128(below, down));
                if (N < 8) {
                    const __m128i live = _mm_cmpgt_epi16(_mm_set1_epi16(short(N - 1)), lanes);   // lane < N - 1
                    moved = _mm_or_si128(_mm_and_si128(live, moved), _mm_andnot_si128(live, c0));
                }
                _mm_storeu_si128(reinterpret_cast<__m128i *>(cdf), moved);
            }
            if (N > 9) {   // (N = 9: the second vector holds 32768, the counter and strangers: nothing of it moves)
                const __m128i lanes1 = _mm_add_epi16(lanes, _mm_set1_epi16(8));
                const __m128i below = _mm_cmpgt_epi16(sym, lanes1);
                const __m128i up = _mm_srl_epi16(_mm_sub_epi16(k32768, c1), sh), down = _mm_srl_epi16(c1, sh);
                __m128i moved = _mm_sub_epi16(_mm_add_epi16(c1, _mm_andnot_si128(below, up)), _mm_and_si128(below, down));
                if (N < 16) {
                    const __m128i live = _mm_cmpgt_epi16(_mm_set1_epi16(short(N - 1)), lanes1);
                    moved = _mm_or_si128(_mm_and_si128(live, moved), _mm_andnot_si128(live, c1));
                }
                _mm_storeu_si128(reinterpret_cast<__m128i *>(cdf + 8), moved);
            }
            cdf[N] = uint16_t(count + (count < 32));
        }
        renorm(prev - cur, dif - (uint64_t(cur) << 48));
        return symbol;
    }
#endif
    // two symbols (most of a stream's elements are flags): the same arithmetic without the loops
    EA1_MSAC_INLINE int decode2(uint32_t f0)   // f0 = 32768 - cdf[0]
    {
        const uint32_t v = uint32_t(dif >> 48);
        const uint32_t cur = (((rng >> 8) * (f0 >> 6)) >> 1) + 4u;
        const int symbol = v < cur;
        renorm(symbol ? cur : rng - cur, symbol ? dif : dif - (uint64_t(cur) << 48));
        return symbol;
    }
    template <bool ADAPT>
    EA1_MSAC_INLINE int sym_n(uint16_t *cdf, int n)
    {
        if (n == 2) {
            const int s = decode2((1u << 15) - cdf[0]);
            if (ADAPT && !disable_update) {
                const int rate = 4 + (cdf[2] > 15) + (cdf[2] > 31);
                if (s) cdf[0] = uint16_t(cdf[0] - (cdf[0] >> rate));
                else cdf[0] = uint16_t(cdf[0] + (((1u << 15) - cdf[0]) >> rate));
                cdf[2] = uint16_t(cdf[2] + (cdf[2] < 32));
            }
            return s;
        }
#if EA1_MSAC_SSE2
        switch (n) {
        case 3: return sym_small<3, ADAPT>(cdf);
        case 4: return sym_small<4, ADAPT>(cdf);
        case 5: return sym_wide<5, ADAPT>(cdf);
        case 6: return sym_wide<6, ADAPT>(cdf);
        case 7: return sym_wide<7, ADAPT>(cdf);
        case 8: return sym_wide<8, ADAPT>(cdf);
        case 9: return sym_wide<9, ADAPT>(cdf);
        case 10: return sym_wide<10, ADAPT>(cdf);
        case 11: return sym_wide<11, ADAPT>(cdf);
        case 12: return sym_wide<12, ADAPT>(cdf);
        case 13: return sym_wide<13, ADAPT>(cdf);
        case 14: return sym_wide<14, ADAPT>(cdf);
        case 15: return sym_wide<15, ADAPT>(cdf);
        case 16: return sym_wide<16, ADAPT>(cdf);
        default: break;
        }
#endif
        const int s = decode(cdf, n);
        if (ADAPT && !disable_update) adapt(cdf, n, s);
        return s;
    }
    EA1_MSAC_INLINE int sym(Elem, uint16_t *cdf, int n, unsigned = ~0u) { return sym_n<true>(cdf, n); }
    EA1_MSAC_INLINE int sym_noadapt(Elem, uint16_t *cdf, int n, unsigned = ~0u) { return sym_n<false>(cdf, n); }
    EA1_MSAC_INLINE int bit(Elem, bool = false) { return decode2(1u << 14); }
    unsigned lit(Elem, int nbits)
    {
        unsigned x = 0;
        for (int i = 0; i < nbits; i++) x = 2 * x + unsigned(bit(E_COUNT));
        return x;
    }
};

}  // namespace av1
}  // namespace ea1
