// Seeded synthetic parsed-frame generator (see include/easyav1_amd_synth.h).  Host-only.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_synth.h"

namespace {

struct Rng {  // xorshift64*
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull)
    {
        if (!s) s = 1;
        for (int i = 0; i < 4; i++) next();
    }
    uint64_t next()
    {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    uint32_t below(uint32_t n) { return uint32_t((next() >> 32) * uint64_t(n) >> 32); }
    double uni() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool chance(double p) { return uni() < p; }
};

struct Gen {
    const ea1_synth_params &prm;
    Rng rng;
    bool inter_frame;
    uint32_t frame_index;
    std::vector<ea1_block> blocks;
    std::vector<ea1_txblock> txs;
    std::vector<int16_t> coefs;
    std::vector<uint8_t> aux;
    std::vector<uint32_t> tile_start;  // in pixels, tile_cols + 1 entries
    uint32_t row_top = 0, row_bottom = 0;  // current tile row, in pixels [row_top, row_bottom)
    std::vector<uint8_t> decoded;      // per 4x4 luma unit: covered by an emitted block
    int w4, h4;

    Gen(const ea1_synth_params &p, uint32_t fi)
        : prm(p), rng(p.seed * 1000003ull + fi), inter_frame(p.gop > 1 ? (fi % p.gop) != 0 : fi != 0), frame_index(fi),
          w4(int(p.width / 4)), h4(int(p.height / 4))
    {
        decoded.assign(size_t(w4) * h4, 0);
    }
    bool has(uint32_t f) const { return (prm.features & f) != 0; }

    // coefficients of one transform block: geometric count of low-frequency positions
    int last_extent = 0;   // max(x, y) over the positions emit_coefs() filled last (what a parser knows from eob + scan)
    int last_diagonal = 0; // max(x + y) over them
    static int corner_log2(int extent) { return extent < 4 ? 2 : (extent < 8 ? 3 : (extent < 16 ? 4 : 5)); }
    uint32_t emit_coefs(int w, int h, bool lossless)
    {
        last_extent = 0;
        last_diagonal = 0;
        const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
        int16_t block[32 * 32];
        memset(block, 0, sizeof block);
        const int area = sw * sh;
        const double mean = 0.12 * area;
        int eob = 1 + int(-std::log(1.0 - rng.uni()) * mean);
        if (eob > area) eob = area;
        // positions by anti-diagonal (x + y), i.e. roughly the coding scan
        int placed = 0;
        for (int d = 0; d < sw + sh - 1 && placed < eob; d++)
            for (int y = 0; y <= d && placed < eob; y++) {
                const int x = d - y;
                if (x >= sw || y >= sh) continue;
                const double u = rng.uni() - 0.5;
                double mag = -20.0 * std::log(1.0 - 2.0 * std::fabs(u));
                if (d == 0) mag *= 4.0;  // DC carries more energy
                long v = std::lround(u < 0 ? -mag : mag);
                if (v > 32767) v = 32767;
                if (v < -32768) v = -32768;
                if (lossless) v /= 4;   // lossless residuals are small integers
                block[y * sw + x] = int16_t(v);
                last_extent = std::max(last_extent, std::max(x, y));
                last_diagonal = std::max(last_diagonal, d);
                placed++;
            }
        if (coefs.size() & 1) coefs.push_back(0);   // offsets stay even in every layout
        const size_t base = coefs.size();
        if (has(EA1_SYN_DIAGONAL_COEFS)) {   // the coded anti-diagonals only (include/easyav1_amd_video.h EA1_COEFS_DIAGONAL)
            for (int d = 0; d <= last_diagonal; d++)
                for (int y = std::max(0, d - (sw - 1)); y <= std::min(d, sh - 1); y++) coefs.push_back(block[y * sw + (d - y)]);
        } else if (has(EA1_SYN_COMPACT_COEFS)) {   // the coded corner only, its width as the row stride
            const int n = 1 << corner_log2(last_extent), cw = std::min(sw, n), ch = std::min(sh, n);
            coefs.resize(base + size_t(cw) * ch);
            for (int y = 0; y < ch; y++)
                for (int x = 0; x < cw; x++) coefs[base + size_t(y) * cw + x] = block[y * sw + x];
        } else {
            coefs.insert(coefs.end(), block, block + area);
        }
        return uint32_t(base);
    }

    uint8_t pick_txtp(int w, int h)
    {
        const int n = w > h ? w : h;
        if (n == 64) return EA1_TXTP(EA1_TX_DCT, EA1_TX_DCT);
        if (n == 32) return rng.chance(0.2) ? EA1_TXTP(EA1_TX_IDENTITY, EA1_TX_IDENTITY) : EA1_TXTP(EA1_TX_DCT, EA1_TX_DCT);
        if (has(EA1_SYN_ALL_TXTP)) return uint8_t(EA1_TXTP(rng.below(4), rng.below(4)));
        // {DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, IDTX}
        switch (rng.below(5)) {
        case 0: return EA1_TXTP(EA1_TX_DCT, EA1_TX_DCT);
        case 1: return EA1_TXTP(EA1_TX_ADST, EA1_TX_DCT);
        case 2: return EA1_TXTP(EA1_TX_DCT, EA1_TX_ADST);
        case 3: return EA1_TXTP(EA1_TX_ADST, EA1_TX_ADST);
        default: return EA1_TXTP(EA1_TX_IDENTITY, EA1_TX_IDENTITY);
        }
    }

    uint8_t pick_intra_mode()
    {
        static const uint8_t base[4] = {EA1_DC_PRED, EA1_V_PRED, EA1_H_PRED, EA1_PAETH_PRED};
        static const uint8_t smooth[3] = {EA1_SMOOTH_PRED, EA1_SMOOTH_V_PRED, EA1_SMOOTH_H_PRED};
        static const uint8_t dirs[6] = {EA1_D45_PRED, EA1_D135_PRED, EA1_D113_PRED, EA1_D157_PRED, EA1_D203_PRED, EA1_D67_PRED};
        const int nb = 4, ns = has(EA1_SYN_SMOOTH) ? 3 : 0, nd = has(EA1_SYN_DIRECTIONAL) ? 6 : 0;
        const int k = int(rng.below(uint32_t(nb + ns + nd)));
        if (k < nb) return base[k];
        if (k < nb + ns) return smooth[k - nb];
        return dirs[k - nb - ns];
    }
    static bool is_dir(int m) { return m >= EA1_V_PRED && m <= EA1_D67_PRED; }

    // one step down the transform-size ladder (AV1 sub-transform map)
    static void sub_tx(int &wl, int &hl)
    {
        if (wl == hl) {
            if (wl > 2) wl--, hl--;
        } else if (wl > hl) {
            wl--;
        } else {
            hl--;
        }
    }

    void emit_tx(int pl, int x, int y, int wl, int hl)
    {
        ea1_txblock t;
        memset(&t, 0, sizeof t);
        t.x4 = uint16_t(x / 4);
        t.y4 = uint16_t(y / 4);
        t.plane = uint8_t(pl);
        t.size = uint8_t(wl | (hl << 4));
        t.txtp = has(EA1_SYN_LOSSLESS) ? uint8_t(EA1_TXTP_WHT) : pick_txtp(1 << wl, 1 << hl);
        t.coef_off = rng.chance(0.3) ? EA1_NO_COEFS : emit_coefs(1 << wl, 1 << hl, has(EA1_SYN_LOSSLESS));
        // the corner bound a parser derives from the end-of-block position (EA1_SYN_NO_NZ_HINT leaves it to the hand-over)
        if (t.coef_off != EA1_NO_COEFS && has(EA1_SYN_DIAGONAL_COEFS)) t.nz_log2 = EA1_NZ_DIAGONAL(corner_log2(last_extent), last_diagonal);
        else if (t.coef_off != EA1_NO_COEFS && (!has(EA1_SYN_NO_NZ_HINT) || has(EA1_SYN_COMPACT_COEFS))) t.nz_log2 = uint8_t(corner_log2(last_extent));
        txs.push_back(t);
    }

    bool unit_decoded(int x, int y) const { return decoded[size_t(y / 4) * w4 + x / 4] != 0; }

    // a block vector whose source -- with the extra column / row a half-sample chroma position
    // reads -- lies inside the frame and in already decoded units
    bool pick_block_vector(int x, int y, int w, int h, ea1_block &b)
    {
        const int W = int(prm.width), H = int(prm.height);
        for (int attempt = 0; attempt < 24; attempt++) {
            const int dx = int(rng.below(257)) - 160, dy = int(rng.below(161)) - 128;
            if (!dx && !dy) continue;
            // luma rectangle, widened to what the chroma planes read (x0 / 2 + floor(dx / 2) .. + w / 2 [+ 1])
            const int cx0 = (x / 2 * 2 + dx) >> 1, cy0 = (y / 2 * 2 + dy) >> 1;   // floor: arithmetic shift
            const int cx1 = cx0 + w / 2 - 1 + (dx & 1), cy1 = cy0 + h / 2 - 1 + (dy & 1);
            const int lx0 = std::min(x + dx, 2 * cx0), ly0 = std::min(y + dy, 2 * cy0);
            const int lx1 = std::max(x + dx + w - 1, 2 * cx1 + 1), ly1 = std::max(y + dy + h - 1, 2 * cy1 + 1);
            if (lx0 < 0 || ly0 < 0 || lx1 >= W || ly1 >= H) continue;
            bool ok = true;
            for (int yy = ly0 & ~3; ok && yy <= ly1; yy += 4)
                for (int xx = lx0 & ~3; xx <= lx1; xx += 4)
                    if (!unit_decoded(xx, yy)) {
                        ok = false;
                        break;
                    }
            if (!ok) continue;
            b.mv[0][0] = int16_t(dy * 8);
            b.mv[0][1] = int16_t(dx * 8);
            return true;
        }
        return false;
    }

    void emit_block(int x, int y, int wl, int hl, int tile)
    {
        ea1_block b;
        memset(&b, 0, sizeof b);
        b.x4 = uint16_t(x / 4);
        b.y4 = uint16_t(y / 4);
        b.bwl = uint8_t(wl);
        b.bhl = uint8_t(hl);
        const int w = 1 << wl, h = 1 << hl;
        const bool have_left = uint32_t(x) > tile_start[tile], have_top = uint32_t(y) > row_top;
        if (have_left) b.flags |= EA1_BLK_HAVE_LEFT;
        if (have_top) b.flags |= EA1_BLK_HAVE_TOP;
        // above-right / below-left: every 4x4 unit of the ew (eh) samples next to the edge is decoded
        auto top_right = [&](int ex, int ey, int ew) {
            if (uint32_t(ey) <= row_top || uint32_t(ex + ew) >= tile_start[tile + 1]) return false;
            for (int k = 0; k < ew && uint32_t(ex + ew + k) < tile_start[tile + 1]; k += 4)
                if (!unit_decoded(ex + ew + k, ey - 4)) return false;
            return true;
        };
        auto bottom_left = [&](int ex, int ey, int eh) {
            if (uint32_t(ex) <= tile_start[tile] || uint32_t(ey + eh) >= row_bottom) return false;
            for (int k = 0; k < eh && uint32_t(ey + eh + k) < row_bottom; k += 4)
                if (!unit_decoded(ex - 4, ey + eh + k)) return false;
            return true;
        };
        if (top_right(x, y, w)) b.flags |= EA1_BLK_HAVE_TOP_RIGHT;
        if (bottom_left(x, y, h)) b.flags |= EA1_BLK_HAVE_BOTTOM_LEFT;
        const bool sub8 = w == 4 || h == 4;
        const bool has_chroma = ((b.x4 & 1) || w > 4) && ((b.y4 & 1) || h > 4);
        const int rx = x & ~7, ry = y & ~7, rw = w < 8 ? 8 : w, rh = h < 8 ? 8 : h;   // luma area of the chroma block
        if (sub8 && has_chroma) {
            if (uint32_t(rx) > tile_start[tile]) b.flags |= EA1_BLK_CHROMA_HAVE_LEFT;
            if (uint32_t(ry) > row_top) b.flags |= EA1_BLK_CHROMA_HAVE_TOP;
            if (top_right(rx, ry, rw)) b.flags |= EA1_BLK_CHROMA_HAVE_TOP_RIGHT;
            if (bottom_left(rx, ry, rh)) b.flags |= EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT;
        }
        bool inter = false;
        if (inter_frame && rng.chance(0.9)) {
            inter = true;
            b.flags |= EA1_BLK_INTER;
            const bool compound = !sub8 && rng.chance(0.3);
            if (compound) b.flags |= EA1_BLK_COMPOUND;
            // references: the two most recently stored slots
            const uint32_t in_gop = prm.gop > 1 ? frame_index % prm.gop : frame_index;
            const uint32_t newest = (frame_index - 1) % 8;   // (the synthetic stream rotates over AV1's eight slots; the context has a ninth picture)
            const uint32_t older = in_gop >= 2 ? (frame_index - 2) % 8 : newest;
            b.ref[0] = uint8_t(rng.chance(0.5) ? newest : older);
            b.ref[1] = uint8_t(b.ref[0] == newest ? older : newest);
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 2; c++) {
                    b.mv[r][c] = int16_t(int(rng.below(1025)) - 512);  // +-64 px, 1/8 pel
                    // diagnostic (tools only, never set by tests or bench.py): EA1_SYN_MV_SHIFT=k divides the vectors by 2^k --
                    // how much of the inter prediction time is the scatter of independent random vectors
                    static const int mv_shift = getenv("EA1_SYN_MV_SHIFT") ? atoi(getenv("EA1_SYN_MV_SHIFT")) : 0;
                    if (mv_shift) b.mv[r][c] = int16_t(b.mv[r][c] >> mv_shift);
                }
            const uint32_t nf = has(EA1_SYN_BILINEAR) ? 4 : 3;
            b.mode_y = uint8_t(rng.below(nf) | (rng.below(nf) << 2));
            const bool wedge_ok = wl <= 5 && hl <= 5 && !sub8;
            if (compound && has(EA1_SYN_COMPOUND)) {
                b.comp_type = uint8_t(rng.below(4));
                if (b.comp_type == EA1_COMP_WEDGE && !wedge_ok) b.comp_type = EA1_COMP_DIFFWTD;
                static const uint8_t dist_w[8] = {9, 11, 12, 13, 7, 5, 4, 3};
                if (b.comp_type == EA1_COMP_DIST) b.comp_arg = dist_w[rng.below(8)];
                else if (b.comp_type == EA1_COMP_WEDGE) b.comp_arg = uint8_t(rng.below(16) | (rng.below(2) << 4));
                else if (b.comp_type == EA1_COMP_DIFFWTD) b.comp_arg = uint8_t(rng.below(2));
            }
            if (has(EA1_SYN_WARP) && !sub8 && rng.chance(0.3)) {
                // a near-identity affine map around the block centre plus a displacement of up to
                // +-32 samples; one in eight matrices is strong enough to fail the shear test
                b.flags |= EA1_BLK_WARP;
                b.warp_refs = uint8_t(compound ? 1 + rng.below(3) : 1);
                while (aux.size() & 3) aux.push_back(0);
                b.aux_off = uint32_t(aux.size());
                for (int r = 0; r < 2; r++) {
                    if (!(b.warp_refs & (1 << r))) continue;
                    const int amp = rng.chance(0.125) ? 14000 : 4096;
                    int32_t mat[6];
                    mat[2] = 65536 + int(rng.below(2 * amp + 1)) - amp;
                    mat[3] = int(rng.below(2 * amp + 1)) - amp;
                    mat[4] = int(rng.below(2 * amp + 1)) - amp;
                    mat[5] = 65536 + int(rng.below(2 * amp + 1)) - amp;
                    const int64_t cx = x + w / 2, cy = y + h / 2;
                    const int64_t dx = int64_t(rng.below(64 * 256 + 1)) - 32 * 256, dy = int64_t(rng.below(64 * 256 + 1)) - 32 * 256;  // 1/256 sample
                    mat[0] = int32_t(cx * 65536 + dx * 256 - (int64_t(mat[2]) * cx + int64_t(mat[3]) * cy));
                    mat[1] = int32_t(cy * 65536 + dy * 256 - (int64_t(mat[4]) * cx + int64_t(mat[5]) * cy));
                    for (int k = 0; k < 6; k++)
                        for (int s8 = 0; s8 < 32; s8 += 8) aux.push_back(uint8_t(uint32_t(mat[k]) >> s8));
                }
            }
            if (!compound && !sub8 && has(EA1_SYN_OBMC) && rng.chance(0.3)) {
                b.flags |= EA1_BLK_OBMC;
            } else if (!compound && has(EA1_SYN_INTERINTRA) && wedge_ok && rng.chance(0.3)) {
                b.flags |= EA1_BLK_INTERINTRA;
                b.ii_mode = uint8_t(rng.below(4));
                if (rng.chance(0.4)) {
                    b.ii_mode |= EA1_II_WEDGE;
                    b.ii_arg = uint8_t(rng.below(16));
                }
            }
        } else if (!inter_frame && has(EA1_SYN_INTRABC) && !sub8 && rng.chance(0.35) && pick_block_vector(x, y, w, h, b)) {
            b.mode_y = b.mode_uv = EA1_INTRABC_PRED;
        } else {
            b.mode_y = pick_intra_mode();
            b.mode_uv = pick_intra_mode();
            if (is_dir(b.mode_y) && has(EA1_SYN_DIRECTIONAL)) b.angle_delta[0] = int8_t(int(rng.below(7)) - 3);
            if (is_dir(b.mode_uv) && has(EA1_SYN_DIRECTIONAL)) b.angle_delta[1] = int8_t(int(rng.below(7)) - 3);
            if (has(EA1_SYN_CFL) && wl <= 5 && hl <= 5 && rng.chance(0.25)) {
                b.mode_uv = EA1_UV_CFL_PRED;
                b.angle_delta[1] = 0;
                b.cfl_alpha[0] = int8_t(int(rng.below(33)) - 16);
                b.cfl_alpha[1] = int8_t(int(rng.below(33)) - 16);
            }
            if (has(EA1_SYN_FILTER_INTRA) && wl <= 5 && hl <= 5 && rng.chance(0.15)) {
                b.mode_y = EA1_DC_PRED;
                b.angle_delta[0] = 0;
                b.filter_intra = uint8_t(1 + rng.below(5));
            } else if (has(EA1_SYN_PALETTE) && !sub8 && rng.chance(0.15)) {
                b.mode_y = EA1_DC_PRED;
                b.angle_delta[0] = 0;
                b.flags |= EA1_BLK_PALETTE_Y;
                b.pal_size[0] = uint8_t(2 + rng.below(7));
                if (b.mode_uv != EA1_UV_CFL_PRED && rng.chance(0.5)) {
                    b.mode_uv = EA1_DC_PRED;
                    b.angle_delta[1] = 0;
                    b.flags |= EA1_BLK_PALETTE_UV;
                    b.pal_size[1] = uint8_t(2 + rng.below(7));
                }
                b.aux_off = uint32_t(aux.size());
                for (int i = 0; i < b.pal_size[0] + 2 * b.pal_size[1]; i++) aux.push_back(uint8_t(rng.below(256)));
                for (int i = 0; i < w * h; i++) aux.push_back(uint8_t(rng.below(b.pal_size[0])));
                if (b.pal_size[1])
                    for (int i = 0; i < w * h / 4; i++) aux.push_back(uint8_t(rng.below(b.pal_size[1])));
            }
        }
        // transform blocks: luma grid (largest size, or one / two steps below it), then chroma at
        // its largest size
        b.first_tx = uint32_t(txs.size());
        int twl = wl, thl = hl;  // <= 6 already
        if (has(EA1_SYN_LOSSLESS)) {
            twl = thl = 2;
        } else if (has(EA1_SYN_TX_SPLIT)) {
            const double u = rng.uni();
            const int depth = u < 0.6 ? 0 : (u < 0.85 ? 1 : 2);
            for (int k = 0; k < depth; k++) sub_tx(twl, thl);
        }
        for (int ty = 0; ty < h; ty += 1 << thl)
            for (int tx = 0; tx < w; tx += 1 << twl) emit_tx(0, x + tx, y + ty, twl, thl);
        int cwl = wl - 1 > 5 ? 5 : wl - 1, chl = hl - 1 > 5 ? 5 : hl - 1;
        if (has(EA1_SYN_LOSSLESS)) cwl = chl = 2;
        if (cwl < 2) cwl = 2;   // chroma block of a 4-wide / 4-high luma block: 4 samples, from the 8-aligned position
        if (chl < 2) chl = 2;
        for (int pl = 1; pl < 3 && has_chroma; pl++)
            for (int ty = 0; ty < rh / 2; ty += 1 << chl)
                for (int tx = 0; tx < rw / 2; tx += 1 << cwl) emit_tx(pl, rx / 2 + tx, ry / 2 + ty, cwl, chl);
        b.n_tx = uint16_t(txs.size() - b.first_tx);
        bool any = false;
        for (uint32_t k = b.first_tx; k < txs.size(); k++) any |= txs[k].coef_off != EA1_NO_COEFS;
        if (!any) b.flags |= EA1_BLK_SKIP;
        if (has(EA1_SYN_LF_DELTA)) {
            // frame levels (20 / 20 / 12 / 12) plus a per-block delta; one block in eight has level 0
            static const int base[4] = {20, 20, 12, 12};
            const int delta = int(rng.below(17)) - 8;
            const bool zero = rng.chance(0.125);
            for (int k = 0; k < 4; k++) {
                int v = base[k] + delta + (k >= 2 ? int(rng.below(5)) - 2 : 0);
                b.lf_level[k] = uint8_t(zero ? 0 : (v < 0 ? 0 : (v > 63 ? 63 : v)));
            }
        }
        (void)inter;
        blocks.push_back(b);
        for (int j = 0; j < h / 4; j++)
            for (int i = 0; i < w / 4; i++) decoded[size_t(y / 4 + j) * w4 + x / 4 + i] = 1;
    }

    void partition(int x, int y, int log2size, int tile)
    {
        const int n = 1 << log2size;
        if (uint32_t(x) >= prm.width || uint32_t(y) >= prm.height) return;
        const bool fits = uint32_t(x + n) <= prm.width && uint32_t(y + n) <= prm.height;
        const int hsz = n / 2, l = log2size;
        enum { NONE, SPLIT, HORZ, VERT, HORZ_A, HORZ_B, VERT_A, VERT_B, HORZ_4, VERT_4 };
        int kind = NONE;
        const int min_l2 = has(EA1_SYN_SUB8) ? 2 : 3;   // smallest block side
        if (log2size > 3) {
            if (!fits) {
                kind = SPLIT;
            } else if (!has(EA1_SYN_RECT)) {
                kind = rng.chance(0.5) ? SPLIT : NONE;
            } else {
                const double u = rng.uni();
                if (u < 0.45) kind = SPLIT;
                else if (u < 0.70) kind = NONE;
                else if (u < 0.77) kind = HORZ;
                else if (u < 0.84) kind = VERT;
                else if (u < 0.865) kind = HORZ_A;
                else if (u < 0.89) kind = HORZ_B;
                else if (u < 0.915) kind = VERT_A;
                else if (u < 0.94) kind = VERT_B;
                else if (u < 0.97) kind = log2size - 2 >= min_l2 ? HORZ_4 : NONE;
                else kind = log2size - 2 >= min_l2 ? VERT_4 : NONE;
            }
        } else if (min_l2 == 2) {   // 8x8 node: 8x8, four 4x4, two 8x4 or two 4x8
            const double u = rng.uni();
            kind = u < 0.4 ? NONE : (u < 0.6 ? SPLIT : (u < 0.8 ? HORZ : VERT));
        }
        switch (kind) {
        case SPLIT:
            if (l == 3) {   // four 4x4 blocks (no further partitioning)
                emit_block(x, y, 2, 2, tile);
                emit_block(x + 4, y, 2, 2, tile);
                emit_block(x, y + 4, 2, 2, tile);
                emit_block(x + 4, y + 4, 2, 2, tile);
                break;
            }
            partition(x, y, l - 1, tile);
            partition(x + hsz, y, l - 1, tile);
            partition(x, y + hsz, l - 1, tile);
            partition(x + hsz, y + hsz, l - 1, tile);
            break;
        case HORZ:
            emit_block(x, y, l, l - 1, tile);
            emit_block(x, y + hsz, l, l - 1, tile);
            break;
        case VERT:
            emit_block(x, y, l - 1, l, tile);
            emit_block(x + hsz, y, l - 1, l, tile);
            break;
        case HORZ_A:
            emit_block(x, y, l - 1, l - 1, tile);
            emit_block(x + hsz, y, l - 1, l - 1, tile);
            emit_block(x, y + hsz, l, l - 1, tile);
            break;
        case HORZ_B:
            emit_block(x, y, l, l - 1, tile);
            emit_block(x, y + hsz, l - 1, l - 1, tile);
            emit_block(x + hsz, y + hsz, l - 1, l - 1, tile);
            break;
        case VERT_A:
            emit_block(x, y, l - 1, l - 1, tile);
            emit_block(x, y + hsz, l - 1, l - 1, tile);
            emit_block(x + hsz, y, l - 1, l, tile);
            break;
        case VERT_B:
            emit_block(x, y, l - 1, l, tile);
            emit_block(x + hsz, y, l - 1, l - 1, tile);
            emit_block(x + hsz, y + hsz, l - 1, l - 1, tile);
            break;
        case HORZ_4:
            for (int k = 0; k < 4; k++) emit_block(x, y + k * (n / 4), l, l - 2, tile);
            break;
        case VERT_4:
            for (int k = 0; k < 4; k++) emit_block(x + k * (n / 4), y, l - 2, l, tile);
            break;
        default: emit_block(x, y, l, l, tile); break;
        }
    }
};

int units(int unit, int size)
{
    int n = (size + (unit >> 1)) / unit;
    return n < 1 ? 1 : n;
}

template <class T>
T *dup_vec(const std::vector<T> &v)
{
    T *p = static_cast<T *>(malloc(sizeof(T) * (v.size() + 1)));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

}  // namespace

extern "C" void ea1_synth_default_params(ea1_synth_params *p, uint32_t width, uint32_t height, uint32_t tile_cols,
                                         uint64_t seed)
{
    memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    p->tile_cols = tile_cols ? tile_cols : 1;
    p->tile_rows = 1;
    p->gop = 60;
    p->seed = seed;
    p->features = EA1_SYN_DIAGONAL_COEFS;   // the graded tool set of SURVEY.md 8(d), coefficients as a parser hands them over
    p->lr_unit_log2 = width > 1920 ? 7 : 6;
    p->enable_deblock = p->enable_cdef = p->enable_lr = 1;
}

extern "C" int ea1_synth_frame_create(const ea1_synth_params *p, uint32_t frame_index, ea1_synth_frame **out)
{
    if (!p || !out) return ea1::fail(EA1_EINVAL, "ea1_synth_frame_create: NULL argument");
    *out = nullptr;
    if (!p->width || !p->height || (p->width & 7) || (p->height & 7) || p->width > 16384 || p->height > 16384)
        return ea1::fail(EA1_EINVAL, "ea1_synth_frame_create: size must be a non-zero multiple of 8");
    if (p->lr_unit_log2 < 6 || p->lr_unit_log2 > 8) return ea1::fail(EA1_EINVAL, "ea1_synth_frame_create: lr_unit_log2");
    Gen g(*p, frame_index);
    const int sb_cols = (p->width + 63) / 64, sb_rows = (p->height + 63) / 64;
    // uniform tile columns (widths in superblocks, last one takes the remainder)
    uint32_t tc = p->tile_cols ? p->tile_cols : 1;
    int l2 = 0;
    while ((1u << l2) < tc) l2++;
    const int tile_w_sb = (sb_cols + (1 << l2) - 1) >> l2;
    for (int sx = 0; sx < sb_cols; sx += tile_w_sb) g.tile_start.push_back(uint32_t(sx) * 64);
    g.tile_start.push_back(p->width);
    const int ntiles = int(g.tile_start.size()) - 1;
    // uniform tile rows (heights in superblocks), tiles in raster order
    uint32_t tr = p->tile_rows ? p->tile_rows : 1;
    int l2r = 0;
    while ((1u << l2r) < tr) l2r++;
    const int tile_h_sb = (sb_rows + (1 << l2r) - 1) >> l2r;
    for (int sy0 = 0; sy0 < sb_rows; sy0 += tile_h_sb) {
        const int sy1 = std::min(sy0 + tile_h_sb, sb_rows);
        g.row_top = uint32_t(sy0) * 64;
        g.row_bottom = std::min(uint32_t(sy1) * 64, p->height);
        for (int t = 0; t < ntiles; t++)
            for (int sy = sy0; sy < sy1; sy++)
                for (uint32_t x = g.tile_start[t]; x < g.tile_start[t + 1]; x += 64) g.partition(int(x), sy * 64, 6, t);
    }

    auto *f = new (std::nothrow) ea1_synth_frame();
    if (!f) return ea1::fail(EA1_ENOMEM, "out of memory");
    memset(f, 0, sizeof *f);
    ea1_frame_desc &d = f->desc;
    d.width = p->width;
    d.height = p->height;
    d.is_inter = g.inter_frame;
    d.dst_slot = uint8_t(frame_index % 8);
    // lossless frames carry no in-loop filtering, nor do intra frames that allow intra block copy
    const bool no_filters = (p->features & EA1_SYN_LOSSLESS) != 0 || ((p->features & EA1_SYN_INTRABC) && !g.inter_frame);
    if (p->enable_deblock && !no_filters) {
        d.lf_level[0] = d.lf_level[1] = 20;
        d.lf_level[2] = d.lf_level[3] = 12;
    }
    d.lf_sharpness = 0;
    d.cdef_enable = p->enable_cdef && !no_filters;
    d.cdef_damping = 5;
    d.intra_edge_filter = (p->features & EA1_SYN_DIRECTIONAL) ? 1 : 0;
    d.crop = uint8_t((p->crop_right & 7) | ((p->crop_bottom & 7) << 4));
    d.coef_layout = (p->features & EA1_SYN_DIAGONAL_COEFS) ? EA1_COEFS_DIAGONAL : ((p->features & EA1_SYN_COMPACT_COEFS) ? EA1_COEFS_CORNER : EA1_COEFS_DENSE);
    d.lf_per_block = (p->features & EA1_SYN_LF_DELTA) && p->enable_deblock && !no_filters ? 1 : 0;
    Rng prng(p->seed * 7919ull + frame_index * 31ull + 5);
    for (int i = 0; i < 8; i++) {
        d.cdef_y_strength[i] = uint8_t((prng.below(16) << 2) | prng.below(4));
        d.cdef_uv_strength[i] = uint8_t((prng.below(16) << 2) | prng.below(4));
    }
    d.num_blocks = uint32_t(g.blocks.size());
    d.num_coefs = uint32_t(g.coefs.size());
    d.num_tx = uint32_t(g.txs.size());
    d.num_aux = uint32_t(g.aux.size());
    f->blocks = dup_vec(g.blocks);
    f->txs = dup_vec(g.txs);
    f->coefs = dup_vec(g.coefs);
    f->aux = dup_vec(g.aux);
    f->num_sb = uint32_t(sb_cols * sb_rows);
    f->cdef_idx = static_cast<uint8_t *>(malloc(f->num_sb));
    for (uint32_t i = 0; i < f->num_sb; i++) f->cdef_idx[i] = uint8_t(prng.below(8));
    for (int pl = 0; pl < 3; pl++) {
        const int ss = pl ? 1 : 0;
        d.lr_type[pl] = (p->enable_lr && !no_filters) ? 1 : 0;
        d.lr_unit_log2[pl] = uint8_t(p->lr_unit_log2 - ss);
        const int usz = 1 << d.lr_unit_log2[pl];
        // unit counts follow the VISIBLE plane (AV1 spec 5.9.20: unitSize against Round2(FrameWidth / Height, ss))
        const int vis_w = int(p->width) - (p->crop_right & 7), vis_h = int(p->height) - (p->crop_bottom & 7);
        const int cnt = units(usz, (vis_h + ss) >> ss) * units(usz, (vis_w + ss) >> ss);
        f->lr_count[pl] = uint32_t(cnt);
        f->lr[pl] = static_cast<ea1_lr_unit *>(calloc(size_t(cnt), sizeof(ea1_lr_unit)));
        for (int i = 0; i < cnt; i++) {
            ea1_lr_unit &u = f->lr[pl][i];
            int type = pl == 0 ? EA1_LR_WIENER : EA1_LR_SGRPROJ, set = 5;
            if (p->features & EA1_SYN_LR_MIX) {
                type = int(prng.below(3));   // EA1_LR_NONE / WIENER / SGRPROJ
                set = int(prng.below(16));
            }
            u.type = uint8_t(type);
            if (type == EA1_LR_WIENER) {
                // legal tap ranges of the bitstream: c0 in [-5,10], c1 in [-23,8], c2 in [-17,46];
                // chroma filters are 5-tap (c0 = 0)
                for (int k = 0; k < 2; k++) {
                    u.coef[k][0] = pl ? int8_t(0) : int8_t(int(prng.below(16)) - 5);
                    u.coef[k][1] = int8_t(int(prng.below(32)) - 23);
                    u.coef[k][2] = int8_t(int(prng.below(64)) - 17);
                }
            } else if (type == EA1_LR_SGRPROJ) {
                u.sgr_set = uint8_t(set);
                // xqd ranges: [-96, 31] and [-32, 95]; sets with one radius 0 carry only one weight
                u.coef[0][0] = int8_t(int(prng.below(128)) - 96);
                u.coef[0][1] = int8_t(int(prng.below(128)) - 32);
                if (set >= 14) {          // r0 == 0: w0 = 0
                    u.coef[0][0] = 0;
                } else if (set >= 10) {   // r1 == 0: w1 = 128 - w0 clipped to [-32, 95]
                    int w1 = 128 - u.coef[0][0];
                    u.coef[0][1] = int8_t(w1 < -32 ? -32 : (w1 > 95 ? 95 : w1));
                }
            }
        }
    }
    *out = f;
    return EA1_OK;
}

extern "C" void ea1_synth_frame_destroy(ea1_synth_frame *f)
{
    if (!f) return;
    free(f->blocks);
    free(f->txs);
    free(f->coefs);
    free(f->aux);
    free(f->cdef_idx);
    for (int pl = 0; pl < 3; pl++) free(f->lr[pl]);
    delete f;
}

extern "C" void ea1_synth_fill_picture(uint32_t width, uint32_t height, uint64_t seed, uint8_t *y, uint8_t *u, uint8_t *v)
{
    Rng rng(seed ^ 0xA5A5A5A5ull);
    uint8_t *planes[3] = {y, u, v};
    for (int pl = 0; pl < 3; pl++) {
        const uint32_t w = pl ? width / 2 : width, h = pl ? height / 2 : height;
        const int gx = int(rng.below(5)) - 2, gy = int(rng.below(5)) - 2, base = 64 + int(rng.below(128));
        for (uint32_t r = 0; r < h; r++)
            for (uint32_t c = 0; c < w; c++) {
                int val = base + (gx * int(c)) / 16 + (gy * int(r)) / 16 + int(rng.below(33)) - 16;
                planes[pl][size_t(r) * w + c] = uint8_t(val < 0 ? 0 : (val > 255 ? 255 : val));
            }
    }
}

static_assert(sizeof(ea1_block) == 48, "ea1_block layout");
static_assert(sizeof(ea1_txblock) == 12, "ea1_txblock layout");
static_assert(sizeof(ea1_lr_unit) == 8, "ea1_lr_unit layout");
static_assert(sizeof(ea1_frame_desc) == 68, "ea1_frame_desc layout");

// Stand-in warp filter table: see easyav1_amd_synth.h.  Row 64 * (ip + 1) + k interpolates at
// position ip + k / 64 (ip = -1, 0, 1) between the samples at tap indices 3 + ip and 4 + ip.
extern "C" void ea1_synth_warp_filter(int8_t filter[193 * 8])
{
    memset(filter, 0, 193 * 8);
    auto rdiv = [](int64_t n) { return int((n + 2048) >> 12); };   // round(n / 4096), arithmetic shift
    for (int row = 0; row < 193; row++) {
        const int ip = row == 192 ? 1 : row / 64 - 1, f = row == 192 ? 64 : row % 64;
        const int64_t f2 = int64_t(f) * f, f3 = f2 * f;
        int w[4] = {rdiv(-f3 + 128 * f2 - 4096 * f), rdiv(3 * f3 - 320 * f2 + 524288), rdiv(-3 * f3 + 256 * f2 + 4096 * f), rdiv(f3 - 64 * f2)};
        int &big = w[1] >= w[2] ? w[1] : w[2], &other = w[1] >= w[2] ? w[2] : w[1];
        big += 128 - (w[0] + w[1] + w[2] + w[3]);
        if (big > 127) {
            other += big - 127;
            big = 127;
        }
        for (int k = 0; k < 4; k++) filter[row * 8 + 2 + ip + k] = int8_t(w[k]);
    }
}

// Stand-in Gaussian sequence for the film grain tests and benchmarks: 2048 12-bit values with a bell
// shape (sum of eight 9-bit uniforms from a fixed xorshift, centred), NOT the AV1 table -- see
// easyav1_amd_synth.h.
extern "C" void ea1_synth_gaussian_sequence(int16_t sequence[2048])
{
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 2048; i++) {
        int sum = 0;
        for (int k = 0; k < 8; k++) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            sum += int(s >> 40) & 511;
        }
        sequence[i] = int16_t(sum - 2044);
    }
}
