/*
 * oracle/av1_recon.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of AV1 frame reconstruction
 * (prediction + residual add, deblocking, CDEF, loop restoration) over the ReconJob IR of
 * include/easyav1_amd_video.h.  It is the checker for the HIP kernels and the "port" CPU
 * baseline of bench.py; nothing in the product path may call it.
 *
 * Parity status: UNPINNED against dav1d (see oracle/av1_itx.c for why).  Units restated, with the
 * dav1d file that holds the reference arithmetic (paths from /root/reference/cmake/dav1d.cmake):
 *   inter prediction  src/mc_tmpl.c (:1013)              AV1 spec 7.11.3.4 (two-pass 8-tap,
 *                                                        Round2 by 3 then 11 / 7; compound average)
 *   intra prediction  src/ipred_tmpl.c (:1007)           AV1 spec 7.11.2 (DC, V, H, PAETH, SMOOTH*)
 *   residual add      src/recon_tmpl.c (:1014)           per block: predict, then itxfm_add
 *   deblocking        src/loopfilter_tmpl.c (:1010),     AV1 spec 7.14 (edge selection, limits,
 *                     src/lf_mask.c (:980)               narrow / 6 / 8 / 14-tap filters)
 *   CDEF              src/cdef_tmpl.c (:1003)            AV1 spec 7.15 (direction search, taps)
 *   loop restoration  src/looprestoration_tmpl.c (:1011) AV1 spec 7.17 (stripes, Wiener, SGR)
 * Everything is written from the published decoding process in the most literal per-pixel form
 * (no SIMD-style restructuring), so that it is easy to audit against the specification text.
 *
 * Picture layout in this file: three tightly packed planes (stride = plane width).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ea1_av1_tables.inc"
#include "easyav1_amd_video.h"

int ea1o_itx_add(uint8_t *dst, int stride, int w, int h, int txtp, const int16_t *coef);

static const int16_t subpel_filters[5][16][8] = {EA1_SUBPEL_FILTERS_VALUES};
static const uint8_t sm_weights[128] = {EA1_SM_WEIGHTS_VALUES};
static const int16_t sgr_params[16][4] = {EA1_SGR_PARAMS_VALUES};
static const int8_t cdef_directions[8][2][2] = {EA1_CDEF_DIRECTIONS_VALUES};
static const uint8_t cdef_pri_taps[2][2] = {EA1_CDEF_PRI_TAPS_VALUES};
static const uint8_t cdef_sec_taps[2] = {EA1_CDEF_SEC_TAPS_VALUES};
static const uint16_t cdef_div_table[9] = {EA1_CDEF_DIV_TABLE_VALUES};

typedef struct {
    int w, h;          /* luma size */
    uint8_t *p[3];     /* planes, stride = plane width */
} pic_t;

static inline int pw(const pic_t *f, int pl) { return pl ? f->w >> 1 : f->w; }
static inline int ph(const pic_t *f, int pl) { return pl ? f->h >> 1 : f->h; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip_px(int v) { return clampi(v, 0, 255); }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------------------------------
 * Inter prediction (AV1 spec 7.11.3.3 / 7.11.3.4, unscaled references)
 * ------------------------------------------------------------------------------------------ */
static int filter_index(int kind, int dim) { return dim <= 4 ? (kind == EA1_FILTER_SMOOTH ? 4 : 3) : kind; }

/* one prediction at "prep" scale: Round2(., 3) horizontally then Round2(., rnd1) vertically */
static void predict_inter(const pic_t *ref, int pl, int x0, int y0, int w, int h, int mvy, int mvx,
                          int filt_h, int filt_v, int rnd1, int32_t *out /* w*h */)
{
    const int ss = pl ? 1 : 0;
    const int lastx = pw(ref, pl) - 1, lasty = ph(ref, pl) - 1;
    /* position in 1/16 sample units of this plane */
    const int px = (x0 << 4) + (ss ? mvx : mvx * 2);
    const int py = (y0 << 4) + (ss ? mvy : mvy * 2);
    const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
    const int16_t *fh = subpel_filters[filter_index(filt_h, w)][fx];
    const int16_t *fv = subpel_filters[filter_index(filt_v, h)][fy];
    const uint8_t *src = ref->p[pl];
    const int stride = pw(ref, pl);
    int32_t *mid = malloc(sizeof(int32_t) * (size_t)w * (h + 7));
    for (int r = 0; r < h + 7; r++) {
        const int sy = clampi(iy + r - 3, 0, lasty);
        for (int c = 0; c < w; c++) {
            int s = 0;
            for (int t = 0; t < 8; t++) s += fh[t] * src[sy * stride + clampi(ix + c + t - 3, 0, lastx)];
            mid[r * w + c] = (s + 4) >> 3;
        }
    }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            int s = 0;
            for (int t = 0; t < 8; t++) s += fv[t] * mid[(r + t) * w + c];
            out[r * w + c] = (s + ((1 << rnd1) >> 1)) >> rnd1;
        }
    free(mid);
}

/* ------------------------------------------------------------------------------------------
 * Intra prediction (AV1 spec 7.11.2), modes without directional edges
 * ------------------------------------------------------------------------------------------ */
static void predict_intra(const pic_t *cur, int pl, int x, int y, int w, int h, int mode, int have_left,
                          int have_top, uint8_t *dst, int dstride)
{
    const uint8_t *f = cur->p[pl];
    const int stride = pw(cur, pl);
    int above[64], left[64], tl;
    for (int i = 0; i < w; i++)
        above[i] = have_top ? f[(y - 1) * stride + x + i] : (have_left ? f[y * stride + x - 1] : 127);
    for (int i = 0; i < h; i++)
        left[i] = have_left ? f[(y + i) * stride + x - 1] : (have_top ? f[(y - 1) * stride + x] : 129);
    if (have_top && have_left) tl = f[(y - 1) * stride + x - 1];
    else if (have_top) tl = f[(y - 1) * stride + x];
    else if (have_left) tl = f[y * stride + x - 1];
    else tl = 128;

    int dc = 128;
    if (mode == EA1_DC_PRED) {
        int sum = 0;
        if (have_top && have_left) {
            for (int i = 0; i < w; i++) sum += above[i];
            for (int i = 0; i < h; i++) sum += left[i];
            dc = (sum + ((w + h) >> 1)) / (w + h);
        } else if (have_left) {
            for (int i = 0; i < h; i++) sum += left[i];
            dc = (sum + (h >> 1)) / h;
        } else if (have_top) {
            for (int i = 0; i < w; i++) sum += above[i];
            dc = (sum + (w >> 1)) / w;
        }
    }
    const uint8_t *wx = sm_weights + w, *wy = sm_weights + h;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            int v;
            switch (mode) {
            case EA1_DC_PRED: v = dc; break;
            case EA1_V_PRED: v = above[j]; break;
            case EA1_H_PRED: v = left[i]; break;
            case EA1_PAETH_PRED: {
                int base = above[j] + left[i] - tl;
                int pl_ = iabs(base - left[i]), pt = iabs(base - above[j]), ptl = iabs(base - tl);
                v = (pl_ <= pt && pl_ <= ptl) ? left[i] : (pt <= ptl ? above[j] : tl);
                break;
            }
            case EA1_SMOOTH_PRED:
                v = (wy[i] * above[j] + (256 - wy[i]) * left[h - 1] + wx[j] * left[i] +
                     (256 - wx[j]) * above[w - 1] + 256) >> 9;
                break;
            case EA1_SMOOTH_V_PRED: v = (wy[i] * above[j] + (256 - wy[i]) * left[h - 1] + 128) >> 8; break;
            default: v = (wx[j] * left[i] + (256 - wx[j]) * above[w - 1] + 128) >> 8; break; /* SMOOTH_H */
            }
            dst[i * dstride + j] = (uint8_t)v;
        }
}

/* ------------------------------------------------------------------------------------------
 * Reconstruction driver: blocks in coded order; predict then add the residual, plane by plane
 * ------------------------------------------------------------------------------------------ */
static void recon_frame(const ea1_frame_desc *d, const ea1_block *blocks, const int16_t *coefs,
                        const pic_t refs[EA1_NUM_REF_SLOTS], pic_t *cur)
{
    int32_t *p0 = malloc(sizeof(int32_t) * 64 * 64), *p1 = malloc(sizeof(int32_t) * 64 * 64);
    for (uint32_t bi = 0; bi < d->num_blocks; bi++) {
        const ea1_block *b = &blocks[bi];
        const int16_t *cf = coefs + b->coef_off;
        for (int pl = 0; pl < 3; pl++) {
            const int ss = pl ? 1 : 0;
            const int w = (1 << b->bwl) >> ss, h = (1 << b->bhl) >> ss;
            const int x = (b->x4 * 4) >> ss, y = (b->y4 * 4) >> ss;
            const int stride = pw(cur, pl);
            uint8_t *dst = cur->p[pl] + y * stride + x;
            if (b->flags & EA1_BLK_INTER) {
                const int fh = b->mode_y & 3, fv = (b->mode_y >> 2) & 3;
                if (b->flags & EA1_BLK_COMPOUND) {
                    predict_inter(&refs[b->ref[0]], pl, x, y, w, h, b->mv[0][0], b->mv[0][1], fh, fv, 7, p0);
                    predict_inter(&refs[b->ref[1]], pl, x, y, w, h, b->mv[1][0], b->mv[1][1], fh, fv, 7, p1);
                    for (int i = 0; i < h; i++)
                        for (int j = 0; j < w; j++) dst[i * stride + j] = clip_px((p0[i * w + j] + p1[i * w + j] + 16) >> 5);
                } else {
                    predict_inter(&refs[b->ref[0]], pl, x, y, w, h, b->mv[0][0], b->mv[0][1], fh, fv, 11, p0);
                    for (int i = 0; i < h; i++)
                        for (int j = 0; j < w; j++) dst[i * stride + j] = clip_px(p0[i * w + j]);
                }
            } else {
                predict_intra(cur, pl, x, y, w, h, pl ? b->mode_uv : b->mode_y, !!(b->flags & EA1_BLK_HAVE_LEFT),
                              !!(b->flags & EA1_BLK_HAVE_TOP), dst, stride);
            }
            if (b->cf_flags & (1 << pl)) {
                ea1o_itx_add(dst, stride, w, h, b->txtp[pl], cf);
                cf += (w < 32 ? w : 32) * (h < 32 ? h : 32);
            }
        }
    }
    free(p0);
    free(p1);
}

/* 8x8-unit block map: index of the block covering each 8x8 luma unit */
static uint32_t *build_block_map(const ea1_frame_desc *d, const ea1_block *blocks)
{
    const int w8 = d->width / 8, h8 = d->height / 8;
    uint32_t *map = malloc(sizeof(uint32_t) * (size_t)w8 * h8);
    memset(map, 0xFF, sizeof(uint32_t) * (size_t)w8 * h8);
    for (uint32_t bi = 0; bi < d->num_blocks; bi++) {
        const ea1_block *b = &blocks[bi];
        const int x8 = b->x4 / 2, y8 = b->y4 / 2, n8w = (1 << b->bwl) / 8, n8h = (1 << b->bhl) / 8;
        for (int j = 0; j < n8h; j++)
            for (int i = 0; i < n8w; i++) map[(y8 + j) * w8 + x8 + i] = bi;
    }
    return map;
}

/* ------------------------------------------------------------------------------------------
 * Deblocking (AV1 spec 7.14)
 * ------------------------------------------------------------------------------------------ */
static inline int f4clamp(int v) { return clampi(v, -128, 127); }

/* filters the 1-D sample line around an edge; s points at q0, step is the distance between taps */
static void filter_edge_sample(uint8_t *s, int step, int plane, int filter_size, int lvl, int sharp)
{
    const int shift = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
    const int limit = sharp > 0 ? clampi(lvl >> shift, 1, 9 - sharp) : (lvl >> shift < 1 ? 1 : lvl >> shift);
    const int blimit = 2 * (lvl + 2) + limit;
    const int thresh = lvl >> 4;
    /* F(i): i < 0 -> p_{-i-1}, i >= 0 -> q_i */
#define F(i) ((int)s[(i) * step])
    const int p0 = F(-1), p1 = F(-2), q0 = F(0), q1 = F(1);
    const int filter_len = filter_size == 4 ? 4 : (plane > 0 ? 6 : (filter_size == 8 ? 8 : 16));
    const int hev = iabs(p1 - p0) > thresh || iabs(q1 - q0) > thresh;
    int mask = iabs(p1 - p0) <= limit && iabs(q1 - q0) <= limit && iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 <= blimit;
    if (filter_len >= 6) mask = mask && iabs(F(-3) - p1) <= limit && iabs(F(2) - q1) <= limit;
    if (filter_len >= 8) mask = mask && iabs(F(-4) - F(-3)) <= limit && iabs(F(3) - F(2)) <= limit;
    if (!mask) return;
    int flat = 0, flat2 = 0;
    if (filter_size >= 8) {
        flat = iabs(p1 - p0) <= 1 && iabs(q1 - q0) <= 1 && iabs(F(-3) - p0) <= 1 && iabs(F(2) - q0) <= 1;
        if (filter_len >= 8) flat = flat && iabs(F(-4) - p0) <= 1 && iabs(F(3) - q0) <= 1;
    }
    if (filter_size >= 16)
        flat2 = iabs(F(-7) - p0) <= 1 && iabs(F(6) - q0) <= 1 && iabs(F(-6) - p0) <= 1 && iabs(F(5) - q0) <= 1 &&
                iabs(F(-5) - p0) <= 1 && iabs(F(4) - q0) <= 1;
    if (filter_size == 4 || !flat) {
        /* narrow filter */
        const int ps1 = p1 - 128, ps0 = p0 - 128, qs0 = q0 - 128, qs1 = q1 - 128;
        int filter = hev ? f4clamp(ps1 - qs1) : 0;
        filter = f4clamp(filter + 3 * (qs0 - ps0));
        const int filter1 = f4clamp(filter + 4) >> 3;
        const int filter2 = f4clamp(filter + 3) >> 3;
        s[0] = (uint8_t)(f4clamp(qs0 - filter1) + 128);
        s[-step] = (uint8_t)(f4clamp(ps0 + filter2) + 128);
        if (!hev) {
            filter = (filter1 + 1) >> 1;
            s[step] = (uint8_t)(f4clamp(qs1 - filter) + 128);
            s[-2 * step] = (uint8_t)(f4clamp(ps1 + filter) + 128);
        }
        return;
    }
    /* wide filters */
    const int log2size = (filter_size == 8 || !flat2) ? 3 : 4;
    const int n = log2size == 4 ? 6 : (plane == 0 ? 3 : 2);
    const int n2 = (log2size == 3 && plane == 0) ? 0 : 1;
    int out[12];
    for (int i = -n; i < n; i++) {
        int t = 0;
        for (int j = -n; j <= n; j++) {
            const int p = clampi(i + j, -(n + 1), n);
            const int tap = iabs(j) <= n2 ? 2 : 1;
            t += F(p) * tap;
        }
        out[i + n] = (t + (1 << (log2size - 1))) >> log2size;
    }
    for (int i = -n; i < n; i++) s[i * step] = (uint8_t)out[i + n];
#undef F
}

static void deblock_frame(const ea1_frame_desc *d, const ea1_block *blocks, const uint32_t *map, pic_t *f)
{
    if (!d->lf_level[0] && !d->lf_level[1]) return;
    const int w8 = d->width / 8;
    for (int pl = 0; pl < 3; pl++) {
        if (pl > 0 && !d->lf_level[1 + pl]) continue;
        const int ss = pl ? 1 : 0;
        const int W = pw(f, pl), H = ph(f, pl);
        const int unit = 8 >> ss; /* size of an 8x8 luma unit in this plane */
        for (int pass = 0; pass < 2; pass++) {
            const int lvl = pl == 0 ? d->lf_level[pass] : d->lf_level[1 + pl];
            if (!lvl) continue;
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    /* (x, y) is q0; the edge lies between it and the previous sample */
                    const int pos = pass == 0 ? x : y;
                    if (pos == 0) continue;
                    const int ux = x / unit, uy = y / unit;
                    const uint32_t bi = map[uy * w8 + ux];
                    const ea1_block *b = &blocks[bi];
                    const int bpos = ((pass == 0 ? b->x4 : b->y4) * 4) >> ss;
                    if (pos != bpos) continue; /* tx == block: only block edges are transform edges */
                    const uint32_t pi = pass == 0 ? map[uy * w8 + (x - 1) / unit] : map[((y - 1) / unit) * w8 + ux];
                    const ea1_block *pb = &blocks[pi];
                    const int cur_dim = (1 << (pass == 0 ? b->bwl : b->bhl)) >> ss;
                    const int prev_dim = (1 << (pass == 0 ? pb->bwl : pb->bhl)) >> ss;
                    const int base = cur_dim < prev_dim ? cur_dim : prev_dim;
                    const int cap = pl == 0 ? 16 : 8;
                    const int fs = base < cap ? base : cap;
                    filter_edge_sample(f->p[pl] + y * W + x, pass == 0 ? 1 : W, pl, fs, lvl, d->lf_sharpness);
                }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * CDEF (AV1 spec 7.15)
 * ------------------------------------------------------------------------------------------ */
static int floor_log2(unsigned v)
{
    int l = 0;
    while (v >>= 1) l++;
    return l;
}

static int constrain(int diff, int threshold, int damping)
{
    if (!threshold) return 0;
    int adj = damping - floor_log2(threshold);
    if (adj < 0) adj = 0;
    const int mag = iabs(diff);
    int v = threshold - (mag >> adj);
    v = clampi(v, 0, mag);
    return diff < 0 ? -v : v;
}

static void cdef_direction(const uint8_t *src, int stride, int *dir, int *var)
{
    int cost[8] = {0}, partial[8][15];
    memset(partial, 0, sizeof partial);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int x = src[i * stride + j] - 128;
            partial[0][i + j] += x;
            partial[1][i + j / 2] += x;
            partial[2][i] += x;
            partial[3][3 + i - j / 2] += x;
            partial[4][7 + i - j] += x;
            partial[5][3 - i / 2 + j] += x;
            partial[6][j] += x;
            partial[7][i / 2 + j] += x;
        }
    for (int i = 0; i < 8; i++) {
        cost[2] += partial[2][i] * partial[2][i];
        cost[6] += partial[6][i] * partial[6][i];
    }
    cost[2] *= cdef_div_table[8];
    cost[6] *= cdef_div_table[8];
    for (int i = 0; i < 7; i++) {
        cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * cdef_div_table[i + 1];
        cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * cdef_div_table[i + 1];
    }
    cost[0] += partial[0][7] * partial[0][7] * cdef_div_table[8];
    cost[4] += partial[4][7] * partial[4][7] * cdef_div_table[8];
    for (int i = 1; i < 8; i += 2) {
        for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
        cost[i] *= cdef_div_table[8];
        for (int j = 0; j < 3; j++)
            cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * cdef_div_table[2 * j + 2];
    }
    int best = 0, bd = 0;
    for (int k = 0; k < 8; k++)
        if (cost[k] > best) {
            best = cost[k];
            bd = k;
        }
    *dir = bd;
    *var = (best - cost[(bd + 4) & 7]) >> 10;
}

static void cdef_filter_block(const pic_t *in, pic_t *out, int pl, int x0, int y0, int bw, int bh, int pri, int sec,
                              int damping, int dir)
{
    const int W = pw(in, pl), H = ph(in, pl);
    const uint8_t *src = in->p[pl];
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) {
            const int x = src[(y0 + i) * W + x0 + j];
            int sum = 0, mx = x, mn = x;
            for (int k = 0; k < 2; k++)
                for (int sign = -1; sign <= 1; sign += 2) {
                    int yy = y0 + i + sign * cdef_directions[dir][k][0];
                    int xx = x0 + j + sign * cdef_directions[dir][k][1];
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const int p = src[yy * W + xx];
                        sum += cdef_pri_taps[pri & 1][k] * constrain(p - x, pri, damping);
                        if (p > mx) mx = p;
                        if (p < mn) mn = p;
                    }
                    for (int doff = -2; doff <= 2; doff += 4) {
                        const int d2 = (dir + doff) & 7;
                        yy = y0 + i + sign * cdef_directions[d2][k][0];
                        xx = x0 + j + sign * cdef_directions[d2][k][1];
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            const int s = src[yy * W + xx];
                            sum += cdef_sec_taps[k] * constrain(s - x, sec, damping);
                            if (s > mx) mx = s;
                            if (s < mn) mn = s;
                        }
                    }
                }
            out->p[pl][(y0 + i) * W + x0 + j] = (uint8_t)clampi(x + ((8 + sum - (sum < 0)) >> 4), mn, mx);
        }
}

static void cdef_frame(const ea1_frame_desc *d, const ea1_block *blocks, const uint32_t *map, const uint8_t *cdef_idx,
                       const pic_t *in, pic_t *out)
{
    for (int pl = 0; pl < 3; pl++) memcpy(out->p[pl], in->p[pl], (size_t)pw(in, pl) * ph(in, pl));
    if (!d->cdef_enable) return;
    const int w8 = d->width / 8, h8 = d->height / 8;
    const int sbw = (d->width + 63) / 64;
    for (int y8 = 0; y8 < h8; y8++)
        for (int x8 = 0; x8 < w8; x8++) {
            const int idx = cdef_idx[(y8 / 8) * sbw + x8 / 8];
            if (idx == 0xFF) continue;
            const ea1_block *b = &blocks[map[y8 * w8 + x8]];
            if (b->flags & EA1_BLK_SKIP) continue;
            int ydir, var;
            cdef_direction(in->p[0] + (y8 * 8) * d->width + x8 * 8, d->width, &ydir, &var);
            int pri = d->cdef_y_strength[idx] >> 2, sec = d->cdef_y_strength[idx] & 3;
            if (sec == 3) sec = 4;
            int dir = pri == 0 ? 0 : ydir;
            const int var_str = (var >> 6) ? (floor_log2(var >> 6) < 12 ? floor_log2(var >> 6) : 12) : 0;
            pri = var ? (pri * (4 + var_str) + 8) >> 4 : 0;
            cdef_filter_block(in, out, 0, x8 * 8, y8 * 8, 8, 8, pri, sec, d->cdef_damping, dir);
            pri = d->cdef_uv_strength[idx] >> 2;
            sec = d->cdef_uv_strength[idx] & 3;
            if (sec == 3) sec = 4;
            dir = pri == 0 ? 0 : ydir;
            for (int pl = 1; pl < 3; pl++)
                cdef_filter_block(in, out, pl, x8 * 4, y8 * 4, 4, 4, pri, sec, d->cdef_damping - 1, dir);
        }
}

/* ------------------------------------------------------------------------------------------
 * Loop restoration (AV1 spec 7.17)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const pic_t *deblocked, *cdef;
    int pl, W, H;
    int stripe_start, stripe_end;
} lr_src_t;

/* get_source_sample: CDEF output inside the stripe, deblocked (pre-CDEF) rows just outside it */
static inline int lr_sample(const lr_src_t *s, int x, int y)
{
    x = clampi(x, 0, s->W - 1);
    y = clampi(y, 0, s->H - 1);
    if (y < s->stripe_start) {
        y = y > s->stripe_start - 2 ? y : s->stripe_start - 2;
        return s->deblocked->p[s->pl][y * s->W + x];
    }
    if (y > s->stripe_end) {
        y = y < s->stripe_end + 2 ? y : s->stripe_end + 2;
        return s->deblocked->p[s->pl][y * s->W + x];
    }
    return s->cdef->p[s->pl][y * s->W + x];
}

static void wiener_block(const lr_src_t *s, const ea1_lr_unit *u, int x, int y, int w, int h, uint8_t *dst)
{
    int vf[7], hf[7];
    vf[3] = hf[3] = 128;
    for (int i = 0; i < 3; i++) {
        vf[i] = vf[6 - i] = u->coef[0][i];
        vf[3] -= 2 * u->coef[0][i];
        hf[i] = hf[6 - i] = u->coef[1][i];
        hf[3] -= 2 * u->coef[1][i];
    }
    int32_t *mid = malloc(sizeof(int32_t) * (size_t)w * (h + 6));
    for (int r = 0; r < h + 6; r++)
        for (int c = 0; c < w; c++) {
            int sum = 0;
            for (int t = 0; t < 7; t++) sum += hf[t] * lr_sample(s, x + c + t - 3, y + r - 3);
            mid[r * w + c] = clampi((sum + 4) >> 3, -(1 << 13), (1 << 13) - 1);
        }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            int sum = 0;
            for (int t = 0; t < 7; t++) sum += vf[t] * mid[(r + t) * w + c];
            dst[(y + r) * s->W + x + c] = (uint8_t)clip_px((sum + 1024) >> 11);
        }
    free(mid);
}

/* box filter of one pass over the (w x h) block at (x, y); flt gets the filtered output */
static void sgr_box(const lr_src_t *s, int x, int y, int w, int h, int pass, int r, int eps, int32_t *flt)
{
    const int n = (2 * r + 1) * (2 * r + 1);
    const int n2e = n * n * eps;
    const unsigned sc = ((1u << 20) + n2e / 2) / n2e;
    const int one_by_n = ((1 << 12) + n / 2) / n;
    const int aw = w + 2;
    int32_t *A = malloc(sizeof(int32_t) * (size_t)aw * (h + 2)), *B = malloc(sizeof(int32_t) * (size_t)aw * (h + 2));
    for (int i = -1; i <= h; i++)
        for (int j = -1; j <= w; j++) {
            unsigned a = 0, b = 0;
            for (int dy = -r; dy <= r; dy++)
                for (int dx = -r; dx <= r; dx++) {
                    const unsigned c = lr_sample(s, x + j + dx, y + i + dy);
                    a += c * c;
                    b += c;
                }
            const int64_t pv = (int64_t)a * n - (int64_t)b * b;
            const uint64_t p = pv < 0 ? 0 : (uint64_t)pv;
            const uint64_t z = (p * sc + (1u << 19)) >> 20;
            int a2;
            if (z >= 255) a2 = 256;
            else if (z == 0) a2 = 1;
            else a2 = (int)(((z << 8) + z / 2) / (z + 1));
            const int64_t b2 = (int64_t)(256 - a2) * b * one_by_n;
            A[(i + 1) * aw + j + 1] = a2;
            B[(i + 1) * aw + j + 1] = (int32_t)((b2 + (1 << 11)) >> 12);
        }
    for (int i = 0; i < h; i++) {
        int shift = 5;
        if (pass == 0 && (i & 1)) shift = 4;
        for (int j = 0; j < w; j++) {
            int a = 0, b = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int weight;
                    if (pass == 0) weight = ((i + dy) & 1) ? (dx == 0 ? 6 : 5) : 0;
                    else weight = (dx == 0 || dy == 0) ? 4 : 3;
                    a += weight * A[(i + 1 + dy) * aw + j + 1 + dx];
                    b += weight * B[(i + 1 + dy) * aw + j + 1 + dx];
                }
            const int v = a * s->cdef->p[s->pl][(y + i) * s->W + x + j] + b;
            const int sh = 8 + shift - 4;
            flt[i * w + j] = (v + (1 << (sh - 1))) >> sh;
        }
    }
    free(A);
    free(B);
}

static void sgr_block(const lr_src_t *s, const ea1_lr_unit *u, int x, int y, int w, int h, uint8_t *dst)
{
    const int16_t *prm = sgr_params[u->sgr_set & 15];
    const int w0 = u->coef[0][0], w1 = u->coef[0][1], w2 = 128 - w0 - w1;
    int32_t *f0 = malloc(sizeof(int32_t) * (size_t)w * h), *f1 = malloc(sizeof(int32_t) * (size_t)w * h);
    if (prm[0]) sgr_box(s, x, y, w, h, 0, prm[0], prm[1], f0);
    if (prm[2]) sgr_box(s, x, y, w, h, 1, prm[2], prm[3], f1);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int uu = s->cdef->p[s->pl][(y + i) * s->W + x + j] << 4;
            int v = w1 * uu;
            v += w0 * (prm[0] ? f0[i * w + j] : uu);
            v += w2 * (prm[2] ? f1[i * w + j] : uu);
            dst[(y + i) * s->W + x + j] = (uint8_t)clip_px((v + (1 << 10)) >> 11);
        }
    free(f0);
    free(f1);
}

static int lr_units(int unit_size, int plane_size)
{
    int n = (plane_size + (unit_size >> 1)) / unit_size;
    return n < 1 ? 1 : n;
}

static void lr_frame(const ea1_frame_desc *d, const ea1_lr_unit *const lr[3], const pic_t *deblocked, const pic_t *cdef,
                     pic_t *out)
{
    for (int pl = 0; pl < 3; pl++) {
        const int W = pw(cdef, pl), H = ph(cdef, pl), ss = pl ? 1 : 0;
        memcpy(out->p[pl], cdef->p[pl], (size_t)W * H);
        if (!d->lr_type[pl] || !lr[pl]) continue;
        const int usz = 1 << d->lr_unit_log2[pl];
        const int urows = lr_units(usz, H), ucols = lr_units(usz, W);
        /* walk the plane in (stripe x unit-column) blocks of 4-luma-row granularity: rows of one
           restoration unit row / stripe share all parameters */
        for (int y = 0; y < H;) {
            const int luma_y = y << ss;
            const int stripe = (luma_y + 8) / 64;
            int urow = (y + (8 >> ss)) / usz;
            if (urow > urows - 1) urow = urows - 1;
            lr_src_t s = {deblocked, cdef, pl, W, H, 0, 0};
            s.stripe_start = (-8 + stripe * 64) >> ss;
            s.stripe_end = s.stripe_start + (64 >> ss) - 1;
            /* rows until the stripe or the unit row changes */
            int y_end = s.stripe_end + 1;
            if (urow < urows - 1) {
                const int next_unit_y = (urow + 1) * usz - (8 >> ss);
                if (next_unit_y < y_end) y_end = next_unit_y;
            }
            if (y_end > H) y_end = H;
            for (int uc = 0; uc < ucols; uc++) {
                const int x = uc * usz;
                const int x_end = uc == ucols - 1 ? W : x + usz;
                const ea1_lr_unit *u = &lr[pl][urow * ucols + uc];
                if (u->type == EA1_LR_WIENER) wiener_block(&s, u, x, y, x_end - x, y_end - y, out->p[pl]);
                else if (u->type == EA1_LR_SGRPROJ) sgr_block(&s, u, x, y, x_end - x, y_end - y, out->p[pl]);
            }
            y = y_end;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Public entry points (ctypes).  Planes are tightly packed; refs: 8 slots x 3 plane pointers
 * (unused slots may be NULL).  Each out_* (3 plane pointers) may be NULL if not wanted.
 * ------------------------------------------------------------------------------------------ */
static void pic_alloc(pic_t *p, int w, int h)
{
    p->w = w;
    p->h = h;
    p->p[0] = malloc((size_t)w * h);
    p->p[1] = malloc((size_t)w * h / 4);
    p->p[2] = malloc((size_t)w * h / 4);
}
static void pic_free(pic_t *p)
{
    for (int i = 0; i < 3; i++) free(p->p[i]);
}
static void pic_export(const pic_t *p, uint8_t *const out[3])
{
    if (!out) return;
    for (int pl = 0; pl < 3; pl++)
        if (out[pl]) memcpy(out[pl], p->p[pl], (size_t)pw(p, pl) * ph(p, pl));
}

int ea1o_decode_frame(const ea1_frame_desc *d, const ea1_block *blocks, const int16_t *coefs, const uint8_t *cdef_idx,
                      const ea1_lr_unit *const lr[3], uint8_t *const refs[EA1_NUM_REF_SLOTS][3], unsigned stages,
                      uint8_t *const out_recon[3], uint8_t *const out_deblock[3], uint8_t *const out_cdef[3],
                      uint8_t *const out_final[3])
{
    pic_t rp[EA1_NUM_REF_SLOTS];
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) {
        rp[i].w = d->width;
        rp[i].h = d->height;
        for (int pl = 0; pl < 3; pl++) rp[i].p[pl] = refs ? refs[i][pl] : NULL;
    }
    pic_t cur, cd, fin;
    pic_alloc(&cur, d->width, d->height);
    pic_alloc(&cd, d->width, d->height);
    pic_alloc(&fin, d->width, d->height);
    for (int pl = 0; pl < 3; pl++) memset(cur.p[pl], 0, (size_t)pw(&cur, pl) * ph(&cur, pl));
    uint32_t *map = build_block_map(d, blocks);
    recon_frame(d, blocks, coefs, rp, &cur);
    pic_export(&cur, out_recon);
    if (stages & EA1_STAGE_DEBLOCK) deblock_frame(d, blocks, map, &cur);
    pic_export(&cur, out_deblock);
    static const ea1_frame_desc zero_desc;
    const ea1_frame_desc *dc = (stages & EA1_STAGE_CDEF) ? d : &zero_desc;
    if (stages & EA1_STAGE_CDEF) cdef_frame(d, blocks, map, cdef_idx, &cur, &cd);
    else
        for (int pl = 0; pl < 3; pl++) memcpy(cd.p[pl], cur.p[pl], (size_t)pw(&cur, pl) * ph(&cur, pl));
    (void)dc;
    pic_export(&cd, out_cdef);
    if (stages & EA1_STAGE_LR) lr_frame(d, lr, &cur, &cd, &fin);
    else
        for (int pl = 0; pl < 3; pl++) memcpy(fin.p[pl], cd.p[pl], (size_t)pw(&cd, pl) * ph(&cd, pl));
    pic_export(&fin, out_final);
    free(map);
    pic_free(&cur);
    pic_free(&cd);
    pic_free(&fin);
    return 0;
}
