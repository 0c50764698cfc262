/*
 * oracle/av1_recon.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of AV1 frame reconstruction
 * (prediction + residual add, deblocking, CDEF, loop restoration) over the ReconJob IR of
 * include/easyav1_amd_video.h.  It is the checker for the HIP kernels and the "port" CPU
 * baseline of bench.py; nothing in the product path may call it.
 *
 * Parity status: PINNED since round 4 against dav1d 1.5.3 (the decoder the reference pins; its dav1d_* entry points out of Pillow's bundled libavif) on libaom 3.13.2 streams, bit-exact: tests/test_real_av1.py, tests/golden/make_video_fixtures.py, DESIGN.md section 2
 * -- every frame of 31 committed streams after reconstruction, deblocking, CDEF and restoration (dav1d's inloop_filters setting).  Units restated, with the
 * dav1d file that holds the reference arithmetic (paths from /root/reference/cmake/dav1d.cmake):
 *   inter prediction  src/mc_tmpl.c (:1013)              AV1 spec 7.11.3.4 (two-pass 8-tap,
 *                                                        Round2 by 3 then 11 / 7; compound average)
 *   warped motion     src/warpmv.c (:994), mc_tmpl.c     AV1 spec 7.11.3.5 / 7.11.3.6 (shear setup,
 *                                                        per-8x8 warp; filter table from the caller)
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

static const int16_t subpel_filters[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};
static const uint8_t sm_weights[128] = {EA1_SM_WEIGHTS_VALUES};
static const int16_t sgr_params[16][4] = {EA1_SGR_PARAMS_VALUES};
static const int8_t cdef_directions[8][2][2] = {EA1_CDEF_DIRECTIONS_VALUES};
static const uint8_t cdef_pri_taps[2][2] = {EA1_CDEF_PRI_TAPS_VALUES};
static const uint8_t cdef_sec_taps[2] = {EA1_CDEF_SEC_TAPS_VALUES};
static const uint16_t cdef_div_table[9] = {EA1_CDEF_DIV_TABLE_VALUES};

typedef struct {
    int w, h;          /* luma size of the coded area (whole 8x8 units) */
    uint8_t *p[3];     /* planes, stride = plane width */
    int vw, vh;        /* visible luma size (ea1_frame_desc.crop): where reference reads and restoration clamp */
} pic_t;

/* Chroma subsampling of the stream (subsampling_x / subsampling_y: 4:2:0 and 4:0:0 1, 1 -- the default; 4:2:2 1, 0; 4:4:4 0, 0), for the next
 * ea1o_decode_frame calls of this thread (round 5: ea1_frame_desc does not carry it yet -- the product's reconstruction path is 4:2:0 only).
 * Stored XOR 1 so that zero-initialised thread storage means 4:2:0. */
static __thread int g_ssx_inv, g_ssy_inv;
void ea1o_set_chroma_format(int ssx, int ssy) { g_ssx_inv = !ssx, g_ssy_inv = !ssy; }
static inline int ssx(int pl) { return pl ? !g_ssx_inv : 0; }
static inline int ssy(int pl) { return pl ? !g_ssy_inv : 0; }
static inline int pw(const pic_t *f, int pl) { return f->w >> ssx(pl); }
static inline int ph(const pic_t *f, int pl) { return f->h >> ssy(pl); }
/* visible plane size: Round2(size, subsampling) (AV1 spec 7.11.3.3 lastX / lastY, 7.17 PlaneEndX / unitRows) */
static inline int pvw(const pic_t *f, int pl) { return (f->vw + ssx(pl)) >> ssx(pl); }
static inline int pvh(const pic_t *f, int pl) { return (f->vh + ssy(pl)) >> ssy(pl); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip_px(int v) { return clampi(v, 0, 255); }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------------------------------
 * Inter prediction (AV1 spec 7.11.3.3 / 7.11.3.4, unscaled references)
 * ------------------------------------------------------------------------------------------ */
static int filter_index(int kind, int dim)
{
    if (kind == EA1_FILTER_BILINEAR) return 5;
    return dim <= 4 ? (kind == EA1_FILTER_SMOOTH ? 4 : 3) : kind;
}

/* Scaled references (7.11.3.3 motion vector scaling, 7.11.3.4 block inter prediction): the size of the frame being
 * decoded, set by ea1o_decode_frame; a reference of another size is read through the scaled position formulas.
 * ea1o_set_ref_sizes states the references' sizes for the next ea1o_decode_frame calls (0 = the frame's own size). */
static __thread int g_frame_w, g_frame_h;   /* per thread: bench.py and the long parity runs decode on several threads */
/* ea1_frame_desc.pad: the TRUE coded size (MiCols * 4 x MiRows * 4) is width x height minus the padding that lets blocks reach
 * beyond it; crop counts from the true coded size */
static inline int d_coded_w(const ea1_frame_desc *d) { return (int)d->width - 8 * (d->pad & 15); }
static inline int d_coded_h(const ea1_frame_desc *d) { return (int)d->height - 8 * (d->pad >> 4); }
static inline int d_vis_w(const ea1_frame_desc *d) { return d_coded_w(d) - (d->crop & 15); }
static inline int d_vis_h(const ea1_frame_desc *d) { return d_coded_h(d) - (d->crop >> 4); }
static __thread int g_coded_w, g_coded_h;   /* of the frame being decoded */
static __thread int g_frame_vw, g_frame_vh; /* the frame's TRUE size (coded size minus crop): what 7.11.3.3 calls FrameWidth / FrameHeight */
static __thread uint32_t g_ref_sizes[EA1_NUM_REF_SLOTS][2];
static __thread int g_ref_crops[EA1_NUM_REF_SLOTS];   /* -1: the frame's own crop; else ea1_frame_desc.crop of the picture in the slot */
/* crop byte (ea1_frame_desc.crop layout) of the picture each slot holds, for the next ea1o_decode_frame calls; NULL = every
 * reference has the crop of the frame being decoded */
void ea1o_set_ref_crops(const int crops[EA1_NUM_REF_SLOTS])
{
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) g_ref_crops[i] = crops ? crops[i] + 1 : 0;   /* stored + 1 so that zero-initialised thread storage means "unset" */
}
void ea1o_set_ref_sizes(const uint32_t sizes[EA1_NUM_REF_SLOTS][2])
{
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) {
        g_ref_sizes[i][0] = sizes ? sizes[i][0] : 0;
        g_ref_sizes[i][1] = sizes ? sizes[i][1] : 0;
    }
}
static int64_t r2s64(int64_t v, int n) { return v >= 0 ? (v + ((int64_t)1 << (n - 1))) >> n : -((-v + ((int64_t)1 << (n - 1))) >> n); }

/* (x0, y0): the BLOCK's top-left sample in plane pl; positions in 1/1024 sample of the reference */
static void predict_inter_scaled(const pic_t *ref, int pl, int x0, int y0, int w, int h, int mvy, int mvx, int filt_h, int filt_v, int rnd1, int32_t *out)
{
    const int sx_ = ssx(pl), sy_ = ssy(pl);
    const int lastx = pvw(ref, pl) - 1, lasty = pvh(ref, pl) - 1;
    /* 7.11.3.3: xScale = ((RefUpscaledWidth << REF_SCALE_SHIFT) + FrameWidth / 2) / FrameWidth on the TRUE sizes of the two
     * pictures (dav1d scales with p.w / p.h the same way), not on their sizes in whole 8x8 units */
    const int xscale = (int)((((int64_t)ref->vw << 14) + g_frame_vw / 2) / g_frame_vw);
    const int yscale = (int)((((int64_t)ref->vh << 14) + g_frame_vh / 2) / g_frame_vh);
    const int64_t orig_x = ((int64_t)x0 << 4) + ((2 * mvx) >> sx_) + 8, orig_y = ((int64_t)y0 << 4) + ((2 * mvy) >> sy_) + 8;
    const int start_x = (int)r2s64(orig_x * xscale - ((int64_t)8 << 14), 8) + 32;
    const int start_y = (int)r2s64(orig_y * yscale - ((int64_t)8 << 14), 8) + 32;
    const int step_x = (int)r2s64(xscale, 4), step_y = (int)r2s64(yscale, 4);
    const int fih = filter_index(filt_h, w), fiv = filter_index(filt_v, h);
    const uint8_t *src = ref->p[pl];
    const int stride = pw(ref, pl);
    /* the rows of the reference the block touches, filtered horizontally once (the specification's intermediate array) */
    const int rows = ((((h - 1) * step_y + (1 << 10) - 1) >> 10) + 8);
    int32_t *mid = malloc(sizeof(int32_t) * (size_t)w * (rows + 1));
    for (int r = 0; r < rows + 1; r++) {
        const int sy = clampi((start_y >> 10) + r - 3, 0, lasty);
        for (int c = 0; c < w; c++) {
            const int p = start_x + step_x * c;
            const int16_t *fh = subpel_filters[fih][(p >> 6) & 15];
            int s = 0;
            for (int t = 0; t < 8; t++) s += fh[t] * src[sy * stride + clampi((p >> 10) + t - 3, 0, lastx)];
            mid[r * w + c] = (s + 4) >> 3;
        }
    }
    for (int r = 0; r < h; r++) {
        const int p = (start_y & 1023) + step_y * r;
        const int16_t *fv = subpel_filters[fiv][(p >> 6) & 15];
        for (int c = 0; c < w; c++) {
            int s = 0;
            for (int t = 0; t < 8; t++) s += fv[t] * mid[((p >> 10) + t) * w + c];
            out[r * w + c] = (s + ((1 << rnd1) >> 1)) >> rnd1;
        }
    }
    free(mid);
}

/* one prediction at "prep" scale: Round2(., 3) horizontally then Round2(., rnd1) vertically.
 * (x0, y0) and w x h are in samples of plane pl; fw/fh_dim select the 4-tap sets (block size in
 * this plane, which for an OBMC strip is NOT the strip size but still the predicted w x h) */
static void predict_inter(const pic_t *ref, int pl, int x0, int y0, int w, int h, int mvy, int mvx,
                          int filt_h, int filt_v, int rnd1, int32_t *out /* w*h */)
{
    if (g_frame_vw && (ref->vw != g_frame_vw || ref->vh != g_frame_vh)) {
        predict_inter_scaled(ref, pl, x0, y0, w, h, mvy, mvx, filt_h, filt_v, rnd1, out);
        return;
    }
    const int lastx = pvw(ref, pl) - 1, lasty = pvh(ref, pl) - 1;
    /* position in 1/16 sample units of this plane (7.11.3.3: (2 * mv) >> subsampling) */
    const int px = (x0 << 4) + (ssx(pl) ? mvx : mvx * 2);
    const int py = (y0 << 4) + (ssy(pl) ? mvy : mvy * 2);
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
 * Warped motion (AV1 spec 7.11.3.5 block warp process, 7.11.3.6 setup shear process; dav1d
 * src/warpmv.c cmake/dav1d.cmake:994 + warp8x8 in src/mc_tmpl.c :1013)
 * ------------------------------------------------------------------------------------------ */
/* Warped_Filters[193][8]: the table of ea1_av1_tables.inc ([recalled]) unless a test hands over another one (as it may to the
 * library: easyav1_amd_video.h, ea1_recon_set_warp_filter); NULL restores the built-in table */
static const int8_t warped_filters_builtin[193][8] = {EA1_WARPED_FILTERS_VALUES};
static int8_t warp_filter[193][8] = {EA1_WARPED_FILTERS_VALUES};

void ea1o_set_warp_filter(const int8_t *f) { memcpy(warp_filter, f ? (const void *)f : (const void *)warped_filters_builtin, sizeof warp_filter); }

static int64_t round2signed64(int64_t v, int n) { return v >= 0 ? (v + ((int64_t)1 << (n - 1))) >> n : -((-v + ((int64_t)1 << (n - 1))) >> n); }

/* Div_Lut[k] = Round(2^14 * 256 / (256 + k)) (7.11.3.7) */
static int div_lut(int k) { return (int)((((int64_t)1 << 22) + (256 + k) / 2) / (256 + k)); }

/* 7.11.3.6: alpha, beta, gamma, delta of an affine matrix; returns warpValid */
int ea1o_setup_shear(const int32_t mat[6], int sh[4])
{
    if (mat[2] <= 0) return 0;
    const int alpha0 = clampi(mat[2] - (1 << 16), -32768, 32767);
    const int beta0 = clampi(mat[3], -32768, 32767);
    /* resolve divisor (7.11.3.7) */
    int n = 0;
    while (((unsigned)mat[2] >> (n + 1)) != 0) n++;
    const int64_t e = (int64_t)mat[2] - ((int64_t)1 << n);
    const int f = (int)(n > 8 ? (e + ((int64_t)1 << (n - 8 - 1))) >> (n - 8) : e << (8 - n));
    const int div_shift = n + 14, div_factor = div_lut(f);
    const int64_t v = (int64_t)mat[4] * 65536;
    const int64_t g = round2signed64(v * div_factor, div_shift);
    const int gamma0 = (int)(g < -32768 ? -32768 : (g > 32767 ? 32767 : g));
    const int64_t w = (int64_t)mat[3] * mat[4];
    const int64_t dl = (int64_t)mat[5] - round2signed64(w * div_factor, div_shift) - (1 << 16);
    const int delta0 = (int)(dl < -32768 ? -32768 : (dl > 32767 ? 32767 : dl));
    const int in[4] = {alpha0, beta0, gamma0, delta0};
    for (int k = 0; k < 4; k++) sh[k] = (int)round2signed64(in[k], 6) * 64;
    if (4 * iabs(sh[0]) + 7 * iabs(sh[1]) >= (1 << 16)) return 0;
    if (4 * iabs(sh[2]) + 4 * iabs(sh[3]) >= (1 << 16)) return 0;
    return 1;
}

/* 7.11.3.5 for every 8x8 of a w x h prediction (w, h multiples of 8) at (x0, y0) of plane pl */
static void predict_warp(const pic_t *ref, int pl, int x0, int y0, int w, int h, const int32_t mat[6], const int sh[4],
                         int rnd1, int32_t *out /* w*h */)
{
    const int sx_ = ssx(pl), sy_ = ssy(pl);
    const int lastx = pvw(ref, pl) - 1, lasty = pvh(ref, pl) - 1;
    const uint8_t *src = ref->p[pl];
    const int stride = pw(ref, pl);
    for (int by = 0; by < h; by += 8)
        for (int bx = 0; bx < w; bx += 8) {
            const int64_t src_x = (int64_t)(x0 + bx + 4) << sx_, src_y = (int64_t)(y0 + by + 4) << sy_;
            const int64_t dst_x = mat[2] * src_x + mat[3] * src_y + mat[0];
            const int64_t dst_y = mat[4] * src_x + mat[5] * src_y + mat[1];
            const int64_t x4 = dst_x >> sx_, y4 = dst_y >> sy_;
            const int ix4 = (int)(x4 >> 16), sx4 = (int)(x4 & 0xFFFF);
            const int iy4 = (int)(y4 >> 16), sy4 = (int)(y4 & 0xFFFF);
            int mid[15][8];
            for (int i1 = -7; i1 < 8; i1++)
                for (int i2 = -4; i2 < 4; i2++) {
                    const int sx = sx4 + sh[0] * i2 + sh[1] * i1;
                    const int offs = ((sx + 512) >> 10) + 64;
                    const int sy = clampi(iy4 + i1, 0, lasty);
                    int s = 0;
                    for (int i3 = 0; i3 < 8; i3++) s += warp_filter[offs][i3] * src[sy * stride + clampi(ix4 + i2 - 3 + i3, 0, lastx)];
                    mid[i1 + 7][i2 + 4] = (s + 4) >> 3;
                }
            for (int i1 = -4; i1 < 4; i1++)
                for (int i2 = -4; i2 < 4; i2++) {
                    const int sy = sy4 + sh[2] * i2 + sh[3] * i1;
                    const int offs = ((sy + 512) >> 10) + 64;
                    int s = 0;
                    for (int i3 = 0; i3 < 8; i3++) s += warp_filter[offs][i3] * mid[i1 + i3 + 4][i2 + 4];
                    out[(by + i1 + 4) * w + bx + i2 + 4] = (s + ((1 << rnd1) >> 1)) >> rnd1;
                }
        }
}

/* ------------------------------------------------------------------------------------------
 * Wedge masks (AV1 spec 7.11.3.11; dav1d src/wedge.c, cmake/dav1d.cmake:995)
 * ------------------------------------------------------------------------------------------ */
static const uint8_t wedge_master_odd[64] = {EA1_WEDGE_MASTER_ODD_VALUES};
static const uint8_t wedge_master_even[64] = {EA1_WEDGE_MASTER_EVEN_VALUES};
static const uint8_t wedge_master_vertical[64] = {EA1_WEDGE_MASTER_VERTICAL_VALUES};
static const uint8_t wedge_codebook[3][16][3] = {EA1_WEDGE_CODEBOOK_VALUES};
static uint8_t master_mask[6][64][64];
static int master_ready;

static void wedge_init_master(void)
{
    if (master_ready) return;
    enum { HOR = 0, VER = 1, O27 = 2, O63 = 3, O117 = 4, O153 = 5 };
    for (int j = 0; j < 64; j++) {
        int shift = 64 / 4;
        for (int i = 0; i < 64; i += 2) {
            master_mask[O63][i][j] = wedge_master_even[clampi(j - shift, 0, 63)];
            shift -= 1;
            master_mask[O63][i + 1][j] = wedge_master_odd[clampi(j - shift, 0, 63)];
            master_mask[VER][i][j] = wedge_master_vertical[j];
            master_mask[VER][i + 1][j] = wedge_master_vertical[j];
        }
    }
    for (int i = 0; i < 64; i++)
        for (int j = 0; j < 64; j++) {
            const int msk = master_mask[O63][i][j];
            master_mask[O27][j][i] = (uint8_t)msk;
            master_mask[O117][i][63 - j] = (uint8_t)(64 - msk);
            master_mask[O153][63 - j][i] = (uint8_t)(64 - msk);
            master_mask[HOR][j][i] = master_mask[VER][i][j];
        }
    master_ready = 1;
}

/* mask of luma block w x h (8..32 each) for (wedge index, sign); out is w*h */
void ea1o_wedge_mask(int w, int h, int index, int sign, uint8_t *out)
{
    wedge_init_master();
    const uint8_t *cb = wedge_codebook[h > w ? 0 : (h < w ? 1 : 2)][index];
    const int dir = cb[0];
    const int xoff = 32 - ((cb[1] * w) >> 3), yoff = 32 - ((cb[2] * h) >> 3);
    int sum = 0;
    for (int i = 0; i < w; i++) sum += master_mask[dir][yoff][xoff + i];
    for (int i = 1; i < h; i++) sum += master_mask[dir][yoff + i][xoff];
    const int avg = (sum + (w + h - 1) / 2) / (w + h - 1);
    const int flip = avg < 32;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int m = master_mask[dir][yoff + i][xoff + j];
            out[i * w + j] = (uint8_t)((flip == sign) ? m : 64 - m);
        }
}

/* ------------------------------------------------------------------------------------------
 * Intra prediction (AV1 spec 7.11.2)
 * ------------------------------------------------------------------------------------------ */
static const uint16_t dr_intra_derivative[88] = {EA1_DR_INTRA_DERIVATIVE_VALUES};
static const uint8_t intra_edge_kernel[3][5] = {EA1_INTRA_EDGE_KERNEL_VALUES};
static const int8_t filter_intra_taps[5][8][7] = {EA1_FILTER_INTRA_TAPS_VALUES};
static const uint8_t mode_to_angle[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};

typedef struct {
    int have_left, have_top, have_tr, have_bl;
    int mode;          /* EA1_*_PRED (never UV_CFL here: the caller passes DC) */
    int angle_delta;   /* -3..3 */
    int smooth_edge;   /* get_filter_type(): a neighbour uses a SMOOTH mode */
    int edge_filter;   /* sequence-level enable_intra_edge_filter */
    int filter_intra;  /* 0 or 1 + EA1_FILTER_* */
} intra_args;

/* 7.11.2.9 */
static int edge_filter_strength(int w, int h, int filter_type, int delta)
{
    const int d = iabs(delta), blk_wh = w + h;
    int strength = 0;
    if (filter_type == 0) {
        if (blk_wh <= 8) {
            if (d >= 56) strength = 1;
        } else if (blk_wh <= 12) {
            if (d >= 40) strength = 1;
        } else if (blk_wh <= 16) {
            if (d >= 40) strength = 1;
        } else if (blk_wh <= 24) {
            if (d >= 8) strength = 1;
            if (d >= 16) strength = 2;
            if (d >= 32) strength = 3;
        } else if (blk_wh <= 32) {
            strength = 1;
            if (d >= 4) strength = 2;
            if (d >= 32) strength = 3;
            if (d < 1) strength = 0;
        } else {
            if (d >= 1) strength = 3;
        }
    } else {
        if (blk_wh <= 8) {
            if (d >= 40) strength = 1;
            if (d >= 64) strength = 2;
        } else if (blk_wh <= 16) {
            if (d >= 20) strength = 1;
            if (d >= 48) strength = 2;
        } else if (blk_wh <= 24) {
            if (d >= 4) strength = 3;
        } else {
            if (d >= 1) strength = 3;
        }
    }
    return strength;
}

/* 7.11.2.12: buf[-1 .. sz-2] is the edge including the corner */
static void edge_filter(int *buf, int sz, int strength)
{
    if (!strength) return;
    int edge[160];
    for (int i = 0; i < sz; i++) edge[i] = buf[i - 1];
    for (int i = 1; i < sz; i++) {
        int s = 0;
        for (int j = 0; j < 5; j++) s += intra_edge_kernel[strength - 1][j] * edge[clampi(i - 2 + j, 0, sz - 1)];
        buf[i - 1] = (s + 8) >> 4;
    }
}

/* 7.11.2.10 */
static int use_upsample(int w, int h, int filter_type, int delta)
{
    const int d = iabs(delta), blk_wh = w + h;
    if (d <= 0 || d >= 40) return 0;
    return filter_type ? blk_wh <= 8 : blk_wh <= 16;
}

/* 7.11.2.11: buf[-1 .. num_px-1] -> buf[-2 .. 2*num_px-2] */
static void edge_upsample(int *buf, int num_px)
{
    int dup[80];
    dup[0] = buf[-1];
    for (int i = -1; i < num_px; i++) dup[i + 2] = buf[i];
    dup[num_px + 2] = buf[num_px - 1];
    buf[-2] = dup[0];
    for (int i = 0; i < num_px; i++) {
        int s = -dup[i] + 9 * dup[i + 1] + 9 * dup[i + 2] - dup[i + 3];
        buf[2 * i - 1] = clip_px((s + 8) >> 4);
        buf[2 * i] = dup[i + 2];
    }
}

static void predict_intra(const pic_t *cur, int pl, int x, int y, int w, int h, const intra_args *a, uint8_t *dst,
                          int dstride)
{
    const uint8_t *f = cur->p[pl];
    const int stride = pw(cur, pl);
    /* 7.11.2 maxX / maxY: the last sample of the TRUE coded plane (blocks may reach beyond it, ea1_frame_desc.pad) */
    const int max_x = (g_coded_w >> ssx(pl)) - 1, max_y = (g_coded_h >> ssy(pl)) - 1;
    const int have_left = a->have_left, have_top = a->have_top;
    int above_buf[16 + 160], left_buf[16 + 160];
    int *above = above_buf + 16, *left = left_buf + 16;
    const int n = w + h;
    /* edge preparation (7.11.2) */
    /* The above-right / below-left extension holds at most w (resp. h) real samples -- what the
     * availability flag vouches for; entries beyond them repeat the last one (libaom
     * build_intra_predictors: n_topright_px = min(txw, ..), n_bottomleft_px = min(txh, ..)). */
    for (int i = 0; i < n; i++) {
        int ai = i < w ? i : (a->have_tr ? (i < 2 * w ? i : 2 * w - 1) : w - 1);
        int li = i < h ? i : (a->have_bl ? (i < 2 * h ? i : 2 * h - 1) : h - 1);
        if (!have_top) above[i] = have_left ? f[y * stride + x - 1] : 127;
        else above[i] = f[(y - 1) * stride + (x + ai < max_x ? x + ai : max_x)];
        if (!have_left) left[i] = have_top ? f[(y - 1) * stride + x] : 129;
        else left[i] = f[(y + li < max_y ? y + li : max_y) * stride + x - 1];
    }
    int tl;
    if (have_top && have_left) tl = f[(y - 1) * stride + x - 1];
    else if (have_top) tl = f[(y - 1) * stride + x];
    else if (have_left) tl = f[y * stride + x - 1];
    else tl = 128;
    above[-1] = left[-1] = tl;

    if (a->filter_intra) {
        /* recursive filter intra (7.11.2.3): 4x2 cells, each from 7 neighbours */
        int pred[64][64];
        const int8_t(*taps)[7] = filter_intra_taps[a->filter_intra - 1];
        for (int i2 = 0; i2 < h / 2; i2++)
            for (int j4 = 0; j4 < w / 4; j4++) {
                int p[7];
                for (int i = 0; i < 7; i++) {
                    if (i < 5) {
                        if (i2 == 0) p[i] = above[(j4 << 2) + i - 1];
                        else if (j4 == 0 && i == 0) p[i] = left[(i2 << 1) - 1];
                        else p[i] = pred[(i2 << 1) - 1][(j4 << 2) + i - 1];
                    } else {
                        if (j4 == 0) p[i] = left[(i2 << 1) + i - 5];
                        else p[i] = pred[(i2 << 1) + i - 5][(j4 << 2) - 1];
                    }
                }
                for (int i1 = 0; i1 < 2; i1++)
                    for (int j1 = 0; j1 < 4; j1++) {
                        int pr = 0;
                        for (int i = 0; i < 7; i++) pr += taps[(i1 << 2) + j1][i] * p[i];
                        /* Round2Signed(pr, 4) */
                        pr = pr >= 0 ? (pr + 8) >> 4 : -((-pr + 8) >> 4);
                        pred[(i2 << 1) + i1][(j4 << 2) + j1] = clip_px(pr);
                    }
            }
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) dst[i * dstride + j] = (uint8_t)pred[i][j];
        return;
    }

    const int mode = a->mode;
    const int directional = mode >= EA1_V_PRED && mode <= EA1_D67_PRED;
    const int p_angle = directional ? mode_to_angle[mode] + 3 * a->angle_delta : 0;
    if (directional && p_angle != 90 && p_angle != 180) {
        int up_above = 0, up_left = 0;
        if (a->edge_filter) {
            if (p_angle > 90 && p_angle < 180 && w + h >= 24)
                above[-1] = left[-1] = (left[0] * 5 + above[-1] * 6 + above[0] * 5 + 8) >> 4;
            if (have_top) {
                const int strength = edge_filter_strength(w, h, a->smooth_edge, p_angle - 90);
                const int num_px = (w < max_x - x + 1 ? w : max_x - x + 1) + (p_angle < 90 ? h : 0) + 1;
                edge_filter(above, num_px, strength);
            }
            if (have_left) {
                const int strength = edge_filter_strength(w, h, a->smooth_edge, p_angle - 180);
                const int num_px = (h < max_y - y + 1 ? h : max_y - y + 1) + (p_angle > 180 ? w : 0) + 1;
                edge_filter(left, num_px, strength);
            }
            up_above = use_upsample(w, h, a->smooth_edge, p_angle - 90);
            if (up_above) edge_upsample(above, w + (p_angle < 90 ? h : 0));
            up_left = use_upsample(w, h, a->smooth_edge, p_angle - 180);
            if (up_left) edge_upsample(left, h + (p_angle > 180 ? w : 0));
        }
        int dx = 0, dy = 0;
        if (p_angle < 90) dx = dr_intra_derivative[p_angle];
        else if (p_angle < 180) dx = dr_intra_derivative[180 - p_angle];
        if (p_angle > 90 && p_angle < 180) dy = dr_intra_derivative[p_angle - 90];
        else if (p_angle > 180) dy = dr_intra_derivative[270 - p_angle];
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int v;
                if (p_angle < 90) {
                    const int idx = (i + 1) * dx;
                    const int base = (idx >> (6 - up_above)) + (j << up_above);
                    const int shift = ((idx << up_above) >> 1) & 0x1F;
                    const int max_base = (w + h - 1) << up_above;
                    v = base < max_base ? (above[base] * (32 - shift) + above[base + 1] * shift + 16) >> 5 : above[max_base];
                } else if (p_angle < 180) {
                    int idx = (j << 6) - (i + 1) * dx;
                    int base = idx >> (6 - up_above);
                    if (base >= -(1 << up_above)) {
                        const int shift = ((idx * (1 << up_above)) >> 1) & 0x1F;
                        v = (above[base] * (32 - shift) + above[base + 1] * shift + 16) >> 5;
                    } else {
                        idx = (i << 6) - (j + 1) * dy;
                        base = idx >> (6 - up_left);
                        const int shift = ((idx * (1 << up_left)) >> 1) & 0x1F;
                        v = (left[base] * (32 - shift) + left[base + 1] * shift + 16) >> 5;
                    }
                } else {
                    const int idx = (j + 1) * dy;
                    const int base = (idx >> (6 - up_left)) + (i << up_left);
                    const int shift = ((idx << up_left) >> 1) & 0x1F;
                    v = (left[base] * (32 - shift) + left[base + 1] * shift + 16) >> 5;
                }
                dst[i * dstride + j] = (uint8_t)v;
            }
        return;
    }

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
            case EA1_V_PRED: v = above[j]; break;   /* also directional with angle 90 */
            case EA1_H_PRED: v = left[i]; break;    /* angle 180 */
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
            case EA1_SMOOTH_H_PRED: v = (wx[j] * left[i] + (256 - wx[j]) * above[w - 1] + 128) >> 8; break;
            default: /* directional mode whose angle came out as exactly 90 or 180 */
                v = p_angle == 90 ? above[j] : left[i];
                break;
            }
            dst[i * dstride + j] = (uint8_t)v;
        }
}

/* ------------------------------------------------------------------------------------------
 * Reconstruction driver (dav1d src/recon_tmpl.c): blocks in coded order.  Inter blocks: predict
 * all planes, then add the residual of every transform block.  Intra blocks: per transform
 * block, predict from the already reconstructed neighbourhood, then add its residual.
 * ------------------------------------------------------------------------------------------ */
static const uint8_t ii_weights_1d[128] = {EA1_II_WEIGHTS_1D_VALUES};
static const uint8_t obmc_mask_tab[62] = {EA1_OBMC_MASK_VALUES};

/* 4:2:0 chroma ownership: a block with a 4-sample side carries the chroma of its 8x8 luma area
 * only if it is the last one in that direction */
static int block_has_chroma(const ea1_block *b) { return (!ssx(1) || (b->x4 & 1) || b->bwl > 2) && (!ssy(1) || (b->y4 & 1) || b->bhl > 2); }
/* the block's area in plane pl (chroma: from the 8-aligned luma position, at least 4x4) */
static void block_rect(const ea1_block *b, int pl, int *x, int *y, int *w, int *h)
{
    const int bw = 1 << b->bwl, bh = 1 << b->bhl;
    if (pl == 0) {
        *x = b->x4 * 4;
        *y = b->y4 * 4;
        *w = bw;
        *h = bh;
    } else {
        /* a subsampled direction: from the 8-aligned luma position, at least 4 samples; an unsubsampled one: the luma geometry */
        *x = ssx(1) ? (b->x4 & ~1) * 2 : b->x4 * 4;
        *y = ssy(1) ? (b->y4 & ~1) * 2 : b->y4 * 4;
        *w = ssx(1) ? (bw < 8 ? 8 : bw) / 2 : bw;
        *h = ssy(1) ? (bh < 8 ? 8 : bh) / 2 : bh;
    }
}
/* a 4-sample side in a direction the chroma planes are subsampled in: the chroma block is that of the 8-sample area */
static int block_sub8(const ea1_block *b) { return (ssx(1) && b->bwl == 2) || (ssy(1) && b->bhl == 2); }
/* edge availability of the block in plane pl: blocks with a 4-sample side have separate chroma flags */
static unsigned block_edge_flags(const ea1_block *b, int pl)
{
    if (pl == 0 || !block_sub8(b)) return b->flags & (EA1_BLK_HAVE_LEFT | EA1_BLK_HAVE_TOP | EA1_BLK_HAVE_TOP_RIGHT | EA1_BLK_HAVE_BOTTOM_LEFT);
    return ((b->flags & EA1_BLK_CHROMA_HAVE_LEFT) ? EA1_BLK_HAVE_LEFT : 0) | ((b->flags & EA1_BLK_CHROMA_HAVE_TOP) ? EA1_BLK_HAVE_TOP : 0) |
           ((b->flags & EA1_BLK_CHROMA_HAVE_TOP_RIGHT) ? EA1_BLK_HAVE_TOP_RIGHT : 0) |
           ((b->flags & EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT) ? EA1_BLK_HAVE_BOTTOM_LEFT : 0);
}

static int is_smooth_mode(int m) { return m == EA1_SMOOTH_PRED || m == EA1_SMOOTH_V_PRED || m == EA1_SMOOTH_H_PRED; }

/* get_filter_type (7.11.2): does the block above or left use a smooth mode in this plane class */
static int smooth_neighbour(const ea1_frame_desc *d, const ea1_block *blocks, const uint32_t *map, const ea1_block *b,
                            int pl)
{
    const int w4 = d->width / 4;
    int r = 0;
    const ea1_block *nb[2] = {NULL, NULL};
    /* luma: the blocks above / left of the block's first 4x4 unit; chroma: of its chroma block */
    const int ux = ssx(pl) ? (b->x4 & ~1) : b->x4, uy = ssy(pl) ? (b->y4 & ~1) : b->y4;
    const unsigned ef = block_edge_flags(b, pl);
    if (ef & EA1_BLK_HAVE_TOP) nb[0] = &blocks[map[(uy - 1) * w4 + (ssx(pl) ? ux + 1 : ux)]];
    if (ef & EA1_BLK_HAVE_LEFT) nb[1] = &blocks[map[(ssy(pl) ? uy + 1 : uy) * w4 + ux - 1]];
    for (int k = 0; k < 2; k++) {
        if (!nb[k] || (nb[k]->flags & EA1_BLK_INTER)) continue;
        r |= is_smooth_mode(pl == 0 ? nb[k]->mode_y : nb[k]->mode_uv);
    }
    return r;
}

static void inter_block(const ea1_frame_desc *d, const ea1_block *blocks, const uint8_t *aux, const uint32_t *map, const ea1_block *b,
                        const pic_t refs[EA1_NUM_REF_SLOTS], pic_t *cur, int32_t *p0, int32_t *p1, uint8_t *mask)
{
    const int bw = 1 << b->bwl, bh = 1 << b->bhl;
    const int fh = b->mode_y & 3, fv = (b->mode_y >> 2) & 3;
    const int w4 = d->width / 4;
    for (int pl = 0; pl < 3; pl++) {
        const int sx_ = ssx(pl), sy_ = ssy(pl);
        if (pl && !block_has_chroma(b)) break;
        int x, y, w, h;
        block_rect(b, pl, &x, &y, &w, &h);
        const int stride = pw(cur, pl);
        uint8_t *dst = cur->p[pl] + y * stride + x;
        if (pl && block_sub8(b)) {
            /* 7.11.3.1: chroma of an 8x8 luma area made of smaller blocks.  If one of them is intra,
             * the whole chroma block takes this (the last) block's motion; otherwise every block
             * of the area predicts its own (w/2 x h/2) piece with its own first reference. */
            const int rx4 = sx_ ? b->x4 & ~1 : b->x4, ry4 = sy_ ? b->y4 & ~1 : b->y4, n4w = (w << sx_) / 4, n4h = (h << sy_) / 4;
            int some_intra = 0;
            for (int r = 0; r < n4h; r++)
                for (int c = 0; c < n4w; c++) some_intra |= !(blocks[map[(ry4 + r) * w4 + rx4 + c]].flags & EA1_BLK_INTER);
            const int pw_ = some_intra ? w : bw >> sx_, ph_ = some_intra ? h : bh >> sy_;
            for (int py = 0, r = 0; py < h; py += ph_, r++)
                for (int px = 0, c = 0; px < w; px += pw_, c++) {
                    const ea1_block *cb = some_intra ? b : &blocks[map[(ry4 + r) * w4 + rx4 + c]];
                    predict_inter(&refs[cb->ref[0]], pl, x + px, y + py, pw_, ph_, cb->mv[0][0], cb->mv[0][1], cb->mode_y & 3,
                                  (cb->mode_y >> 2) & 3, 11, p0);
                    for (int i = 0; i < ph_; i++)
                        for (int j = 0; j < pw_; j++) dst[(py + i) * stride + px + j] = (uint8_t)clip_px(p0[i * pw_ + j]);
                }
            continue;
        }
        /* warped predictions (7.11.3.1: only when the block is at least 8x8 in this plane and the
         * matrix passes the shear test; otherwise plain translation) */
        const int32_t *wmat[2] = {NULL, NULL};
        int32_t mats[2][6];
        int wsh[2][4];
        if ((b->flags & EA1_BLK_WARP) && w >= 8 && h >= 8) {
            const uint8_t *ap = aux + b->aux_off;
            for (int r = 0; r < 2; r++)
                if (b->warp_refs & (1 << r)) {
                    memcpy(mats[r], ap, 24);
                    ap += 24;
                    if (ea1o_setup_shear(mats[r], wsh[r])) wmat[r] = mats[r];
                }
        }
        if (b->flags & EA1_BLK_COMPOUND) {
            if (wmat[0]) predict_warp(&refs[b->ref[0]], pl, x, y, w, h, wmat[0], wsh[0], 7, p0);
            else predict_inter(&refs[b->ref[0]], pl, x, y, w, h, b->mv[0][0], b->mv[0][1], fh, fv, 7, p0);
            if (wmat[1]) predict_warp(&refs[b->ref[1]], pl, x, y, w, h, wmat[1], wsh[1], 7, p1);
            else predict_inter(&refs[b->ref[1]], pl, x, y, w, h, b->mv[1][0], b->mv[1][1], fh, fv, 7, p1);
            if (pl == 0 && b->comp_type == EA1_COMP_WEDGE) ea1o_wedge_mask(bw, bh, b->comp_arg & 15, (b->comp_arg >> 4) & 1, mask);
            if (pl == 0 && b->comp_type == EA1_COMP_DIFFWTD)   /* 7.11.3.12, from the luma predictions */
                for (int i = 0; i < bh * bw; i++) {
                    const int diff = (iabs(p0[i] - p1[i]) + 8) >> 4;
                    const int m = clampi(38 + diff / 16, 0, 64);
                    mask[i] = (uint8_t)(b->comp_arg ? 64 - m : m);
                }
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) {
                    const int a = p0[i * w + j], c = p1[i * w + j];
                    int v;
                    switch (b->comp_type) {
                    case EA1_COMP_AVG: v = (a + c + 16) >> 5; break;
                    case EA1_COMP_DIST: v = (a * b->comp_arg + c * (16 - b->comp_arg) + 128) >> 8; break;
                    default: {
                        int m;
                        /* 7.11.3.14: the luma mask, averaged over the samples a chroma sample covers */
                        if (!sx_ && !sy_) m = mask[i * bw + j];
                        else if (sx_ && !sy_) m = (mask[i * bw + 2 * j] + mask[i * bw + 2 * j + 1] + 1) >> 1;
                        else
                            m = (mask[(2 * i) * bw + 2 * j] + mask[(2 * i) * bw + 2 * j + 1] + mask[(2 * i + 1) * bw + 2 * j] +
                                 mask[(2 * i + 1) * bw + 2 * j + 1] + 2) >> 2;
                        v = (m * a + (64 - m) * c + 512) >> 10;
                        break;
                    }
                    }
                    dst[i * stride + j] = (uint8_t)clip_px(v);
                }
        } else {
            if (wmat[0]) predict_warp(&refs[b->ref[0]], pl, x, y, w, h, wmat[0], wsh[0], 11, p0);
            else predict_inter(&refs[b->ref[0]], pl, x, y, w, h, b->mv[0][0], b->mv[0][1], fh, fv, 11, p0);
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) dst[i * stride + j] = (uint8_t)clip_px(p0[i * w + j]);
        }
        if (b->flags & EA1_BLK_OBMC) {
            /* 7.11.3.10: blend predictions made with the motion of the inter neighbours above
             * (pass 0) and to the left (pass 1) into the top / left part of the block.  The size
             * test -- the spec's "plane residual size >= BLOCK_8X8" in block-size enumeration order -- guards the pass from ABOVE
             * only: a 4-sample-wide chroma block still takes the pass from the left (libaom av1_skip_u4x4_pred_in_obmc: `dir == 0`;
             * dav1d obmc(): the `w + h >= 16` test sits on the first loop alone).  Round 4, found beside dav1d 1.5.3. */
            const int big_enough = w >= 8 && h >= 8 ? 1 : (w == 4 && h == 16) || (w == 16 && h == 4);
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 0 && !big_enough) continue;
                if (!(b->flags & (pass == 0 ? EA1_BLK_HAVE_TOP : EA1_BLK_HAVE_LEFT))) continue;
                const int len4 = (pass == 0 ? bw : bh) / 4;           /* block extent along the edge, 4-px luma units */
                const int l2 = pass == 0 ? b->bwl - 2 : b->bhl - 2;
                const int limit = l2 < 4 ? l2 : 4;
                const int start4 = pass == 0 ? b->x4 : b->y4;
                const int frame4 = (pass == 0 ? d_coded_w(d) : d_coded_h(d)) / 4;   /* neighbours inside the TRUE coded frame */
                int count = 0;
                for (int p4 = start4; count < limit && p4 < (start4 + len4 < frame4 ? start4 + len4 : frame4);) {
                    const int cx4 = pass == 0 ? (p4 | 1) : b->x4 - 1, cy4 = pass == 0 ? b->y4 - 1 : (p4 | 1);
                    const ea1_block *c = &blocks[map[cy4 * w4 + cx4]];
                    const int csz4 = (1 << (pass == 0 ? c->bwl : c->bhl)) / 4;
                    const int step4 = clampi(csz4, 2, 16);
                    if (c->flags & EA1_BLK_INTER) {
                        count++;
                        int pw_, ph_, px, py;
                        if (pass == 0) {
                            pw_ = (w < ((step4 * 4) >> sx_) ? w : (step4 * 4) >> sx_);
                            ph_ = (h >> 1) < (32 >> sy_) ? (h >> 1) : (32 >> sy_);
                            px = (p4 * 4) >> sx_;
                            py = y;
                        } else {
                            pw_ = (w >> 1) < (32 >> sx_) ? (w >> 1) : (32 >> sx_);
                            ph_ = (h < ((step4 * 4) >> sy_) ? h : (step4 * 4) >> sy_);
                            px = x;
                            py = (p4 * 4) >> sy_;
                        }
                        const uint8_t *om = obmc_mask_tab + (pass == 0 ? ph_ : pw_) - 2;
                        predict_inter(&refs[c->ref[0]], pl, px, py, pw_, ph_, c->mv[0][0], c->mv[0][1], c->mode_y & 3,
                                      (c->mode_y >> 2) & 3, 11, p1);
                        for (int i = 0; i < ph_; i++)
                            for (int j = 0; j < pw_; j++) {
                                const int m = pass == 0 ? om[i] : om[j];
                                uint8_t *q = cur->p[pl] + (py + i) * stride + px + j;
                                *q = (uint8_t)((m * *q + (64 - m) * clip_px(p1[i * pw_ + j]) + 32) >> 6);
                            }
                    }
                    p4 += step4;
                }
            }
        }
        if (b->flags & EA1_BLK_INTERINTRA) {
            /* 7.11.3.13 / 7.11.3.14: blend an intra prediction of the whole block into the inter one */
            static const uint8_t ii_to_intra[4] = {EA1_DC_PRED, EA1_V_PRED, EA1_H_PRED, EA1_SMOOTH_PRED};
            intra_args ia;
            memset(&ia, 0, sizeof ia);
            ia.have_left = !!(b->flags & EA1_BLK_HAVE_LEFT);
            ia.have_top = !!(b->flags & EA1_BLK_HAVE_TOP);
            ia.mode = ii_to_intra[b->ii_mode & 3];
            uint8_t ipred[32 * 32];
            predict_intra(cur, pl, x, y, w, h, &ia, ipred, w);
            if (pl == 0 && (b->ii_mode & EA1_II_WEDGE)) ea1o_wedge_mask(bw, bh, b->ii_arg & 15, 0, mask);
            const int scale = 128 / (w > h ? w : h);
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) {
                    int m;
                    if (b->ii_mode & EA1_II_WEDGE) {
                        /* 7.11.3.14: the luma mask, averaged over the samples a chroma sample covers */
                        if (!sx_ && !sy_) m = mask[i * bw + j];
                        else if (sx_ && !sy_) m = (mask[i * bw + 2 * j] + mask[i * bw + 2 * j + 1] + 1) >> 1;
                        else
                            m = (mask[(2 * i) * bw + 2 * j] + mask[(2 * i) * bw + 2 * j + 1] + mask[(2 * i + 1) * bw + 2 * j] +
                                 mask[(2 * i + 1) * bw + 2 * j + 1] + 2) >> 2;
                    } else {
                        switch (b->ii_mode & 3) {
                        case EA1_II_V: m = ii_weights_1d[i * scale]; break;
                        case EA1_II_H: m = ii_weights_1d[j * scale]; break;
                        case EA1_II_SMOOTH: m = ii_weights_1d[(i < j ? i : j) * scale]; break;
                        default: m = 32; break;
                        }
                    }
                    dst[i * stride + j] = (uint8_t)((m * ipred[i * w + j] + (64 - m) * dst[i * stride + j] + 32) >> 6);
                }
        }
    }
}

/* strip_x0 / strip_x1: only the blocks whose luma columns lie in [strip_x0, strip_x1) are reconstructed (one rank's
 * share of a tile-column split; 0 .. width = the whole frame) */
static void recon_frame(const ea1_frame_desc *d, const ea1_block *blocks, const ea1_txblock *txs, const int16_t *coefs,
                        const uint8_t *aux, const uint32_t *map, const pic_t refs[EA1_NUM_REF_SLOTS], pic_t *cur, int strip_x0, int strip_x1)
{
    int32_t *p0 = malloc(sizeof(int32_t) * 128 * 128), *p1 = malloc(sizeof(int32_t) * 128 * 128);
    uint8_t *mask = malloc(128 * 128);
    for (uint32_t bi = 0; bi < d->num_blocks; bi++) {
        const ea1_block *b = &blocks[bi];
        if (b->x4 * 4 < strip_x0 || b->x4 * 4 >= strip_x1) continue;
        const int inter = b->flags & EA1_BLK_INTER;
        if (inter) inter_block(d, blocks, aux, map, b, refs, cur, p0, p1, mask);
        const int intrabc = !inter && b->mode_y == EA1_INTRABC_PRED;
        if (intrabc) {
            /* intra block copy: the block is predicted like a single-reference inter block whose
             * reference is the current picture as reconstructed so far, with the bilinear filter
             * (AV1 spec 7.11.3.2: refIdx -1; 5.11.x: interp_filter = BILINEAR when use_intrabc) */
            for (int pl = 0; pl < 3; pl++) {
                int x, y, w, h;
                if (pl && !block_has_chroma(b)) break;   /* (a block with a 4-sample side: the last one of its 8x8 area copies the chroma block) */
                block_rect(b, pl, &x, &y, &w, &h);
                const int stride = pw(cur, pl);
                predict_inter(cur, pl, x, y, w, h, b->mv[0][0], b->mv[0][1], EA1_FILTER_BILINEAR, EA1_FILTER_BILINEAR, 11, p0);
                for (int i = 0; i < h; i++)
                    for (int j = 0; j < w; j++) cur->p[pl][(y + i) * stride + x + j] = (uint8_t)clip_px(p0[i * w + j]);
            }
        }
        for (uint32_t ti = b->first_tx; ti < b->first_tx + b->n_tx; ti++) {
            const ea1_txblock *t = &txs[ti];
            const int pl = t->plane;
            const int w = 1 << (t->size & 15), h = 1 << (t->size >> 4);
            const int x = t->x4 * 4, y = t->y4 * 4;
            const int stride = pw(cur, pl);
            uint8_t *dst = cur->p[pl] + y * stride + x;
            if (!inter && !intrabc) {
                int bx, by, bw, bh;
                block_rect(b, pl, &bx, &by, &bw, &bh);
                const unsigned ef = block_edge_flags(b, pl);
                const int pal = pl == 0 ? (b->flags & EA1_BLK_PALETTE_Y) : (b->flags & EA1_BLK_PALETTE_UV);
                if (pal) {
                    /* 7.11.4: colour index map -> palette colours */
                    const int ny = b->pal_size[0], nuv = b->pal_size[1];
                    const int lw = 1 << b->bwl, lh = 1 << b->bhl;
                    const uint8_t *colors = aux + b->aux_off + (pl == 0 ? 0 : ny + (pl - 1) * nuv);
                    const uint8_t *cmap = aux + b->aux_off + ny + 2 * nuv + (pl == 0 ? 0 : lw * lh);
                    for (int i = 0; i < h; i++)
                        for (int j = 0; j < w; j++) dst[i * stride + j] = colors[cmap[(y - by + i) * bw + (x - bx + j)]];
                } else {
                    intra_args ia;
                    memset(&ia, 0, sizeof ia);
                    ia.have_left = (ef & EA1_BLK_HAVE_LEFT) || x > bx;
                    ia.have_top = (ef & EA1_BLK_HAVE_TOP) || y > by;
                    /* above-right / below-left (w resp. h samples): decoded iff they belong to an earlier block or an
                     * earlier transform block of this one (spec: BlockDecoded) */
                    const int last_col = x + w >= bx + bw, first_row = y == by;
                    const int first_col = x == bx, last_row = y + h >= by + bh;
                    /* transform blocks are decoded in raster order inside a 64x64 luma chunk of the block (32x32 in chroma), the chunks
                     * of a 128-sample block in raster order (5.11.34 residual): across a chunk boundary the sample up and to the right
                     * is decoded iff it lies in the chunk row above, the sample down and to the left iff it lies in the same chunk row */
                    const int csw = 64 >> ssx(pl), csh = 64 >> ssy(pl);
                    const int chunk_last_col = (x + w - bx) % csw == 0, chunk_first_row = (y - by) % csh == 0;
                    const int chunk_first_col = (x - bx) % csw == 0, chunk_last_row = (y + h - by) % csh == 0;
                    ia.have_tr = first_row ? (last_col ? !!(ef & EA1_BLK_HAVE_TOP_RIGHT) : 1) : (last_col ? 0 : (chunk_last_col ? chunk_first_row : 1));
                    ia.have_bl = first_col ? (last_row ? !!(ef & EA1_BLK_HAVE_BOTTOM_LEFT) : 1) : (last_row ? 0 : (chunk_first_col ? !chunk_last_row : 0));
                    const int mode = pl ? b->mode_uv : b->mode_y;
                    ia.mode = mode == EA1_UV_CFL_PRED ? EA1_DC_PRED : mode;
                    ia.angle_delta = b->angle_delta[pl ? 1 : 0];
                    ia.edge_filter = d->intra_edge_filter;
                    ia.smooth_edge = smooth_neighbour(d, blocks, map, b, pl);
                    ia.filter_intra = pl == 0 ? b->filter_intra : 0;
                    predict_intra(cur, pl, x, y, w, h, &ia, dst, stride);
                    if (pl && mode == EA1_UV_CFL_PRED) {
                        /* 7.11.5 chroma from luma: scaled AC of the reconstructed luma on top of DC */
                        const int alpha = b->cfl_alpha[pl - 1];
                        const uint8_t *luma = cur->p[0];
                        const int ls = pw(cur, 0);
                        int L[32 * 32], avg = 0;
                        /* MaxLumaW / MaxLumaH (5.11.35): the end of the block's last CODED luma transform block -- one that starts
                         * beyond the true coded frame is not coded (the list carries it without coefficients, and what is predicted
                         * there is never looked at): the luma beyond repeats the last coded column / row */
                        int max_lw = (b->x4 * 4) + (1 << b->bwl), max_lh = (b->y4 * 4) + (1 << b->bhl);
                        for (uint32_t k = b->first_tx; k < b->first_tx + b->n_tx; k++)
                            if (txs[k].plane == 0) {
                                const int tw = 1 << (txs[k].size & 15), th = 1 << (txs[k].size >> 4);
                                const int in_w = g_coded_w - b->x4 * 4, in_h = g_coded_h - b->y4 * 4;
                                if (in_w < (1 << b->bwl)) max_lw = b->x4 * 4 + (in_w + tw - 1) / tw * tw;
                                if (in_h < (1 << b->bhl)) max_lh = b->y4 * 4 + (in_h + th - 1) / th * th;
                                break;   /* the luma grid of an intra block is uniform */
                            }
                        const int csx = ssx(1), csy = ssy(1);
                        for (int i = 0; i < h; i++)
                            for (int j = 0; j < w; j++) {
                                /* 7.11.5: the luma samples a chroma sample covers, summed and scaled to 8 x their mean */
                                int ly = (y + i) << csy, lx = (x + j) << csx;
                                if (lx > max_lw - (1 << csx)) lx = max_lw - (1 << csx);
                                if (ly > max_lh - (1 << csy)) ly = max_lh - (1 << csy);
                                int t = 0;
                                for (int dy = 0; dy <= csy; dy++)
                                    for (int dx = 0; dx <= csx; dx++) t += luma[(ly + dy) * ls + lx + dx];
                                const int v8 = t << (3 - csx - csy);
                                L[i * w + j] = v8;
                                avg += v8;
                            }
                        int lw2 = 0, lh2 = 0;
                        while ((1 << lw2) < w) lw2++;
                        while ((1 << lh2) < h) lh2++;
                        avg = (avg + ((1 << (lw2 + lh2)) >> 1)) >> (lw2 + lh2);
                        for (int i = 0; i < h; i++)
                            for (int j = 0; j < w; j++) {
                                const int v = alpha * (L[i * w + j] - avg);
                                const int sc = v >= 0 ? (v + 32) >> 6 : -((-v + 32) >> 6);
                                dst[i * stride + j] = (uint8_t)clip_px(dst[i * stride + j] + sc);
                            }
                    }
                }
            }
            if (t->coef_off != EA1_NO_COEFS) {
                if (d->coef_layout == EA1_COEFS_DIAGONAL) {
                    /* include/easyav1_amd_video.h EA1_COEFS_DIAGONAL: the positions with x + y <= D in anti-diagonal
                     * order (y ascending inside a diagonal); what lies beyond the stated corner is not part of the frame */
                    int16_t dense[32 * 32];
                    const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32, n = 1 << (2 + (t->nz_log2 & 3)), D = t->nz_log2 >> 2;
                    const int cw = n < sw ? n : sw, ch = n < sh ? n : sh;
                    memset(dense, 0, sizeof dense);
                    uint32_t at = t->coef_off;
                    for (int dd = 0; dd <= D && dd <= sw + sh - 2; dd++)
                        for (int yy = dd - (sw - 1) > 0 ? dd - (sw - 1) : 0; yy <= (dd < sh - 1 ? dd : sh - 1); yy++, at++) {
                            const int xx = dd - yy;
                            if (xx < cw && yy < ch) dense[yy * sw + xx] = coefs[at];
                        }
                    ea1o_itx_add(dst, stride, w, h, t->txtp, dense);
                } else if (t->nz_log2 >= 2 && t->nz_log2 <= 5) {
                    /* include/easyav1_amd_video.h, ea1_txblock.nz_log2: the parser's bound on the coded coefficients;
                     * whatever the arena holds beyond that top-left corner is not part of the frame.  With
                     * ea1_frame_desc.coef_layout == EA1_COEFS_CORNER the arena holds that corner and nothing else. */
                    int16_t dense[32 * 32];
                    const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32, n = 1 << t->nz_log2;
                    const int cw = n < sw ? n : sw, ch = n < sh ? n : sh;
                    const int cstride = d->coef_layout == EA1_COEFS_CORNER ? cw : sw;
                    for (int yy = 0; yy < sh; yy++)
                        for (int xx = 0; xx < sw; xx++) dense[yy * sw + xx] = (xx < cw && yy < ch) ? coefs[t->coef_off + yy * cstride + xx] : 0;
                    ea1o_itx_add(dst, stride, w, h, t->txtp, dense);
                } else {
                    ea1o_itx_add(dst, stride, w, h, t->txtp, coefs + t->coef_off);
                }
            }
        }
    }
    free(p0);
    free(p1);
    free(mask);
}

/* block map: index of the block covering each 4x4 luma unit */
static uint32_t *build_block_map(const ea1_frame_desc *d, const ea1_block *blocks)
{
    const int w4 = d->width / 4, h4 = d->height / 4;
    uint32_t *map = malloc(sizeof(uint32_t) * (size_t)w4 * h4);
    memset(map, 0xFF, sizeof(uint32_t) * (size_t)w4 * h4);
    for (uint32_t bi = 0; bi < d->num_blocks; bi++) {
        const ea1_block *b = &blocks[bi];
        const int n4w = (1 << b->bwl) / 4, n4h = (1 << b->bhl) / 4;
        for (int j = 0; j < n4h; j++)
            for (int i = 0; i < n4w; i++) map[(b->y4 + j) * w4 + b->x4 + i] = bi;
    }
    return map;
}

/* per plane, per 4x4-sample unit: index of the transform block covering it */
static uint32_t *build_tx_map(const ea1_frame_desc *d, const ea1_txblock *txs, int pl)
{
    const int w4 = (int)(d->width >> ssx(pl)) / 4, h4 = (int)(d->height >> ssy(pl)) / 4;
    uint32_t *map = malloc(sizeof(uint32_t) * (size_t)w4 * h4);
    memset(map, 0xFF, sizeof(uint32_t) * (size_t)w4 * h4);
    for (uint32_t ti = 0; ti < d->num_tx; ti++) {
        const ea1_txblock *t = &txs[ti];
        if (t->plane != pl) continue;
        const int tw4 = (1 << (t->size & 15)) / 4, th4 = (1 << (t->size >> 4)) / 4;
        for (int j = 0; j < th4; j++)
            for (int i = 0; i < tw4; i++) map[(t->y4 + j) * w4 + t->x4 + i] = ti;
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

/* 7.14.2: an edge is filtered where a transform edge lies, except inside skipped inter blocks;
 * 7.14.3: filter size from the smaller of the two transform sizes across the edge */
static void deblock_frame(const ea1_frame_desc *d, const ea1_block *blocks, const uint32_t *map, const ea1_txblock *txs,
                          uint32_t *const txmap[3], pic_t *f)
{
    if (!d->lf_level[0] && !d->lf_level[1]) return;
    const int bw4 = d->width / 4;
    for (int pl = 0; pl < 3; pl++) {
        if (pl > 0 && !d->lf_level[1 + pl]) continue;
        const int W = pw(f, pl), H = ph(f, pl);
        const int w4 = W / 4;
        /* block of a 4x4 unit of this plane: chroma units belong to the block that owns the chroma
         * of the 8x8 luma area, i.e. the one at its last (odd) 4x4 luma unit */
#define BLOCK_AT(ux, uy) (&blocks[map[(ssy(pl) ? 2 * (uy) + 1 : (uy)) * bw4 + (ssx(pl) ? 2 * (ux) + 1 : (ux))]])
        for (int pass = 0; pass < 2; pass++) {
            const int frame_lvl = pl == 0 ? d->lf_level[pass] : d->lf_level[1 + pl];
            const int li = pl == 0 ? pass : 1 + pl;   /* index into ea1_block.lf_level */
            /* a luma pass whose frame level is 0 still runs when the levels are per block: deltas may raise a block above 0
             * (7.14.4; dav1d's calc_lf_values zeroes the table only when both luma levels are 0) */
            if (!frame_lvl && !(pl == 0 && d->lf_per_block)) continue;
            /* 7.14.2 onScreen: a 4x4 unit whose LUMA position is at or beyond the frame's true size is not filtered */
            const int vis_w = d_vis_w(d), vis_h = d_vis_h(d);
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    /* (x, y) is q0; the edge lies between it and the previous sample */
                    const int pos = pass == 0 ? x : y;
                    if (pos == 0) continue;
                    if ((((x & ~3) << ssx(pl)) >= vis_w) || (((y & ~3) << ssy(pl)) >= vis_h)) continue;
                    const ea1_txblock *t = &txs[txmap[pl][(y / 4) * w4 + x / 4]];
                    const int tpos = (pass == 0 ? t->x4 : t->y4) * 4;
                    if (pos != tpos) continue; /* not a transform edge */
                    const ea1_block *b = BLOCK_AT(x / 4, y / 4);
                    int bx, by, bw_, bh_;
                    block_rect(b, pl, &bx, &by, &bw_, &bh_);
                    const int is_block_edge = pos == (pass == 0 ? bx : by);
                    if (!is_block_edge && (b->flags & EA1_BLK_SKIP) && (b->flags & EA1_BLK_INTER)) continue;
                    const ea1_txblock *pt = pass == 0 ? &txs[txmap[pl][(y / 4) * w4 + (x - 1) / 4]]
                                                      : &txs[txmap[pl][((y - 1) / 4) * w4 + x / 4]];
                    const int cur_dim = 1 << (pass == 0 ? (t->size & 15) : (t->size >> 4));
                    const int prev_dim = 1 << (pass == 0 ? (pt->size & 15) : (pt->size >> 4));
                    const int base = cur_dim < prev_dim ? cur_dim : prev_dim;
                    const int cap = pl == 0 ? 16 : 8;
                    const int fs = base < cap ? base : cap;
                    /* 7.14.4: the level of the current block, or of the block across the edge if that is 0 */
                    int lvl = frame_lvl;
                    if (d->lf_per_block) {
                        lvl = b->lf_level[li];
                        if (!lvl) {
                            const int qx = pass == 0 ? x - 1 : x, qy = pass == 0 ? y : y - 1;
                            lvl = BLOCK_AT(qx / 4, qy / 4)->lf_level[li];
                        }
                        if (!lvl) continue;
                    }
                    filter_edge_sample(f->p[pl] + y * W + x, pass == 0 ? 1 : W, pl, fs, lvl, d->lf_sharpness);
                }
        }
#undef BLOCK_AT
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
    const int W = pw(in, pl);
    /* CdefAvailable: samples of the TRUE coded plane */
    const int AW = g_coded_w >> ssx(pl), H = g_coded_h >> ssy(pl);
    const uint8_t *src = in->p[pl];
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) {
            const int x = src[(y0 + i) * W + x0 + j];
            int sum = 0, mx = x, mn = x;
            for (int k = 0; k < 2; k++)
                for (int sign = -1; sign <= 1; sign += 2) {
                    int yy = y0 + i + sign * cdef_directions[dir][k][0];
                    int xx = x0 + j + sign * cdef_directions[dir][k][1];
                    if (yy >= 0 && yy < H && xx >= 0 && xx < AW) {
                        const int p = src[yy * W + xx];
                        sum += cdef_pri_taps[pri & 1][k] * constrain(p - x, pri, damping);
                        if (p > mx) mx = p;
                        if (p < mn) mn = p;
                    }
                    for (int doff = -2; doff <= 2; doff += 4) {
                        const int d2 = (dir + doff) & 7;
                        yy = y0 + i + sign * cdef_directions[d2][k][0];
                        xx = x0 + j + sign * cdef_directions[d2][k][1];
                        if (yy >= 0 && yy < H && xx >= 0 && xx < AW) {
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
    const int w8 = d_coded_w(d) / 8, h8 = d_coded_h(d) / 8;   /* 8x8 blocks of the TRUE coded frame */
    const int sbw = (d->width + 63) / 64;
    for (int y8 = 0; y8 < h8; y8++)
        for (int x8 = 0; x8 < w8; x8++) {
            const int idx = cdef_idx[(y8 / 8) * sbw + x8 / 8];
            if (idx == 0xFF) continue;
            /* an 8x8 is left alone when every block inside it is skipped */
            const int bw4 = d->width / 4;
            int all_skip = 1;
            for (int k = 0; k < 4; k++) all_skip &= !!(blocks[map[(2 * y8 + (k >> 1)) * bw4 + 2 * x8 + (k & 1)]].flags & EA1_BLK_SKIP);
            if (all_skip) continue;
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
            /* 7.15.1: the luma direction, translated when the chroma planes are subsampled in one direction only (Cdef_Uv_Dir: 4:2:2) */
            static const uint8_t uv_dir_422[8] = {7, 0, 2, 4, 5, 6, 6, 6};
            dir = pri == 0 ? 0 : (ssx(1) && !ssy(1) ? uv_dir_422[ydir] : ydir);
            for (int pl = 1; pl < 3; pl++)
                cdef_filter_block(in, out, pl, (x8 * 8) >> ssx(pl), (y8 * 8) >> ssy(pl), 8 >> ssx(pl), 8 >> ssy(pl), pri, sec, d->cdef_damping - 1, dir);
        }
}

/* ------------------------------------------------------------------------------------------
 * Loop restoration (AV1 spec 7.17)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const pic_t *deblocked, *cdef;
    int pl, W, H;      /* VISIBLE plane size: samples are replicated beyond it */
    int stripe_start, stripe_end;
    int stride;        /* row stride of the planes (coded plane width) */
} lr_src_t;

/* get_source_sample: CDEF output inside the stripe, deblocked (pre-CDEF) rows just outside it */
static inline int lr_sample(const lr_src_t *s, int x, int y)
{
    x = clampi(x, 0, s->W - 1);
    y = clampi(y, 0, s->H - 1);
    if (y < s->stripe_start) {
        y = y > s->stripe_start - 2 ? y : s->stripe_start - 2;
        return s->deblocked->p[s->pl][y * s->stride + x];
    }
    if (y > s->stripe_end) {
        y = y < s->stripe_end + 2 ? y : s->stripe_end + 2;
        return s->deblocked->p[s->pl][y * s->stride + x];
    }
    return s->cdef->p[s->pl][y * s->stride + x];
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
            /* AV1 spec 7.17.4: offset = 1 << (BitDepth + FILTER_BITS - InterRound0 - 1) = 2048, limit = (1 << (BitDepth + 1 +
             * FILTER_BITS - InterRound0)) - 1 = 8191; the intermediate is Clip3(-offset, limit - offset, v) = [-2048, 6143] */
            mid[r * w + c] = clampi((sum + 4) >> 3, -2048, 6143);
        }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            int sum = 0;
            for (int t = 0; t < 7; t++) sum += vf[t] * mid[(r + t) * w + c];
            dst[(y + r) * s->stride + x + c] = (uint8_t)clip_px((sum + 1024) >> 11);
        }
    free(mid);
}

/* box filter of one pass over the (w x h) block at (x, y); flt gets the filtered output */
static void sgr_box(const lr_src_t *s, int x, int y, int w, int h, int pass, int r, int eps, int32_t *flt)
{
    const int n = (2 * r + 1) * (2 * r + 1);
    /* 7.17.3: z = Round2(p * s, SGRPROJ_MTABLE_BITS) with s = Sgr_Params[set][2 * pass + 1] AS TABULATED (the table holds the scale, not
     * an epsilon to derive it from: found in round 4 on a libaom stream decoded beside dav1d 1.5.3) */
    const unsigned sc = (unsigned)eps;
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
            const int v = a * s->cdef->p[s->pl][(y + i) * s->stride + x + j] + b;
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
            const int uu = s->cdef->p[s->pl][(y + i) * s->stride + x + j] << 4;
            int v = w1 * uu;
            v += w0 * (prm[0] ? f0[i * w + j] : uu);
            v += w2 * (prm[2] ? f1[i * w + j] : uu);
            dst[(y + i) * s->stride + x + j] = (uint8_t)clip_px((v + (1 << 10)) >> 11);
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
        /* restoration covers the visible plane; the rest of the coded area keeps the CDEF output */
        const int W = pvw(cdef, pl), H = pvh(cdef, pl), ss = ssy(pl), stride = pw(cdef, pl);   /* (stripes and unit rows: the vertical subsampling) */
        memcpy(out->p[pl], cdef->p[pl], (size_t)stride * ph(cdef, pl));
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
            lr_src_t s = {deblocked, cdef, pl, W, H, 0, 0, stride};
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
    p->w = p->vw = w;
    p->h = p->vh = h;
    p->p[0] = malloc((size_t)w * h);
    p->p[1] = malloc((size_t)(w >> ssx(1)) * (h >> ssy(1)));
    p->p[2] = malloc((size_t)(w >> ssx(1)) * (h >> ssy(1)));
}
/* ---- super-resolution (AV1 spec 7.16): horizontal upscaling of a plane with the 8-tap, 64-phase Upscale_Filter ---- */
static __thread int16_t g_upscale[64][8];
void ea1o_set_upscale_filter(const int16_t *taps)
{
    if (taps) memcpy(g_upscale, taps, sizeof g_upscale);
    else memset(g_upscale, 0, sizeof g_upscale);
}
/* down_w / up_w: Round2(FrameWidth, subX) / Round2(UpscaledWidth, subX); last_x: the last column of the CODED downscaled plane */
static void upscale_plane(const uint8_t *src, int sstride, int down_w, int last_x, int rows, uint8_t *dst, int dstride, int up_w)
{
    const int step = ((down_w << 14) + up_w / 2) / up_w;
    const int err = up_w * step - (down_w << 14);
    /* (the division of a negative number truncates towards zero, as in the specification's integer arithmetic) */
    const int initial = ((-((up_w - down_w) << 13) + up_w / 2) / up_w + (1 << 7) - err / 2) & 16383;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < up_w; x++) {
            const int src_x = -(1 << 14) + initial + x * step;
            const int px = src_x >> 14, ph_ = (src_x & 16383) >> 8;
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += g_upscale[ph_][k] * src[y * sstride + clampi(px + k - 3, 0, last_x)];
            dst[y * dstride + x] = (uint8_t)clip_px((sum + 64) >> 7);
        }
}

/* (for the known-answer tests of the upscaling alone) */
void ea1o_upscale_rows(const uint8_t *src, int sstride, int down_w, int last_x, int rows, uint8_t *dst, int dstride, int up_w)
{
    upscale_plane(src, sstride, down_w, last_x, rows, dst, dstride, up_w);
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

int ea1o_decode_frame(const ea1_frame_desc *d, const ea1_block *blocks, const ea1_txblock *txs, const int16_t *coefs,
                      const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3], uint8_t *const refs[EA1_NUM_REF_SLOTS][3], unsigned stages,
                      uint8_t *const out_recon[3], uint8_t *const out_deblock[3], uint8_t *const out_cdef[3],
                      uint8_t *const out_final[3])
{
    pic_t rp[EA1_NUM_REF_SLOTS];
    g_frame_w = (int)d->width;
    g_frame_h = (int)d->height;
    g_coded_w = d_coded_w(d);
    g_coded_h = d_coded_h(d);
    g_frame_vw = d_vis_w(d);
    g_frame_vh = d_vis_h(d);
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) {
        rp[i].w = d->width;
        rp[i].h = d->height;
        if (g_ref_sizes[i][0] && g_ref_sizes[i][1]) {   /* a reference of its own size (ea1o_set_ref_sizes), tightly packed planes */
            rp[i].w = (int)g_ref_sizes[i][0];
            rp[i].h = (int)g_ref_sizes[i][1];
        }
        const int same = rp[i].w == (int)d->width && rp[i].h == (int)d->height;
        /* crop byte of the picture in the slot | its pad byte << 8 (ea1_frame_desc.crop / .pad of the frame that made it) */
        const int rcrop = g_ref_crops[i] ? g_ref_crops[i] - 1 : 0, rpad = (rcrop >> 8) & 255;
        rp[i].vw = (!g_ref_crops[i] && same) ? d_vis_w(d) : rp[i].w - 8 * (rpad & 15) - (rcrop & 15);
        rp[i].vh = (!g_ref_crops[i] && same) ? d_vis_h(d) : rp[i].h - 8 * (rpad >> 4) - ((rcrop >> 4) & 15);
        for (int pl = 0; pl < 3; pl++) rp[i].p[pl] = refs ? refs[i][pl] : NULL;
    }
    pic_t cur, cd, fin;
    pic_alloc(&cur, d->width, d->height);
    pic_alloc(&cd, d->width, d->height);
    pic_alloc(&fin, d->width, d->height);
    cur.vw = cd.vw = fin.vw = d_vis_w(d);
    cur.vh = cd.vh = fin.vh = d_vis_h(d);
    for (int pl = 0; pl < 3; pl++) memset(cur.p[pl], 0, (size_t)pw(&cur, pl) * ph(&cur, pl));
    uint32_t *map = build_block_map(d, blocks);
    uint32_t *txmap[3];
    for (int pl = 0; pl < 3; pl++) txmap[pl] = build_tx_map(d, txs, pl);
    recon_frame(d, blocks, txs, coefs, aux, map, rp, &cur, 0, (int)d->width);
    pic_export(&cur, out_recon);
    if (stages & EA1_STAGE_DEBLOCK) deblock_frame(d, blocks, map, txs, txmap, &cur);
    pic_export(&cur, out_deblock);
    static const ea1_frame_desc zero_desc;
    const ea1_frame_desc *dc = (stages & EA1_STAGE_CDEF) ? d : &zero_desc;
    if (stages & EA1_STAGE_CDEF) cdef_frame(d, blocks, map, cdef_idx, &cur, &cd);
    else
        for (int pl = 0; pl < 3; pl++) memcpy(cd.p[pl], cur.p[pl], (size_t)pw(&cur, pl) * ph(&cur, pl));
    (void)dc;
    pic_export(&cd, out_cdef);
    if (d->upscaled_width) {
        /* super-resolution: the deblocked and the CDEF-filtered picture are upscaled, restoration runs on the upscaled pair, the
         * picture that comes out is upscaled_width visible samples wide inside (upscaled_width + 7) & ~7 stored ones */
        pic_t ucur, ucd, ufin;
        const int uw = ((int)d->upscaled_width + 7) & ~7;
        pic_alloc(&ucur, uw, d->height);
        pic_alloc(&ucd, uw, d->height);
        pic_alloc(&ufin, uw, d->height);
        ucur.vw = ucd.vw = ufin.vw = (int)d->upscaled_width;
        ucur.vh = ucd.vh = ufin.vh = d_vis_h(d);
        for (int pl = 0; pl < 3; pl++) {
            const int ss = ssx(pl), down = (d_vis_w(d) + ss) >> ss, up = ((int)d->upscaled_width + ss) >> ss, last = (d_coded_w(d) >> ss) - 1;
            memset(ucur.p[pl], 0, (size_t)pw(&ucur, pl) * ph(&ucur, pl));
            memset(ucd.p[pl], 0, (size_t)pw(&ucd, pl) * ph(&ucd, pl));
            upscale_plane(cur.p[pl], pw(&cur, pl), down, last, ph(&cur, pl), ucur.p[pl], pw(&ucur, pl), up);
            upscale_plane(cd.p[pl], pw(&cd, pl), down, last, ph(&cd, pl), ucd.p[pl], pw(&ucd, pl), up);
        }
        if (stages & EA1_STAGE_LR) lr_frame(d, lr, &ucur, &ucd, &ufin);
        else
            for (int pl = 0; pl < 3; pl++) memcpy(ufin.p[pl], ucd.p[pl], (size_t)pw(&ucd, pl) * ph(&ucd, pl));
        pic_export(&ufin, out_final);
        pic_free(&ucur);
        pic_free(&ucd);
        pic_free(&ufin);
    } else {
        if (stages & EA1_STAGE_LR) lr_frame(d, lr, &cur, &cd, &fin);
        else
            for (int pl = 0; pl < 3; pl++) memcpy(fin.p[pl], cd.p[pl], (size_t)pw(&cd, pl) * ph(&cd, pl));
        pic_export(&fin, out_final);
    }
    free(map);
    for (int pl = 0; pl < 3; pl++) free(txmap[pl]);
    pic_free(&cur);
    pic_free(&cd);
    pic_free(&fin);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Tile-column split, CPU model of one rank (include/easyav1_amd_video.h, "Tile-column split"): the two
 * compute phases between which the ranks exchange columns.  `work` are the CALLER's full-frame planes
 * (tightly packed): ea1o_strip_recon writes the strip's own blocks into them and leaves every other
 * column alone; ea1o_strip_filter runs the in-loop filters over the whole planes -- whatever the columns
 * far from the strip hold -- so that a test can check which output columns depend on which input columns.
 * ------------------------------------------------------------------------------------------ */
int ea1o_strip_recon(const ea1_frame_desc *d, const ea1_block *blocks, const ea1_txblock *txs, const int16_t *coefs, const uint8_t *aux,
                     uint8_t *const refs[EA1_NUM_REF_SLOTS][3], int strip_x0, int strip_x1, uint8_t *const work[3])
{
    pic_t rp[EA1_NUM_REF_SLOTS], cur;
    g_frame_w = (int)d->width;
    g_frame_h = (int)d->height;
    g_coded_w = d_coded_w(d);
    g_coded_h = d_coded_h(d);
    g_frame_vw = d_vis_w(d);
    g_frame_vh = d_vis_h(d);
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) {
        rp[i].w = d->width;
        rp[i].h = d->height;
        rp[i].vw = d_vis_w(d);
        rp[i].vh = d_vis_h(d);
        for (int pl = 0; pl < 3; pl++) rp[i].p[pl] = refs ? refs[i][pl] : NULL;
    }
    cur = rp[0];
    for (int pl = 0; pl < 3; pl++) cur.p[pl] = work[pl];
    uint32_t *map = build_block_map(d, blocks);
    recon_frame(d, blocks, txs, coefs, aux, map, rp, &cur, strip_x0, strip_x1);
    free(map);
    return 0;
}

int ea1o_strip_filter(const ea1_frame_desc *d, const ea1_block *blocks, const ea1_txblock *txs, const uint8_t *cdef_idx,
                      const ea1_lr_unit *const lr[3], unsigned stages, uint8_t *const work[3], uint8_t *const out_final[3])
{
    pic_t cur, cd, fin;
    pic_alloc(&cur, d->width, d->height);
    pic_alloc(&cd, d->width, d->height);
    pic_alloc(&fin, d->width, d->height);
    cur.vw = cd.vw = fin.vw = d_vis_w(d);
    cur.vh = cd.vh = fin.vh = d_vis_h(d);
    for (int pl = 0; pl < 3; pl++) memcpy(cur.p[pl], work[pl], (size_t)pw(&cur, pl) * ph(&cur, pl));
    uint32_t *map = build_block_map(d, blocks);
    uint32_t *txmap[3];
    for (int pl = 0; pl < 3; pl++) txmap[pl] = build_tx_map(d, txs, pl);
    if (stages & EA1_STAGE_DEBLOCK) deblock_frame(d, blocks, map, txs, txmap, &cur);
    if (stages & EA1_STAGE_CDEF) cdef_frame(d, blocks, map, cdef_idx, &cur, &cd);
    else
        for (int pl = 0; pl < 3; pl++) memcpy(cd.p[pl], cur.p[pl], (size_t)pw(&cur, pl) * ph(&cur, pl));
    if (stages & EA1_STAGE_LR) lr_frame(d, lr, &cur, &cd, &fin);
    else
        for (int pl = 0; pl < 3; pl++) memcpy(fin.p[pl], cd.p[pl], (size_t)pw(&cd, pl) * ph(&cd, pl));
    pic_export(&fin, out_final);
    free(map);
    for (int pl = 0; pl < 3; pl++) free(txmap[pl]);
    pic_free(&cur);
    pic_free(&cd);
    pic_free(&fin);
    return 0;
}
