/*
 * oracle/av1_grain.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of AV1 film grain synthesis
 * (AV1 specification 7.18.3; dav1d src/filmgrain_tmpl.c + src/fg_apply_tmpl.c,
 * /root/reference/cmake/dav1d.cmake:1005,:1004) for 8-bit 4:2:0 pictures.  Checker for the HIP
 * kernel of easyav1_amd/csrc/film_grain.hip; nothing in the product path may call it.
 *
 * Parity status: PINNED since round 4 against dav1d 1.5.3 (the decoder the reference pins; its dav1d_* entry points out of Pillow's bundled libavif) on libaom 3.13.2 streams, bit-exact: tests/test_real_av1.py, tests/golden/make_video_fixtures.py, DESIGN.md section 2
 * (film-grain-test vectors 1..16 of libaom, 4:2:0 and 4:0:0, beside dav1d's apply_grain).  The Gaussian sequence is specification data that the caller
 * supplies (easyav1_amd/data/av1_spec_tables.bin).
 * Written in the specification's own structure (random register, grain templates, noise stripes
 * with the overlap blends, noise image, final blend), NOT in the closed per-sample form the GPU
 * kernel uses, so that agreement between the two says something.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "easyav1_amd_grain.h"

static int16_t gaussian_sequence[2048];

void ea1o_set_gaussian_sequence(const int16_t *seq) { memcpy(gaussian_sequence, seq, sizeof gaussian_sequence); }

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int round2(int v, int n) { return n ? (v + (1 << (n - 1))) >> n : v; }

/* 7.18.3.2 */
static unsigned random_register;
static int get_random_number(int bits)
{
    unsigned r = random_register;
    const unsigned bit = ((r >> 0) ^ (r >> 1) ^ (r >> 3) ^ (r >> 12)) & 1;
    r = (r >> 1) | (bit << 15);
    random_register = r;
    return (int)((r >> (16 - bits)) & ((1u << bits) - 1));
}

enum { GRAIN_MIN = -128, GRAIN_MAX = 127 };   /* 8-bit: GrainCenter 128 */

static int luma_grain[73][82], cb_grain[73][82], cr_grain[73][82];   /* (chroma templates: 38 x 44 with both subsamplings, up to 73 x 82 without) */

/* Chroma subsampling for the next ea1o_film_grain calls of this thread (4:2:0 = the default; round 5: 4:2:2 = 1, 0 and 4:4:4 = 0, 0 for the CPU checker).
 * Stored inverted so that zero-initialised thread storage means 4:2:0. */
static __thread int g_gsx_inv, g_gsy_inv;
void ea1o_grain_set_chroma_format(int ssx, int ssy) { g_gsx_inv = !ssx, g_gsy_inv = !ssy; }
#define GSX (!g_gsx_inv)
#define GSY (!g_gsy_inv)

/* 7.18.3.3 */
static void generate_grain(const ea1_film_grain *fg)
{
    int shift = 12 - 8 + fg->grain_scale_shift;
    random_register = fg->grain_seed;
    for (int y = 0; y < 73; y++)
        for (int x = 0; x < 82; x++) {
            const int g = fg->num_y_points > 0 ? gaussian_sequence[get_random_number(11)] : 0;
            luma_grain[y][x] = round2(g, shift);
        }
    const int lag = fg->ar_coeff_lag;
    shift = fg->ar_coeff_shift;
    for (int y = 3; y < 73; y++)
        for (int x = 3; x < 82 - 3; x++) {
            int sum = 0, pos = 0;
            for (int dr = -lag; dr <= 0; dr++)
                for (int dc = -lag; dc <= lag; dc++) {
                    if (dr == 0 && dc == 0) break;
                    sum += fg->ar_coeffs_y[pos++] * luma_grain[y + dr][x + dc];
                }
            luma_grain[y][x] = clip3(GRAIN_MIN, GRAIN_MAX, luma_grain[y][x] + round2(sum, shift));
        }
    shift = 12 - 8 + fg->grain_scale_shift;
    const int sx = GSX, sy = GSY, chroma_w = sx ? 44 : 82, chroma_h = sy ? 38 : 73;
    for (int pl = 0; pl < 2; pl++) {
        int(*grain)[82] = pl ? cr_grain : cb_grain;
        random_register = fg->grain_seed ^ (pl ? 0x49d8u : 0xb524u);
        for (int y = 0; y < chroma_h; y++)
            for (int x = 0; x < chroma_w; x++) {
                const int g = (fg->num_uv_points[pl] || fg->chroma_scaling_from_luma) ? gaussian_sequence[get_random_number(11)] : 0;
                grain[y][x] = round2(g, shift);
            }
    }
    shift = fg->ar_coeff_shift;
    for (int y = 3; y < chroma_h; y++)
        for (int x = 3; x < chroma_w - 3; x++) {
            int sum0 = 0, sum1 = 0, pos = 0;
            for (int dr = -lag; dr <= 0; dr++)
                for (int dc = -lag; dc <= lag; dc++) {
                    const int c0 = fg->ar_coeffs_uv[0][pos], c1 = fg->ar_coeffs_uv[1][pos];
                    if (dr == 0 && dc == 0) {
                        if (fg->num_y_points > 0) {
                            int luma = 0;
                            const int lx = ((x - 3) << sx) + 3, ly = ((y - 3) << sy) + 3;
                            for (int i = 0; i <= sy; i++)
                                for (int j = 0; j <= sx; j++) luma += luma_grain[ly + i][lx + j];
                            luma = round2(luma, sx + sy);
                            sum0 += luma * c0;
                            sum1 += luma * c1;
                        }
                        break;
                    }
                    sum0 += c0 * cb_grain[y + dr][x + dc];
                    sum1 += c1 * cr_grain[y + dr][x + dc];
                    pos++;
                }
            cb_grain[y][x] = clip3(GRAIN_MIN, GRAIN_MAX, cb_grain[y][x] + round2(sum0, shift));
            cr_grain[y][x] = clip3(GRAIN_MIN, GRAIN_MAX, cr_grain[y][x] + round2(sum1, shift));
        }
}

/* 7.18.3.4 */
static void scaling_lut_init(int n, const uint8_t pts[][2], uint8_t lut[256])
{
    memset(lut, 0, 256);
    if (!n) return;
    for (int i = 0; i < pts[0][0]; i++) lut[i] = pts[0][1];
    for (int i = 0; i < n - 1; i++) {
        const int dy = pts[i + 1][1] - pts[i][1], dx = pts[i + 1][0] - pts[i][0];
        const int delta = dy * ((65536 + (dx >> 1)) / dx);
        for (int x = 0; x < dx; x++) lut[pts[i][0] + x] = (uint8_t)(pts[i][1] + ((x * delta + 32768) >> 16));
    }
    for (int i = pts[n - 1][0]; i < 256; i++) lut[i] = pts[n - 1][1];
}

/* planes: tightly packed coded area cw x chh (whole 8x8 units); the grain is synthesised for the visible
 * picture (cw - crop_right) x (chh - crop_bottom), chroma Round2(size, 1); the rest is copied */
int ea1o_film_grain(const ea1_film_grain *fg, int cw_, int chh_, const uint8_t *const in[3], uint8_t *const out[3])
{
    const int w = cw_ - (fg->crop & 15), h = chh_ - (fg->crop >> 4);
    const int sx = GSX, sy = GSY;
    const int ls = cw_, cs = (cw_ + sx) >> sx;   /* row strides (Round2: the caller may hand over the VISIBLE planes of an odd-sized picture) */
    memcpy(out[0], in[0], (size_t)cw_ * chh_);
    memcpy(out[1], in[1], (size_t)cs * ((chh_ + sy) >> sy));
    memcpy(out[2], in[2], (size_t)cs * ((chh_ + sy) >> sy));
    generate_grain(fg);
    uint8_t lut[3][256];
    scaling_lut_init(fg->num_y_points, fg->point_y, lut[0]);
    for (int pl = 0; pl < 2; pl++) {
        if (fg->chroma_scaling_from_luma) memcpy(lut[1 + pl], lut[0], 256);
        else scaling_lut_init(fg->num_uv_points[pl], fg->point_uv[pl], lut[1 + pl]);
    }
    /* 7.18.3.5: noise stripes of 34 rows, one per 32 luma rows */
    const int nstripes = (h + 31) / 32;
    const int sw = w + 64;   /* stripe row storage (the last block may reach past the picture) */
    int *stripe[3];
    for (int pl = 0; pl < 3; pl++) stripe[pl] = calloc((size_t)nstripes * 34 * sw, sizeof(int));
#define STRIPE(pl, n, i, x) stripe[pl][((size_t)(n) * 34 + (i)) * sw + (x)]
    int luma_num = 0;
    for (int y = 0; y < (h + 1) / 2; y += 16) {
        random_register = fg->grain_seed;
        random_register ^= (unsigned)((luma_num * 37 + 178) & 255) << 8;
        random_register ^= (unsigned)((luma_num * 173 + 105) & 255);
        for (int x = 0; x < (w + 1) / 2; x += 16) {
            const int rnd = get_random_number(8);
            const int offx = rnd >> 4, offy = rnd & 15;
            for (int pl = 0; pl < 3; pl++) {
                const int subx = pl > 0 && sx, suby = pl > 0 && sy;
                const int pox = subx ? 6 + offx : 9 + offx * 2, poy = suby ? 6 + offy : 9 + offy * 2;
                for (int i = 0; i < (34 >> suby); i++)
                    for (int j = 0; j < (34 >> subx); j++) {
                        int g = pl == 0 ? luma_grain[poy + i][pox + j] : (pl == 1 ? cb_grain[poy + i][pox + j] : cr_grain[poy + i][pox + j]);
                        if (!subx) {
                            if (j < 2 && fg->overlap_flag && x > 0) {
                                const int old = STRIPE(pl, luma_num, i, x * 2 + j);
                                g = j == 0 ? old * 27 + g * 17 : old * 17 + g * 27;
                                g = clip3(GRAIN_MIN, GRAIN_MAX, round2(g, 5));
                            }
                            STRIPE(pl, luma_num, i, x * 2 + j) = g;
                        } else {
                            if (j == 0 && fg->overlap_flag && x > 0) {
                                const int old = STRIPE(pl, luma_num, i, x + j);
                                g = old * 23 + g * 22;
                                g = clip3(GRAIN_MIN, GRAIN_MAX, round2(g, 5));
                            }
                            STRIPE(pl, luma_num, i, x + j) = g;
                        }
                    }
            }
        }
        luma_num++;
    }
    /* noise image */
    int *noise[3];
    for (int pl = 0; pl < 3; pl++) {
        const int subx = pl > 0 && sx, sub = pl > 0 && sy, pw = (w + subx) >> subx, ph = (h + sub) >> sub;
        noise[pl] = malloc(sizeof(int) * (size_t)pw * ph);
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                const int n = y / (32 >> sub), i = y - n * (32 >> sub);   /* (stripes: the vertical subsampling) */
                int g = STRIPE(pl, n, i, x);
                if (!sub) {
                    if (i < 2 && n > 0 && fg->overlap_flag) {
                        const int old = STRIPE(pl, n - 1, i + 32, x);
                        g = i == 0 ? old * 27 + g * 17 : old * 17 + g * 27;
                        g = clip3(GRAIN_MIN, GRAIN_MAX, round2(g, 5));
                    }
                } else if (i < 1 && n > 0 && fg->overlap_flag) {
                    const int old = STRIPE(pl, n - 1, i + 16, x);
                    g = old * 23 + g * 22;
                    g = clip3(GRAIN_MIN, GRAIN_MAX, round2(g, 5));
                }
                noise[pl][(size_t)y * pw + x] = g;
            }
    }
    /* final blend: chroma first (it reads the luma WITHOUT noise), then luma */
    const int min_value = fg->clip_to_restricted_range ? 16 : 0;
    const int max_luma = fg->clip_to_restricted_range ? 235 : 255;
    const int max_chroma = fg->clip_to_restricted_range ? (fg->matrix_is_identity ? 235 : 240) : 255;
    const int cw = (w + sx) >> sx, ch = (h + sy) >> sy;
    for (int pl = 1; pl < 3; pl++)
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) {
                const int orig = in[pl][(size_t)y * cs + x];
                int v = orig;
                if (fg->num_uv_points[pl - 1] > 0 || fg->chroma_scaling_from_luma) {
                    const int lx = x << sx, ly = y << sy, lnx = lx + 1 < w - 1 ? lx + 1 : w - 1;
                    const int avg = sx ? round2(in[0][(size_t)ly * ls + lx] + in[0][(size_t)ly * ls + lnx], 1) : in[0][(size_t)ly * ls + lx];
                    int merged;
                    if (fg->chroma_scaling_from_luma) {
                        merged = avg;
                    } else {
                        const int combined = avg * (fg->uv_luma_mult[pl - 1] - 128) + orig * (fg->uv_mult[pl - 1] - 128);
                        merged = clip3(0, 255, (combined >> 6) + (fg->uv_offset[pl - 1] - 256));
                    }
                    const int n = round2(lut[pl][merged] * noise[pl][(size_t)y * cw + x], fg->scaling_shift);
                    v = clip3(min_value, max_chroma, orig + n);
                }
                out[pl][(size_t)y * cs + x] = (uint8_t)v;
            }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int orig = in[0][(size_t)y * ls + x];
            int v = orig;
            if (fg->num_y_points > 0) {
                const int n = round2(lut[0][orig] * noise[0][(size_t)y * w + x], fg->scaling_shift);
                v = clip3(min_value, max_luma, orig + n);
            }
            out[0][(size_t)y * ls + x] = (uint8_t)v;
        }
    for (int pl = 0; pl < 3; pl++) {
        free(stripe[pl]);
        free(noise[pl]);
    }
    return 0;
}
