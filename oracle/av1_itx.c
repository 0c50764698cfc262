/*
 * oracle/av1_itx.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the AV1 inverse transforms.
 *
 * Parity status: PINNED since round 4 against dav1d 1.5.3 (the decoder the reference pins; its dav1d_* entry points out of Pillow's bundled libavif) on libaom 3.13.2 streams, bit-exact: tests/test_real_av1.py, tests/golden/make_video_fixtures.py, DESIGN.md section 2.
 * The reference's arithmetic for this unit lives in
 * dav1d src/itx_1d.c + src/itx_tmpl.c (listed at /root/reference/cmake/dav1d.cmake:979,:1008;
 * the submodule itself is empty and the AV1 specification is not in this container), so this file
 * restates the published algorithm (AV1 spec section 7.13: butterfly network with one
 * Round2(.,12) per rotation; 2-D flow with row shift / int16 clamps / Round2(.,4)) from
 * knowledge.  What IS pinned here (tests/test_oracle_video.py): every 1-D transform equals the
 * mathematical DCT-II^-1 / AV1 ADST / identity to within its rounding error for all lengths,
 * the cos128 table equals its closed form, and DC-only blocks reproduce dav1d's dc shortcut
 * identity.  Fixture format (tests/golden/) is laid out so dav1d vectors can be dropped in.
 *
 * Structure: the DCT is written as ONE recursive routine for all lengths 4..64 (bit-reversed
 * slots, even half = half-length DCT, odd half = rotation/add-sub network, final mirror merge),
 * i.e. the structure of the spec's generic "inverse DCT process"; ADST4/8/16 follow the spec's
 * fixed networks.  The GPU kernels (easyav1_amd/csrc/av1_itx.hip) are written independently as
 * fully unrolled compile-time code; agreement bit-for-bit is what the parity tests assert.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ea1_av1_tables.inc"
#include "easyav1_amd_video.h"

static const int32_t cospi[65] = {EA1_COSPI_VALUES};

static inline int32_t clip16(int64_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : (int32_t)v); }

/* Conformance monitor (AV1 spec 7.13.2.1 / 7.13.3: "it is a requirement of bitstream conformance that all values stored in the variable T by this
 * process are representable by a signed integer using 8 + BitDepth bits"): how many values of the one-dimensional transforms of this thread left that
 * range.  A stream that makes this count is not a conformant one, and decoders are free to differ on it (dav1d clamps inside its butterflies, this
 * restatement follows the specification's clamps between the passes only); the sweep that feeds generated streams to dav1d (tools/dbg/gen_vs_dav1d.py)
 * sets such streams aside instead of counting them as mismatches. */
static __thread unsigned long g_range_violations;
static inline int32_t range_chk(int64_t v)
{
    if (v < -32768 || v > 32767) g_range_violations++;
    return (int32_t)v;
}
unsigned long ea1o_itx_range_violations(int reset)
{
    const unsigned long n = g_range_violations;
    if (reset) g_range_violations = 0;
    return n;
}

/* one butterfly rotation output: Round2(w0*a + w1*b, 12) */
static inline int32_t half_btf(int32_t w0, int32_t a, int32_t w1, int32_t b)
{
    return range_chk(((int64_t)w0 * a + (int64_t)w1 * b + 2048) >> 12);
}

static int bitrev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++)
        if (v & (1 << i)) r |= 1 << (bits - 1 - i);
    return r;
}

static int ilog2i(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

/* rotation with lo' = -c[A] lo + c[64-A] hi, hi' = c[64-A] lo + c[A] hi */
static void rot_p1(int32_t *T, int lo, int hi, int A)
{
    int32_t x = T[lo], y = T[hi];
    T[lo] = half_btf(-cospi[A], x, cospi[64 - A], y);
    T[hi] = half_btf(cospi[64 - A], x, cospi[A], y);
}
/* rotation with lo' = -c[64-A] lo - c[A] hi, hi' = -c[A] lo + c[64-A] hi */
static void rot_p2(int32_t *T, int lo, int hi, int A)
{
    int32_t x = T[lo], y = T[hi];
    T[lo] = half_btf(-cospi[64 - A], x, -cospi[A], y);
    T[hi] = half_btf(-cospi[A], x, cospi[64 - A], y);
}

/* odd half of an inverse DCT of length 2M: slots [M, 2M) */
static void idct_odd(int32_t *T, int M)
{
    const int N = 2 * M, L = ilog2i(M);
    /* input rotations: slot lo holds input k (odd), slot hi holds input N-k */
    for (int j = 0; j < M / 2; j++) {
        int lo = M + j, hi = 2 * M - 1 - j;
        int k = bitrev(lo, L + 1);
        int A = 64 * k / N;
        int32_t x = T[lo], y = T[hi];
        T[lo] = half_btf(cospi[64 - A], x, -cospi[A], y);
        T[hi] = half_btf(cospi[A], x, cospi[64 - A], y);
    }
    for (int t = 1; t <= L - 1; t++) {
        /* add/sub inside groups of g = 2^t, alternating orientation */
        int g = 1 << t;
        for (int s = M, q = 0; s < 2 * M; s += g, q++) {
            for (int i = 0; i < g / 2; i++) {
                int lo = s + i, hi = s + g - 1 - i;
                int32_t a = T[lo], b = T[hi];
                if ((q & 1) == 0) {
                    T[lo] = clip16((int64_t)a + b);
                    T[hi] = clip16((int64_t)a - b);
                } else {
                    T[lo] = clip16((int64_t)b - a);
                    T[hi] = clip16((int64_t)b + a);
                }
            }
        }
        if (t == L - 1) {
            for (int lo = M + M / 4; lo < M + M / 2; lo++) rot_p1(T, lo, 3 * M - 1 - lo, 32);
        } else {
            int B = 1 << (t + 1);
            int nblocks = (M / 2) / B;
            int bits = ilog2i(nblocks) + 1;
            for (int b = 0; b < nblocks; b++) {
                int base = M + b * B;
                int kb = 2 * bitrev(b, bits) + 1;
                int A = 16 * kb / nblocks;
                for (int i = 0; i < B / 4; i++) {
                    int lo1 = base + B / 4 + i, lo2 = base + B / 2 + i;
                    rot_p1(T, lo1, 3 * M - 1 - lo1, A);
                    rot_p2(T, lo2, 3 * M - 1 - lo2, A);
                }
            }
        }
    }
}

/* T in bit-reversed input order, natural output order */
static void idct_core(int32_t *T, int n)
{
    if (n == 2) {
        int32_t a = T[0], b = T[1];
        T[0] = half_btf(cospi[32], a, cospi[32], b);
        T[1] = half_btf(cospi[32], a, -cospi[32], b);
        return;
    }
    idct_core(T, n / 2);
    idct_odd(T, n / 2);
    for (int i = 0; i < n / 2; i++) {
        int32_t a = T[i], b = T[n - 1 - i];
        T[i] = clip16((int64_t)a + b);
        T[n - 1 - i] = clip16((int64_t)a - b);
    }
}

static void inv_dct(int32_t *io, int n)
{
    int32_t T[64];
    const int bits = ilog2i(n);
    for (int i = 0; i < n; i++) T[i] = io[bitrev(i, bits)];
    idct_core(T, n);
    memcpy(io, T, sizeof(int32_t) * n);
}

static void inv_adst4(int32_t *io)
{
    const int64_t x0 = io[0], x1 = io[1], x2 = io[2], x3 = io[3];
    int64_t s0 = EA1_SINPI_1_9 * x0;
    int64_t s1 = EA1_SINPI_2_9 * x0;
    int64_t s2 = EA1_SINPI_3_9 * x1;
    int64_t s3 = EA1_SINPI_4_9 * x2;
    int64_t s4 = EA1_SINPI_1_9 * x2;
    int64_t s5 = EA1_SINPI_2_9 * x3;
    int64_t s6 = EA1_SINPI_4_9 * x3;
    int64_t a7 = x0 - x2;
    int64_t b7 = a7 + x3;
    s0 = s0 + s3;
    s1 = s1 - s4;
    s3 = s2;
    s2 = EA1_SINPI_3_9 * b7;
    s0 = s0 + s5;
    s1 = s1 - s6;
    int64_t o0 = s0 + s3, o1 = s1 + s3, o2 = s2, o3 = s0 + s1;
    o3 = o3 - s3;
    io[0] = (int32_t)((o0 + 2048) >> 12);
    io[1] = (int32_t)((o1 + 2048) >> 12);
    io[2] = (int32_t)((o2 + 2048) >> 12);
    io[3] = (int32_t)((o3 + 2048) >> 12);
}

#define ADDSUB(lo, hi)                                  \
    do {                                                \
        int32_t a_ = b[lo], b_ = b[hi];                 \
        b[lo] = clip16((int64_t)a_ + b_);               \
        b[hi] = clip16((int64_t)a_ - b_);               \
    } while (0)
/* (lo, hi) <- (c[A] lo + c[64-A] hi, c[64-A] lo - c[A] hi) */
#define ROT_A(lo, hi, A)                                               \
    do {                                                               \
        int32_t a_ = b[lo], b_ = b[hi];                                \
        b[lo] = half_btf(cospi[A], a_, cospi[64 - (A)], b_);           \
        b[hi] = half_btf(cospi[64 - (A)], a_, -cospi[A], b_);          \
    } while (0)
/* (lo, hi) <- (-c[64-A] lo + c[A] hi, c[A] lo + c[64-A] hi) */
#define ROT_B(lo, hi, A)                                               \
    do {                                                               \
        int32_t a_ = b[lo], b_ = b[hi];                                \
        b[lo] = half_btf(-cospi[64 - (A)], a_, cospi[A], b_);          \
        b[hi] = half_btf(cospi[A], a_, cospi[64 - (A)], b_);           \
    } while (0)

static void inv_adst8(int32_t *io)
{
    int32_t b[8] = {io[7], io[0], io[5], io[2], io[3], io[4], io[1], io[6]};
    ROT_A(0, 1, 4);
    ROT_A(2, 3, 20);
    ROT_A(4, 5, 36);
    ROT_A(6, 7, 52);
    ADDSUB(0, 4); ADDSUB(1, 5); ADDSUB(2, 6); ADDSUB(3, 7);
    ROT_A(4, 5, 16);
    ROT_B(6, 7, 16);
    ADDSUB(0, 2); ADDSUB(1, 3); ADDSUB(4, 6); ADDSUB(5, 7);
    ROT_A(2, 3, 32);
    ROT_A(6, 7, 32);
    io[0] = b[0]; io[1] = -b[4]; io[2] = b[6]; io[3] = -b[2];
    io[4] = b[3]; io[5] = -b[7]; io[6] = b[5]; io[7] = -b[1];
}

static void inv_adst16(int32_t *io)
{
    int32_t b[16] = {io[15], io[0], io[13], io[2], io[11], io[4], io[9], io[6],
                     io[7], io[8], io[5], io[10], io[3], io[12], io[1], io[14]};
    for (int i = 0; i < 8; i++) ROT_A(2 * i, 2 * i + 1, 2 + 8 * i);
    for (int i = 0; i < 8; i++) ADDSUB(i, i + 8);
    ROT_A(8, 9, 8);
    ROT_A(10, 11, 40);
    ROT_B(12, 13, 8);
    ROT_B(14, 15, 40);
    for (int i = 0; i < 4; i++) { ADDSUB(i, i + 4); ADDSUB(8 + i, 12 + i); }
    ROT_A(4, 5, 16);
    ROT_B(6, 7, 16);
    ROT_A(12, 13, 16);
    ROT_B(14, 15, 16);
    for (int g = 0; g < 16; g += 4) { ADDSUB(g, g + 2); ADDSUB(g + 1, g + 3); }
    for (int g = 0; g < 16; g += 4) ROT_A(g + 2, g + 3, 32);
    io[0] = b[0];   io[1] = -b[8];   io[2] = b[12];  io[3] = -b[4];
    io[4] = b[6];   io[5] = -b[14];  io[6] = b[10];  io[7] = -b[2];
    io[8] = b[3];   io[9] = -b[11];  io[10] = b[15]; io[11] = -b[7];
    io[12] = b[5];  io[13] = -b[13]; io[14] = b[9];  io[15] = -b[1];
}

static void inv_identity(int32_t *io, int n)
{
    for (int i = 0; i < n; i++) {
        int64_t v = io[i];
        switch (n) {
        case 4: io[i] = (int32_t)((v * 5793 + 2048) >> 12); break;
        case 8: io[i] = (int32_t)(v * 2); break;
        case 16: io[i] = (int32_t)((v * 11586 + 2048) >> 12); break;
        default: io[i] = (int32_t)(v * 4); break;
        }
    }
}

/* kind: EA1_TX_*; returns 0 if the (kind, n) pair exists in AV1 */
int ea1o_inv_txfm_1d(int kind, int n, int32_t *io)
{
    switch (kind) {
    case EA1_TX_DCT:
        if (n < 4 || n > 64) return -1;
        inv_dct(io, n);
        return 0;
    case EA1_TX_ADST:
    case EA1_TX_FLIPADST:
        if (n == 4) inv_adst4(io);
        else if (n == 8) inv_adst8(io);
        else if (n == 16) inv_adst16(io);
        else return -1;
        if (kind == EA1_TX_FLIPADST)
            for (int i = 0; i < n / 2; i++) {
                int32_t t = io[i];
                io[i] = io[n - 1 - i];
                io[n - 1 - i] = t;
            }
        return 0;
    case EA1_TX_IDENTITY:
        if (n > 32) return -1;
        inv_identity(io, n);
        return 0;
    }
    return -1;
}

/* inverse Walsh-Hadamard, 4 points (AV1 spec 7.13.2.10) */
static void inv_wht4(int32_t *io, int shift)
{
    int32_t a = io[0] >> shift, c = io[1] >> shift, d = io[2] >> shift, b = io[3] >> shift;
    a += c;
    d -= b;
    const int32_t e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= b;
    d += c;
    io[0] = a;
    io[1] = b;
    io[2] = c;
    io[3] = d;
}

static int row_shift(int w, int h)
{
    /* Transform_Row_Shift: by area / shape */
    const int lw = ilog2i(w), lh = ilog2i(h);
    static const int8_t tbl[5][5] = {
        /* h:    4  8  16 32 64      w */
        /* 4 */ {0, 0, 1, -1, -1},
        /* 8 */ {0, 1, 1, 2, -1},
        /*16 */ {1, 1, 2, 1, 2},
        /*32 */ {-1, 2, 1, 2, 1},
        /*64 */ {-1, -1, 2, 1, 2},
    };
    return tbl[lw - 2][lh - 2];
}

/* dav1d's itxfm_add(dst, stride, coeff, eob): coef is min(w,32) x min(h,32), row-major */
int ea1o_itx_add(uint8_t *dst, int stride, int w, int h, int txtp, const int16_t *coef)
{
    if (txtp == EA1_TXTP_WHT) {
        /* lossless 4x4 Walsh-Hadamard (AV1 spec 7.13.2.10): rows with shift 2, columns with shift
         * 0, no rounding shift on the residual */
        if (w != 4 || h != 4) return -1;
        int32_t t[4][4];
        for (int y = 0; y < 4; y++) {
            for (int x = 0; x < 4; x++) t[y][x] = coef[y * 4 + x];
            inv_wht4(t[y], 2);
        }
        for (int x = 0; x < 4; x++) {
            int32_t col[4] = {t[0][x], t[1][x], t[2][x], t[3][x]};
            inv_wht4(col, 0);
            for (int y = 0; y < 4; y++) {
                int v = dst[y * stride + x] + col[y];
                dst[y * stride + x] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        }
        return 0;
    }
    if (txtp >> 4) return -1;
    const int rk = txtp & 3, ck = (txtp >> 2) & 3;
    const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
    const int shift = row_shift(w, h);
    if (shift < 0) return -1;
    const int rect2 = (w == 2 * h) || (h == 2 * w);
    const int rnd = (1 << shift) >> 1;
    int32_t *tmp = calloc((size_t)w * h, sizeof(int32_t));
    int32_t line[64];
    int rc = 0;
    for (int y = 0; y < sh && !rc; y++) {
        memset(line, 0, sizeof line);
        for (int x = 0; x < sw; x++) {
            int32_t c = coef[y * sw + x];
            line[x] = rect2 ? (c * 181 + 128) >> 8 : c;
        }
        rc = ea1o_inv_txfm_1d(rk, w, line);
        for (int x = 0; x < w; x++) range_chk(line[x]);
        for (int x = 0; x < w; x++) tmp[y * w + x] = clip16(((int64_t)line[x] + rnd) >> shift);
    }
    for (int x = 0; x < w && !rc; x++) {
        for (int y = 0; y < h; y++) line[y] = tmp[y * w + x];
        rc = ea1o_inv_txfm_1d(ck, h, line);
        for (int y = 0; y < h; y++) range_chk(line[y]);
        for (int y = 0; y < h; y++) {
            int v = dst[y * stride + x] + ((line[y] + 8) >> 4);
            dst[y * stride + x] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
    }
    free(tmp);
    return rc;
}
