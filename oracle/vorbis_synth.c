/*
 * oracle/vorbis_synth.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, float32, no FMA contraction: build with -ffp-contract=off) of the
 * Vorbis synthesis half of the easyav1 hot path.  It is the checker for the HIP kernels in
 * easyav1_amd/csrc/vorbis_synth.hip; nothing in the product path may call it.
 *
 * Parity status: PINNED.  tests/test_oracle_audio.py checks every routine here bit-for-bit
 * against the reference's own code compiled from /root/reference (oracle/_ref, see
 * oracle/ref_minivorbis.c) and against the committed golden vectors in tests/golden/ that were
 * generated from that reference build (tests/golden/make_audio_golden.py).
 *
 * Reference followed (all in /root/reference/ext/minivorbis/minivorbis.h):
 *   tables        mdct_init                      :6709-6748
 *   IMDCT         mdct_backward                  :7054-7148
 *                 mdct_butterflies               :6974-6994
 *                 mdct_butterfly_first/_generic  :6874-6972
 *                 mdct_butterfly_32/_16/_8       :6751-6871
 *                 mdct__bitreverse               :7004-7052
 *   window        vwin64..vwin8192 literals      :9917-11979 (regenerated, see ea1o_window)
 *   overlap-add   vorbis_synthesis_blockin       :9195-9408
 *                 vorbis_synthesis_pcmout/read   :9411-9430
 *                 vorbis_synthesis_restart       :9159-9180
 *   PCM ring      decode_audio / prepare_audio_buffer_for_new_samples / easyav1_get_audio_frame
 *                 /root/reference/src/easyav1.c:2434-2532, :3604-3630
 *
 * The reference walks pointers through the buffer; this restatement is written in "flat index"
 * form (every stage is a set of independent little operations indexed by one integer), which is
 * the form the GPU kernel parallelises.  Each arithmetic expression keeps the reference's
 * operand pairing: a*b + c*d is two float multiplies and one float add, never an FMA.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define cPI3_8 .38268343236508977175F
#define cPI2_8 .70710678118654752441F
#define cPI1_8 .92387953251128675613F

static int ilog2_exact(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

/* mdct_init :6709.  trig has n + n/4 floats, bitrev n/4 ints. */
void ea1o_mdct_tables(int n, float *T, int *bitrev)
{
    const int n2 = n >> 1;
    const int log2n = ilog2_exact(n);
    for (int i = 0; i < n / 4; i++) {
        T[i * 2] = (float)cos((M_PI / n) * (4 * i));
        T[i * 2 + 1] = (float)-sin((M_PI / n) * (4 * i));
        T[n2 + i * 2] = (float)cos((M_PI / (2 * n)) * (2 * i + 1));
        T[n2 + i * 2 + 1] = (float)sin((M_PI / (2 * n)) * (2 * i + 1));
    }
    for (int i = 0; i < n / 8; i++) {
        T[n + i * 2] = (float)(cos((M_PI / n) * (4 * i + 2)) * .5);
        T[n + i * 2 + 1] = (float)(-sin((M_PI / n) * (4 * i + 2)) * .5);
    }
    const int mask = (1 << (log2n - 1)) - 1;
    const int msb = 1 << (log2n - 2);
    for (int i = 0; i < n / 8; i++) {
        int acc = 0;
        for (int j = 0; msb >> j; j++)
            if ((msb >> j) & i) acc |= 1 << j;
        bitrev[i * 2] = ((~acc) & mask) - 1;
        bitrev[i * 2 + 1] = acc;
    }
}

/*
 * Half window for blocksize n (n/2 floats), vwin tables :9917-11979.  The reference stores
 * sin(pi/2 * sin^2((i+.5)/n * pi)) as decimal literals with 10 fractional digits; the float the
 * C compiler makes of such a literal is reproduced here by printing the double with "%.10f" and
 * parsing it back (verified bit-for-bit against all eight tables in tests/test_oracle_audio.py).
 */
void ea1o_window(int n, float *w)
{
    char buf[64];
    for (int i = 0; i < n / 2; i++) {
        double s = sin((i + 0.5) / n * M_PI);
        double v = sin(0.5 * M_PI * s * s);
        snprintf(buf, sizeof buf, "%.10f", v);
        w[i] = (float)strtod(buf, NULL);
    }
}

/* one (x1,x2) twiddle pair of mdct_butterfly_first/_generic (:6889-6894) */
static void pair_op(float *x, int a, int b, float t0, float t1)
{
    float r0 = x[a] - x[b];
    float r1 = x[a + 1] - x[b + 1];
    x[a] += x[b];
    x[a + 1] += x[b + 1];
    x[b] = r1 * t1 + r0 * t0;
    x[b + 1] = r1 * t0 - r0 * t1;
}

/* first stage of mdct_butterfly_32 (:6811-6866): pair k works on x[16+2k],x[17+2k] vs x[2k],x[2k+1] */
static void b32_pair(float *x, int k)
{
    const int hi = 16 + 2 * k, lo = 2 * k;
    float r0, r1;
    switch (k) {
    case 7:
        r0 = x[30] - x[14]; r1 = x[31] - x[15];
        x[30] += x[14]; x[31] += x[15];
        x[14] = r0; x[15] = r1;
        break;
    case 6:
        r0 = x[28] - x[12]; r1 = x[29] - x[13];
        x[28] += x[12]; x[29] += x[13];
        x[12] = r0 * cPI1_8 - r1 * cPI3_8;
        x[13] = r0 * cPI3_8 + r1 * cPI1_8;
        break;
    case 5:
        r0 = x[26] - x[10]; r1 = x[27] - x[11];
        x[26] += x[10]; x[27] += x[11];
        x[10] = (r0 - r1) * cPI2_8;
        x[11] = (r0 + r1) * cPI2_8;
        break;
    case 4:
        r0 = x[24] - x[8]; r1 = x[25] - x[9];
        x[24] += x[8]; x[25] += x[9];
        x[8] = r0 * cPI3_8 - r1 * cPI1_8;
        x[9] = r1 * cPI3_8 + r0 * cPI1_8;
        break;
    case 3:
        r0 = x[22] - x[6]; r1 = x[7] - x[23];
        x[22] += x[6]; x[23] += x[7];
        x[6] = r1; x[7] = r0;
        break;
    case 2:
        r0 = x[4] - x[20]; r1 = x[5] - x[21];
        x[20] += x[4]; x[21] += x[5];
        x[4] = r1 * cPI1_8 + r0 * cPI3_8;
        x[5] = r1 * cPI3_8 - r0 * cPI1_8;
        break;
    case 1:
        r0 = x[2] - x[18]; r1 = x[3] - x[19];
        x[18] += x[2]; x[19] += x[3];
        x[2] = (r1 + r0) * cPI2_8;
        x[3] = (r1 - r0) * cPI2_8;
        break;
    default:
        r0 = x[0] - x[16]; r1 = x[1] - x[17];
        x[16] += x[0]; x[17] += x[1];
        x[0] = r1 * cPI3_8 + r0 * cPI1_8;
        x[1] = r1 * cPI1_8 - r0 * cPI3_8;
        break;
    }
    (void)hi; (void)lo;
}

/* first stage of mdct_butterfly_16 (:6776-6803): pair k of a 16-float group */
static void b16_pair(float *x, int k)
{
    float r0, r1;
    switch (k) {
    case 0:
        r0 = x[1] - x[9]; r1 = x[0] - x[8];
        x[8] += x[0]; x[9] += x[1];
        x[0] = (r0 + r1) * cPI2_8;
        x[1] = (r0 - r1) * cPI2_8;
        break;
    case 1:
        r0 = x[3] - x[11]; r1 = x[10] - x[2];
        x[10] += x[2]; x[11] += x[3];
        x[2] = r0; x[3] = r1;
        break;
    case 2:
        r0 = x[12] - x[4]; r1 = x[13] - x[5];
        x[12] += x[4]; x[13] += x[5];
        x[4] = (r0 - r1) * cPI2_8;
        x[5] = (r0 + r1) * cPI2_8;
        break;
    default:
        r0 = x[14] - x[6]; r1 = x[15] - x[7];
        x[14] += x[6]; x[15] += x[7];
        x[6] = r0; x[7] = r1;
        break;
    }
}

/* mdct_butterfly_8 (:6751-6772) */
static void b8(float *x)
{
    float r0 = x[6] + x[2];
    float r1 = x[6] - x[2];
    float r2 = x[4] + x[0];
    float r3 = x[4] - x[0];

    x[6] = r0 + r2;
    x[4] = r0 - r2;

    r0 = x[5] - x[1];
    r2 = x[7] - x[3];
    x[0] = r1 + r0;
    x[2] = r1 - r0;

    r0 = x[5] + x[1];
    r1 = x[7] + x[3];
    x[3] = r2 + r3;
    x[1] = r2 - r3;
    x[7] = r1 + r0;
    x[5] = r1 - r0;
}

/*
 * mdct_backward :7054.  in: n/2 floats, out: n floats; in may equal out (as at :17825).
 * T/bitrev from ea1o_mdct_tables.
 */
void ea1o_mdct_backward(int n, const float *T, const int *bitrev, const float *in, float *out)
{
    const int n2 = n >> 1, n4 = n >> 2;
    const int log2n = ilog2_exact(n);

    /* pre-rotation (:7061-7087); iteration `it` of each reference loop */
    for (int it = 0; it < n / 16; it++) {
        const float *iX = in + n2 - 7 - 8 * it;
        float *oX = out + n2 + n4 - 4 * (it + 1);
        const float *t = T + n4 + 4 * it;
        oX[0] = -iX[2] * t[3] - iX[0] * t[2];
        oX[1] = iX[0] * t[3] - iX[2] * t[2];
        oX[2] = -iX[6] * t[1] - iX[4] * t[0];
        oX[3] = iX[4] * t[1] - iX[6] * t[0];
    }
    for (int it = 0; it < n / 16; it++) {
        const float *iX = in + n2 - 8 - 8 * it;
        float *oX = out + n2 + n4 + 4 * it;
        const float *t = T + n4 - 4 * (it + 1);
        oX[0] = iX[4] * t[3] + iX[6] * t[2];
        oX[1] = iX[4] * t[2] - iX[6] * t[3];
        oX[2] = iX[0] * t[1] + iX[2] * t[0];
        oX[3] = iX[0] * t[0] - iX[2] * t[1];
    }

    /* mdct_butterflies(out+n2, n2) :6974: log2n-6 radix-2 stages, then 32-point kernels */
    {
        float *x = out + n2;
        const int points = n2;
        const int nstage = log2n - 6;
        for (int s = 0; s < nstage; s++) {
            const int sub = points >> s;          /* size of a sub-block */
            const int trigint = 4 << s;
            for (int j = 0; j < (1 << s); j++) {
                float *xs = x + sub * j;
                for (int p = 0; p < sub / 4; p++)  /* p-th pair of _first/_generic */
                    pair_op(xs, sub - 2 - 2 * p, sub / 2 - 2 - 2 * p, T[trigint * p], T[trigint * p + 1]);
            }
        }
        for (int j = 0; j < points; j += 32) {
            float *xs = x + j;
            for (int k = 0; k < 8; k++) b32_pair(xs, k);
            for (int h = 0; h < 2; h++)
                for (int k = 0; k < 4; k++) b16_pair(xs + 16 * h, k);
            for (int q = 0; q < 4; q++) b8(xs + 8 * q);
        }
    }

    /* mdct__bitreverse :7004, sub-operation q (two per reference iteration) */
    {
        const float *x = out + n2;
        const float *t = T + n;
        for (int q = 0; q < n / 8; q++) {
            const float *x0 = x + bitrev[2 * q];
            const float *x1 = x + bitrev[2 * q + 1];
            float r0 = x0[1] - x1[1];
            float r1 = x0[0] + x1[0];
            float r2 = r1 * t[2 * q] + r0 * t[2 * q + 1];
            float r3 = r1 * t[2 * q + 1] - r0 * t[2 * q];
            r0 = (x0[1] + x1[1]) * .5f;
            r1 = (x0[0] - x1[0]) * .5f;
            const int lo = 2 * q, hi = n2 - 2 * q - 2;
            out[lo] = r0 + r2;
            out[lo + 1] = r1 + r3;
            out[hi] = r0 - r2;
            out[hi + 1] = r3 - r1;
        }
    }

    /* post-rotation + unfold (:7094-7146).  Every m reads out[2m],out[2m+1] (< n2) and writes
       four final samples at indices >= ... the reads of later m are still untouched only for the
       upper half, so stage through the two rotated values first (reference does the same via its
       three loops). */
    {
        const float *t = T + n2;
        for (int m = 0; m < n4; m++) {
            float a = out[2 * m], b = out[2 * m + 1];
            float v1 = a * t[2 * m + 1] - b * t[2 * m];
            float v2 = -(a * t[2 * m] + b * t[2 * m + 1]);
            out[n2 + n4 - 1 - m] = v1;   /* oX1 */
            out[n2 + n4 + m] = v2;       /* oX2 */
        }
        for (int m = 0; m < n4; m++) {
            float v1 = out[n2 + n4 - 1 - m];
            out[n4 - 1 - m] = v1;
            out[n4 + m] = -v1;
        }
        for (int m = 0; m < n4; m++)
            out[n2 + n4 - 1 - m] = out[n2 + n4 + m];
    }
}

/* ------------------------------------------------------------------------------------------
 * Overlap-add state machine: vorbis_synthesis_blockin/pcmout/read/restart, halfrate_flag = 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int channels;
    int bs[2];         /* blocksizes */
    float *win[2];     /* half windows */
    float **pcm;       /* [ch][bs[1]] two-stage double buffer (:8725-8731) */
    int lW, W;
    int centerW;
    int pcm_current, pcm_returned;
} ea1o_synth;

void *ea1o_synth_new(int channels, int bs0, int bs1)
{
    ea1o_synth *s = calloc(1, sizeof(*s));
    s->channels = channels;
    s->bs[0] = bs0;
    s->bs[1] = bs1;
    for (int k = 0; k < 2; k++) {
        s->win[k] = malloc(sizeof(float) * s->bs[k] / 2);
        ea1o_window(s->bs[k], s->win[k]);
    }
    s->pcm = malloc(sizeof(float *) * channels);
    for (int c = 0; c < channels; c++) s->pcm[c] = calloc(bs1, sizeof(float));
    s->lW = s->W = 0;
    /* vorbis_synthesis_restart :9170-9173 */
    s->centerW = bs1 >> 1;
    s->pcm_current = s->centerW;
    s->pcm_returned = -1;
    return s;
}

void ea1o_synth_free(void *h)
{
    ea1o_synth *s = h;
    for (int c = 0; c < s->channels; c++) free(s->pcm[c]);
    free(s->pcm);
    free(s->win[0]);
    free(s->win[1]);
    free(s);
}

int ea1o_synth_restart(void *h)
{
    ea1o_synth *s = h;
    s->centerW = s->bs[1] >> 1;
    s->pcm_current = s->centerW;
    s->pcm_returned = -1;
    return 0;
}

/* pcm_all: channel-major, bs[W] floats per channel (IMDCT outputs).  :9195 */
int ea1o_synth_blockin(void *h, int W, const float *pcm_all)
{
    ea1o_synth *s = h;
    if (s->pcm_current > s->pcm_returned && s->pcm_returned != -1) return -131; /* OV_EINVAL */
    s->lW = s->W;
    s->W = W;
    const int n = s->bs[W] >> 1, n0 = s->bs[0] >> 1, n1 = s->bs[1] >> 1;
    const int thisCenter = s->centerW ? n1 : 0;
    const int prevCenter = s->centerW ? 0 : n1;
    for (int j = 0; j < s->channels; j++) {
        const float *p = pcm_all + (size_t)j * s->bs[W];
        if (s->lW) {
            if (s->W) {
                const float *w = s->win[1];
                float *pcm = s->pcm[j] + prevCenter;
                for (int i = 0; i < n1; i++) pcm[i] = pcm[i] * w[n1 - i - 1] + p[i] * w[i];
            } else {
                const float *w = s->win[0];
                float *pcm = s->pcm[j] + prevCenter + n1 / 2 - n0 / 2;
                for (int i = 0; i < n0; i++) pcm[i] = pcm[i] * w[n0 - i - 1] + p[i] * w[i];
            }
        } else {
            if (s->W) {
                const float *w = s->win[0];
                float *pcm = s->pcm[j] + prevCenter;
                const float *pp = p + n1 / 2 - n0 / 2;
                int i;
                for (i = 0; i < n0; i++) pcm[i] = pcm[i] * w[n0 - i - 1] + pp[i] * w[i];
                for (; i < n1 / 2 + n0 / 2; i++) pcm[i] = pp[i];
            } else {
                const float *w = s->win[0];
                float *pcm = s->pcm[j] + prevCenter;
                for (int i = 0; i < n0; i++) pcm[i] = pcm[i] * w[n0 - i - 1] + p[i] * w[i];
            }
        }
        memcpy(s->pcm[j] + thisCenter, p + n, sizeof(float) * n);
    }
    s->centerW = s->centerW ? 0 : n1;
    if (s->pcm_returned == -1) {
        s->pcm_returned = thisCenter;
        s->pcm_current = thisCenter;
    } else {
        s->pcm_returned = prevCenter;
        s->pcm_current = prevCenter + s->bs[s->lW] / 4 + s->bs[s->W] / 4;
    }
    return 0;
}

/* channel-major copy-out with per-channel capacity `cap`; returns samples (or -needed) */
int ea1o_synth_pcmout(void *h, float *out, int cap)
{
    ea1o_synth *s = h;
    if (s->pcm_returned > -1 && s->pcm_returned < s->pcm_current) {
        int n = s->pcm_current - s->pcm_returned;
        if (!out) return n;
        if (n > cap) return -n;
        for (int c = 0; c < s->channels; c++)
            memcpy(out + (size_t)c * cap, s->pcm[c] + s->pcm_returned, sizeof(float) * n);
        return n;
    }
    return 0;
}

int ea1o_synth_read(void *h, int n)
{
    ea1o_synth *s = h;
    if (n && s->pcm_returned + n > s->pcm_current) return -131;
    s->pcm_returned += n;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * easyav1's 4096-sample PCM ring (src/easyav1.c:38 AUDIO_BUFFER_SIZE, :2471-2532, :3604-3630).
 * ------------------------------------------------------------------------------------------ */
#define EA1O_AUDIO_BUFFER_SIZE 4096

typedef struct {
    int channels, interlace;
    float *buffer;          /* 4096 * channels */
    unsigned samples;
    int has_samples;
} ea1o_ring;

void *ea1o_ring_new(int channels, int interlace)
{
    ea1o_ring *r = calloc(1, sizeof(*r));
    r->channels = channels;
    r->interlace = interlace;
    r->buffer = calloc((size_t)EA1O_AUDIO_BUFFER_SIZE * channels, sizeof(float));
    return r;
}

void ea1o_ring_free(void *h)
{
    ea1o_ring *r = h;
    free(r->buffer);
    free(r);
}

/* prepare_audio_buffer_for_new_samples :2501 */
static unsigned ring_prepare(ea1o_ring *r, int decoded)
{
    if (decoded > EA1O_AUDIO_BUFFER_SIZE) {
        r->samples = 0;
        return decoded - EA1O_AUDIO_BUFFER_SIZE;
    }
    if (decoded + r->samples <= EA1O_AUDIO_BUFFER_SIZE) return 0;
    unsigned mv = decoded + r->samples - EA1O_AUDIO_BUFFER_SIZE;
    if (r->interlace) {
        memmove(r->buffer, r->buffer + (size_t)mv * r->channels,
                (size_t)(EA1O_AUDIO_BUFFER_SIZE - mv) * r->channels * sizeof(float));
    } else {
        for (int c = 0; c < r->channels; c++)
            memmove(r->buffer + (size_t)c * EA1O_AUDIO_BUFFER_SIZE,
                    r->buffer + (size_t)c * EA1O_AUDIO_BUFFER_SIZE + mv,
                    (EA1O_AUDIO_BUFFER_SIZE - mv) * sizeof(float));
    }
    r->samples = EA1O_AUDIO_BUFFER_SIZE - decoded;
    return 0;
}

/* the body of decode_audio's copy loop for one pcmout result (:2468-2492).
   pcm: channel-major, `stride` floats per channel, `decoded` valid samples. */
void ea1o_ring_push(void *h, const float *pcm, int stride, int decoded)
{
    ea1o_ring *r = h;
    if (!r->has_samples && decoded >= 0) {
        /* :2441-2444 happens once per packet; caller signals a new packet with decoded < 0 */
    }
    unsigned off = ring_prepare(r, decoded);
    for (int s = 0; s < decoded; s++) {
        /* note: reference loops s over [0, decoded) reading pcm[s + off]; when off > 0 the tail
           reads run past the decoded range in the reference too (it only happens for
           decoded > 4096, impossible for blocksizes <= 8192: max 4096 samples per block). */
        for (int c = 0; c < r->channels; c++) {
            float v = pcm[(size_t)c * stride + s + off];
            if (r->interlace) r->buffer[(size_t)(r->samples + s) * r->channels + c] = v;
            else r->buffer[(size_t)c * EA1O_AUDIO_BUFFER_SIZE + r->samples + s] = v;
        }
    }
    r->samples += decoded;
    if (r->samples > EA1O_AUDIO_BUFFER_SIZE) r->samples = EA1O_AUDIO_BUFFER_SIZE;
    if (r->samples > 0) r->has_samples = 1;
}

/* :2441-2444: at the start of each packet, if the consumer took the previous frame */
void ea1o_ring_begin_packet(void *h)
{
    ea1o_ring *r = h;
    if (!r->has_samples) r->samples = 0;
}

/* easyav1_get_audio_frame :3604: returns samples (0 = NULL frame), exposes buffer */
unsigned ea1o_ring_get(void *h, const float **buf, size_t *bytes)
{
    ea1o_ring *r = h;
    if (!r->has_samples) return 0;
    r->has_samples = 0;
    if (r->samples == 0) return 0;
    *bytes = r->samples * sizeof(float) * (r->interlace ? r->channels : 1);
    *buf = r->buffer;
    return r->samples;
}
