/*
 * oracle/ref_minivorbis.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin harness that compiles the reference's vendored Vorbis decoder *where it lies*
 * (/root/reference/ext/minivorbis/minivorbis.h, included, never copied) into
 * oracle/_ref/libref_minivorbis.so, so that the CPU restatement in oracle/vorbis_synth.c can be
 * pinned against the real reference routines:
 *   mdct_init / mdct_backward          minivorbis.h:6709 / :7054
 *   _vorbis_window_get                 minivorbis.h:11992
 *   vorbis_synthesis_blockin/pcmout/read/restart   minivorbis.h:9195 / :9411 / :9426 / :9159
 * The dsp state is set up the way _vds_shared_init (minivorbis.h:8644) expects, with an empty
 * codebook list, so blockin can be driven with arbitrary (W, pcm) blocks exactly like
 * easyav1's decode_audio does after vorbis_synthesis (src/easyav1.c:2458-2490).
 *
 * Built only when /root/reference exists (see oracle/Makefile); output stays in oracle/_ref/.
 */
#define OGG_IMPL
#define VORBIS_IMPL
#include "minivorbis/minivorbis.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    mdct_lookup l;
} ref_mdct;

void *ref_mdct_new(int n)
{
    ref_mdct *m = calloc(1, sizeof(*m));
    mdct_init(&m->l, n);
    return m;
}

void ref_mdct_free(void *h)
{
    ref_mdct *m = h;
    mdct_clear(&m->l);
    free(m);
}

/* in-place capable, exactly like mapping0_inverse calls it (minivorbis.h:17825) */
void ref_mdct_backward(void *h, float *in, float *out)
{
    mdct_backward(&((ref_mdct *)h)->l, in, out);
}

const float *ref_mdct_trig(void *h) { return ((ref_mdct *)h)->l.trig; }
const int *ref_mdct_bitrev(void *h) { return ((ref_mdct *)h)->l.bitrev; }

/* idx = ilog(blocksize) - 7, i.e. 0 -> 64 ... 7 -> 8192 (minivorbis.h:8681) */
const float *ref_window(int idx) { return _vorbis_window_get(idx); }

typedef struct {
    vorbis_info vi;
    vorbis_dsp_state v;
    vorbis_block vb;
    float **pcm;
    long seq;
} ref_synth;

void *ref_synth_new(int channels, int bs0, int bs1)
{
    ref_synth *s = calloc(1, sizeof(*s));
    vorbis_info_init(&s->vi);
    codec_setup_info *ci = s->vi.codec_setup;
    s->vi.channels = channels;
    s->vi.rate = 48000;
    ci->blocksizes[0] = bs0;
    ci->blocksizes[1] = bs1;
    ci->modes = 1;
    if (vorbis_synthesis_init(&s->v, &s->vi)) {
        free(s);
        return NULL;
    }
    memset(&s->vb, 0, sizeof(s->vb));
    s->vb.vd = &s->v;
    s->pcm = calloc(channels, sizeof(float *));
    return s;
}

/* pcm_all: channels * blocksize[W] floats, channel-major (the mdct_backward outputs) */
int ref_synth_blockin(void *h, int W, float *pcm_all)
{
    ref_synth *s = h;
    codec_setup_info *ci = s->vi.codec_setup;
    long n = ci->blocksizes[W];
    for (int c = 0; c < s->vi.channels; c++) s->pcm[c] = pcm_all + (size_t)c * n;
    s->vb.W = W;
    s->vb.pcm = s->pcm;
    s->vb.pcmend = n;
    /* easyav1 never sets packetno (src/easyav1.c:2436-2439): sequence stays constant */
    s->vb.sequence = 0;
    return vorbis_synthesis_blockin(&s->v, &s->vb);
}

/* copies the available samples channel-major into out (cap samples per channel), returns count */
int ref_synth_pcmout(void *h, float *out, int cap)
{
    ref_synth *s = h;
    float **pcm = NULL;   /* left untouched by pcmout when nothing is available */
    int n = vorbis_synthesis_pcmout(&s->v, &pcm);
    if (n > cap) return -n;
    for (int c = 0; n > 0 && c < s->vi.channels; c++) memcpy(out + (size_t)c * cap, pcm[c], n * sizeof(float));
    return n;
}

int ref_synth_read(void *h, int n) { return vorbis_synthesis_read(&((ref_synth *)h)->v, n); }
int ref_synth_restart(void *h) { return vorbis_synthesis_restart(&((ref_synth *)h)->v); }

void ref_synth_free(void *h)
{
    ref_synth *s = h;
    vorbis_dsp_clear(&s->v);
    /* codec_setup was filled by hand: let vorbis_info_clear release it */
    vorbis_info_clear(&s->vi);
    free(s->pcm);
    free(s);
}

/* ------------------------------------------------------------------------------------------
 * bench.py's audio cpu_baseline (kind "reference"): the reference's OWN mdct_backward (minivorbis.h:7054) and
 * vorbis_synthesis_blockin / pcmout / read (:9195 / :9411 / :9426) over a whole stream of blocks, the loop in C and the clock around
 * it (no per-transform ctypes call), on `threads` host threads with a dsp state each -- what SURVEY.md probe 4 measured by hand.
 * spectra: [nb][channels][bs1 / 2] floats (a short block uses the first bs0 / 2 of its row); returns the transforms done by all
 * threads together, *seconds = wall time of the slowest thread.
 * ------------------------------------------------------------------------------------------ */
#include <pthread.h>
#include <time.h>

typedef struct {
    int channels, bs0, bs1, nb, repeats;
    const unsigned char *flags;
    const float *spectra;
    double seconds;
    double checksum;
} ref_bench_arg;

static void *ref_bench_thread(void *p)
{
    ref_bench_arg *a = p;
    const int ch = a->channels;
    ref_synth *s = ref_synth_new(ch, a->bs0, a->bs1);
    mdct_lookup m[2];
    mdct_init(&m[0], a->bs0);
    mdct_init(&m[1], a->bs1);
    float *block = malloc(sizeof(float) * (size_t)ch * a->bs1);
    struct timespec t0, t1;
    double sum = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < a->repeats; r++) {
        vorbis_synthesis_restart(&s->v);
        for (int k = 0; k < a->nb; k++) {
            const int W = a->flags[k];
            const long n = W ? a->bs1 : a->bs0;
            for (int c = 0; c < ch; c++) {
                /* mapping0_inverse transforms in place (minivorbis.h:17825): the spectrum sits in the first half of the block's buffer */
                float *pcm = block + (size_t)c * n;
                memcpy(pcm, a->spectra + ((size_t)k * ch + c) * (a->bs1 / 2), sizeof(float) * (n / 2));
                mdct_backward(&m[W], pcm, pcm);
            }
            ref_synth_blockin(s, W, block);
            float **out = NULL;
            const int got = vorbis_synthesis_pcmout(&s->v, &out);
            if (got > 0) sum += out[0][0] + out[ch - 1][got - 1];
            vorbis_synthesis_read(&s->v, got);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    a->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    a->checksum = sum;
    free(block);
    mdct_clear(&m[0]);
    mdct_clear(&m[1]);
    ref_synth_free(s);
    return NULL;
}

long long ref_bench_synth(int channels, int bs0, int bs1, const unsigned char *flags, int nb, const float *spectra, int threads, int repeats, double *seconds)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    ref_bench_arg arg[256];
    for (int t = 0; t < threads; t++) {
        ref_bench_arg a = {channels, bs0, bs1, nb, repeats, flags, spectra, 0.0, 0.0};
        arg[t] = a;
        pthread_create(&th[t], NULL, ref_bench_thread, &arg[t]);
    }
    double worst = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        if (arg[t].seconds > worst) worst = arg[t].seconds;
    }
    *seconds = worst;
    return (long long)threads * repeats * nb * channels;
}

/* ------------------------------------------------------------------------------------------
 * Full packet decode through the reference, the way easyav1 drives it (src/easyav1.c:1440-1480,
 * :2458-2490): three header packets through vorbis_synthesis_headerin, then per audio packet
 * vorbis_synthesis + vorbis_synthesis_blockin + pcmout/read.  Pins the packet front end of
 * easyav1_amd/csrc/vorbis_front.cpp.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    vorbis_info vi;
    vorbis_comment vc;
    vorbis_dsp_state v;
    vorbis_block vb;
    int ready;
} ref_stream;

void *ref_stream_new(const unsigned char *id, long id_n, const unsigned char *cm, long cm_n, const unsigned char *su, long su_n)
{
    ref_stream *s = calloc(1, sizeof(*s));
    vorbis_info_init(&s->vi);
    vorbis_comment_init(&s->vc);
    const unsigned char *hp[3] = {id, cm, su};
    const long hn[3] = {id_n, cm_n, su_n};
    for (int i = 0; i < 3; i++) {
        ogg_packet op;
        memset(&op, 0, sizeof op);
        op.packet = (unsigned char *)hp[i];
        op.bytes = hn[i];
        op.b_o_s = i == 0;
        op.packetno = i;
        if (vorbis_synthesis_headerin(&s->vi, &s->vc, &op) < 0) goto fail;
    }
    if (vorbis_synthesis_init(&s->v, &s->vi)) goto fail;
    vorbis_block_init(&s->v, &s->vb);
    s->ready = 1;
    return s;
fail:
    vorbis_comment_clear(&s->vc);
    vorbis_info_clear(&s->vi);
    free(s);
    return NULL;
}

int ref_stream_channels(void *h) { return ((ref_stream *)h)->vi.channels; }
long ref_stream_rate(void *h) { return ((ref_stream *)h)->vi.rate; }
int ref_stream_blocksize(void *h, int which) { return (int)((codec_setup_info *)((ref_stream *)h)->vi.codec_setup)->blocksizes[which]; }

/* decodes one packet; out: channels x cap floats (planar); returns the sample count, -1 if the packet
 * was rejected by vorbis_synthesis, -2 if cap is too small */
int ref_stream_decode(void *h, const unsigned char *packet, long bytes, float *out, int cap)
{
    ref_stream *s = h;
    ogg_packet op;
    memset(&op, 0, sizeof op);
    op.packet = (unsigned char *)packet;
    op.bytes = bytes;
    if (vorbis_synthesis(&s->vb, &op) != 0) return -1;
    vorbis_synthesis_blockin(&s->v, &s->vb);
    float **pcm = NULL;   /* left untouched by pcmout when nothing is available */
    int n = vorbis_synthesis_pcmout(&s->v, &pcm);
    if (n > cap) return -2;
    for (int c = 0; n > 0 && c < s->vi.channels; c++) memcpy(out + (size_t)c * cap, pcm[c], n * sizeof(float));
    vorbis_synthesis_read(&s->v, n);
    return n;
}

int ref_stream_restart(void *h) { return vorbis_synthesis_restart(&((ref_stream *)h)->v); }

void ref_stream_free(void *h)
{
    ref_stream *s = h;
    vorbis_block_clear(&s->vb);
    vorbis_dsp_clear(&s->v);
    vorbis_comment_clear(&s->vc);
    vorbis_info_clear(&s->vi);
    free(s);
}

const float *ref_floor1_db_table(void) { return FLOOR1_fromdB_LOOKUP; }
