/*
 * easyav1_amd_audio.h -- C ABI of the MI355X Vorbis synthesis path (IMDCT + window + overlap-add).
 *
 * Drop-in boundary: these entry points replace the *bodies* behind the libvorbis calls that
 * easyav1's decode_audio makes (reference: /root/reference/src/easyav1.c:2458-2490) and the
 * synthesis setup in init_audio (:1474).  Everything before the IMDCT (packet parse, floor,
 * residue, channel coupling: minivorbis.h:17746-17819) stays on the host and hands over the
 * dequantised, floor-multiplied spectra -- the hand-off of mapping0_inverse (minivorbis.h:17823).
 *
 *   reference routine (ext/minivorbis/minivorbis.h)            replaced by
 *   ---------------------------------------------------------  ---------------------------------
 *   mdct_init                     :6709  (x2, _vds_shared_init :8671)   ea1_vorbis_synth_create
 *   _vorbis_window_get            :11992 (window tables)               ea1_vorbis_synth_create
 *   mdct_backward                 :7054  (per channel, :17823-17826)   ea1_vorbis_synth_decode /
 *                                                                      ea1_mdct_backward_batch
 *   vorbis_synthesis_blockin      :9195                                ea1_vorbis_synth_decode
 *   vorbis_synthesis_pcmout/_read :9411 / :9426                        ea1_vorbis_synth_decode
 *   vorbis_synthesis_restart      :9159  (seek, easyav1.c:3077)        ea1_vorbis_synth_restart
 *   vorbis_dsp_clear              (easyav1.c:3976)                     ea1_vorbis_synth_destroy
 *   decode_audio copy-out + 4096-sample ring  easyav1.c:2471-2532      ea1_audio_ring_*
 *   easyav1_get_audio_frame       easyav1.c:3604                       ea1_audio_ring_get
 *
 * The GPU path is batched: the pull-based API hands over one packet at a time, so the shim
 * decodes ahead over the queued packets (easyav1 pre-queues them, easyav1.c:2077) and the state
 * carried between calls (previous block's right half, lW, "first block" flag) makes N calls of
 * one block bit-identical to one call of N blocks and to the reference called block by block.
 *
 * Conventions: plain pointers and sizes only.  `d_` pointers are device (HBM) addresses on the
 * synth's device, everything else is host memory.  `stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  Returns 0 on success, a negative EA1_E* code otherwise; errors are
 * also kept in a thread-local string (ea1_last_error).  No call falls back to a CPU path.
 */
#ifndef EASYAV1_AMD_AUDIO_H
#define EASYAV1_AMD_AUDIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EA1_OK 0
#define EA1_EINVAL (-1)    /* bad argument (NULL, size not a power of two, ...)            */
#define EA1_ENOMEM (-2)    /* host or device allocation failed                              */
#define EA1_EHIP (-3)      /* a HIP runtime call or kernel launch failed                    */
#define EA1_ENODEV (-4)    /* no usable gfx950 device                                       */
#define EA1_ECAPACITY (-5) /* output buffer too small for what the call would produce      */
#define EA1_ESTATE (-6)    /* call not valid in the object's current state                 */
#define EA1_ENOTSUP (-8)   /* well-formed input that uses a coding tool this implementation does not decode */
#define EA1_ENOENT (-7)    /* the data holds no item of the requested kind (not an error of the data) */

/* last error text of the calling thread ("" if none) */
const char *ea1_last_error(void);
/* number of visible HIP devices (<0: error) */
int ea1_device_count(void);

typedef struct ea1_vorbis_synth ea1_vorbis_synth;

/* blocksize0 <= blocksize1, both powers of two in [64, 8192] (minivorbis.h:8651-8654). */
int ea1_vorbis_synth_create(ea1_vorbis_synth **out, int channels, int blocksize0, int blocksize1,
                            int device);
void ea1_vorbis_synth_destroy(ea1_vorbis_synth *s);

/* vorbis_synthesis_restart: next block is a "first block" again (returns no samples). */
int ea1_vorbis_synth_restart(ea1_vorbis_synth *s);

/*
 * Decode `nblocks` consecutive audio packets' worth of spectra.
 *   d_spectra   [nblocks][channels][blocksize1/2] float32; block k uses the first
 *               blocksize[flags[k]]/2 entries of each row (rest ignored)
 *   flags       host, nblocks bytes: 0 = short block, 1 = long block (vb->W)
 *   d_pcm       [channels][pcm_stride] float32, samples of all blocks back to back per channel
 *   counts      host, optional (may be NULL): samples per channel block k contributed
 *               ( = vorbis_synthesis_pcmout after its blockin: bs[lW]/4 + bs[W]/4, 0 for a
 *               first block )
 *   total       host, optional: sum of counts
 * Asynchronous on `stream`; host outputs (counts/total) are valid on return.
 */
int ea1_vorbis_synth_decode(ea1_vorbis_synth *s, const float *d_spectra, const uint8_t *flags,
                            int nblocks, float *d_pcm, size_t pcm_stride, int *counts,
                            int64_t *total, void *stream);

/*
 * The same for n streams at once (each with its own synth object, block sequence and output): one kernel launch per
 * 16 streams instead of one per stream.  Arrays of n entries; totals (optional): samples per channel of each stream.
 * Results are those of n separate ea1_vorbis_synth_decode calls.
 */
int ea1_vorbis_synth_decode_batch(int n, ea1_vorbis_synth *const *synths, const float *const *d_spectra,
                                  const uint8_t *const *flags, const int *nblocks, float *const *d_pcm,
                                  const size_t *pcm_stride, int64_t *totals, void *stream);
/*
 * Measurement (bench.py's roofline of the audio workload): with profiling on, every decode launch of this process is bracketed by HIP events on
 * its stream; ea1_vorbis_profile_read synchronises, returns the kernels' summed duration and their number since the last read, and resets.
 */
int ea1_vorbis_profile_enable(int on);
int ea1_vorbis_profile_read(double *ms_total, uint64_t *launches);
/* samples the next ea1_vorbis_synth_decode(flags, nblocks) would produce per channel */
int64_t ea1_vorbis_synth_count(const ea1_vorbis_synth *s, const uint8_t *flags, int nblocks);

/*
 * Plain batched mdct_backward (minivorbis.h:7054): `count` transforms of size n.
 *   d_in  : count rows of n/2 float32, row stride in_stride floats
 *   d_out : count rows of n float32, row stride out_stride floats (may not alias d_in)
 */
int ea1_mdct_backward_batch(int n, const float *d_in, size_t in_stride, float *d_out,
                            size_t out_stride, int count, int device, void *stream);

/* ---- easyav1's PCM hand-off ring (host memory; src/easyav1.c:38, :2471-2532, :3604-3630) ---- */
typedef struct ea1_audio_ring ea1_audio_ring;

#define EA1_AUDIO_BUFFER_SIZE 4096

int ea1_audio_ring_create(ea1_audio_ring **out, int channels, int interlace);
void ea1_audio_ring_destroy(ea1_audio_ring *r);
/* start of a packet (easyav1.c:2441-2444) */
void ea1_audio_ring_begin_packet(ea1_audio_ring *r);
/* one pcmout result: channel-major host PCM, `stride` floats per channel (easyav1.c:2468-2492) */
int ea1_audio_ring_push(ea1_audio_ring *r, const float *pcm, size_t stride, int samples);
/* easyav1_get_audio_frame: returns samples (0 = no frame), consumes the frame */
unsigned ea1_audio_ring_get(ea1_audio_ring *r, const float **buffer, size_t *bytes);
/* easyav1_is_audio_buffer_filled (easyav1.c:3589) */
int ea1_audio_ring_is_filled(const ea1_audio_ring *r);

#ifdef __cplusplus
}
#endif
#endif
