/*
 * easyav1_amd_vorbis.h -- C ABI of the Vorbis packet front end:
 * "the step before the path".  easyav1 hands Vorbis packets from the WebM demuxer to libvorbis
 * (reference: /root/reference/src/easyav1.c:1440-1480 header packets -> vorbis_synthesis_headerin,
 * :2458 vorbis_synthesis per audio packet).  libvorbis then decodes, ON THE HOST, the packet into
 * one spectrum per channel (ext/minivorbis/minivorbis.h: vorbis_synthesis :12786, mapping0_inverse
 * :17729-17819: floor1 curve decode, residue decode, inverse channel coupling, floor multiply) and
 * runs the IMDCT + overlap-add (the GPU path of easyav1_amd_audio.h).
 *
 * This front end is that host stage, written from the Vorbis I specification, and the glue that
 * feeds its spectra to the GPU synthesis:
 *   packets --(host: this file)--> spectra [block][channel][n/2] --(H2D)--> ea1_vorbis_synth_decode
 *           --(D2H)--> PCM, bit-exact with vorbis_synthesis + vorbis_synthesis_blockin + pcmout.
 * It runs on the CPU because BASELINE.json's north_star keeps bitstream parsing on the host; it is
 * not a fallback for anything the GPU path does.
 *
 * Supported: everything a Vorbis I stream can carry except floor type 0 (obsolete, never produced by
 * any libvorbis release since 2001): codebooks (ordered / sparse, lookup types 0-2), floor 1,
 * residues 0 / 1 / 2, mapping 0 with channel coupling and submaps, any mode set, end-of-packet
 * behaviour of the specification (truncated packets decode what they hold).
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_VORBIS_H
#define EASYAV1_AMD_VORBIS_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_audio.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ea1_vorbis_decoder ea1_vorbis_decoder;

/*
 * The three header packets (identification, comment, setup), exactly the bytes easyav1 passes to
 * vorbis_synthesis_headerin (src/easyav1.c:1455-1470; in WebM they are the Xiph-laced CodecPrivate).
 * device < 0: host-only decoder (spectra only, no GPU synthesis; used by the CPU tests).
 */
int ea1_vorbis_decoder_create(ea1_vorbis_decoder **out, const uint8_t *id_header, size_t id_size,
                              const uint8_t *comment_header, size_t comment_size, const uint8_t *setup_header,
                              size_t setup_size, int device);
void ea1_vorbis_decoder_destroy(ea1_vorbis_decoder *d);

/* stream parameters from the identification header */
int ea1_vorbis_decoder_info(const ea1_vorbis_decoder *d, int *channels, long *rate, int *blocksize0,
                            int *blocksize1);

/*
 * Host stage only: one audio packet -> spectra.
 *   spectra  channels x spectrum_stride floats; the first blocksize[W]/2 entries of each row are
 *            written (what mapping0_inverse holds before mdct_backward, minivorbis.h:17823)
 *   long_block  out: vb->W of the packet
 * Returns EA1_OK, or EA1_EINVAL for a packet that is not a decodable audio packet (libvorbis:
 * OV_ENOTAUDIO / OV_EBADPACKET; easyav1 skips such packets, src/easyav1.c:2458).
 */
int ea1_vorbis_decoder_spectra(ea1_vorbis_decoder *d, const uint8_t *packet, size_t size, float *spectra,
                               size_t spectrum_stride, int *long_block);

/*
 * Decode `npackets` consecutive audio packets to PCM: host spectra -> GPU IMDCT + window +
 * overlap-add (ea1_vorbis_synth_decode).  Undecodable packets are skipped (counts[k] = 0).
 *   pcm      host, channels x pcm_stride floats (planar), samples of all packets back to back
 *   counts   host, optional: samples per channel contributed by packet k
 *   total    host, optional: sum of counts
 * Synchronous (the PCM is on the host when it returns).
 */
int ea1_vorbis_decoder_decode(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes,
                              int npackets, float *pcm, size_t pcm_stride, int *counts, int64_t *total,
                              void *stream);

/* The same, and accepted[k] (host, npackets bytes) tells whether packet k was a decodable audio packet
 * (1) or was skipped (0): easyav1 treats a packet that vorbis_synthesis refuses as a decoder error
 * (src/easyav1.c:2458-2462), so its shim has to know. */
int ea1_vorbis_decoder_decode_checked(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes,
                                      int npackets, float *pcm, size_t pcm_stride, int *counts, int64_t *total,
                                      uint8_t *accepted, void *stream);
/* seek: forget the overlap state (vorbis_synthesis_restart, src/easyav1.c:3077) */
int ea1_vorbis_decoder_restart(ea1_vorbis_decoder *d);

/* floor 1 inverse dB table (Vorbis I specification 10.1), 256 entries: exposed for the parity tests */
const float *ea1_vorbis_floor1_db_table(void);

#ifdef __cplusplus
}
#endif
#endif
