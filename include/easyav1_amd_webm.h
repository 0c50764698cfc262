/*
 * easyav1_amd_webm.h -- C ABI of the WebM reader: the demux side
 * of the path.  easyav1 reads its input through nestegg (reference:
 * /root/reference/ext/nestegg/nestegg.c; call sites /root/reference/src/easyav1.c:1279-1506 track
 * discovery through nestegg_track_* and :1904-2017 packet loop through nestegg_read_packet).
 * This reader covers what easyav1 asks of nestegg: EBML header check, Segment / Info / Tracks
 * parsing, codec private data (split per header for Vorbis), sequential packet reads from Clusters
 * (SimpleBlock and BlockGroup, no / Xiph / EBML / fixed lacing, unknown-size Segments and Clusters,
 * Void / CRC-32 / unknown elements skipped), timestamps in nanoseconds.  Host code (demuxing stays
 * on the CPU by BASELINE.json's north_star).  Seeking through Cues: ea1_webm_seek.  Not covered: encryption,
 * block additions.
 *
 * Conventions: as easyav1_amd_audio.h (0 = OK, negative EA1_E* code, ea1_last_error()).
 */
#ifndef EASYAV1_AMD_WEBM_H
#define EASYAV1_AMD_WEBM_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_audio.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { EA1_WEBM_TRACK_VIDEO = 0, EA1_WEBM_TRACK_AUDIO = 1, EA1_WEBM_TRACK_UNKNOWN = 2 };   /* = NESTEGG_TRACK_* */
enum { EA1_WEBM_CODEC_VP8 = 0, EA1_WEBM_CODEC_VORBIS = 1, EA1_WEBM_CODEC_VP9 = 2, EA1_WEBM_CODEC_OPUS = 3,
       EA1_WEBM_CODEC_AV1 = 4, EA1_WEBM_CODEC_UNKNOWN = 0x7fffffff };                       /* = NESTEGG_CODEC_* */

typedef struct ea1_webm ea1_webm;

typedef struct ea1_webm_track {
    int type;                  /* EA1_WEBM_TRACK_*                                              */
    int codec;                 /* EA1_WEBM_CODEC_*                                              */
    uint64_t number;           /* TrackNumber as blocks reference it                            */
    uint64_t default_duration; /* ns per frame, 0 = absent (easyav1 derives the fps from it)    */
    uint32_t width, height;    /* video: PixelWidth / PixelHeight                               */
    double rate;               /* audio: SamplingFrequency                                      */
    uint32_t channels;         /* audio                                                         */
    uint32_t codec_data_count; /* Vorbis: 3 (the Xiph-laced headers), else 1 or 0               */
    uint64_t codec_delay;      /* ns, CodecDelay (nestegg_audio_params.codec_delay: easyav1 adds it to the audio offset, src/easyav1.c:1495, :3909); 0 = absent */
} ea1_webm_track;

#define EA1_WEBM_MAX_FRAMES 256
typedef struct ea1_webm_packet {
    uint32_t track;            /* track INDEX (order of the TrackEntry elements), as nestegg   */
    uint64_t tstamp;           /* ns: (cluster timecode + block timecode) * TimecodeScale      */
    uint64_t duration;         /* ns, BlockDuration * TimecodeScale; 0 = not given             */
    int keyframe;              /* 1 yes, 0 no (SimpleBlock flag; BlockGroup with a ReferenceBlock), -1 unknown */
    uint32_t count;            /* frames in the block (lacing)                                 */
    const uint8_t *data[EA1_WEBM_MAX_FRAMES];   /* pointers into the caller's buffer           */
    size_t size[EA1_WEBM_MAX_FRAMES];
} ea1_webm_packet;

/* `data` is borrowed and must outlive the reader (easyav1_init_from_memory semantics,
 * src/easyav1.c:1642-1681) */
int ea1_webm_open_memory(ea1_webm **out, const uint8_t *data, size_t size);
/* The same over a buffer that is FILLED ON DEMAND (easyav1_init_from_custom_stream / _file / _filename: the reference reads through the
 * caller's callbacks as it goes, src/easyav1.c:717-848): `data` is a mapping of the stream's size whose pages the owner brings in when the
 * reader calls need(userdata, offset, length) before it touches [offset, offset + length).  Pages, once in, stay (packets point into them). */
int ea1_webm_open_paged(ea1_webm **out, const uint8_t *data, size_t size, void (*need)(void *userdata, size_t offset, size_t length), void *userdata);
void ea1_webm_destroy(ea1_webm *w);

int ea1_webm_track_count(const ea1_webm *w, uint32_t *count);
int ea1_webm_track_info(const ea1_webm *w, uint32_t track, ea1_webm_track *out);
/* item-th private data blob of the track (nestegg_track_codec_data) */
int ea1_webm_codec_data(const ea1_webm *w, uint32_t track, uint32_t item, const uint8_t **data, size_t *size);
/* Segment duration in ns (EA1_ESTATE if the file has none) and TimecodeScale */
int ea1_webm_duration(const ea1_webm *w, uint64_t *ns);
int ea1_webm_tstamp_scale(const ea1_webm *w, uint64_t *scale);

/* next packet in file order: returns 1 and fills `out`, 0 at the end of the stream, negative on a
 * malformed file (nestegg_read_packet: 1 / 0 / -1) */
int ea1_webm_read_packet(ea1_webm *w, ea1_webm_packet *out);
/* back to the first cluster */
int ea1_webm_rewind(ea1_webm *w);
/* Cues (nestegg_track_seek / nestegg_get_cue_point, ext/nestegg/nestegg.c:2304-2433; src/easyav1.c:2942,:3026): positions the reader at the Cluster
 * of the last cue point of `track` (index) at or before `ns`.  EA1_ENOENT: the file has no such cue point (no Cues at all: ea1_webm_has_cues
 * = 0) -- the caller walks from the first cluster instead.  The index is located through the SeekHead and read on the first call. */
int ea1_webm_has_cues(ea1_webm *w);
int ea1_webm_seek(ea1_webm *w, uint32_t track, uint64_t ns);

#ifdef __cplusplus
}
#endif
#endif
