/*
 * easyav1.h -- the public interface of easyav1 as exported by libeasyav1.so of THIS repository
 * (easyav1_amd/csrc/easyav1_api.cpp): the drop-in boundary of SURVEY.md section 8(b).
 *
 * Binary compatible with the reference's src/easyav1.h (names, enumerator values, struct layouts,
 * 34 entry points; checked by tests/test_easyav1_abi.py, which also builds the reference's own
 * tools/easyav1_benchmark.c against this header where the reference tree is present).  Every
 * declaration cites the reference declaration it matches (easyav1.h:line) and the implementation
 * it replaces (easyav1.c:line), paths relative to /root/reference/src/.
 *
 * What is behind it here: demux = the WebM reader of easyav1_amd_webm.h; audio = Vorbis packets ->
 * host front end -> GPU IMDCT / window / overlap-add (easyav1_amd_vorbis.h) -> the 4096-sample PCM
 * hand-off buffer; video = track facts and OBU / sequence-header parsing on the host, frames from
 * the GPU reconstruction path (easyav1_amd_video.h) for tracks that carry parsed frames.  See
 * INTEGRATION.md for the state of each entry point.
 */
#ifndef EASYAV1_H
#define EASYAV1_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types (easyav1.h:77-112) -------------------------------------------------------- */
typedef struct easyav1_t easyav1_t;      /* opaque instance                                      */
typedef uint64_t easyav1_timestamp;      /* milliseconds                                         */
typedef enum { EASYAV1_FALSE = 0, EASYAV1_TRUE = 1 } easyav1_bool;

/* 0 / 1 / 2 are what functions return (FINISHED only from easyav1_decode_*); the negative values
 * are detail codes kept inside the instance, which easyav1_get_status folds into ERROR
 * (easyav1.c:3642-3656) */
typedef enum {
    EASYAV1_STATUS_ERROR = 0, EASYAV1_STATUS_OK = 1, EASYAV1_STATUS_FINISHED = 2,
    EASYAV1_STATUS_INVALID_ARGUMENT = -1, EASYAV1_STATUS_OUT_OF_MEMORY = -2, EASYAV1_STATUS_IO_ERROR = -3,
    EASYAV1_STATUS_DECODER_ERROR = -4, EASYAV1_STATUS_NOT_IMPLEMENTED = -5, EASYAV1_STATUS_INVALID_STATE = -6,
    EASYAV1_STATUS_INVALID_DATA = -7, EASYAV1_STATUS_UNSUPPORTED = -8
} easyav1_status;

/* ---- caller-supplied input (easyav1.h:114-160) ---------------------------------------------- */
/* read: exactly `size` bytes -> 1, end of input -> 0, failure -> -1;  seek: origin as fseek, 0 = ok;
 * tell: position or -1 */
typedef int (*easyav1_read_func)(void *buffer, size_t size, void *userdata);
typedef int (*easyav1_seek_func)(int64_t offset, int origin, void *userdata);
typedef int64_t (*easyav1_tell_func)(void *userdata);
typedef struct {
    easyav1_read_func read_func;
    easyav1_seek_func seek_func;
    easyav1_tell_func tell_func;
    void *userdata;
} easyav1_stream;

/* ---- picture description (easyav1.h:162-285), values follow the AV1 sequence header ---------- */
typedef enum {
    EASYAV1_PIXEL_LAYOUT_UNKNOWN = 0, EASYAV1_PIXEL_LAYOUT_YUV400 = 1, EASYAV1_PIXEL_LAYOUT_YUV420 = 2,
    EASYAV1_PIXEL_LAYOUT_YUV422 = 3, EASYAV1_PIXEL_LAYOUT_YUV444 = 4
} easyav1_pixel_layout;
typedef enum { EASYAV1_COLOR_SPACE_UNKNOWN = 0, EASYAV1_COLOR_SPACE_LIMITED = 1, EASYAV1_COLOR_SPACE_FULL = 2 } easyav1_color_space;
typedef enum {
    EASYAV1_BITS_PER_COLOR_UNKNOWN = 0, EASYAV1_BITS_PER_COLOR_8 = 1, EASYAV1_BITS_PER_COLOR_10 = 2, EASYAV1_BITS_PER_COLOR_12 = 3
} easyav1_bits_per_color;
typedef enum {
    EASYAV1_COLOR_PRIMARIES_UNSPECIFIED = 0, EASYAV1_COLOR_PRIMARIES_BT709 = 1, EASYAV1_COLOR_PRIMARIES_UNKNOWN = 2,
    EASYAV1_COLOR_PRIMARIES_BT470M = 3, EASYAV1_COLOR_PRIMARIES_BT470BG = 4, EASYAV1_COLOR_PRIMARIES_BT601 = 5,
    EASYAV1_COLOR_PRIMARIES_SMPTE240 = 6, EASYAV1_COLOR_PRIMARIES_FILM = 7, EASYAV1_COLOR_PRIMARIES_BT2020 = 8,
    EASYAV1_COLOR_PRIMARIES_XYZ = 9, EASYAV1_COLOR_PRIMARIES_SMPTE431 = 10, EASYAV1_COLOR_PRIMARIES_SMPTE432 = 11,
    EASYAV1_COLOR_PRIMARIES_EBU3213 = 12
} easyav1_color_primaries;
typedef enum {
    EASYAV1_TRANSFER_CHARACTERISTICS_UNSPECIFIED = 0, EASYAV1_TRANSFER_CHARACTERISTICS_BT709 = 1,
    EASYAV1_TRANSFER_CHARACTERISTICS_UNKNOWN = 2, EASYAV1_TRANSFER_CHARACTERISTICS_BT470M = 3,
    EASYAV1_TRANSFER_CHARACTERISTICS_BT470BG = 4, EASYAV1_TRANSFER_CHARACTERISTICS_BT601 = 5,
    EASYAV1_TRANSFER_CHARACTERISTICS_SMPTE240 = 6, EASYAV1_TRANSFER_CHARACTERISTICS_LINEAR = 7,
    EASYAV1_TRANSFER_CHARACTERISTICS_LOG_100 = 8, EASYAV1_TRANSFER_CHARACTERISTICS_LOG_100_SQRT = 9,
    EASYAV1_TRANSFER_CHARACTERISTICS_IEC61966 = 10, EASYAV1_TRANSFER_CHARACTERISTICS_BT1361 = 11,
    EASYAV1_TRANSFER_CHARACTERISTICS_SRGB = 12, EASYAV1_TRANSFER_CHARACTERISTICS_BT2020_10 = 13,
    EASYAV1_TRANSFER_CHARACTERISTICS_BT2020_12 = 14, EASYAV1_TRANSFER_CHARACTERISTICS_SMPTE2084 = 15,
    EASYAV1_TRANSFER_CHARACTERISTICS_SMPTE428 = 16, EASYAV1_TRANSFER_CHARACTERISTICS_HLG = 17
} easyav1_transfer_characteristics;
typedef enum {
    EASYAV1_MATRIX_COEFFICIENTS_UNSPECIFIED = 0, EASYAV1_MATRIX_COEFFICIENTS_IDENTITY = 1, EASYAV1_MATRIX_COEFFICIENTS_BT709 = 2,
    EASYAV1_MATRIX_COEFFICIENTS_UNKNOWN = 3, EASYAV1_MATRIX_COEFFICIENTS_FCC = 4, EASYAV1_MATRIX_COEFFICIENTS_BT470BG = 5,
    EASYAV1_MATRIX_COEFFICIENTS_BT601 = 6, EASYAV1_MATRIX_COEFFICIENTS_SMPTE240 = 7, EASYAV1_MATRIX_COEFFICIENTS_SMPTE_YCGCO = 8,
    EASYAV1_MATRIX_COEFFICIENTS_BT2020_NCL = 9, EASYAV1_MATRIX_COEFFICIENTS_BT2020_CL = 10,
    EASYAV1_MATRIX_COEFFICIENTS_SMPTE2085 = 11, EASYAV1_MATRIX_COEFFICIENTS_CHROMATICITY_NCL = 12,
    EASYAV1_MATRIX_COEFFICIENTS_CHROMATICITY_CL = 13, EASYAV1_MATRIX_COEFFICIENTS_ICTCP = 14
} easyav1_matrix_coefficients;
typedef enum {
    EASYAV1_CHROMA_SAMPLE_POSITION_UNKNOWN = 0, EASYAV1_CHROMA_SAMPLE_POSITION_VERTICAL = 1, EASYAV1_CHROMA_SAMPLE_POSITION_COLOCATED = 2
} easyav1_chroma_sample_position;

/* one decoded picture (easyav1.h:288-304; filled at easyav1.c:3557-3568).  The plane pointers stay
 * valid until the next easyav1_get_video_frame / seek / destroy; U and V share stride[1] == stride[2] */
typedef struct {
    struct {
        easyav1_pixel_layout pixel_layout;
        easyav1_bits_per_color bits_per_color;
        easyav1_color_space color_space;
        easyav1_color_primaries color_primaries;
        easyav1_transfer_characteristics transfer_characteristics;
        easyav1_matrix_coefficients matrix_coefficients;
        easyav1_chroma_sample_position chroma_sample_position;
        unsigned int width;
        unsigned int height;
    } properties;
    easyav1_timestamp timestamp;
    const void *data[3];
    size_t stride[3];
} easyav1_video_frame;

/* PCM collected since the last easyav1_get_audio_frame, at most 4096 samples per channel
 * (easyav1.h:309-319).  bytes = samples * 4, times `channels` when interlaced.  Interlaced:
 * sample s of channel c at interlaced[s * channels + c]; otherwise deinterlaced[c][s] */
typedef struct {
    unsigned int channels;
    unsigned int samples;
    easyav1_timestamp timestamp;
    size_t bytes;
    union {
        const float **deinterlaced;
        const float *interlaced;
    } pcm;
} easyav1_audio_frame;

typedef void (*easyav1_video_callback)(const easyav1_video_frame *frame, void *userdata);
typedef void (*easyav1_audio_callback)(const easyav1_audio_frame *frame, void *userdata);

typedef enum { EASYAV1_LOG_LEVEL_ERROR, EASYAV1_LOG_LEVEL_WARNING, EASYAV1_LOG_LEVEL_INFO } easyav1_log_level_t;

/* easyav1.h:412-428; defaults (easyav1.c:307-323): video, audio, skip_unprocessed_frames and
 * interlace_audio on, everything else 0 / NULL, log level WARNING */
typedef struct {
    easyav1_bool enable_video;
    easyav1_bool enable_audio;
    easyav1_bool skip_unprocessed_frames;
    easyav1_bool interlace_audio;
    easyav1_bool close_handle_on_destroy;   /* easyav1_init_from_file: fclose the FILE in easyav1_destroy */
    struct {
        easyav1_video_callback video;
        easyav1_audio_callback audio;
        void *userdata;
    } callbacks;
    unsigned int video_track;               /* n-th video / audio track of the file */
    unsigned int audio_track;
    easyav1_bool use_fast_seeking;
    int64_t audio_offset_time;              /* ms added to audio timestamps */
    easyav1_log_level_t log_level;
} easyav1_settings;

/* ---- the 34 entry points: easyav1.h declaration -> easyav1.c implementation ------------------ */
easyav1_settings easyav1_default_settings(void);                                                        /* :449 -> :3637 */

/* constructors return NULL on failure; a NULL `settings` means the defaults */
easyav1_t *easyav1_init_from_filename(const char *filename, const easyav1_settings *settings);          /* :460 -> :1717 */
easyav1_t *easyav1_init_from_file(FILE *f, const easyav1_settings *settings);                           /* :471 -> :1683 */
easyav1_t *easyav1_init_from_memory(const void *data, size_t size, const easyav1_settings *settings);   /* :483 -> :1642, `data` is borrowed */
easyav1_t *easyav1_init_from_custom_stream(const easyav1_stream *stream, const easyav1_settings *settings); /* :494 -> :1569 */

/* one packet / up to a timestamp / for a time span; OK, FINISHED or ERROR (errors are sticky) */
easyav1_status easyav1_decode_next(easyav1_t *easyav1);                                                 /* :504 -> :2618 */
easyav1_status easyav1_decode_until(easyav1_t *easyav1, easyav1_timestamp timestamp);                   /* :516 -> :2774 */
easyav1_status easyav1_decode_for(easyav1_t *easyav1, easyav1_timestamp time);                          /* :528 -> :2789 */

/* wall-clock paced decoding on a thread of the instance; callbacks then run on that thread */
easyav1_status easyav1_play(easyav1_t *easyav1);                                                        /* :551 -> :2855 */
void easyav1_stop(easyav1_t *easyav1);                                                                  /* :564 -> :2895 */

easyav1_status easyav1_seek_forward(easyav1_t *easyav1, easyav1_timestamp time);                        /* :575 -> :3238 */
easyav1_status easyav1_seek_backward(easyav1_t *easyav1, easyav1_timestamp time);                       /* :586 -> :3248 */
easyav1_status easyav1_seek_to_timestamp(easyav1_t *easyav1, easyav1_timestamp timestamp);              /* :597 -> :3218 */

easyav1_status easyav1_get_status(easyav1_t *easyav1);                                                  /* :615 -> :3642 */
easyav1_timestamp easyav1_get_current_timestamp(easyav1_t *easyav1);                                    /* :624 -> :3658 */
easyav1_bool easyav1_has_video_track(const easyav1_t *easyav1);                                         /* :634 -> :3674 */
easyav1_bool easyav1_has_audio_track(const easyav1_t *easyav1);                                         /* :644 -> :3684 */
unsigned int easyav1_get_total_video_tracks(const easyav1_t *easyav1);                                  /* :655 -> :3754 */
unsigned int easyav1_get_total_audio_tracks(const easyav1_t *easyav1);                                  /* :665 -> :3759 */
unsigned int easyav1_get_video_width(const easyav1_t *easyav1);                                         /* :675 -> :3694 */
unsigned int easyav1_get_video_height(const easyav1_t *easyav1);                                        /* :685 -> :3704 */
unsigned int easyav1_get_video_fps(const easyav1_t *easyav1);                                           /* :695 -> :3714 */
unsigned int easyav1_get_audio_channels(const easyav1_t *easyav1);                                      /* :705 -> :3724 */
unsigned int easyav1_get_audio_sample_rate(const easyav1_t *easyav1);                                   /* :715 -> :3734 */

/* a picture whose timestamp has been reached is waiting; each picture is handed out once */
easyav1_bool easyav1_has_video_frame(easyav1_t *easyav1);                                               /* :725 -> :3267 */
const easyav1_video_frame *easyav1_get_video_frame(easyav1_t *easyav1);                                 /* :738 -> :3509 */
uint64_t easyav1_get_total_video_frames_processed(easyav1_t *easyav1);                                  /* :748 -> :3573 */

/* 4096 samples are waiting / take what is waiting (NULL when nothing is) */
easyav1_bool easyav1_is_audio_buffer_filled(const easyav1_t *easyav1);                                  /* :758 -> :3589 */
const easyav1_audio_frame *easyav1_get_audio_frame(easyav1_t *easyav1);                                 /* :771 -> :3604 */

easyav1_timestamp easyav1_get_duration(const easyav1_t *easyav1);                                       /* :781 -> :3744 */
easyav1_settings easyav1_get_current_settings(const easyav1_t *easyav1);                                /* :791 -> :3785 */
easyav1_status easyav1_update_settings(easyav1_t *easyav1, const easyav1_settings *settings);           /* :804 -> :3854 */
easyav1_bool easyav1_is_finished(easyav1_t *easyav1);                                                   /* :814 -> :3764 */

/* frees the instance and sets *easyav1 to NULL (easyav1.c:3990-4039) */
void easyav1_destroy(easyav1_t **easyav1);                                                              /* :822 -> :3990 */

#ifdef __cplusplus
}
#endif
#endif /* EASYAV1_H */
