// Host-side PCM hand-off ring with easyav1's exact semantics.
// Mirrors (own code, same behaviour) /root/reference/src/easyav1.c:
//   AUDIO_BUFFER_SIZE :38, decode_audio copy loop :2468-2492,
//   prepare_audio_buffer_for_new_samples :2501-2532, easyav1_get_audio_frame :3604-3630,
//   easyav1_is_audio_buffer_filled :3588-3601.
#include <cstdlib>
#include <cstring>
#include <new>

#include "ea1_common.h"

struct ea1_audio_ring {
    int channels;
    bool interlace;
    float *buffer;
    unsigned samples;
    bool has_samples;
};

extern "C" int ea1_audio_ring_create(ea1_audio_ring **out, int channels, int interlace)
{
    if (!out || channels <= 0) return ea1::fail(EA1_EINVAL, "ea1_audio_ring_create: bad arguments");
    auto *r = new (std::nothrow) ea1_audio_ring();
    if (!r) return ea1::fail(EA1_ENOMEM, "ea1_audio_ring_create: out of memory");
    r->channels = channels;
    r->interlace = interlace != 0;
    r->buffer = static_cast<float *>(calloc(size_t(EA1_AUDIO_BUFFER_SIZE) * channels, sizeof(float)));
    r->samples = 0;
    r->has_samples = false;
    if (!r->buffer) {
        delete r;
        return ea1::fail(EA1_ENOMEM, "ea1_audio_ring_create: out of memory");
    }
    *out = r;
    return EA1_OK;
}

extern "C" void ea1_audio_ring_destroy(ea1_audio_ring *r)
{
    if (!r) return;
    free(r->buffer);
    delete r;
}

extern "C" void ea1_audio_ring_begin_packet(ea1_audio_ring *r)
{
    if (r && !r->has_samples) r->samples = 0;
}

// returns the offset into the incoming samples at which copying starts
static unsigned make_room(ea1_audio_ring *r, int incoming)
{
    const unsigned cap = EA1_AUDIO_BUFFER_SIZE;
    if (unsigned(incoming) > cap) {  // more than a whole ring at once: keep only the newest cap
        r->samples = 0;
        return unsigned(incoming) - cap;
    }
    if (incoming + r->samples <= cap) return 0;
    const unsigned drop = incoming + r->samples - cap;  // oldest samples fall out
    if (r->interlace) {
        memmove(r->buffer, r->buffer + size_t(drop) * r->channels,
                size_t(cap - drop) * r->channels * sizeof(float));
    } else {
        for (int c = 0; c < r->channels; c++)
            memmove(r->buffer + size_t(c) * cap, r->buffer + size_t(c) * cap + drop,
                    (cap - drop) * sizeof(float));
    }
    r->samples = cap - incoming;
    return 0;
}

extern "C" int ea1_audio_ring_push(ea1_audio_ring *r, const float *pcm, size_t stride, int samples)
{
    if (!r || samples < 0 || (samples > 0 && !pcm)) return ea1::fail(EA1_EINVAL, "ea1_audio_ring_push: bad arguments");
    const unsigned cap = EA1_AUDIO_BUFFER_SIZE;
    const unsigned off = make_room(r, samples);
    // the reference copies `samples` values starting at `off`; with off > 0 that would run past
    // the decoded range, so the copy is bounded to what exists (only reachable for > 4096
    // samples per block, which Vorbis blocksizes <= 8192 never produce)
    const int ncopy = samples - int(off);
    for (int c = 0; c < r->channels; c++) {
        const float *src = pcm + size_t(c) * stride + off;
        if (r->interlace) {
            float *dst = r->buffer + size_t(r->samples) * r->channels + c;
            for (int s = 0; s < ncopy; s++) dst[size_t(s) * r->channels] = src[s];
        } else {
            memcpy(r->buffer + size_t(c) * cap + r->samples, src, size_t(ncopy) * sizeof(float));
        }
    }
    r->samples += samples;
    if (r->samples > cap) r->samples = cap;
    if (r->samples > 0) r->has_samples = true;
    return EA1_OK;
}

extern "C" unsigned ea1_audio_ring_get(ea1_audio_ring *r, const float **buffer, size_t *bytes)
{
    if (!r || !r->has_samples) return 0;
    r->has_samples = false;
    if (r->samples == 0) return 0;
    if (bytes) *bytes = size_t(r->samples) * sizeof(float) * (r->interlace ? r->channels : 1);
    if (buffer) *buffer = r->buffer;
    return r->samples;
}

extern "C" int ea1_audio_ring_is_filled(const ea1_audio_ring *r)
{
    return r && r->has_samples && r->samples == EA1_AUDIO_BUFFER_SIZE;
}
