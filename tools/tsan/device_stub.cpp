// Device-less stand-in for the GPU half of libeasyav1_amd.so, for the ThreadSanitizer build of the HOST shim (tools/tsan/run.sh; the
// reference builds its Debug configuration with -fsanitize=thread,undefined: /root/reference/CMakeLists.txt:50-58).  TEST INFRASTRUCTURE:
// the HIP runtime calls and the reconstruction / hand-over entry points the shim uses, as plain host code -- pictures are grey, copies are
// memcpy, every event has happened when it is recorded -- so that easyav1_api.cpp + easyav1_video_sink.cpp + the parsers (the code whose
// locking is under test) run on a machine without a GPU and with every one of THEIR memory accesses instrumented.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>

#include "av1_dev.h"
#include "ea1_common.h"
#include "easyav1_amd_audio.h"
#include "easyav1_amd_grain.h"
#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_video.h"
#include "easyav1_amd_vorbis.h"

// ---- HIP runtime: what the shim calls ----
struct StubEvent { std::atomic<int> recorded{0}; };
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(new int(0)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<int *>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new StubEvent); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<StubEvent *>(e)->recorded.store(1, std::memory_order_release); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { (void)reinterpret_cast<StubEvent *>(e)->recorded.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)reinterpret_cast<StubEvent *>(e)->recorded.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memcpy(dst, src, n); return hipSuccess; }
}

// ---- the reconstruction context: grey pictures in one layout ----
struct ea1_recon {
    uint32_t width, height;
    int nchains;
    size_t off[3], stride[3], bytes;
    uint8_t *display;
};
struct ea1_job { uint8_t dst_slot; };

extern "C" hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
extern "C" int ea1_tx_size_index(int log2w, int log2h)   // (lives in av1_recon.cpp in the library)
{
    if (log2w < 2 || log2w > 6 || log2h < 2 || log2h > 6) return -1;
    return ea1::tx_size_index(log2w, log2h);
}
extern "C" int ea1_host_alloc(size_t bytes, void **out) { *out = malloc(bytes ? bytes : 1); return *out ? EA1_OK : EA1_ENOMEM; }
extern "C" void ea1_host_free(void *p) { free(p); }
extern "C" int ea1_recon_warm(int) { return EA1_OK; }
extern "C" int ea1_recon_create_layout(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device, int layout);
extern "C" int ea1_recon_create_batch(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device) { return ea1_recon_create_layout(out, width, height, nchains, device, EA1_LAYOUT_420); }
extern "C" int ea1_recon_create_layout(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int, int layout)
{
    ea1_recon *r = new (std::nothrow) ea1_recon;
    if (!r) return EA1_ENOMEM;
    const int ssx = layout == EA1_LAYOUT_444 ? 0 : 1, ssy = layout == EA1_LAYOUT_420 ? 1 : 0;
    r->width = width, r->height = height, r->nchains = nchains;
    r->stride[0] = (width + 127) & ~size_t(127), r->stride[1] = r->stride[2] = ((width >> ssx) + 127) & ~size_t(127);
    r->off[0] = 0, r->off[1] = r->stride[0] * height, r->off[2] = r->off[1] + r->stride[1] * (height >> ssy);
    r->bytes = r->off[2] + r->stride[2] * (height >> ssy);
    r->display = static_cast<uint8_t *>(malloc(r->bytes));
    memset(r->display, 128, r->bytes);
    *out = r;
    return EA1_OK;
}
extern "C" void ea1_recon_destroy(ea1_recon *r)
{
    if (!r) return;
    free(r->display);
    delete r;
}
extern "C" int ea1_recon_flush(ea1_recon *) { return EA1_OK; }
extern "C" int ea1_recon_picture_layout(const ea1_recon *r, size_t off[3], size_t stride[3], size_t *bytes)
{
    for (int p = 0; p < 3; p++) off[p] = r->off[p], stride[p] = r->stride[p];
    *bytes = r->bytes;
    return EA1_OK;
}
extern "C" int ea1_recon_set_gaussian_sequence(ea1_recon *, const int16_t *) { return EA1_OK; }
extern "C" int ea1_recon_set_upscale_filter(ea1_recon *, const int16_t *) { return EA1_OK; }
extern "C" int ea1_job_prepare(ea1_recon *, const ea1_ir_frame *frame, ea1_job **out)
{
    // (touch the records like the real hand-over does: the parse storage must be stable while a helper reads it)
    unsigned sum = 0;
    for (uint32_t i = 0; i < frame->desc.num_blocks; i += 7) sum += frame->blocks[i].x4;
    ea1_job *j = new (std::nothrow) ea1_job;
    if (!j) return EA1_ENOMEM;
    j->dst_slot = uint8_t(frame->desc.dst_slot + (sum & 0));
    *out = j;
    return EA1_OK;
}
extern "C" int ea1_job_commit(ea1_recon *, ea1_job *const *, int) { return EA1_OK; }
extern "C" void ea1_job_destroy(ea1_job *j) { delete j; }
extern "C" int ea1_recon_submit_batch(ea1_recon *, ea1_job *const *, unsigned, void *) { return EA1_OK; }
extern "C" int ea1_recon_download_batch_async(ea1_recon *r, const int *slots, uint8_t *const *dst, void *)
{
    for (int c = 0; c < r->nchains; c++)
        if (slots[c] >= 0 && dst[c]) memset(dst[c], 128, r->bytes);
    return EA1_OK;
}
extern "C" int ea1_recon_film_grain_chain(ea1_recon *, int, int, const ea1_film_grain *, void *) { return EA1_OK; }
extern "C" int ea1_recon_get_display_chain(ea1_recon *r, int, ea1_picture *out)
{
    memset(out, 0, sizeof *out);
    for (int p = 0; p < 3; p++) out->data[p] = r->display + r->off[p], out->stride[p] = r->stride[p];
    out->width = r->width, out->height = r->height;
    return EA1_OK;
}
// audio on the GPU: not part of this build (the driver switches audio off); a file with an audio track fails to open like on a machine without a device
extern "C" int ea1_vorbis_decoder_create(ea1_vorbis_decoder **, const uint8_t *, size_t, const uint8_t *, size_t, const uint8_t *, size_t, int)
{
    return ea1::fail(EA1_ENODEV, "device stub: no audio");
}
extern "C" void ea1_vorbis_decoder_destroy(ea1_vorbis_decoder *) {}
extern "C" int ea1_vorbis_decoder_info(const ea1_vorbis_decoder *, int *, long *, int *, int *) { return EA1_ENODEV; }
extern "C" int ea1_vorbis_decoder_decode_checked(ea1_vorbis_decoder *, const uint8_t *const *, const size_t *, int, float *, size_t, int *, int64_t *, uint8_t *, void *) { return EA1_ENODEV; }
extern "C" int ea1_vorbis_decoder_restart(ea1_vorbis_decoder *) { return EA1_ENODEV; }
