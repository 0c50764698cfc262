// See easyav1_video_sink.h.  One decoding chain; every frame is reconstructed on the sink's HIP
// stream, copied in device layout (one copy) into a pinned host buffer and fenced by an event, so
// that frames submitted ahead of the playback position (the shim submits the packets it has
// prefetched, as the reference's decoder thread runs ahead of the position, src/easyav1.c:1813-1824)
// overlap with the caller consuming earlier ones.
#include <hip/hip_runtime.h>

#include <cstring>
#include <algorithm>
#include <deque>
#include <new>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_video.h"
#include "easyav1_video_sink.h"

namespace ea1 {

namespace {
constexpr size_t kRing = 11;   // decoded pictures that may wait; the oldest is dropped beyond (src/easyav1.c:41, :2141-2143)

struct Pending {
    uint64_t timestamp;
    unsigned vis_w, vis_h;
    int buffer;          // index into VideoSink::buffers
    hipEvent_t done;     // recorded behind the copy
    ea1_job *job;
};
}  // namespace

struct VideoSink {
    unsigned width = 0, height = 0;   // size of the context: the track's (sequence header's) size in whole 8x8 units, at least the first frame's
    unsigned track_w = 0, track_h = 0;
    ea1_recon *ctx = nullptr;
    hipStream_t stream = nullptr;
    size_t plane_off[3] = {0, 0, 0}, plane_stride[3] = {0, 0, 0}, pic_bytes = 0;
    std::vector<uint8_t *> buffers;   // pinned
    std::vector<int> free_buffers;
    std::deque<Pending> waiting;
    int shown = -1;                   // buffer of the picture the caller holds
    std::vector<ea1_job *> retired;   // jobs whose frames are done: destroyed in batches (a destroy drains the device)
};

bool is_parsed_frame_metadata(const uint8_t *payload, size_t size) { return ea1_ir_packet_probe(payload, size) != 0; }

int video_sink_create(VideoSink **out, unsigned width, unsigned height)
{
    *out = nullptr;
    auto *s = new (std::nothrow) VideoSink();
    if (!s) return fail(EA1_ENOMEM, "out of memory");
    s->track_w = (width + 7) & ~7u;
    s->track_h = (height + 7) & ~7u;
    if (ea1_device_count() <= 0) {
        delete s;
        return fail(EA1_ENODEV, "video sink: no HIP device (there is no CPU fallback)");
    }
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s;
        return fail(EA1_EHIP, "hipStreamCreate failed");
    }
    *out = s;
    return EA1_OK;
}

static void retire(VideoSink *s, Pending &p, bool wait)
{
    if (wait) (void)hipEventSynchronize(p.done);
    (void)hipEventDestroy(p.done);
    s->free_buffers.push_back(p.buffer);
    s->retired.push_back(p.job);
}

static void destroy_retired(VideoSink *s)
{
    for (ea1_job *j : s->retired) ea1_job_destroy(j);
    s->retired.clear();
}

void video_sink_flush(VideoSink *s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s->stream);
    for (Pending &p : s->waiting) retire(s, p, false);
    s->waiting.clear();
    destroy_retired(s);
    if (s->ctx) ea1_recon_flush(s->ctx);
}

void video_sink_destroy(VideoSink *s)
{
    if (!s) return;
    video_sink_flush(s);
    if (s->ctx) ea1_recon_destroy(s->ctx);
    for (uint8_t *b : s->buffers) ea1_host_free(b);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static int units(int unit, int size)
{
    const int n = (size + (unit >> 1)) / unit;
    return n < 1 ? 1 : n;
}

int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp)
{
    ea1_ir_storage *st = nullptr;
    ea1_ir_frame f;
    int rc = ea1_ir_packet_parse(payload, size, &st, &f);
    if (rc != EA1_OK) return rc;
    struct Guard {
        ea1_ir_storage *s;
        ~Guard() { ea1_ir_storage_destroy(s); }
    } guard{st};
    const ea1_frame_desc &d = f.desc;
    // the two arrays whose length ea1_job_create derives from the frame size must really be that long
    const uint64_t sbs = uint64_t((d.width + 63) / 64) * ((d.height + 63) / 64);
    if (d.cdef_enable && f.num_sb < sbs) return fail(EA1_EINVAL, "parsed frame: %u CDEF indices for %llu superblocks", f.num_sb, (unsigned long long)sbs);
    const int vis_w = int(d.width) - (d.crop & 15), vis_h = int(d.height) - (d.crop >> 4);
    for (int pl = 0; pl < 3; pl++) {
        if (!d.lr_type[pl]) continue;
        if (d.lr_unit_log2[pl] < 5 || d.lr_unit_log2[pl] > 8) return fail(EA1_EINVAL, "parsed frame: restoration unit size");
        const int ss = pl ? 1 : 0, usz = 1 << d.lr_unit_log2[pl];
        const uint64_t need = uint64_t(units(usz, (vis_h + ss) >> ss)) * units(usz, (vis_w + ss) >> ss);
        if (f.lr_count[pl] < need) return fail(EA1_EINVAL, "parsed frame: %u restoration units of plane %d, %llu needed", f.lr_count[pl], pl, (unsigned long long)need);
    }
    if (!s->ctx) {
        // frames may be smaller than the track's size (resolution switches, scaled references): the context holds the largest
        s->width = std::max(s->track_w, d.width);
        s->height = std::max(s->track_h, d.height);
        rc = ea1_recon_create(&s->ctx, s->width, s->height, 0);
        if (rc != EA1_OK) return rc;
        rc = ea1_recon_picture_layout(s->ctx, s->plane_off, s->plane_stride, &s->pic_bytes);
        if (rc != EA1_OK) return rc;
    }
    // freeing a job drains the device: done when nothing is in flight anyway, or when enough of them have piled up
    if ((s->retired.size() >= 8 && s->waiting.empty()) || s->retired.size() >= 64) destroy_retired(s);
    // a free pinned buffer; beyond kRing waiting pictures the oldest one is given up
    if (s->waiting.size() >= kRing) {
        retire(s, s->waiting.front(), true);
        s->waiting.pop_front();
    }
    if (s->free_buffers.empty()) {
        void *b = nullptr;
        rc = ea1_host_alloc(s->pic_bytes, &b);
        if (rc != EA1_OK) return rc;
        s->buffers.push_back(static_cast<uint8_t *>(b));
        s->free_buffers.push_back(int(s->buffers.size()) - 1);
    }
    ea1_job *job = nullptr;
    const ea1_lr_unit *lr[3] = {f.lr[0], f.lr[1], f.lr[2]};
    rc = ea1_job_create(s->ctx, &f.desc, f.blocks, f.txs, f.coefs, f.aux, f.cdef_idx, lr, &job);
    if (rc != EA1_OK) return rc;
    rc = ea1_recon_submit(s->ctx, job, s->stream);
    if (rc != EA1_OK) {
        ea1_job_destroy(job);
        return rc;
    }
    Pending p;
    p.timestamp = timestamp;
    p.vis_w = unsigned(vis_w);
    p.vis_h = unsigned(vis_h);
    p.buffer = s->free_buffers.back();
    p.job = job;
    s->free_buffers.pop_back();
    const int slot = d.dst_slot;
    uint8_t *dst = s->buffers[size_t(p.buffer)];
    rc = ea1_recon_download_batch_async(s->ctx, &slot, &dst, s->stream);
    if (rc != EA1_OK || hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamSynchronize(s->stream);
        s->free_buffers.push_back(p.buffer);
        ea1_job_destroy(job);
        return rc != EA1_OK ? rc : fail(EA1_EHIP, "hipEventCreate failed");
    }
    (void)hipEventRecord(p.done, s->stream);
    s->waiting.push_back(p);
    return EA1_OK;
}

bool video_sink_has_frame(const VideoSink *s, uint64_t position) { return s && !s->waiting.empty() && s->waiting.front().timestamp <= position; }

bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out)
{
    if (!video_sink_has_frame(s, position)) return false;
    if (s->shown >= 0) {
        s->free_buffers.push_back(s->shown);
        s->shown = -1;
    }
    Pending p = s->waiting.front();
    s->waiting.pop_front();
    (void)hipEventSynchronize(p.done);
    (void)hipEventDestroy(p.done);
    s->retired.push_back(p.job);
    s->shown = p.buffer;
    const uint8_t *base = s->buffers[size_t(p.buffer)];
    for (int pl = 0; pl < 3; pl++) {
        out->data[pl] = base + s->plane_off[pl];
        out->stride[pl] = s->plane_stride[pl];
    }
    out->width = p.vis_w;
    out->height = p.vis_h;
    out->timestamp = p.timestamp;
    return true;
}

}  // namespace ea1
