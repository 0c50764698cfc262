// See easyav1_video_sink.h.  One decoding chain per sink; packets are prepared on helper threads, committed and
// submitted in stream order by one issuer thread; pictures travel to pinned host memory on a second HIP stream, in
// device layout (one copy per picture), each fenced by an event.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_video.h"
#include "easyav1_video_sink.h"

namespace ea1 {

namespace {
constexpr size_t kRing = 11;    // finished pictures that may wait; the oldest is dropped beyond (src/easyav1.c:41, :2141-2143)
constexpr int kMaxCommit = 8;   // prepared frames committed with one launch of the derivation kernels

enum class St { Queued, Preparing, Prepared, Issued, Failed };

struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
};

struct Item {
    uint64_t ticket = 0;
    const uint8_t *payload = nullptr;
    size_t size = 0;
    uint64_t timestamp = 0;
    bool discard = false;
    St state = St::Queued;
    PinBlock *pin = nullptr;         // parse storage (pinned), owned by VideoSink::pins
    ea1_ir_frame frame;
    ea1_job *job = nullptr;
    int rc = EA1_OK;
    std::string msg;
    unsigned vis_w = 0, vis_h = 0;
    int buffer = -1;              // picture buffer: index into VideoSink::buffers
    hipEvent_t done = nullptr;    // behind the picture copy (deliverable frames) or the reconstruction (discarded ones)
    hipEvent_t mid = nullptr;     // deliverable frames: behind the reconstruction, what the copy stream waits for
};

int units(int unit, int size)
{
    const int n = (size + (unit >> 1)) / unit;
    return n < 1 ? 1 : n;
}
}  // namespace

struct VideoSink {
    unsigned width = 0, height = 0;   // size of the context: the track's size in whole 8x8 units, at least the first frame's
    unsigned track_w = 0, track_h = 0;
    int device = 0;
    ea1_recon *ctx = nullptr;
    hipStream_t recon_st = nullptr, copy_st = nullptr;
    size_t plane_off[3] = {0, 0, 0}, plane_stride[3] = {0, 0, 0}, pic_bytes = 0;

    std::mutex mu;
    std::condition_variable cv_work, cv_issue, cv_done;
    std::deque<std::unique_ptr<Item>> items;   // submission order; tickets are consecutive: items[k]->ticket == items[0]->ticket + k
    uint64_t next_ticket = 0, next_issue = 0;  // next_issue: ticket of the first item the issuer has not dealt with
    bool stop = false, issuer_busy = false;
    int preparing = 0;
    int error_rc = EA1_OK;                     // sticky: a frame failed; every later frame of the stream fails with it
    uint64_t error_ticket = 0;
    std::string error_msg;
    std::vector<std::thread> workers;
    std::thread issuer;

    std::vector<std::unique_ptr<PinBlock>> pins;   // pinned parse storage, recycled (the blocks themselves never move)
    std::vector<uint8_t *> buffers;            // pinned picture buffers
    std::vector<int> free_buffers;
    std::vector<hipEvent_t> free_events;
    int shown = -1;                            // buffer of the picture the caller holds
    struct {
        uint64_t ticket;
        hipEvent_t ev;
    } slot_copy[EA1_NUM_REF_SLOTS] = {};       // the picture copy that last read each reference slot (ev == nullptr: none in flight)
};

bool is_parsed_frame_metadata(const uint8_t *payload, size_t size) { return ea1_ir_packet_probe(payload, size) != 0; }

namespace {

Item *find(VideoSink *s, uint64_t ticket)
{
    if (s->items.empty() || ticket < s->items.front()->ticket) return nullptr;
    const uint64_t k = ticket - s->items.front()->ticket;
    return k < s->items.size() ? s->items[size_t(k)].get() : nullptr;
}

// consumer side, under the lock: gives the item's resources back (its event must have completed or been waited for)
void retire(VideoSink *s, Item &it, bool keep_buffer)
{
    if (it.done) {
        for (auto &sc : s->slot_copy)
            if (sc.ev == it.done && sc.ticket == it.ticket) sc.ev = nullptr;
        s->free_events.push_back(it.done);
        it.done = nullptr;
    }
    if (it.mid) s->free_events.push_back(it.mid);
    it.mid = nullptr;
    if (it.buffer >= 0 && !keep_buffer) s->free_buffers.push_back(it.buffer);
    if (it.pin) it.pin->busy = false;
    if (it.job) ea1_job_destroy(it.job);   // never synchronises: the arena returns to the pool behind the job's last submit
    it.job = nullptr;
    it.pin = nullptr;
}

// the host half of one frame, on a helper thread, outside the lock
void prepare(VideoSink *s, Item &it)
{
    ea1_ir_frame &f = it.frame;
    int rc = ea1_ir_packet_parse_into(it.payload, it.size, it.pin->p, it.pin->cap, &f);
    if (rc == EA1_OK) {
        const ea1_frame_desc &d = f.desc;
        // the two arrays whose length the hand-over derives from the frame size must really be that long
        const uint64_t sbs = uint64_t((d.width + 63) / 64) * ((d.height + 63) / 64);
        const int vis_w = int(d.width) - (d.crop & 15), vis_h = int(d.height) - (d.crop >> 4);
        if (d.cdef_enable && f.num_sb < sbs) rc = fail(EA1_EINVAL, "parsed frame: %u CDEF indices for %llu superblocks", f.num_sb, (unsigned long long)sbs);
        for (int pl = 0; pl < 3 && rc == EA1_OK; pl++) {
            if (!d.lr_type[pl]) continue;
            if (d.lr_unit_log2[pl] < 5 || d.lr_unit_log2[pl] > 8) {
                rc = fail(EA1_EINVAL, "parsed frame: restoration unit size");
                break;
            }
            const int ss = pl ? 1 : 0, usz = 1 << d.lr_unit_log2[pl];
            const uint64_t need = uint64_t(units(usz, (vis_h + ss) >> ss)) * units(usz, (vis_w + ss) >> ss);
            if (f.lr_count[pl] < need) rc = fail(EA1_EINVAL, "parsed frame: %u restoration units of plane %d, %llu needed", f.lr_count[pl], pl, (unsigned long long)need);
        }
        it.vis_w = unsigned(vis_w);
        it.vis_h = unsigned(vis_h);
    }
    if (rc == EA1_OK) {
        std::unique_lock<std::mutex> lk(s->mu);
        if (!s->ctx) {
            // frames may be smaller than the track's size (resolution switches, scaled references): the context holds the largest
            s->width = std::max(s->track_w, f.desc.width);
            s->height = std::max(s->track_h, f.desc.height);
            rc = ea1_recon_create(&s->ctx, s->width, s->height, s->device);
            if (rc == EA1_OK) rc = ea1_recon_picture_layout(s->ctx, s->plane_off, s->plane_stride, &s->pic_bytes);
        }
    }
    if (rc == EA1_OK) rc = ea1_job_prepare(s->ctx, &f, &it.job);
    it.rc = rc;
    if (rc != EA1_OK) it.msg = ea1_last_error();
}

void worker_loop(VideoSink *s)
{
    (void)hipSetDevice(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        Item *it = nullptr;
        for (auto &p : s->items)
            if (p->state == St::Queued) {
                it = p.get();
                break;
            }
        if (s->stop) return;
        if (!it || s->error_rc != EA1_OK) {
            s->cv_work.wait(lk);
            continue;
        }
        // parse storage: a free pinned block that is large enough, else a new one (the payload bounds what the arrays need)
        const size_t need = it->size + 64;
        PinBlock *pbp = nullptr;
        for (auto &b : s->pins)
            if (!b->busy && b->cap >= need && !pbp) pbp = b.get();
        if (!pbp)
            for (auto &b : s->pins)
                if (!b->busy && !pbp) pbp = b.get();   // too small: replaced below
        if (!pbp) {
            s->pins.emplace_back(new PinBlock());
            pbp = s->pins.back().get();
        }
        PinBlock &pb = *pbp;
        pb.busy = true;
        it->pin = pbp;
        it->state = St::Preparing;
        s->preparing++;
        lk.unlock();
        int rc = EA1_OK;
        if (pb.cap < need) {   // only this thread touches a busy block
            if (pb.p) ea1_host_free(pb.p);
            pb.p = nullptr;
            pb.cap = 0;
            const size_t cap = (need + need / 4 + 0xFFFFF) & ~size_t(0xFFFFF);
            rc = ea1_host_alloc(cap, &pb.p);
            if (rc == EA1_OK) pb.cap = cap;
        }
        if (rc == EA1_OK) {
            prepare(s, *it);
        } else {
            it->rc = rc;
            it->msg = ea1_last_error();
        }
        lk.lock();
        s->preparing--;
        it->state = it->rc == EA1_OK ? St::Prepared : St::Failed;
        s->cv_issue.notify_all();
        s->cv_done.notify_all();
    }
}

void issuer_loop(VideoSink *s)
{
    (void)hipSetDevice(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        if (!s->items.empty() && s->next_issue < s->items.front()->ticket) s->next_issue = s->items.front()->ticket;
        Item *first = find(s, s->next_issue);
        if (s->stop) return;
        if (!first || (first->state != St::Prepared && first->state != St::Failed)) {
            s->cv_issue.wait(lk);
            continue;
        }
        if (first->state == St::Failed || s->error_rc != EA1_OK) {
            // the stream is broken from here on: later frames would predict from pictures that do not exist
            if (s->error_rc == EA1_OK) {
                s->error_rc = first->rc;
                s->error_msg = first->msg;
                s->error_ticket = first->ticket;
            }
            first->state = St::Failed;
            s->next_issue++;
            s->cv_done.notify_all();
            continue;
        }
        Item *run[kMaxCommit];
        ea1_job *jobs[kMaxCommit];
        int n = 0;
        for (uint64_t t = s->next_issue; n < kMaxCommit; t++) {
            Item *it = find(s, t);
            if (!it || it->state != St::Prepared) break;
            run[n] = it;
            jobs[n++] = it->job;
        }
        // events and picture buffers of the run
        int rc = EA1_OK;
        std::string msg;
        for (int k = 0; k < n && rc == EA1_OK; k++) {
            Item &it = *run[k];
            while (s->free_events.size() < 2) {
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                    rc = EA1_EHIP;
                    msg = "hipEventCreate failed";
                    break;
                }
                s->free_events.push_back(e);
            }
            if (rc != EA1_OK) break;
            it.done = s->free_events.back();
            s->free_events.pop_back();
            if (it.discard) continue;
            it.mid = s->free_events.back();
            s->free_events.pop_back();
            if (s->free_buffers.empty()) {
                void *b = nullptr;
                lk.unlock();   // pinning 12 MB takes a millisecond: not under the lock (only the first dozen frames get here)
                const int arc = ea1_host_alloc(s->pic_bytes, &b);
                if (arc != EA1_OK) msg = ea1_last_error();
                lk.lock();
                if (arc != EA1_OK) {
                    rc = arc;
                    break;
                }
                s->buffers.push_back(static_cast<uint8_t *>(b));
                s->free_buffers.push_back(int(s->buffers.size()) - 1);
            }
            it.buffer = s->free_buffers.back();
            s->free_buffers.pop_back();
        }
        // what the submits must wait for: the copy that still reads the slot a frame is about to overwrite
        hipEvent_t wait_for[kMaxCommit] = {};
        for (int k = 0; k < n; k++) wait_for[k] = s->slot_copy[run[k]->frame.desc.dst_slot & 7].ev;
        s->issuer_busy = true;
        lk.unlock();
        int issued = 0;
        if (rc == EA1_OK) {
            rc = ea1_job_commit(s->ctx, jobs, n);
            if (rc != EA1_OK) msg = ea1_last_error();
        }
        hipEvent_t own_copy[EA1_NUM_REF_SLOTS] = {};   // copies issued inside this run
        for (int k = 0; k < n && rc == EA1_OK; k++) {
            Item &it = *run[k];
            const int slot = it.frame.desc.dst_slot & 7;
            hipEvent_t w = own_copy[slot] ? own_copy[slot] : wait_for[k];
            if (w && hipStreamWaitEvent(s->recon_st, w, 0) != hipSuccess) {
                rc = EA1_EHIP;
                msg = "hipStreamWaitEvent failed";
                break;
            }
            rc = ea1_recon_submit(s->ctx, it.job, s->recon_st);
            if (rc != EA1_OK) {
                msg = ea1_last_error();
                break;
            }
            if (it.discard) {
                if (hipEventRecord(it.done, s->recon_st) != hipSuccess) rc = EA1_EHIP;
            } else {
                // reconstruction -> (event) -> copy stream: the copy of picture n runs beside the reconstruction of n + 1
                uint8_t *dst = s->buffers[size_t(it.buffer)];
                if (hipEventRecord(it.mid, s->recon_st) != hipSuccess || hipStreamWaitEvent(s->copy_st, it.mid, 0) != hipSuccess) rc = EA1_EHIP;
                if (rc == EA1_OK) rc = ea1_recon_download_batch_async(s->ctx, &slot, &dst, s->copy_st);
                if (rc == EA1_OK && hipEventRecord(it.done, s->copy_st) != hipSuccess) rc = EA1_EHIP;
                own_copy[slot] = it.done;
            }
            if (rc != EA1_OK) {
                if (msg.empty()) msg = rc == EA1_EHIP ? "event plumbing of the picture copy failed" : ea1_last_error();
                break;
            }
            issued++;
        }
        if (rc != EA1_OK) {   // whatever did get out must not outlive the resources the failed items give back
            (void)hipStreamSynchronize(s->recon_st);
            (void)hipStreamSynchronize(s->copy_st);
        }
        lk.lock();
        s->issuer_busy = false;
        for (int k = 0; k < n; k++) {
            Item &it = *run[k];
            if (k < issued) {
                it.state = St::Issued;
                if (!it.discard) {
                    auto &sc = s->slot_copy[it.frame.desc.dst_slot & 7];
                    sc.ticket = it.ticket;
                    sc.ev = it.done;
                }
            } else {
                it.state = St::Failed;
                it.rc = rc;
                it.msg = msg;
                if (s->error_rc == EA1_OK) {
                    s->error_rc = rc;
                    s->error_msg = msg;
                    s->error_ticket = it.ticket;
                }
            }
        }
        s->next_issue += uint64_t(n);
        s->cv_done.notify_all();
        s->cv_issue.notify_all();
    }
}

// consumer side, under the lock: the front of the queue that nobody will ask for any more
void sweep(VideoSink *s)
{
    while (!s->items.empty()) {
        Item &it = *s->items.front();
        const bool dead_discard = it.discard && it.state == St::Issued && hipEventQuery(it.done) == hipSuccess;
        const bool dead_failed = it.state == St::Failed && it.ticket < s->next_issue;
        if (!dead_discard && !dead_failed) break;
        retire(s, it, false);
        s->items.pop_front();
    }
    (void)hipGetLastError();   // hipEventQuery reports "not ready" through the sticky error too
}

// oldest deliverable frame that has not been handed out
Item *oldest_picture(VideoSink *s)
{
    for (auto &p : s->items)
        if (!p->discard && p->state != St::Failed) return p.get();
    return nullptr;
}

}  // namespace

int video_sink_create(VideoSink **out, unsigned width, unsigned height)
{
    *out = nullptr;
    auto *s = new (std::nothrow) VideoSink();
    if (!s) return fail(EA1_ENOMEM, "out of memory");
    s->track_w = (width + 7) & ~7u;
    s->track_h = (height + 7) & ~7u;
    const int ndev = ea1_device_count();
    if (ndev <= 0) {
        delete s;
        return fail(EA1_ENODEV, "video sink: no HIP device (there is no CPU fallback)");
    }
    // the reference's settings struct cannot grow (ABI): the device and the helper count come from the environment
    const char *dev = getenv("EASYAV1_AMD_DEVICE");
    s->device = dev ? atoi(dev) : 0;
    if (s->device < 0 || s->device >= ndev) {
        const int bad = s->device;
        delete s;
        return fail(EA1_ENODEV, "video sink: EASYAV1_AMD_DEVICE=%d, %d device(s) present", bad, ndev);
    }
    DeviceGuard g(s->device);
    if (hipStreamCreateWithFlags(&s->recon_st, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s->copy_st, hipStreamNonBlocking) != hipSuccess) {
        if (s->recon_st) (void)hipStreamDestroy(s->recon_st);
        delete s;
        return fail(EA1_EHIP, "hipStreamCreate failed");
    }
    const char *thr = getenv("EASYAV1_AMD_THREADS");
    int nthreads = thr ? atoi(thr) : int(std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
    nthreads = std::min(std::max(nthreads, 1), 32);
    try {
        s->issuer = std::thread(issuer_loop, s);
        for (int k = 0; k < nthreads; k++) s->workers.emplace_back(worker_loop, s);
    } catch (...) {
        video_sink_destroy(s);
        return fail(EA1_ENOMEM, "video sink: could not start its threads");
    }
    *out = s;
    return EA1_OK;
}

void video_sink_flush(VideoSink *s)
{
    if (!s) return;
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    // nothing new starts: queued frames are dropped where they stand, the ones in the helpers' hands are waited for
    for (auto &p : s->items)
        if (p->state == St::Queued) p->state = St::Failed;
    const int saved_rc = s->error_rc;
    s->error_rc = EA1_ESTATE;   // holds the helpers and makes the issuer skip what is prepared
    s->cv_issue.notify_all();
    s->cv_done.wait(lk, [&] { return s->preparing == 0 && !s->issuer_busy && (s->items.empty() || s->next_issue > s->items.back()->ticket); });
    (void)saved_rc;
    lk.unlock();
    (void)hipStreamSynchronize(s->recon_st);
    (void)hipStreamSynchronize(s->copy_st);
    lk.lock();
    for (auto &p : s->items) retire(s, *p, false);
    s->items.clear();
    if (s->shown >= 0) s->free_buffers.push_back(s->shown);
    s->shown = -1;
    s->next_issue = s->next_ticket;
    s->error_rc = EA1_OK;   // a seek starts the stream afresh
    s->error_msg.clear();
    for (auto &sc : s->slot_copy) sc.ev = nullptr;
    lk.unlock();
    if (s->ctx) ea1_recon_flush(s->ctx);
}

void video_sink_destroy(VideoSink *s)
{
    if (!s) return;
    video_sink_flush(s);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv_work.notify_all();
    s->cv_issue.notify_all();
    if (s->issuer.joinable()) s->issuer.join();
    for (std::thread &t : s->workers)
        if (t.joinable()) t.join();
    DeviceGuard g(s->device);
    if (s->ctx) ea1_recon_destroy(s->ctx);
    for (uint8_t *b : s->buffers) ea1_host_free(b);
    for (auto &p : s->pins)
        if (p->p) ea1_host_free(p->p);
    for (hipEvent_t e : s->free_events) (void)hipEventDestroy(e);
    if (s->recon_st) (void)hipStreamDestroy(s->recon_st);
    if (s->copy_st) (void)hipStreamDestroy(s->copy_st);
    delete s;
}

int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool discard, uint64_t *ticket)
{
    if (!s || !payload) return fail(EA1_EINVAL, "video sink: NULL argument");
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    sweep(s);
    // beyond kRing pictures that nobody took the oldest one is given up (the reference's decoded-frame ring drops its oldest entry)
    for (;;) {
        size_t waiting = 0;
        for (auto &p : s->items)
            if (!p->discard && p->state != St::Failed) waiting++;
        if (waiting < kRing) break;
        Item *old = oldest_picture(s);
        s->cv_done.wait(lk, [&] { return (old->state == St::Issued || old->state == St::Failed) && old->ticket < s->next_issue; });
        if (old->state == St::Issued) {
            hipEvent_t ev = old->done;
            lk.unlock();
            (void)hipEventSynchronize(ev);
            lk.lock();
        }
        // the oldest picture is the first deliverable item; everything in front of it is discarded or failed work
        while (!s->items.empty()) {
            Item &it = *s->items.front();
            const bool last = &it == old;
            if (it.state == St::Issued && it.done) (void)hipEventSynchronize(it.done);
            retire(s, it, false);
            s->items.pop_front();
            if (last) break;
        }
    }
    auto it = std::unique_ptr<Item>(new (std::nothrow) Item());
    if (!it) return fail(EA1_ENOMEM, "out of memory");
    it->ticket = s->next_ticket++;
    it->payload = payload;
    it->size = size;
    it->timestamp = timestamp;
    it->discard = discard;
    if (s->items.empty()) s->next_issue = std::min(s->next_issue, it->ticket);
    if (ticket) *ticket = it->ticket;
    s->items.push_back(std::move(it));
    s->cv_work.notify_one();
    return EA1_OK;
}

int video_sink_wait(VideoSink *s, uint64_t ticket)
{
    if (!s) return fail(EA1_EINVAL, "video sink: NULL argument");
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    Item *it = find(s, ticket);
    if (!it) {   // retired already (dropped from the ring, or a discarded frame that has finished)
        if (s->error_rc != EA1_OK && ticket >= s->error_ticket) return fail(s->error_rc, "%s", s->error_msg.c_str());
        return EA1_OK;
    }
    s->cv_done.wait(lk, [&] { return it->state == St::Issued || it->state == St::Failed; });
    if (it->state == St::Failed) {
        const int rc = it->rc != EA1_OK ? it->rc : s->error_rc;
        return fail(rc != EA1_OK ? rc : EA1_ESTATE, "%s", it->msg.empty() ? s->error_msg.c_str() : it->msg.c_str());
    }
    hipEvent_t ev = it->done;
    lk.unlock();
    if (hipEventSynchronize(ev) != hipSuccess) return fail(EA1_EHIP, "video sink: waiting for a picture failed: %s", hipGetErrorString(hipGetLastError()));
    return EA1_OK;
}

bool video_sink_has_frame(VideoSink *s, uint64_t position)
{
    if (!s) return false;
    std::lock_guard<std::mutex> lk(s->mu);
    Item *it = oldest_picture(s);
    if (!it || it->timestamp > position || it->state != St::Issued) return false;
    const bool landed = hipEventQuery(it->done) == hipSuccess;
    (void)hipGetLastError();
    return landed;
}

bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out)
{
    if (!s) return false;
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    sweep(s);
    Item *it = oldest_picture(s);
    if (!it || it->timestamp > position || it->state != St::Issued) return false;
    hipEvent_t ev = it->done;
    lk.unlock();
    (void)hipEventSynchronize(ev);
    lk.lock();
    if (s->shown >= 0) s->free_buffers.push_back(s->shown);
    s->shown = it->buffer;
    const uint8_t *base = s->buffers[size_t(it->buffer)];
    for (int pl = 0; pl < 3; pl++) {
        out->data[pl] = base + s->plane_off[pl];
        out->stride[pl] = s->plane_stride[pl];
    }
    out->width = it->vis_w;
    out->height = it->vis_h;
    out->timestamp = it->timestamp;
    // the item and whatever discarded / failed work lies in front of it leave the queue
    while (!s->items.empty()) {
        Item &f = *s->items.front();
        const bool last = &f == it;
        if (!last && f.state == St::Issued && f.done) (void)hipEventSynchronize(f.done);
        retire(s, f, last);
        s->items.pop_front();
        if (last) break;
    }
    return true;
}

}  // namespace ea1
