// See easyav1_video_sink.h.  Packets are prepared on helper threads; one issuer thread commits them (copies + derivation
// kernels, several frames per launch) and submits the reconstruction: ONE FRAME OF EVERY GOP CHAIN PER LAUNCH -- closed GOPs
// are independent, a key-frame packet starts the next chain in turn, and the shim reads far enough ahead
// in the file for `chains` GOPs to be in flight.  Pictures travel to pinned host memory on a second HIP stream, in device
// layout (one copy per picture), each batch fenced by an event; they are handed out in stream order.
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
constexpr size_t kRing = 11;       // finished pictures whose turn has come and that nobody took: the oldest is dropped beyond (src/easyav1.c:41, :2141-2143)
constexpr int kMaxChains = 8;
constexpr int kMaxCommit = 8;      // prepared frames committed with one launch of the derivation kernels
constexpr int kPrepareAhead = 12;  // frames of one chain that may be prepared ahead of the chain's next submit (bounds the arenas held)

// Queued -> Preparing -> Prepared -> Committed -> Issued; Failed: its hand-over or submit failed; Dead: given up (error, flush)
enum class St { Queued, Preparing, Prepared, Committed, Issued, Failed, Dead };

struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
};

// one submit: a frame of up to `chains` GOP chains, their picture copies, one event behind all of it
struct Batch {
    hipEvent_t done = nullptr;   // behind the picture copies (or behind the reconstruction when no picture of the batch is delivered)
    hipEvent_t mid = nullptr;    // behind the reconstruction: what the copy stream waits for
    int refs = 0;
};

struct Item {
    uint64_t ticket = 0;
    const uint8_t *payload = nullptr;
    size_t size = 0;
    uint64_t timestamp = 0;
    bool discard = false;
    int chain = 0;
    St state = St::Queued;
    PinBlock *pin = nullptr;      // parse storage (pinned), owned by VideoSink::pins
    ea1_ir_frame frame;
    ea1_job *job = nullptr;
    int rc = EA1_OK;
    std::string msg;
    unsigned vis_w = 0, vis_h = 0;
    int buffer = -1;              // picture buffer: index into VideoSink::buffers
    Batch *batch = nullptr;
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
    int device = 0, chains = 1, lookahead = 10;
    ea1_recon *ctx = nullptr;
    hipStream_t recon_st = nullptr, copy_st = nullptr;
    size_t plane_off[3] = {0, 0, 0}, plane_stride[3] = {0, 0, 0}, pic_bytes = 0;

    std::mutex mu;
    std::condition_variable cv_work, cv_issue, cv_done;
    std::deque<std::unique_ptr<Item>> items;   // submission order; tickets are consecutive: items[k]->ticket == items[0]->ticket + k
    uint64_t next_ticket = 0;
    uint64_t gop_count = 0;                    // key-frame packets seen: GOP g runs on chain g % chains
    bool any_submitted = false;
    uint64_t waited_through = 0;               // tickets below this have had their turn (video_sink_wait)
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
    std::vector<Batch *> free_batches;
    int shown = -1;                            // buffer of the picture the caller holds
    Batch *slot_copy[kMaxChains][EA1_NUM_REF_SLOTS] = {};   // the batch whose picture copy last read each reference slot (nullptr: none in flight)
};

bool is_parsed_frame_metadata(const uint8_t *payload, size_t size) { return ea1_ir_packet_probe(payload, size) != 0; }

namespace {

Item *find(VideoSink *s, uint64_t ticket)
{
    if (s->items.empty() || ticket < s->items.front()->ticket) return nullptr;
    const uint64_t k = ticket - s->items.front()->ticket;
    return k < s->items.size() ? s->items[size_t(k)].get() : nullptr;
}

bool settled(const Item &it) { return it.state == St::Issued || it.state == St::Dead; }   // the issuer is done with it

void release_batch(VideoSink *s, Batch *b)
{
    if (!b || --b->refs > 0) return;
    for (auto &row : s->slot_copy)
        for (Batch *&sc : row)
            if (sc == b) sc = nullptr;
    if (b->done) s->free_events.push_back(b->done);
    if (b->mid) s->free_events.push_back(b->mid);
    b->done = b->mid = nullptr;
    s->free_batches.push_back(b);
}

// consumer side, under the lock: gives the item's resources back (its batch's event must have completed or been waited for)
void retire(VideoSink *s, Item &it, bool keep_buffer)
{
    release_batch(s, it.batch);
    it.batch = nullptr;
    if (it.buffer >= 0 && !keep_buffer) s->free_buffers.push_back(it.buffer);
    it.buffer = -1;
    if (it.pin) it.pin->busy = false;
    if (it.job) ea1_job_destroy(it.job);   // never synchronises: the arena returns to the pool behind the job's last submit
    it.job = nullptr;
    it.pin = nullptr;
}

void set_error(VideoSink *s, const Item &it)
{
    if (s->error_rc != EA1_OK) return;
    s->error_rc = it.rc != EA1_OK ? it.rc : EA1_ESTATE;
    s->error_msg = it.msg;
    s->error_ticket = it.ticket;
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
        if (rc == EA1_OK && d.dst_slot >= EA1_NUM_REF_SLOTS) rc = fail(EA1_EINVAL, "parsed frame: destination slot %u", d.dst_slot);
        it.vis_w = unsigned(vis_w);
        it.vis_h = unsigned(vis_h);
    }
    if (rc == EA1_OK) {
        std::unique_lock<std::mutex> lk(s->mu);
        if (!s->ctx) {
            // frames may be smaller than the track's size (resolution switches, scaled references): the context holds the largest
            s->width = std::max(s->track_w, f.desc.width);
            s->height = std::max(s->track_h, f.desc.height);
            rc = ea1_recon_create_batch(&s->ctx, s->width, s->height, s->chains, s->device);
            if (rc == EA1_OK) rc = ea1_recon_picture_layout(s->ctx, s->plane_off, s->plane_stride, &s->pic_bytes);
        }
    }
    if (rc == EA1_OK) rc = ea1_job_prepare(s->ctx, &f, &it.job);
    it.rc = rc;
    if (rc != EA1_OK) it.msg = ea1_last_error();
}

// under the lock: the queued frame to prepare next -- the one closest to its chain's next submit (with several chains the
// heads of all of them come first: a batch wants one frame of each), not further ahead than kPrepareAhead
Item *pick_work(VideoSink *s)
{
    int ahead[kMaxChains] = {};
    Item *best = nullptr;
    int best_rank = 1 << 30;
    for (auto &p : s->items) {
        Item &it = *p;
        if (settled(it)) continue;
        if (s->error_rc != EA1_OK && it.ticket >= s->error_ticket) continue;   // given up: the issuer marks it
        const int rank = ahead[it.chain]++;
        if (it.state == St::Queued && rank < kPrepareAhead && rank < best_rank) {
            best = &it;
            best_rank = rank;
        }
    }
    return best;
}

void worker_loop(VideoSink *s)
{
    (void)hipSetDevice(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        if (s->stop) return;
        Item *it = pick_work(s);
        if (!it) {
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

hipEvent_t take_event(VideoSink *s)
{
    if (s->free_events.empty()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        return e;
    }
    hipEvent_t e = s->free_events.back();
    s->free_events.pop_back();
    return e;
}

void issuer_loop(VideoSink *s)
{
    (void)hipSetDevice(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        if (s->stop) return;
        // ---- errors: a failed frame breaks the stream from its ticket on (later frames would predict from pictures that do not
        //      exist; frames of other chains that lie BEFORE it in the stream still come out) ----
        for (auto &p : s->items)
            if (p->state == St::Failed) set_error(s, *p);
        if (s->error_rc != EA1_OK) {
            bool changed = false;
            for (auto &p : s->items) {
                Item &it = *p;
                if (it.ticket < s->error_ticket) continue;
                if (it.state == St::Queued || it.state == St::Prepared || it.state == St::Committed || it.state == St::Failed) {
                    if (it.rc == EA1_OK) {
                        it.rc = s->error_rc;
                        it.msg = s->error_msg;
                    }
                    it.state = St::Dead;
                    changed = true;
                }
            }
            if (changed) s->cv_done.notify_all();
        }
        // ---- commit: the copies and the derivation kernels of whatever the helpers have finished, up to kMaxCommit frames a launch ----
        Item *run[kMaxCommit];
        ea1_job *jobs[kMaxCommit];
        int n = 0;
        for (auto &p : s->items)
            if (p->state == St::Prepared && n < kMaxCommit) {
                run[n] = p.get();
                jobs[n++] = p->job;
            }
        if (n) {
            s->issuer_busy = true;
            lk.unlock();
            const int rc = ea1_job_commit(s->ctx, jobs, n);
            const std::string msg = rc != EA1_OK ? ea1_last_error() : "";
            lk.lock();
            s->issuer_busy = false;
            for (int k = 0; k < n; k++) {
                run[k]->state = rc == EA1_OK ? St::Committed : St::Failed;
                if (rc != EA1_OK) {
                    run[k]->rc = rc;
                    run[k]->msg = msg;
                }
            }
            s->cv_done.notify_all();
            continue;
        }
        // ---- submit: the next frame of every chain that has one committed -- once the OLDEST unsubmitted frame of the stream is
        //      among them (it is what the caller waits for next; in the steady state every chain's head is ready by then) ----
        Item *head[kMaxChains] = {};
        bool seen[kMaxChains] = {};
        Item *oldest = nullptr;
        for (auto &p : s->items) {
            Item &it = *p;
            if (settled(it) || seen[it.chain]) continue;
            seen[it.chain] = true;
            if (!oldest) oldest = &it;
            // one submit carries frames of ONE size (ea1_recon_submit_batch): a chain whose next frame has another size (a stream
            // that switches resolution) waits for its own turn
            if (it.state == St::Committed && oldest->state == St::Committed && it.frame.desc.width == oldest->frame.desc.width &&
                it.frame.desc.height == oldest->frame.desc.height)
                head[it.chain] = &it;
        }
        if (!oldest || oldest->state != St::Committed) {
            s->cv_issue.wait(lk);
            continue;
        }
        int rc = EA1_OK;
        std::string msg;
        Batch *b = nullptr;
        if (s->free_batches.empty()) {
            b = new (std::nothrow) Batch();
        } else {
            b = s->free_batches.back();
            s->free_batches.pop_back();
        }
        if (b) {
            b->done = take_event(s);
            b->mid = take_event(s);
            b->refs = 1;   // the issuer's own reference until the batch is out
        }
        if (!b || !b->done || !b->mid) {
            rc = EA1_EHIP;
            msg = "video sink: no event for a submit";
        }
        bool any_picture = false;
        for (int c = 0; c < s->chains && rc == EA1_OK; c++) {
            Item *it = head[c];
            if (!it || it->discard) continue;
            any_picture = true;
            if (s->free_buffers.empty()) {
                void *mem = nullptr;
                lk.unlock();   // pinning 12 MB takes a millisecond: not under the lock
                const int arc = ea1_host_alloc(s->pic_bytes, &mem);
                if (arc != EA1_OK) msg = ea1_last_error();
                lk.lock();
                if (arc != EA1_OK) {
                    rc = arc;
                    break;
                }
                s->buffers.push_back(static_cast<uint8_t *>(mem));
                s->free_buffers.push_back(int(s->buffers.size()) - 1);
            }
            it->buffer = s->free_buffers.back();
            s->free_buffers.pop_back();
        }
        // what the submit must wait for: the copies that still read the slots it is about to overwrite
        hipEvent_t wait_for[kMaxChains] = {};
        ea1_job *bjobs[kMaxChains] = {};
        int slots[kMaxChains];
        uint8_t *dst[kMaxChains] = {};
        for (int c = 0; c < s->chains; c++) {
            slots[c] = -1;
            Item *it = head[c];
            if (!it) continue;
            bjobs[c] = it->job;
            const int slot = it->frame.desc.dst_slot;
            if (Batch *old = s->slot_copy[c][slot]) wait_for[c] = old->done;
            if (!it->discard && it->buffer >= 0) {
                slots[c] = slot;
                dst[c] = s->buffers[size_t(it->buffer)];
            }
        }
        s->issuer_busy = true;
        lk.unlock();
        for (int c = 0; c < s->chains && rc == EA1_OK; c++)
            if (wait_for[c] && hipStreamWaitEvent(s->recon_st, wait_for[c], 0) != hipSuccess) {
                rc = EA1_EHIP;
                msg = "hipStreamWaitEvent failed";
            }
        if (rc == EA1_OK) {
            rc = ea1_recon_submit_batch(s->ctx, bjobs, EA1_STAGE_ALL, s->recon_st);
            if (rc != EA1_OK) msg = ea1_last_error();
        }
        if (rc == EA1_OK) {
            if (any_picture) {
                // reconstruction -> (event) -> copy stream: the copies of this batch run beside the reconstruction of the next
                if (hipEventRecord(b->mid, s->recon_st) != hipSuccess || hipStreamWaitEvent(s->copy_st, b->mid, 0) != hipSuccess) rc = EA1_EHIP;
                if (rc == EA1_OK) rc = ea1_recon_download_batch_async(s->ctx, slots, dst, s->copy_st);
                if (rc == EA1_OK && hipEventRecord(b->done, s->copy_st) != hipSuccess) rc = EA1_EHIP;
            } else if (hipEventRecord(b->done, s->recon_st) != hipSuccess) {
                rc = EA1_EHIP;
            }
            if (rc != EA1_OK && msg.empty()) msg = rc == EA1_EHIP ? "event plumbing of the picture copy failed" : ea1_last_error();
        }
        if (rc != EA1_OK) {   // whatever did get out must not outlive the resources the failed items give back
            (void)hipStreamSynchronize(s->recon_st);
            (void)hipStreamSynchronize(s->copy_st);
        }
        lk.lock();
        s->issuer_busy = false;
        for (int c = 0; c < s->chains; c++) {
            Item *it = head[c];
            if (!it) continue;
            if (rc == EA1_OK) {
                it->state = St::Issued;
                it->batch = b;
                b->refs++;
                if (slots[c] >= 0) s->slot_copy[c][slots[c]] = b;
            } else {
                it->state = St::Failed;
                it->rc = rc;
                it->msg = msg;
                if (it->buffer >= 0) s->free_buffers.push_back(it->buffer);
                it->buffer = -1;
            }
        }
        if (b) release_batch(s, b);   // the issuer's own reference
        s->cv_done.notify_all();
        s->cv_work.notify_all();      // chains moved on: frames further ahead may be prepared now
    }
}

bool batch_landed(const Item &it)
{
    const bool ok = it.batch && hipEventQuery(it.batch->done) == hipSuccess;
    (void)hipGetLastError();   // hipEventQuery reports "not ready" through the sticky error too
    return ok;
}

// consumer side, under the lock: the front of the queue that nobody will ask for any more
void sweep(VideoSink *s)
{
    while (!s->items.empty()) {
        Item &it = *s->items.front();
        const bool dead_discard = it.discard && it.state == St::Issued && batch_landed(it);
        if (!dead_discard && it.state != St::Dead) break;
        retire(s, it, false);
        s->items.pop_front();
    }
}

// oldest deliverable frame that has not been handed out
Item *oldest_picture(VideoSink *s)
{
    for (auto &p : s->items)
        if (!p->discard && p->state != St::Dead && p->state != St::Failed) return p.get();
    return nullptr;
}

// consumer side: takes the queue's front up to and including `last` out (everything in front of a picture is discarded or dead work)
void pop_through(VideoSink *s, Item *last, bool keep_last_buffer)
{
    while (!s->items.empty()) {
        Item &f = *s->items.front();
        const bool is_last = &f == last;
        if (f.state == St::Issued && f.batch && !(is_last && keep_last_buffer)) (void)hipEventSynchronize(f.batch->done);
        retire(s, f, is_last && keep_last_buffer);
        s->items.pop_front();
        if (is_last) break;
    }
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
    // the reference's settings struct cannot grow (ABI): the device, the helper count and the GOP look-ahead come from the environment
    const char *dev = getenv("EASYAV1_AMD_DEVICE");
    s->device = dev ? atoi(dev) : 0;
    if (s->device < 0 || s->device >= ndev) {
        const int bad = s->device;
        delete s;
        return fail(EA1_ENODEV, "video sink: EASYAV1_AMD_DEVICE=%d, %d device(s) present", bad, ndev);
    }
    // GOP chains in flight: a launch carries one frame of each (a single 1080p frame is a few microseconds of HBM traffic: the
    // launches dominate).  What it costs: the pictures of the later GOPs wait in pinned host memory until their turn --
    // about (chains - 1) x GOP / 2 of them.  Default: 4 up to 1080p, 2 above; EASYAV1_AMD_GOPS=1 switches it off.
    const char *gops = getenv("EASYAV1_AMD_GOPS");
    s->chains = gops ? atoi(gops) : (uint64_t(s->track_w) * s->track_h <= 1920ull * 1088 ? 4 : 2);
    s->chains = std::min(std::max(s->chains, 1), kMaxChains);
    const char *la = getenv("EASYAV1_AMD_LOOKAHEAD");
    s->lookahead = la ? atoi(la) : (s->chains > 1 ? 64 * s->chains : 10);   // packets the shim reads ahead of the position (the reference: 10)
    s->lookahead = std::min(std::max(s->lookahead, 10), 4096);
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

int video_sink_lookahead(const VideoSink *s) { return s ? s->lookahead : 10; }

void video_sink_flush(VideoSink *s)
{
    if (!s) return;
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    // nothing new starts: the helpers hold, the issuer gives up whatever it has not submitted
    s->error_rc = EA1_ESTATE;
    s->error_ticket = 0;
    s->cv_issue.notify_all();
    s->cv_done.wait(lk, [&] {
        if (s->preparing || s->issuer_busy) return false;
        for (auto &p : s->items)
            if (!settled(*p)) {
                s->cv_issue.notify_all();
                return false;
            }
        return true;
    });
    lk.unlock();
    (void)hipStreamSynchronize(s->recon_st);
    (void)hipStreamSynchronize(s->copy_st);
    lk.lock();
    for (auto &p : s->items) retire(s, *p, false);
    s->items.clear();
    if (s->shown >= 0) s->free_buffers.push_back(s->shown);
    s->shown = -1;
    s->waited_through = s->next_ticket;
    s->any_submitted = false;   // the next packet starts a GOP whatever its flag says
    s->error_rc = EA1_OK;       // a seek starts the stream afresh
    s->error_msg.clear();
    for (auto &row : s->slot_copy)
        for (Batch *&sc : row) sc = nullptr;
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
    for (Batch *b : s->free_batches) delete b;
    if (s->recon_st) (void)hipStreamDestroy(s->recon_st);
    if (s->copy_st) (void)hipStreamDestroy(s->copy_st);
    delete s;
}

int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool keyframe, bool discard, uint64_t *ticket)
{
    if (!s || !payload) return fail(EA1_EINVAL, "video sink: NULL argument");
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    sweep(s);
    // beyond kRing pictures whose turn has come and that nobody took, the oldest one is given up (the reference's decoded-frame
    // ring drops its oldest entry); pictures of packets worked on ahead of their turn do not count
    for (;;) {
        size_t waiting = 0;
        for (auto &p : s->items)
            if (!p->discard && p->state != St::Dead && p->state != St::Failed && p->ticket < s->waited_through) waiting++;
        if (waiting < kRing) break;
        Item *old = oldest_picture(s);
        s->cv_done.wait(lk, [&] { return settled(*old); });
        pop_through(s, old, false);
    }
    auto it = std::unique_ptr<Item>(new (std::nothrow) Item());
    if (!it) return fail(EA1_ENOMEM, "out of memory");
    it->ticket = s->next_ticket++;
    it->payload = payload;
    it->size = size;
    it->timestamp = timestamp;
    it->discard = discard;
    if (keyframe && s->any_submitted) s->gop_count++;   // a key frame starts a GOP: the next chain in turn takes it
    s->any_submitted = true;
    it->chain = int(s->gop_count % uint64_t(s->chains));
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
    s->waited_through = std::max(s->waited_through, ticket + 1);
    Item *it = find(s, ticket);
    if (!it) {   // retired already (dropped from the ring, or a discarded frame that has finished)
        if (s->error_rc != EA1_OK && ticket >= s->error_ticket) return fail(s->error_rc, "%s", s->error_msg.c_str());
        return EA1_OK;
    }
    s->cv_done.wait(lk, [&] { return settled(*it); });
    if (it->state == St::Dead) {
        const int rc = it->rc != EA1_OK ? it->rc : (s->error_rc != EA1_OK ? s->error_rc : EA1_ESTATE);
        return fail(rc, "%s", it->msg.empty() ? s->error_msg.c_str() : it->msg.c_str());
    }
    hipEvent_t ev = it->batch->done;
    lk.unlock();
    if (hipEventSynchronize(ev) != hipSuccess) return fail(EA1_EHIP, "video sink: waiting for a picture failed: %s", hipGetErrorString(hipGetLastError()));
    return EA1_OK;
}

bool video_sink_has_frame(VideoSink *s, uint64_t position)
{
    if (!s) return false;
    std::lock_guard<std::mutex> lk(s->mu);
    Item *it = oldest_picture(s);
    return it && it->timestamp <= position && it->state == St::Issued && batch_landed(*it);
}

bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out)
{
    if (!s) return false;
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    sweep(s);
    Item *it = oldest_picture(s);
    if (!it || it->timestamp > position || it->state != St::Issued) return false;
    hipEvent_t ev = it->batch->done;
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
    pop_through(s, it, true);
    return true;
}

}  // namespace ea1
