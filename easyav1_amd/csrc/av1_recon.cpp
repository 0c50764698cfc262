// Host side of the AV1 reconstruction path: decoder context (reference slots, work pictures),
// hand-over of a parsed frame (upload + schedule derivation) and the per-frame launch sequence.
// This is the GPU counterpart of dav1d's frame driver (src/recon_tmpl.c + lf_mask.c +
// *_apply_tmpl.c, /root/reference/cmake/dav1d.cmake:1014,:980,:1002,:1009,:1012) behind the calls
// easyav1 makes at /root/reference/src/easyav1.c:1405 (open), :2390-2403 (send/get), :3059 (flush).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "av1_dev.h"
#include "film_grain.h"
#include "ea1_common.h"
#include "ea1_av1_tables.inc"
#include "easyav1_amd_ir_packet.h"

using ea1::DevPic;
using ea1::FrameView;
using ea1::JobStats;
using ea1::TxItem;

namespace {

// HIP-event timing of every kernel launch of a submit (bench.py's roofline figure)
struct EventProf final : ea1::Prof {
    struct Rec {
        const char *name;
        uint64_t bytes;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t take()
    {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(const char *kernel, uint64_t bytes, hipStream_t st) override
    {
        Rec r{kernel, bytes, take(), take()};
        (void)hipEventRecord(r.a, st);
        recs.push_back(r);
    }
    void end(hipStream_t st) override { (void)hipEventRecord(recs.back().b, st); }
    ~EventProf() override
    {
        for (auto &r : recs) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

}  // namespace

struct Chain {
    DevPic slots[EA1_NUM_REF_SLOTS];
    DevPic work;                 // reconstruction + in-place deblocking
    DevPic cdef;                 // CDEF output
    DevPic display;              // film grain output (easyav1_amd_grain.h)
    bool display_valid = false;
    int16_t *resid[3] = {nullptr, nullptr, nullptr};
    uint8_t *arena = nullptr;    // one allocation for all pictures + residual planes of the chain
    uint8_t *up_arena = nullptr; // super-resolution: the two upscaled pictures restoration reads (made with the first such frame)
    DevPic up[2];
    DevPic *d_slots = nullptr;   // device copy of slots[] for the inter-prediction kernel
    uint8_t *d_dir_var = nullptr;
    bool valid[EA1_NUM_REF_SLOTS] = {false};
    // size of the picture each slot holds: coded (multiples of 8) and visible; a context decodes frames of any size up
    // to its own, and a frame may predict from a reference of another size (scaled references)
    int slot_w[EA1_NUM_REF_SLOTS] = {0}, slot_h[EA1_NUM_REF_SLOTS] = {0}, slot_vw[EA1_NUM_REF_SLOTS] = {0}, slot_vh[EA1_NUM_REF_SLOTS] = {0};
    uint8_t slot_pad[EA1_NUM_REF_SLOTS] = {0};   // ea1_frame_desc.pad of the frame a submit stored there (film grain: crop counts from the true coded size)
};

// Memory of parsed frames in flight.  A 1080p frame is handed over every ~50 us at the rates the kernels reach, so
// nothing on that path may allocate, free or synchronise: device arenas and pinned staging buffers are recycled,
// the upload runs on the pool's own stream, and memory whose last use may still be in flight waits in `held`
// behind an event instead of behind a hipDeviceSynchronize.  Shared by a context and all of its jobs (a job may
// outlive its context); thread safe (job creation runs on several host threads).
// hipEventQuery reports "not ready" through the error channel: keep that out of the thread's last-error slot, which
// the launch checks read
static bool event_done(hipEvent_t e)
{
    if (hipEventQuery(e) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

struct SubmitMark;
struct JobPool {
    struct Block {
        uint8_t *p;
        size_t cap;
    };
    struct Held {
        Block b;
        bool pinned;
        hipEvent_t ev;                       // owned event, or
        std::shared_ptr<SubmitMark> mark;    // the submit whose completion frees the block
    };
    std::recursive_mutex mu;   // recursive: dropping the last reference to a SubmitMark under the lock returns its event to the pool
    int device = 0;
    // Two streams of the pool's own: ALL host-to-device copies go down `copy` back to back (measured on this box,
    // profiles/r02_b_host_link_probe.txt: one stream of 8 MB copies reaches 54 GB/s, two streams share 46 GB/s), the
    // derivation kernels of a job run on `derive` behind an event, so a kernel never holds the DMA queue up.
    // (the kernels' stream is picked per hand-over thread out of four: 16 threads on one stream queue at its lock)
    static constexpr int kDeriveStreams = 4;
    hipStream_t copy = nullptr, copy2 = nullptr, derive[kDeriveStreams] = {nullptr, nullptr, nullptr, nullptr};
    std::atomic<unsigned> next_stream{0};
    hipStream_t derive_stream()
    {
        static thread_local int mine = -1;
        if (mine < 0) mine = int(next_stream.fetch_add(1) % kDeriveStreams);
        return derive[mine];
    }
    std::vector<Block> free_dev, free_pin;
    std::deque<Held> held;
    std::vector<hipEvent_t> events;

    ~JobPool()
    {
        ea1::DeviceGuard g(device);
        (void)hipDeviceSynchronize();
        for (Held &h : held) {
            if (h.ev) (void)hipEventDestroy(h.ev);
            h.mark.reset();
            if (h.pinned) (void)hipHostFree(h.b.p);
            else (void)hipFree(h.b.p);
        }
        for (Block &b : free_dev) (void)hipFree(b.p);
        for (Block &b : free_pin) (void)hipHostFree(b.p);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (copy) (void)hipStreamDestroy(copy);
        if (copy2) (void)hipStreamDestroy(copy2);
        for (hipStream_t st : derive)
            if (st) (void)hipStreamDestroy(st);
    }
    hipEvent_t event()   // call with `mu` held
    {
        if (!events.empty()) {
            hipEvent_t e = events.back();
            events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return e;
    }
    void reclaim();      // call with `mu` held: memory whose event has completed becomes free
    bool take(bool pinned, size_t bytes, Block *out)
    {
        std::lock_guard<std::recursive_mutex> g(mu);
        reclaim();
        std::vector<Block> &fl = pinned ? free_pin : free_dev;
        size_t best = fl.size();
        for (size_t i = 0; i < fl.size(); i++)
            if (fl[i].cap >= bytes && fl[i].cap <= 2 * bytes + (size_t(1) << 20) && (best == fl.size() || fl[i].cap < fl[best].cap)) best = i;
        if (best != fl.size()) {
            *out = fl[best];
            fl[best] = fl.back();
            fl.pop_back();
            return true;
        }
        const size_t cap = (bytes + (size_t(1) << 18) - 1) & ~((size_t(1) << 18) - 1);
        void *p = nullptr;
        if ((pinned ? hipHostMalloc(&p, cap, hipHostMallocDefault) : hipMalloc(&p, cap)) != hipSuccess) {
            // recycle what can be recycled before giving up: drop the free lists and retry once
            for (Block &b : fl) (void)(pinned ? hipHostFree(b.p) : hipFree(b.p));
            fl.clear();
            if ((pinned ? hipHostMalloc(&p, cap, hipHostMallocDefault) : hipMalloc(&p, cap)) != hipSuccess) return false;
        }
        *out = Block{static_cast<uint8_t *>(p), cap};
        return true;
    }
    // `after`: stream whose work so far is the last user of the block (nullptr: free at once)
    void give(Block b, bool pinned, hipStream_t after, bool has_user)
    {
        std::lock_guard<std::recursive_mutex> g(mu);
        if (!has_user) {
            (pinned ? free_pin : free_dev).push_back(b);
            return;
        }
        hipEvent_t e = event();
        (void)hipEventRecord(e, after);
        held.push_back(Held{b, pinned, e, nullptr});
    }
    void give_behind(Block b, bool pinned, hipEvent_t ev)   // takes ownership of a recorded event
    {
        std::lock_guard<std::recursive_mutex> g(mu);
        held.push_back(Held{b, pinned, ev, nullptr});
    }
    void give_behind(Block b, bool pinned, std::shared_ptr<SubmitMark> mark)
    {
        std::lock_guard<std::recursive_mutex> g(mu);
        held.push_back(Held{b, pinned, nullptr, std::move(mark)});
    }
};

// host threads of ea1_job_create_batch: started on first use, parked on a condition variable in between
struct Workers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable wake, idle;
    std::function<void(int)> fn;
    std::vector<int> done;   // indices whose fn() has returned, in completion order (run_with_issuer)
    std::condition_variable progress;
    int next = 0, total = 0, running = 0, active = 0;
    uint64_t generation = 0;
    bool stop = false;
    void loop(int id)
    {
        (void)pthread_setname_np(pthread_self(), "ea1-prepare");
        std::unique_lock<std::mutex> lk(mu);
        uint64_t seen = 0;
        for (;;) {
            wake.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            if (id >= active) {   // a call that asked for fewer threads than there are
                seen = generation;
                continue;
            }
            while (next < total) {
                const int i = next++;
                lk.unlock();
                fn(i);
                lk.lock();
                done.push_back(i);
                progress.notify_one();
            }
            seen = generation;
            if (--running == 0) idle.notify_all();
        }
    }
    // fn(0) .. fn(n - 1) on nthreads threads (nthreads - 1 helpers and the caller) while the CALLER ALONE runs
    // on_done(i) for every finished i, in completion order (the hand-over: validation and staging on all threads,
    // only the caller talks to the HIP runtime, so the copies of finished frames overlap the host work on the
    // others and no two threads queue at the runtime's lock).
    // One thread: fn(i); on_done(i) in turn.
    std::mutex call_mu;   // one batch at a time per context (two caller threads may both hand batches over)
    void run_with_issuer(int n, int nthreads, std::function<void(int)> f, const std::function<void(int)> &on_done)
    {
        if (n <= 0) return;
        std::lock_guard<std::mutex> one_call(call_mu);
        nthreads = std::max(1, std::min(nthreads, n + 1));
        if (nthreads == 1) {
            for (int i = 0; i < n; i++) {
                f(i);
                on_done(i);
            }
            return;
        }
        std::unique_lock<std::mutex> lk(mu);
        while (int(threads.size()) < nthreads - 1) {
            const int id = int(threads.size());
            threads.emplace_back([this, id] { loop(id); });
        }
        fn = std::move(f);
        done.clear();
        next = 0;
        total = n;
        running = active = nthreads - 1;
        generation++;
        wake.notify_all();
        // the caller issues what is finished first; with nothing to issue it takes a frame itself (n frames on
        // nthreads workers: 32 frames on 16 threads are two rounds, on 15 helpers three)
        size_t taken = 0;
        while (taken < size_t(n)) {
            if (taken < done.size()) {
                const int i = done[taken++];
                lk.unlock();
                on_done(i);
                lk.lock();
            } else if (next < total) {
                const int i = next++;
                lk.unlock();
                fn(i);
                lk.lock();
                done.push_back(i);
            } else {
                progress.wait(lk, [&] { return taken < done.size(); });
            }
        }
        idle.wait(lk, [&] { return running == 0; });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        wake.notify_all();
        for (std::thread &t : threads) t.join();
    }
};

struct ea1_recon {
    EventProf *prof = nullptr;
    std::shared_ptr<JobPool> pool;
    std::unique_ptr<Workers> workers;
    uint8_t *d_wedge = nullptr;           // wedge mask table (wedge.cpp)
    int8_t *d_warp_filter = nullptr;      // warp filter table, 193 x 8 (ea1_recon_set_warp_filter)
    bool warp_filter_set = false;
    int16_t *d_upscale = nullptr;         // Upscale_Filter, 64 x 8 (ea1_recon_set_upscale_filter): the library does not own it
    bool upscale_set = false;
    std::vector<int16_t> gaussian;        // Gaussian sequence of the film grain synthesis (host: the templates are made there)
    uint8_t *d_grain = nullptr;           // film grain: GrainView[nchains] + one blob per chain
    std::vector<uint8_t> h_grain;         // host staging of the same
    size_t grain_blob = 0;
    uint32_t *d_wave_counters = nullptr;  // scratch counters of the intra wavefront (kept for experiments; the grid barrier that used word 0 is gone: av1_intra.hip)
    int device = 0;
    uint32_t width = 0, height = 0;
    int layout = EA1_LAYOUT_420, ssx = 1, ssy = 1;   // chroma layout of every picture of the context
    size_t pic_bytes = 0;
    std::vector<Chain> chains;
    std::mutex commit_mu;                 // ea1_job_commit: one caller at a time issues copies + derivation on the pool's streams
};

// the submit a job was last part of: one event per submit call, shared by its jobs
struct SubmitMark {
    std::weak_ptr<JobPool> pool;   // weak: marks are parked inside the pool's own `held` list
    hipEvent_t ev = nullptr;
    ~SubmitMark()
    {
        if (!ev) return;
        if (auto p = pool.lock()) {
            std::lock_guard<std::recursive_mutex> g(p->mu);
            p->events.push_back(ev);
        } else {
            (void)hipEventDestroy(ev);
        }
    }
};

void JobPool::reclaim()
{
    // entries complete roughly in order; stop at the first one that has not (a later one is picked up next time)
    while (!held.empty()) {
        Held &f = held.front();
        if (!event_done(f.mark ? f.mark->ev : f.ev)) break;
        Held h = std::move(f);
        held.pop_front();
        if (h.ev) events.push_back(h.ev);
        (h.pinned ? free_pin : free_dev).push_back(h.b);
        h.mark.reset();
    }
}

struct ea1_job {
    int device = 0;
    ea1::FrameView view;         // device pointers of the frame data; chain buffers filled per chain
    ea1::JobStats stats;
    std::shared_ptr<JobPool> pool;
    JobPool::Block arena{nullptr, 0};   // everything of the job on the device: uploaded records + derived maps + its FrameView
    // recorded on the pool's derive stream behind the copies and the derivation kernels; shared by the jobs of one
    // hand-over batch (one event, one launch of the derivation kernels for all of them)
    std::shared_ptr<SubmitMark> ready;
    bool ready_seen = false;            // ... and known to have completed
    // between job_prepare (host work, any thread) and job_issue (the HIP calls, one thread): what is to be copied
    struct Copy {
        void *dst;
        const void *src;
        size_t bytes;
        bool pinned;   // the source is pinned memory of this library (staging buffer, ea1_host_alloc): a kernel can read it
    };
    struct Pending {
        std::vector<Copy> copies;
        JobPool::Block pin{nullptr, 0};
        const ea1::DeriveArgs *d_args = nullptr;
        uint32_t nb = 0, ntx = 0, max_items = 0;
        int units = 0;
    };
    std::unique_ptr<Pending> pending;
    std::shared_ptr<SubmitMark> last_use;
    ea1::FrameView *d_view_slot = nullptr;   // place of the first bound FrameView inside the arena
    // device copies of `view` specialised for (context, chain), created on first use
    struct Bound {
        const ea1_recon *ctx;
        int chain;
        ea1::FrameView *d_view;
        bool own_alloc;
        uint32_t scaled_mask;   // views with scaled references are made anew for every submit (the references' sizes may change)
    };
    std::vector<Bound> bound;
    double refs_per_pixel = 0;  // r-bar of SURVEY.md 8(d)
    uint64_t coef_count = 0;    // C
    uint64_t block_bytes = 0;   // M
    size_t intra_count = 0;
    std::vector<uint32_t> level_start;   // host copy (JobStats::level_start points here)
};

namespace {

int align_up(int v, int a) { return (v + a - 1) / a * a; }

// interpolation filter set of a translated prediction: bilinear is set 5, a 4-sample side of the
// predicted block takes the 4-tap sets (dav1d mc_tmpl.c; AV1 spec 7.11.3.4)
int mc_filter_set(int kind, int dim)
{
    if (kind == EA1_FILTER_BILINEAR) return 5;
    return dim <= 4 ? (kind == EA1_FILTER_SMOOTH ? 4 : 3) : kind;
}

// ---- warped motion: shear parameters of an affine matrix (AV1 spec 7.11.3.6 setup shear process,
// 7.11.3.7 resolve divisor process; dav1d src/warpmv.c dav1d_get_shear_params) ----
struct WarpSetup {
    bool valid;
    int32_t mat[6];
    int sh[4];   // alpha, beta, gamma, delta (multiples of 64)
};

int64_t round2_signed(int64_t v, int n)
{
    const int64_t half = int64_t(1) << (n - 1);
    return v < 0 ? -((-v + half) >> n) : (v + half) >> n;
}

bool shear_params(const int32_t m[6], int sh[4])
{
    if (m[2] <= 0) return false;
    auto clamp16 = [](int64_t v) { return int(std::min<int64_t>(std::max<int64_t>(v, -32768), 32767)); };
    // 1 / m[2] as Div_Lut[f] * 2^-shift, Div_Lut[k] = Round(2^22 / (256 + k))
    int n = 31;
    while (!((uint32_t(m[2]) >> n) & 1)) n--;
    const int64_t e = int64_t(m[2]) - (int64_t(1) << n);
    const int f = int(n > 8 ? (e + (int64_t(1) << (n - 9))) >> (n - 8) : e << (8 - n));
    const int64_t factor = ((int64_t(1) << 22) + (256 + f) / 2) / (256 + f);
    const int shift = n + 14;
    const int raw[4] = {clamp16(int64_t(m[2]) - 65536), clamp16(m[3]),
                        clamp16(round2_signed(int64_t(m[4]) * 65536 * factor, shift)),
                        clamp16(int64_t(m[5]) - round2_signed(int64_t(m[3]) * m[4] * factor, shift) - 65536)};
    for (int k = 0; k < 4; k++) sh[k] = int(round2_signed(raw[k], 6)) * 64;
    return 4 * std::abs(sh[0]) + 7 * std::abs(sh[1]) < 65536 && 4 * std::abs(sh[2]) + 4 * std::abs(sh[3]) < 65536;
}

// matrices of a block's warped predictions; false when none of them passes the shear test (the
// block is then an ordinary translated block)
bool warp_setup(const ea1_block &b, const uint8_t *aux, WarpSetup out[2])
{
    const uint8_t *p = aux + b.aux_off;
    bool any = false;
    for (int k = 0; k < 2; k++) {
        out[k].valid = false;
        if (!(b.warp_refs & (1 << k))) continue;
        memcpy(out[k].mat, p, 24);
        p += 24;
        out[k].valid = shear_params(out[k].mat, out[k].sh);
        any |= out[k].valid;
    }
    return any;
}

void layout_pic(DevPic &p, uint8_t *base, int w, int h, int ssx, int ssy, size_t *bytes)
{
    p.w = w;
    p.h = h;
    p.ssx = uint8_t(ssx);
    p.ssy = uint8_t(ssy);
    size_t off = 0;
    for (int pl = 0; pl < 3; pl++) {
        const int pw = pl ? w >> ssx : w, ph = pl ? h >> ssy : h;
        p.stride[pl] = align_up(pw, 128);
        p.p[pl] = base ? base + off : nullptr;
        // 8 guard rows keep the widest vector accesses of the filters inside the allocation
        off += size_t(p.stride[pl]) * (ph + 8);
        off = (off + 255) & ~size_t(255);
    }
    if (bytes) *bytes = off;
}

// All per-frame device data lives in ONE arena: the small records and lists are laid out (256-byte aligned) in a
// pinned staging buffer and travel in one asynchronous copy; the coefficients -- most of the bytes -- are copied
// straight from the caller's array (no second pass over them on the host); regions that only the device writes
// (the derived maps, the FrameView) follow behind.  Arena and staging come from the pool, nothing is allocated
// per frame once the pool is warm.
// pinned host memory handed out by ea1_host_alloc: [begin, end) ranges (a hand-over whose coefficient array lies in
// one is pulled by a kernel instead of copied call by call)
struct PinnedRegistry {
    std::mutex mu;
    std::map<uintptr_t, uintptr_t> ranges;   // begin -> end
    void add(void *p, size_t n)
    {
        std::lock_guard<std::mutex> g(mu);
        ranges[reinterpret_cast<uintptr_t>(p)] = reinterpret_cast<uintptr_t>(p) + n;
    }
    void remove(void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        ranges.erase(reinterpret_cast<uintptr_t>(p));
    }
    bool covers(const void *p, size_t n)
    {
        std::lock_guard<std::mutex> g(mu);
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        auto it = ranges.upper_bound(a);
        if (it == ranges.begin()) return false;
        --it;
        return a >= it->first && a + n <= it->second;
    }
};
PinnedRegistry &pinned_registry()
{
    static PinnedRegistry reg;
    return reg;
}
bool pinned_by_library(const void *p, size_t n) { return pinned_registry().covers(p, n); }

struct Packer {
    struct Part {
        const void *src;
        size_t bytes, reserve;   // reserve >= bytes: a staged part's tail up to the 256-byte boundary is zero
        const void **out;
        size_t off;
        bool direct;             // copied from `src` directly, not through the staging buffer
        bool device_tail;        // the reservation beyond `bytes` is written on the device: nothing is uploaded for it
    };
    std::vector<Part> parts;
    size_t total = 0;            // bytes of the uploaded part of the arena
    size_t device_only = 0;      // bytes behind it that no copy touches
    void reset()
    {
        parts.clear();
        total = device_only = 0;
    }
    template <class T>
    void add(const T *host, size_t count, const T **dev_out, size_t extra_zero_elems = 0, bool direct = false, bool device_tail = false)
    {
        *dev_out = nullptr;
        if (!count && !extra_zero_elems) return;
        static const bool all_staged = getenv("EA1_COEF_STAGED") != nullptr;   // diagnostic: everything through the staging buffer
        if (all_staged) direct = false;
        Part p{host, count * sizeof(T), (count + extra_zero_elems) * sizeof(T), reinterpret_cast<const void **>(dev_out), total, direct, device_tail};
        total += (p.reserve + 255) & ~size_t(255);
        parts.push_back(p);
    }
    // `count` uploaded elements followed by `tail` elements that a kernel writes
    template <class T>
    void add_with_device_tail(const T *host, size_t count, size_t tail, const T **dev_out)
    {
        add(host, count, dev_out, tail, false, true);
    }
    // a region of the arena that is written on the device only; returns its offset from the arena start once
    // the uploaded part has been placed
    size_t reserve_device(size_t bytes)
    {
        const size_t off = device_only;
        device_only += (bytes + 255) & ~size_t(255);
        return off;
    }
    static size_t pad(size_t n) { return (n + 255) & ~size_t(255); }
    size_t upload_len(const Part &p) const { return p.device_tail ? (p.bytes ? pad(p.bytes) : 0) : pad(p.reserve); }
    size_t staged_bytes() const
    {
        size_t staged = 0;
        for (const Part &p : parts)
            if (!p.direct && upload_len(p)) staged = std::max(staged, p.off + upload_len(p));
        return staged;
    }
    // device addresses of the parts, once the arena is known
    void place(uint8_t *d)
    {
        for (const Part &p : parts) *p.out = d + p.off;
    }
    // Fills the staging buffer (same layout as the arena) and lists the copies: staged parts that follow each other
    // without a gap travel as one run, direct parts from their source.  No HIP call.
    void stage(uint8_t *d, uint8_t *pin, std::vector<ea1_job::Copy> *copies) const
    {
        size_t run_begin = 0, run_end = 0;
        bool open = false;
        auto flush = [&] {
            if (open && run_end > run_begin) copies->push_back(ea1_job::Copy{d + run_begin, pin + run_begin, run_end - run_begin, true});
            open = false;
        };
        for (const Part &p : parts) {
            const size_t up = upload_len(p);
            if (p.direct || !up) {
                flush();
                continue;
            }
            if (open && p.off != run_end) flush();
            if (p.bytes) memcpy(pin + p.off, p.src, p.bytes);
            if (up > p.bytes) memset(pin + p.off + p.bytes, 0, up - p.bytes);
            if (!open) {
                run_begin = p.off;
                open = true;
            }
            run_end = p.off + up;
            if (p.device_tail && pad(p.reserve) != up) flush();
        }
        flush();
        // pageable source: the runtime has staged the bytes when the copy call returns; pinned source (ea1_host_alloc):
        // a true asynchronous copy, the caller keeps the array until ea1_job_wait / the first submit
        for (const Part &p : parts)
            if (p.direct && p.bytes) copies->push_back(ea1_job::Copy{d + p.off, p.src, p.bytes, pinned_by_library(p.src, p.bytes)});
    }
};

int lr_units(int unit, int size)
{
    int n = (size + (unit >> 1)) / unit;
    return n < 1 ? 1 : n;
}

}  // namespace


static void destroy_chain(Chain &c)
{
    if (c.arena) (void)hipFree(c.arena);
    if (c.up_arena) (void)hipFree(c.up_arena);
    if (c.d_slots) (void)hipFree(c.d_slots);
    if (c.d_dir_var) (void)hipFree(c.d_dir_var);
    c = Chain();
}

extern "C" int ea1_recon_create_batch(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device)
{
    return ea1_recon_create_layout(out, width, height, nchains, device, EA1_LAYOUT_420);
}

extern "C" int ea1_recon_layout(const ea1_recon *r) { return r ? r->layout : ea1::fail(EA1_EINVAL, "NULL handle"); }

extern "C" int ea1_recon_create_layout(ea1_recon **out, uint32_t width, uint32_t height, int nchains, int device, int layout)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_recon_create: out is NULL");
    *out = nullptr;
    if (layout != EA1_LAYOUT_420 && layout != EA1_LAYOUT_444 && layout != EA1_LAYOUT_422) return ea1::fail(EA1_EINVAL, "ea1_recon_create: layout %d", layout);
    if (!width || !height || (width & 7) || (height & 7) || width > 16384 || height > 16384)
        return ea1::fail(EA1_EINVAL, "ea1_recon_create: %ux%u must be a non-zero multiple of 8", width, height);
    if (nchains < 1 || nchains > ea1::kMaxChains) return ea1::fail(EA1_EINVAL, "ea1_recon_create: nchains %d not in [1, %d]", nchains, ea1::kMaxChains);
    const int ndev = ea1_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev)
        return ea1::fail(EA1_ENODEV, "ea1_recon_create: device %d not available (%d visible)", device, ndev);
    ea1::DeviceGuard g(device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", device);
    auto *r = new (std::nothrow) ea1_recon();
    if (!r) return ea1::fail(EA1_ENOMEM, "out of memory");
    r->device = device;
    r->width = width;
    r->height = height;
    r->layout = layout;
    r->ssx = layout == EA1_LAYOUT_444 ? 0 : 1;
    r->ssy = layout == EA1_LAYOUT_420 ? 1 : 0;
    const int ssx = r->ssx, ssy = r->ssy;
    DevPic probe;
    layout_pic(probe, nullptr, int(width), int(height), ssx, ssy, &r->pic_bytes);
    const int npics = EA1_NUM_REF_SLOTS + 3;   // reference slots, work, CDEF, display
    const int w8 = width / 8, h8 = height / 8;
    const size_t arena_bytes = r->pic_bytes * (npics + 2);  // + int16 residual planes (2 bytes per sample)
    r->chains.resize(size_t(nchains));
    int rc = EA1_OK;
    for (int ci = 0; ci < nchains && rc == EA1_OK; ci++) {
        Chain &c = r->chains[size_t(ci)];
        if (hipMalloc(&c.arena, arena_bytes) != hipSuccess) rc = ea1::fail(EA1_ENOMEM, "hipMalloc(pictures) failed");
        if (rc == EA1_OK && hipMemset(c.arena, 128, arena_bytes) != hipSuccess) rc = ea1::fail(EA1_EHIP, "hipMemset failed");
        if (rc != EA1_OK) break;
        for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) layout_pic(c.slots[i], c.arena + r->pic_bytes * i, int(width), int(height), ssx, ssy, nullptr);
        layout_pic(c.work, c.arena + r->pic_bytes * EA1_NUM_REF_SLOTS, int(width), int(height), ssx, ssy, nullptr);
        layout_pic(c.cdef, c.arena + r->pic_bytes * (EA1_NUM_REF_SLOTS + 1), int(width), int(height), ssx, ssy, nullptr);
        layout_pic(c.display, c.arena + r->pic_bytes * (EA1_NUM_REF_SLOTS + 2), int(width), int(height), ssx, ssy, nullptr);
        {
            // residual planes: same geometry as a picture, two bytes per sample
            uint8_t *base = c.arena + r->pic_bytes * npics;
            for (int pl = 0; pl < 3; pl++) c.resid[pl] = reinterpret_cast<int16_t *>(base + 2 * size_t(c.work.p[pl] - c.work.p[0]));
        }
        if (hipMalloc(&c.d_slots, sizeof(c.slots)) != hipSuccess ||
            hipMemcpy(c.d_slots, c.slots, sizeof(c.slots), hipMemcpyHostToDevice) != hipSuccess)
            rc = ea1::fail(EA1_EHIP, "slot table upload failed");
        if (rc == EA1_OK && hipMalloc(&c.d_dir_var, size_t(w8) * h8 * 2) != hipSuccess) rc = ea1::fail(EA1_ENOMEM, "hipMalloc failed");
    }
    if (rc == EA1_OK) {
        std::vector<uint8_t> wt(ea1::wedge_table_bytes());
        ea1::build_wedge_table(wt.data());
        if (hipMalloc(&r->d_wedge, wt.size()) != hipSuccess || hipMemcpy(r->d_wedge, wt.data(), wt.size(), hipMemcpyHostToDevice) != hipSuccess)
            rc = ea1::fail(EA1_ENOMEM, "wedge table upload failed");
    }
    // the warp filters: the table of include/ea1_av1_tables.inc ([recalled], structurally cross-checked) unless the caller brings its own
    static const int8_t kWarpedFilters[EA1_WARP_FILTER_ROWS][8] = {EA1_WARPED_FILTERS_VALUES};
    if (rc == EA1_OK && (hipMalloc(&r->d_warp_filter, EA1_WARP_FILTER_ROWS * 8) != hipSuccess ||
                         hipMemcpy(r->d_warp_filter, kWarpedFilters, EA1_WARP_FILTER_ROWS * 8, hipMemcpyHostToDevice) != hipSuccess))
        rc = ea1::fail(EA1_ENOMEM, "hipMalloc(warp filter) failed");
    if (rc == EA1_OK) r->warp_filter_set = true;
    if (rc == EA1_OK) {
        r->pool = std::make_shared<JobPool>();
        r->pool->device = device;
        r->workers.reset(new (std::nothrow) Workers());   // threads start with the first batch
        // the hand-over streams run at the highest priority: their kernels (the pull of the next frames' bytes, the
        // derivation) are small and feed the next submit, while the reconstruction kernels they run beside fill the
        // device; at equal priority the pull waits for wave slots behind thousands of workgroups
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        static const bool flat_prio = getenv("EA1_FLAT_PRIORITY") != nullptr;   // diagnostic
        if (flat_prio) prio_greatest = prio_least;
        if (hipStreamCreateWithPriority(&r->pool->copy, hipStreamNonBlocking, prio_greatest) != hipSuccess) rc = ea1::fail(EA1_EHIP, "hipStreamCreate failed");
        if (rc == EA1_OK && getenv("EA1_COPY_STREAMS") && atoi(getenv("EA1_COPY_STREAMS")) == 2 &&
            hipStreamCreateWithPriority(&r->pool->copy2, hipStreamNonBlocking, prio_greatest) != hipSuccess)
            rc = ea1::fail(EA1_EHIP, "hipStreamCreate failed");
        for (hipStream_t &st : r->pool->derive)
            if (rc == EA1_OK && hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_greatest) != hipSuccess) rc = ea1::fail(EA1_EHIP, "hipStreamCreate failed");
    }
    if (rc == EA1_OK && (hipMalloc(&r->d_wave_counters, 2 * ea1::kMaxChains * sizeof(uint32_t)) != hipSuccess ||
                         hipMemset(r->d_wave_counters, 0, 2 * ea1::kMaxChains * sizeof(uint32_t)) != hipSuccess))
        rc = ea1::fail(EA1_ENOMEM, "hipMalloc(counters) failed");
    if (rc != EA1_OK) {
        ea1_recon_destroy(r);
        return rc;
    }
    *out = r;
    return EA1_OK;
}

extern "C" int ea1_recon_create(ea1_recon **out, uint32_t width, uint32_t height, int device)
{
    return ea1_recon_create_batch(out, width, height, 1, device);
}

extern "C" int ea1_recon_num_chains(const ea1_recon *r) { return r ? int(r->chains.size()) : ea1::fail(EA1_EINVAL, "NULL handle"); }

extern "C" void ea1_recon_destroy(ea1_recon *r)
{
    if (!r) return;
    ea1::DeviceGuard g(r->device);
    (void)hipDeviceSynchronize();
    for (Chain &c : r->chains) destroy_chain(c);
    if (r->d_wave_counters) (void)hipFree(r->d_wave_counters);
    if (r->d_wedge) (void)hipFree(r->d_wedge);
    if (r->d_warp_filter) (void)hipFree(r->d_warp_filter);
    if (r->d_upscale) (void)hipFree(r->d_upscale);
    if (r->d_grain) (void)hipFree(r->d_grain);
    delete r->prof;
    delete r;
}

extern "C" int ea1_recon_set_warp_filter(ea1_recon *r, const int8_t *filter)
{
    if (!r || !filter) return ea1::fail(EA1_EINVAL, "ea1_recon_set_warp_filter: NULL argument");
    for (int row = 0; row < EA1_WARP_FILTER_ROWS; row++) {
        int sum = 0;
        for (int k = 0; k < 8; k++) sum += filter[row * 8 + k];
        if (sum != 128) return ea1::fail(EA1_EINVAL, "ea1_recon_set_warp_filter: row %d sums to %d, not 128", row, sum);
    }
    ea1::DeviceGuard g(r->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", r->device);
    // the table may be in use by frames in flight: drain first (set-up call, not on the frame path)
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(r->d_warp_filter, filter, EA1_WARP_FILTER_ROWS * 8, hipMemcpyHostToDevice) != hipSuccess)
        return ea1::fail(EA1_EHIP, "ea1_recon_set_warp_filter: upload failed");
    r->warp_filter_set = true;
    return EA1_OK;
}

extern "C" int ea1_recon_flush(ea1_recon *r)
{
    if (!r) return ea1::fail(EA1_EINVAL, "ea1_recon_flush: NULL handle");
    for (Chain &c : r->chains)
        for (bool &v : c.valid) v = false;
    return EA1_OK;
}

extern "C" int ea1_recon_set_ref_sized(ea1_recon *r, int chain, int slot, uint32_t width, uint32_t height, const uint8_t *y, size_t y_stride,
                                       const uint8_t *u, const uint8_t *v, size_t c_stride)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS || !y || !u || !v)
        return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref: bad arguments");
    if (!width || !height || (width & 7) || (height & 7) || width > r->width || height > r->height)
        return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref: picture %ux%u: multiples of 8 up to the context's %ux%u", width, height, r->width, r->height);
    ea1::DeviceGuard g(r->device);
    Chain &c = r->chains[size_t(chain)];
    const DevPic &p = c.slots[slot];
    const uint8_t *src[3] = {y, u, v};
    const size_t ss[3] = {y_stride, c_stride, c_stride};
    for (int pl = 0; pl < 3; pl++)
        EA1_HIP(hipMemcpy2D(p.p[pl], p.stride[pl], src[pl], ss[pl], pl ? width >> r->ssx : width, pl ? height >> r->ssy : height, hipMemcpyHostToDevice));
    c.valid[slot] = true;
    c.slot_w[slot] = c.slot_vw[slot] = int(width);
    c.slot_h[slot] = c.slot_vh[slot] = int(height);
    c.slot_pad[slot] = 0;
    return EA1_OK;
}

extern "C" int ea1_recon_set_ref_crop(ea1_recon *r, int chain, int slot, uint8_t crop)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref_crop: bad arguments");
    if ((crop & 15) > 7 || (crop >> 4) > 7) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref_crop: crop 0x%02x", crop);
    Chain &c = r->chains[size_t(chain)];
    if (!c.valid[slot]) return ea1::fail(EA1_ESTATE, "ea1_recon_set_ref_crop: slot %d holds no picture", slot);
    c.slot_vw[slot] = c.slot_w[slot] - (crop & 15);
    c.slot_vh[slot] = c.slot_h[slot] - (crop >> 4);
    return EA1_OK;
}

extern "C" int ea1_recon_set_ref_pad(ea1_recon *r, int chain, int slot, uint8_t pad, uint8_t crop)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref_pad: bad arguments");
    if ((crop & 15) > 7 || (crop >> 4) > 7) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref_pad: pad 0x%02x crop 0x%02x", pad, crop);
    Chain &c = r->chains[size_t(chain)];
    if (!c.valid[slot]) return ea1::fail(EA1_ESTATE, "ea1_recon_set_ref_pad: slot %d holds no picture", slot);
    c.slot_pad[slot] = pad;
    c.slot_vw[slot] = c.slot_w[slot] - 8 * (pad & 15) - (crop & 15);
    c.slot_vh[slot] = c.slot_h[slot] - 8 * (pad >> 4) - (crop >> 4);
    if (c.slot_vw[slot] <= 0 || c.slot_vh[slot] <= 0) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref_pad: nothing left of the picture");
    return EA1_OK;
}

extern "C" int ea1_recon_set_ref_chain(ea1_recon *r, int chain, int slot, const uint8_t *y, size_t y_stride,
                                       const uint8_t *u, const uint8_t *v, size_t c_stride)
{
    if (!r) return ea1::fail(EA1_EINVAL, "ea1_recon_set_ref: bad arguments");
    return ea1_recon_set_ref_sized(r, chain, slot, r->width, r->height, y, y_stride, u, v, c_stride);
}

extern "C" int ea1_recon_slot_size(const ea1_recon *r, int chain, int slot, uint32_t *width, uint32_t *height)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS || !width || !height)
        return ea1::fail(EA1_EINVAL, "ea1_recon_slot_size: bad arguments");
    const Chain &c = r->chains[size_t(chain)];
    if (!c.valid[slot]) return ea1::fail(EA1_ESTATE, "ea1_recon_slot_size: slot %d holds no picture", slot);
    *width = uint32_t(c.slot_w[slot]);
    *height = uint32_t(c.slot_h[slot]);
    return EA1_OK;
}

extern "C" int ea1_recon_set_ref(ea1_recon *r, int slot, const uint8_t *y, size_t y_stride, const uint8_t *u,
                                 const uint8_t *v, size_t c_stride)
{
    return ea1_recon_set_ref_chain(r, 0, slot, y, y_stride, u, v, c_stride);
}

// No synchronisation: the arena goes back to the pool behind the event of the job's last use (its last submit, or
// its upload if it was never submitted) and is handed out again once that event has completed.
extern "C" void ea1_job_destroy(ea1_job *job)
{
    if (!job) return;
    ea1::DeviceGuard g(job->device);
    bool drained = false;
    for (auto &b : job->bound)
        if (b.own_alloc) {   // a job bound to more than one chain: rare, freed the blunt way
            if (!drained) (void)hipDeviceSynchronize();
            drained = true;
            (void)hipFree(b.d_view);
        }
    if (job->pool) {
        JobPool &pool = *job->pool;
        if (job->arena.p) {
            if (job->last_use && job->last_use->ev && !drained) {
                if (event_done(job->last_use->ev)) pool.give(job->arena, false, nullptr, false);
                else pool.give_behind(job->arena, false, job->last_use);   // parked until that submit has finished
            } else if (job->ready && job->ready->ev && !job->ready_seen && !drained && !event_done(job->ready->ev)) {
                pool.give_behind(job->arena, false, job->ready);
            } else {
                pool.give(job->arena, false, nullptr, false);
            }
        }
        if (job->pending && job->pending->pin.p) pool.give(job->pending->pin, true, nullptr, false);   // prepared, never issued
    }
    delete job;
}

// blocks until the job's upload and derivation have finished (its input arrays may then be reused even when they
// are pinned memory); submits do not need it -- they order themselves behind the upload on the device
extern "C" int ea1_job_wait(ea1_job *job)
{
    if (!job) return ea1::fail(EA1_EINVAL, "ea1_job_wait: NULL job");
    if (job->ready_seen || !job->ready || !job->ready->ev) return EA1_OK;
    ea1::DeviceGuard g(job->device);
    EA1_HIP(hipEventSynchronize(job->ready->ev));
    job->ready_seen = true;
    return EA1_OK;
}
extern "C" int ea1_tx_size_index(int log2w, int log2h)
{
    if (log2w < 2 || log2w > 6 || log2h < 2 || log2h > 6) return -1;
    return ea1::tx_size_index(log2w, log2h);
}

namespace {

bool is_smooth_mode(int m) { return m == EA1_SMOOTH_PRED || m == EA1_SMOOTH_V_PRED || m == EA1_SMOOTH_H_PRED; }
bool is_directional(int m) { return m >= EA1_V_PRED && m <= EA1_D67_PRED; }
const int kModeAngle[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};

// a 4-sample luma side: such inter blocks are predicted piece by piece on the host-made list, whatever the chroma layout
bool is_sub8(const ea1_block &b) { return b.bwl == 2 || b.bhl == 2; }
struct Rect {
    int x, y, w, h;
};
// chroma ownership and geometry of blocks (see include/easyav1_amd_video.h) for the chroma subsampling of the context: in a subsampled
// direction the chroma of two 4-sample blocks belongs to the second one and covers both
struct ChromaGeom {
    int ssx, ssy;
    bool has_chroma(const ea1_block &b) const { return (!ssx || (b.x4 & 1) || b.bwl > 2) && (!ssy || (b.y4 & 1) || b.bhl > 2); }
    bool sub8(const ea1_block &b) const { return (ssx && b.bwl == 2) || (ssy && b.bhl == 2); }
    Rect rect(const ea1_block &b, int pl) const
    {
        const int bw = 1 << b.bwl, bh = 1 << b.bhl;
        if (pl == 0) return Rect{b.x4 * 4, b.y4 * 4, bw, bh};
        return Rect{ssx ? (b.x4 & ~1) * 2 : b.x4 * 4, ssy ? (b.y4 & ~1) * 2 : b.y4 * 4, ssx ? std::max(bw, 8) / 2 : bw, ssy ? std::max(bh, 8) / 2 : bh};
    }
    unsigned edge_flags(const ea1_block &b, int pl) const
    {
        if (pl == 0 || !sub8(b)) return b.flags & (EA1_BLK_HAVE_LEFT | EA1_BLK_HAVE_TOP | EA1_BLK_HAVE_TOP_RIGHT | EA1_BLK_HAVE_BOTTOM_LEFT);
        return ((b.flags & EA1_BLK_CHROMA_HAVE_LEFT) ? EA1_BLK_HAVE_LEFT : 0u) | ((b.flags & EA1_BLK_CHROMA_HAVE_TOP) ? EA1_BLK_HAVE_TOP : 0u) |
               ((b.flags & EA1_BLK_CHROMA_HAVE_TOP_RIGHT) ? EA1_BLK_HAVE_TOP_RIGHT : 0u) |
               ((b.flags & EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT) ? EA1_BLK_HAVE_BOTTOM_LEFT : 0u);
    }
};

// rows of one wavefront item: a wave covers 64 / (w / 4) rows per step and at most 4 steps
int rows_per_part(int w) { return w == 4 ? 64 : (w == 8 ? 32 : (w == 16 ? 64 : (w == 32 ? 32 : 16))); }

}  // namespace

namespace {

struct LevelItem {
    uint32_t level;
    ea1::IntraItem it;
};

// Host scratch of the hand-over, one per thread: the vectors keep their capacity from frame to frame, so a frame
// costs no allocation and no first-touch page faults (a fifth of the time before).
struct HostLists {
    std::vector<uint32_t> map;     // block index per 4x4 luma unit, kept for the units neighbour look-ups read (a block's bottom row and right column)
    std::vector<uint64_t> done;    // one bit per 4x4 luma unit: covered by a block earlier in the list
    std::vector<uint16_t> lvl[3];  // wavefront level of the intra item covering each 4x4-sample unit (0: inter / none); all zero between frames (levels stay below 65 535: checked)
    bool lvl_clean = false;
    struct Touched {
        uint32_t pl, ux0, uy0, nw, nh;
    };
    std::vector<Touched> touched;
    std::vector<ea1::McItem> mc[3], mc_ext[3];
    std::vector<ea1::ObmcItem> obmc[2];
    std::vector<ea1::PieceItem> pieces;
    std::vector<ea1::WarpItem> warps, warps_chroma;   // luma rectangles / chroma rectangles (launched after the luma ones)
    std::vector<LevelItem> intra;
    std::vector<ea1::IntraItem> intra_items;
    std::vector<uint32_t> level_start, at;
    std::vector<ea1::PalRef> pal_refs;
    std::vector<uint32_t> host_blocks;   // indices of the blocks the lists phase visits
    Packer pk;
};

// transform size byte -> list index (-1: not a transform size), and which transform types a size takes
struct TxTables {
    int8_t cls[256];
    uint8_t size_of[ea1::kTxSizes];   // list index -> size byte
    uint8_t ok[ea1::kTxSizes][256];
    TxTables()
    {
        memset(ok, 0, sizeof ok);
        for (int sz = 0; sz < 256; sz++) {
            const int wl = sz & 15, hl = sz >> 4;
            cls[sz] = int8_t(ea1_tx_size_index(wl, hl));
            if (cls[sz] < 0) continue;
            size_of[cls[sz]] = uint8_t(sz);
            const int tw = 1 << wl, th = 1 << hl;
            for (int tp = 0; tp < 256; tp++) {
                bool good;
                if (tp == EA1_TXTP_WHT) {
                    good = tw == 4 && th == 4;
                } else {
                    const int rk = tp & 3, ck = (tp >> 2) & 3;
                    auto k_ok = [](int k, int n) { return k == EA1_TX_DCT || (k == EA1_TX_IDENTITY && n <= 32) || ((k == EA1_TX_ADST || k == EA1_TX_FLIPADST) && n <= 16); };
                    good = !(tp >> 4) && k_ok(rk, tw) && k_ok(ck, th) && !((tw == 64 || th == 64) && tp);
                }
                ok[cls[sz]][tp] = good ? 1 : 0;
            }
        }
    }
};

}  // namespace

// strip_x0 / strip_x1: the luma columns whose blocks the job reconstructs (tile-column split);
// the whole frame is validated and its deblock / CDEF / restoration side information derived either way, because
// the filters of a strip also cover the margin next to it
//
// job_prepare is the host half (any thread, no HIP call once the pool is warm): validation, the work lists the host
// still makes, staging into pinned memory.  job_issue is the device half (one thread): copies, derivation kernels.
// lengths of the two arrays whose size follows from the frame (CDEF indices per 64x64, restoration units per plane): an ea1_ir_frame states them, and the
// entry points that take one have them checked (a caller of the array-wise ea1_job_create vouches for its arrays, as before)
struct ArrayLengths {
    uint32_t num_sb;
    uint32_t lr_count[3];
};
static int job_prepare(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                       const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx,
                       const ea1_lr_unit *const lr[3], int strip_x0, int strip_x1, ea1_job **out, const ArrayLengths *lengths = nullptr)
{
    if (!r || !desc || !out || (desc->num_blocks && !blocks) || (desc->num_tx && !txs) || (desc->num_coefs && !coefs) ||
        (desc->num_aux && !aux))
        return ea1::fail(EA1_EINVAL, "ea1_job_create: NULL argument");
    *out = nullptr;
    if (!desc->width || !desc->height || (desc->width & 7) || (desc->height & 7) || desc->width > r->width || desc->height > r->height)
        return ea1::fail(EA1_EINVAL, "ea1_job_create: frame %ux%u: multiples of 8 up to the context's %ux%u", desc->width,
                         desc->height, r->width, r->height);
    if (desc->dst_slot >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "ea1_job_create: dst_slot %u", desc->dst_slot);
    if (desc->layout != r->layout) return ea1::fail(EA1_EINVAL, "ea1_job_create: the frame's chroma layout is %u, the context's %d (ea1_recon_create_layout)", desc->layout, r->layout);
    if (r->layout != EA1_LAYOUT_420 && (strip_x0 > 0 || strip_x1 < int(desc->width))) return ea1::fail(EA1_ENOTSUP, "ea1_job_create: a tile-column split of a 4:4:4 / 4:2:2 frame");
    if (desc->upscaled_width) {
        // super-resolution: the frame above is the downscaled one; SuperresDenom 9..16 eighths
        const uint32_t vis = desc->width - 8u * (desc->pad & 15u) - (desc->crop & 15u);
        if (!r->upscale_set) return ea1::fail(EA1_ESTATE, "ea1_job_create: super-resolution, but the context has no Upscale_Filter table (ea1_recon_set_upscale_filter)");
        if (desc->upscaled_width <= vis || desc->upscaled_width > 2 * vis + 16 || ((desc->upscaled_width + 7u) & ~7u) > r->width)
            return ea1::fail(EA1_EINVAL, "ea1_job_create: upscaled_width %u for a frame %u samples wide in a context of %u", desc->upscaled_width, vis, r->width);
        if (strip_x0 > 0 || strip_x1 < int(desc->width)) return ea1::fail(EA1_ENOTSUP, "ea1_job_create: super-resolution with a tile-column split");
    }
    const int W = int(desc->width), H = int(desc->height);
    const int w8 = W / 8, h8 = H / 8, w4 = W / 4, h4 = H / 4;
    const int ssx = r->ssx, ssy = r->ssy;
    const int cw4 = w4 >> ssx, ch4 = h4 >> ssy;   // 4x4 units of a chroma plane
    const ChromaGeom cg{ssx, ssy};
    const int sb_cols = (W + 63) / 64, sb_rows = (H + 63) / 64;
    const uint32_t nb = desc->num_blocks, ntx = desc->num_tx;

    // EA1_JOB_TIMING=1: per-phase host time of the hand-over on stderr (tools/time_job_create.py)
    static const bool timing = getenv("EA1_JOB_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_start = now();
    decltype(t_start) t_valid, t_lists;
    // ---- validation, 8x8 block map, per-plane transform maps (4x4-sample units) --------------
    // (host copies of the maps serve the validation and the list derivation below; the kernels get theirs from
    // av1_derive.hip.  The luma transform grid is checked per block on a bitmap, only the small chroma grids are
    // kept as maps -- for the rule that U and V share one grid)
    static thread_local HostLists tl_lists;
    static const TxTables tt;
    HostLists &hs = tl_lists;
    std::vector<uint32_t> &map = hs.map;
    if (map.size() != size_t(w4) * h4) map.resize(size_t(w4) * h4);
    const int dw = (w4 + 63) / 64;   // words per row of the coverage bitmap
    hs.done.assign(size_t(dw) * h4, 0);
    hs.host_blocks.clear();
    uint64_t *const done = hs.done.data();
    // units [ux0, ux1) of row uy all covered (at most 33 units: two words)
    auto row_done = [&](int uy, int ux0, int ux1) {
        const int wa = ux0 >> 6, wb = (ux1 - 1) >> 6;
        const uint64_t ma = ~0ull << (ux0 & 63), mb = ~0ull >> (63 - ((ux1 - 1) & 63));
        const uint64_t *rw = done + size_t(uy) * dw;
        if (wa == wb) return (rw[wa] & ma & mb) == (ma & mb);
        return (rw[wa] & ma) == ma && (rw[wb] & mb) == mb;
    };
    // luma rectangle, clipped to the frame: every unit it touches belongs to a block earlier in the list
    // (clipped to the TRUE coded frame, ea1_frame_desc.pad: edge fetches repeat its last sample beyond it, and what lies out there
    // belongs to filler blocks at the end of the list)
    const int coded_W = W - 8 * (desc->pad & 15), coded_H = H - 8 * (desc->pad >> 4);
    auto decoded = [&](int px0, int py0, int pw_, int ph_) {
        const int ux0 = std::max(px0, 0) / 4, ux1 = (std::min(px0 + pw_, coded_W) + 3) / 4;
        const int uy0 = std::max(py0, 0) / 4, uy1 = (std::min(py0 + ph_, coded_H) + 3) / 4;
        if (ux0 >= ux1) return true;
        for (int uy = uy0; uy < uy1; uy++)
            if (!row_done(uy, ux0, ux1)) return false;
        return true;
    };
    uint64_t ref_pixels = 0, coef_count = 0, inter_px = 0, covered4 = 0;
    // what the device-made lists will hold (av1_derive.hip derive_items_kernel follows the same rules)
    uint32_t tx_bucket[ea1::kTxSizes][4];
    memset(tx_bucket, 0, sizeof tx_bucket);
    uint32_t tx_zero_fill[ea1::kTxSizes] = {0};
    uint32_t mc_dev[3] = {0, 0, 0}, mc_ext_dev[3] = {0, 0, 0};
    // only the coefficients of the strip's own transform blocks travel to the device: [coef_lo, coef_hi) of the arena
    uint32_t coef_lo = desc->num_coefs, coef_hi = 0;
    if (desc->coef_layout > EA1_COEFS_DIAGONAL) return ea1::fail(EA1_EINVAL, "ea1_job_create: coef_layout %u", desc->coef_layout);
    const bool corner_layout = desc->coef_layout == EA1_COEFS_CORNER, diagonal_layout = desc->coef_layout == EA1_COEFS_DIAGONAL;
    uint64_t expanded_entries = 0;   // EA1_COEFS_DIAGONAL: entries of the strip's coded blocks in the corner layout the device expands them to
    bool luma_odd_row_edges = false;   // a luma transform block starts on an odd 4-row unit (4-sample-high transforms)
    uint32_t ref_mask = 0;
    for (uint32_t bi = 0; bi < nb; bi++) {
        const ea1_block &b = blocks[bi];
        // (the 128-sample sizes are 128x128, 128x64 and 64x128 only)
        if (b.bwl < 2 || b.bwl > 7 || b.bhl < 2 || b.bhl > 7 || std::abs(int(b.bwl) - int(b.bhl)) > 2 || (std::max(b.bwl, b.bhl) == 7 && std::min(b.bwl, b.bhl) < 6))
            return ea1::fail(EA1_EINVAL, "block %u: unsupported size 2^%u x 2^%u", bi, b.bwl, b.bhl);
        const int bw = 1 << b.bwl, bh = 1 << b.bhl;
        const int x = b.x4 * 4, y = b.y4 * 4;
        if ((x & (bw - 1)) || (y & (bh - 1)) || x + bw > W || y + bh > H)
            return ea1::fail(EA1_EINVAL, "block %u: position (%d,%d) size %dx%d outside the frame or misaligned", bi, x, y, bw, bh);
        const bool in_strip = x >= strip_x0 && x < strip_x1;
        // blocks the host itself makes work items for (the intra wavefront, OBMC, warped and 4-sample predictions): all
        // others -- plain translated inter blocks, nine in ten of an inter frame -- are the device's
        if (in_strip && (!(b.flags & EA1_BLK_INTER) || (b.flags & (EA1_BLK_OBMC | EA1_BLK_INTERINTRA | EA1_BLK_WARP)) || is_sub8(b))) hs.host_blocks.push_back(bi);
        if (x < strip_x1 && x + bw > strip_x0 && (x < strip_x0 || x + bw > strip_x1))
            return ea1::fail(EA1_EINVAL, "block %u: columns %d..%d cross the strip boundary (%d..%d): strips must follow tile columns", bi, x, x + bw, strip_x0, strip_x1);
        {
            // edge availability must be geometrically possible: the flags steer reads of neighbouring samples and
            // of the block map (a block at the frame's top cannot have a top neighbour, ...)
            const unsigned fl = b.flags;
            const int cx = ssx ? x & ~7 : x, cy = ssy ? y & ~7 : y, cwl = ssx ? std::max(bw, 8) : bw, chl = ssy ? std::max(bh, 8) : bh;   // luma area of the chroma block of a sub-8 block
            if (((fl & EA1_BLK_HAVE_TOP) && y == 0) || ((fl & EA1_BLK_HAVE_LEFT) && x == 0) ||
                ((fl & EA1_BLK_HAVE_TOP_RIGHT) && (!(fl & EA1_BLK_HAVE_TOP) || x + bw >= W)) ||
                ((fl & EA1_BLK_HAVE_BOTTOM_LEFT) && (!(fl & EA1_BLK_HAVE_LEFT) || y + bh >= H)) ||
                ((fl & EA1_BLK_CHROMA_HAVE_TOP) && cy == 0) || ((fl & EA1_BLK_CHROMA_HAVE_LEFT) && cx == 0) ||
                ((fl & EA1_BLK_CHROMA_HAVE_TOP_RIGHT) && (!(fl & EA1_BLK_CHROMA_HAVE_TOP) || cx + cwl >= W)) ||
                ((fl & EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT) && (!(fl & EA1_BLK_CHROMA_HAVE_LEFT) || cy + chl >= H)))
                return ea1::fail(EA1_EINVAL, "block %u: edge availability flags 0x%04x impossible at (%d,%d)", bi, fl, x, y);
            // ... and must point at samples of blocks EARLIER in the list: the intra wavefront orders an item behind the
            // items its edges come from, a neighbour that is decoded later would be read before it exists
            bool ok = true;
            // (inter blocks read no neighbouring samples -- OBMC takes its neighbours' MOTION, from the finished block
            // map -- unless they blend an intra prediction in)
            const bool reads_edges = !(fl & EA1_BLK_INTER) || (fl & EA1_BLK_INTERINTRA);
            if (reads_edges) {
                if (fl & EA1_BLK_HAVE_TOP) ok &= decoded(x, y - 4, bw, 4);
                if (fl & EA1_BLK_HAVE_LEFT) ok &= decoded(x - 4, y, 4, bh);
                if ((fl & EA1_BLK_HAVE_TOP) && (fl & EA1_BLK_HAVE_LEFT)) ok &= decoded(x - 4, y - 4, 4, 4);
                if (fl & EA1_BLK_HAVE_TOP_RIGHT) ok &= decoded(x + bw, y - 4, bw, 4);
                if (fl & EA1_BLK_HAVE_BOTTOM_LEFT) ok &= decoded(x - 4, y + bh, 4, bh);
                if (fl & EA1_BLK_CHROMA_HAVE_TOP) ok &= decoded(cx, cy - 4, cwl, 4);
                if (fl & EA1_BLK_CHROMA_HAVE_LEFT) ok &= decoded(cx - 4, cy, 4, chl);
                if ((fl & EA1_BLK_CHROMA_HAVE_TOP) && (fl & EA1_BLK_CHROMA_HAVE_LEFT)) ok &= decoded(cx - 4, cy - 4, 4, 4);
                if (fl & EA1_BLK_CHROMA_HAVE_TOP_RIGHT) ok &= decoded(cx + cwl, cy - 4, cwl, 4);
                if (fl & EA1_BLK_CHROMA_HAVE_BOTTOM_LEFT) ok &= decoded(cx - 4, cy + chl, 4, chl);
            }
            if (!ok) return ea1::fail(EA1_EINVAL, "block %u: an edge it is flagged to have (0x%04x) belongs to a block later in the list", bi, fl);
        }
        if (is_sub8(b) && (b.flags & (EA1_BLK_COMPOUND | EA1_BLK_OBMC | EA1_BLK_INTERINTRA)))
            return ea1::fail(EA1_EINVAL, "block %u: compound / OBMC / inter-intra need block sides of at least 8", bi);
        if ((b.flags & (EA1_BLK_PALETTE_Y | EA1_BLK_PALETTE_UV)) && (bw * bh < 64 || bw > 64 || bh > 64))   // from 8x8 on, 4x16 and 16x4 included
            return ea1::fail(EA1_EINVAL, "block %u: a palette needs a block of 64 to 64x64 samples", bi);
        if (b.flags & EA1_BLK_INTER) {
            if (!desc->is_inter) return ea1::fail(EA1_EINVAL, "block %u: inter block in an intra frame", bi);
            const int nref = (b.flags & EA1_BLK_COMPOUND) ? 2 : 1;
            for (int k = 0; k < nref; k++) {
                if (b.ref[k] >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "block %u: reference slot %u", bi, b.ref[k]);
                ref_mask |= 1u << b.ref[k];
            }
            if ((b.mode_y >> 4)) return ea1::fail(EA1_EINVAL, "block %u: filter kind", bi);
            const bool wedge_size = bw <= 32 && bh <= 32;
            if (b.flags & EA1_BLK_COMPOUND) {
                if (b.comp_type > EA1_COMP_DIFFWTD || (b.comp_type == EA1_COMP_DIST && (b.comp_arg < 1 || b.comp_arg > 15)) ||
                    (b.comp_type == EA1_COMP_WEDGE && (!wedge_size || b.comp_arg > 31)) || (b.comp_type == EA1_COMP_DIFFWTD && b.comp_arg > 1))
                    return ea1::fail(EA1_EINVAL, "block %u: compound type %u / argument %u", bi, b.comp_type, b.comp_arg);
                if (b.flags & (EA1_BLK_OBMC | EA1_BLK_INTERINTRA)) return ea1::fail(EA1_EINVAL, "block %u: OBMC / inter-intra need a single reference", bi);
            }
            if ((b.flags & EA1_BLK_OBMC) && (b.flags & EA1_BLK_INTERINTRA)) return ea1::fail(EA1_EINVAL, "block %u: OBMC and inter-intra exclude each other", bi);
            if ((b.flags & EA1_BLK_INTERINTRA) && (!wedge_size || (b.ii_mode & 0x7C) || ((b.ii_mode & EA1_II_WEDGE) && b.ii_arg > 15)))
                return ea1::fail(EA1_EINVAL, "block %u: inter-intra needs a block of at most 32x32, mode 0..3, wedge 0..15", bi);
            if (b.flags & EA1_BLK_WARP) {
                if (!r->warp_filter_set) return ea1::fail(EA1_EINVAL, "block %u: warped motion, but the context has no warp filter table (ea1_recon_set_warp_filter)", bi);
                if (is_sub8(b) || !b.warp_refs || b.warp_refs >= (1u << nref))
                    return ea1::fail(EA1_EINVAL, "block %u: warped motion needs a block of at least 8x8 and warp_refs within its references", bi);
                const int nm = (b.warp_refs & 1) + ((b.warp_refs >> 1) & 1);
                if ((b.aux_off & 3) || uint64_t(b.aux_off) + 24u * nm > desc->num_aux) return ea1::fail(EA1_EINVAL, "block %u: warp matrix range / alignment", bi);
                // the bitstream bounds the parameters (global motion: 12-bit magnitudes at their precision; local warp:
                // clamped fits): non-translational part within 2^15 of the identity, translation within 2^30.  Keeps the
                // 64-bit arithmetic of the shear set-up far from overflow on corrupt input.
                for (int k = 0; k < 6 * nm; k++) {
                    int32_t mv;
                    memcpy(&mv, aux + b.aux_off + 4 * k, 4);
                    const int idx = k % 6;
                    const int64_t dev = idx < 2 ? mv : (idx == 2 || idx == 5 ? int64_t(mv) - 65536 : mv);
                    if (std::llabs(dev) > (idx < 2 ? (int64_t(1) << 30) : (int64_t(1) << 15)))
                        return ea1::fail(EA1_EINVAL, "block %u: warp matrix entry %d = %d out of range", bi, idx, mv);
                }
            } else if (b.warp_refs) {
                return ea1::fail(EA1_EINVAL, "block %u: warp_refs without EA1_BLK_WARP", bi);
            }
            ref_pixels += uint64_t(nref) * bw * bh;
            inter_px += uint64_t(bw) * bh;
            if (in_strip && !is_sub8(b) && !(b.flags & EA1_BLK_WARP)) {   // tiles the device makes
                const int tl = std::min(std::min(int(b.bwl), int(b.bhl)), 5), cls = tl - 3;   // tiles of 8, 16 or 32 samples a side
                const bool ext = (b.flags & EA1_BLK_COMPOUND) && b.comp_type != EA1_COMP_AVG;
                (ext ? mc_ext_dev : mc_dev)[cls] += 1u << (b.bwl + b.bhl - 2 * tl);
            }
        } else {
            if ((b.flags & EA1_BLK_WARP) || b.warp_refs) return ea1::fail(EA1_EINVAL, "block %u: warped motion on an intra block", bi);
            if (b.mode_y == EA1_INTRABC_PRED || b.mode_uv == EA1_INTRABC_PRED) {
                // intra block copy: an intra frame without in-loop filters, both planes' modes say so,
                // whole-sample luma vector, and every unit the copy reads (the extra column / row of
                // a half-sample chroma position included) belongs to an earlier block
                if (desc->is_inter || b.mode_y != b.mode_uv || b.angle_delta[0] || b.angle_delta[1] || b.filter_intra ||
                    (b.flags & (EA1_BLK_PALETTE_Y | EA1_BLK_PALETTE_UV)) || b.pal_size[0] || b.pal_size[1])
                    return ea1::fail(EA1_EINVAL, "block %u: intra block copy needs an intra frame, mode_y == mode_uv and no other intra tool", bi);
                if (desc->lf_level[0] || desc->lf_level[1] || desc->lf_per_block || desc->cdef_enable || desc->lr_type[0] || desc->lr_type[1] || desc->lr_type[2])
                    return ea1::fail(EA1_EINVAL, "block %u: intra block copy in a frame with in-loop filters", bi);
                if ((b.mv[0][0] & 7) || (b.mv[0][1] & 7) || (!b.mv[0][0] && !b.mv[0][1])) return ea1::fail(EA1_EINVAL, "block %u: block vector (%d,%d)", bi, b.mv[0][0], b.mv[0][1]);
                const int dy = b.mv[0][0] / 8, dx = b.mv[0][1] / 8;
                // the chroma block (of a block with a 4-sample side that owns chroma: of the whole 8-sample area) moves by the same vector
                const Rect rc_ = cg.rect(b, 1);
                const bool own_c = cg.has_chroma(b);
                const int cx0 = own_c ? rc_.x + (dx >> ssx) : (x + dx) >> ssx, cy0 = own_c ? rc_.y + (dy >> ssy) : (y + dy) >> ssy;
                const int cx1 = own_c ? cx0 + rc_.w - 1 + (dx & ssx) : (x + dx + bw - 1) >> ssx, cy1 = own_c ? cy0 + rc_.h - 1 + (dy & ssy) : (y + dy + bh - 1) >> ssy;
                const int lx0 = std::min(x + dx, cx0 * (1 << ssx)), ly0 = std::min(y + dy, cy0 * (1 << ssy));
                const int lx1 = std::max(x + dx + bw - 1, cx1 * (1 << ssx) + ssx), ly1 = std::max(y + dy + bh - 1, cy1 * (1 << ssy) + ssy);
                if (lx0 < 0 || ly0 < 0 || lx1 >= W - 8 * int(desc->pad & 15) - int(desc->crop & 15) || ly1 >= H - 8 * int(desc->pad >> 4) - int(desc->crop >> 4))
                    return ea1::fail(EA1_EINVAL, "block %u: block vector leaves the frame", bi);
                if (!decoded(lx0, ly0, lx1 - lx0 + 1, ly1 - ly0 + 1)) return ea1::fail(EA1_EINVAL, "block %u: block vector points at samples that are not decoded yet", bi);
            } else if (b.mode_y > EA1_PAETH_PRED || b.mode_uv > EA1_UV_CFL_PRED) {
                return ea1::fail(EA1_EINVAL, "block %u: intra mode", bi);
            }
            for (int k = 0; k < 2; k++) {
                const int m = k ? b.mode_uv : b.mode_y;
                if (b.angle_delta[k] < -3 || b.angle_delta[k] > 3 || (b.angle_delta[k] && !is_directional(m)))
                    return ea1::fail(EA1_EINVAL, "block %u: angle delta %d with mode %d", bi, b.angle_delta[k], m);
            }
            if (b.mode_uv == EA1_UV_CFL_PRED && (bw > 32 || bh > 32 || b.cfl_alpha[0] < -16 || b.cfl_alpha[0] > 16 || b.cfl_alpha[1] < -16 || b.cfl_alpha[1] > 16))
                return ea1::fail(EA1_EINVAL, "block %u: chroma-from-luma needs a block of at most 32x32 and alpha in [-16, 16]", bi);
            if (b.filter_intra && (b.filter_intra > 5 || bw > 32 || bh > 32 || b.mode_y != EA1_DC_PRED || (b.flags & EA1_BLK_PALETTE_Y)))
                return ea1::fail(EA1_EINVAL, "block %u: filter intra needs mode_y DC, a block of at most 32x32 and no palette", bi);
            const bool py = b.flags & EA1_BLK_PALETTE_Y, puv = b.flags & EA1_BLK_PALETTE_UV;
            if ((py && (b.pal_size[0] < 2 || b.pal_size[0] > 8 || b.mode_y != EA1_DC_PRED)) || (puv && (b.pal_size[1] < 2 || b.pal_size[1] > 8 || b.mode_uv != EA1_DC_PRED)) ||
                (!py && b.pal_size[0]) || (!puv && b.pal_size[1]))
                return ea1::fail(EA1_EINVAL, "block %u: palette sizes / modes", bi);
            if (py || puv) {
                const Rect prc = cg.rect(b, 1);
                const uint64_t carea_ = uint64_t(prc.w) * prc.h;   // the chroma block (of the whole 8-sample area for a 4-sample side in a subsampled direction)
                const uint64_t need = uint64_t(b.pal_size[0]) + 2 * b.pal_size[1] + uint64_t(bw) * bh + (puv ? carea_ : 0);
                if (uint64_t(b.aux_off) + need > desc->num_aux) return ea1::fail(EA1_EINVAL, "block %u: palette data range", bi);
                const uint8_t *mp = aux + b.aux_off + b.pal_size[0] + 2 * b.pal_size[1];
                for (int i = 0; i < bw * bh; i++)
                    if (py ? mp[i] >= b.pal_size[0] : false) return ea1::fail(EA1_EINVAL, "block %u: luma colour index %u", bi, mp[i]);
                if (puv && !cg.has_chroma(b)) return ea1::fail(EA1_EINVAL, "block %u: a chroma palette on a block that owns no chroma", bi);
                for (int i = 0; puv && i < int(carea_); i++)
                    if (mp[bw * bh + i] >= b.pal_size[1]) return ea1::fail(EA1_EINVAL, "block %u: chroma colour index", bi);
            }
        }
        if (desc->lf_per_block && (b.lf_level[0] > 63 || b.lf_level[1] > 63 || b.lf_level[2] > 63 || b.lf_level[3] > 63))
            return ea1::fail(EA1_EINVAL, "block %u: deblock level above 63", bi);
        if (uint64_t(b.first_tx) + b.n_tx > ntx) return ea1::fail(EA1_EINVAL, "block %u: transform-block range", bi);
        uint64_t area[3] = {0, 0, 0};
        // units of the block's rectangle covered so far, per plane: 32 rows of 32 bits (a 128x128 block has 32 x 32 units)
        uint64_t cover[3][16];
        const Rect rect_y = cg.rect(b, 0), rect_c = cg.rect(b, 1);
        const bool chroma_owner = cg.has_chroma(b);
        // U and V share one transform grid (only U's is derived for the filters): the same set of (position, size)
        static thread_local uint64_t grid_uv[2][1024];   // (a lossless 128x128 block: 16 x 16 chroma transform blocks of 4x4; 4:4:4: 32 x 32)
        int n_uv[2] = {0, 0};
        bool plane_has[3] = {false, false, false};
        const bool interintra = (b.flags & EA1_BLK_INTER) && (b.flags & EA1_BLK_INTERINTRA);
        // the usual shape -- ONE transform block of the block's own size per plane, in plane order -- is recognised up front:
        // its geometry (position, alignment, tiling, the shared U / V grid) then holds by construction
        bool plain = b.n_tx == (chroma_owner ? 3u : 1u);
        if (plain) {
            const ea1_txblock *t3 = txs + b.first_tx;
            plain = t3[0].plane == 0 && t3[0].size == (b.bwl | (b.bhl << 4)) && t3[0].x4 == b.x4 && t3[0].y4 == b.y4;
            if (plain && chroma_owner) {
                const unsigned sc = unsigned(ssx ? std::max(int(b.bwl), 3) - 1 : int(b.bwl)) | (unsigned(ssy ? std::max(int(b.bhl), 3) - 1 : int(b.bhl)) << 4);
                const unsigned cx4 = unsigned(rect_c.x) / 4, cy4 = unsigned(rect_c.y) / 4;
                plain = t3[1].plane == 1 && t3[2].plane == 2 && t3[1].size == sc && t3[2].size == sc && t3[1].x4 == cx4 && t3[1].y4 == cy4 &&
                        t3[2].x4 == cx4 && t3[2].y4 == cy4;
            }
        }
        if (!plain) memset(cover, 0, sizeof cover);
        for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++) {
            const ea1_txblock &t = txs[ti];
            const int cls = t.plane > 2 ? -1 : tt.cls[t.size];
            if (cls < 0) return ea1::fail(EA1_EINVAL, "transform block %u: plane / size", ti);
            const int wl = t.size & 15, hl = t.size >> 4;
            const int tw = 1 << wl, th = 1 << hl, tx_ = t.x4 * 4, ty = t.y4 * 4;
            const Rect &br = t.plane ? rect_c : rect_y;
            const int bx = br.x, by = br.y, pbw = br.w, pbh = br.h;
            if (!plain) {
                if (t.plane && !chroma_owner) return ea1::fail(EA1_EINVAL, "transform block %u: block %u owns no chroma", ti, bi);
                if ((tx_ & (tw - 1)) || (ty & (th - 1)) || tx_ < bx || ty < by || tx_ + tw > bx + pbw || ty + th > by + pbh)
                    return ea1::fail(EA1_EINVAL, "transform block %u: %dx%d at (%d,%d) leaves block %u or is misaligned", ti, tw, th, tx_, ty, bi);
            }
            if (!tt.ok[cls][t.txtp]) return ea1::fail(EA1_EINVAL, "transform block %u: type %u invalid for size %dx%d", ti, t.txtp, tw, th);
            if (t.coef_off != EA1_NO_COEFS) {
                const uint32_t cn = uint32_t((tw < 32 ? tw : 32) * (th < 32 ? th : 32));
                uint32_t stored = cn;   // entries of the arena the block owns
                int bucket = (t.nz_log2 >= 2 && t.nz_log2 <= 5) ? t.nz_log2 - 2 : 0;
                if (diagonal_layout) {
                    const int sw = std::min(tw, 32), sh = std::min(th, 32), D = t.nz_log2 >> 2, n = 4 << (t.nz_log2 & 3);
                    if (D > sw + sh - 2) return ea1::fail(EA1_EINVAL, "transform block %u: last diagonal %d of a %dx%d block", ti, D, tw, th);
                    stored = uint32_t(ea1::diag_before(sw, sh, D + 1));
                    bucket = t.nz_log2 & 3;
                    if (in_strip) expanded_entries += uint64_t(std::min(sw, n) * std::min(sh, n));
                } else if (corner_layout) {
                    if (t.nz_log2 < 2 || t.nz_log2 > 5) return ea1::fail(EA1_EINVAL, "transform block %u: the corner layout needs nz_log2 in 2..5, not %u", ti, t.nz_log2);
                    stored = uint32_t(std::min(std::min(tw, 32), 1 << t.nz_log2) * std::min(std::min(th, 32), 1 << t.nz_log2));
                }
                if ((t.coef_off & 1) || uint64_t(t.coef_off) + stored > desc->num_coefs) return ea1::fail(EA1_EINVAL, "transform block %u: coefficient range", ti);
                coef_count += cn;
                plane_has[t.plane] = true;
                if (in_strip) {
                    tx_bucket[cls][bucket]++;
                    coef_lo = std::min(coef_lo, t.coef_off);
                    coef_hi = std::max(coef_hi, t.coef_off + stored);
                }
            }
            if (plain) {
                if (!t.plane && (t.y4 & 1)) luma_odd_row_edges = true;
                continue;
            }
            if (tw != pbw || th != pbh) {   // (a transform block of the block's own size covers it alone: the area check below sees any other)
                const int cx = (tx_ - bx) / 4, cy = (ty - by) / 4, nw4 = tw / 4, nh4 = th / 4;
                const uint64_t row_bits = ((1ull << nw4) - 1) << cx;   // (nw4 <= 16, cx + nw4 <= 32)
                for (int j = cy; j < cy + nh4; j++) {
                    uint64_t &wd = cover[t.plane][j >> 1];
                    const uint64_t m = row_bits << (32 * (j & 1));
                    if (wd & m) return ea1::fail(EA1_EINVAL, "transform block %u overlaps another transform block of block %u", ti, bi);
                    wd |= m;
                }
            }
            if (t.plane) {
                int &n = n_uv[t.plane - 1];
                if (n >= 1024) return ea1::fail(EA1_EINVAL, "block %u: more chroma transform blocks than chroma units", bi);
                grid_uv[t.plane - 1][n++] = uint64_t(t.x4) | (uint64_t(t.y4) << 16) | (uint64_t(t.size) << 32);
            } else if (t.y4 & 1) {
                luma_odd_row_edges = true;
            }
            area[t.plane] += uint64_t(tw) * th;
        }
        const uint64_t carea = chroma_owner ? uint64_t(rect_c.w) * rect_c.h : 0;
        if (!plain && (area[0] != uint64_t(bw) * bh || area[1] != carea || area[2] != carea))
            return ea1::fail(EA1_EINVAL, "block %u: its transform blocks do not tile it", bi);
        if (n_uv[0] != n_uv[1]) return ea1::fail(EA1_EINVAL, "block %u: U and V transform grids differ", bi);
        for (int i = 0; i < n_uv[0]; i++) {
            bool found = grid_uv[1][i] == grid_uv[0][i];   // the usual order: U and V listed alike
            for (int k = 0; !found && k < n_uv[1]; k++) found = grid_uv[1][k] == grid_uv[0][i];
            if (!found) return ea1::fail(EA1_EINVAL, "block %u: U and V transform grids differ", bi);
        }
        if (interintra && in_strip)   // transform blocks that only clear the residual plane (see derive_items_kernel)
            for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++)
                if (txs[ti].coef_off == EA1_NO_COEFS && plane_has[txs[ti].plane]) tx_zero_fill[tt.cls[txs[ti].size]]++;
        // the block itself: no unit taken twice (holes are caught by the count below)
        {
            const int n4w = bw / 4;
            const uint64_t m = ((1ull << n4w) - 1) << (b.x4 & 63);   // (n4w <= 32) aligned to its width: inside one word
            uint64_t *rw = done + size_t(b.y4) * dw + (b.x4 >> 6);
            for (int j = 0; j < bh / 4; j++, rw += dw) {
                if (*rw & m) return ea1::fail(EA1_EINVAL, "block %u overlaps an earlier block", bi);
                *rw |= m;
            }
        }
        // the block map serves neighbour look-ups only (OBMC, smooth-edge, the chroma of 4-sample blocks): all of them read
        // the unit just above a block's top row or just left of its left column, or units of blocks at most two units
        // wide or high -- a block's bottom row and right column of units are all that is ever read
        {
            uint32_t *const m0 = map.data() + size_t(b.y4) * w4 + b.x4;
            const int n4w = bw / 4, n4h = bh / 4;
            for (int j = 0; j < n4h - 1; j++) m0[size_t(j) * w4 + n4w - 1] = bi;
            std::fill_n(m0 + size_t(n4h - 1) * w4, size_t(n4w), bi);
        }
        covered4 += uint64_t(bw / 4) * (bh / 4);
    }
    if (covered4 != uint64_t(w4) * h4) return ea1::fail(EA1_EINVAL, "the blocks cover %llu 4x4 units, the frame has %llu: blocks overlap or leave holes", (unsigned long long)covered4, (unsigned long long)(uint64_t(w4) * h4));
    // (no unit is taken twice and the count is the frame's: every unit is covered, and with it -- transform blocks
    // tile their block in every plane -- every transform unit)
    if (desc->cdef_enable && !cdef_idx) return ea1::fail(EA1_EINVAL, "ea1_job_create: cdef_idx is NULL");
    if (cdef_idx && lengths && uint64_t(lengths->num_sb) < uint64_t(sb_cols) * sb_rows)
        return ea1::fail(EA1_EINVAL, "parsed frame: %u CDEF indices for %d superblocks of 64x64", lengths->num_sb, sb_cols * sb_rows);
    if (desc->cdef_enable) {
        for (int i = 0; i < sb_cols * sb_rows; i++)
            if (cdef_idx[i] > 7 && cdef_idx[i] != 0xFF) return ea1::fail(EA1_EINVAL, "ea1_job_create: cdef_idx[%d] = %u", i, cdef_idx[i]);
        if (desc->cdef_damping < 3 || desc->cdef_damping > 6) return ea1::fail(EA1_EINVAL, "ea1_job_create: cdef_damping %u", desc->cdef_damping);
    }
    for (int k = 0; k < 4; k++)
        if (desc->lf_level[k] > 63) return ea1::fail(EA1_EINVAL, "ea1_job_create: lf_level[%d] = %u", k, desc->lf_level[k]);
    if (desc->lf_sharpness > 7) return ea1::fail(EA1_EINVAL, "ea1_job_create: lf_sharpness %u", desc->lf_sharpness);
    if ((desc->crop & 15) > 7 || (desc->crop >> 4) > 7) return ea1::fail(EA1_EINVAL, "ea1_job_create: crop 0x%02x (0..7 samples per direction)", desc->crop);
    if (8 * (desc->pad & 15) >= W || 8 * (desc->pad >> 4) >= H)
        return ea1::fail(EA1_EINVAL, "ea1_job_create: pad 0x%02x (0..15 units of 8 samples per direction, less than the frame)", desc->pad);

    t_valid = now();
    // ---- work lists the host makes: the intra wavefront, OBMC strips, warped / sub-8 predictions -------------
    // (transform items and inter tiles -- the bulk -- are made on the device from the uploaded records)
    std::vector<ea1::McItem> *mc = hs.mc, *mc_ext = hs.mc_ext;   // host part: EA1_BLK_WARP blocks whose warp is unusable
    std::vector<ea1::ObmcItem> *obmc = hs.obmc;
    std::vector<ea1::PieceItem> &pieces = hs.pieces;
    std::vector<ea1::WarpItem> &warps = hs.warps, &warps_chroma = hs.warps_chroma;
    std::vector<LevelItem> &intra = hs.intra;
    std::vector<ea1::PalRef> &pal_refs = hs.pal_refs;
    for (int k = 0; k < 3; k++) {
        mc[k].clear();
        mc_ext[k].clear();
    }
    obmc[0].clear();
    obmc[1].clear();
    pieces.clear();
    warps.clear();
    warps_chroma.clear();
    intra.clear();
    pal_refs.clear();
    WarpSetup wsu[2];
    JobStats acc;
    memset(&acc, 0, sizeof acc);
    std::vector<uint16_t> *lvl = hs.lvl;
    if (!hs.lvl_clean || lvl[0].size() != size_t(w4) * h4 || lvl[1].size() != size_t(cw4) * ch4) {
        lvl[0].assign(size_t(w4) * h4, 0);
        lvl[1].assign(size_t(cw4) * ch4, 0);
        lvl[2].assign(size_t(cw4) * ch4, 0);
    }
    hs.lvl_clean = false;
    hs.touched.clear();
    // an intra frame touches the whole map: it is cleared as a whole afterwards instead of area by area (387 000 areas of a 4K key frame)
    const bool dense_levels = !desc->is_inter;
    auto set_level = [&](int pl, int ux0, int uy0, int nw, int nh, uint32_t lv) {
        const int pw4 = pl ? cw4 : w4;
        for (int j = 0; j < nh; j++) std::fill_n(lvl[pl].begin() + (size_t(uy0 + j) * pw4 + ux0), size_t(nw), uint16_t(lv));
        if (!dense_levels) hs.touched.push_back(HostLists::Touched{uint32_t(pl), uint32_t(ux0), uint32_t(uy0), uint32_t(nw), uint32_t(nh)});
    };
    uint32_t max_level = 0;
    const bool whole = strip_x0 <= 0 && strip_x1 >= W;
    if (whole) {
        coef_lo = 0;
        coef_hi = desc->num_coefs;
    } else if (coef_hi <= coef_lo) {
        coef_lo = coef_hi = 0;
    }
    for (const uint32_t bi : hs.host_blocks) {   // in list (= decoding) order; the strip's own blocks only
        const ea1_block &b = blocks[bi];
        const int bw = 1 << b.bwl, bh = 1 << b.bhl, x = b.x4 * 4, y = b.y4 * 4;
        const bool inter = b.flags & EA1_BLK_INTER;
        const bool interintra = inter && (b.flags & EA1_BLK_INTERINTRA);
        if (inter && is_sub8(b)) {
            // AV1 spec 7.11.3.1: luma as one small prediction; chroma (if this block owns it) as one
            // block with this block's motion when the 8x8 luma area contains an intra block, else
            // piecewise, every block of the area predicting its (w/2 x h/2) part with its own motion
            auto piece = [&](const ea1_block &src, int pl, int px, int py, int pw_, int ph_) {
                ea1::PieceItem pi;
                memset(&pi, 0, sizeof pi);
                pi.x = uint16_t(px);
                pi.y = uint16_t(py);
                pi.w = uint8_t(pw_);
                pi.h = uint8_t(ph_);
                pi.plane = uint8_t(pl);
                pi.filt = src.mode_y;
                pi.ref = src.ref[0];
                pi.mv[0] = src.mv[0][0];
                pi.mv[1] = src.mv[0][1];
                pieces.push_back(pi);
                acc.piece_px += uint64_t(pw_) * ph_;
            };
            piece(b, 0, x, y, bw, bh);
            if (cg.has_chroma(b) && !cg.sub8(b)) {
                // (4:4:4, or the 4-sample side lies in a direction that is not subsampled): the chroma block is the block's own
                const Rect cr = cg.rect(b, 1);
                for (int pl = 1; pl < 3; pl++) piece(b, pl, cr.x, cr.y, cr.w, cr.h);   // (at most 64 samples, like the luma block)
            } else if (cg.has_chroma(b)) {
                const Rect cr = cg.rect(b, 1);
                const int rx4 = ssx ? b.x4 & ~1 : b.x4, ry4 = ssy ? b.y4 & ~1 : b.y4, n4w = (cr.w << ssx) / 4, n4h = (cr.h << ssy) / 4;
                bool some_intra = false;
                for (int r = 0; r < n4h; r++)
                    for (int c = 0; c < n4w; c++) some_intra |= !(blocks[map[size_t(ry4 + r) * w4 + rx4 + c]].flags & EA1_BLK_INTER);
                const int pw_ = some_intra ? cr.w : bw >> ssx, ph_ = some_intra ? cr.h : bh >> ssy;
                for (int pl = 1; pl < 3; pl++)
                    for (int py = 0, r = 0; py < cr.h; py += ph_, r++)
                        for (int px = 0, c = 0; px < cr.w; px += pw_, c++)
                            piece(some_intra ? b : blocks[map[size_t(ry4 + r) * w4 + rx4 + c]], pl, cr.x + px, cr.y + py, pw_, ph_);
            }
            ref_pixels += 0;
        } else if (inter && (b.flags & EA1_BLK_WARP) && warp_setup(b, aux, wsu)) {
            // AV1 spec 7.11.3.1 / 7.11.3.5: per plane, the block is warped in 8x8 units when it is at
            // least 8x8 there; each unit gets the affine map evaluated at its centre.  Everything of
            // the block (translated chroma of small blocks, the translated prediction of a compound
            // pair) goes through the warp list so that the pair is combined in one place.
            const int nref = (b.flags & EA1_BLK_COMPOUND) ? 2 : 1;
            for (int k = 0; k < nref; k++)
                if (wsu[k].valid) acc.warp_ref_mask |= 1u << b.ref[k];   // the references a prediction really warps from
            for (int pl = 0; pl < 3; pl++) {
                const int sx_ = pl ? ssx : 0, sy_ = pl ? ssy : 0, pw_ = bw >> sx_, ph_ = bh >> sy_, px0 = x >> sx_, py0 = y >> sy_;
                const bool plane_warp = pw_ >= 8 && ph_ >= 8;
                const int uw = std::min(pw_, 8), uh = std::min(ph_, 8);
                for (int uy = 0; uy < ph_; uy += uh)
                    for (int ux = 0; ux < pw_; ux += uw) {
                        ea1::WarpItem wi;
                        memset(&wi, 0, sizeof wi);
                        wi.x = uint16_t(px0 + ux);
                        wi.y = uint16_t(py0 + uy);
                        wi.plane = uint8_t(pl);
                        wi.w = uint8_t(uw);
                        wi.h = uint8_t(uh);
                        wi.bx = uint8_t(ux);
                        wi.by = uint8_t(uy);
                        wi.ref[0] = b.ref[0];
                        wi.ref[1] = b.ref[1];
                        wi.fset = uint8_t(mc_filter_set(b.mode_y & 3, pw_) | (mc_filter_set((b.mode_y >> 2) & 3, ph_) << 4));
                        wi.comp_type = (b.flags & EA1_BLK_COMPOUND) ? b.comp_type : uint8_t(EA1_COMP_AVG);
                        wi.comp_arg = b.comp_arg;
                        if (nref == 2 && b.comp_type == EA1_COMP_WEDGE) {
                            wi.mask_stride_log2 = b.bwl;
                            wi.mask_off = ea1::wedge_mask_offset(b.bwl, b.bhl, (b.comp_arg >> 4) & 1, b.comp_arg & 15) + uint32_t(((uy << sy_) * bw) + (ux << sx_));
                        }
                        wi.flags = uint8_t(nref == 2 ? 1 : 0);
                        for (int k = 0; k < nref; k++) {
                            ea1::WarpRef &wr = wi.r[k];
                            wr.mv[0] = b.mv[k][0];
                            wr.mv[1] = b.mv[k][1];
                            if (!plane_warp || !wsu[k].valid) continue;
                            wi.flags |= uint8_t(2 << k);
                            const int32_t *m = wsu[k].mat;
                            const int64_t sx = int64_t(px0 + ux + 4) << sx_, sy = int64_t(py0 + uy + 4) << sy_;
                            const int64_t x4 = (m[2] * sx + m[3] * sy + m[0]) >> sx_, y4 = (m[4] * sx + m[5] * sy + m[1]) >> sy_;
                            wr.ix4 = int32_t(x4 >> 16);
                            wr.sx4 = uint16_t(x4 & 0xFFFF);
                            wr.iy4 = int32_t(y4 >> 16);
                            wr.sy4 = uint16_t(y4 & 0xFFFF);
                            wr.alpha = int16_t(wsu[k].sh[0]);
                            wr.beta = int16_t(wsu[k].sh[1]);
                            wr.gamma = int16_t(wsu[k].sh[2]);
                            wr.delta = int16_t(wsu[k].sh[3]);
                        }
                        (pl ? warps_chroma : warps).push_back(wi);
                        acc.warp_px += uint64_t(uw) * uh;
                    }
            }
        } else if (inter && (b.flags & EA1_BLK_WARP)) {   // flagged for warped motion, warp unusable: translated (the other blocks' tiles are made on the device)
            ea1::McItem mi;
            memset(&mi, 0, sizeof mi);
            mi.flags = uint8_t(((b.flags & EA1_BLK_COMPOUND) ? 2 : 0) | ((bw >> ssx) == 4 ? 4 : 0) | ((bh >> ssy) == 4 ? 8 : 0));
            mi.filt = b.mode_y;
            mi.ref[0] = b.ref[0];
            mi.ref[1] = b.ref[1];
            memcpy(mi.mv, b.mv, sizeof mi.mv);
            const bool ext = (b.flags & EA1_BLK_COMPOUND) && b.comp_type != EA1_COMP_AVG;
            mi.comp_type = ext ? b.comp_type : uint8_t(EA1_COMP_AVG);
            mi.comp_arg = uint8_t(b.comp_type == EA1_COMP_WEDGE ? 0 : b.comp_arg);
            mi.mask_stride_log2 = b.bwl;
            const int mn = std::min(bw, bh);
            const int ts = mn < 32 ? mn : 32;            // tile size
            const int cls = ts == 8 ? 0 : (ts == 16 ? 1 : 2);
            for (int ty = 0; ty < bh; ty += ts)
                for (int tx_ = 0; tx_ < bw; tx_ += ts) {
                    mi.x8 = uint16_t((x + tx_) / 8);
                    mi.y8 = uint16_t((y + ty) / 8);
                    mi.ox8 = uint8_t(tx_ / 8);
                    mi.oy8 = uint8_t(ty / 8);
                    if (ext && b.comp_type == EA1_COMP_WEDGE)
                        mi.mask_off = ea1::wedge_mask_offset(b.bwl, b.bhl, (b.comp_arg >> 4) & 1, b.comp_arg & 15) + uint32_t(ty * bw + tx_);
                    (ext ? mc_ext : mc)[cls].push_back(mi);
                }
        }
        if (inter && (b.flags & EA1_BLK_OBMC)) {
            // AV1 spec 7.11.3.10: up to min(4, log2(size / 4)) inter neighbours per side, visited in
            // steps of the neighbour's size (at least 8, at most 64 samples)
            for (int pass = 0; pass < 2; pass++) {
                if (!(b.flags & (pass == 0 ? EA1_BLK_HAVE_TOP : EA1_BLK_HAVE_LEFT))) continue;
                const int len4 = (pass == 0 ? bw : bh) / 4, l2 = (pass == 0 ? b.bwl : b.bhl) - 2;
                const int limit = std::min(l2, 4), start4 = pass == 0 ? b.x4 : b.y4;
                const int end4 = std::min(start4 + len4, (pass == 0 ? W - 8 * (desc->pad & 15) : H - 8 * (desc->pad >> 4)) / 4);   // neighbours inside the TRUE coded frame
                int count = 0;
                for (int p4 = start4; count < limit && p4 < end4;) {
                    const int cx4 = pass == 0 ? (p4 | 1) : b.x4 - 1, cy4 = pass == 0 ? b.y4 - 1 : (p4 | 1);
                    const ea1_block &c = blocks[map[size_t(cy4) * w4 + cx4]];
                    const int step4 = std::min(std::max((1 << (pass == 0 ? c.bwl : c.bhl)) / 4, 2), 16);
                    if (c.flags & EA1_BLK_INTER) {
                        count++;
                        ea1::ObmcItem oi;
                        memset(&oi, 0, sizeof oi);
                        if (pass == 0) {
                            oi.x = uint16_t(p4 * 4);
                            oi.y = uint16_t(y);
                            oi.w = uint8_t(std::min(bw, step4 * 4));
                            oi.h = uint8_t(std::min(bh >> 1, 32));
                        } else {
                            oi.x = uint16_t(x);
                            oi.y = uint16_t(p4 * 4);
                            oi.w = uint8_t(std::min(bw >> 1, 32));
                            oi.h = uint8_t(std::min(bh, step4 * 4));
                        }
                        // chroma takes part in the pass from ABOVE when the block's chroma is at least 8x8 in block-size
                        // order (which includes 4x16 and 16x4); the pass from the LEFT always has it (7.11.3.10: the size test
                        // guards the first pass only -- libaom av1_skip_u4x4_pred_in_obmc, dav1d obmc(); round 4)
                        const int cw = bw >> ssx, chh = bh >> ssy;
                        oi.flags = uint8_t((pass == 1 || (cw >= 8 && chh >= 8) || (cw == 4 && chh == 16) || (cw == 16 && chh == 4)) ? 1 : 0);
                        oi.filt = c.mode_y;
                        oi.ref = c.ref[0];
                        oi.mv[0] = c.mv[0][0];
                        oi.mv[1] = c.mv[0][1];
                        obmc[pass].push_back(oi);
                        acc.obmc_px[pass] += uint64_t(oi.w) * oi.h * ((oi.flags & 1) ? 3 : 2) / 2;
                    }
                    p4 += step4;
                }
            }
        }
        if (interintra) {
            // one wavefront item per plane (and row part): intra prediction of the WHOLE block from
            // its reconstructed neighbours, blended into the inter prediction; residual afterwards
            static const uint8_t ii_to_intra[4] = {EA1_DC_PRED, EA1_V_PRED, EA1_H_PRED, EA1_SMOOTH_PRED};
            acc.intra_special = true;
            acc.interintra_blocks++;
            for (int pl = 0; pl < 3; pl++) {
                const int sx_ = pl ? ssx : 0, sy_ = pl ? ssy : 0, pw_ = bw >> sx_, ph_ = bh >> sy_, px_ = x >> sx_, py_ = y >> sy_;
                const int pw4 = pl ? cw4 : w4, ph4 = pl ? ch4 : h4;
                const bool have_left = b.flags & EA1_BLK_HAVE_LEFT, have_top = b.flags & EA1_BLK_HAVE_TOP;
                uint32_t lv = 0;
                auto dep = [&](int ux, int uy) {
                    if (ux < 0 || uy < 0 || ux >= pw4 || uy >= ph4) return;
                    lv = std::max<uint32_t>(lv, lvl[pl][size_t(uy) * pw4 + ux]);
                };
                const int ux0 = px_ / 4, uy0 = py_ / 4, nw = pw_ / 4, nh = ph_ / 4;
                if (have_left)
                    for (int k = have_top ? -1 : 0; k < nh; k++) dep(ux0 - 1, uy0 + k);
                if (have_top)
                    for (int k = 0; k < nw; k++) dep(ux0 + k, uy0 - 1);
                lv += 1;
                max_level = std::max(max_level, lv);
                set_level(pl, ux0, uy0, nw, nh, lv);
                bool any = false;
                for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++) any |= txs[ti].plane == pl && txs[ti].coef_off != EA1_NO_COEFS;
                ea1::IntraItem ii;
                memset(&ii, 0, sizeof ii);
                ii.x = uint16_t(px_);
                ii.y = uint16_t(py_);
                ii.plane = uint8_t(pl);
                ii.size = uint8_t((b.bwl - sx_) | ((b.bhl - sy_) << 4));
                ii.mode = ii_to_intra[b.ii_mode & 3];
                ii.flags = uint8_t((have_left ? ea1::kIIHaveLeft : 0) | (have_top ? ea1::kIIHaveTop : 0) | (any ? ea1::kIIResid : 0));
                ii.ii = uint8_t(((b.ii_mode & 3) + 1) | (b.ii_mode & EA1_II_WEDGE));
                if (b.ii_mode & EA1_II_WEDGE) {
                    ii.aux = ea1::wedge_mask_offset(b.bwl, b.bhl, 0, b.ii_arg & 15);
                    ii.alpha = int8_t(b.bwl);
                }
                const int rpp = rows_per_part(pw_);
                for (int part = 0; part * rpp < ph_; part++) {
                    ii.part = uint8_t(part);
                    intra.push_back(LevelItem{lv, ii});
                }
                acc.intra_pixels += uint64_t(pw_) * ph_;
                if (any) acc.intra_resid_pixels += uint64_t(pw_) * ph_;
            }
        }
        bool smooth_edge[2] = {false, false};   // luma / chroma: get_filter_type() of the spec
        if (!inter) {
            for (int k = 0; k < 2; k++) {   // k = 0 luma, 1 chroma (from the chroma block's position)
                const int ux = (k && ssx) ? (b.x4 & ~1) : b.x4, uy = (k && ssy) ? (b.y4 & ~1) : b.y4;
                const unsigned ef = cg.edge_flags(b, k);
                const ea1_block *nbk[2] = {nullptr, nullptr};
                if (ef & EA1_BLK_HAVE_TOP) nbk[0] = &blocks[map[size_t(uy - 1) * w4 + ((k && ssx) ? ux + 1 : ux)]];
                if (ef & EA1_BLK_HAVE_LEFT) nbk[1] = &blocks[map[size_t((k && ssy) ? uy + 1 : uy) * w4 + ux - 1]];
                for (const ea1_block *n : nbk) {
                    if (!n || (n->flags & EA1_BLK_INTER)) continue;
                    smooth_edge[k] |= is_smooth_mode(k ? n->mode_uv : n->mode_y);
                }
            }
        }
        for (uint32_t ti = b.first_tx; !inter && ti < b.first_tx + b.n_tx; ti++) {
            const ea1_txblock &t = txs[ti];
            const int pl = t.plane;
            const int tw = 1 << (t.size & 15), th = 1 << (t.size >> 4);
            const int tx_ = t.x4 * 4, ty = t.y4 * 4;
            const bool has = t.coef_off != EA1_NO_COEFS;
            // ---- intra wavefront item: level = 1 + deepest intra item among the samples it reads
            const Rect br = cg.rect(b, pl);
            const int bx = br.x, by = br.y, pbw = br.w, pbh = br.h;
            const unsigned ef = cg.edge_flags(b, pl);
            const int pw4 = pl ? cw4 : w4, ph4 = pl ? ch4 : h4;
            const bool have_left = (ef & EA1_BLK_HAVE_LEFT) || tx_ > bx, have_top = (ef & EA1_BLK_HAVE_TOP) || ty > by;
            const bool last_col = tx_ + tw >= bx + pbw, first_row = ty == by, first_col = tx_ == bx, last_row = ty + th >= by + pbh;
            // transform blocks are decoded in raster order inside a 64x64 luma chunk of the block (32x32 in chroma), the chunks of a
            // 128-sample block in raster order: across a chunk boundary the sample up and to the right is decoded iff it lies in the
            // chunk row above, the sample down and to the left iff it lies in the same chunk row
            const int cmx = (pl ? 64 >> ssx : 64) - 1, cmy = (pl ? 64 >> ssy : 64) - 1;   // (offsets inside the block: never negative)
            const bool chunk_last_col = ((tx_ + tw - bx) & cmx) == 0, chunk_first_row = ((ty - by) & cmy) == 0;
            const bool chunk_first_col = ((tx_ - bx) & cmx) == 0, chunk_last_row = ((ty + th - by) & cmy) == 0;
            const bool have_tr = first_row ? (last_col ? (ef & EA1_BLK_HAVE_TOP_RIGHT) != 0 : true) : (last_col ? false : (chunk_last_col ? chunk_first_row : true));
            const bool have_bl = first_col ? (last_row ? (ef & EA1_BLK_HAVE_BOTTOM_LEFT) != 0 : true) : (last_row ? false : (chunk_first_col ? !chunk_last_row : false));
            int mode = pl ? b.mode_uv : b.mode_y;
            const bool ibc = mode == EA1_INTRABC_PRED;
            const int p_angle = is_directional(mode) ? kModeAngle[mode] + 3 * b.angle_delta[pl ? 1 : 0] : 0;
            const bool palette = pl ? (b.flags & EA1_BLK_PALETTE_UV) != 0 : (b.flags & EA1_BLK_PALETTE_Y) != 0;
            if (pl == 0 && b.filter_intra) mode = ea1::kIntraFilter + b.filter_intra - 1;
            if (palette) mode = ea1::kIntraPalette;
            if (ibc) mode = ea1::kIntraBC;
            uint32_t lv = 0;
            const std::vector<uint16_t> &lm = lvl[pl];
            auto dep = [&](int ux, int uy) {
                if (ux < 0 || uy < 0 || ux >= pw4 || uy >= ph4) return;
                lv = std::max<uint32_t>(lv, lm[size_t(uy) * pw4 + ux]);
            };
            const int ux0 = tx_ / 4, uy0 = ty / 4, nw = tw / 4, nh = th / 4;
            if (ibc) {
                // the copy reads a tw x th rectangle (one more column / row at a half-sample chroma
                // position) of this plane, displaced by the block vector
                const int px = (tx_ << 4) + ((pl && ssx) ? b.mv[0][1] : b.mv[0][1] * 2), py = (ty << 4) + ((pl && ssy) ? b.mv[0][0] : b.mv[0][0] * 2);
                const int sx0 = px >> 4, sy0 = py >> 4, sx1 = sx0 + tw - 1 + ((px & 15) ? 1 : 0), sy1 = sy0 + th - 1 + ((py & 15) ? 1 : 0);
                for (int uy = sy0 / 4; uy <= sy1 / 4; uy++)
                    for (int ux = sx0 / 4; ux <= sx1 / 4; ux++) dep(ux, uy);
            }
            if (have_left && !palette && !ibc) {
                const int ext = (have_bl && p_angle > 180) ? std::min(nw, nh) : 0;
                for (int k = have_top ? -1 : 0; k < nh + ext; k++) dep(ux0 - 1, uy0 + k);
            }
            if (have_top && !palette && !ibc) {
                const int ext = (have_tr && p_angle > 0 && p_angle < 90) ? std::min(nw, nh) : 0;
                for (int k = 0; k < nw + ext; k++) dep(ux0 + k, uy0 - 1);
            }
            if (pl && mode == EA1_UV_CFL_PRED)   // reads the reconstructed luma of its own area
                for (int j = 0; j < (nh << ssy); j++)
                    for (int i = 0; i < (nw << ssx); i++) lv = std::max<uint32_t>(lv, lvl[0][size_t((uy0 << ssy) + j) * w4 + (ux0 << ssx) + i]);
            lv += 1;
            max_level = std::max(max_level, lv);
            set_level(pl, ux0, uy0, nw, nh, lv);
            ea1::IntraItem ii;
            memset(&ii, 0, sizeof ii);
            ii.x = uint16_t(tx_);
            ii.y = uint16_t(ty);
            ii.plane = uint8_t(pl);
            ii.size = t.size;
            ii.mode = uint8_t(mode);
            ii.flags = uint8_t((have_left ? ea1::kIIHaveLeft : 0) | (have_top ? ea1::kIIHaveTop : 0) | (has ? ea1::kIIResid : 0) |
                               (have_tr ? ea1::kIIHaveTopRight : 0) | (have_bl ? ea1::kIIHaveBottomLeft : 0) |
                               (smooth_edge[pl ? 1 : 0] ? ea1::kIISmoothEdge : 0) | (desc->intra_edge_filter ? ea1::kIIEdgeFilter : 0));
            ii.angle_delta = b.angle_delta[pl ? 1 : 0];
            ii.alpha = pl ? b.cfl_alpha[pl - 1] : 0;
            if (pl && mode == EA1_UV_CFL_PRED && desc->pad) {
                // MaxLumaW / MaxLumaH (AV1 spec 5.11.35 / 7.11.5) of a block that reaches beyond the coded frame: its luma transform blocks
                // that start out there are not coded, chroma-from-luma repeats the last coded luma column / row.  aux = limits in luma
                // samples (0 = none)
                const int cw_ = W - 8 * (desc->pad & 15), ch_ = H - 8 * (desc->pad >> 4);
                const ea1_txblock &l0 = txs[b.first_tx];   // the luma grid of an intra block is uniform, its first entry is luma
                const int ltw = 1 << (l0.size & 15), lth = 1 << (l0.size >> 4);
                int mlw = x + bw, mlh = y + bh;
                if (cw_ - x < bw) mlw = x + (cw_ - x + ltw - 1) / ltw * ltw;
                if (ch_ - y < bh) mlh = y + (ch_ - y + lth - 1) / lth * lth;
                if (mlw < x + bw || mlh < y + bh) ii.aux = (uint32_t(mlh) << 16) | uint32_t(mlw);
            }
            if (ibc) {
                ii.flags = uint8_t(has ? ea1::kIIResid : 0);   // no edges
                ii.aux = (uint32_t(uint16_t(b.mv[0][0])) << 16) | uint16_t(b.mv[0][1]);
            }
            if (mode > EA1_PAETH_PRED || (p_angle && p_angle != 90 && p_angle != 180)) acc.intra_special = true;
            if (palette) {
                const int ny = b.pal_size[0], nuv = b.pal_size[1];
                ea1::PalRef pr;
                pr.colors_off = b.aux_off + uint32_t(pl == 0 ? 0 : ny + (pl - 1) * nuv);
                pr.map_stride = uint32_t(pbw);
                pr.map_off = b.aux_off + uint32_t(ny + 2 * nuv + (pl ? bw * bh : 0) + (ty - by) * pbw + (tx_ - bx));
                ii.aux = uint32_t(pal_refs.size());
                pal_refs.push_back(pr);
            }
            const int rpp = rows_per_part(tw);
            for (int part = 0; part * rpp < th; part++) {
                ii.part = uint8_t(part);
                intra.push_back(LevelItem{lv, ii});
            }
            acc.intra_pixels += uint64_t(tw) * th;
            if (has) acc.intra_resid_pixels += uint64_t(tw) * th;
        }
    }
    if (max_level > 65534u) {
        std::fill(lvl[0].begin(), lvl[0].end(), uint16_t(0)), std::fill(lvl[1].begin(), lvl[1].end(), uint16_t(0)), std::fill(lvl[2].begin(), lvl[2].end(), uint16_t(0));
        return ea1::fail(EA1_ENOTSUP, "ea1_job_create: %u dependency levels of intra prediction (more than the level map holds)", max_level);
    }
    auto t_loop = now();
    // the wavefront items by level, order inside a level kept: counting sort again (levels are 1 .. max_level)
    std::vector<ea1::IntraItem> &intra_items = hs.intra_items;
    std::vector<uint32_t> &level_start = hs.level_start;
    intra_items.resize(intra.size());
    level_start.assign(max_level + 1, 0);
    {
        std::vector<uint32_t> &at = hs.at;
        at.assign(max_level + 2, 0);
        for (const LevelItem &li : intra) at[li.level + 1]++;
        for (uint32_t lv = 1; lv <= max_level + 1; lv++) at[lv] += at[lv - 1];   // at[lv] = first index of level lv
        for (uint32_t lv = 1; lv <= max_level; lv++) level_start[lv - 1] = at[lv];
        level_start[max_level] = uint32_t(intra.size());
        for (const LevelItem &li : intra) intra_items[at[li.level]++] = li.it;
    }
    auto t_sort = now();
    // the level map goes back to all zero for the next frame of this thread
    if (dense_levels) {
        for (int pl = 0; pl < 3; pl++) std::fill(lvl[pl].begin(), lvl[pl].end(), uint16_t(0));
    } else {
        for (const HostLists::Touched &t : hs.touched) {
            const int pw4 = t.pl ? cw4 : w4;
            for (uint32_t j = 0; j < t.nh; j++) std::fill_n(lvl[t.pl].begin() + (size_t(t.uy0 + j) * pw4 + t.ux0), size_t(t.nw), uint16_t(0));
        }
    }
    hs.lvl_clean = true;

    t_lists = now();
    // ---- layout of the job's arena, staging (no HIP call: arena and staging buffer come from the warm pool) ----
    auto *job = new (std::nothrow) ea1_job();
    if (!job) return ea1::fail(EA1_ENOMEM, "out of memory");
    job->device = r->device;
    FrameView &d = job->view;
    memset(&d, 0, sizeof d);
    d.desc = *desc;
    d.w8 = w8;
    d.h8 = h8;
    d.cw4 = cw4;
    d.ch4 = ch4;
    d.ssx = ssx;
    d.ssy = ssy;
    d.sb_cols = sb_cols;
    d.sb_rows = sb_rows;
    d.vis_w = int(desc->width) - 8 * (desc->pad & 15) - (desc->crop & 15);   // crop counts from the TRUE coded size (ea1_frame_desc.pad)
    d.vis_h = int(desc->height) - 8 * (desc->pad >> 4) - (desc->crop >> 4);
    bool lr_sgr[3] = {false, false, false};
    int rc = EA1_OK;
    Packer &pk = hs.pk;
    pk.reset();
    ea1::DeriveArgs da;
    memset(&da, 0, sizeof da);
    const ea1::DeriveArgs *d_da = nullptr;
    pk.add(&da, 1, &d_da);   // filled in below, once the addresses are known, before it is staged
    // the block and transform-block records themselves: input of the derivation kernels (av1_derive.hip)
    const ea1_block *d_blocks = nullptr;
    const ea1_txblock *d_txs = nullptr;
    pk.add(blocks, nb, &d_blocks);
    pk.add(txs, ntx, &d_txs);
    pk.add(aux, desc->num_aux, &d.aux, 16);   // margin: palette lookups of unused colour slots never happen, but keep one
    pk.add(pal_refs.data(), pal_refs.size(), &d.pal_refs);
    // (the inter tile lists are placed further down: behind everything that travels through the staging buffer, so
    // that the staged bytes of a job go in ONE copy)
    d.piece_count = int(pieces.size());
    pk.add(pieces.data(), pieces.size(), &d.piece_items);
    d.warp_luma_count = int(warps.size());
    warps.insert(warps.end(), warps_chroma.begin(), warps_chroma.end());
    d.warp_count = int(warps.size());
    pk.add(warps.data(), warps.size(), &d.warp_items);
    for (int k = 0; k < 2; k++) {
        d.obmc_count[k] = int(obmc[k].size());
        pk.add(obmc[k].data(), obmc[k].size(), &d.obmc_items[k]);
    }
    if (cdef_idx) pk.add(cdef_idx, size_t(sb_cols) * sb_rows, &d.cdef_idx);
    for (int pl = 0; pl < 3 && rc == EA1_OK; pl++) {
        const int sx_ = pl ? ssx : 0, sy_ = pl ? ssy : 0;
        if (!desc->lr_type[pl] || !lr || !lr[pl]) continue;
        if (desc->lr_unit_log2[pl] < 5 || desc->lr_unit_log2[pl] > 8) {
            rc = ea1::fail(EA1_EINVAL, "ea1_job_create: lr_unit_log2[%d] = %u", pl, desc->lr_unit_log2[pl]);
            break;
        }
        const int usz = 1 << desc->lr_unit_log2[pl];
        if (usz < ((pl && ssx && ssy) ? 32 : 64)) {
            rc = ea1::fail(EA1_EINVAL, "ea1_job_create: restoration unit %d too small for plane %d", usz, pl);
            break;
        }
        d.lr_rows[pl] = lr_units(usz, (d.vis_h + sy_) >> sy_);
        d.lr_cols[pl] = lr_units(usz, ((desc->upscaled_width ? int(desc->upscaled_width) : d.vis_w) + sx_) >> sx_);   // (super-resolution: counted on the upscaled width)
        if (lengths && uint64_t(lengths->lr_count[pl]) < uint64_t(d.lr_rows[pl]) * d.lr_cols[pl]) {
            rc = ea1::fail(EA1_EINVAL, "parsed frame: %u restoration units of plane %d, %d needed", lengths->lr_count[pl], pl, d.lr_rows[pl] * d.lr_cols[pl]);
            break;
        }
        pk.add(lr[pl], size_t(d.lr_rows[pl]) * d.lr_cols[pl], &d.lr[pl]);
        for (int k = 0; k < d.lr_rows[pl] * d.lr_cols[pl]; k++) lr_sgr[pl] |= lr[pl][k].type == EA1_LR_SGRPROJ;
    }
    pk.add(intra_items.data(), intra_items.size(), &d.intra_items);
    pk.add(level_start.data(), level_start.size(), &d.level_start);
    for (int k = 0; k < 3; k++) {
        // host-made tiles first (usually none), the device-made ones behind them
        d.mc_count[k] = int(mc[k].size() + mc_dev[k]);
        pk.add_with_device_tail(mc[k].data(), mc[k].size(), mc_dev[k], &d.mc_items[k]);
        d.mc_ext_count[k] = int(mc_ext[k].size() + mc_ext_dev[k]);
        pk.add_with_device_tail(mc_ext[k].data(), mc_ext[k].size(), mc_ext_dev[k], &d.mc_ext_items[k]);
    }
    // last: kZeroCoefs zero coefficients (written by the first derivation kernel) and, straight behind them and not
    // through the staging buffer, the coefficients.  Offsets of the transform items count from the zeros.
    // (EA1_COEFS_DIAGONAL: the zeros are followed by the corner-layout region the device expands the blocks into, the
    // uploaded packed diagonals come behind that)
    const int16_t *d_zero_coefs = nullptr, *d_own_coefs = nullptr;
    if (diagonal_layout && (expanded_entries >> 31 || uint64_t(coef_hi - coef_lo) >> 25)) {
        delete job;
        return ea1::fail(EA1_EINVAL, "ea1_job_create: coefficient arena too large for the diagonal layout");
    }
    pk.add_with_device_tail(static_cast<const int16_t *>(nullptr), 0, ea1::kZeroCoefs + (diagonal_layout ? size_t(expanded_entries) : 0), &d_zero_coefs);
    pk.add(coefs ? coefs + coef_lo : coefs, coef_hi - coef_lo, &d_own_coefs, 16, true);
    // device-only regions: the FrameView of the chain the job gets bound to, the derived maps and lists, their scratch
    const size_t n4 = size_t(w4) * h4, n8 = size_t(w8) * h8, nc4 = size_t(cw4) * ch4;
    const size_t o_view = pk.reserve_device(sizeof(FrameView));
    const size_t o_ledge = pk.reserve_device(n4), o_cedge = pk.reserve_device(nc4), o_uflags = pk.reserve_device(n8);
    const size_t o_lvly = pk.reserve_device(2 * n4), o_lvlc = pk.reserve_device(2 * nc4);
    const size_t o_map = pk.reserve_device(4 * n4), o_txm0 = pk.reserve_device(4 * n4), o_txm1 = pk.reserve_device(4 * nc4);
    const size_t o_cursors = pk.reserve_device(sizeof(uint32_t) * ea1::kDeriveCursors);
    size_t o_tx[ea1::kTxSizes];
    for (int k = 0; k < ea1::kTxSizes; k++) {
        uint32_t n = tx_zero_fill[k];
        for (int bk = 0; bk < 4; bk++) n += tx_bucket[k][bk];
        d.tx_count[k] = int(n);
        o_tx[k] = pk.reserve_device(size_t(n) * sizeof(TxItem));
    }
    job->pool = r->pool;
    if (rc == EA1_OK && !r->pool) rc = ea1::fail(EA1_ENOMEM, "ea1_job_create: the context has no device memory pool (no device)");
    if (rc == EA1_OK) {
        JobPool &pool = *r->pool;
        std::unique_ptr<ea1_job::Pending> pend(new (std::nothrow) ea1_job::Pending());
        const size_t staged = pk.staged_bytes();
        if (!pend) rc = ea1::fail(EA1_ENOMEM, "out of memory");
        else if (!pool.take(false, pk.total + pk.device_only + 256, &job->arena)) rc = ea1::fail(EA1_ENOMEM, "hipMalloc(%zu) failed", pk.total + pk.device_only);
        else if (!pool.take(true, std::max<size_t>(staged, 256), &pend->pin)) rc = ea1::fail(EA1_ENOMEM, "hipHostMalloc(%zu) failed", staged);
        if (rc == EA1_OK) {
            uint8_t *dbase = job->arena.p, *dob = dbase + pk.total;
            pk.place(dbase);
            d.coefs = d_zero_coefs;   // kZeroCoefs * 2 bytes is a multiple of the 256-byte part alignment: the coefficients follow directly
            if (!diagonal_layout && coef_hi > coef_lo && d_own_coefs != d_zero_coefs + ea1::kZeroCoefs) rc = ea1::fail(EA1_EINVAL, "ea1_job_create: internal layout error");
            job->d_view_slot = reinterpret_cast<FrameView *>(dob + o_view);
            da.blocks = d_blocks;
            da.txs = d_txs;
            da.num_blocks = nb;
            da.num_tx = ntx;
            da.luma_edge = dob + o_ledge;
            da.chroma_edge = dob + o_cedge;
            da.unit_flags = dob + o_uflags;
            da.lvl_y = reinterpret_cast<uint16_t *>(dob + o_lvly);
            da.lvl_c = reinterpret_cast<uint16_t *>(dob + o_lvlc);
            da.map = reinterpret_cast<uint32_t *>(dob + o_map);
            da.txmap[0] = reinterpret_cast<uint32_t *>(dob + o_txm0);
            da.txmap[1] = reinterpret_cast<uint32_t *>(dob + o_txm1);
            da.txmap[2] = nullptr;
            da.w4 = w4;
            da.h4 = h4;
            da.w8 = w8;
            da.h8 = h8;
            da.cw4 = cw4;
            da.ch4 = ch4;
            da.ssx = ssx;
            da.ssy = ssy;
            da.vis_w = int(desc->width) - 8 * (desc->pad & 15) - (desc->crop & 15);
            da.vis_h = int(desc->height) - 8 * (desc->pad >> 4) - (desc->crop >> 4);
            memcpy(da.lf_level, desc->lf_level, 4);
            da.lf_per_block = desc->lf_per_block;
            da.corner_layout = corner_layout ? 1 : 0;
            da.strip_x0 = strip_x0;
            da.strip_x1 = strip_x1;
            for (int k = 0; k < ea1::kTxSizes; k++) {
                da.tx_items[k] = reinterpret_cast<TxItem *>(dob + o_tx[k]);
                d.tx_items[k] = d.tx_count[k] ? da.tx_items[k] : nullptr;
                // bucket 0 = unhinted + zero-fill + hinted 4x4 corners, then the larger corners: equal corner classes
                // next to each other (fewer divergent groups per wave)
                uint32_t run = 0;
                for (int bk = 0; bk < 4; bk++) {
                    da.tx_base[k][bk] = run;
                    run += tx_bucket[k][bk] + (bk == 0 ? tx_zero_fill[k] : 0);
                }
            }
            for (int k = 0; k < 3; k++) {
                da.mc_items[k] = const_cast<ea1::McItem *>(d.mc_items[k]);
                da.mc_ext_items[k] = const_cast<ea1::McItem *>(d.mc_ext_items[k]);
                da.mc_base[k] = uint32_t(mc[k].size());
                da.mc_ext_base[k] = uint32_t(mc_ext[k].size());
            }
            da.cursors = reinterpret_cast<uint32_t *>(dob + o_cursors);
            da.coefs = const_cast<int16_t *>(d_zero_coefs);
            da.coef_rebase = ea1::kZeroCoefs - coef_lo;
            da.packed = d_own_coefs;
            da.packed_lo = coef_lo;
            da.diagonal_layout = diagonal_layout ? 1 : 0;
            for (int k = 0; k < ea1::kTxSizes; k++) da.tx_count[k] = uint32_t(d.tx_count[k]);
            d.luma_edge = da.luma_edge;
            d.chroma_edge = da.chroma_edge;
            d.unit_flags = da.unit_flags;
            d.lvl_y = da.lvl_y;
            d.lvl_c = da.lvl_c;
            pk.stage(dbase, pend->pin.p, &pend->copies);
            pend->d_args = d_da;
            pend->nb = nb;
            pend->ntx = ntx;
            pend->units = int(n4 + nc4);
            if (diagonal_layout)
                for (int k = 0; k < ea1::kTxSizes; k++) pend->max_items = std::max(pend->max_items, uint32_t(d.tx_count[k]));
        }
        job->pending = std::move(pend);
    }
    d.num_levels = int(max_level);
    // per size: coded transform blocks, coefficient positions and pixels they touch (SURVEY.md 8(d) accounting)
    for (int k = 0; k < ea1::kTxSizes; k++) {
        uint32_t coded = 0;
        for (int bk = 0; bk < 4; bk++) coded += tx_bucket[k][bk];
        const int wl = tt.size_of[k] & 15, hl = tt.size_of[k] >> 4;
        acc.tx_coefs[k] = uint64_t(coded) * uint64_t(std::min(1 << wl, 32) * std::min(1 << hl, 32));
        acc.tx_pixels[k] = uint64_t(coded + tx_zero_fill[k]) << (wl + hl);
    }
    job->stats = acc;
    for (int pl = 0; pl < 3; pl++) job->stats.lr_has_sgr[pl] = lr_sgr[pl];
    job->intra_count = intra_items.size();
    for (int k = 0; k < ea1::kTxSizes; k++) job->stats.tx_count[k] = d.tx_count[k];
    for (int k = 0; k < 3; k++) job->stats.mc_count[k] = d.mc_count[k];
    for (int k = 0; k < 3; k++) job->stats.mc_ext_count[k] = d.mc_ext_count[k];
    for (int k = 0; k < 2; k++) job->stats.obmc_count[k] = int(obmc[k].size());
    job->stats.piece_count = int(pieces.size());
    job->stats.warp_count = int(warps.size());
    job->stats.warp_luma_count = d.warp_luma_count;
    job->stats.ref_mask = ref_mask;
    job->stats.num_levels = int(max_level);
    for (uint32_t lv = 0; lv < max_level; lv++)
        job->stats.max_level_items = std::max(job->stats.max_level_items, int(level_start[lv + 1] - level_start[lv]));
    job->stats.luma_odd_row_edges = luma_odd_row_edges;
    job->level_start = level_start;
    job->stats.level_start = job->level_start.data();
    job->stats.level0_items = max_level ? int(level_start[1] - level_start[0]) : 0;
    job->stats.ref_luma_px = ref_pixels;
    job->stats.inter_luma_px = inter_px;
    job->stats.w8 = w8;
    job->stats.h8 = h8;
    job->stats.width = int(desc->width);
    job->stats.height = int(desc->height);
    job->stats.desc = *desc;
    job->stats.strip_x0 = std::max(strip_x0, 0);
    job->stats.strip_x1 = std::min(strip_x1, W);
    if (timing) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "ea1_job_create: validate %.3f ms, lists %.3f ms, pack + stage %.3f ms (loop %.3f sort %.3f clear %.3f)\n", ms(t_start, t_valid), ms(t_valid, t_lists), ms(t_lists, now()), ms(t_valid, t_loop), ms(t_loop, t_sort), ms(t_sort, t_lists));
    }
    if (rc != EA1_OK) {
        ea1_job_destroy(job);
        return rc;
    }
    const uint64_t luma_px = uint64_t(desc->width) * desc->height;
    job->refs_per_pixel = luma_px ? double(ref_pixels) / double(luma_px) : 0.0;
    job->coef_count = coef_count;
    job->block_bytes = uint64_t(nb) * sizeof(ea1_block) + uint64_t(ntx) * sizeof(ea1_txblock);
    *out = job;
    return EA1_OK;
}

// The device half of the hand-over for n prepared jobs (one thread): their copies down the pool's copy stream, then
// ONE launch of each derivation kernel for all of them on a derive stream behind one event, and one `ready` event
// the jobs share.  issue_copies may be called job by job as the host halves finish (ea1_job_create_batch).
// The records and coefficients of a frame travel by hipMemcpyAsync on the pool's copy stream (the DMA engines).
// EA1_PULL=1 moves what lies in pinned memory with ONE kernel per kPullSegs buffers instead (av1_split.hip
// pull_kernel): on an idle device that is the faster way for frames of this size (17.6 k against 14.0 k 1080p frames/s
// handed over, profiles/r02_h_handover_copy_paths.txt), but beside the reconstruction kernels -- the case that
// matters -- the pull waits for wave slots and loses (14.0 k against 16.9 k), whatever its stream priority or grid.
// The FrameViews of a submit (1.5 KB each, one per chain) do go through the pull kernel: see bind_job.
struct PullQueue {
    ea1::PullArgs args;
    hipStream_t st, st2 = nullptr;   // st2: optional second copy stream, jobs alternate (EA1_COPY_STREAMS=2)
    unsigned turn = 0;
    explicit PullQueue(hipStream_t s, hipStream_t s2 = nullptr) : st(s), st2(s2) { args.n = 0; }
    int flush()
    {
        if (!args.n) return EA1_OK;
        ea1::launch_pull(args, st);
        args.n = 0;
        if (hipGetLastError() != hipSuccess) return ea1::fail(EA1_EHIP, "ea1_job_create: launch of the pull kernel failed");
        return EA1_OK;
    }
};

static int issue_copies(ea1_job *job, PullQueue &q)
{
    if (!job->pending) return EA1_OK;
    static const bool no_pull = getenv("EA1_PULL") == nullptr;
    hipStream_t mst = (q.st2 && (q.turn++ & 1)) ? q.st2 : q.st;
    static const size_t pull_max = getenv("EA1_PULL_MAX_KB") ? size_t(atol(getenv("EA1_PULL_MAX_KB"))) << 10 : ~size_t(0);
    for (const ea1_job::Copy &c : job->pending->copies) {
        const bool aligned = !((reinterpret_cast<uintptr_t>(c.dst) | reinterpret_cast<uintptr_t>(c.src)) & 15) && !(c.bytes & 1);
        if (c.pinned && aligned && !no_pull && c.bytes <= pull_max) {
            if (q.args.n == ea1::kPullSegs) {
                const int rc = q.flush();
                if (rc != EA1_OK) return rc;
            }
            ea1::PullSeg &sg = q.args.seg[q.args.n++];
            sg.dst = c.dst;
            sg.src = c.src;
            sg.bytes = c.bytes;
        } else if (hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, mst) != hipSuccess) {
            return ea1::fail(EA1_EHIP, "ea1_job_create: a host-to-device copy failed");
        }
    }
    job->pending->copies.clear();
    return EA1_OK;
}

static int issue_derive(ea1_recon *r, ea1_job *const *jobs, int n)
{
    JobPool &pool = *r->pool;
    hipStream_t up = pool.copy, dv = pool.derive_stream();
    auto copied = std::make_shared<SubmitMark>(), ready = std::make_shared<SubmitMark>();
    copied->pool = r->pool;
    ready->pool = r->pool;
    {
        std::lock_guard<std::recursive_mutex> lk(pool.mu);
        copied->ev = pool.event();
        ready->ev = pool.event();
    }
    int rc = EA1_OK;
    // copies -> (event) -> derivation kernels -> ready.  The copy stream goes straight on to the next batch's bytes.
    if (pool.copy2) {   // the second copy stream joins the first
        auto joined = std::make_shared<SubmitMark>();
        joined->pool = r->pool;
        {
            std::lock_guard<std::recursive_mutex> lk(pool.mu);
            joined->ev = pool.event();
        }
        if (hipEventRecord(joined->ev, pool.copy2) != hipSuccess || hipStreamWaitEvent(up, joined->ev, 0) != hipSuccess) rc = ea1::fail(EA1_EHIP, "event plumbing of the upload failed");
    }
    if (rc == EA1_OK && (hipEventRecord(copied->ev, up) != hipSuccess || hipStreamWaitEvent(dv, copied->ev, 0) != hipSuccess)) rc = ea1::fail(EA1_EHIP, "event plumbing of the upload failed");
    for (int k0 = 0; k0 < n && rc == EA1_OK; k0 += ea1::kMaxChains) {
        ea1::DeriveBatch db;
        db.n = 0;
        uint32_t max_blocks = 0, max_tx = 0, max_items = 0;
        int max_units = 0;
        for (int k = k0; k < n && k < k0 + ea1::kMaxChains; k++) {
            const ea1_job::Pending &p = *jobs[k]->pending;
            db.a[db.n++] = p.d_args;
            max_blocks = std::max(max_blocks, p.nb);
            max_tx = std::max(max_tx, p.ntx);
            max_units = std::max(max_units, p.units);
            max_items = std::max(max_items, p.max_items);
        }
        ea1::launch_derive(db, max_blocks, max_tx, max_units, max_items, dv);
        if (hipGetLastError() != hipSuccess) rc = ea1::fail(EA1_EHIP, "ea1_job_create: launch of the derivation kernels failed");
    }
    if (rc == EA1_OK && hipEventRecord(ready->ev, dv) != hipSuccess) rc = ea1::fail(EA1_EHIP, "hipEventRecord failed");
    // the staging buffers go back to the pool behind the copies, the jobs wait for `ready`
    for (int k = 0; k < n; k++) {
        ea1_job *job = jobs[k];
        if (job->pending->pin.p) pool.give_behind(job->pending->pin, true, copied);
        job->pending.reset();
        job->ready = ready;
    }
    return rc;
}

static int job_create(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                      const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx,
                      const ea1_lr_unit *const lr[3], int strip_x0, int strip_x1, ea1_job **out)
{
    ea1_job *job = nullptr;
    int rc = job_prepare(r, desc, blocks, txs, coefs, aux, cdef_idx, lr, strip_x0, strip_x1, &job);
    if (rc != EA1_OK) return rc;
    ea1::DeviceGuard g(r->device);
    PullQueue q(r->pool->copy, r->pool->copy2);
    rc = issue_copies(job, q);
    if (rc == EA1_OK) rc = q.flush();
    if (rc == EA1_OK) rc = issue_derive(r, &job, 1);
    if (rc != EA1_OK) {
        ea1_job_destroy(job);
        *out = nullptr;
        return rc;
    }
    *out = job;
    return EA1_OK;
}

extern "C" int ea1_job_create(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                              const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx,
                              const ea1_lr_unit *const lr[3], ea1_job **out)
{
    return job_create(r, desc, blocks, txs, coefs, aux, cdef_idx, lr, 0, 1 << 30, out);
}

extern "C" int ea1_job_create_strip(ea1_recon *r, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                    const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx,
                                    const ea1_lr_unit *const lr[3], uint32_t x_begin, uint32_t x_end, ea1_job **out)
{
    if (!r || !desc) return ea1::fail(EA1_EINVAL, "ea1_job_create_strip: NULL argument");
    if (x_begin >= x_end || x_end > desc->width || (x_begin & 63) || ((x_end & 63) && x_end != desc->width))
        return ea1::fail(EA1_EINVAL, "ea1_job_create_strip: columns %u..%u are not a superblock-aligned range of the frame", x_begin, x_end);
    // super-resolution (7.16) stretches every row over the whole upscaled width between CDEF and restoration: a strip of the downscaled frame does not
    // map onto a strip of the picture that comes out, and the halo exchange of the split (include/easyav1_amd_video.h) knows one width only.  Refused
    // rather than wrong; such streams decode on one GPU (GOP chains shard them without a split).
    if (desc->upscaled_width) return ea1::fail(EA1_ENOTSUP, "ea1_job_create_strip: a frame with super-resolution cannot be split by tile columns");
    return job_create(r, desc, blocks, txs, coefs, aux, cdef_idx, lr, int(x_begin), int(x_end), out);
}

// n parsed frames handed over on up to `threads` host threads of the library (validation and list building are
// per-frame host work with nothing shared but the memory pool).  out[k] = job of frames[k]; on failure every job made
// so far is destroyed, the first error is returned and its text is the calling thread's ea1_last_error().
extern "C" int ea1_job_create_batch(ea1_recon *r, const ea1_ir_frame *const *frames, int n, int threads, ea1_job **out)
{
    if (!r || !frames || !out || n < 0) return ea1::fail(EA1_EINVAL, "ea1_job_create_batch: bad arguments");
    for (int k = 0; k < n; k++) out[k] = nullptr;
    if (!r->workers) return ea1::fail(EA1_ENOMEM, "ea1_job_create_batch: the context has no worker pool");
    std::vector<int> rcs(static_cast<size_t>(n), EA1_OK);
    std::vector<std::string> msgs{static_cast<size_t>(n)};
    ea1::DeviceGuard g(r->device);
    int issue_rc = EA1_OK;
    std::string issue_msg;
    PullQueue q(r->pool ? r->pool->copy : nullptr, r->pool ? r->pool->copy2 : nullptr);
    int queued_jobs = 0;
    // helpers: the host half of every frame; this thread: the copies of each frame as soon as its host half is done
    r->workers->run_with_issuer(
        n, threads,
        [&](int k) {
            const ea1_ir_frame *f = frames[k];
            if (!f) {
                rcs[size_t(k)] = EA1_EINVAL;
                msgs[size_t(k)] = "ea1_job_create_batch: NULL frame";
                return;
            }
            const ea1_lr_unit *lr[3] = {f->lr[0], f->lr[1], f->lr[2]};
            const ArrayLengths len{f->num_sb, {f->lr_count[0], f->lr_count[1], f->lr_count[2]}};
            rcs[size_t(k)] = job_prepare(r, &f->desc, f->blocks, f->txs, f->coefs, f->aux, f->cdef_idx, lr, 0, 1 << 30, &out[k], &len);
            if (rcs[size_t(k)] != EA1_OK) msgs[size_t(k)] = ea1_last_error();   // thread-local: carried over to the caller below
        },
        [&](int k) {
            if (rcs[size_t(k)] != EA1_OK || issue_rc != EA1_OK) return;
            issue_rc = issue_copies(out[k], q);
            // a launch per few frames: the link starts on the first frames while the helpers work on the rest
            static const int pull_jobs = getenv("EA1_PULL_JOBS") ? atoi(getenv("EA1_PULL_JOBS")) : 8;
            if (issue_rc == EA1_OK && ++queued_jobs >= pull_jobs) {
                issue_rc = q.flush();
                queued_jobs = 0;
            }
            if (issue_rc != EA1_OK) issue_msg = ea1_last_error();
        });
    if (issue_rc == EA1_OK) {
        issue_rc = q.flush();
        if (issue_rc != EA1_OK) issue_msg = ea1_last_error();
    }
    int first_bad = -1;
    for (int k = 0; k < n && first_bad < 0; k++)
        if (rcs[size_t(k)] != EA1_OK) first_bad = k;
    if (first_bad < 0 && issue_rc == EA1_OK && n > 0) {
        issue_rc = issue_derive(r, out, n);
        if (issue_rc != EA1_OK) issue_msg = ea1_last_error();
    }
    if (first_bad >= 0 || issue_rc != EA1_OK) {
        // copies of the frames that did pass may be in flight out of staging blocks and caller-pinned arrays: the jobs are about to
        // go (their arenas and staging return to the pool at once: no submit ever marked them), so wait for the copy streams first
        (void)hipStreamSynchronize(r->pool->copy);
        if (r->pool->copy2) (void)hipStreamSynchronize(r->pool->copy2);
        for (int j = 0; j < n; j++) {
            ea1_job_destroy(out[j]);
            out[j] = nullptr;
        }
        if (first_bad >= 0) return ea1::fail(rcs[size_t(first_bad)], "frame %d: %s", first_bad, msgs[size_t(first_bad)].c_str());
        return ea1::fail(issue_rc, "%s", issue_msg.c_str());
    }
    return EA1_OK;
}

// The two halves of the hand-over as calls of their own, for a caller that pipelines frames itself (libeasyav1.so's video
// sink: helper threads prepare, one thread commits in stream order).
extern "C" int ea1_job_prepare(ea1_recon *r, const ea1_ir_frame *f, ea1_job **out)
{
    if (!r || !f || !out) return ea1::fail(EA1_EINVAL, "ea1_job_prepare: NULL argument");
    *out = nullptr;
    const ea1_lr_unit *lr[3] = {f->lr[0], f->lr[1], f->lr[2]};
    const ArrayLengths len{f->num_sb, {f->lr_count[0], f->lr_count[1], f->lr_count[2]}};
    return job_prepare(r, &f->desc, f->blocks, f->txs, f->coefs, f->aux, f->cdef_idx, lr, 0, 1 << 30, out, &len);
}

extern "C" int ea1_job_commit(ea1_recon *r, ea1_job *const *jobs, int n)
{
    if (!r || !jobs || n <= 0) return ea1::fail(EA1_EINVAL, "ea1_job_commit: bad arguments");
    for (int k = 0; k < n; k++)
        if (!jobs[k] || !jobs[k]->pending) return ea1::fail(EA1_ESTATE, "ea1_job_commit: job %d is not a prepared job (or was committed before)", k);
    ea1::DeviceGuard g(r->device);
    std::lock_guard<std::mutex> one(r->commit_mu);
    PullQueue q(r->pool->copy, r->pool->copy2);
    int rc = EA1_OK;
    for (int k = 0; k < n && rc == EA1_OK; k++) rc = issue_copies(jobs[k], q);
    if (rc == EA1_OK) rc = q.flush();
    if (rc == EA1_OK) rc = issue_derive(r, jobs, n);
    if (rc != EA1_OK) {   // the jobs stay the caller's (to destroy); nothing of them may still be on the wire then
        (void)hipStreamSynchronize(r->pool->copy);
        if (r->pool->copy2) (void)hipStreamSynchronize(r->pool->copy2);
    }
    return rc;
}

extern "C" int ea1_job_stats(const ea1_job *job, double *ref_pixels_per_pixel, uint64_t *coef_count,
                             uint64_t *block_bytes, uint64_t *algorithmic_bytes)
{
    if (!job) return ea1::fail(EA1_EINVAL, "ea1_job_stats: NULL job");
    const double P = 1.5 * double(job->view.desc.width) * double(job->view.desc.height);
    if (ref_pixels_per_pixel) *ref_pixels_per_pixel = job->refs_per_pixel;
    if (coef_count) *coef_count = job->coef_count;
    if (block_bytes) *block_bytes = job->block_bytes;
    if (algorithmic_bytes)
        *algorithmic_bytes = uint64_t(P * (1.0 + job->refs_per_pixel)) + 2 * job->coef_count + job->block_bytes;
    return EA1_OK;
}

extern "C" int ea1_job_schedule_info(const ea1_job *job, int *num_levels, int *max_level_items, int *intra_items,
                                     int *mc_tiles /* [3] */, int *tx_blocks /* [EA1_NUM_TX_SIZES] */)
{
    if (!job) return ea1::fail(EA1_EINVAL, "ea1_job_schedule_info: NULL job");
    if (num_levels) *num_levels = job->stats.num_levels;
    if (max_level_items) *max_level_items = job->stats.max_level_items;
    if (intra_items) *intra_items = int(job->intra_count);
    for (int k = 0; k < 3 && mc_tiles; k++) mc_tiles[k] = job->stats.mc_count[k];
    for (int k = 0; k < ea1::kTxSizes && tx_blocks; k++) tx_blocks[k] = job->stats.tx_count[k];
    return EA1_OK;
}

// device FrameView of `job` specialised for chain `ci` of `r` (created on first use)
// first-time uploads of a submit's FrameViews: one pinned block for all of them, pulled by a kernel on the submit
// stream ahead of the kernels that read them (a launch per kPullSegs views instead of a copy call per job)
struct ViewUpload {
    JobPool::Block pin{nullptr, 0};
    size_t used = 0;
    PullQueue q;
    explicit ViewUpload(hipStream_t st) : q(st) {}
};

static int bind_job(ea1_recon *r, int ci, ea1_job *job, ViewUpload &vu, const FrameView **out)
{
    const Chain &c = r->chains[size_t(ci)];
    FrameView v = job->view;
    const int fw = int(v.desc.width), fh = int(v.desc.height);
    // scale factors of the references this frame reads (AV1 spec 7.11.3.3: xScale = ((RefUpscaledWidth << 14) + FrameWidth / 2) / FrameWidth)
    v.scaled_mask = 0;
    for (int i = 0; i < EA1_NUM_REF_SLOTS; i++) {
        v.slots[i] = c.slots[i];
        v.ref_scale[i][0] = v.ref_scale[i][1] = 1 << 14;
        const bool used = (job->stats.ref_mask >> i) & 1u;
        // on the TRUE sizes of the two pictures (coded size minus crop: the specification's RefUpscaledWidth / FrameWidth, dav1d's
        // p.w / p.h), not on their sizes in whole 8x8 units: 1080 -> 540 rows is exactly 2 : 1 although 1080 / 544 is not
        const int vw = v.vis_w, vh = v.vis_h;
        if (used && c.valid[i] && (c.slot_vw[i] != vw || c.slot_vh[i] != vh)) {
            const int rw = c.slot_vw[i], rh = c.slot_vh[i];
            if (2 * vw < rw || 2 * vh < rh || vw > 16 * rw || vh > 16 * rh)
                return ea1::fail(EA1_EINVAL, "ea1_recon_submit: chain %d: reference slot %d is %dx%d, the frame %dx%d: beyond the 2 : 1 / 1 : 16 limits of scaled references", ci, i, rw, rh, vw, vh);
            v.ref_scale[i][0] = int(((int64_t(rw) << 14) + vw / 2) / vw);
            v.ref_scale[i][1] = int(((int64_t(rh) << 14) + vh / 2) / vh);
            v.scaled_mask |= 1u << i;
            v.slots[i].w = c.slot_vw[i];   // reads clamp at the REFERENCE's visible picture
            v.slots[i].h = c.slot_vh[i];
        } else {
            v.slots[i].w = v.vis_w;   // reference reads clamp at the visible picture (AV1 spec 7.11.3.3 lastX / lastY)
            v.slots[i].h = v.vis_h;
        }
    }
    if (v.scaled_mask) {
        // the scaled path covers every translated prediction (blocks of any size, every compound kind, OBMC, the inter half
        // of inter-intra blocks).  Warped motion never meets a scaled reference in a valid stream (7.11.3.1: a scaled
        // reference switches warping off)
        // -- per reference: a block flagged for warped motion translates from its scaled references (the hand-over marks only the predictions
        // that really warp; a compound block may warp from one reference and translate from a scaled other one: mc_warp_kernel's translated
        // half goes through the scaled arithmetic then.  Round 5: refused until the generated streams that dav1d decodes showed it is common)
        if (job->stats.warp_ref_mask & v.scaled_mask)
            return ea1::fail(EA1_EINVAL, "ea1_recon_submit: chain %d: a prediction warps from a scaled reference (AV1 spec 7.11.3.1 rules that out)", ci);
    }
    ea1_job::Bound *hit = nullptr;
    for (auto &b : job->bound)
        if (b.ctx == r && b.chain == ci) hit = &b;
    if (hit && !hit->scaled_mask && !v.scaled_mask) {
        *out = hit->d_view;
        return EA1_OK;
    }
    v.work = c.work;
    v.cdef = c.cdef;
    v.dst = c.slots[v.desc.dst_slot];
    // the frame's own TRUE coded size (a context decodes frames up to its size; blocks may reach beyond the coded size into the
    // padding of ea1_frame_desc.pad): what intra edge fetches and CDEF clamp at
    v.work.w = v.cdef.w = v.dst.w = fw - 8 * (v.desc.pad & 15);
    v.work.h = v.cdef.h = v.dst.h = fh - 8 * (v.desc.pad >> 4);
    for (int pl = 0; pl < 3; pl++) v.resid[pl] = c.resid[pl];
    v.up_cw = v.up_vis_w = 0;
    v.upscale_filter = r->d_upscale;
    if (v.desc.upscaled_width) {
        // super-resolution: the two upscaled pictures of the chain (made with its first such frame)
        if (!r->upscale_set) return ea1::fail(EA1_ESTATE, "ea1_recon_submit: chain %d: super-resolution without an Upscale_Filter table", ci);
        Chain &cm = r->chains[size_t(ci)];
        if (!cm.up_arena) {
            if (hipMalloc(&cm.up_arena, 2 * r->pic_bytes) != hipSuccess) return ea1::fail(EA1_ENOMEM, "hipMalloc(upscaled pictures) failed");
            for (int k = 0; k < 2; k++) layout_pic(cm.up[k], cm.up_arena + r->pic_bytes * size_t(k), int(r->width), int(r->height), r->ssx, r->ssy, nullptr);
        }
        v.up_deb = cm.up[0];
        v.up_cdef = cm.up[1];
        v.up_vis_w = int(v.desc.upscaled_width);
        v.up_cw = (v.up_vis_w + 7) & ~7;
        v.up_deb.w = v.up_cdef.w = v.up_cw;
        v.up_deb.h = v.up_cdef.h = fh - 8 * (v.desc.pad >> 4);
    }
    v.dir_var = c.d_dir_var;
    v.wedge = r->d_wedge;
    v.warp_filter = r->d_warp_filter;
    // where the view lives: again where it was (views with scaled references are refreshed on every submit), else -- the
    // usual case, a job reconstructed on one chain -- in the job's own arena, else in an allocation of its own
    FrameView *d = nullptr;
    bool own = false;
    if (hit) {
        d = hit->d_view;
    } else if (job->bound.empty() && job->d_view_slot) {
        d = job->d_view_slot;
    } else {
        own = true;
        if (hipMalloc(&d, sizeof(FrameView)) != hipSuccess) return ea1::fail(EA1_ENOMEM, "hipMalloc(FrameView) failed");
    }
    // it goes there on the submit stream, ahead of the kernels that read it: staged in the submit's pinned block, pulled
    // with the others by one kernel
    {
        constexpr size_t slot = (sizeof(FrameView) + 255) & ~size_t(255);
        if (!vu.pin.p && !r->pool->take(true, slot * r->chains.size(), &vu.pin)) {
            if (own) (void)hipFree(d);
            return ea1::fail(EA1_ENOMEM, "hipHostMalloc(FrameView staging) failed");
        }
        memcpy(vu.pin.p + vu.used, &v, sizeof(FrameView));
        if (vu.q.args.n == ea1::kPullSegs) {
            const int rc = vu.q.flush();
            if (rc != EA1_OK) {
                if (own) (void)hipFree(d);
                return rc;
            }
        }
        ea1::PullSeg &sg = vu.q.args.seg[vu.q.args.n++];
        sg.dst = d;
        sg.src = vu.pin.p + vu.used;
        sg.bytes = (sizeof(FrameView) + 15) & ~size_t(15);
        vu.used += slot;
    }
    if (hit) hit->scaled_mask = v.scaled_mask;
    else job->bound.push_back(ea1_job::Bound{r, ci, d, own, v.scaled_mask});
    *out = d;
    return EA1_OK;
}

static int submit_impl(ea1_recon *r, ea1_job *const *jobs, unsigned stages, unsigned phases, void *stream);

extern "C" int ea1_recon_submit_batch(ea1_recon *r, ea1_job *const *jobs, unsigned stages, void *stream)
{
    return submit_impl(r, jobs, stages, EA1_PHASE_RECON | EA1_PHASE_FILTER, stream);
}

extern "C" int ea1_recon_submit_phase(ea1_recon *r, ea1_job *const *jobs, unsigned phase, void *stream)
{
    if (phase != EA1_PHASE_RECON && phase != EA1_PHASE_FILTER) return ea1::fail(EA1_EINVAL, "ea1_recon_submit_phase: phase %u", phase);
    return submit_impl(r, jobs, EA1_STAGE_ALL, phase, stream);
}

// one event behind the launches of a submit, shared by its jobs: what ea1_job_destroy parks their memory behind
static void mark_submit(ea1_recon *r, ea1_job *const *jobs, hipStream_t st)
{
    auto mark = std::make_shared<SubmitMark>();
    mark->pool = r->pool;
    {
        std::lock_guard<std::recursive_mutex> lk(r->pool->mu);
        mark->ev = r->pool->event();
    }
    (void)hipEventRecord(mark->ev, st);
    for (size_t c = 0; c < r->chains.size(); c++)
        if (jobs[c]) jobs[c]->last_use = mark;
}

static int submit_impl(ea1_recon *r, ea1_job *const *jobs, unsigned stages, unsigned phases, void *stream)
{
    if (!r || !jobs) return ea1::fail(EA1_EINVAL, "ea1_recon_submit: NULL argument");
    ea1::DeviceGuard g(r->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", r->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = int(r->chains.size());
    ea1::SubmitInfo s;
    memset(&s, 0, sizeof s);
    s.batch.n = n;
    s.wave_counters = r->d_wave_counters;
    s.ssx = r->ssx;
    s.ssy = r->ssy;
    s.width = s.height = 0;   // the frames' own size (all jobs of a submit alike), set below
    bool any = false, any_inter = false;
    hipEvent_t waited_for = nullptr, known_done = nullptr;
    ViewUpload views(st);
    for (int c = 0; c < n; c++) {
        ea1_job *job = jobs[c];
        if (!job) continue;
        if (job->device != r->device || job->view.desc.width > r->width || job->view.desc.height > r->height)
            return ea1::fail(EA1_EINVAL, "ea1_recon_submit: job of chain %d does not belong to this context", c);
        if (!s.width) {
            s.width = int(job->view.desc.width);
            s.height = int(job->view.desc.height);
            s.up_cw = job->view.desc.upscaled_width ? int((job->view.desc.upscaled_width + 7u) & ~7u) : 0;
        } else if (s.width != int(job->view.desc.width) || s.height != int(job->view.desc.height) ||
                   s.up_cw != (job->view.desc.upscaled_width ? int((job->view.desc.upscaled_width + 7u) & ~7u) : 0)) {
            return ea1::fail(EA1_EINVAL, "ea1_recon_submit: chain %d: the frames of one submit must have one size (with super-resolution: one upscaled width in whole 8-sample units)", c);
        }
        for (int k = 0; k < EA1_NUM_REF_SLOTS; k++)
            if ((job->stats.ref_mask >> k) & 1u && !r->chains[size_t(c)].valid[k])
                return ea1::fail(EA1_ESTATE, "ea1_recon_submit: chain %d: the frame predicts from reference slot %d, which holds no picture", c, k);
        if (job->pending) return ea1::fail(EA1_ESTATE, "ea1_recon_submit: job of chain %d was never uploaded", c);
        if (!job->ready_seen && job->ready && job->ready->ev) {
            // the job's upload and derivation run on the pool's streams: order this submit behind them on the device
            // (the jobs of one hand-over batch share the event: one query, one wait)
            if (job->ready->ev == known_done) {
                job->ready_seen = true;
            } else if (job->ready->ev == waited_for) {
                ;
            } else if (event_done(job->ready->ev)) {
                job->ready_seen = true;
                known_done = job->ready->ev;
            } else {
                EA1_HIP(hipStreamWaitEvent(st, job->ready->ev, 0));
                waited_for = job->ready->ev;
            }
        }
        int rc = bind_job(r, c, job, views, &s.batch.v[c]);
        if (rc != EA1_OK) {
            if (views.pin.p) r->pool->give(views.pin, true, st, true);
            return rc;
        }
        for (const ea1_job::Bound &b : job->bound)
            if (b.ctx == r && b.chain == c && b.scaled_mask) s.any_scaled = true;
        s.stats[c] = &job->stats;
        if (!any) {
            s.lr_x0 = job->stats.strip_x0;
            s.lr_x1 = job->stats.strip_x1;
        } else if (s.lr_x0 != job->stats.strip_x0 || s.lr_x1 != job->stats.strip_x1) {
            return ea1::fail(EA1_EINVAL, "ea1_recon_submit: chain %d: the jobs of one submit must cover the same strip of columns", c);
        }
        any = true;
        any_inter |= job->view.desc.is_inter != 0;
    }
    if (!any) return EA1_OK;
    {
        const int rc = views.q.flush();
        if (views.pin.p) r->pool->give(views.pin, true, st, true);   // free again behind the pull (an event on the submit stream)
        if (rc != EA1_OK) return rc;
    }
    // the filters of a strip also cover 16 luma columns on either side (exchanged between the ranks before this
    // phase): deblocking is exact from 8 columns inside that margin on, CDEF from 3 -- what restoration reads
    s.win_x0 = std::max(s.lr_x0 - 16, 0);
    s.win_x1 = std::min(s.lr_x1 + 16, s.width);
    ea1::Prof *prof = r->prof;
    if (phases & EA1_PHASE_RECON) {
        if (any_inter) ea1::launch_inter_pred(s, st, prof);
        ea1::launch_itx(s, st, prof);
        ea1::launch_intra(s, st, prof);
    }
    if (!(phases & EA1_PHASE_FILTER)) {
        mark_submit(r, jobs, st);
        EA1_LAUNCH_CHECK();
        return EA1_OK;
    }
    // each chain has two pictures, "work" (0) and "second" (1).  Reconstruction lands in work; the
    // deblock stage moves the picture to second (its vertical pass is out of place); CDEF writes the
    // other picture; restoration reads both (deblocked + CDEF) and writes the reference slot.
    int deb = 0;  // where the deblocked (pre-CDEF) picture is
    if (stages & EA1_STAGE_DEBLOCK) {
        ea1::launch_deblock(s, st, prof);
        deb = 1;
    }
    int cd = deb;  // where the CDEF output is
    if (stages & EA1_STAGE_CDEF) {
        ea1::launch_cdef(s, st, prof, deb);
        cd = deb ^ 1;
    }
    // super-resolution: both filtered pictures are upscaled; restoration then reads the upscaled pair
    if (s.up_cw) ea1::launch_upscale(s, st, prof, deb, cd);
    if (stages & EA1_STAGE_LR) {
        ea1::launch_lr(s, st, prof, deb, cd);
    } else {
        for (int c = 0; c < n; c++) {
            if (!jobs[c]) continue;
            const Chain &ch = r->chains[size_t(c)];
            const DevPic &src = s.up_cw ? ch.up[1] : (cd ? ch.cdef : ch.work);
            EA1_HIP(hipMemcpyAsync(ch.slots[jobs[c]->view.desc.dst_slot].p[0], src.p[0], r->pic_bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    mark_submit(r, jobs, st);
    EA1_LAUNCH_CHECK();
    for (int c = 0; c < n; c++)
        if (jobs[c]) {
            Chain &ch = r->chains[size_t(c)];
            const int slot = jobs[c]->view.desc.dst_slot;
            ch.valid[slot] = true;
            ch.slot_w[slot] = int(jobs[c]->view.desc.width);
            ch.slot_h[slot] = int(jobs[c]->view.desc.height);
            ch.slot_vw[slot] = jobs[c]->view.vis_w;
            ch.slot_vh[slot] = jobs[c]->view.vis_h;
            ch.slot_pad[slot] = jobs[c]->view.desc.pad;
            if (jobs[c]->view.desc.upscaled_width) {   // kept upscaled: upscaled_width visible samples in whole 8-sample units, nothing padded to the right
                ch.slot_w[slot] = s.up_cw;
                ch.slot_vw[slot] = int(jobs[c]->view.desc.upscaled_width);
                ch.slot_pad[slot] = uint8_t(jobs[c]->view.desc.pad & 0xF0);
            }
        }
    return EA1_OK;
}

extern "C" int ea1_recon_submit_stages(ea1_recon *r, const ea1_job *job, unsigned stages, void *stream)
{
    if (!r || !job) return ea1::fail(EA1_EINVAL, "ea1_recon_submit: NULL argument");
    ea1_job *jobs[ea1::kMaxChains] = {nullptr};
    jobs[0] = const_cast<ea1_job *>(job);
    return ea1_recon_submit_batch(r, jobs, stages, stream);
}

extern "C" int ea1_recon_submit(ea1_recon *r, const ea1_job *job, void *stream)
{
    return ea1_recon_submit_stages(r, job, EA1_STAGE_ALL, stream);
}

extern "C" int ea1_recon_get_picture_chain(ea1_recon *r, int chain, int slot, ea1_picture *out)
{
    if (!r || !out || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS)
        return ea1::fail(EA1_EINVAL, "ea1_recon_get_picture: bad arguments");
    const Chain &c = r->chains[size_t(chain)];
    if (!c.valid[slot]) return ea1::fail(EA1_ESTATE, "ea1_recon_get_picture: slot %d holds no picture", slot);
    const DevPic &p = c.slots[slot];
    for (int pl = 0; pl < 3; pl++) {
        out->data[pl] = p.p[pl];
        out->stride[pl] = size_t(p.stride[pl]);
    }
    out->width = uint32_t(c.slot_w[slot]);
    out->height = uint32_t(c.slot_h[slot]);
    return EA1_OK;
}

extern "C" int ea1_recon_get_picture(ea1_recon *r, int slot, ea1_picture *out) { return ea1_recon_get_picture_chain(r, 0, slot, out); }

extern "C" int ea1_recon_download_chain(ea1_recon *r, int chain, int slot, uint8_t *y, size_t y_stride, uint8_t *u,
                                        uint8_t *v, size_t c_stride, void *stream)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || slot < 0 || slot >= EA1_NUM_REF_SLOTS || !y || !u || !v)
        return ea1::fail(EA1_EINVAL, "ea1_recon_download: bad arguments");
    const Chain &c = r->chains[size_t(chain)];
    if (!c.valid[slot]) return ea1::fail(EA1_ESTATE, "ea1_recon_download: slot %d holds no picture", slot);
    ea1::DeviceGuard g(r->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevPic &p = c.slots[slot];
    uint8_t *dst[3] = {y, u, v};
    const size_t ds[3] = {y_stride, c_stride, c_stride};
    for (int pl = 0; pl < 3; pl++)   // the slot's own picture (frames may be smaller than the context)
        EA1_HIP(hipMemcpy2DAsync(dst[pl], ds[pl], p.p[pl], p.stride[pl], pl ? c.slot_w[slot] >> r->ssx : c.slot_w[slot], pl ? c.slot_h[slot] >> r->ssy : c.slot_h[slot],
                                 hipMemcpyDeviceToHost, st));
    EA1_HIP(hipStreamSynchronize(st));
    return EA1_OK;
}

extern "C" int ea1_host_alloc(size_t bytes, void **out)
{
    if (!out || !bytes) return ea1::fail(EA1_EINVAL, "ea1_host_alloc: bad arguments");
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) return ea1::fail(EA1_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
    pinned_registry().add(*out, bytes);
    return EA1_OK;
}

extern "C" void ea1_host_free(void *p)
{
    if (!p) return;
    pinned_registry().remove(p);
    (void)hipHostFree(p);
}

extern "C" int ea1_recon_picture_layout(const ea1_recon *r, size_t plane_offset[3], size_t plane_stride[3], size_t *bytes)
{
    if (!r || !plane_offset || !plane_stride || !bytes) return ea1::fail(EA1_EINVAL, "ea1_recon_picture_layout: NULL argument");
    const DevPic &p = r->chains[0].slots[0];
    for (int pl = 0; pl < 3; pl++) {
        plane_offset[pl] = size_t(p.p[pl] - p.p[0]);
        plane_stride[pl] = size_t(p.stride[pl]);
    }
    *bytes = r->pic_bytes;
    return EA1_OK;
}

extern "C" int ea1_recon_download_batch_async(ea1_recon *r, const int *slots, uint8_t *const *dst, void *stream)
{
    if (!r || !slots || !dst) return ea1::fail(EA1_EINVAL, "ea1_recon_download_batch_async: NULL argument");
    ea1::DeviceGuard g(r->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t c = 0; c < r->chains.size(); c++) {
        if (slots[c] < 0) continue;
        if (slots[c] >= EA1_NUM_REF_SLOTS || !dst[c]) return ea1::fail(EA1_EINVAL, "ea1_recon_download_batch_async: chain %zu: slot / destination", c);
        const Chain &ch = r->chains[c];
        if (!ch.valid[slots[c]]) return ea1::fail(EA1_ESTATE, "ea1_recon_download_batch_async: chain %zu: slot %d holds no picture", c, slots[c]);
        EA1_HIP(hipMemcpyAsync(dst[c], ch.slots[slots[c]].p[0], r->pic_bytes, hipMemcpyDefault, st));   // dst: pinned host or device memory
    }
    return EA1_OK;
}

extern "C" int ea1_recon_download(ea1_recon *r, int slot, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                                  size_t c_stride, void *stream)
{
    return ea1_recon_download_chain(r, 0, slot, y, y_stride, u, v, c_stride, stream);
}

// ---- tile-column split: column ranges of a chain's pictures <-> packed device buffers -----------------
extern "C" size_t ea1_recon_columns_bytes(const ea1_recon *r, uint32_t w) { return r ? ea1::columns_bytes(int(w), int(r->height)) : 0; }

static int copy_columns(ea1_recon *r, int chain, int pic, uint32_t x, uint32_t w, void *d_buf, bool to_picture, void *stream)
{
    if (!r || !d_buf || chain < 0 || chain >= int(r->chains.size()) || pic < EA1_PIC_WORK || pic >= EA1_NUM_REF_SLOTS)
        return ea1::fail(EA1_EINVAL, "ea1_recon_*_columns: bad arguments");
    if (r->layout != EA1_LAYOUT_420) return ea1::fail(EA1_ENOTSUP, "ea1_recon_*_columns: the tile-column split is 4:2:0 only");
    if (!w || (x & 7) || (w & 7) || uint64_t(x) + w > r->width || (reinterpret_cast<uintptr_t>(d_buf) & 3))
        return ea1::fail(EA1_EINVAL, "ea1_recon_*_columns: columns %u..%u (multiples of 8 inside the frame, buffer 4-byte aligned)", x, x + w);
    Chain &c = r->chains[size_t(chain)];
    if (pic >= 0 && !to_picture && !c.valid[pic]) return ea1::fail(EA1_ESTATE, "ea1_recon_get_columns: slot %d holds no picture", pic);
    ea1::DeviceGuard g(r->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", r->device);
    ea1::launch_copy_columns(pic == EA1_PIC_WORK ? c.work : c.slots[pic], int(x), int(w), static_cast<uint8_t *>(d_buf), to_picture, static_cast<hipStream_t>(stream));
    EA1_LAUNCH_CHECK();
    return EA1_OK;
}

extern "C" int ea1_recon_get_columns(ea1_recon *r, int chain, int pic, uint32_t x, uint32_t w, void *d_buf, void *stream)
{
    return copy_columns(r, chain, pic, x, w, d_buf, false, stream);
}

extern "C" int ea1_recon_put_columns(ea1_recon *r, int chain, int pic, uint32_t x, uint32_t w, const void *d_buf, void *stream)
{
    return copy_columns(r, chain, pic, x, w, const_cast<void *>(d_buf), true, stream);
}

// ---- film grain (easyav1_amd_grain.h) ---------------------------------------------------------
extern "C" int ea1_recon_set_upscale_filter(ea1_recon *r, const int16_t *taps)
{
    if (!r || !taps) return ea1::fail(EA1_EINVAL, "ea1_recon_set_upscale_filter: NULL argument");
    bool any = false;
    for (int ph = 0; ph < 64; ph++) {
        int sum = 0;
        for (int k = 0; k < 8; k++) {
            if (taps[ph * 8 + k] < -128 || taps[ph * 8 + k] > 255) return ea1::fail(EA1_EINVAL, "ea1_recon_set_upscale_filter: tap [%d][%d] = %d", ph, k, taps[ph * 8 + k]);
            sum += taps[ph * 8 + k];
            any |= taps[ph * 8 + k] != 0;
        }
        if (any && sum != 128) return ea1::fail(EA1_EINVAL, "ea1_recon_set_upscale_filter: row %d sums to %d, not 128", ph, sum);
    }
    if (!any) {   // an all-zero table (a blob without it): frames with super-resolution stay refused
        r->upscale_set = false;
        return EA1_OK;
    }
    ea1::DeviceGuard g(r->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", r->device);
    if (!r->d_upscale && hipMalloc(&r->d_upscale, 64 * 8 * sizeof(int16_t)) != hipSuccess) return ea1::fail(EA1_ENOMEM, "hipMalloc(upscale filter) failed");
    if (hipMemcpy(r->d_upscale, taps, 64 * 8 * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess) return ea1::fail(EA1_EHIP, "upscale filter upload failed");
    r->upscale_set = true;
    return EA1_OK;
}

extern "C" int ea1_recon_set_gaussian_sequence(ea1_recon *r, const int16_t *sequence)
{
    if (!r || !sequence) return ea1::fail(EA1_EINVAL, "ea1_recon_set_gaussian_sequence: NULL argument");
    for (int i = 0; i < EA1_GAUSSIAN_SEQUENCE_SIZE; i++)
        if (sequence[i] < -2048 || sequence[i] > 2047) return ea1::fail(EA1_EINVAL, "ea1_recon_set_gaussian_sequence: entry %d = %d is not a 12-bit value", i, sequence[i]);
    r->gaussian.assign(sequence, sequence + EA1_GAUSSIAN_SEQUENCE_SIZE);
    return EA1_OK;
}

extern "C" int ea1_recon_film_grain_batch(ea1_recon *r, const int *slots, const ea1_film_grain *const *params, void *stream)
{
    if (!r || !slots || !params) return ea1::fail(EA1_EINVAL, "ea1_recon_film_grain: NULL argument");
    if (r->gaussian.empty()) return ea1::fail(EA1_ESTATE, "ea1_recon_film_grain: the context has no Gaussian sequence (ea1_recon_set_gaussian_sequence)");
    const int n = int(r->chains.size()), W = int(r->width), H = int(r->height);
    ea1::DeviceGuard g(r->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", r->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t views_bytes = (sizeof(ea1::GrainView) * size_t(n) + 255) & ~size_t(255);
    if (!r->d_grain) {
        r->grain_blob = (ea1::grain_blob_bytes(W, H) + 255) & ~size_t(255);
        const size_t total = views_bytes + r->grain_blob * size_t(n);
        if (hipMalloc(&r->d_grain, total) != hipSuccess) return ea1::fail(EA1_ENOMEM, "hipMalloc(film grain tables) failed");
        r->h_grain.assign(total, 0);
    }
    int active = 0;
    std::vector<int> prepare;   // chains whose templates / tables are to be made
    for (int c = 0; c < n; c++) {
        ea1::GrainView &v = reinterpret_cast<ea1::GrainView *>(r->h_grain.data())[c];
        memset(&v, 0, sizeof v);
        if (slots[c] < 0 || !params[c]) continue;
        if (slots[c] >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "ea1_recon_film_grain: chain %d: slot %d", c, slots[c]);
        Chain &ch = r->chains[size_t(c)];
        if (!ch.valid[slots[c]]) return ea1::fail(EA1_ESTATE, "ea1_recon_film_grain: chain %d: slot %d holds no picture", c, slots[c]);
        // (a picture may be smaller than the context -- a stream that changes its frame size: the synthesis runs on the picture's own size)
        const int PW = ch.slot_w[slots[c]], PH = ch.slot_h[slots[c]];
        if (PW <= 0 || PH <= 0 || PW > W || PH > H) return ea1::fail(EA1_EINVAL, "ea1_recon_film_grain: chain %d: the picture is %dx%d, the context %dx%d", c, PW, PH, W, H);
        const ea1_film_grain &fg = *params[c];
        const int rc = ea1::grain_validate(fg);
        if (rc != EA1_OK) return rc;
        prepare.push_back(c);
        for (int pl = 0; pl < 3; pl++) {
            v.src[pl] = ch.slots[slots[c]].p[pl];
            v.dst[pl] = ch.display.p[pl];
            v.stride[pl] = ch.display.stride[pl];
        }
        v.blob = r->d_grain + views_bytes + r->grain_blob * size_t(c);
        if ((fg.crop & 15) > 7 || (fg.crop >> 4) > 7) return ea1::fail(EA1_EINVAL, "film grain: crop 0x%02x", fg.crop);
        v.cw = PW;
        v.ch = PH;
        const int padw = 8 * (ch.slot_pad[slots[c]] & 15), padh = 8 * (ch.slot_pad[slots[c]] >> 4);   // crop counts from the TRUE coded size
        v.w = PW - padw - (fg.crop & 15);
        v.h = PH - padh - (fg.crop >> 4);
        v.nbx = (v.w + 31) / 32;
        for (int p = 0; p < 2; p++) {
            v.uv_mult[p] = int16_t(int(fg.uv_mult[p]) - 128);
            v.uv_luma_mult[p] = int16_t(int(fg.uv_luma_mult[p]) - 128);
            v.uv_offset[p] = int16_t(int(fg.uv_offset[p]) - 256);
        }
        v.flags = uint8_t((fg.num_y_points ? ea1::kGrainY : 0) | ((fg.num_uv_points[0] || fg.chroma_scaling_from_luma) ? ea1::kGrainCbOn : 0) |
                          ((fg.num_uv_points[1] || fg.chroma_scaling_from_luma) ? ea1::kGrainCrOn : 0) | (fg.overlap_flag ? ea1::kGrainOverlap : 0) |
                          (fg.chroma_scaling_from_luma ? ea1::kGrainFromLuma : 0));
        v.scaling_shift = fg.scaling_shift;
        v.min_value = fg.clip_to_restricted_range ? 16 : 0;
        v.max_luma = fg.clip_to_restricted_range ? 235 : 255;
        v.max_chroma = fg.clip_to_restricted_range ? (fg.matrix_is_identity ? 235 : 240) : 255;
        v.ssx = uint8_t(r->ssx);
        v.ssy = uint8_t(r->ssy);
        ch.display_valid = true;
        active++;
    }
    if (!active) return EA1_OK;
    {
        // the sequential part of the synthesis (LFSR, auto-regression) per picture: ~66 us each on one host
        // thread, independent between chains -> a few threads keep a wide batch from waiting on the host
        auto work = [&](size_t first, size_t step) {
            for (size_t k = first; k < prepare.size(); k += step) {
                const int c = prepare[k];
                const Chain &gc = r->chains[size_t(c)];
                ea1::grain_prepare(*params[c], r->gaussian.data(), gc.slot_w[slots[c]] - 8 * (gc.slot_pad[slots[c]] & 15) - (params[c]->crop & 15),
                                   gc.slot_h[slots[c]] - 8 * (gc.slot_pad[slots[c]] >> 4) - (params[c]->crop >> 4), r->ssx, r->ssy,
                                   r->h_grain.data() + views_bytes + r->grain_blob * size_t(c));
            }
        };
        const size_t nthreads = std::min<size_t>({prepare.size() / 4, size_t(8), size_t(std::max(1u, std::thread::hardware_concurrency()))});
        if (nthreads <= 1) {
            work(0, 1);
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nthreads; t++) pool.emplace_back(work, t, nthreads);
            work(0, nthreads);
            for (std::thread &t : pool) t.join();
        }
    }
    // pageable source: the copy has left the staging buffer when the call returns
    EA1_HIP(hipMemcpyAsync(r->d_grain, r->h_grain.data(), r->h_grain.size(), hipMemcpyHostToDevice, st));
    ea1::launch_film_grain(reinterpret_cast<const ea1::GrainView *>(r->d_grain), n, W, H, r->ssx, r->ssy, st, r->prof, uint64_t(active));
    EA1_HIP(hipGetLastError());
    return EA1_OK;
}

extern "C" int ea1_recon_film_grain_chain(ea1_recon *r, int chain, int slot, const ea1_film_grain *params, void *stream)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || !params) return ea1::fail(EA1_EINVAL, "ea1_recon_film_grain_chain: bad arguments");
    std::vector<int> slots(r->chains.size(), -1);
    std::vector<const ea1_film_grain *> ps(r->chains.size(), nullptr);
    slots[size_t(chain)] = slot;
    ps[size_t(chain)] = params;
    return ea1_recon_film_grain_batch(r, slots.data(), ps.data(), stream);
}

extern "C" int ea1_recon_get_display_chain(ea1_recon *r, int chain, ea1_picture *out)
{
    if (!r || !out || chain < 0 || chain >= int(r->chains.size())) return ea1::fail(EA1_EINVAL, "ea1_recon_get_display: bad arguments");
    const Chain &c = r->chains[size_t(chain)];
    if (!c.display_valid) return ea1::fail(EA1_ESTATE, "ea1_recon_get_display: chain %d has no display picture", chain);
    for (int pl = 0; pl < 3; pl++) {
        out->data[pl] = c.display.p[pl];
        out->stride[pl] = size_t(c.display.stride[pl]);
    }
    out->width = r->width;
    out->height = r->height;
    return EA1_OK;
}

extern "C" int ea1_recon_download_display_chain(ea1_recon *r, int chain, uint8_t *y, size_t y_stride, uint8_t *u, uint8_t *v,
                                                size_t c_stride, void *stream)
{
    if (!r || chain < 0 || chain >= int(r->chains.size()) || !y || !u || !v) return ea1::fail(EA1_EINVAL, "ea1_recon_download_display: bad arguments");
    const Chain &c = r->chains[size_t(chain)];
    if (!c.display_valid) return ea1::fail(EA1_ESTATE, "ea1_recon_download_display: chain %d has no display picture", chain);
    ea1::DeviceGuard g(r->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint8_t *dst[3] = {y, u, v};
    const size_t ds[3] = {y_stride, c_stride, c_stride};
    for (int pl = 0; pl < 3; pl++)
        EA1_HIP(hipMemcpy2DAsync(dst[pl], ds[pl], c.display.p[pl], c.display.stride[pl], c.display.pw(pl), c.display.ph(pl), hipMemcpyDeviceToHost, st));
    EA1_HIP(hipStreamSynchronize(st));
    return EA1_OK;
}

extern "C" int ea1_dsp_itx_add(uint8_t *d_plane, size_t stride, int w, int h, const int16_t *d_coefs,
                               const uint32_t *pos, const uint8_t *txtp, int count, int device, void *stream)
{
    if (!d_plane || !d_coefs || !pos || !txtp || count < 0) return ea1::fail(EA1_EINVAL, "ea1_dsp_itx_add: bad arguments");
    if (count == 0) return EA1_OK;
    ea1::DeviceGuard g(device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", device);
    return ea1::launch_dsp_itx_add(d_plane, stride, w, h, d_coefs, pos, txtp, count, static_cast<hipStream_t>(stream));
}

extern "C" int ea1_recon_warm(int device)
{
    const int ndev = ea1_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return ea1::fail(EA1_ENODEV, "ea1_recon_warm: device %d not available (%d visible)", device, ndev);
    // A key frame and two inter frames with every coding tool, 256 x 192, through a context of their own: the HIP runtime sets a kernel up at its
    // first launch (a few ms each, ~70 kernels on a first frame's path), whatever the frame size.  The frames are DATA shipped beside the library
    // (data/warm_frames.bin: three parsed-frame packets of easyav1_amd_ir_packet.h, each behind its 32-bit size; made by tools/make_warm_frames.py
    // with the test-side stream generator, which is not part of this library); EASYAV1_AMD_WARM_FRAMES names another file.  Without the file
    // nothing is warmed: the kernels load at their first real launch, as they would anyway.
    std::string path;
    if (const char *env = getenv("EASYAV1_AMD_WARM_FRAMES")) {
        path = env;
    } else {
        Dl_info di;
        if (dladdr(reinterpret_cast<const void *>(&ea1_recon_warm), &di) && di.dli_fname) {
            path = di.dli_fname;
            const size_t slash = path.rfind('/');
            path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/data/warm_frames.bin";
        }
    }
    std::vector<uint8_t> blob;
    if (FILE *fp = path.empty() ? nullptr : fopen(path.c_str(), "rb")) {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) blob.insert(blob.end(), buf, buf + n);
        fclose(fp);
    }
    if (blob.empty()) return EA1_OK;
    ea1_recon *r = nullptr;
    int rc = ea1_recon_create(&r, 256, 192, device);
    if (rc != EA1_OK) return rc;
    std::vector<ea1_job *> jobs;
    for (size_t pos = 0; pos + 4 <= blob.size() && rc == EA1_OK;) {
        uint32_t n;
        memcpy(&n, &blob[pos], 4);
        pos += 4;
        if (n > blob.size() - pos) {
            rc = ea1::fail(EA1_EINVAL, "ea1_recon_warm: %s is truncated", path.c_str());
            break;
        }
        ea1_ir_storage *st = nullptr;
        ea1_ir_frame f;
        rc = ea1_ir_packet_parse(&blob[pos], n, &st, &f);
        pos += n;
        if (rc != EA1_OK) break;
        if (f.desc.width > 256 || f.desc.height > 192) rc = ea1::fail(EA1_EINVAL, "ea1_recon_warm: a frame of %s is larger than 256 x 192", path.c_str());
        ea1_job *job = nullptr;
        const ea1_lr_unit *lr[3] = {f.lr[0], f.lr[1], f.lr[2]};
        if (rc == EA1_OK) rc = ea1_job_create(r, &f.desc, f.blocks, f.txs, f.coefs, f.aux, f.cdef_idx, lr, &job);
        if (rc == EA1_OK) {
            jobs.push_back(job);
            rc = ea1_recon_submit(r, job, nullptr);
        }
        ea1_ir_storage_destroy(st);
    }
    (void)ea1_recon_flush(r);
    for (ea1_job *j : jobs) ea1_job_destroy(j);
    ea1_recon_destroy(r);
    return rc;
}

extern "C" int ea1_recon_profile_enable(ea1_recon *r, int on)
{
    if (!r) return ea1::fail(EA1_EINVAL, "ea1_recon_profile_enable: NULL handle");
    ea1::DeviceGuard g(r->device);
    (void)hipDeviceSynchronize();
    delete r->prof;
    r->prof = on ? new (std::nothrow) EventProf() : nullptr;
    return EA1_OK;
}

extern "C" int ea1_recon_profile_read(ea1_recon *r, ea1_prof_entry *out, int cap, int *count)
{
    if (!r || !count || (cap > 0 && !out)) return ea1::fail(EA1_EINVAL, "ea1_recon_profile_read: bad arguments");
    *count = 0;
    if (!r->prof) return ea1::fail(EA1_ESTATE, "ea1_recon_profile_read: profiling is not enabled");
    ea1::DeviceGuard g(r->device);
    EA1_HIP(hipDeviceSynchronize());
    int n = 0;
    for (auto &rec : r->prof->recs) {
        float ms = 0.f;
        EA1_HIP(hipEventElapsedTime(&ms, rec.a, rec.b));
        int k = 0;
        for (; k < n; k++)
            if (!strcmp(out[k].name, rec.name)) break;
        if (k == n) {
            if (n >= cap) continue;
            memset(&out[n], 0, sizeof out[n]);
            strncpy(out[n].name, rec.name, sizeof(out[n].name) - 1);
            n++;
        }
        out[k].ms_total += ms;
        out[k].launches += 1;
        out[k].algorithmic_bytes += rec.bytes;
        r->prof->pool.push_back(rec.a);
        r->prof->pool.push_back(rec.b);
    }
    r->prof->recs.clear();
    *count = n;
    return EA1_OK;
}

#ifdef EA1_FUZZ_HOOK
// tools/fuzz/fuzz_job.cpp (CPU, AddressSanitizer): the host part of ea1_job_create -- validation and the
// derivation of the schedule -- on a context that exists only as a size; the device allocation at
// the end fails without a GPU, after every host phase has run.
extern "C" int ea1_fuzz_job_create_layout(uint32_t width, uint32_t height, int layout, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                          const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3]);
// ... through the entry point that takes a whole parsed frame (its array lengths are checked)
extern "C" int ea1_fuzz_job_prepare_layout(uint32_t width, uint32_t height, int layout, const ea1_ir_frame *frame)
{
    ea1_recon r;
    r.width = width;
    r.height = height;
    r.layout = layout;
    r.ssx = layout == EA1_LAYOUT_444 ? 0 : 1;
    r.ssy = layout == EA1_LAYOUT_420 ? 1 : 0;
    r.warp_filter_set = true;
    r.chains.resize(1);
    ea1_job *job = nullptr;
    const int rc = ea1_job_prepare(&r, frame, &job);
    if (job) ea1_job_destroy(job);
    return rc;
}
extern "C" int ea1_fuzz_job_create(uint32_t width, uint32_t height, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                   const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3])
{
    return ea1_fuzz_job_create_layout(width, height, EA1_LAYOUT_420, desc, blocks, txs, coefs, aux, cdef_idx, lr);
}
extern "C" int ea1_fuzz_job_create_layout(uint32_t width, uint32_t height, int layout, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                          const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3])
{
    ea1_recon r;
    r.width = width;
    r.height = height;
    r.layout = layout;
    r.ssx = layout == EA1_LAYOUT_444 ? 0 : 1;
    r.ssy = layout == EA1_LAYOUT_420 ? 1 : 0;
    r.warp_filter_set = true;
    r.chains.resize(1);
    ea1_job *job = nullptr;
    const int rc = ea1_job_create(&r, desc, blocks, txs, coefs, aux, cdef_idx, lr, &job);
    if (job) ea1_job_destroy(job);
    return rc;
}
#endif
