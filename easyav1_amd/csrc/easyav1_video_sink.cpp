// See easyav1_video_sink.h.  Packets are prepared on helper threads; one issuer thread commits them (copies + derivation
// kernels, several frames per launch) and submits the reconstruction: ONE FRAME OF EVERY GOP CHAIN PER LAUNCH -- closed GOPs
// are independent, a key-frame packet starts the next chain in turn, and the shim reads far enough ahead
// in the file for `chains` GOPs to be in flight.  A finished picture is parked in HBM (one device-to-device copy on the
// reconstruction stream: pictures of GOPs decoded ahead of their turn wait THERE, not in pinned host memory -- pinning a
// gigabyte costs a third of a second) and travels to a small ring of pinned host buffers in PRESENTATION order on a second HIP
// stream, in device layout (one copy per picture), fenced by an event of its own.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>

#include "ea1_common.h"
#include "easyav1_amd_grain.h"
#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_obu.h"
#include "easyav1_amd_video.h"
#include "easyav1_video_sink.h"

#include <dlfcn.h>
#include <sched.h>

namespace ea1 {

namespace {
constexpr size_t kRing = 11;       // finished pictures whose turn has come and that nobody took: the oldest is dropped beyond (src/easyav1.c:41, :2141-2143)
constexpr int kMaxChains = 16;
constexpr int kMaxCommit = 8;      // prepared frames committed with one launch of the derivation kernels
constexpr int kPrepareAhead = 6;   // frames of one chain that may be prepared ahead of the chain's next submit (bounds the arenas held)
constexpr int kHostRing = int(kRing) + 3;   // pinned host buffers: the waiting pictures, the one the caller holds, two in flight

// Queued -> Preparing -> Prepared -> Committed -> Issued; Failed: its hand-over or submit failed; Dead: given up (error, flush)
enum class St { Queued, Preparing, Prepared, Committed, Issued, Failed, Dead };

struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
    bool slab = false;   // part of the slab made at open: freed with it
};

// one submit: a frame of up to `chains` GOP chains and the parking copies of their pictures, one event behind all of it
struct Batch {
    hipEvent_t done = nullptr;   // on the reconstruction stream, behind the reconstruction and the device-to-device copies
    int refs = 0;
};

struct Item {
    uint64_t ticket = 0;
    uint32_t tag = 0;          // the submitter's tag in force when the packet came (video_sink_set_tag)
    const uint8_t *payload = nullptr;
    size_t size = 0;
    uint64_t timestamp = 0;
    bool discard = false;
    bool av1 = false;             // a plain AV1 temporal unit (block-level parser) instead of a parsed-frame payload
    bool parsed = false;          // av1: its symbol parse is over (the next AV1 packet may be parsed)
    bool show_existing = false;   // av1: a show_existing_frame header -- no job, the picture of frame.desc.dst_slot is the output
    bool grain = false;           // av1: film grain is synthesised on the output picture (the reference slot stays clean)
    std::vector<uint8_t> seq_before;   // av1: the sequence header OBU in force in front of this frame (primes a chain's parser that has seen none)
    ea1_film_grain fg;
    int chain = 0;
    St state = St::Queued;
    PinBlock *pin = nullptr;      // parse storage (pinned), owned by VideoSink::pins
    ea1_ir_frame frame;
    ea1_job *job = nullptr;
    int rc = EA1_OK;
    std::string msg;
    unsigned vis_w = 0, vis_h = 0;
    int parked = -1;              // where the finished picture waits in HBM: index into VideoSink::parks
    int buffer = -1;              // pinned host buffer it was sent to (delivery, in stream order): index into VideoSink::buffers
    hipEvent_t landed = nullptr;  // on the copy stream, behind that copy
    Batch *batch = nullptr;
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// EASYAV1_AMD_TRACE: milestones of the pipeline on stderr, in milliseconds since the first of them
bool tracing()
{
    static const bool on = getenv("EASYAV1_AMD_TRACE") != nullptr;
    return on;
}
double trace_ms()
{
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define EA1_TRACE(...)                                   \
    do {                                                 \
        if (tracing()) {                                 \
            fprintf(stderr, "[sink %9.3f ms] ", trace_ms()); \
            fprintf(stderr, __VA_ARGS__);                \
            fputc('\n', stderr);                         \
        }                                                \
    } while (0)

// host CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota when there is one (cgroup v2 cpu.max "quota period",
// v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A container that shows 256 logical CPUs and grants 16 CPUs of quota THROTTLES a process whose runnable
// threads add up to more (every thread of the group stands still for the rest of the 100 ms period): on the MI355X box of this round 24 / 32 helpers
// decode 8 / 14 % fewer frames per second than 16 (profiles/r05_g / r05_h).
int host_cpu_budget()
{
    int n = int(std::max(1u, std::thread::hardware_concurrency()));
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n, CPU_COUNT(&set));
    auto read_two = [](const char *path, long long *a, long long *b) {
        FILE *fp = fopen(path, "r");
        if (!fp) return 0;
        char t0[32] = {}, t1[32] = {};
        const int got = fscanf(fp, "%31s %31s", t0, t1);
        fclose(fp);
        if (got >= 1) *a = strcmp(t0, "max") ? atoll(t0) : -1;
        if (got >= 2) *b = atoll(t1);
        return got;
    };
    long long quota = -1, period = 100000;
    if (read_two("/sys/fs/cgroup/cpu.max", &quota, &period) < 1) {
        long long dummy = 0;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, &dummy) >= 1) (void)read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, &dummy);
    }
    if (quota > 0 && period > 0) n = std::min(n, int(std::max(1ll, (quota + period / 2) / period)));
    return n;
}

int units(int unit, int size)
{
    const int n = (size + (unit >> 1)) / unit;
    return n < 1 ? 1 : n;
}
}  // namespace

struct VideoSink {
    unsigned width = 0, height = 0;   // size of the context: the track's size in whole 8x8 units, at least the first frame's
    unsigned track_w = 0, track_h = 0;
    int device = 0, chains = 1, lookahead = 10, prepare_ahead = kPrepareAhead, cpu_budget = 0;
    ea1_recon *ctx = nullptr;
    hipStream_t recon_st = nullptr, copy_st = nullptr;
    size_t plane_off[3] = {0, 0, 0}, plane_stride[3] = {0, 0, 0}, pic_bytes = 0;
    int layout = EA1_LAYOUT_420;               // chroma layout of the context (of the stream's first frame: ea1_frame_desc.layout)
    bool declared = false;                     // video_sink_declare named layout and size: the first frame a helper happens to prepare does not decide them

    std::mutex mu;
    std::condition_variable cv_work, cv_issue, cv_done;
    std::deque<std::unique_ptr<Item>> items;   // submission order; tickets are consecutive: items[k]->ticket == items[0]->ticket + k
    uint64_t next_ticket = 0;
    uint32_t tag = 0;          // video_sink_set_tag
    uint64_t gop_count = 0;                    // key-frame packets seen: GOP g runs on chain g % chains
    bool any_submitted = false;
    uint64_t waited_through = 0;               // tickets below this have had their turn (video_sink_wait)
    bool stop = false;
    int stage_busy = 0;                        // stage threads inside their HIP calls
    int preparing = 0, allocating = 0;         // helpers inside prepare() / inside an allocation of a picture buffer
    int error_rc = EA1_OK;                     // sticky: a frame failed; every later frame of the stream fails with it
    uint64_t error_ticket = 0;
    std::string error_msg;
    std::vector<std::thread> workers;
    std::thread stages[3];

    std::vector<std::unique_ptr<PinBlock>> pins;   // pinned parse storage, recycled (the blocks themselves never move)
    void *pin_slab = nullptr;                  // the blocks made at open, one allocation
    std::vector<uint8_t *> buffers;            // pinned picture buffers: at most kHostRing
    std::vector<int> free_buffers;
    std::vector<uint8_t *> parks;              // device picture buffers (pic_bytes each): finished pictures until their turn
    std::vector<int> free_parks;
    std::vector<hipEvent_t> free_events;
    std::vector<Batch *> free_batches;
    int shown = -1;                            // buffer of the picture the caller holds

    // plain AV1 packets: the table blob named by EASYAV1_AMD_AV1_TABLES (the library owns no default CDFs / scans / quantiser
    // tables, include/easyav1_amd_obu.h)
    // (one parser PER GOP CHAIN: a closed GOP starts from a key frame that resets every context and reference, so the chains' symbol
    // parses are independent and run on different helpers at once; inside a chain packets are parsed one at a time, in order)
    ea1_av1_tables *tables = nullptr;
    ea1_av1_frame_parser *av1 = nullptr;            // = av1_chain[0]: "plain AV1 is decodable"
    ea1_av1_frame_parser *av1_chain[kMaxChains] = {};
    std::vector<uint8_t> seq_obu;                   // the last sequence header OBU SUBMITTED (items carry the one in force at their position: Item::seq_before)
    bool seq_reduced_still = false;                 // its reduced_still_picture_header
    bool chain_has_seq[kMaxChains] = {};
    double av1_parse_seconds = 0;
    uint64_t av1_frames = 0;
    // ---- several devices (EASYAV1_AMD_DEVICES=0,1,...): this sink is then only a dealer -- it owns one CHILD sink per entry of the list (a
    // device may be named twice: two contexts on it), gives every GOP to the next child in turn (closed GOPs are independent: SURVEY.md 8(e)(i),
    // no collective) and hands the pictures out in stream order.  The children are complete sinks: their own context, streams, stage thread,
    // helpers, pinned ring.  All calls come from one thread at a time (the shim's state lock), so the dealer needs no lock of its own.
    std::vector<VideoSink *> children;
    int deal_to = 0;                 // the child that takes the current GOP
    bool dealt_any = false;
    struct Dealt { int child; uint64_t local; };
    std::deque<Dealt> dealt;         // global ticket - dealt_base -> where it went
    uint64_t dealt_base = 0;
    int taken_from = -1;             // the child whose buffer the caller holds
    // EASYAV1_AMD_STATS: where the pipeline's time goes (printed when the sink is destroyed); all under `mu`
    double st_copy = 0, st_prepare = 0, st_alloc = 0, st_commit = 0, st_submit = 0, st_deliver = 0, st_stage_idle = 0, st_helper_idle = 0;
    uint64_t st_allocs = 0, st_alloc_bytes = 0, st_commits = 0, st_commit_frames = 0, st_submits = 0, st_submit_frames = 0, st_delivers = 0;
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
    if (b->done) s->free_events.push_back(b->done);
    b->done = nullptr;
    s->free_batches.push_back(b);
}

// consumer side, under the lock: gives the item's resources back (its events must have completed or been waited for)
void retire(VideoSink *s, Item &it, bool keep_buffer)
{
    release_batch(s, it.batch);
    it.batch = nullptr;
    if (it.landed) s->free_events.push_back(it.landed);
    it.landed = nullptr;
    if (it.parked >= 0) s->free_parks.push_back(it.parked);
    it.parked = -1;
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

// under the lock: the decoder context, sized for the track and for the first frame (frames may be smaller than the track's size --
// resolution switches, scaled references: the context holds the largest)
int make_context(VideoSink *s, unsigned frame_w, unsigned frame_h)
{
    if (s->ctx) return EA1_OK;
    // (a plain AV1 stream may change its frame size from frame to frame, up to the track's; its frames arrive padded to whole
    // superblocks of up to 128 samples)
    const unsigned sbm = s->tables ? 127u : 0u;
    s->width = std::max((s->track_w + sbm) & ~sbm, frame_w);
    s->height = std::max((s->track_h + sbm) & ~sbm, frame_h);
    EA1_TRACE("creating the context: %ux%u, %d chain(s), layout %d", s->width, s->height, s->chains, s->layout);
    int rc = ea1_recon_create_layout(&s->ctx, s->width, s->height, s->chains, s->device, s->layout);
    if (rc == EA1_OK) rc = ea1_recon_picture_layout(s->ctx, s->plane_off, s->plane_stride, &s->pic_bytes);
    if (rc == EA1_OK && s->tables) rc = ea1_recon_set_gaussian_sequence(s->ctx, ea1_av1_tables_gaussian_sequence(s->tables));
    if (rc == EA1_OK && s->tables) rc = ea1_recon_set_upscale_filter(s->ctx, ea1_av1_tables_upscale_filter(s->tables));   // (all zero: none)
    EA1_TRACE("context created (rc %d)", rc);
    return rc;
}

// a parsed frame's arrays back to back in one block (8-byte aligned each), as ea1_ir_packet_parse_into lays them out
size_t ir_frame_bytes(const ea1_ir_frame &f)
{
    auto up = [](size_t n) { return (n + 7) & ~size_t(7); };
    size_t n = up(size_t(f.desc.num_blocks) * sizeof(ea1_block)) + up(size_t(f.desc.num_tx) * sizeof(ea1_txblock)) + up(size_t(f.desc.num_coefs) * 2) +
               up(f.desc.num_aux) + up(f.num_sb);
    for (int p = 0; p < 3; p++) n += up(size_t(f.lr_count[p]) * sizeof(ea1_lr_unit));
    return n + 64;
}
void copy_ir_frame(const ea1_ir_frame &src, void *mem, ea1_ir_frame *out)
{
    auto up = [](size_t n) { return (n + 7) & ~size_t(7); };
    uint8_t *p = static_cast<uint8_t *>(mem);
    auto put = [&](const void *from, size_t bytes) -> void * {
        void *at = bytes ? p : nullptr;
        if (bytes) memcpy(p, from, bytes);
        p += up(bytes);
        return at;
    };
    *out = src;
    out->blocks = static_cast<const ea1_block *>(put(src.blocks, size_t(src.desc.num_blocks) * sizeof(ea1_block)));
    out->txs = static_cast<const ea1_txblock *>(put(src.txs, size_t(src.desc.num_tx) * sizeof(ea1_txblock)));
    out->coefs = static_cast<const int16_t *>(put(src.coefs, size_t(src.desc.num_coefs) * 2));
    out->aux = static_cast<const uint8_t *>(put(src.aux, src.desc.num_aux));
    out->cdef_idx = static_cast<const uint8_t *>(put(src.cdef_idx, src.num_sb));
    for (int pl = 0; pl < 3; pl++) out->lr[pl] = static_cast<const ea1_lr_unit *>(put(src.lr[pl], size_t(src.lr_count[pl]) * sizeof(ea1_lr_unit)));
}

// the host half of one frame, on a helper thread, outside the lock
void prepare(VideoSink *s, Item &it)
{
    ea1_ir_frame &f = it.frame;
    int rc;
    if (it.av1) {
        // the symbol parse (this packet's turn: pick_work hands AV1 packets out one at a time, in order), then the records move
        // into pinned memory and the parser is free for the next packet while this thread does the hand-over's host half
        ea1_av1_parsed parsed;
        ea1_av1_frame_parser *parser = s->av1_chain[it.chain];
        {
            // the chain's parser needs a sequence header before its first frame
            std::vector<uint8_t> seq;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                if (!s->chain_has_seq[it.chain]) seq = it.seq_before;
                s->chain_has_seq[it.chain] = true;
            }
            if (!seq.empty()) (void)ea1_av1_frame_parser_parse(parser, seq.data(), seq.size(), &parsed);   // EA1_ENOENT: no frame, the header is active
        }
        {
            // tile threads for THIS frame.  When the stream keeps at least half as many GOP chains in work as there are helpers, every frame is parsed on
            // ONE thread: the chains are the parallelism, and a parse that borrows "idle" helpers' CPUs for its tiles collides with them a moment later (on a
            // box whose cgroup grants as many CPUs as there are helpers the whole process is then throttled; measured neutral within +- 4 %,
            // profiles/r05_h_threads_probe.txt).  A stream with FEW chains in work (one open GOP; GOPs longer than the look-ahead) splits
            // the helpers between them: the tiles of its frames side by side.
            int helpers, chains_in_work = 0;
            bool head = false, first_picture = false;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                helpers = int(s->workers.size());
                first_picture = s->st_submits == 0;
                bool seen[kMaxChains] = {};
                bool first = true;
                for (auto &p : s->items) {
                    if (settled(*p)) continue;
                    if (first) head = p.get() == &it, first = false;
                    if (!seen[p->chain]) seen[p->chain] = true, chains_in_work++;
                }
            }
            chains_in_work = std::max(1, chains_in_work);
            int tile_threads = 2 * chains_in_work <= helpers ? helpers / chains_in_work : 1;
            // the frame everybody waits for -- the stream's oldest unsettled one -- always gets four: its chain is the only one whose pictures can leave, and
            // with GOPs of hundreds of frames the first GOP would otherwise come out at a sixteenth of the host's rate (measured neutral on the GOP-16
            // bench stream, +- 4 %: profiles/r05_h_threads_probe.txt)
            if (head) tile_threads = std::max(tile_threads, 4);
            (void)first_picture;
            (void)ea1_av1_frame_parser_set_thread_limit(parser, tile_threads);
        }
        rc = ea1_av1_frame_parser_parse(parser, it.payload, it.size, &parsed);
        if (rc == EA1_OK) {
            // the records' pinned home, chosen now that their size is known: the SMALLEST free block that holds them (a key frame of a 4K
            // stream is 12 MB of records, its inter frames half a megabyte: first fit would grow every block to key-frame size, one pinned
            // allocation after the other -- 1.7 GB in 440 allocations over a 1 500-frame run, each stalling the HIP runtime), else a new block
            // of the next power-of-two size class, which is then there for good
            const size_t need = ir_frame_bytes(parsed.frame);
            PinBlock *best = nullptr;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                for (auto &b : s->pins)
                    if (!b->busy && b->cap >= need && (!best || b->cap < best->cap)) best = b.get();
                if (!best) {
                    s->pins.emplace_back(new PinBlock());
                    best = s->pins.back().get();
                }
                best->busy = true;
                it.pin = best;
            }
            PinBlock &pb = *best;
            if (pb.cap < need) {   // a new block (only this thread touches a busy block)
                size_t cap = size_t(1) << 20;
                while (cap < need) cap <<= 1;
                const double t = now_s();
                rc = ea1_host_alloc(cap, &pb.p);
                if (rc == EA1_OK) pb.cap = cap;
                std::unique_lock<std::mutex> lk(s->mu);
                s->st_alloc += now_s() - t, s->st_allocs++, s->st_alloc_bytes += cap;
            }
            const double t = now_s();
            if (rc == EA1_OK) copy_ir_frame(parsed.frame, pb.p, &f);
            std::unique_lock<std::mutex> lk(s->mu);
            s->st_copy += now_s() - t;
        }
        {
            // (what other threads look at under the lock -- discard: the ring's accounting, the deliverer; found by the ThreadSanitizer build,
            // tools/tsan -- is written under the lock)
            std::unique_lock<std::mutex> lk(s->mu);
            if (rc == EA1_OK && !parsed.show_frame) it.discard = true;   // a frame that is decoded for its successors only
            if (rc == EA1_OK && parsed.show_existing) it.show_existing = true;
            if (rc == EA1_OK && parsed.has_film_grain && parsed.show_frame) {
                it.grain = true;
                it.fg = parsed.film_grain;
            }
            it.parsed = true;
            if (rc == EA1_OK) s->av1_parse_seconds += parsed.parse_seconds, s->av1_frames++;
            s->cv_work.notify_all();
        }
    } else {
        rc = ea1_ir_packet_parse_into(it.payload, it.size, it.pin->p, it.pin->cap, &f);
    }
    if (rc == EA1_OK && it.show_existing) {
        // nothing to reconstruct: the picture is in the context already
        it.vis_w = f.desc.width - 8u * (f.desc.pad & 15u) - (f.desc.crop & 15u);
        it.vis_h = f.desc.height - 8u * (f.desc.pad >> 4) - (f.desc.crop >> 4);
        {
            std::unique_lock<std::mutex> lk(s->mu);
            if (!s->ctx && !s->declared) s->layout = f.desc.layout;
            rc = make_context(s, f.desc.width, f.desc.height);
        }
        it.rc = rc;
        if (rc != EA1_OK) it.msg = ea1_last_error();
        return;
    }
    if (rc == EA1_OK) {
        const ea1_frame_desc &d = f.desc;
        // the two arrays whose length the hand-over derives from the frame size must really be that long
        const uint64_t sbs = uint64_t((d.width + 63) / 64) * ((d.height + 63) / 64);
        // (super-resolution: the picture that comes out, and the restoration units, have the upscaled width)
        const int vis_w = d.upscaled_width ? int(d.upscaled_width) : int(d.width) - 8 * (d.pad & 15) - (d.crop & 15), vis_h = int(d.height) - 8 * (d.pad >> 4) - (d.crop >> 4);
        if (d.upscaled_width && it.grain) it.fg.crop = uint8_t((it.fg.crop & 0xF0) | (((d.upscaled_width + 7u) & ~7u) - d.upscaled_width));   // the grain runs on the upscaled picture
        if (d.cdef_enable && f.num_sb < sbs) rc = fail(EA1_EINVAL, "parsed frame: %u CDEF indices for %llu superblocks", f.num_sb, (unsigned long long)sbs);
        for (int pl = 0; pl < 3 && rc == EA1_OK; pl++) {
            if (!d.lr_type[pl]) continue;
            if (d.lr_unit_log2[pl] < 5 || d.lr_unit_log2[pl] > 8) {
                rc = fail(EA1_EINVAL, "parsed frame: restoration unit size");
                break;
            }
            const int sx = pl && d.layout != EA1_LAYOUT_444 ? 1 : 0, sy = pl && d.layout == EA1_LAYOUT_420 ? 1 : 0, usz = 1 << d.lr_unit_log2[pl];
            const uint64_t need = uint64_t(units(usz, (vis_h + sy) >> sy)) * units(usz, (vis_w + sx) >> sx);
            if (f.lr_count[pl] < need) rc = fail(EA1_EINVAL, "parsed frame: %u restoration units of plane %d, %llu needed", f.lr_count[pl], pl, (unsigned long long)need);
        }
        if (rc == EA1_OK && d.dst_slot >= EA1_NUM_REF_SLOTS) rc = fail(EA1_EINVAL, "parsed frame: destination slot %u", d.dst_slot);
        it.vis_w = unsigned(vis_w);
        it.vis_h = unsigned(vis_h);
    }
    if (rc == EA1_OK) {
        std::unique_lock<std::mutex> lk(s->mu);
        if (!s->ctx && !s->declared) s->layout = f.desc.layout;   // (a frame of another layout than the context's is refused by the hand-over)
        rc = make_context(s, f.desc.width, f.desc.height);
    }
    if (rc == EA1_OK) {
        const double t = now_s();
        rc = ea1_job_prepare(s->ctx, &f, &it.job);
        std::unique_lock<std::mutex> lk(s->mu);
        s->st_prepare += now_s() - t;
    }
    if (it.ticket < 4 || (it.ticket & 63) == 0) EA1_TRACE("ticket %llu prepared (rc %d)", (unsigned long long)it.ticket, rc);
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
    bool av1_turn_taken[kMaxChains] = {};   // an AV1 packet of the chain whose symbol parse is still to come or under way: its later ones wait
    for (auto &p : s->items) {
        Item &it = *p;
        if (settled(it)) continue;
        if (s->error_rc != EA1_OK && it.ticket >= s->error_ticket) continue;   // given up: the issuer marks it
        const int rank = ahead[it.chain]++;
        if (it.av1 && !it.parsed && it.state != St::Failed) {
            const bool mine = !av1_turn_taken[it.chain] && it.state == St::Queued;
            av1_turn_taken[it.chain] = true;
            if (!mine) continue;
            // the parse of a chain is serial: its oldest unparsed packet goes first, ahead of hand-over-only work
            if (rank < s->prepare_ahead * 4) return &it;
            continue;
        }
        if (it.state == St::Queued && rank < s->prepare_ahead && rank < best_rank) {
            best = &it;
            best_rank = rank;
        }
    }
    return best;
}

void worker_loop(VideoSink *s)
{
    (void)pthread_setname_np(pthread_self(), "ea1-helper");   // (top -H, per-thread CPU accounting: tools/easyav1_bench --threads)
    (void)hipSetDevice(s->device);
    // one helper loads the kernels' code objects while the others parse the first frames (the runtime would load them at the first launches:
    // 230 ms between the first submit and the first picture of a 4K stream, profiles/r04_ac_first_frame.txt)
    static std::atomic<uint32_t> warmed{0};
    const uint32_t bit = 1u << (s->device & 31);
    if (!(warmed.fetch_or(bit) & bit)) {
        const double t = now_s();
        (void)ea1_recon_warm(s->device);
        EA1_TRACE("kernels' code objects loaded in %.1f ms", (now_s() - t) * 1e3);
    }
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        if (s->stop) return;
        // picture buffers ahead of need: pinning 12 MB takes milliseconds, which the issuer should not spend between two submits
        // (with several GOPs in flight the pictures of the later ones pile up while the stream warms up)
        if (s->pic_bytes && int(s->buffers.size()) + s->allocating < kHostRing && s->error_rc == EA1_OK && !s->items.empty()) {
            s->allocating++;
            lk.unlock();
            void *mem = nullptr;
            const int arc = ea1_host_alloc(s->pic_bytes, &mem);
            lk.lock();
            s->allocating--;
            if (arc == EA1_OK) {
                s->buffers.push_back(static_cast<uint8_t *>(mem));
                s->free_buffers.push_back(int(s->buffers.size()) - 1);
            }
            s->cv_done.notify_all();
            s->cv_issue.notify_all();   // a delivery may have been waiting for it
            if (arc == EA1_OK) continue;
        }
        Item *it = pick_work(s);
        if (!it) {
            const double t = now_s();
            s->cv_work.wait(lk);
            s->st_helper_idle += now_s() - t;
            continue;
        }
        // parse storage: a free pinned block that is large enough, else a new one (the payload bounds what the arrays of a parsed-frame packet
        // need; a plain AV1 packet says nothing about the size of its records: prepare() picks its block AFTER the parse)
        const size_t need = it->av1 ? 0 : it->size + 64;
        PinBlock *pbp = nullptr;
        if (!it->av1) {
            for (auto &b : s->pins)
                if (!b->busy && b->cap >= need && !pbp) pbp = b.get();
            if (!pbp)
                for (auto &b : s->pins)
                    if (!b->busy && !b->slab && !pbp) pbp = b.get();   // too small: replaced below
            if (!pbp) {
                s->pins.emplace_back(new PinBlock());
                pbp = s->pins.back().get();
            }
            pbp->busy = true;
        }
        static PinBlock none;
        PinBlock &pb = pbp ? *pbp : none;
        it->pin = pbp;
        it->state = St::Preparing;
        s->preparing++;
        lk.unlock();
        int rc = EA1_OK;
        if (pbp && pb.cap < need) {   // only this thread touches a busy block
            if (pb.p) ea1_host_free(pb.p);
            pb.p = nullptr;
            pb.cap = 0;
            const size_t cap = (need + need / 4 + 0xFFFFF) & ~size_t(0xFFFFF);
            const double t = now_s();
            rc = ea1_host_alloc(cap, &pb.p);
            if (rc == EA1_OK) pb.cap = cap;
            std::unique_lock<std::mutex> lk2(s->mu);
            s->st_alloc += now_s() - t, s->st_allocs++, s->st_alloc_bytes += cap;
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

// The three stages behind the helpers.  ONE thread runs them in turn by default: three threads (EASYAV1_AMD_STAGE_THREADS=3), each
// with its own stream, were measured slower (the runtime's calls serialise, and a committer of its own commits frame by frame
// instead of eight at a time): 1080p 3 360 -> 2 780 frames/s with 4 GOPs in flight.
using Lock = std::unique_lock<std::mutex>;

// commit: the copies and the derivation kernels of whatever the helpers have finished, up to kMaxCommit frames a launch
bool stage_commit(VideoSink *s, Lock &lk)
{
    Item *run[kMaxCommit];
    ea1_job *jobs[kMaxCommit];
    int n = 0;
    bool moved = false;
    for (auto &p : s->items)
        if (p->state == St::Prepared && !p->job) {   // show_existing_frame: nothing to commit
            p->state = St::Committed;
            moved = true;
        } else if (p->state == St::Prepared && n < kMaxCommit) {
            run[n] = p.get();
            jobs[n++] = p->job;
        }
    if (!n) {
        if (moved) s->cv_issue.notify_all();
        return moved;
    }
    s->stage_busy++;
    s->st_commits++, s->st_commit_frames += uint64_t(n);
    lk.unlock();
    const int rc = ea1_job_commit(s->ctx, jobs, n);
    const std::string msg = rc != EA1_OK ? ea1_last_error() : "";
    if (run[0]->ticket < 8) EA1_TRACE("committed %d frame(s) from ticket %llu", n, (unsigned long long)run[0]->ticket);
    lk.lock();
    s->stage_busy--;
    for (int k = 0; k < n; k++) {
        run[k]->state = rc == EA1_OK ? St::Committed : St::Failed;
        if (rc != EA1_OK) {
            run[k]->rc = rc;
            run[k]->msg = msg;
        }
    }
    s->cv_done.notify_all();
    s->cv_issue.notify_all();
    return true;
}

// delivery: finished pictures to the pinned ring, strictly in stream order, as far ahead as the ring has room
bool stage_deliver(VideoSink *s, Lock &lk)
{
    Item *next = nullptr;
    for (auto &p : s->items) {
        Item &it = *p;
        if (it.discard || it.state == St::Dead) continue;
        if (it.buffer >= 0 || it.state == St::Failed) continue;   // sent already / about to be given up
        next = &it;
        break;
    }
    hipEvent_t ev = next && next->state == St::Issued && !s->free_buffers.empty() ? take_event(s) : nullptr;
    if (!ev) return false;
    const int buf = s->free_buffers.back();
    s->free_buffers.pop_back();
    uint8_t *dst = s->buffers[size_t(buf)];
    const uint8_t *src = s->parks[size_t(next->parked)];
    hipEvent_t after = next->batch->done;
    s->stage_busy++;
    lk.unlock();
    const bool ok = hipStreamWaitEvent(s->copy_st, after, 0) == hipSuccess && hipMemcpyAsync(dst, src, s->pic_bytes, hipMemcpyDeviceToHost, s->copy_st) == hipSuccess &&
                    hipEventRecord(ev, s->copy_st) == hipSuccess;
    if (!ok) (void)hipStreamSynchronize(s->copy_st);
    lk.lock();
    s->stage_busy--;
    // the item gets its buffer and its event only now: whoever sees them may wait on an event that HAS been recorded
    // (waiting on one that has not returns at once: found as a picture "there" before it had landed)
    next->buffer = buf;
    next->landed = ev;
    if (!ok) {
        next->state = St::Failed;
        next->rc = EA1_EHIP;
        next->msg = "video sink: the copy of a picture to host memory failed";
        s->cv_issue.notify_all();
    }
    s->cv_done.notify_all();
    return true;
}

// errors, parse storage, and the submit: one frame of every chain that has one committed
bool stage_submit(VideoSink *s, Lock &lk)
{
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
    // ---- parse storage of submitted frames goes back as soon as their batch has run (the copies out of it are in front of the
    //      reconstruction): not when the picture is shown, which for a GOP decoded ahead is a long time off ----
    for (auto &p : s->items) {
        Item &it = *p;
        if (it.state == St::Issued && it.pin && it.batch && hipEventQuery(it.batch->done) == hipSuccess) {
            it.pin->busy = false;
            it.pin = nullptr;
            s->cv_work.notify_one();
        }
    }
    (void)hipGetLastError();
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
            it.frame.desc.height == oldest->frame.desc.height && ((it.frame.desc.upscaled_width + 7u) & ~7u) == ((oldest->frame.desc.upscaled_width + 7u) & ~7u))
            head[it.chain] = &it;
    }
    if (!oldest || oldest->state != St::Committed) return false;
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
        b->refs = 1;   // the stage's own reference until the batch is out
    }
    if (!b || !b->done) {
        rc = EA1_EHIP;
        msg = "video sink: no event for a submit";
    }
    // a place in HBM for every picture of the batch that will be shown
    for (int c = 0; c < s->chains && rc == EA1_OK; c++) {
        Item *it = head[c];
        if (!it || it->discard) continue;
        if (s->free_parks.empty()) {
            void *mem = nullptr;
            lk.unlock();
            const bool ok = hipMalloc(&mem, s->pic_bytes) == hipSuccess;
            lk.lock();
            if (!ok) {
                rc = EA1_ENOMEM;
                msg = "video sink: out of device memory for a finished picture";
                break;
            }
            s->parks.push_back(static_cast<uint8_t *>(mem));
            s->free_parks.push_back(int(s->parks.size()) - 1);
        }
        it->parked = s->free_parks.back();
        s->free_parks.pop_back();
    }
    ea1_job *bjobs[kMaxChains] = {};
    int slots[kMaxChains];
    uint8_t *dst[kMaxChains] = {};
    bool any_picture = false;
    struct Grain {
        int chain, slot;
        const ea1_film_grain *fg;
        uint8_t *dst;
    } grains[kMaxChains];
    int ngrain = 0;
    for (int c = 0; c < s->chains; c++) {
        slots[c] = -1;
        Item *it = head[c];
        if (!it) continue;
        bjobs[c] = it->job;
        if (!it->discard && it->parked >= 0) {
            if (it->grain) {   // the grained copy is made in the chain's display picture and parked from there
                grains[ngrain++] = Grain{c, int(it->frame.desc.dst_slot), &it->fg, s->parks[size_t(it->parked)]};
                continue;
            }
            slots[c] = it->frame.desc.dst_slot;
            dst[c] = s->parks[size_t(it->parked)];
            any_picture = true;
        }
    }
    s->stage_busy++;
    lk.unlock();
    bool any_job = false;
    for (int c = 0; c < s->chains; c++) any_job |= bjobs[c] != nullptr;
    if (rc == EA1_OK && any_job) {   // (a batch of show_existing_frame headers only parks pictures)
        rc = ea1_recon_submit_batch(s->ctx, bjobs, EA1_STAGE_ALL, s->recon_st);
        if (rc != EA1_OK) msg = ea1_last_error();
    }
    // the pictures leave their reference slots on the SAME stream (the next submit may overwrite a slot: stream order keeps
    // the copy in front of it)
    if (rc == EA1_OK && any_picture) {
        rc = ea1_recon_download_batch_async(s->ctx, slots, dst, s->recon_st);
        if (rc != EA1_OK) msg = ea1_last_error();
    }
    for (int k = 0; k < ngrain && rc == EA1_OK; k++) {
        ea1_picture disp;
        rc = ea1_recon_film_grain_chain(s->ctx, grains[k].chain, grains[k].slot, grains[k].fg, s->recon_st);
        if (rc == EA1_OK) rc = ea1_recon_get_display_chain(s->ctx, grains[k].chain, &disp);
        if (rc != EA1_OK) msg = ea1_last_error();
        // (the display picture has the layout of every picture of the context: one block of pic_bytes from its luma plane on)
        else if (hipMemcpyAsync(grains[k].dst, disp.data[0], s->pic_bytes, hipMemcpyDeviceToDevice, s->recon_st) != hipSuccess) {
            rc = EA1_EHIP;
            msg = "video sink: parking a grained picture failed";
        }
    }
    if (rc == EA1_OK && hipEventRecord(b->done, s->recon_st) != hipSuccess) {
        rc = EA1_EHIP;
        msg = "hipEventRecord failed";
    }
    if (rc != EA1_OK) (void)hipStreamSynchronize(s->recon_st);   // whatever did get out must not outlive the resources the failed items give back
    if (oldest->ticket < 8 || (oldest->ticket & 63) == 0) {
        int nb = 0;
        for (int c = 0; c < s->chains; c++) nb += head[c] != nullptr;
        EA1_TRACE("submitted a batch of %d frame(s), oldest ticket %llu (rc %d)", nb, (unsigned long long)oldest->ticket, rc);
    }
    lk.lock();
    s->stage_busy--;
    for (int c = 0; c < s->chains; c++) {
        Item *it = head[c];
        if (!it) continue;
        if (rc == EA1_OK) {
            it->state = St::Issued;
            it->batch = b;
            b->refs++;
            // the job has done its duty (its arena returns to the pool behind the submit: ea1_job_destroy never synchronises);
            // frames decoded a GOP ahead of their turn would otherwise hold their arenas until they are shown
            if (it->job) ea1_job_destroy(it->job);
            it->job = nullptr;
        } else {
            it->state = St::Failed;
            it->rc = rc;
            it->msg = msg;
            if (it->parked >= 0) s->free_parks.push_back(it->parked);
            it->parked = -1;
        }
    }
    if (b) release_batch(s, b);   // the stage's own reference
    s->cv_done.notify_all();
    s->cv_issue.notify_all();     // the deliverer: a picture is parked
    s->cv_work.notify_all();      // chains moved on: frames further ahead may be prepared now
    return true;
}

enum Role { kAll = -1, kSubmit = 0, kCommit = 1, kDeliver = 2 };

void stage_loop(VideoSink *s, int role)
{
    (void)pthread_setname_np(pthread_self(), "ea1-stage");
    (void)hipSetDevice(s->device);
    Lock lk(s->mu);
    for (;;) {
        if (s->stop) return;
        // one thread: in this order of priority, starting over after every piece of work (a submit parks up to `chains` pictures:
        // all of them must be able to leave before the next one)
        double t = now_s();
        if ((role == kAll || role == kCommit) && stage_commit(s, lk)) {
            s->st_commit += now_s() - t;
            continue;
        }
        if ((role == kAll || role == kDeliver) && stage_deliver(s, lk)) {
            s->st_deliver += now_s() - t, s->st_delivers++;
            continue;
        }
        if ((role == kAll || role == kSubmit) && stage_submit(s, lk)) {
            s->st_submit += now_s() - t, s->st_submits++;
            continue;
        }
        t = now_s();
        if (!s->stop) s->cv_issue.wait(lk);
        s->st_stage_idle += now_s() - t;
    }
}

// the item's work on the device has finished: the picture is in its host buffer (shown frames), the reconstruction is done (discarded ones)
bool batch_landed(const Item &it)
{
    hipEvent_t ev = it.discard ? (it.batch ? it.batch->done : nullptr) : it.landed;
    const bool ok = ev && hipEventQuery(ev) == hipSuccess;
    (void)hipGetLastError();   // hipEventQuery reports "not ready" through the sticky error too
    return ok;
}

void wait_item(const Item &it)
{
    if (it.landed) (void)hipEventSynchronize(it.landed);
    else if (it.batch && it.batch->done) (void)hipEventSynchronize(it.batch->done);
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
        if (f.state == St::Issued && !(is_last && keep_last_buffer)) wait_item(f);
        retire(s, f, is_last && keep_last_buffer);
        s->items.pop_front();
        if (is_last) break;
    }
    s->cv_issue.notify_all();   // a host buffer may have come free: the next picture can travel
}

}  // namespace

static int create_on(VideoSink **out, unsigned width, unsigned height, int device, int share);

int video_sink_create(VideoSink **out, unsigned width, unsigned height)
{
    *out = nullptr;
    // the reference's settings struct cannot grow (ABI): the devices, the helper count and the GOP look-ahead come from the environment
    const char *list = getenv("EASYAV1_AMD_DEVICES");
    if (!list || !strchr(list, ',')) {
        const char *dev = list && *list ? list : getenv("EASYAV1_AMD_DEVICE");
        return create_on(out, width, height, dev ? atoi(dev) : 0, 1);
    }
    std::vector<int> devs;
    for (const char *p = list; *p;) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p) return fail(EA1_EINVAL, "video sink: EASYAV1_AMD_DEVICES=%s", list);
        devs.push_back(int(v));
        p = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') return fail(EA1_EINVAL, "video sink: EASYAV1_AMD_DEVICES=%s", list);
    }
    if (devs.size() > 16) return fail(EA1_EINVAL, "video sink: EASYAV1_AMD_DEVICES names %zu contexts (at most 16)", devs.size());
    auto *s = new (std::nothrow) VideoSink();
    if (!s) return fail(EA1_ENOMEM, "out of memory");
    for (int d : devs) {
        VideoSink *c = nullptr;
        const int rc = create_on(&c, width, height, d, int(devs.size()));
        if (rc != EA1_OK) {
            video_sink_destroy(s);
            return rc;
        }
        s->children.push_back(c);
    }
    s->lookahead = 0;
    for (VideoSink *c : s->children) s->lookahead += c->lookahead;
    s->lookahead = std::min(s->lookahead, 4096);
    EA1_TRACE("dealer over %zu device contexts, look-ahead %d packets", s->children.size(), s->lookahead);
    *out = s;
    return EA1_OK;
}

// one complete sink on one device; `share`: how many such sinks split the host between them (helper threads)
static int create_on(VideoSink **out, unsigned width, unsigned height, int device, int share)
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
    s->device = device;
    if (s->device < 0 || s->device >= ndev) {
        const int bad = s->device;
        delete s;
        return fail(EA1_ENODEV, "video sink: device %d named (EASYAV1_AMD_DEVICE / EASYAV1_AMD_DEVICES), %d device(s) present", bad, ndev);
    }
    // GOP chains in flight: a launch carries one frame of each (a single 1080p frame is a few microseconds of HBM traffic: the
    // launches dominate).  What it costs: the pictures of the later GOPs wait until their turn --
    // in HBM, about (chains - 1) x GOP / 2 of them.  Default: 16 up to 1080p, 8 above (16 / 8 against 8 / 4: 1080p 4 660 -> 7 460 frames/s behind the start-up, 4K 2 770 -> 4 240); EASYAV1_AMD_GOPS=1 switches it off.
    // With the table blob given (plain AV1: the symbol parse on the helper threads is the bound, one parser per chain) twice as many, up to
    // kMaxChains: the chains' parses run side by side, 16 chains on 16 helpers decode 2 100 1080p frames/s where 8 on 8 reach 1 200.
    // the AV1 specification's tables (default CDFs, scans, quantisers ...): the blob named by EASYAV1_AMD_AV1_TABLES, else the one that ships
    // beside this library (data/av1_spec_tables.bin: assembled by tests/golden/make_av1_tables.py, accepted by bit-exact pictures beside dav1d)
    std::string tables_path;
    if (const char *path = getenv("EASYAV1_AMD_AV1_TABLES")) {
        tables_path = path;
    } else {
        Dl_info di;
        if (dladdr(reinterpret_cast<const void *>(&ea1::video_sink_create), &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
            const std::string cand = dir + "/data/av1_spec_tables.bin";
            if (FILE *fp = fopen(cand.c_str(), "rb")) {
                fclose(fp);
                tables_path = cand;
            }
        }
    }
    const bool plain_av1 = !tables_path.empty();
    const char *gops = getenv("EASYAV1_AMD_GOPS");
    s->chains = gops ? atoi(gops) : (uint64_t(s->track_w) * s->track_h <= 1920ull * 1088 ? 16 : 8) * (plain_av1 ? 2 : 1);
    s->chains = std::min(std::max(s->chains, 1), kMaxChains);
    const char *la = getenv("EASYAV1_AMD_LOOKAHEAD");
    s->lookahead = la ? atoi(la) : (s->chains > 1 ? 64 * s->chains : 10);   // packets the shim reads ahead of the position (the reference: 10)
    s->lookahead = std::min(std::max(s->lookahead, 10), 4096);
    // plain AV1 needs the specification's tables, which come from outside (include/easyav1_amd_obu.h): without them such a track
    // yields its headers and no pictures, as before
    if (plain_av1) {
        const char *path = tables_path.c_str();
        std::vector<uint8_t> blob;
        if (FILE *fp = fopen(path, "rb")) {
            uint8_t buf[65536];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, fp)) > 0) blob.insert(blob.end(), buf, buf + n);
            fclose(fp);
        }
        int rc = blob.empty() ? fail(EA1_EINVAL, "video sink: the AV1 table blob %s cannot be read", path) : ea1_av1_tables_load(blob.data(), blob.size(), &s->tables);
        for (int c = 0; c < s->chains && rc == EA1_OK; c++) rc = ea1_av1_frame_parser_create(s->tables, &s->av1_chain[c]);
        // the tiles of a frame side by side (dav1d's tile tasks): what one GOP chain alone can use of the host -- a stream whose GOPs are longer than
        // the look-ahead has ONE chain in flight, and its key frames carry most of the bits.  (The workers sleep unless a frame has several tiles.)
        if (!getenv("EASYAV1_AMD_TILE_THREADS")) {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const int tile_threads = hw >= 16 ? 4 : (hw >= 4 ? 2 : 1);
            for (int c = 0; c < s->chains && rc == EA1_OK; c++) rc = ea1_av1_frame_parser_set_threads(s->av1_chain[c], tile_threads);
        }
        s->av1 = s->av1_chain[0];
        if (rc != EA1_OK) {
            for (int c = 0; c < s->chains; c++) ea1_av1_frame_parser_destroy(s->av1_chain[c]);
            if (s->tables) ea1_av1_tables_destroy(s->tables);
            delete s;
            return rc;
        }
    }
    DeviceGuard g(s->device);
    if (hipStreamCreateWithFlags(&s->recon_st, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s->copy_st, hipStreamNonBlocking) != hipSuccess) {
        if (s->recon_st) (void)hipStreamDestroy(s->recon_st);
        delete s;
        return fail(EA1_EHIP, "hipStreamCreate failed");
    }
    // plain AV1: a chain must be able to parse through its GOP while another chain's frames are the stream's oldest (submits wait for the oldest
    // unsubmitted frame): 6 frames ahead left the helpers idle 40 % of the time on a GOP-16 4K stream, 16 ahead 595 -> 893 frames/s
    // (profiles/r04_h_sink_stats.txt)
    if (s->av1) s->prepare_ahead = 16;
    if (const char *pa = getenv("EASYAV1_AMD_PREPARE_AHEAD")) s->prepare_ahead = std::min(std::max(atoi(pa), 1), 64);
    const char *thr = getenv("EASYAV1_AMD_THREADS");
    // plain AV1 (the symbol parse is the work): up to 16, and no more than the CPUs this process may really use (host_cpu_budget: a group that asks for
    // more than its quota is throttled as a whole -- 24 / 32 helpers on a 16-CPU share lose 8 / 14 %; 14 / 15 / 16 are within the run-to-run spread of
    // +- 4 %, 13 and fewer lose: profiles/r05_h_threads_probe.txt)
    const int budget = host_cpu_budget();
    int nthreads = thr ? atoi(thr) : (s->tables ? std::min(16, std::max(2, budget)) : std::min(8, budget));
    s->cpu_budget = budget;
    if (!thr && share > 1) nthreads = std::max(4, nthreads * 2 / share);   // several sinks (EASYAV1_AMD_DEVICES) split about twice one sink's helpers
    nthreads = std::min(std::max(nthreads, 1), 32);
    try {
        const char *st = getenv("EASYAV1_AMD_STAGE_THREADS");
        if (st && atoi(st) == 3) {
            for (int r = 0; r < 3; r++) s->stages[r] = std::thread(stage_loop, s, r);
        } else {
            s->stages[0] = std::thread(stage_loop, s, int(kAll));
        }
        for (int k = 0; k < nthreads; k++) s->workers.emplace_back(worker_loop, s);
    } catch (...) {
        video_sink_destroy(s);
        return fail(EA1_ENOMEM, "video sink: could not start its threads");
    }
    EA1_TRACE("sink created: device %d, %d chain(s), %d helper thread(s) of a budget of %d CPUs, look-ahead %d packets", s->device, s->chains, nthreads, budget, s->lookahead);
    *out = s;
    return EA1_OK;
}

int video_sink_lookahead(const VideoSink *s) { return s ? s->lookahead : 10; }
int video_sink_chains(const VideoSink *s)
{
    if (!s) return 1;
    if (s->children.empty()) return s->chains;
    int n = 0;
    for (const VideoSink *c : s->children) n += c->chains;
    return n;
}

// What "opening the decoder" costs, done at once (the reference's dav1d_open sits in easyav1_init_* too, src/easyav1.c:1405): the
// context, the pinned ring of picture buffers and the parse storage -- pinning memory takes a millisecond per few megabytes and
// stalls other HIP calls meanwhile, which a running pipeline pays for dearly.
void video_sink_declare(VideoSink *s, unsigned max_w, unsigned max_h, int layout)
{
    if (!s) return;
    for (VideoSink *c : s->children) video_sink_declare(c, max_w, max_h, layout);
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->ctx) return;
    s->layout = layout;
    s->declared = true;
    s->track_w = std::max(s->track_w, (max_w + 7) & ~7u);
    s->track_h = std::max(s->track_h, (max_h + 7) & ~7u);
}

// would the context this sink has (or will make from what was declared) hold the frames of a sequence header with this chroma layout and these largest
// frame dimensions?  The shim asks before it queues a packet that brings a new sequence header: one that does not fit gets a sink of its own
bool video_sink_fits(VideoSink *s, unsigned max_w, unsigned max_h, int layout)
{
    if (!s) return true;
    if (!s->children.empty()) {
        for (VideoSink *c : s->children)
            if (!video_sink_fits(c, max_w, max_h, layout)) return false;
        return true;
    }
    std::unique_lock<std::mutex> lk(s->mu);
    if (!s->ctx && !s->declared) return true;   // nothing decided yet: the first frame decides
    if (layout != s->layout) return false;
    // (frames arrive padded to whole superblocks of up to 128 samples: make_context)
    const unsigned sbm = s->tables ? 127u : 7u;
    const unsigned w = s->ctx ? s->width : (s->track_w + sbm) & ~sbm, h = s->ctx ? s->height : (s->track_h + sbm) & ~sbm;
    return ((max_w + sbm) & ~sbm) <= w && ((max_h + sbm) & ~sbm) <= h;
}

int video_sink_open(VideoSink *s, unsigned frame_w, unsigned frame_h, size_t payload_bytes, int layout)
{
    if (!s) return fail(EA1_EINVAL, "video sink: NULL argument");
    if (!s->children.empty()) {
        for (VideoSink *c : s->children) {
            const int rc = video_sink_open(c, frame_w, frame_h, payload_bytes, layout);
            if (rc != EA1_OK) return rc;
        }
        return EA1_OK;
    }
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    if (!s->ctx) s->layout = layout;
    int rc = make_context(s, frame_w, frame_h);
    if (rc != EA1_OK) return rc;
    // (payload_bytes = 0: the context only -- a plain AV1 track opened from its sequence header; the helpers pin picture buffers and parse
    // blocks as they go)
    while (payload_bytes && int(s->buffers.size()) < kHostRing) {
        void *mem = nullptr;
        rc = ea1_host_alloc(s->pic_bytes, &mem);
        if (rc != EA1_OK) return rc;
        s->buffers.push_back(static_cast<uint8_t *>(mem));
        s->free_buffers.push_back(int(s->buffers.size()) - 1);
    }
    if (!s->pin_slab && payload_bytes) {
        size_t cap = (payload_bytes + payload_bytes / 2 + 64 + 0xFFFF) & ~size_t(0xFFFF);   // later frames may be larger than the first
        if (s->av1) {
            // plain AV1: the packet's size says nothing about its records'.  Blocks for ordinary inter frames (a fifth of a byte per sample, at
            // least 1 MB); the few key-frame-sized ones are made on demand, once (prepare(): best fit, power-of-two classes)
            cap = size_t(1) << 20;
            while (cap < size_t(frame_w) * frame_h / 5) cap <<= 1;
        }
        const int n = s->chains * s->prepare_ahead + kMaxCommit + int(s->workers.size());
        if (ea1_host_alloc(cap * size_t(n), &s->pin_slab) == EA1_OK)   // (not fatal: the helpers allocate blocks as they go)
            for (int k = 0; k < n; k++) {
                s->pins.emplace_back(new PinBlock());
                s->pins.back()->p = static_cast<uint8_t *>(s->pin_slab) + size_t(k) * cap;
                s->pins.back()->cap = cap;
                s->pins.back()->slab = true;
            }
    }
    EA1_TRACE("opened: %zu picture buffers of %zu bytes, %zu parse blocks", s->buffers.size(), s->pic_bytes, s->pins.size());
    return EA1_OK;
}

void video_sink_flush(VideoSink *s)
{
    if (!s) return;
    if (!s->children.empty()) {
        for (VideoSink *c : s->children) video_sink_flush(c);
        s->dealt_base += s->dealt.size();
        s->dealt.clear();
        s->deal_to = 0, s->dealt_any = false, s->taken_from = -1;
        return;
    }
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    // nothing new starts: the helpers hold, the issuer gives up whatever it has not submitted
    s->error_rc = EA1_ESTATE;
    s->error_ticket = 0;
    s->cv_issue.notify_all();
    s->cv_done.wait(lk, [&] {
        if (s->preparing || s->allocating || s->stage_busy) return false;
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
    lk.unlock();
    if (s->ctx) ea1_recon_flush(s->ctx);
    for (int c = 0; c < kMaxChains; c++)
        if (s->av1_chain[c]) ea1_av1_frame_parser_reset(s->av1_chain[c]);   // the next packet is a key frame (the shim seeks to one)
}

void video_sink_destroy(VideoSink *s)
{
    if (s && !s->children.empty()) {
        for (VideoSink *c : s->children) video_sink_destroy(c);
        s->children.clear();
        delete s;
        return;
    }
    if (!s) return;
    video_sink_flush(s);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv_work.notify_all();
    s->cv_issue.notify_all();
    for (std::thread &t : s->stages)
        if (t.joinable()) t.join();
    for (std::thread &t : s->workers)
        if (t.joinable()) t.join();
    DeviceGuard g(s->device);
    if (s->ctx) ea1_recon_destroy(s->ctx);
    for (uint8_t *b : s->buffers) ea1_host_free(b);
    for (uint8_t *b : s->parks) (void)hipFree(b);
    for (auto &p : s->pins)
        if (p->p && !p->slab) ea1_host_free(p->p);
    if (s->pin_slab) ea1_host_free(s->pin_slab);
    for (hipEvent_t e : s->free_events) (void)hipEventDestroy(e);
    for (Batch *b : s->free_batches) delete b;
    if (s->recon_st) (void)hipStreamDestroy(s->recon_st);
    if (s->copy_st) (void)hipStreamDestroy(s->copy_st);
    if (s->av1) {
        if (getenv("EASYAV1_AMD_STATS"))
            fprintf(stderr, "[sink stats] helpers: parse %.3f s (%llu frames), copy to pinned %.3f s, job_prepare %.3f s, pinned allocs %llu (%.1f MB, %.3f s), idle %.3f s | "
                            "stage thread: commit %.3f s (%llu calls, %llu frames), submit %.3f s (%llu), deliver %.3f s (%llu), idle %.3f s\n",
                    s->av1_parse_seconds, (unsigned long long)s->av1_frames, s->st_copy, s->st_prepare, (unsigned long long)s->st_allocs, s->st_alloc_bytes / 1e6, s->st_alloc,
                    s->st_helper_idle, s->st_commit, (unsigned long long)s->st_commits, (unsigned long long)s->st_commit_frames, s->st_submit, (unsigned long long)s->st_submits,
                    s->st_deliver, (unsigned long long)s->st_delivers, s->st_stage_idle);
        if (tracing() && s->av1_frames)
            fprintf(stderr, "[sink] AV1 symbol parse: %llu frames, %.3f ms per frame on one host thread (%.1f frames/s)\n", (unsigned long long)s->av1_frames,
                    1e3 * s->av1_parse_seconds / double(s->av1_frames), double(s->av1_frames) / s->av1_parse_seconds);
        for (int c = 0; c < kMaxChains; c++) ea1_av1_frame_parser_destroy(s->av1_chain[c]);
    }
    if (s->tables) ea1_av1_tables_destroy(s->tables);
    delete s;
}

bool video_sink_parses_av1(const VideoSink *s) { return s && (s->children.empty() ? s->av1 != nullptr : video_sink_parses_av1(s->children[0])); }
void video_sink_av1_stats(const VideoSink *s, uint64_t *frames, double *seconds)
{
    if (s && !s->children.empty()) {
        *frames = 0, *seconds = 0;
        for (const VideoSink *c : s->children) {
            uint64_t f;
            double t;
            video_sink_av1_stats(c, &f, &t);
            *frames += f, *seconds += t;
        }
        return;
    }
    *frames = s ? s->av1_frames : 0;
    *seconds = s ? s->av1_parse_seconds : 0;
}

static int submit_item(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool keyframe, bool discard, bool av1, uint64_t *ticket,
                       const std::vector<uint8_t> *seq_before = nullptr);
// the dealer: which child takes this packet (a GOP start moves on to the next child), and the book-keeping of its ticket
static VideoSink *deal(VideoSink *s, bool gop_start)
{
    if (gop_start && s->dealt_any) s->deal_to = (s->deal_to + 1) % int(s->children.size());
    s->dealt_any = true;
    return s->children[size_t(s->deal_to)];
}
static void dealt_ticket(VideoSink *s, uint64_t local, uint64_t *ticket)
{
    // (tickets nobody will wait for any more are dropped from the front: the shim waits for them in order)
    while (s->dealt.size() > 16384) s->dealt.pop_front(), s->dealt_base++;
    s->dealt.push_back(VideoSink::Dealt{s->deal_to, local});
    if (ticket) *ticket = s->dealt_base + s->dealt.size() - 1;
}

int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool keyframe, bool discard, uint64_t *ticket)
{
    if (s && !s->children.empty()) {
        VideoSink *c = deal(s, keyframe);
        uint64_t local = 0;
        const int rc = submit_item(c, payload, size, timestamp, keyframe, discard, false, &local);
        if (rc == EA1_OK) dealt_ticket(s, local, ticket);
        return rc;
    }
    return submit_item(s, payload, size, timestamp, keyframe, discard, false, ticket);
}
// A temporal unit may hold hidden frames in front of its shown frame (alternate references), or a lone show_existing_frame header:
// every frame becomes an item of its own -- [the OBUs in front of it, the frame] -- so that the pipeline sees one frame per ticket.
// Frame boundaries need no symbol decoding: an OBU_FRAME is a whole frame, an OBU_FRAME_HEADER starts one (and is one when its first
// bit, show_existing_frame, is set) that its tile groups complete.
static void split_unit(const uint8_t *data, size_t size, std::vector<size_t> *ends)
{
    ends->clear();
    bool in_frame = false, seen_tiles = false;
    for (size_t pos = 0; pos < size;) {
        ea1_obu o;
        if (ea1_av1_next_obu(data + pos, size - pos, &o) != EA1_OK) break;   // the parser will say what is wrong with it
        if (o.type == EA1_OBU_FRAME) {
            if (in_frame) ends->push_back(pos);
            ends->push_back(pos + o.total_size);
            in_frame = false;
        } else if (o.type == EA1_OBU_FRAME_HEADER) {
            if (in_frame && seen_tiles) {
                ends->push_back(pos);
                in_frame = false;
            }
            if (!in_frame) {
                if (o.payload_size && (o.payload[0] & 0x80)) {
                    ends->push_back(pos + o.total_size);
                } else {
                    in_frame = true;
                    seen_tiles = false;
                }
            }
        } else if (o.type == EA1_OBU_TILE_GROUP) {
            seen_tiles = true;
        }
        pos += o.total_size;
    }
    if (ends->empty() || in_frame || ends->back() < size) {
        if (!ends->empty() && !in_frame) ends->back() = size;   // trailing OBUs that belong to no frame go with the last one
        else ends->push_back(size);
    }
}

// does the first frame of [data, data + size) start a closed GOP?  Only a SHOWN KEY_FRAME resets every reference and context (AV1
// specification 7.20 / 7.21; a key frame with show_frame = 0, an intra-only frame and show_existing_frame of a key frame keep
// references alive).  Read from the BITSTREAM: the first bits of the frame header are show_existing_frame f(1), frame_type f(2),
// show_frame f(1) unless the sequence header says reduced_still_picture_header (then every frame is a shown key frame).  The
// container's key flag is not trusted (a BlockGroup without ReferenceBlock on an inter frame, a SimpleBlock that lost its flag):
// dav1d, behind the reference's decode_video (src/easyav1.c:2381-2428), never sees it either.
static bool starts_gop(const uint8_t *data, size_t size, bool reduced_still)
{
    for (size_t pos = 0; pos < size;) {
        ea1_obu o;
        if (ea1_av1_next_obu(data + pos, size - pos, &o) != EA1_OK) return false;
        if (o.type == EA1_OBU_FRAME || o.type == EA1_OBU_FRAME_HEADER) {
            if (reduced_still) return true;
            if (!o.payload_size) return false;
            const unsigned b = o.payload[0];
            const bool show_existing = b & 0x80, show_frame = b & 0x10;
            const unsigned frame_type = (b >> 5) & 3;
            return !show_existing && frame_type == 0 && show_frame;
        }
        pos += o.total_size;
    }
    return false;
}

int video_sink_submit_av1(VideoSink *s, const uint8_t *unit, size_t size, uint64_t timestamp, bool keyframe, bool discard, uint64_t *ticket)
{
    (void)keyframe;   // the container's opinion: see starts_gop()
    if (s && !s->children.empty()) {
        if (!unit) return fail(EA1_EINVAL, "video sink: NULL argument");
        // the unit goes where its FIRST frame belongs (a shown key frame opens a GOP); every child keeps the sequence header in force
        // (a child may see its first GOP long after the header was sent)
        for (size_t pos = 0; pos < size;) {
            ea1_obu o;
            if (ea1_av1_next_obu(unit + pos, size - pos, &o) != EA1_OK) break;
            if (o.type == EA1_OBU_SEQUENCE_HEADER) {
                s->seq_reduced_still = o.payload_size && (o.payload[0] & 0x08);
                for (VideoSink *c : s->children) {
                    std::unique_lock<std::mutex> lk(c->mu);
                    c->seq_obu.assign(unit + pos, unit + pos + o.total_size);
                    c->seq_reduced_still = s->seq_reduced_still;
                }
            }
            pos += o.total_size;
        }
        VideoSink *c = deal(s, starts_gop(unit, size, s->seq_reduced_still));
        uint64_t local = 0;
        const int rc = video_sink_submit_av1(c, unit, size, timestamp, keyframe, discard, &local);
        if (rc == EA1_OK) dealt_ticket(s, local, ticket);
        return rc;
    }
    if (s && !s->av1) return fail(EA1_ESTATE, "video sink: plain AV1 needs the AV1 table blob (data/av1_spec_tables.bin beside the library, or EASYAV1_AMD_AV1_TABLES)");
    if (!s || !unit) return fail(EA1_EINVAL, "video sink: NULL argument");
    std::vector<size_t> ends;
    split_unit(unit, size, &ends);
    size_t start = 0;
    int rc = EA1_OK;
    for (size_t k = 0; k < ends.size() && rc == EA1_OK; k++) {
        // the sequence header in force BEFORE this frame (a chain's first packet is primed with it when it carries none of its own),
        // then the ones this frame brings along
        std::vector<uint8_t> seq_before;
        bool reduced_still;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            seq_before = s->seq_obu;
        }
        for (size_t pos = start; pos < ends[k];) {
            ea1_obu o;
            if (ea1_av1_next_obu(unit + pos, ends[k] - pos, &o) != EA1_OK) break;
            if (o.type == EA1_OBU_SEQUENCE_HEADER) {
                std::unique_lock<std::mutex> lk(s->mu);
                s->seq_obu.assign(unit + pos, unit + pos + o.total_size);
                s->seq_reduced_still = o.payload_size && (o.payload[0] & 0x08);   // seq_profile f(3), still_picture f(1), reduced_still_picture_header f(1)
            }
            pos += o.total_size;
        }
        {
            std::unique_lock<std::mutex> lk(s->mu);
            reduced_still = s->seq_reduced_still;
        }
        // (which of them is shown the parser finds out: a frame with show_frame = 0 turns its item into a discarded one)
        rc = submit_item(s, unit + start, ends[k] - start, timestamp, starts_gop(unit + start, ends[k] - start, reduced_still), discard, true, ticket, &seq_before);
        start = ends[k];
    }
    return rc;
}
static int submit_item(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool keyframe, bool discard, bool av1, uint64_t *ticket,
                       const std::vector<uint8_t> *seq_before)
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
        // (a picture whose turn has come was waited for, so it is in its host buffer or on the way: the deliverer is not inside a
        // copy of it, which pop_through below must not pull the source from under)
        s->cv_done.wait(lk, [&] { return old->state == St::Dead || (old->state == St::Issued && old->buffer >= 0); });
        pop_through(s, old, false);
    }
    auto it = std::unique_ptr<Item>(new (std::nothrow) Item());
    if (!it) return fail(EA1_ENOMEM, "out of memory");
    it->ticket = s->next_ticket++;
    it->tag = s->tag;
    it->payload = payload;
    it->size = size;
    it->timestamp = timestamp;
    it->discard = discard;
    it->av1 = av1;
    if (keyframe && s->any_submitted) s->gop_count++;   // a key frame starts a GOP: the next chain in turn takes it
    s->any_submitted = true;
    it->chain = int(s->gop_count % uint64_t(s->chains));
    if (seq_before) it->seq_before = *seq_before;   // (a few bytes: the header in force at THIS position of the stream, not whatever is newest at parse time)
    if (ticket) *ticket = it->ticket;
    s->items.push_back(std::move(it));
    s->cv_work.notify_one();
    return EA1_OK;
}

bool video_sink_ready(VideoSink *s, uint64_t ticket)
{
    if (!s) return true;
    if (!s->children.empty()) {
        if (ticket < s->dealt_base || ticket - s->dealt_base >= s->dealt.size()) return true;
        const VideoSink::Dealt d = s->dealt[size_t(ticket - s->dealt_base)];
        return video_sink_ready(s->children[size_t(d.child)], d.local);
    }
    DeviceGuard g(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    Item *it = find(s, ticket);
    if (!it) return true;
    if (it->state == St::Failed || it->state == St::Dead || s->error_rc != EA1_OK) return true;   // video_sink_wait has an error to report
    if (it->state != St::Issued) return false;
    if (it->discard) return batch_landed(*it);
    return it->buffer >= 0 && batch_landed(*it);
}

int video_sink_wait(VideoSink *s, uint64_t ticket)
{
    if (!s) return fail(EA1_EINVAL, "video sink: NULL argument");
    if (!s->children.empty()) {
        if (ticket < s->dealt_base || ticket - s->dealt_base >= s->dealt.size()) return EA1_OK;   // flushed / long gone
        const VideoSink::Dealt d = s->dealt[size_t(ticket - s->dealt_base)];
        return video_sink_wait(s->children[size_t(d.child)], d.local);
    }
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    s->waited_through = std::max(s->waited_through, ticket + 1);
    Item *it = find(s, ticket);
    if (!it) {   // retired already (dropped from the ring, or a discarded frame that has finished)
        if (s->error_rc != EA1_OK && ticket >= s->error_ticket) return fail(s->error_rc, "%s", s->error_msg.c_str());
        return EA1_OK;
    }
    for (;;) {
        s->cv_done.wait(lk, [&] { return settled(*it) || it->state == St::Failed; });
        if (it->state != St::Issued) break;
        if (it->discard || it->buffer >= 0) break;   // reconstructed / on its way to the host
        // its picture waits in HBM for a host buffer.  When the ring is full of pictures nobody took, the oldest goes (the
        // reference's decoded-frame ring drops its oldest entry, src/easyav1.c:2141-2143): this is the thread that would take them
        if (s->free_buffers.empty() && int(s->buffers.size()) + s->allocating >= kHostRing) {
            Item *old = nullptr;
            for (auto &p : s->items)
                if (!p->discard && p->state == St::Issued && p->buffer >= 0) {
                    old = p.get();
                    break;
                }
            if (old && old != it) {
                pop_through(s, old, false);
                s->cv_issue.notify_all();
                continue;
            }
        }
        s->cv_issue.notify_all();
        s->cv_done.wait(lk);
    }
    if (it->state != St::Issued) {
        const int rc = it->rc != EA1_OK ? it->rc : (s->error_rc != EA1_OK ? s->error_rc : EA1_ESTATE);
        return fail(rc, "%s", it->msg.empty() ? s->error_msg.c_str() : it->msg.c_str());
    }
    hipEvent_t ev = it->discard ? it->batch->done : it->landed;
    lk.unlock();
    if (hipEventSynchronize(ev) != hipSuccess) return fail(EA1_EHIP, "video sink: waiting for a picture failed: %s", hipGetErrorString(hipGetLastError()));
    if (ticket < 8 || (ticket & 63) == 0) EA1_TRACE("ticket %llu waited for", (unsigned long long)ticket);
    return EA1_OK;
}

// the dealer: the child whose oldest due picture comes first in the stream (timestamps grow with the stream; the pictures of one packet
// sit in one child), -1 = none is due
static int next_child(VideoSink *s, uint64_t position)
{
    int best = -1;
    uint64_t best_ts = 0;
    for (size_t k = 0; k < s->children.size(); k++) {
        VideoSink *c = s->children[k];
        std::lock_guard<std::mutex> lk(c->mu);
        Item *it = oldest_picture(c);
        if (!it || it->timestamp > position || it->state != St::Issued || it->buffer < 0) continue;
        if (best < 0 || it->timestamp < best_ts) best = int(k), best_ts = it->timestamp;
    }
    return best;
}

bool video_sink_has_frame(VideoSink *s, uint64_t position)
{
    if (!s) return false;
    if (!s->children.empty()) {
        const int k = next_child(s, position);
        return k >= 0 && video_sink_has_frame(s->children[size_t(k)], position);
    }
    std::lock_guard<std::mutex> lk(s->mu);
    Item *it = oldest_picture(s);
    return it && it->timestamp <= position && it->state == St::Issued && it->buffer >= 0 && batch_landed(*it);
}

bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out)
{
    if (!s) return false;
    if (!s->children.empty()) {
        const int k = next_child(s, position);
        if (k < 0) return false;
        s->taken_from = k;
        return video_sink_take(s->children[size_t(k)], position, out);
    }
    DeviceGuard g(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    sweep(s);
    Item *it = oldest_picture(s);
    if (!it || it->timestamp > position || it->state != St::Issued || it->buffer < 0) return false;
    hipEvent_t ev = it->landed;
    lk.unlock();
    (void)hipEventSynchronize(ev);
    lk.lock();
    if (s->shown >= 0) s->free_buffers.push_back(s->shown);   // (pop_through below tells the issuer)
    s->shown = it->buffer;
    const uint8_t *base = s->buffers[size_t(it->buffer)];
    for (int pl = 0; pl < 3; pl++) {
        out->data[pl] = base + s->plane_off[pl];
        out->stride[pl] = s->plane_stride[pl];
    }
    out->width = it->vis_w;
    out->height = it->vis_h;
    out->timestamp = it->timestamp;
    out->tag = it->tag;
    pop_through(s, it, true);
    return true;
}

void video_sink_set_tag(VideoSink *s, uint32_t tag)
{
    if (!s) return;
    for (VideoSink *c : s->children) video_sink_set_tag(c, tag);   // (a child takes whole GOPs: the tag in force when it is dealt one)
    std::unique_lock<std::mutex> lk(s->mu);
    s->tag = tag;
}

}  // namespace ea1
