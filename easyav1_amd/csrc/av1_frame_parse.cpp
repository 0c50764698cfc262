// ea1_av1_frame_parser: the block-level AV1 parse behind the C ABI (include/easyav1_amd_obu.h).  One temporal unit in, one
// ea1_ir_frame out.  See av1_tile_syntax.h for what is parsed and av1_frame_driver.h for the frame-level bookkeeping.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "av1_frame_driver.h"

using namespace ea1::av1;

// helper threads of ONE parser for the tiles of a frame (ea1_av1_frame_parser_set_threads): the caller's thread takes tiles too
struct TilePool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> job;
    int next = 0, count = 0, running = 0;
    int slots = 0;               // workers that may still join the current run (ea1_av1_frame_parser_set_thread_limit)
    uint64_t generation = 0;
    bool stop = false;

    void start(int nworkers)
    {
        shutdown();
        stop = false;
        for (int i = 0; i < nworkers; i++) workers.emplace_back([this] { loop(); });
    }
    void shutdown()
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (std::thread &t : workers) t.join();
        workers.clear();
    }
    ~TilePool() { shutdown(); }
    bool take(std::unique_lock<std::mutex> &lk, int *k)
    {
        (void)lk;
        if (next >= count) return false;
        *k = next++;
        running++;
        return true;
    }
    void loop()
    {
        (void)pthread_setname_np(pthread_self(), "ea1-tile");
        std::unique_lock<std::mutex> lk(mu);
        uint64_t seen = 0;
        for (;;) {
            cv_work.wait(lk, [&] { return stop || (generation != seen && next < count); });
            if (stop) return;
            if (slots <= 0) {      // this run has all the threads it may use
                seen = generation;
                continue;
            }
            slots--;
            int k;
            while (take(lk, &k)) {
                lk.unlock();
                job(k);
                lk.lock();
                if (--running == 0 && next >= count) cv_done.notify_all();
            }
            seen = generation;
        }
    }
    // job(0) .. job(n - 1), each exactly once, on the workers and on this thread; returns when all have finished
    void run(int n, std::function<void(int)> f, int max_threads)
    {
        std::unique_lock<std::mutex> lk(mu);
        job = std::move(f);
        next = 0, count = n, running = 0;
        slots = max_threads - 1;
        generation++;
        cv_work.notify_all();
        int k;
        while (take(lk, &k)) {
            lk.unlock();
            job(k);
            lk.lock();
            --running;
        }
        cv_done.wait(lk, [&] { return running == 0 && next >= count; });
        count = 0;
    }
};

struct ea1_av1_frame_parser {
    FrameDriver d;
    std::vector<ea1_av1_tile> tiles;
    TilePool pool;
    int threads = 1;
    std::atomic<int> limit{64};   // of `threads`, how many the next frames may use
};

extern "C" int ea1_av1_frame_parser_create(const ea1_av1_tables *tables, ea1_av1_frame_parser **out)
{
    if (!tables || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_create: NULL argument");
    ea1_av1_frame_parser *p = new (std::nothrow) ea1_av1_frame_parser;
    if (!p) return ea1::fail(EA1_ENOMEM, "ea1_av1_frame_parser_create: out of memory");
    p->d.tab = tables_of(tables);
    const int rc = ea1_av1_parser_create(&p->d.hp);
    if (rc != EA1_OK) {
        delete p;
        return rc;
    }
    p->tiles.resize(size_t(EA1_MAX_TILE_COLS) * EA1_MAX_TILE_ROWS);
    if (const char *tt = getenv("EASYAV1_AMD_TILE_THREADS")) {   // default for parsers the caller does not configure
        const int n = atoi(tt);
        if (n > 1 && n <= 64) {
            p->pool.start(n - 1);
            p->threads = n;
        }
    }
    *out = p;
    return EA1_OK;
}

extern "C" void ea1_av1_frame_parser_destroy(ea1_av1_frame_parser *p)
{
    if (!p) return;
    ea1_av1_parser_destroy(p->d.hp);
    delete p;
}

extern "C" int ea1_av1_frame_parser_set_threads(ea1_av1_frame_parser *p, int threads)
{
    if (!p || threads < 1 || threads > 64) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_set_threads: %d threads (1..64)", threads);
    if (threads == p->threads) return EA1_OK;
    p->pool.start(threads - 1);
    p->threads = threads;
    return EA1_OK;
}

extern "C" int ea1_av1_frame_parser_allow_chroma_formats(ea1_av1_frame_parser *p, int on)
{
    if (!p) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_allow_chroma_formats: NULL handle");
    p->d.allow_chroma_formats = on != 0;
    return EA1_OK;
}

extern "C" int ea1_av1_frame_parser_set_thread_limit(ea1_av1_frame_parser *p, int threads)
{
    if (!p || threads < 1) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_set_thread_limit: %d", threads);
    p->limit.store(threads, std::memory_order_relaxed);
    return EA1_OK;
}

extern "C" void ea1_av1_frame_parser_reset(ea1_av1_frame_parser *p)
{
    if (!p) return;
    // the header parser keeps its sequence header; its slots are overwritten by the key frame that must follow
    p->d.reset_slots();
}

static int parse_frame(ea1_av1_frame_parser *p, const uint8_t *data, size_t size, ea1_av1_parsed *out);

extern "C" int ea1_av1_frame_parser_parse(ea1_av1_frame_parser *p, const uint8_t *data, size_t size, ea1_av1_parsed *out)
{
    if (!p || !data || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_parse: NULL argument");
    // (the record arrays grow as the stream demands: an allocation failure must come back as a code, not unwind through the C caller)
    try {
        return parse_frame(p, data, size, out);
    } catch (const std::bad_alloc &) {
        return ea1::fail(EA1_ENOMEM, "ea1_av1_frame_parser_parse: out of memory");
    }
}

static int parse_frame(ea1_av1_frame_parser *p, const uint8_t *data, size_t size, ea1_av1_parsed *out)
{
    const auto t0 = std::chrono::steady_clock::now();
    memset(out, 0, sizeof *out);
    size_t consumed = 0;
    int count = 0;
    int rc = ea1_av1_parser_next_frame(p->d.hp, data, size, &consumed, &out->header, p->tiles.data(), int(p->tiles.size()), &count);
    if (rc != EA1_OK) return rc;
    const ea1_av1_frame_header &hdr = out->header;
    out->consumed = consumed;
    if (hdr.show_existing_frame) {
        rc = p->d.show_existing(hdr, &out->frame);
        if (rc != EA1_OK) return rc;
        out->show_existing = 1;
        out->show_frame = 1;
        out->has_film_grain = ea1_av1_frame_header_film_grain(&hdr, &p->d.seq, &out->film_grain) == EA1_OK;
        out->parse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return EA1_OK;
    }
    static const bool phases = getenv("EA1_AV1_PHASES") != nullptr;   // diagnostics: where a frame's parse time goes (stderr)
    const auto t1 = std::chrono::steady_clock::now();
    rc = p->d.begin_frame(hdr);
    if (rc != EA1_OK) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    static const bool slow = getenv("EA1_AV1_SLOW_MSAC") != nullptr;
    const int use_threads = std::min(p->threads, p->limit.load(std::memory_order_relaxed));
    if (use_threads > 1 && count > 1 && !slow) {
        // the tiles side by side (FrameDriver::run_tile_into / merge_tiles): the same records as the loop below, byte for byte
        p->d.begin_tiles(count);
        const int disable = hdr.disable_cdf_update;
        const unsigned ctx_tile = hdr.context_update_tile_id;
        p->pool.run(count, [&](int k) {
            const ea1_av1_tile &t = p->tiles[size_t(k)];
            MsacReader io;
            if (io.init(data + t.offset, t.size, disable) != EA1_OK) {
                p->d.tile_rc[size_t(k)] = EA1_EINVAL;
                p->d.tile_msg[size_t(k)] = "AV1 block parser: empty tile";
                return;
            }
            const auto a = std::chrono::steady_clock::now();
            p->d.run_tile_into(io, t, t.tile_num == ctx_tile, k);
            if (phases) fprintf(stderr, "[ea1 av1 parse]   tile %d: %.3f ms on thread %lx (%zu bytes)\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(),
                                (unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xFFFF, size_t(t.size));
        }, use_threads);
        const auto tm = std::chrono::steady_clock::now();
        rc = p->d.merge_tiles(count);
        if (rc != EA1_OK) return rc;
        if (phases) fprintf(stderr, "[ea1 av1 parse]   merge %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm).count());
        count = 0;   // (nothing left for the loop)
    }
    for (int k = 0; k < count; k++) {
        const ea1_av1_tile &t = p->tiles[size_t(k)];
        if (slow) {
            SlowMsacReader io;
            rc = io.init(data + t.offset, t.size, hdr.disable_cdf_update);
            if (rc == EA1_OK) rc = p->d.run_tile(io, t, t.tile_num == hdr.context_update_tile_id);
        } else {
            MsacReader io;
            rc = io.init(data + t.offset, t.size, hdr.disable_cdf_update);
            if (rc == EA1_OK) rc = p->d.run_tile(io, t, t.tile_num == hdr.context_update_tile_id);
        }
        if (rc != EA1_OK) return rc;
    }
    const auto t3 = std::chrono::steady_clock::now();
    rc = p->d.end_frame(hdr, &out->frame);
    if (rc != EA1_OK) return rc;
    if (phases) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[ea1 av1 parse] headers %.3f ms, begin_frame %.3f, tiles %.3f (%d tiles, %d threads), end_frame %.3f; %u blocks\n", ms(t0, t1), ms(t1, t2), ms(t2, t3),
                int(hdr.tile_cols) * int(hdr.tile_rows), p->threads, ms(t3, t4), out->frame.desc.num_blocks);
    }
    out->show_frame = hdr.show_frame;
    out->has_film_grain = ea1_av1_frame_header_film_grain(&hdr, &p->d.seq, &out->film_grain) == EA1_OK;
    out->parse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return EA1_OK;
}
