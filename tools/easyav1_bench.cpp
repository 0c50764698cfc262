// easyav1_bench -- this repository's own benchmark tool for the public interface (include/easyav1.h): what the
// reference's tools/easyav1_benchmark.c measures (/root/reference/tools/easyav1_benchmark.c:72-187: audio off, decode
// every packet as fast as possible, time from one displayed frame to the next, report frames per second and the
// slowest frame) with the clock the reference lacks: it adds up whole MILLISECONDS per frame (:54-63, :124-125), so a
// frame that takes less than a millisecond adds nothing and above 1 000 frames/s its figure is meaningless.  Here the
// per-frame times are nanoseconds, and the wall time of the whole decode loop is printed beside their sum.
//
//   easyav1_bench <file.webm> [--json] [--touch] [--repeat N] [--skip K]
//     --json    one machine-readable line at the end (bench.py's api_single_stream leg reads it)
//     --touch   read one byte per 4 KiB page of every picture (a consumer that looks at its pictures)
//     --repeat  decode the file N times in this process (the first pass pays start-up: contexts, pinned buffers)
//     --skip    the first K displayed frames are decoded but not counted (steady state: the context, its pinned
//               buffers and the code objects exist by then); the JSON line says so
//
// Output lines as the reference prints them ("Video duration", "Video size", "Decoded N frames in T milliseconds
// (F fps average, Sx).", "Slowest frame: ...") with T in fractional milliseconds.
#include <dirent.h>
#include <sys/resource.h>
#include <unistd.h>

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "easyav1.h"

namespace {
using Clock = std::chrono::steady_clock;
int64_t ns_since(Clock::time_point t0) { return std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count(); }

struct Pass {
    uint64_t frames = 0;
    int64_t sum_ns = 0, wall_ns = 0, slowest_ns = 0, open_ns = 0, wall_all_ns = 0;   // wall_all: the whole decode loop, skipped frames included
    double cpu_s = 0;   // user + system time of the whole process (every thread) over the decode loop: how many host CPUs the pipeline kept busy
    uint64_t frames_all = 0;
    uint64_t slowest_frame = 0, slowest_ts = 0, duration_ms = 0;
    unsigned width = 0, height = 0, fps = 0;
    uint64_t checksum = 0;
    int rc = 0;
};

// CPU seconds by thread name (/proc/self/task/*/stat: utime + stime), for --threads: the library names its threads (ea1-helper, ea1-stage, ea1-tile,
// ea1-prepare); the first thread is the caller, unnamed ones belong to the HIP / HSA runtime
bool g_threads = false;
void print_thread_cpu()
{
    struct Row { std::string name; double s = 0; int n = 0; };
    std::vector<Row> rows;
    const double tick = 1.0 / double(sysconf(_SC_CLK_TCK));
    const long self = long(getpid());
    if (DIR *d = opendir("/proc/self/task")) {
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            char path[128], line[1024];
            snprintf(path, sizeof path, "/proc/self/task/%s/stat", e->d_name);
            FILE *f = fopen(path, "r");
            if (!f) continue;
            const bool ok = fgets(line, sizeof line, f) != nullptr;
            fclose(f);
            if (!ok) continue;
            const char *l = strchr(line, '('), *r = strrchr(line, ')');
            if (!l || !r) continue;
            std::string name(l + 1, r);
            if (atol(e->d_name) == self) name = "caller (main thread)";
            unsigned long ut = 0, st = 0;
            // after ") ": state ppid pgrp session tty tpgid flags minflt cminflt majflt cmajflt utime stime
            if (sscanf(r + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st) != 2) continue;
            size_t k = 0;
            while (k < rows.size() && rows[k].name != name) k++;
            if (k == rows.size()) rows.push_back(Row{name, 0, 0});
            rows[k].s += double(ut + st) * tick;
            rows[k].n++;
        }
        closedir(d);
    }
    double all = 0;
    for (const Row &r : rows) all += r.s;
    fprintf(stderr, "[threads] CPU seconds of the threads alive at the end of the pass (total %.2f):\n", all);
    for (const Row &r : rows) fprintf(stderr, "[threads]   %-24s x %-3d %8.2f s  %5.1f %%\n", r.name.c_str(), r.n, r.s, all > 0 ? 100.0 * r.s / all : 0.0);
}

double process_cpu_s()
{
    rusage u;
    getrusage(RUSAGE_SELF, &u);
    return double(u.ru_utime.tv_sec + u.ru_stime.tv_sec) + 1e-6 * double(u.ru_utime.tv_usec + u.ru_stime.tv_usec);
}

Pass run(const char *path, bool touch, bool verbose, uint64_t skip)
{
    Pass r;
    easyav1_settings settings = easyav1_default_settings();
    settings.enable_audio = EASYAV1_FALSE;
    settings.skip_unprocessed_frames = EASYAV1_FALSE;
    settings.log_level = EASYAV1_LOG_LEVEL_ERROR;
    const Clock::time_point t_open = Clock::now();
    easyav1_t *h = easyav1_init_from_filename(path, &settings);
    r.open_ns = ns_since(t_open);
    if (!h) {
        printf("Failed to initialize easyav1.\n");
        r.rc = 2;
        return r;
    }
    if (!easyav1_has_video_track(h)) {
        printf("The video does not contain a video track.\n");
        easyav1_destroy(&h);
        r.rc = 3;
        return r;
    }
    r.duration_ms = easyav1_get_duration(h);
    r.width = easyav1_get_video_width(h);
    r.height = easyav1_get_video_height(h);
    r.fps = easyav1_get_video_fps(h);
    if (verbose) {
        printf("Video duration: %" PRIu64 ":%02" PRIu64 " (%" PRIu64 " ms).\n", r.duration_ms / 60000, (r.duration_ms / 1000) % 60, r.duration_ms);
        printf("Video size: %ux%u, %u FPS.\n", r.width, r.height, r.fps);
        fflush(stdout);
    }
    static const bool timeline = getenv("EASYAV1_BENCH_TIMELINE") != nullptr;
    Clock::time_point t_loop = Clock::now();
    const Clock::time_point t_all = t_loop;
    const double cpu0 = process_cpu_s();
    Clock::time_point t_frame = t_loop, t_print = t_loop;
    uint64_t shown = 0;
    while (easyav1_decode_next(h) == EASYAV1_STATUS_OK) {
        if (easyav1_has_video_frame(h) == EASYAV1_FALSE) continue;
        // the frame counts from the previous displayed frame to the moment its picture is in the caller's hands
        const easyav1_video_frame *f = easyav1_get_video_frame(h);
        if (f && touch)
            for (int pl = 0; pl < 3; pl++) {
                const bool half_rows = f->properties.pixel_layout == EASYAV1_PIXEL_LAYOUT_YUV420;   // 4:2:2 / 4:4:4: chroma planes of full height
                const size_t rows = pl && half_rows ? (f->properties.height + 1) / 2 : f->properties.height;
                const uint8_t *p = static_cast<const uint8_t *>(f->data[pl]);
                for (size_t o = 0; o < rows * f->stride[pl]; o += 4096) r.checksum += p[o];
            }
        r.frames_all++;
        if (timeline && (r.frames_all & (r.frames_all - 1)) == 0)   // pictures 1, 2, 4, 8, ...: when they reached the caller, from before the file was opened
            fprintf(stderr, "[timeline] picture %" PRIu64 " at %.1f ms\n", r.frames_all, ns_since(t_open) / 1e6);
        if (shown++ < skip) {   // warm-up frames: the clocks start behind the last of them
            t_frame = Clock::now();
            if (shown == skip) t_loop = t_frame;
            continue;
        }
        const int64_t dt = ns_since(t_frame);
        r.sum_ns += dt;
        r.frames++;
        if (dt > r.slowest_ns) {
            r.slowest_ns = dt;
            r.slowest_frame = easyav1_get_total_video_frames_processed(h);
            r.slowest_ts = easyav1_get_current_timestamp(h);
        }
        if (verbose && ns_since(t_print) > 1000000000) {
            const uint64_t ts = easyav1_get_current_timestamp(h);
            printf("\rDecoding (%" PRIu64 ":%02" PRIu64 "): Decoded %" PRIu64 " frames in %.3f ms (%f fps average, %.3fx).   ", ts / 60000, (ts / 1000) % 60,
                   r.frames, r.sum_ns / 1e6, r.frames / (r.sum_ns / 1e9), ts / (r.sum_ns / 1e6));
            fflush(stdout);
            t_print = Clock::now();
        }
        t_frame = Clock::now();
    }
    r.wall_ns = ns_since(t_loop);
    r.wall_all_ns = ns_since(t_all);
    r.cpu_s = process_cpu_s() - cpu0;
    if (!easyav1_is_finished(h)) {
        printf("\nFailed to decode the video.\n");
        r.rc = 4;
    }
    if (g_threads) print_thread_cpu();
    easyav1_destroy(&h);
    return r;
}
}  // namespace

int main(int argc, const char **argv)
{
    const char *path = nullptr;
    bool json = false, touch = false;
    int repeat = 1;
    uint64_t skip = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--json")) json = true;
        else if (!strcmp(argv[i], "--touch")) touch = true;
        else if (!strcmp(argv[i], "--threads")) g_threads = true;   // CPU seconds per thread name at the end of a pass (stderr)
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--skip") && i + 1 < argc) skip = strtoull(argv[++i], nullptr, 10);
        else path = argv[i];
    }
    if (!path || repeat < 1) {
        fprintf(stderr, "Usage: %s <filename> [--json] [--touch] [--repeat N] [--skip K]\n", argv[0]);
        return 1;
    }
    Pass last, best;
    for (int k = 0; k < repeat; k++) {
        last = run(path, touch, k == 0, skip);
        if (last.rc) return last.rc;
        if (k == 0 || last.wall_ns < best.wall_ns) best = last;
        const double fps = last.frames / (last.sum_ns / 1e9);
        printf("\rDecoded %" PRIu64 " frames in %.3f milliseconds (%f fps average, %.3fx).                    \n", last.frames, last.sum_ns / 1e6, fps,
               last.duration_ms / (last.sum_ns / 1e6));
        printf("Slowest frame: #%" PRIu64 " (at %" PRIu64 ":%02" PRIu64 ") - %.3f milliseconds (%f fps).\n", last.slowest_frame, last.slowest_ts / 60000,
               (last.slowest_ts / 1000) % 60, last.slowest_ns / 1e6, 1e9 / double(last.slowest_ns ? last.slowest_ns : 1));
        printf("Wall time of the decode loop: %.3f ms (%f frames/s); opening the file took %.3f ms.\n", last.wall_ns / 1e6, last.frames / (last.wall_ns / 1e9),
               last.open_ns / 1e6);
        if (skip) printf("With the %" PRIu64 " skipped frames: %" PRIu64 " frames in %.3f ms (%f frames/s).\n", skip, last.frames_all, last.wall_all_ns / 1e6, last.frames_all / (last.wall_all_ns / 1e9));
    }
    if (json)
        printf("{\"frames\": %" PRIu64 ", \"sum_frame_ms\": %.6f, \"wall_ms\": %.6f, \"fps_frames\": %.3f, \"fps_wall\": %.3f, \"slowest_ms\": %.6f, \"open_ms\": %.3f, "
               "\"width\": %u, \"height\": %u, \"passes\": %d, \"skipped_frames\": %" PRIu64 ", \"frames_all\": %" PRIu64 ", \"fps_wall_all\": %.3f, \"cpu_seconds\": %.3f, \"cpus_busy\": %.2f, \"checksum\": %" PRIu64 "}\n",
               best.frames, best.sum_ns / 1e6, best.wall_ns / 1e6, best.frames / (best.sum_ns / 1e9), best.frames / (best.wall_ns / 1e9), best.slowest_ns / 1e6,
               best.open_ns / 1e6, best.width, best.height, repeat, skip, best.frames_all, best.frames_all / (best.wall_all_ns / 1e9), best.cpu_s, best.cpu_s / (best.wall_all_ns / 1e9), best.checksum);
    return 0;
}
