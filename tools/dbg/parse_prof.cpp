// Where the block-level AV1 parser spends its time on a real stream: a SIGPROF sampler (1 kHz, instruction pointers) around
// ea1_av1_frame_parser_parse, attributed with addr2line afterwards (tools/dbg/parse_prof.sh).  Build-container diagnostics only.
//   usage: parse_prof <stream file of tools/fuzz/make_av1_corpus.py or tools/dbg/make_parse_stream.py> [repeats]
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <ctime>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "easyav1_amd_obu.h"

static uint64_t g_pc[1 << 20];
static volatile int g_n = 0;
static void on_prof(int, siginfo_t *, void *uc)
{
    if (g_n < (1 << 20)) g_pc[g_n++] = uint64_t(static_cast<ucontext_t *>(uc)->uc_mcontext.gregs[REG_RIP]);
}
static double cpu_s()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);   // (the build container is shared: wall time of a pass is noise)
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}
static std::vector<uint8_t> read_file(const char *p)
{
    std::vector<uint8_t> v;
    if (FILE *f = fopen(p, "rb")) {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        fclose(f);
    }
    return v;
}
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int repeats = argc > 2 ? atoi(argv[2]) : 3;
    const std::vector<uint8_t> d = read_file(argv[1]);
    size_t pos = 0;
    auto take = [&](std::vector<uint8_t> *out) {
        if (pos + 4 > d.size()) return false;
        uint32_t n;
        memcpy(&n, &d[pos], 4);
        pos += 4;
        if (pos + n > d.size()) return false;
        out->assign(d.begin() + long(pos), d.begin() + long(pos + n));
        pos += n;
        return true;
    };
    std::vector<uint8_t> blob, u;
    std::vector<std::vector<uint8_t>> units;
    if (!take(&blob)) return 1;
    while (take(&u)) units.push_back(u);
    ea1_av1_tables *t = nullptr;
    if (ea1_av1_tables_load(blob.data(), blob.size(), &t) != EA1_OK) return 1;
    struct sigaction sa {};
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    itimerval tv{{0, 1000}, {0, 1000}};
    int frames = 0;
    uint64_t blocks = 0, txs = 0, coefs = 0, bytes = 0;
    const auto t0 = std::chrono::steady_clock::now();
    setitimer(ITIMER_PROF, &tv, nullptr);
    double best = 1e30, best_inter = 1e30, t_inter0 = 0;
    const bool inter_only = getenv("EA1_PROF_INTER_ONLY") != nullptr;
    // ONE parser for all passes, as a GOP chain of the sink keeps its parser: its arrays have their capacity after the first pass (every pass
    // starts with a key frame, which resets the parser's state)
    ea1_av1_frame_parser *p = nullptr;
    if (ea1_av1_frame_parser_create(t, &p) != EA1_OK) return 1;
    for (int r = 0; r < repeats; r++) {
        const double tr = cpu_s();
        for (size_t ui = 0; ui < units.size(); ui++) {
            const auto &un = units[ui];
            if (inter_only) {   // EA1_PROF_INTER_ONLY: sample (and time) everything but the first unit of a pass, the key frame
                itimerval tv2 = ui ? tv : itimerval{};
                setitimer(ITIMER_PROF, &tv2, nullptr);
                if (ui == 1) t_inter0 = cpu_s();
            }
            size_t at = 0;
            bytes += un.size();
            for (int guard = 0; at < un.size() && guard < 64; guard++) {
                ea1_av1_parsed out;
                if (ea1_av1_frame_parser_parse(p, un.data() + at, un.size() - at, &out) != EA1_OK) {
                    fprintf(stderr, "parse failed: %s\n", ea1_last_error());
                    return 1;
                }
                frames++;
                blocks += out.frame.desc.num_blocks;
                txs += out.frame.desc.num_tx;
                coefs += out.frame.desc.num_coefs;
                if (!out.consumed) break;
                at += out.consumed;
            }
        }
        best = std::min(best, cpu_s() - tr);
        if (inter_only && units.size() > 1) best_inter = std::min(best_inter, (cpu_s() - t_inter0) / double(units.size() - 1));
    }
    ea1_av1_frame_parser_destroy(p);
    itimerval off{};
    setitimer(ITIMER_PROF, &off, nullptr);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (inter_only) fprintf(stderr, "frames behind the first of a pass: best %.3f ms of CPU per frame\n", 1e3 * best_inter);
    fprintf(stderr, "best pass %.2f ms of CPU per frame | %d frames in %.3f s: %.2f ms per frame, %.1f frames/s; per frame %.0f bytes, %.0f blocks, %.0f transform blocks, %.0f coefficient slots; %d samples\n", 1e3 * best * repeats / frames, frames, s,
            1e3 * s / frames, frames / s, double(bytes) / frames, double(blocks) / frames, double(txs) / frames, double(coefs) / frames, g_n);
    std::map<uint64_t, int> h;
    for (int i = 0; i < g_n; i++) h[g_pc[i]]++;
    for (auto &kv : h) printf("%llx %d\n", (unsigned long long)kv.first, kv.second);
    return 0;
}
