// Host cost of ea1_job_create on a REAL 4K key frame (the block parser's output for the first frame of a stream file of tools/dbg/parse_prof.cpp),
// without a device (EA1_FUZZ_HOOK build of av1_recon.cpp), SIGPROF-sampled: build-container diagnostics.  usage: job_prof_real <stream file> [iterations] [frame]
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "easyav1_amd_obu.h"
#include "easyav1_amd_video.h"

extern "C" int ea1_fuzz_job_create(uint32_t width, uint32_t height, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                   const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3]);
static uint64_t g_pc[1 << 20];
static volatile int g_n = 0;
static void on_prof(int, siginfo_t *, void *uc)
{
    if (g_n < (1 << 20)) g_pc[g_n++] = uint64_t(static_cast<ucontext_t *>(uc)->uc_mcontext.gregs[REG_RIP]);
}
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 20, want = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<uint8_t> d;
    if (FILE *f = fopen(argv[1], "rb")) {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
    }
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
    if (!take(&blob)) return 1;
    ea1_av1_tables *t = nullptr;
    if (ea1_av1_tables_load(blob.data(), blob.size(), &t) != EA1_OK) return 1;
    ea1_av1_frame_parser *p = nullptr;
    if (ea1_av1_frame_parser_create(t, &p) != EA1_OK) return 1;
    ea1_av1_parsed out;
    int fi = 0;
    bool have = false;
    while (!have && take(&u)) {
        size_t at = 0;
        while (at < u.size()) {
            if (ea1_av1_frame_parser_parse(p, u.data() + at, u.size() - at, &out) != EA1_OK) return 1;
            if (fi++ == want) { have = true; break; }
            if (!out.consumed) break;
            at += out.consumed;
        }
    }
    if (!have) return 1;
    const ea1_ir_frame &f = out.frame;
    fprintf(stderr, "frame %d: %ux%u, %u blocks, %u transform blocks, %u coefficient slots\n", want, f.desc.width, f.desc.height, f.desc.num_blocks, f.desc.num_tx, f.desc.num_coefs);
    struct sigaction sa {};
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    itimerval tv{{0, 1000}, {0, 1000}};
    const ea1_lr_unit *lr[3] = {f.lr[0], f.lr[1], f.lr[2]};
    const auto t0 = std::chrono::steady_clock::now();
    setitimer(ITIMER_PROF, &tv, nullptr);
    int rc = 0;
    for (int i = 0; i < iters; i++) rc = ea1_fuzz_job_create(f.desc.width, f.desc.height, &f.desc, f.blocks, f.txs, f.coefs, f.aux, f.cdef_idx, lr);
    itimerval off{};
    setitimer(ITIMER_PROF, &off, nullptr);
    fprintf(stderr, "%d calls, %.3f ms each (rc %d: the device allocation at the end fails without a GPU)\n", iters,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters, rc);
    std::map<uint64_t, int> h;
    for (int i = 0; i < g_n; i++) h[g_pc[i]]++;
    for (auto &kv : h) printf("%llx %d\n", (unsigned long long)kv.first, kv.second);
    return 0;
}
