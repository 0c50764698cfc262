// Where the block-level AV1 parser spends its time, by source line: SIGPROF sampler over a stream file
//   bash tools/host_profile_av1.sh <table blob> <units file: (u32 size + temporal unit)*> [repeats]
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "easyav1_amd_obu.h"

static uint64_t pcs[1 << 20];
static volatile size_t npcs = 0;
static void on_prof(int, siginfo_t *, void *uc)
{
    if (npcs < (sizeof pcs / sizeof pcs[0])) pcs[npcs++] = uint64_t(static_cast<ucontext_t *>(uc)->uc_mcontext.gregs[REG_RIP]);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int repeats = argc > 3 ? atoi(argv[3]) : 10;
    std::vector<uint8_t> blob(1 << 21);
    FILE *fp = fopen(argv[1], "rb");
    blob.resize(fread(blob.data(), 1, blob.size(), fp));
    fclose(fp);
    ea1_av1_tables *t;
    if (ea1_av1_tables_load(blob.data(), blob.size(), &t) != EA1_OK) return printf("tables: %s\n", ea1_last_error()), 1;
    std::vector<std::vector<uint8_t>> units;
    fp = fopen(argv[2], "rb");
    uint32_t n;
    while (fread(&n, 4, 1, fp) == 1) {
        units.emplace_back(n);
        if (fread(units.back().data(), 1, n, fp) != n) return 2;
    }
    fclose(fp);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    itimerval tv = {{0, 200}, {0, 200}};
    setitimer(ITIMER_PROF, &tv, nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    size_t frames = 0;
    for (int rep = 0; rep < repeats; rep++) {
        ea1_av1_frame_parser *p;
        ea1_av1_frame_parser_create(t, &p);
        for (auto &u : units) {
            ea1_av1_parsed out;
            if (ea1_av1_frame_parser_parse(p, u.data(), u.size(), &out) != EA1_OK) return printf("parse: %s\n", ea1_last_error()), 3;
            frames++;
        }
        ea1_av1_frame_parser_destroy(p);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    itimerval off = {{0, 0}, {0, 0}};
    setitimer(ITIMER_PROF, &off, nullptr);
    printf("%.3f ms per frame (%zu frames)\n", ms / double(frames), frames);
    FILE *o = fopen("/tmp/hp/pcs_av1.txt", "w");
    for (size_t i = 0; i < npcs; i++) fprintf(o, "0x%llx\n", (unsigned long long)pcs[i]);
    fclose(o);
    return 0;
}
