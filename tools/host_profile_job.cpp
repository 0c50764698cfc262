// Host cost of ea1_job_create without a device (the EA1_FUZZ_HOOK build of av1_recon.cpp: every host phase runs, the
// device allocation at the end fails), with a SIGPROF sampler that attributes the time to source lines:
//   bash tools/host_profile_job.sh [1080p|4k] [iterations]
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "easyav1_amd_synth.h"
#include "easyav1_amd_video.h"

extern "C" int ea1_fuzz_job_create(uint32_t width, uint32_t height, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                   const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3]);

static uint64_t pcs[1 << 20];
static volatile size_t npcs = 0;
static void on_prof(int, siginfo_t *, void *uc)
{
    if (npcs < (sizeof pcs / sizeof pcs[0])) pcs[npcs++] = uint64_t(static_cast<ucontext_t *>(uc)->uc_mcontext.gregs[REG_RIP]);
}

int main(int argc, char **argv)
{
    const bool k4 = argc > 1 && !strcmp(argv[1], "4k");
    const int iters = argc > 2 ? atoi(argv[2]) : 100;
    ea1_synth_params p;
    ea1_synth_default_params(&p, k4 ? 3840 : 1920, k4 ? 2160 : 1080, k4 ? 4 : 1, 1);
    ea1_synth_frame *f[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; k++)
        if (ea1_synth_frame_create(&p, uint32_t(k), &f[k]) != EA1_OK) return 3;
    printf("frame 1: %u blocks, %u transform blocks, %u coefficients\n", f[1]->desc.num_blocks, f[1]->desc.num_tx, f[1]->desc.num_coefs);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    itimerval tv = {{0, 200}, {0, 200}};
    for (int k = 0; k < 4; k++) {   // warm
        const ea1_synth_frame *s = f[k];
        ea1_fuzz_job_create(p.width, p.height, &s->desc, s->blocks, s->txs, s->coefs, s->aux, s->cdef_idx, s->lr);
    }
    // first without the sampler: the best of batches of 30 frames (a shared machine's noise is one-sided)
    double best = 1e9;
    for (int batch = 0; batch < std::max(iters / 30, 1); batch++) {
        const auto b0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 30; it++) {
            const ea1_synth_frame *s = f[1 + it % 3];
            ea1_fuzz_job_create(p.width, p.height, &s->desc, s->blocks, s->txs, s->coefs, s->aux, s->cdef_idx, s->lr);
        }
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count() / 30);
    }
    printf("%.3f ms per inter frame: best batch of 30, no sampler\n", best);
    setitimer(ITIMER_PROF, &tv, nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++) {
        const ea1_synth_frame *s = f[1 + it % 3];
        ea1_fuzz_job_create(p.width, p.height, &s->desc, s->blocks, s->txs, s->coefs, s->aux, s->cdef_idx, s->lr);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    itimerval off = {{0, 0}, {0, 0}};
    setitimer(ITIMER_PROF, &off, nullptr);
    printf("%.3f ms per inter frame (host phases only, %d iterations), last error: %s\n", ms / iters, iters, ea1_last_error());
    FILE *o = fopen("/tmp/hp/pcs.txt", "w");
    for (size_t i = 0; i < npcs; i++) fprintf(o, "0x%llx\n", (unsigned long long)pcs[i]);
    fclose(o);
    return 0;
}
