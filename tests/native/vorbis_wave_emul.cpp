// TEST INFRASTRUCTURE: the lane program of the Vorbis synthesis kernels (easyav1_amd/csrc/vorbis_synth_wave.h) run on the HOST --
// 64 threads stand for the 64 lanes of a wave, a barrier for the wave-level fence, plain arrays for LDS and device memory -- so that
// the schedule (who computes what, in which order, through which LDS slot) is checked against the oracle without a GPU.  The
// arithmetic is the header's own code compiled for the host with -ffp-contract=off; nothing here is shipped.
#include <cstring>
#include <thread>
#include <vector>

#include <pthread.h>

#include "vorbis_synth_wave.h"

using namespace ea1::vsynth;

namespace {
struct HostSync {
    pthread_barrier_t *b;
    void operator()() const { pthread_barrier_wait(b); }
};
template <class F>
void run_wave(F &&body)
{
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, kWave);
    std::vector<std::thread> t;
    for (int lane = 0; lane < kWave; lane++) t.emplace_back([&, lane] { body(lane, HostSync{&bar}); });
    for (auto &x : t) x.join();
    pthread_barrier_destroy(&bar);
}
int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}
// the tables as a workgroup sees them: the compact stage twiddles made from mdct_init's table, everything copied to "LDS" by copy_tables
struct Tables {
    std::vector<float> tw, lds;
    MdctTables view;
    Tables(int n, const float *trig, const float *win) : tw(tw_floats(n) + 1), lds(table_floats(n))
    {
        build_stage_twiddles(trig, n, tw.data());
        const MdctTables g{trig, tw.data(), win, n, ilog2(n)};
        for (int tid = 0; tid < 512; tid++) view = copy_tables(lds.data(), g, tid, 512);
    }
};
}  // namespace

extern "C" void ea1emul_mdct_backward(int n, const float *trig, const float *in, size_t in_stride, float *out, size_t out_stride, int count)
{
    const Tables tabs(n, trig, nullptr);
    const MdctTables tb = tabs.view;
    std::vector<float> lds(lds_transform_floats(n) + 4);
    float *l = lds.data();
    while (reinterpret_cast<uintptr_t>(l) & 15) l++;
    for (int k = 0; k < count; k++) {
        const float *src = in + size_t(k) * in_stride;
        const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        for (int i = 0; i < lds_transform_floats(n); i++) l[i] = -12345.0f;
        run_wave([&](int lane, HostSync s) { wave_mdct_backward(l, tb, src, aligned, out + size_t(k) * out_stride, lane, s); });
    }
}

// one decode call of one stream, every (run, channel) in turn; same arguments as the kernel's
extern "C" void ea1emul_decode(int bs0, int bs1, const float *trig0, const float *win0, const float *trig1, const float *win1, const float *spectra,
                               const BlockMeta *meta, int nblocks, int channels, float *pcm, size_t pcm_stride, const float *tail_in, float *tail_out)
{
    const Tables tabs0(bs0, trig0, win0), tabs1(bs1, trig1, win1);
    const MdctTables tb0 = tabs0.view, tb1 = tabs1.view;
    std::vector<float> lds(lds_decode_floats(bs1) + 4);
    float *l = lds.data();
    while (reinterpret_cast<uintptr_t>(l) & 15) l++;
    for (int run = 0; run * kRun < nblocks; run++)
        for (int ch = 0; ch < channels; ch++) {
            for (int i = 0; i < lds_decode_floats(bs1); i++) l[i] = -12345.0f;
            run_wave([&](int lane, HostSync s) {
                wave_decode_run(tb0, tb1, spectra, meta, nblocks, channels, pcm, pcm_stride, tail_in, tail_out, l, run, ch, lane, s);
            });
        }
}

extern "C" int ea1emul_block_meta_size() { return int(sizeof(BlockMeta)); }
extern "C" int ea1emul_run_length() { return kRun; }
