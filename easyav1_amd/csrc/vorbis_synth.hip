// Vorbis synthesis on CDNA4 (gfx950): IMDCT + window + overlap-add, float32, bit-exact with
// libvorbis 1.3.7's scalar C code (the reference's ext/minivorbis/minivorbis.h).
//
// Must be compiled with -ffp-contract=off: the reference computes a*b + c*d as two rounded
// multiplies and one rounded add; an FMA changes the last bit (SURVEY.md finding 8).
//
// What is restated (reference line numbers are minivorbis.h):
//   mdct_backward :7054-7148 with mdct_butterflies :6974, _first/_generic :6874-6972,
//   _32/_16/_8 :6751-6871, mdct__bitreverse :7004-7052; tables as mdct_init :6709-6748 (built on
//   the HOST with libm in double precision and uploaded -- device cos/sin would not match);
//   windows vwin* :9917-11979; vorbis_synthesis_blockin :9195-9408; pcmout/read :9411-9430.
//
// Mapping to the hardware: ONE 64-lane wave per transform (vorbis_synth_wave.h has the lane program and the reasons): the
// butterflies run on registers, radix 8 / 4 between LDS round trips and the whole 32-float butterfly at the end; the lanes of
// a wave need no barrier, only a wave-level fence, so a workgroup is one wave and a CU keeps ~18 transforms in flight.
// The decode kernel fuses bit-reverse step, post-rotation, windowing, overlap-add and the pcmout copy into the consumer, so
// HBM traffic is the algorithmic minimum: n/2 floats of spectrum in (one coalesced float4 per work item), bs[lW]/4+bs[W]/4
// floats of PCM out per (block, channel).  A wave walks a run of kRun consecutive blocks of one channel and keeps the previous
// block's right half in LDS; only the block before a run is transformed twice (once here for its right half, once by the
// neighbouring run), i.e. 1/kRun extra work and no extra traffic except that block's spectrum.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include <algorithm>

#include "ea1_common.h"
#include "vorbis_synth_wave.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

using namespace ea1::vsynth;

struct WaveSync {  // lanes of one wave exchange data through LDS: order the LDS accesses; loads and stores of global memory stay in flight
    __device__ __forceinline__ void operator()() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
};

// Up to kMaxWaves waves share a workgroup and its LDS copy of the tables: one wave per SIMD, so that workgroups stack evenly; three such
// workgroups per CU are 12 waves = 3 per SIMD, which is what the registers allow (~160 VGPRs: a transform's 16 elements, their
// twiddles, and the next block's spectrum already in flight).  (Six-wave workgroups place 2+2+1+1 waves on the SIMDs and a second one
// no longer fits: measured 1.66 ms against 1.11 ms per launch of the bench's audio workload, profiles/r04_r_audio_stage_probe.txt.)
constexpr int kMaxWaves = 4;

// LDS of a workgroup: [tables of bs1][tables of bs0, when the sizes differ][one region per wave]; without `lds_tables` the waves read
// the tables from global memory (sizes whose tables do not fit) and the tables' part is absent
struct WgTables {
    MdctTables tb0, tb1;
    float *waves;
};
template <bool LDS_TABLES>
__device__ __forceinline__ WgTables stage_tables(float *smem, const MdctTables &g0, const MdctTables &g1)
{
    WgTables t{g0, g1, smem};
    if (LDS_TABLES) {
        t.tb1 = copy_tables(smem, g1, int(threadIdx.x), int(blockDim.x));
        t.waves = smem + table_floats(g1.n);
        if (g0.n != g1.n) {
            t.tb0 = copy_tables(t.waves, g0, int(threadIdx.x), int(blockDim.x));
            t.waves += table_floats(g0.n);
        } else {
            t.tb0 = t.tb1;
        }
        __syncthreads();  // the only block-wide barrier: from here on the waves run on their own
    }
    return t;
}

// a wave per transform, `waves` transforms per workgroup: grid.x = ceil(count / waves)
template <bool LDS_TABLES>
__global__ void __launch_bounds__(kMaxWaves * kWave, 3)
mdct_backward_kernel(MdctTables tb, const float *__restrict__ in, size_t in_stride, float *__restrict__ out, size_t out_stride, int count, int aligned)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgTables t = stage_tables<LDS_TABLES>(smem, tb, tb);
    const int wave = int(threadIdx.x) / kWave, lane = int(threadIdx.x) % kWave;
    const int k = int(blockIdx.x) * (int(blockDim.x) / kWave) + wave;
    if (k >= count) return;
    wave_mdct_backward(t.waves + wave * lds_transform_floats(tb.n), t.tb1, in + size_t(k) * in_stride, aligned != 0, out + size_t(k) * out_stride, lane, WaveSync{});
}

// one stream's share of a decode launch
struct StreamArgs {
    MdctTables tb0, tb1;
    const float *spectra;
    const BlockMeta *meta;
    float *pcm;
    const float *tail_in;
    float *tail_out;
    size_t pcm_stride;
    int nblocks, channels;
};
constexpr int kBatchStreams = 16;   // streams served by one launch (their arguments travel by value: ~2 KB of kernel arguments)
struct BatchArgs {
    StreamArgs s[kBatchStreams];
};

// workgroup (bx, ch) of one stream: its waves take the runs bx * waves ... of channel ch
template <bool LDS_TABLES>
__device__ __forceinline__ void decode_workgroup(const StreamArgs &s, float *smem)
{
    const int waves = int(blockDim.x) / kWave;
    if (int(blockIdx.x) * waves * kRun >= s.nblocks || int(blockIdx.y) >= s.channels) return;   // uniform per workgroup
    const WgTables t = stage_tables<LDS_TABLES>(smem, s.tb0, s.tb1);
    const int wave = int(threadIdx.x) / kWave, lane = int(threadIdx.x) % kWave;
    const int run = int(blockIdx.x) * waves + wave;
    if (run * kRun >= s.nblocks) return;   // uniform per wave, after the barrier
    wave_decode_run(t.tb0, t.tb1, s.spectra, s.meta, s.nblocks, s.channels, s.pcm, s.pcm_stride, s.tail_in, s.tail_out,
                    t.waves + wave * lds_decode_floats(s.tb1.n), run, int(blockIdx.y), lane, WaveSync{});
}

// grid = (ceil(runs / waves), channels)
template <bool LDS_TABLES>
__global__ void __launch_bounds__(kMaxWaves * kWave, 3)
synth_decode_kernel(StreamArgs s)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    decode_workgroup<LDS_TABLES>(s, smem);
}

// several streams in one launch: grid = (max workgroups, max channels, streams); a stream with fewer runs / channels
// leaves the surplus workgroups idle (they return at once)
template <bool LDS_TABLES>
__global__ void __launch_bounds__(kMaxWaves * kWave, 3)
synth_decode_batch_kernel(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    decode_workgroup<LDS_TABLES>(a.s[blockIdx.z], smem);
}

// how many waves share a workgroup (and its LDS copy of the tables) for the block sizes (bs0, bs1)
struct Shape {
    int waves;
    int lds_tables;
    size_t lds_bytes;
};
constexpr int kLdsFloats = 160 * 1024 / 4;
Shape shape_for(int bs0, int bs1, int per_wave)
{
    const int tf = table_floats(bs1) + (bs0 != bs1 ? table_floats(bs0) : 0);
    int w = (kLdsFloats / 3 - tf) / per_wave;          // three workgroups per CU
    if (w >= kMaxWaves) return Shape{kMaxWaves, 1, size_t(tf + kMaxWaves * per_wave) * sizeof(float)};
    w = (kLdsFloats / 2 - tf) / per_wave;              // two
    if (w >= 3) return Shape{std::min(w, kMaxWaves), 1, size_t(tf + std::min(w, kMaxWaves) * per_wave) * sizeof(float)};
    w = (kLdsFloats - tf) / per_wave;                  // one
    if (w >= 1) return Shape{std::min(w, kMaxWaves), 1, size_t(tf + std::min(w, kMaxWaves) * per_wave) * sizeof(float)};
    return Shape{1, 0, size_t(per_wave) * sizeof(float)};   // tables stay in global memory
}
template <class K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return EA1_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)) != hipSuccess)
        return ea1::fail(EA1_EHIP, "hipFuncSetAttribute(%zu bytes of LDS) failed", bytes);
    return EA1_OK;
}

// ---------------------------------------------------------------- host side
struct HostTables {
    std::vector<float> trig, tw, win;
};

// mdct_init :6709 + window literal reconstruction (see oracle/vorbis_synth.c:ea1o_window for why
// the "%.10f" round trip reproduces the reference's tables bit-for-bit).
void build_tables(int n, HostTables &h)
{
    const int n2 = n >> 1;
    h.trig.resize(n + n / 4);
    h.win.resize(n / 2);
    float *T = h.trig.data();
    for (int i = 0; i < n / 4; i++) {
        T[i * 2] = float(cos((M_PI / n) * (4 * i)));
        T[i * 2 + 1] = float(-sin((M_PI / n) * (4 * i)));
        T[n2 + i * 2] = float(cos((M_PI / (2 * n)) * (2 * i + 1)));
        T[n2 + i * 2 + 1] = float(sin((M_PI / (2 * n)) * (2 * i + 1)));
    }
    for (int i = 0; i < n / 8; i++) {
        T[n + i * 2] = float(cos((M_PI / n) * (4 * i + 2)) * .5);
        T[n + i * 2 + 1] = float(-sin((M_PI / n) * (4 * i + 2)) * .5);
    }
    // (the bit-reverse table of :6726-6736 is index arithmetic in the kernels: vorbis_synth_wave.h final_pass / combine)
    h.tw.assign(std::max(tw_floats(n), 1), 0.f);
    build_stage_twiddles(T, n, h.tw.data());
    char buf[64];
    for (int i = 0; i < n / 2; i++) {
        const double s = sin((i + 0.5) / n * M_PI);
        snprintf(buf, sizeof buf, "%.10f", sin(0.5 * M_PI * s * s));
        h.win[i] = float(strtod(buf, nullptr));
    }
}

struct DeviceTables {
    float *trig = nullptr, *tw = nullptr, *win = nullptr;
    int n = 0;
    int upload(int n_)
    {
        HostTables h;
        build_tables(n_, h);
        n = n_;
        EA1_HIP(hipMalloc(&trig, h.trig.size() * sizeof(float)));
        EA1_HIP(hipMalloc(&win, h.win.size() * sizeof(float)));
        EA1_HIP(hipMalloc(&tw, h.tw.size() * sizeof(float)));
        EA1_HIP(hipMemcpy(tw, h.tw.data(), h.tw.size() * sizeof(float), hipMemcpyHostToDevice));
        EA1_HIP(hipMemcpy(trig, h.trig.data(), h.trig.size() * sizeof(float), hipMemcpyHostToDevice));
        EA1_HIP(hipMemcpy(win, h.win.data(), h.win.size() * sizeof(float), hipMemcpyHostToDevice));
        return EA1_OK;
    }
    void release()
    {
        if (trig) (void)hipFree(trig);
        if (win) (void)hipFree(win);
        if (tw) (void)hipFree(tw);
        trig = win = tw = nullptr;
    }
    MdctTables view() const { return MdctTables{trig, tw, win, n, ea1::ilog2(n)}; }
};

bool valid_blocksize(int n) { return ea1::is_pow2(n) && n >= 64 && n <= 8192; }

}  // namespace

struct ea1_vorbis_synth {
    int device = 0;
    int channels = 0;
    int bs[2] = {0, 0};
    DeviceTables tb[2];
    float *d_tail = nullptr;      // 2 x [channels][bs1/2]: right half of the last block seen (double-buffered)
    int tail_cur = 0;             // which half holds the current state
    BlockMeta *d_meta = nullptr;  // device copy of the per-block metadata
    BlockMeta *h_meta = nullptr;  // pinned staging
    int meta_cap = 0;
    hipEvent_t meta_free = nullptr;  // h_meta may be rewritten once this has fired
    // blockin state (v->W, pcm_returned == -1)
    int W = 0;
    bool has_prev = false;
};

namespace {

// fills counts for a flag sequence starting from state (W, has_prev); returns the total
int64_t plan(const ea1_vorbis_synth *s, const uint8_t *flags, int nblocks, BlockMeta *meta, int *counts)
{
    int lW = s->W;
    bool has_prev = s->has_prev;
    int64_t off = 0;
    for (int k = 0; k < nblocks; k++) {
        const int W = flags[k] ? 1 : 0;
        const int cnt = has_prev ? s->bs[lW] / 4 + s->bs[W] / 4 : 0;
        if (meta) meta[k] = BlockMeta{off, cnt, uint8_t(W), uint8_t(lW), uint8_t(has_prev), 0};
        if (counts) counts[k] = cnt;
        off += cnt;
        lW = W;
        has_prev = true;
    }
    return off;
}

}  // namespace

extern "C" int ea1_vorbis_synth_create(ea1_vorbis_synth **out, int channels, int blocksize0,
                                       int blocksize1, int device)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_create: out is NULL");
    *out = nullptr;
    if (channels <= 0 || channels > 255 || !valid_blocksize(blocksize0) ||
        !valid_blocksize(blocksize1) || blocksize1 < blocksize0)
        return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_create: bad channels/blocksizes (%d, %d, %d)",
                         channels, blocksize0, blocksize1);
    int ndev = ea1_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev)
        return ea1::fail(EA1_ENODEV, "ea1_vorbis_synth_create: device %d not available (%d visible)", device, ndev);
    ea1::DeviceGuard g(device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", device);
    auto *s = new (std::nothrow) ea1_vorbis_synth();
    if (!s) return ea1::fail(EA1_ENOMEM, "out of memory");
    s->device = device;
    s->channels = channels;
    s->bs[0] = blocksize0;
    s->bs[1] = blocksize1;
    int rc = s->tb[0].upload(blocksize0);
    if (rc == EA1_OK) rc = s->tb[1].upload(blocksize1);
    const size_t tail_bytes = 2 * size_t(channels) * (blocksize1 / 2) * sizeof(float);
    if (rc == EA1_OK && hipMalloc(&s->d_tail, tail_bytes) != hipSuccess)
        rc = ea1::fail(EA1_ENOMEM, "hipMalloc(tail) failed");
    if (rc == EA1_OK && hipMemset(s->d_tail, 0, tail_bytes) != hipSuccess)
        rc = ea1::fail(EA1_EHIP, "hipMemset(tail) failed");
    if (rc == EA1_OK && hipEventCreateWithFlags(&s->meta_free, hipEventDisableTiming) != hipSuccess)
        rc = ea1::fail(EA1_EHIP, "hipEventCreate failed");
    if (rc != EA1_OK) {
        ea1_vorbis_synth_destroy(s);
        return rc;
    }
    *out = s;
    return EA1_OK;
}

extern "C" void ea1_vorbis_synth_destroy(ea1_vorbis_synth *s)
{
    if (!s) return;
    ea1::DeviceGuard g(s->device);
    (void)hipDeviceSynchronize();
    s->tb[0].release();
    s->tb[1].release();
    if (s->d_tail) (void)hipFree(s->d_tail);
    if (s->d_meta) (void)hipFree(s->d_meta);
    if (s->h_meta) (void)hipHostFree(s->h_meta);
    if (s->meta_free) (void)hipEventDestroy(s->meta_free);
    delete s;
}

extern "C" int ea1_vorbis_synth_restart(ea1_vorbis_synth *s)
{
    if (!s) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_restart: NULL handle");
    // vorbis_synthesis_restart :9159 resets pcm_returned to -1 and leaves W alone
    s->has_prev = false;
    return EA1_OK;
}

extern "C" int64_t ea1_vorbis_synth_count(const ea1_vorbis_synth *s, const uint8_t *flags, int nblocks)
{
    if (!s || nblocks < 0 || (nblocks > 0 && !flags)) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_count: bad arguments");
    return plan(s, flags, nblocks, nullptr, nullptr);
}

namespace {

// room for `blocks` block plans in s's staging buffers; on return the previous upload has left the pinned one
int reserve_meta(ea1_vorbis_synth *s, int blocks)
{
    if (blocks > s->meta_cap) {
        EA1_HIP(hipDeviceSynchronize());
        if (s->d_meta) (void)hipFree(s->d_meta);
        if (s->h_meta) (void)hipHostFree(s->h_meta);
        s->d_meta = nullptr;
        s->h_meta = nullptr;
        s->meta_cap = 0;
        const int cap = blocks + blocks / 2 + 64;
        EA1_HIP(hipMalloc(&s->d_meta, size_t(cap) * sizeof(BlockMeta)));
        EA1_HIP(hipHostMalloc(&s->h_meta, size_t(cap) * sizeof(BlockMeta)));
        s->meta_cap = cap;
    } else {
        EA1_HIP(hipEventSynchronize(s->meta_free));
    }
    return EA1_OK;
}

// host part of a decode call: the block plan into h_meta (its device copy will be d_meta) and the kernel's arguments into `a`
int prepare_decode(ea1_vorbis_synth *s, const float *d_spectra, const uint8_t *flags, int nblocks, float *d_pcm, size_t pcm_stride,
                   int *counts, int64_t *total, BlockMeta *h_meta, const BlockMeta *d_meta, StreamArgs *a)
{
    const int64_t tot = plan(s, flags, nblocks, h_meta, counts);
    if (size_t(tot) > pcm_stride)
        return ea1::fail(EA1_ECAPACITY, "ea1_vorbis_synth_decode: %lld samples > pcm_stride %zu",
                         (long long)tot, pcm_stride);
    a->tb0 = s->tb[0].view();
    a->tb1 = s->tb[1].view();
    a->spectra = d_spectra;
    a->meta = d_meta;
    a->pcm = d_pcm;
    a->pcm_stride = pcm_stride;
    a->tail_in = s->d_tail + size_t(s->tail_cur) * s->channels * (s->bs[1] / 2);
    a->tail_out = s->d_tail + size_t(s->tail_cur ^ 1) * s->channels * (s->bs[1] / 2);
    a->nblocks = nblocks;
    a->channels = s->channels;
    if (total) *total = tot;
    return EA1_OK;
}

// block plans staged in s's pinned buffer -> device, ordered on `st`
int upload_meta(ea1_vorbis_synth *s, int blocks, hipStream_t st)
{
    EA1_HIP(hipMemcpyAsync(s->d_meta, s->h_meta, size_t(blocks) * sizeof(BlockMeta), hipMemcpyHostToDevice, st));
    EA1_HIP(hipEventRecord(s->meta_free, st));
    return EA1_OK;
}

// ea1_vorbis_profile_*: events around the decode launches (process-wide; a measurement aid, not a hot path)
struct LaunchProf {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
} g_prof;
struct ProfSpan {   // brackets one launch on `st` while profiling is on
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    explicit ProfSpan(hipStream_t s) : st(s)
    {
        std::lock_guard<std::mutex> g(g_prof.mu);
        if (!g_prof.on) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            if (a) (void)hipEventDestroy(a);
            a = b = nullptr;
            return;
        }
        (void)hipEventRecord(a, st);
    }
    ~ProfSpan()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> g(g_prof.mu);
        g_prof.spans.emplace_back(a, b);
    }
};

void commit_decode(ea1_vorbis_synth *s, const uint8_t *flags, int nblocks)
{
    s->tail_cur ^= 1;
    s->W = flags[nblocks - 1] ? 1 : 0;
    s->has_prev = true;
}

}  // namespace

extern "C" int ea1_vorbis_synth_decode(ea1_vorbis_synth *s, const float *d_spectra,
                                       const uint8_t *flags, int nblocks, float *d_pcm,
                                       size_t pcm_stride, int *counts, int64_t *total, void *stream)
{
    if (!s || nblocks < 0) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode: bad arguments");
    if (total) *total = 0;
    if (nblocks == 0) return EA1_OK;
    if (!d_spectra || !flags || !d_pcm) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode: NULL buffer");
    ea1::DeviceGuard g(s->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", s->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    StreamArgs a;
    int rc = reserve_meta(s, nblocks);
    if (rc == EA1_OK) rc = prepare_decode(s, d_spectra, flags, nblocks, d_pcm, pcm_stride, counts, total, s->h_meta, s->d_meta, &a);
    if (rc == EA1_OK) rc = upload_meta(s, nblocks, st);
    if (rc != EA1_OK) return rc;
    const Shape sh = shape_for(s->bs[0], s->bs[1], lds_decode_floats(s->bs[1]));
    const int runs = (nblocks + kRun - 1) / kRun;
    auto kernel = sh.lds_tables ? synth_decode_kernel<true> : synth_decode_kernel<false>;
    const int arc = allow_lds(kernel, sh.lds_bytes);
    if (arc != EA1_OK) return arc;
    {
        ProfSpan span(st);
        hipLaunchKernelGGL(kernel, dim3((runs + sh.waves - 1) / sh.waves, s->channels), dim3(sh.waves * kWave), sh.lds_bytes, st, a);
    }
    EA1_LAUNCH_CHECK();
    commit_decode(s, flags, nblocks);
    return EA1_OK;
}

// The audio of several streams (a player with many files open; bench.py's 16 x 5.1 workload) in ONE launch per 16
// streams instead of one launch per stream: a 60 s stream is ~2800 transforms per channel, a launch of its own for
// every stream leaves the chip idle between them.  Same results as n calls of ea1_vorbis_synth_decode.
extern "C" int ea1_vorbis_synth_decode_batch(int n, ea1_vorbis_synth *const *synths, const float *const *d_spectra, const uint8_t *const *flags,
                                             const int *nblocks, float *const *d_pcm, const size_t *pcm_stride, int64_t *totals, void *stream)
{
    if (n < 0 || (n && (!synths || !d_spectra || !flags || !nblocks || !d_pcm || !pcm_stride))) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode_batch: bad arguments");
    for (int k = 0; k < n; k++) {
        if (totals) totals[k] = 0;
        if (!synths[k] || nblocks[k] < 0 || (nblocks[k] && (!d_spectra[k] || !flags[k] || !d_pcm[k])))
            return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode_batch: stream %d: bad arguments", k);
        if (synths[k]->device != synths[0]->device) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode_batch: streams on different devices");
        for (int j = 0; j < k; j++)
            if (synths[j] == synths[k]) return ea1::fail(EA1_EINVAL, "ea1_vorbis_synth_decode_batch: stream %d given twice", k);
    }
    if (!n) return EA1_OK;
    ea1::DeviceGuard g(synths[0]->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", synths[0]->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += kBatchStreams) {
        BatchArgs ba;
        memset(&ba, 0, sizeof ba);
        int live = 0, max_runs = 0, max_ch = 0;
        Shape sh{kMaxWaves, 1, 0};   // the launch's shape: the fewest waves and the most LDS any of its streams asks for
        int idx[kBatchStreams];
        // the block plans of the launch travel in ONE copy, through the staging buffers of its first stream
        ea1_vorbis_synth *lead = nullptr;
        int plans = 0;
        for (int k = base; k < n && k < base + kBatchStreams; k++) {
            if (nblocks[k] && !lead) lead = synths[k];
            plans += nblocks[k];
        }
        if (!lead) continue;
        int rc = reserve_meta(lead, plans);
        if (rc != EA1_OK) return rc;
        plans = 0;
        for (int k = base; k < n && k < base + kBatchStreams; k++) {
            if (!nblocks[k]) continue;
            rc = prepare_decode(synths[k], d_spectra[k], flags[k], nblocks[k], d_pcm[k], pcm_stride[k], nullptr, totals ? &totals[k] : nullptr, lead->h_meta + plans,
                                lead->d_meta + plans, &ba.s[live]);
            if (rc != EA1_OK) return rc;   // no stream of the launch has been committed: their state is unchanged
            plans += nblocks[k];
            max_runs = std::max(max_runs, (nblocks[k] + kRun - 1) / kRun);
            max_ch = std::max(max_ch, synths[k]->channels);
            const Shape one = shape_for(synths[k]->bs[0], synths[k]->bs[1], lds_decode_floats(synths[k]->bs[1]));
            sh.waves = std::min(sh.waves, one.waves);
            sh.lds_tables &= one.lds_tables;
            idx[live++] = k;
        }
        rc = upload_meta(lead, plans, st);
        if (rc != EA1_OK) return rc;
        for (int i = 0; i < live; i++) {
            const ea1_vorbis_synth *y = synths[idx[i]];
            const size_t tf = sh.lds_tables ? size_t(table_floats(y->bs[1]) + (y->bs[0] != y->bs[1] ? table_floats(y->bs[0]) : 0)) : 0;
            sh.lds_bytes = std::max(sh.lds_bytes, (tf + size_t(sh.waves) * lds_decode_floats(y->bs[1])) * sizeof(float));
        }
        auto kernel = sh.lds_tables ? synth_decode_batch_kernel<true> : synth_decode_batch_kernel<false>;
        const int arc = allow_lds(kernel, sh.lds_bytes);
        if (arc != EA1_OK) return arc;
        {
            ProfSpan span(st);
            hipLaunchKernelGGL(kernel, dim3((max_runs + sh.waves - 1) / sh.waves, max_ch, live), dim3(sh.waves * kWave), sh.lds_bytes, st, ba);
        }
        EA1_LAUNCH_CHECK();
        for (int i = 0; i < live; i++) commit_decode(synths[idx[i]], flags[idx[i]], nblocks[idx[i]]);
    }
    return EA1_OK;
}

extern "C" int ea1_vorbis_profile_enable(int on)
{
    std::lock_guard<std::mutex> g(g_prof.mu);
    g_prof.on = on != 0;
    return EA1_OK;
}

extern "C" int ea1_vorbis_profile_read(double *ms_total, uint64_t *launches)
{
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    {
        std::lock_guard<std::mutex> g(g_prof.mu);
        spans.swap(g_prof.spans);
    }
    double ms = 0;
    for (auto &sp : spans) {
        float t = 0;
        if (hipEventSynchronize(sp.second) == hipSuccess && hipEventElapsedTime(&t, sp.first, sp.second) == hipSuccess) ms += t;
        (void)hipEventDestroy(sp.first);
        (void)hipEventDestroy(sp.second);
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = spans.size();
    return EA1_OK;
}

extern "C" int ea1_mdct_backward_batch(int n, const float *d_in, size_t in_stride, float *d_out,
                                       size_t out_stride, int count, int device, void *stream)
{
    if (!valid_blocksize(n) || count < 0 || in_stride < size_t(n / 2) || out_stride < size_t(n))
        return ea1::fail(EA1_EINVAL, "ea1_mdct_backward_batch: bad n/count/strides");
    if (count == 0) return EA1_OK;
    if (!d_in || !d_out) return ea1::fail(EA1_EINVAL, "ea1_mdct_backward_batch: NULL buffer");
    ea1::DeviceGuard g(device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    DeviceTables tb;
    int rc = tb.upload(n);
    if (rc != EA1_OK) {
        tb.release();
        return rc;
    }
    const int aligned = (reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && in_stride % 4 == 0;
    const Shape sh = shape_for(n, n, lds_transform_floats(n));
    auto kernel = sh.lds_tables ? mdct_backward_kernel<true> : mdct_backward_kernel<false>;
    rc = allow_lds(kernel, sh.lds_bytes);
    if (rc != EA1_OK) {
        tb.release();
        return rc;
    }
    MdctTables view = tb.view();
    view.win = nullptr;   // a bare transform has no use for the window
    hipLaunchKernelGGL(kernel, dim3((count + sh.waves - 1) / sh.waves), dim3(sh.waves * kWave), sh.lds_bytes, st, view, d_in, in_stride, d_out, out_stride, count,
                       aligned);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // tables are freed below
    tb.release();
    if (e != hipSuccess) return ea1::fail(EA1_EHIP, "mdct_backward_kernel: %s", hipGetErrorString(e));
    return EA1_OK;
}
