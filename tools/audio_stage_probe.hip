// Where a wave of the Vorbis synthesis kernel spends its time: the lane program of easyav1_amd/csrc/vorbis_synth_wave.h with a Sync
// functor that also records s_memtime at every stage boundary (lane 0 of a few sampled waves), under a full grid.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I easyav1_amd/csrc tools/audio_stage_probe.hip -o tools/bin/audio_stage_probe && tools/bin/audio_stage_probe
// Table VALUES are random (timing does not depend on them); sizes and access patterns are the bench's (6 channels, 256 / 2048).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vorbis_synth_wave.h"
using namespace ea1::vsynth;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMaxMarks = 96;
struct ProbeSync {
    unsigned long long *log;  // nullptr: not a sampled wave
    int *n;
    __device__ __forceinline__ void operator()() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if (log && threadIdx.x % kWave == 0 && *n < kMaxMarks) log[(*n)++] = __builtin_amdgcn_s_memtime();
    }
};

// the workgroup shape of vorbis_synth.hip's decode kernels: `waves` waves share one LDS copy of the tables (lds_tables = 0: tables in global memory)
__global__ void __launch_bounds__(768, 3)
probe_kernel(MdctTables g0, MdctTables g1, const float *spectra, const BlockMeta *meta, int nblocks, int channels, float *pcm, size_t pcm_stride,
             const float *tail_in, float *tail_out, unsigned long long *logs, int sample_every, size_t spec_per_stream, int lds_tables)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int counts[16];
    spectra += spec_per_stream * blockIdx.z;   // every stream its own spectra and PCM (same block plan)
    pcm += pcm_stride * channels * blockIdx.z;
    const int waves = blockDim.x / kWave, wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    MdctTables tb0 = g0, tb1 = g1;
    float *wl = smem;
    if (lds_tables) {
        tb1 = copy_tables(smem, g1, threadIdx.x, blockDim.x);
        tb0 = copy_tables(smem + table_floats(g1.n), g0, threadIdx.x, blockDim.x);
        wl = smem + table_floats(g1.n) + table_floats(g0.n);
        __syncthreads();
    }
    const int run = blockIdx.x * waves + wave;
    if (run * kRun >= nblocks) return;
    const int id = run + ((nblocks + kRun - 1) / kRun) * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long *log = (id % sample_every == 0) ? logs + size_t(id / sample_every) * kMaxMarks : nullptr;
    if (lane == 0) {
        counts[wave] = 0;
        if (log) log[counts[wave]++] = __builtin_amdgcn_s_memtime();
    }
    wave_decode_run(tb0, tb1, spectra, meta, nblocks, channels, pcm, pcm_stride, tail_in, tail_out, wl + wave * lds_decode_floats(g1.n), run, int(blockIdx.y), lane,
                    ProbeSync{log, &counts[wave]});
}

int main(int argc, char **argv)
{
    const int bs0 = 256, bs1 = 2048, ch = 6, streams = 16, nblocks = 3213;
    const int long_percent = argc > 1 ? atoi(argv[1]) : 85;
    std::vector<float> t0(bs0 + bs0 / 4), t1(bs1 + bs1 / 4), w0(bs0 / 2), w1(bs1 / 2);
    srand(1);
    auto fill = [](std::vector<float> &v) { for (auto &x : v) x = float(rand() % 20001) / 10000.0f - 1; };
    fill(t0); fill(t1); fill(w0); fill(w1);
    std::vector<BlockMeta> meta(nblocks);
    int lW = 0; int64_t off = 0;
    int bs[2] = {bs0, bs1};
    int shorts_left = 0;
    for (int k = 0; k < nblocks; k++) {
        if (!shorts_left && (rand() % 10000) < (100 - long_percent) * 100 / 8 * 100 / long_percent) shorts_left = 8;   // shorts in runs of 8, like bench.py
        const int W = shorts_left ? 0 : 1;
        if (shorts_left) shorts_left--;
        const int cnt = k ? bs[lW] / 4 + bs[W] / 4 : 0;
        meta[k] = BlockMeta{off, cnt, uint8_t(W), uint8_t(lW), uint8_t(k > 0), 0};
        off += cnt; lW = W;
    }
    const size_t spec_n = size_t(nblocks) * ch * (bs1 / 2);
    std::vector<float> spec(spec_n);
    fill(spec);
    float *d_t0, *d_t1, *d_w0, *d_w1, *d_spec, *d_pcm, *d_tail;
    BlockMeta *d_meta;
    CK(hipMalloc(&d_t0, t0.size() * 4)); CK(hipMalloc(&d_t1, t1.size() * 4)); CK(hipMalloc(&d_w0, w0.size() * 4)); CK(hipMalloc(&d_w1, w1.size() * 4));
    CK(hipMemcpy(d_t0, t0.data(), t0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_t1, t1.data(), t1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w0, w0.data(), w0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_spec, spec_n * 4 * streams));
    for (int s = 0; s < streams; s++) CK(hipMemcpy(d_spec + spec_n * s, spec.data(), spec_n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_meta, meta.size() * sizeof(BlockMeta))); CK(hipMemcpy(d_meta, meta.data(), meta.size() * sizeof(BlockMeta), hipMemcpyHostToDevice));
    const size_t stride = size_t(off + 64);
    CK(hipMalloc(&d_pcm, stride * ch * 4 * streams));
    CK(hipMalloc(&d_tail, size_t(2) * ch * (bs1 / 2) * 4)); CK(hipMemset(d_tail, 0, size_t(2) * ch * (bs1 / 2) * 4));
    const int runs = (nblocks + kRun - 1) / kRun;
    const int sample_every = 997;
    const int waves = runs * ch * streams, sampled = waves / sample_every + 1;
    unsigned long long *d_log;
    CK(hipMalloc(&d_log, size_t(sampled) * kMaxMarks * 8)); CK(hipMemset(d_log, 0, size_t(sampled) * kMaxMarks * 8));
    const MdctTables tb0{d_t0, d_t0, d_w0, bs0, 8}, tb1{d_t1, d_t1, d_w1, bs1, 11};   // tw: any finite values of the right size
    // all streams share tables / meta; stream z reads its own spectra and writes its own pcm through pointer offsets in blockIdx.z:
    // simplest: one launch per "stream" would serialise, so the z dimension is folded into channels (6 x 16 rows of spectra / pcm)
    // -- spectra of stream z are a copy, so folding is just more (block, channel) rows with the same meta.
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; variant++) {
        const int wv = (int[]){4, 4, 12, 8}[variant], lt = (int[]){1, 0, 1, 1}[variant];
        const size_t lds = size_t((lt ? table_floats(bs0) + table_floats(bs1) : 0) + wv * lds_decode_floats(bs1)) * 4;
        if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(d_log, 0, size_t(sampled) * kMaxMarks * 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe_kernel, dim3((runs + wv - 1) / wv, ch, streams), dim3(wv * kWave), lds, 0, tb0, tb1, d_spec, d_meta, nblocks, ch, d_pcm, stride, d_tail,
                               d_tail + ch * (bs1 / 2), d_log, sample_every, spec_n, lt);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d waves per workgroup, tables in %s, %zu bytes of LDS: launch %d: %.3f ms for %d waves (%d transforms incl. the recomputed ones)\n", wv, lt ? "LDS" : "global memory",
                   lds, rep, ms, waves, waves * 9);
        }
        if (variant > 0) continue;
    std::vector<unsigned long long> log(size_t(sampled) * kMaxMarks);
    CK(hipMemcpy(log.data(), d_log, log.size() * 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on gfx9: print in ticks and us
    printf("marks of sampled waves (delta in s_memtime ticks of 10 ns): index 0 = entry\n");
    for (int w = 0; w < sampled && w < 12; w++) {
        printf("wave %5d:", w * sample_every);
        for (int i = 1; i < kMaxMarks && log[size_t(w) * kMaxMarks + i]; i++) printf(" %llu", log[size_t(w) * kMaxMarks + i] - log[size_t(w) * kMaxMarks + i - 1]);
        printf("\n");
    }
    }
    return 0;
}
