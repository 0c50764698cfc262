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
// Mapping to the hardware.  The reference is a pointer-walking in-place routine; every one of its
// stages is really a set of *independent* 2-point (complex) butterflies, so a transform is run by
// one 256-thread workgroup with the n floats resident in LDS and one __syncthreads() per stage:
//   stage                         independent work items      LDS traffic per item
//   pre-rotation  :7061-7087      n/8  (4 outputs each)       4 reads, 4 writes
//   radix-2 stages (log2n-6 of them) n/8                      4 reads, 4 writes, 2 twiddles
//   32/16-point stages            n/8 each                    4 reads, 4 writes
//   8-point stage                 n/16                        8 reads, 8 writes
//   bit-reverse   :7004           n/8                         4 reads, 4 writes
//   post-rotation :7094-7118      n/4                         2 reads, 2 writes
// The two "unfold" copy loops of the reference (:7120-7146) are never executed: the consumer reads
// the final sample i through an index map (final_sample), which is free.
// The decode kernel fuses windowing + overlap-add + the pcmout copy into the consumer, so HBM
// traffic is the algorithmic minimum: n/2 floats of spectrum in, bs[lW]/4+bs[W]/4 floats of PCM
// out per (block, channel).  A workgroup walks a run of consecutive blocks of one channel and
// keeps the previous block's right half in LDS; only the first block of a run is transformed
// twice (once here, once by the neighbouring run), i.e. 1/RUN extra work and no extra traffic
// except that block's spectrum.
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include <algorithm>

#include "ea1_common.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kRun = 8;  // consecutive blocks per workgroup in the decode kernel

constexpr float cPI3_8 = .38268343236508977175F;
constexpr float cPI2_8 = .70710678118654752441F;
constexpr float cPI1_8 = .92387953251128675613F;

struct MdctTables {  // device pointers
    const float *trig;  // n + n/4
    const int *bitrev;  // n/4
    const float *win;   // n/2
    int n;
    int log2n;
};

// one twiddled pair of mdct_butterfly_first/_generic (:6889-6894)
__device__ __forceinline__ void pair_op(float *x, int a, int b, float t0, float t1)
{
    const float xa0 = x[a], xa1 = x[a + 1], xb0 = x[b], xb1 = x[b + 1];
    const float r0 = xa0 - xb0;
    const float r1 = xa1 - xb1;
    x[a] = xa0 + xb0;
    x[a + 1] = xa1 + xb1;
    x[b] = r1 * t1 + r0 * t0;
    x[b + 1] = r1 * t0 - r0 * t1;
}

// first stage of mdct_butterfly_32 (:6811-6866); k selects the pair (x[16+2k..], x[2k..])
__device__ __forceinline__ void b32_pair(float *x, int k)
{
    float r0, r1;
    switch (k) {
    case 7:
        r0 = x[30] - x[14]; r1 = x[31] - x[15];
        x[30] += x[14]; x[31] += x[15];
        x[14] = r0; x[15] = r1;
        break;
    case 6:
        r0 = x[28] - x[12]; r1 = x[29] - x[13];
        x[28] += x[12]; x[29] += x[13];
        x[12] = r0 * cPI1_8 - r1 * cPI3_8;
        x[13] = r0 * cPI3_8 + r1 * cPI1_8;
        break;
    case 5:
        r0 = x[26] - x[10]; r1 = x[27] - x[11];
        x[26] += x[10]; x[27] += x[11];
        x[10] = (r0 - r1) * cPI2_8;
        x[11] = (r0 + r1) * cPI2_8;
        break;
    case 4:
        r0 = x[24] - x[8]; r1 = x[25] - x[9];
        x[24] += x[8]; x[25] += x[9];
        x[8] = r0 * cPI3_8 - r1 * cPI1_8;
        x[9] = r1 * cPI3_8 + r0 * cPI1_8;
        break;
    case 3:
        r0 = x[22] - x[6]; r1 = x[7] - x[23];
        x[22] += x[6]; x[23] += x[7];
        x[6] = r1; x[7] = r0;
        break;
    case 2:
        r0 = x[4] - x[20]; r1 = x[5] - x[21];
        x[20] += x[4]; x[21] += x[5];
        x[4] = r1 * cPI1_8 + r0 * cPI3_8;
        x[5] = r1 * cPI3_8 - r0 * cPI1_8;
        break;
    case 1:
        r0 = x[2] - x[18]; r1 = x[3] - x[19];
        x[18] += x[2]; x[19] += x[3];
        x[2] = (r1 + r0) * cPI2_8;
        x[3] = (r1 - r0) * cPI2_8;
        break;
    default:
        r0 = x[0] - x[16]; r1 = x[1] - x[17];
        x[16] += x[0]; x[17] += x[1];
        x[0] = r1 * cPI3_8 + r0 * cPI1_8;
        x[1] = r1 * cPI1_8 - r0 * cPI3_8;
        break;
    }
}

// first stage of mdct_butterfly_16 (:6776-6803)
__device__ __forceinline__ void b16_pair(float *x, int k)
{
    float r0, r1;
    switch (k) {
    case 0:
        r0 = x[1] - x[9]; r1 = x[0] - x[8];
        x[8] += x[0]; x[9] += x[1];
        x[0] = (r0 + r1) * cPI2_8;
        x[1] = (r0 - r1) * cPI2_8;
        break;
    case 1:
        r0 = x[3] - x[11]; r1 = x[10] - x[2];
        x[10] += x[2]; x[11] += x[3];
        x[2] = r0; x[3] = r1;
        break;
    case 2:
        r0 = x[12] - x[4]; r1 = x[13] - x[5];
        x[12] += x[4]; x[13] += x[5];
        x[4] = (r0 - r1) * cPI2_8;
        x[5] = (r0 + r1) * cPI2_8;
        break;
    default:
        r0 = x[14] - x[6]; r1 = x[15] - x[7];
        x[14] += x[6]; x[15] += x[7];
        x[6] = r0; x[7] = r1;
        break;
    }
}

// mdct_butterfly_8 (:6751-6772)
__device__ __forceinline__ void b8(float *x)
{
    const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4], x5 = x[5], x6 = x[6], x7 = x[7];
    float r0 = x6 + x2;
    float r1 = x6 - x2;
    float r2 = x4 + x0;
    float r3 = x4 - x0;
    x[6] = r0 + r2;
    x[4] = r0 - r2;
    r0 = x5 - x1;
    r2 = x7 - x3;
    x[0] = r1 + r0;
    x[2] = r1 - r0;
    r0 = x5 + x1;
    r1 = x7 + x3;
    x[3] = r2 + r3;
    x[1] = r2 - r3;
    x[7] = r1 + r0;
    x[5] = r1 - r0;
}

// In-LDS inverse MDCT.  On entry x[0, n/2) holds the spectrum; on exit (after the trailing
// barrier) x[n/2, n) holds the rotated values from which final_sample() reads every output.
__device__ void imdct_lds(float *x, const MdctTables &tb, int tid)
{
    const int n = tb.n, n2 = n >> 1, n4 = n >> 2;
    const float *__restrict__ T = tb.trig;

    // pre-rotation (:7061-7087): work item = (loop, iteration)
    for (int w = tid; w < n / 8; w += kThreads) {
        const int it = w & (n / 16 - 1);
        if (w < n / 16) {
            const float *iX = x + n2 - 7 - 8 * it;
            float *oX = x + n2 + n4 - 4 * (it + 1);
            const float *t = T + n4 + 4 * it;
            const float i0 = iX[0], i2 = iX[2], i4 = iX[4], i6 = iX[6];
            oX[0] = -i2 * t[3] - i0 * t[2];
            oX[1] = i0 * t[3] - i2 * t[2];
            oX[2] = -i6 * t[1] - i4 * t[0];
            oX[3] = i4 * t[1] - i6 * t[0];
        } else {
            const float *iX = x + n2 - 8 - 8 * it;
            float *oX = x + n2 + n4 + 4 * it;
            const float *t = T + n4 - 4 * (it + 1);
            const float i0 = iX[0], i2 = iX[2], i4 = iX[4], i6 = iX[6];
            oX[0] = i4 * t[3] + i6 * t[2];
            oX[1] = i4 * t[2] - i6 * t[3];
            oX[2] = i0 * t[1] + i2 * t[0];
            oX[3] = i0 * t[0] - i2 * t[1];
        }
    }
    __syncthreads();

    // mdct_butterflies(x + n2, n2) (:6974-6994)
    float *y = x + n2;
    const int nstage = tb.log2n - 6;
    for (int s = 0; s < nstage; s++) {
        const int sub = n2 >> s;  // sub-block length, sub/4 pairs in each
        const int lsub4 = tb.log2n - 1 - s - 2;
        for (int i = tid; i < n2 / 4; i += kThreads) {
            const int j = i >> lsub4;
            const int p = i & ((sub >> 2) - 1);
            const int ti = (4 << s) * p;
            pair_op(y + sub * j, sub - 2 - 2 * p, (sub >> 1) - 2 - 2 * p, T[ti], T[ti + 1]);
        }
        __syncthreads();
    }
    for (int i = tid; i < n2 / 4; i += kThreads) b32_pair(y + 32 * (i >> 3), i & 7);
    __syncthreads();
    for (int i = tid; i < n2 / 4; i += kThreads) b16_pair(y + 16 * (i >> 2), i & 3);
    __syncthreads();
    for (int i = tid; i < n2 / 8; i += kThreads) b8(y + 8 * i);
    __syncthreads();

    // mdct__bitreverse (:7004-7052): reads y, writes x[0, n2)
    {
        const float *t = T + n;
        const int *__restrict__ bit = tb.bitrev;
        for (int q = tid; q < n / 8; q += kThreads) {
            const float *x0 = y + bit[2 * q];
            const float *x1 = y + bit[2 * q + 1];
            const float a0 = x0[0], a1 = x0[1], b0 = x1[0], b1 = x1[1];
            float r0 = a1 - b1;
            float r1 = a0 + b0;
            const float r2 = r1 * t[2 * q] + r0 * t[2 * q + 1];
            const float r3 = r1 * t[2 * q + 1] - r0 * t[2 * q];
            r0 = (a1 + b1) * .5f;
            r1 = (a0 - b0) * .5f;
            const int lo = 2 * q, hi = n2 - 2 * q - 2;
            x[lo] = r0 + r2;
            x[lo + 1] = r1 + r3;
            x[hi] = r0 - r2;
            x[hi + 1] = r3 - r1;
        }
    }
    __syncthreads();

    // post-rotation (:7100-7118): reads x[0, n2), writes x[n2, n)
    {
        const float *t = T + n2;
        for (int m = tid; m < n4; m += kThreads) {
            const float a = x[2 * m], b = x[2 * m + 1];
            const float t0 = t[2 * m], t1 = t[2 * m + 1];
            x[n2 + n4 - 1 - m] = a * t1 - b * t0;
            x[n2 + n4 + m] = -(a * t0 + b * t1);
        }
    }
    __syncthreads();
}

// Sample i of the finished transform, i.e. what the reference's out[i] holds after :7120-7146.
__device__ __forceinline__ float final_sample(const float *x, int n, int i)
{
    const int n2 = n >> 1, n4 = n >> 2;
    if (i < n4) return x[n2 + i];
    if (i < n2) return -x[n - 1 - i];
    if (i < n2 + n4) return x[n + n2 - 1 - i];
    return x[i];
}

__global__ void __launch_bounds__(kThreads)
mdct_backward_kernel(MdctTables tb, const float *__restrict__ in, size_t in_stride,
                     float *__restrict__ out, size_t out_stride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int n = tb.n;
    const float *src = in + size_t(blockIdx.x) * in_stride;
    for (int i = tid; i < n / 2; i += kThreads) smem[i] = src[i];
    __syncthreads();
    imdct_lds(smem, tb, tid);
    float *dst = out + size_t(blockIdx.x) * out_stride;
    for (int i = tid; i < n; i += kThreads) dst[i] = final_sample(smem, n, i);
}

struct BlockMeta {     // one per block of the batch, built on the host
    int64_t out_off;   // sample offset of this block's output in each channel row
    int count;         // samples produced (0 for a first block)
    uint8_t W;         // this block long?
    uint8_t lW;        // previous block long?
    uint8_t has_prev;  // previous right half is meaningful
    uint8_t pad;
};

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

// workgroup (run, ch) of one stream: grid.x = ceil(nblocks / kRun), grid.y = channels
__device__ __forceinline__ void synth_decode_run(const MdctTables &tb0, const MdctTables &tb1, const float *__restrict__ spectra,
                                                 const BlockMeta *__restrict__ meta, int nblocks, int channels,
                                                 float *__restrict__ pcm, size_t pcm_stride, const float *__restrict__ tail_in,
                                                 float *__restrict__ tail_out, float *smem, int run, int ch)
{
    const int bs1 = tb1.n, bs0 = tb0.n;
    float *x = smem;            // bs1 floats: the transform in flight
    float *prevR = smem + bs1;  // bs1/2 floats: right half of the previous block (raw)
    const int tid = threadIdx.x;
    const int first = run * kRun;
    const int last = min(first + kRun, nblocks);
    const size_t row = size_t(bs1 / 2);
    float *out = pcm + size_t(ch) * pcm_stride;
    // the carried right half is double-buffered: the first run reads last call's copy while the
    // last run of THIS launch writes the other one (same buffer would be a read/write race)
    const float *tl_in = tail_in + size_t(ch) * row;
    float *tl_out = tail_out + size_t(ch) * row;

    // right half of block first-1: carried state for the first run, recomputed otherwise
    if (first == 0) {
        for (int i = tid; i < bs1 / 2; i += kThreads) prevR[i] = tl_in[i];
        __syncthreads();
    } else {
        const int k = first - 1;
        const MdctTables &tb = meta[k].W ? tb1 : tb0;
        const float *src = spectra + (size_t(k) * channels + ch) * row;
        for (int i = tid; i < tb.n / 2; i += kThreads) x[i] = src[i];
        __syncthreads();
        imdct_lds(x, tb, tid);
        for (int i = tid; i < tb.n / 2; i += kThreads) prevR[i] = final_sample(x, tb.n, tb.n / 2 + i);
        __syncthreads();
    }

    for (int k = first; k < last; k++) {
        const BlockMeta m = meta[k];
        const MdctTables &tb = m.W ? tb1 : tb0;
        const int n = tb.n;
        const float *src = spectra + (size_t(k) * channels + ch) * row;
        for (int i = tid; i < n / 2; i += kThreads) x[i] = src[i];
        __syncthreads();
        imdct_lds(x, tb, tid);

        // windowed overlap-add + pcmout (:9243-9288, :9299-9306), n0/n1 = half block sizes
        if (m.count > 0) {
            const int n0 = bs0 / 2, n1 = bs1 / 2;
            float *o = out + m.out_off;
            if (m.lW && m.W) {  // long/long
                const float *w = tb1.win;
                for (int i = tid; i < n1; i += kThreads)
                    o[i] = prevR[i] * w[n1 - i - 1] + final_sample(x, n, i) * w[i];
            } else if (m.lW) {  // long/short: head of the long right half passes through
                const float *w = tb0.win;
                const int lead = n1 / 2 - n0 / 2;
                for (int s = tid; s < lead + n0; s += kThreads) {
                    const int i = s - lead;
                    o[s] = i < 0 ? prevR[s] : prevR[s] * w[n0 - i - 1] + final_sample(x, n, i) * w[i];
                }
            } else if (m.W) {   // short/long: long left half starts at n1/2 - n0/2
                const float *w = tb0.win;
                const int skip = n1 / 2 - n0 / 2;
                for (int s = tid; s < n1 / 2 + n0 / 2; s += kThreads) {
                    const float p = final_sample(x, n, skip + s);
                    o[s] = s < n0 ? prevR[s] * w[n0 - s - 1] + p * w[s] : p;
                }
            } else {            // short/short
                const float *w = tb0.win;
                for (int i = tid; i < n0; i += kThreads)
                    o[i] = prevR[i] * w[n0 - i - 1] + final_sample(x, n, i) * w[i];
            }
        }
        __syncthreads();
        // the copy section (:9282-9287): keep this block's right half for the next one
        for (int i = tid; i < n / 2; i += kThreads) prevR[i] = final_sample(x, n, n / 2 + i);
        __syncthreads();
    }
    if (last == nblocks)
        for (int i = tid; i < bs1 / 2; i += kThreads) tl_out[i] = prevR[i];
}

// grid = (ceil(nblocks / kRun), channels)
__global__ void __launch_bounds__(kThreads)
synth_decode_kernel(MdctTables tb0, MdctTables tb1, const float *__restrict__ spectra,
                    const BlockMeta *__restrict__ meta, int nblocks, int channels,
                    float *__restrict__ pcm, size_t pcm_stride, const float *__restrict__ tail_in,
                    float *__restrict__ tail_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    synth_decode_run(tb0, tb1, spectra, meta, nblocks, channels, pcm, pcm_stride, tail_in, tail_out, smem, blockIdx.x, blockIdx.y);
}

// several streams in one launch: grid = (max runs, max channels, streams); a stream with fewer runs / channels
// leaves the surplus workgroups idle (they return at once)
__global__ void __launch_bounds__(kThreads)
synth_decode_batch_kernel(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const StreamArgs &s = a.s[blockIdx.z];
    if (int(blockIdx.x) * kRun >= s.nblocks || int(blockIdx.y) >= s.channels) return;   // uniform per workgroup
    synth_decode_run(s.tb0, s.tb1, s.spectra, s.meta, s.nblocks, s.channels, s.pcm, s.pcm_stride, s.tail_in, s.tail_out, smem, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------- host side
struct HostTables {
    std::vector<float> trig, win;
    std::vector<int> bitrev;
};

// mdct_init :6709 + window literal reconstruction (see oracle/vorbis_synth.c:ea1o_window for why
// the "%.10f" round trip reproduces the reference's tables bit-for-bit).
void build_tables(int n, HostTables &h)
{
    const int n2 = n >> 1, log2n = ea1::ilog2(n);
    h.trig.resize(n + n / 4);
    h.bitrev.resize(n / 4);
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
    const int mask = (1 << (log2n - 1)) - 1, msb = 1 << (log2n - 2);
    for (int i = 0; i < n / 8; i++) {
        int acc = 0;
        for (int j = 0; msb >> j; j++)
            if ((msb >> j) & i) acc |= 1 << j;
        h.bitrev[i * 2] = ((~acc) & mask) - 1;
        h.bitrev[i * 2 + 1] = acc;
    }
    char buf[64];
    for (int i = 0; i < n / 2; i++) {
        const double s = sin((i + 0.5) / n * M_PI);
        snprintf(buf, sizeof buf, "%.10f", sin(0.5 * M_PI * s * s));
        h.win[i] = float(strtod(buf, nullptr));
    }
}

struct DeviceTables {
    float *trig = nullptr, *win = nullptr;
    int *bitrev = nullptr;
    int n = 0;
    int upload(int n_)
    {
        HostTables h;
        build_tables(n_, h);
        n = n_;
        EA1_HIP(hipMalloc(&trig, h.trig.size() * sizeof(float)));
        EA1_HIP(hipMalloc(&win, h.win.size() * sizeof(float)));
        EA1_HIP(hipMalloc(&bitrev, h.bitrev.size() * sizeof(int)));
        EA1_HIP(hipMemcpy(trig, h.trig.data(), h.trig.size() * sizeof(float), hipMemcpyHostToDevice));
        EA1_HIP(hipMemcpy(win, h.win.data(), h.win.size() * sizeof(float), hipMemcpyHostToDevice));
        EA1_HIP(hipMemcpy(bitrev, h.bitrev.data(), h.bitrev.size() * sizeof(int), hipMemcpyHostToDevice));
        return EA1_OK;
    }
    void release()
    {
        if (trig) (void)hipFree(trig);
        if (win) (void)hipFree(win);
        if (bitrev) (void)hipFree(bitrev);
        trig = win = nullptr;
        bitrev = nullptr;
    }
    MdctTables view() const { return MdctTables{trig, bitrev, win, n, ea1::ilog2(n)}; }
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

// host part of a decode call: block plan -> pinned staging -> device, ordered on `st`; fills `a`
int prepare_decode(ea1_vorbis_synth *s, const float *d_spectra, const uint8_t *flags, int nblocks, float *d_pcm, size_t pcm_stride,
                   int *counts, int64_t *total, hipStream_t st, StreamArgs *a)
{
    if (nblocks > s->meta_cap) {
        EA1_HIP(hipDeviceSynchronize());
        if (s->d_meta) (void)hipFree(s->d_meta);
        if (s->h_meta) (void)hipHostFree(s->h_meta);
        s->d_meta = nullptr;
        s->h_meta = nullptr;
        s->meta_cap = 0;
        const int cap = nblocks + nblocks / 2 + 64;
        EA1_HIP(hipMalloc(&s->d_meta, size_t(cap) * sizeof(BlockMeta)));
        EA1_HIP(hipHostMalloc(&s->h_meta, size_t(cap) * sizeof(BlockMeta)));
        s->meta_cap = cap;
    } else {
        EA1_HIP(hipEventSynchronize(s->meta_free));  // previous upload has left the staging buffer
    }
    const int64_t tot = plan(s, flags, nblocks, s->h_meta, counts);
    if (size_t(tot) > pcm_stride)
        return ea1::fail(EA1_ECAPACITY, "ea1_vorbis_synth_decode: %lld samples > pcm_stride %zu",
                         (long long)tot, pcm_stride);
    EA1_HIP(hipMemcpyAsync(s->d_meta, s->h_meta, size_t(nblocks) * sizeof(BlockMeta), hipMemcpyHostToDevice, st));
    EA1_HIP(hipEventRecord(s->meta_free, st));
    a->tb0 = s->tb[0].view();
    a->tb1 = s->tb[1].view();
    a->spectra = d_spectra;
    a->meta = s->d_meta;
    a->pcm = d_pcm;
    a->pcm_stride = pcm_stride;
    a->tail_in = s->d_tail + size_t(s->tail_cur) * s->channels * (s->bs[1] / 2);
    a->tail_out = s->d_tail + size_t(s->tail_cur ^ 1) * s->channels * (s->bs[1] / 2);
    a->nblocks = nblocks;
    a->channels = s->channels;
    if (total) *total = tot;
    return EA1_OK;
}

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
    const int rc = prepare_decode(s, d_spectra, flags, nblocks, d_pcm, pcm_stride, counts, total, st, &a);
    if (rc != EA1_OK) return rc;
    const dim3 grid((nblocks + kRun - 1) / kRun, s->channels);
    const size_t lds = size_t(s->bs[1] + s->bs[1] / 2) * sizeof(float);
    hipLaunchKernelGGL(synth_decode_kernel, grid, dim3(kThreads), lds, st, a.tb0, a.tb1, a.spectra, a.meta, a.nblocks, a.channels, a.pcm,
                       a.pcm_stride, a.tail_in, a.tail_out);
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
        size_t lds = 0;
        int idx[kBatchStreams];
        for (int k = base; k < n && k < base + kBatchStreams; k++) {
            if (!nblocks[k]) continue;
            const int rc = prepare_decode(synths[k], d_spectra[k], flags[k], nblocks[k], d_pcm[k], pcm_stride[k], nullptr, totals ? &totals[k] : nullptr, st, &ba.s[live]);
            if (rc != EA1_OK) return rc;   // streams prepared before this one have not been committed: their state is unchanged
            max_runs = std::max(max_runs, (nblocks[k] + kRun - 1) / kRun);
            max_ch = std::max(max_ch, synths[k]->channels);
            lds = std::max(lds, size_t(synths[k]->bs[1] + synths[k]->bs[1] / 2) * sizeof(float));
            idx[live++] = k;
        }
        if (!live) continue;
        hipLaunchKernelGGL(synth_decode_batch_kernel, dim3(max_runs, max_ch, live), dim3(kThreads), lds, st, ba);
        EA1_LAUNCH_CHECK();
        for (int i = 0; i < live; i++) commit_decode(synths[idx[i]], flags[idx[i]], nblocks[idx[i]]);
    }
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
    hipLaunchKernelGGL(mdct_backward_kernel, dim3(count), dim3(kThreads), size_t(n) * sizeof(float), st,
                       tb.view(), d_in, in_stride, d_out, out_stride);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // tables are freed below
    tb.release();
    if (e != hipSuccess) return ea1::fail(EA1_EHIP, "mdct_backward_kernel: %s", hipGetErrorString(e));
    return EA1_OK;
}
