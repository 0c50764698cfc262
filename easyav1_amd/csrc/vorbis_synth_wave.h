// Vorbis synthesis, one 64-lane wave per transform: the lane program of vorbis_synth.hip's kernels.
//
// Everything here is a restatement of libvorbis 1.3.7's scalar routines (the reference's ext/minivorbis/minivorbis.h; line
// numbers below are that file's) as a schedule of INDEPENDENT operations: each rounded multiply, add and subtract of the
// reference is performed once, on the same operands, in the same association -- only who performs it and when changes,
// which is why the results are bit-identical.  Must be compiled with -ffp-contract=off (an FMA changes the last bit).
//
//   mdct_backward :7054-7148 = pre-rotation :7061-7087, mdct_butterflies :6974-6994 (mdct_butterfly_first / _generic
//   :6874-6972, mdct_butterfly_32 / _16 / _8 :6751-6871), mdct_bitreverse :7004-7052, post-rotation :7094-7118; the window
//   and overlap-add of vorbis_synthesis_blockin :9243-9306.
//
// The butterflies are a decimation-in-frequency radix-2 transform of C = n/4 complex points: log2(C) - 4 twiddled stages at
// distances C/2 ... 16, then the fixed 32-float butterfly (distances 8, 4, 2, 1).  The schedule:
//
//   load     one float4 of the spectrum per work item (coalesced, read once) -> BOTH pre-rotated elements made of it -> LDS
//   passes   up to three twiddled stages at a time on 8 / 4 / 2 elements held in registers (radix 8 / 4 / 2), in place in LDS
//   final    the 32-float butterfly of 16 consecutive elements in registers, stored to LDS in BIT-REVERSED order
//   combine  the bit-reverse step reads its two operands at q and C-1-q (contiguous), applies both post-rotations and hands
//            the four samples z[q], z[C-1-q], z[C+q], z[2C-1-q] to the consumer: window + overlap-add + PCM store, or the
//            plain unfolded output.  The two "unfold" copy loops of the reference (:7120-7146) are index arithmetic here.
//
// LDS holds C complex elements in a skewed layout (two elements of padding after every 16, so that the final pass's 128-bit
// reads of 16 consecutive elements per lane and every pass's 64-bit reads of runs of >= 16 consecutive elements are free of
// bank conflicts) and the right half of the previous block (C floats: the half is mirror-symmetric).  A transform of 2048
// moves 36 KB through LDS in 5 round trips; the block-wide schedule this replaces moved 100 KB in 12, with 32-way conflicts
// in its bit-reverse step and a block barrier per stage (profiles/r04_n_audio_counters.txt).
//
// Lanes of a wave synchronise through `Sync` (a wave-level fence on the device).  The same code runs on the host with 64
// threads and a barrier: tests/native/vorbis_wave_emul.cpp, checked against the oracle without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define EA1_VS_HD __host__ __device__ __forceinline__
#else
#define EA1_VS_HD inline
#endif

namespace ea1 {
namespace vsynth {

constexpr int kWave = 64;
constexpr int kRun = 8;  // consecutive blocks of one channel per wave

constexpr float cPI3_8 = .38268343236508977175F;
constexpr float cPI2_8 = .70710678118654752441F;
constexpr float cPI1_8 = .92387953251128675613F;

struct alignas(8) F2 {
    float x, y;
};
struct alignas(16) F4 {
    float x, y, z, w;
};

struct MdctTables {     // device pointers (host pointers under emulation)
    const float *trig;  // n + n/4 floats, mdct_init :6709
    const float *win;   // n/2
    int n;
    int log2n;
};

struct BlockMeta {     // one per block of a decode call, built on the host
    int64_t out_off;   // sample offset of this block's output in each channel row
    int count;         // samples produced (0 for a first block)
    uint8_t W;         // this block long?
    uint8_t lW;        // previous block long?
    uint8_t has_prev;  // previous right half is meaningful
    uint8_t pad;
};

// LDS floats a wave needs for transforms up to `n` (n * 9/16) and for two copies of a right half (n/4 each)
constexpr int lds_transform_floats(int n) { return (n / 4 + n / 32) * 2; }
constexpr int lds_decode_floats(int bs1) { return lds_transform_floats(bs1) + 2 * (bs1 / 4); }

// complex element c lives at float offset 2 * skew(c)
EA1_VS_HD int skew(int c) { return c + ((c >> 4) << 1); }

EA1_VS_HD uint32_t bit_reverse(uint32_t v, int bits)
{
    if (bits <= 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - bits);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    v = (v >> 16) | (v << 16);
    return v >> (32 - bits);
#endif
}

// ---------------------------------------------------------------- the fixed 32-float butterfly, on registers
// first stage of mdct_butterfly_32 (:6811-6866); k selects the pair (x[16+2k..], x[2k..])
EA1_VS_HD void b32_pair(float *x, int k)
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
EA1_VS_HD void b16_pair(float *x, int k)
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
EA1_VS_HD void b8(float *x)
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

// mdct_butterfly_32 (:6805-6871) on 32 floats in registers (every index below is a constant after unrolling)
EA1_VS_HD void butterfly_32(float *v)
{
#pragma unroll
    for (int k = 0; k < 8; k++) b32_pair(v, k);
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 4; k++) b16_pair(v + 16 * h, k);
    }
#pragma unroll
    for (int h = 0; h < 4; h++) b8(v + 8 * h);
}

// ---------------------------------------------------------------- the stages of one transform
// load + pre-rotation (:7061-7087).  The float4 x[n2-4-4w .. n2-1-4w] = (R, Q, S, P) makes element C/2-1-w of the first loop
// (from the odd entries Q, P) and element C/2+w of the second (from the even entries R, S).
EA1_VS_HD void load_prerotate(float *y, const MdctTables &tb, const float *src, bool aligned, int lane)
{
    const int n2 = tb.n >> 1, C = tb.n >> 2, half = C >> 1;
    const float *T = tb.trig;
    for (int w = lane; w < half; w += kWave) {
        F4 s;
        const float *p = src + n2 - 4 - 4 * w;
        if (aligned) {
            s = *reinterpret_cast<const F4 *>(p);
        } else {
            s.x = p[0]; s.y = p[1]; s.z = p[2]; s.w = p[3];
        }
        const F2 ta = *reinterpret_cast<const F2 *>(T + C + 2 * w);       // first loop's t[2], t[3] (or t[0], t[1])
        const F2 tc = *reinterpret_cast<const F2 *>(T + C - 2 * w - 2);   // second loop's t[2], t[3] (or t[0], t[1])
        F2 lo, hi;
        lo.x = -s.w * ta.y - s.y * ta.x;
        lo.y = s.y * ta.y - s.w * ta.x;
        hi.x = s.x * tc.y + s.z * tc.x;
        hi.y = s.x * tc.x - s.z * tc.y;
        *reinterpret_cast<F2 *>(y + 2 * skew(half - 1 - w)) = lo;
        *reinterpret_cast<F2 *>(y + 2 * skew(half + w)) = hi;
    }
}

// K twiddled stages (mdct_butterfly_first / _generic, :6874-6972) starting at stage s, on 2^K elements per work item
template <int K>
EA1_VS_HD void twiddled_pass(float *y, const float *T, int log2C, int s, int lane)
{
    constexpr int R = 1 << K;
    const int log2D = log2C - s - K;  // D = smallest distance of the pass, >= 16
    const int D = 1 << log2D;
    const int groups = 1 << (log2C - K);
    for (int grp = lane; grp < groups; grp += kWave) {
        const int lo = grp & (D - 1);
        const int base = ((grp >> log2D) << (log2D + K)) + lo;
        float re[R], im[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const F2 v = *reinterpret_cast<const F2 *>(y + 2 * skew(base + m * D));
            re[m] = v.x;
            im[m] = v.y;
        }
#pragma unroll
        for (int t = 0; t < K; t++) {
            const int span = 1 << (K - 1 - t);
            const int dist = D * span;
            const int step = 4 << (s + t);
#pragma unroll
            for (int m = 0; m < R; m++) {
                if (m & span) continue;
                const int u = m + span;
                const int j = lo + (m & (span - 1)) * D;  // position of the lower element in its half
                const F2 tw = *reinterpret_cast<const F2 *>(T + step * (dist - 1 - j));
                const float r0 = re[u] - re[m];
                const float r1 = im[u] - im[m];
                re[u] = re[u] + re[m];
                im[u] = im[u] + im[m];
                re[m] = r1 * tw.y + r0 * tw.x;
                im[m] = r1 * tw.x - r0 * tw.y;
            }
        }
#pragma unroll
        for (int m = 0; m < R; m++) {
            F2 v;
            v.x = re[m];
            v.y = im[m];
            *reinterpret_cast<F2 *>(y + 2 * skew(base + m * D)) = v;
        }
    }
}

// the 32-float butterflies; element e leaves at position bitrev(e) of the UNSKEWED array that overlays y
template <int ITER, class Sync>
EA1_VS_HD void final_pass(float *y, int log2C, int lane, const Sync &sync)
{
    const int groups = 1 << (log2C - 4);
    float v[ITER][32];
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int g = lane + kWave * it;
        if (g < groups) {
            const F4 *p = reinterpret_cast<const F4 *>(y + 2 * skew(16 * g));
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const F4 q = p[j];
                v[it][4 * j] = q.x; v[it][4 * j + 1] = q.y; v[it][4 * j + 2] = q.z; v[it][4 * j + 3] = q.w;
            }
        }
    }
    sync();  // every element is in a register: the overlay may be written
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int g = lane + kWave * it;
        if (g < groups) {
            butterfly_32(v[it]);
            const int rg = int(bit_reverse(uint32_t(g), log2C - 4));
#pragma unroll
            for (int m = 0; m < 16; m++) {
                constexpr int rev4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
                F2 o;
                o.x = v[it][2 * m];
                o.y = v[it][2 * m + 1];
                *reinterpret_cast<F2 *>(y + 2 * (rev4[m] * groups + rg)) = o;
            }
        }
    }
}

// spectrum in global memory -> bit-reversed butterfly output in LDS (y[0, n/2) unskewed); ends with a sync
template <class Sync>
EA1_VS_HD void transform_to_lds(float *y, const MdctTables &tb, const float *src, bool aligned, int lane, const Sync &sync)
{
    const int log2C = tb.log2n - 2;
    load_prerotate(y, tb, src, aligned, lane);
    sync();
    int s = 0;
    const int stages = log2C - 4;
    while (s < stages) {
        const int rem = stages - s;
        if (rem == 1) {
            twiddled_pass<1>(y, tb.trig, log2C, s, lane);
            s += 1;
        } else if (rem == 2 || rem == 4) {
            twiddled_pass<2>(y, tb.trig, log2C, s, lane);
            s += 2;
        } else {
            twiddled_pass<3>(y, tb.trig, log2C, s, lane);
            s += 3;
        }
        sync();
    }
    if ((1 << (log2C - 4)) > kWave)
        final_pass<2>(y, log2C, lane, sync);  // n = 8192: 128 butterflies
    else
        final_pass<1>(y, log2C, lane, sync);
    sync();
}

// mdct_bitreverse (:7004-7052) for step q and both post-rotations (:7100-7118) of its two results
struct Quad {
    float zl0, zl1;  // z[q], z[C-1-q]: the left half is z[i] for i < C and -z[2C-1-i] above
    float zr0, zr1;  // z[C+q], z[2C-1-q]: the right half is z[2C-1-i] for i < C and z[i] above
};
EA1_VS_HD Quad combine(const float *P, const MdctTables &tb, int q)
{
    const int n = tb.n, n2 = n >> 1, C = n >> 2;
    const float *T = tb.trig;
    const F2 b = *reinterpret_cast<const F2 *>(P + 2 * q);            // x1 = butterfly output bitrev(q)
    const F2 a = *reinterpret_cast<const F2 *>(P + 2 * (C - 1 - q));  // x0 = butterfly output C-1-bitrev(q)
    const F2 tq = *reinterpret_cast<const F2 *>(T + n + 2 * q);
    float r0 = a.y - b.y;
    float r1 = a.x + b.x;
    const float r2 = r1 * tq.x + r0 * tq.y;
    const float r3 = r1 * tq.y - r0 * tq.x;
    r0 = (a.y + b.y) * .5f;
    r1 = (a.x - b.x) * .5f;
    const float wq0 = r0 + r2, wq1 = r1 + r3;  // w[q]
    const float wp0 = r0 - r2, wp1 = r3 - r1;  // w[C-1-q]
    const F2 uq = *reinterpret_cast<const F2 *>(T + n2 + 2 * q);
    const F2 up = *reinterpret_cast<const F2 *>(T + n2 + 2 * (C - 1 - q));
    Quad z;
    z.zl1 = wq0 * uq.y - wq1 * uq.x;       // m = q:     z[C-1-q]
    z.zr0 = -(wq0 * uq.x + wq1 * uq.y);    //            z[C+q]
    z.zl0 = wp0 * up.y - wp1 * up.x;       // m = C-1-q: z[q]
    z.zr1 = -(wp0 * up.x + wp1 * up.y);    //            z[2C-1-q]
    return z;
}

// ---------------------------------------------------------------- mdct_backward alone: n/2 floats in, n floats out
template <class Sync>
EA1_VS_HD void wave_mdct_backward(float *lds, const MdctTables &tb, const float *src, bool aligned, float *dst, int lane, const Sync &sync)
{
    transform_to_lds(lds, tb, src, aligned, lane, sync);
    const int n = tb.n, n2 = n >> 1, C = n >> 2;
    for (int q = lane; q < (C >> 1); q += kWave) {
        const Quad z = combine(lds, tb, q);
        // out[i] = z[i], out[n2-1-i] = -z[i] (i < C); out[n-1-j] = out[n2+j] = z[j] (j >= C)     (:7120-7146)
        dst[q] = z.zl0;
        dst[n2 - 1 - q] = -z.zl0;
        dst[C - 1 - q] = z.zl1;
        dst[C + q] = -z.zl1;
        dst[n - 1 - (C + q)] = z.zr0;
        dst[n2 + C + q] = z.zr0;
        dst[n - 2 * C + q] = z.zr1;  // n - 1 - (2C-1-q)
        dst[n2 + 2 * C - 1 - q] = z.zr1;
    }
    sync();
}

// ---------------------------------------------------------------- decode: transform + window + overlap-add + PCM
// sample i of a right half kept as its upper half R[0, Cp): the reference's pcm[...] tail after :9282-9287
EA1_VS_HD float right_half(const float *R, int Cp, int i) { return i < Cp ? R[Cp - 1 - i] : R[i - Cp]; }

struct Emit {      // where sample i of this block's left half goes and what it overlaps (vorbis_synthesis_blockin :9243-9288)
    float *o;          // this block's output
    const float *Rp;   // previous right half (upper half), of Cp floats
    const float *win;  // window of the overlap
    int Cp;
    int shift;         // output index = i + shift
    int wh;            // overlap length (window half size)
    int wmode;         // 0: window index = i, every sample overlaps; 1: window index = output index, overlap below wh
    EA1_VS_HD void operator()(int i, float f) const
    {
        const int s = i + shift;
        if (s < 0) return;
        if (wmode == 0) {
            o[s] = right_half(Rp, Cp, s) * win[wh - i - 1] + f * win[i];
        } else if (s < wh) {
            o[s] = right_half(Rp, Cp, s) * win[wh - s - 1] + f * win[s];
        } else {
            o[s] = f;
        }
    }
};

// one (run, channel) of one stream: blocks [run * kRun, ...) of channel ch.  lds: lds_decode_floats(bs1) floats.
template <class Sync>
EA1_VS_HD void wave_decode_run(const MdctTables &tb0, const MdctTables &tb1, const float *spectra, const BlockMeta *meta, int nblocks,
                               int channels, float *pcm, size_t pcm_stride, const float *tail_in, float *tail_out, float *lds, int run,
                               int ch, int lane, const Sync &sync)
{
    const int bs1 = tb1.n, bs0 = tb0.n;
    float *y = lds;
    float *Rbuf[2] = {lds + lds_transform_floats(bs1), lds + lds_transform_floats(bs1) + bs1 / 4};
    const int first = run * kRun;
    const int last = first + kRun < nblocks ? first + kRun : nblocks;
    const size_t row = size_t(bs1 / 2);
    const bool aligned = (reinterpret_cast<uintptr_t>(spectra) & 15) == 0;
    float *out = pcm + size_t(ch) * pcm_stride;
    // the carried right half is double-buffered in global memory: the first run reads last call's copy while the last run of
    // THIS launch writes the other one
    const float *tl_in = tail_in + size_t(ch) * row;
    float *tl_out = tail_out + size_t(ch) * row;
    int cur = 0;  // Rbuf[cur] holds the right half of the previous block

    if (first == 0) {
        const int Cp = (meta[0].lW ? bs1 : bs0) >> 2;
        for (int i = lane; i < Cp; i += kWave) Rbuf[0][i] = tl_in[Cp + i];
        sync();
    } else {  // recomputed: block first-1 belongs to the neighbouring run
        const int k = first - 1;
        const MdctTables &tb = meta[k].W ? tb1 : tb0;
        transform_to_lds(y, tb, spectra + (size_t(k) * channels + ch) * row, aligned, lane, sync);
        const int C = tb.n >> 2;
        for (int q = lane; q < (C >> 1); q += kWave) {
            const Quad z = combine(y, tb, q);
            Rbuf[0][q] = z.zr0;
            Rbuf[0][C - 1 - q] = z.zr1;
        }
        sync();
    }

    for (int k = first; k < last; k++) {
        const BlockMeta m = meta[k];
        const MdctTables &tb = m.W ? tb1 : tb0;
        transform_to_lds(y, tb, spectra + (size_t(k) * channels + ch) * row, aligned, lane, sync);
        const int C = tb.n >> 2;
        const int n0 = bs0 / 2, n1 = bs1 / 2;
        const float *Rp = Rbuf[cur];
        float *Rn = Rbuf[cur ^ 1];
        Emit e;
        e.o = out + m.out_off;
        e.Rp = Rp;
        e.Cp = (m.lW ? bs1 : bs0) >> 2;
        if (m.lW == m.W) {  // long/long, short/short
            e.win = tb.win; e.shift = 0; e.wh = tb.n >> 1; e.wmode = 0;
        } else if (m.lW) {  // long/short: the head of the long right half passes through
            const int lead = n1 / 2 - n0 / 2;
            e.win = tb0.win; e.shift = lead; e.wh = n0; e.wmode = 0;
            if (m.count > 0)
                for (int s = lane; s < lead; s += kWave) e.o[s] = right_half(Rp, e.Cp, s);
        } else {            // short/long: the long left half starts at n1/2 - n0/2
            e.win = tb0.win; e.shift = -(n1 / 2 - n0 / 2); e.wh = n0; e.wmode = 1;
        }
        for (int q = lane; q < (C >> 1); q += kWave) {
            const Quad z = combine(y, tb, q);
            if (m.count > 0) {
                e(q, z.zl0);
                e(C - 1 - q, z.zl1);
                e(C + q, -z.zl1);
                e(2 * C - 1 - q, -z.zl0);
            }
            Rn[q] = z.zr0;
            Rn[C - 1 - q] = z.zr1;
        }
        cur ^= 1;
        sync();
    }
    if (last == nblocks) {
        const int n2 = (meta[nblocks - 1].W ? bs1 : bs0) >> 1;
        for (int i = lane; i < n2; i += kWave) tl_out[i] = right_half(Rbuf[cur], n2 >> 1, i);
    }
}

}  // namespace vsynth
}  // namespace ea1
