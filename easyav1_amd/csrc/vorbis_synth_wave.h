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
// The trigonometric tables and windows are read from wherever MdctTables points: the kernels copy them to LDS once per
// workgroup of several waves (every stage of every transform would otherwise wait for an L2 round trip: the first form of this
// schedule spent 20 us per transform that way: profiles/r04_p_audio_counters.txt, r04_r_audio_stage_probe.txt), the butterflies' twiddles as compact
// per-stage arrays so that consecutive lanes read consecutive entries.
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
constexpr int kRun = 16;  // consecutive blocks of one channel per wave

constexpr float cPI3_8 = .38268343236508977175F;
constexpr float cPI2_8 = .70710678118654752441F;
constexpr float cPI1_8 = .92387953251128675613F;

struct alignas(8) F2 {
    float x, y;
};
struct alignas(16) F4 {
    float x, y, z, w;
};

struct MdctTables {     // device pointers -- global memory, or the workgroup's LDS copy (host pointers under emulation)
    const float *trig;  // n + n/4 floats, mdct_init :6709
    const float *tw;    // the butterflies' twiddles stage by stage, compact: stage s holds (T[(4<<s)*p], T[(4<<s)*p+1]) for p < C >> (s+1)
    const float *win;   // n/2 (may be null for a bare transform)
    int n;
    int log2n;
};

// floats of one size's tables, back to back: trig | tw | win
constexpr int trig_floats(int n) { return n + n / 4; }
constexpr int tw_floats(int n) { return n / 2 - 32; }   // 2 * (C/2 + C/4 + ... + 16)
constexpr int table_floats(int n) { return trig_floats(n) + tw_floats(n) + n / 2; }

// the compact stage tables out of mdct_init's T (the values themselves are not touched)
inline void build_stage_twiddles(const float *T, int n, float *tw)
{
    const int C = n >> 2;
    int log2C = 0;
    while ((1 << log2C) < C) log2C++;
    for (int s = 0; s < log2C - 4; s++) {
        const int dist = C >> (s + 1);
        float *o = tw + 2 * (C - (C >> s));
        for (int p = 0; p < dist; p++) {
            o[2 * p] = T[(4 << s) * p];
            o[2 * p + 1] = T[(4 << s) * p + 1];
        }
    }
}

// one size's tables copied to `dst` (LDS) by `nthreads` threads; returns the view of the copy.  The caller synchronises.
EA1_VS_HD MdctTables copy_tables(float *dst, const MdctTables &g, int tid, int nthreads)
{
    const int n = g.n;
    const int a = trig_floats(n), b = tw_floats(n), c = g.win ? n / 2 : 0;
    for (int i = tid; i < a; i += nthreads) dst[i] = g.trig[i];
    for (int i = tid; i < b; i += nthreads) dst[a + i] = g.tw[i];
    for (int i = tid; i < c; i += nthreads) dst[a + b + i] = g.win[i];
    MdctTables l = g;
    l.trig = dst;
    l.tw = dst + a;
    l.win = g.win ? dst + a + b : nullptr;
    return l;
}

struct BlockMeta {     // one per block of a decode call, built on the host
    int64_t out_off;   // sample offset of this block's output in each channel row
    int count;         // samples produced (0 for a first block)
    uint8_t W;         // this block long?
    uint8_t lW;        // previous block long?
    uint8_t has_prev;  // previous right half is meaningful
    uint8_t pad;
};

// a value every lane of the wave holds alike, said so to the compiler (scalar registers, uniform branches)
EA1_VS_HD int uniform(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
EA1_VS_HD BlockMeta load_meta(const BlockMeta *p)
{
    BlockMeta m = *p;
    const uint64_t off = uint64_t(m.out_off);
    m.out_off = int64_t(uint64_t(uint32_t(uniform(int(uint32_t(off))))) | (uint64_t(uint32_t(uniform(int(uint32_t(off >> 32))))) << 32));
    m.count = uniform(m.count);
    m.W = uint8_t(uniform(m.W));
    m.lW = uint8_t(uniform(m.lW));
    return m;
}

// LDS floats a wave needs for transforms up to `n` (n * 9/16) and for the upper half of a right half (n/4)
constexpr int lds_transform_floats(int n) { return (n / 4 + n / 32) * 2 + (n > 4096 ? n / 2 : 0); }
constexpr int lds_decode_floats(int bs1) { return lds_transform_floats(bs1) + bs1 / 4; }

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
// The spectrum of one block as a lane reads it: float4 number w = lane + 64 * i at x[n2-4-4w].  The first kAhead of a lane's
// float4s can be fetched a block ahead (the loads then fly while the previous block is transformed and its samples stored).
constexpr int kAhead = 4;  // covers block sizes up to 2048; larger ones fetch the rest when they need it
struct Spectrum {
    F4 v[kAhead];
};
EA1_VS_HD F4 load_spectrum(const float *src, int n2, int w, bool aligned)
{
    const float *p = src + n2 - 4 - 4 * w;
    F4 s;
    if (aligned) {
        s = *reinterpret_cast<const F4 *>(p);
    } else {
        s.x = p[0]; s.y = p[1]; s.z = p[2]; s.w = p[3];
    }
    return s;
}
EA1_VS_HD void fetch_spectrum(Spectrum &sp, int n, const float *src, bool aligned, int lane)
{
    const int n2 = n >> 1, half = n >> 3;
#pragma unroll
    for (int i = 0; i < kAhead; i++) {
        const int w = lane + kWave * i;
        sp.v[i] = load_spectrum(src, n2, w < half ? w : half - 1, aligned);  // unconditional (a lane beyond the block re-reads its last float4)
    }
}

// pre-rotation (:7061-7087).  The float4 x[n2-4-4w .. n2-1-4w] = (R, Q, S, P) makes element C/2-1-w of the first loop
// (from the odd entries Q, P) and element C/2+w of the second (from the even entries R, S).
EA1_VS_HD void prerotate_one(float *y, const float *T, int C, int w, const F4 &s)
{
    const int half = C >> 1;
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
EA1_VS_HD void load_prerotate(float *y, const MdctTables &tb, const Spectrum &sp, const float *src, bool aligned, int lane)
{
    const int n2 = tb.n >> 1, C = tb.n >> 2, half = C >> 1;
#pragma unroll
    for (int i = 0; i < kAhead; i++) {
        const int w = lane + kWave * i;
        if (w < half) prerotate_one(y, tb.trig, C, w, sp.v[i]);
    }
    for (int w = lane + kWave * kAhead; w < half; w += kWave) prerotate_one(y, tb.trig, C, w, load_spectrum(src, n2, w, aligned));
}

// K twiddled stages (mdct_butterfly_first / _generic, :6874-6972) starting at stage s, on 2^K elements per work item
template <int K>
EA1_VS_HD void twiddled_pass(float *y, const float *tw, int log2C, int s, int lane)
{
    constexpr int R = 1 << K;
    const int log2D = log2C - s - K;  // D = smallest distance of the pass, >= 16
    const int D = 1 << log2D;
    const int groups = 1 << (log2C - K);
#pragma unroll 1
    for (int grp = lane; grp < groups; grp += kWave) {
        const int lo = grp & (D - 1);
        const int base = ((grp >> log2D) << (log2D + K)) + lo;
        // D is a multiple of 16: skew(base + m * D) = skew(base) + m * (D + D / 8)
        float *yb = y + 2 * skew(base);
        const int stride = 2 * (D + (D >> 3));
        float re[R], im[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const F2 v = *reinterpret_cast<const F2 *>(yb + m * stride);
            re[m] = v.x;
            im[m] = v.y;
        }
#pragma unroll
        for (int t = 0; t < K; t++) {
            const int span = 1 << (K - 1 - t);
            const int dist = D * span;
            const float *T = tw + 2 * ((1 << log2C) - (1 << (log2C - s - t)));  // stage s+t: entry p = T[(4 << (s+t)) * p] of mdct_init's table
#pragma unroll
            for (int m = 0; m < R; m++) {
                if (m & span) continue;
                const int u = m + span;
                const int j = lo + (m & (span - 1)) * D;  // position of the lower element in its half
                const F2 w = *reinterpret_cast<const F2 *>(T + 2 * (dist - 1 - j));
                const float r0 = re[u] - re[m];
                const float r1 = im[u] - im[m];
                re[u] = re[u] + re[m];
                im[u] = im[u] + im[m];
                re[m] = r1 * w.y + r0 * w.x;
                im[m] = r1 * w.x - r0 * w.y;
            }
        }
#pragma unroll
        for (int m = 0; m < R; m++) {
            F2 v;
            v.x = re[m];
            v.y = im[m];
            *reinterpret_cast<F2 *>(yb + m * stride) = v;
        }
    }
}

// the 32-float butterflies; element e leaves at position bitrev(e) of the unskewed array P.  Up to 64 butterflies (n <= 4096) P
// overlays y: every lane holds its 16 elements in registers before any is stored.  128 butterflies (n = 8192) would need 64
// registers per lane for that: P then lies behind y.
constexpr int kOverlayMaxN = 4096;
EA1_VS_HD void store_bit_reversed(float *P, const float *v, int g, int log2C)
{
    const int groups = 1 << (log2C - 4);
    const int rg = int(bit_reverse(uint32_t(g), log2C - 4));
#pragma unroll
    for (int m = 0; m < 16; m++) {
        constexpr int rev4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
        F2 o;
        o.x = v[2 * m];
        o.y = v[2 * m + 1];
        *reinterpret_cast<F2 *>(P + 2 * (rev4[m] * groups + rg)) = o;
    }
}
EA1_VS_HD void load_group(float *v, const float *y, int g)
{
    const F4 *p = reinterpret_cast<const F4 *>(y + 2 * skew(16 * g));
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const F4 q = p[j];
        v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
}
template <class Sync>
EA1_VS_HD float *final_pass(float *y, int n, int log2C, int lane, const Sync &sync)
{
    const int groups = 1 << (log2C - 4);
    if (n <= kOverlayMaxN) {
        float v[32];
        if (lane < groups) load_group(v, y, lane);
        sync();  // every element is in a register: the overlay may be written
        if (lane < groups) {
            butterfly_32(v);
            store_bit_reversed(y, v, lane, log2C);
        }
        return y;
    }
    float *P = y + (n / 4 + n / 32) * 2;
    for (int g = lane; g < groups; g += kWave) {
        float v[32];
        load_group(v, y, g);
        butterfly_32(v);
        store_bit_reversed(P, v, g, log2C);
    }
    return P;
}

// spectrum (its head fetched by the caller) -> bit-reversed butterfly output in LDS (returned: n/2 floats, unskewed); ends with a sync
template <class Sync>
EA1_VS_HD const float *transform_to_lds(float *y, const MdctTables &tb, const Spectrum &sp, const float *src, bool aligned, int lane, const Sync &sync)
{
    const int log2C = tb.log2n - 2;
    load_prerotate(y, tb, sp, src, aligned, lane);
    sync();
    int s = 0;
    const int stages = log2C - 4;
    while (s < stages) {
        const int rem = stages - s;
        if (rem == 1) {
            twiddled_pass<1>(y, tb.tw, log2C, s, lane);
            s += 1;
        } else if (rem == 2 || rem == 4) {
            twiddled_pass<2>(y, tb.tw, log2C, s, lane);
            s += 2;
        } else {
            twiddled_pass<3>(y, tb.tw, log2C, s, lane);
            s += 3;
        }
        sync();
    }
    float *P = final_pass(y, tb.n, log2C, lane, sync);
    sync();
    return P;
}

// mdct_bitreverse (:7004-7052) for step q and both post-rotations (:7100-7118) of its two results
struct Quad {
    float zl0, zl1;  // z[q], z[C-1-q]: the left half is z[i] for i < C and -z[2C-1-i] above
    float zr0, zr1;  // z[C+q], z[2C-1-q]: the right half is z[2C-1-i] for i < C and z[i] above
};
EA1_VS_HD Quad combine_values(const F2 &a, const F2 &b, const F2 &tq, const F2 &uq, const F2 &up)
{
    float r0 = a.y - b.y;
    float r1 = a.x + b.x;
    const float r2 = r1 * tq.x + r0 * tq.y;
    const float r3 = r1 * tq.y - r0 * tq.x;
    r0 = (a.y + b.y) * .5f;
    r1 = (a.x - b.x) * .5f;
    const float wq0 = r0 + r2, wq1 = r1 + r3;  // w[q]
    const float wp0 = r0 - r2, wp1 = r3 - r1;  // w[C-1-q]
    Quad z;
    z.zl1 = wq0 * uq.y - wq1 * uq.x;       // m = q:     z[C-1-q]
    z.zr0 = -(wq0 * uq.x + wq1 * uq.y);    //            z[C+q]
    z.zl0 = wp0 * up.y - wp1 * up.x;       // m = C-1-q: z[q]
    z.zr1 = -(wp0 * up.x + wp1 * up.y);    //            z[2C-1-q]
    return z;
}
EA1_VS_HD Quad combine(const float *P, const MdctTables &tb, int q)
{
    const int n = tb.n, n2 = n >> 1, C = n >> 2;
    const float *T = tb.trig;
    const F2 b = *reinterpret_cast<const F2 *>(P + 2 * q);            // x1 = butterfly output bitrev(q)
    const F2 a = *reinterpret_cast<const F2 *>(P + 2 * (C - 1 - q));  // x0 = butterfly output C-1-bitrev(q)
    const F2 tq = *reinterpret_cast<const F2 *>(T + n + 2 * q);
    const F2 uq = *reinterpret_cast<const F2 *>(T + n2 + 2 * q);
    const F2 up = *reinterpret_cast<const F2 *>(T + n2 + 2 * (C - 1 - q));
    return combine_values(a, b, tq, uq, up);
}
// steps q and q+1 (q even) from 128-bit reads: element pairs (q, q+1) ascending, (C-2-q, C-1-q) for the mirrored operands
EA1_VS_HD void combine_pair(const float *P, const MdctTables &tb, int q, Quad &z0, Quad &z1)
{
    const int n = tb.n, n2 = n >> 1, C = n >> 2;
    const float *T = tb.trig;
    const F4 b = *reinterpret_cast<const F4 *>(P + 2 * q);
    const F4 a = *reinterpret_cast<const F4 *>(P + 2 * (C - 2 - q));
    const F4 tq = *reinterpret_cast<const F4 *>(T + n + 2 * q);
    const F4 uq = *reinterpret_cast<const F4 *>(T + n2 + 2 * q);
    const F4 up = *reinterpret_cast<const F4 *>(T + n2 + 2 * (C - 2 - q));
    z0 = combine_values(F2{a.z, a.w}, F2{b.x, b.y}, F2{tq.x, tq.y}, F2{uq.x, uq.y}, F2{up.z, up.w});
    z1 = combine_values(F2{a.x, a.y}, F2{b.z, b.w}, F2{tq.z, tq.w}, F2{uq.z, uq.w}, F2{up.x, up.y});
}

// ---------------------------------------------------------------- mdct_backward alone: n/2 floats in, n floats out
template <class Sync>
EA1_VS_HD void wave_mdct_backward(float *lds, const MdctTables &tb, const float *src, bool aligned, float *dst, int lane, const Sync &sync)
{
    Spectrum sp;
    fetch_spectrum(sp, tb.n, src, aligned, lane);
    const float *P = transform_to_lds(lds, tb, sp, src, aligned, lane, sync);
    const int n = tb.n, n2 = n >> 1, C = n >> 2;
#pragma unroll 1
    for (int q = lane; q < (C >> 1); q += kWave) {
        const Quad z = combine(P, tb, q);
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

// combine + window + overlap-add of a block that follows one of its own size, in place: step q overlaps with R[q] and R[C-1-q] only, the two
// slots it replaces.  Emit's four samples written out: sample i meets right_half(i) * win[2C-1-i] (:9243-9250 long/long, :9270-9277
// short/short).  ITEMS > 0: the trip count (C/4 / 64) known at compile time.
template <int ITEMS>
EA1_VS_HD void overlap_same_size(const float *P, const MdctTables &tb, float *R, float *o, bool emit, int lane)
{
    const int C = tb.n >> 2;
    const float *w = tb.win;
    // two adjacent steps per work item: every operand pair is one 64- or 128-bit access, the arithmetic pairs up
    auto step2 = [&](int q) {
        Quad z0, z1;
        combine_pair(P, tb, q, z0, z1);
        const int qc = C - 2 - q;  // mirrored pair (qc, qc+1) = (C-2-q, C-1-q)
        const F2 Rq = *reinterpret_cast<const F2 *>(R + q), Rc = *reinterpret_cast<const F2 *>(R + qc);
        if (emit) {
            const F2 wq = *reinterpret_cast<const F2 *>(w + q), wc1 = *reinterpret_cast<const F2 *>(w + qc);
            const F2 wc = *reinterpret_cast<const F2 *>(w + C + q), wn = *reinterpret_cast<const F2 *>(w + C + qc);
            // step q:   o[q], o[C-1-q], o[C+q], o[2C-1-q];   step q+1: o[q+1], o[C-2-q], o[C+q+1], o[2C-2-q]
            o[q] = Rc.y * wn.y + z0.zl0 * wq.x;
            o[q + 1] = Rc.x * wn.x + z1.zl0 * wq.y;
            o[qc] = Rq.y * wc.y + z1.zl1 * wc1.x;
            o[qc + 1] = Rq.x * wc.x + z0.zl1 * wc1.y;
            o[C + q] = Rq.x * wc1.y + -z0.zl1 * wc.x;
            o[C + q + 1] = Rq.y * wc1.x + -z1.zl1 * wc.y;
            o[C + qc] = Rc.x * wq.y + -z1.zl0 * wn.x;
            o[C + qc + 1] = Rc.y * wq.x + -z0.zl0 * wn.y;
        }
        F2 r;
        r.x = z0.zr0; r.y = z1.zr0;
        *reinterpret_cast<F2 *>(R + q) = r;
        r.x = z1.zr1; r.y = z0.zr1;
        *reinterpret_cast<F2 *>(R + qc) = r;
    };
    if (ITEMS > 0) {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) step2(2 * (lane + kWave * j));
    } else {
#pragma unroll 1
        for (int p = lane; p < (C >> 2); p += kWave) step2(2 * p);
    }
}

// one (run, channel) of one stream: blocks [run * kRun, ...) of channel ch.  lds: lds_decode_floats(bs1) floats of this wave.
template <class Sync>
EA1_VS_HD void wave_decode_run(const MdctTables &tb0, const MdctTables &tb1, const float *spectra, const BlockMeta *meta, int nblocks,
                               int channels, float *pcm, size_t pcm_stride, const float *tail_in, float *tail_out, float *lds, int run,
                               int ch, int lane, const Sync &sync)
{
    const int bs1 = tb1.n, bs0 = tb0.n;
    float *y = lds;
    float *R = lds + lds_transform_floats(bs1);  // upper half of the previous block's right half
    const int first = run * kRun;
    const int last = first + kRun < nblocks ? first + kRun : nblocks;
    const size_t row = size_t(bs1 / 2);
    const bool aligned = (reinterpret_cast<uintptr_t>(spectra) & 15) == 0;
    float *out = pcm + size_t(ch) * pcm_stride;
    // the carried right half is double-buffered in global memory: the first run reads last call's copy while the last run of
    // THIS launch writes the other one
    const float *tl_in = tail_in + size_t(ch) * row;
    float *tl_out = tail_out + size_t(ch) * row;

    Spectrum sp;  // head of the next block to transform
    const BlockMeta mfirst = load_meta(meta + first);
    if (first == 0) {
        fetch_spectrum(sp, mfirst.W ? bs1 : bs0, spectra + size_t(ch) * row, aligned, lane);
        const int Cp = (mfirst.lW ? bs1 : bs0) >> 2;
        for (int i = lane; i < Cp; i += kWave) R[i] = tl_in[Cp + i];
        sync();
    } else {  // recomputed: block first-1 belongs to the neighbouring run
        const int k = first - 1;
        const MdctTables &tb = mfirst.lW ? tb1 : tb0;
        const float *src = spectra + (size_t(k) * channels + ch) * row;
        fetch_spectrum(sp, tb.n, src, aligned, lane);
        const float *P = transform_to_lds(y, tb, sp, src, aligned, lane, sync);
        fetch_spectrum(sp, mfirst.W ? bs1 : bs0, spectra + (size_t(first) * channels + ch) * row, aligned, lane);
        const int C = tb.n >> 2;
#pragma unroll 1
        for (int q = lane; q < (C >> 1); q += kWave) {
            const Quad z = combine(P, tb, q);
            R[q] = z.zr0;
            R[C - 1 - q] = z.zr1;
        }
        sync();
    }

    BlockMeta m = mfirst;
    for (int k = first; k < last; k++) {
        const MdctTables &tb = m.W ? tb1 : tb0;
        const BlockMeta mnext = load_meta(meta + (k + 1 < last ? k + 1 : k));
        const float *P = transform_to_lds(y, tb, sp, spectra + (size_t(k) * channels + ch) * row, aligned, lane, sync);
        if (k + 1 < last) fetch_spectrum(sp, mnext.W ? bs1 : bs0, spectra + (size_t(k + 1) * channels + ch) * row, aligned, lane);
        const int C = tb.n >> 2;
        const int n0 = bs0 / 2, n1 = bs1 / 2;
        Emit e;
        e.o = out + m.out_off;
        e.Rp = R;
        e.Cp = (m.lW ? bs1 : bs0) >> 2;
        const bool same = m.lW == m.W || bs0 == bs1;
        if (same) {         // long/long, short/short
            e.win = tb.win; e.shift = 0; e.wh = tb.n >> 1; e.wmode = 0;
        } else if (m.lW) {  // long/short: the head of the long right half passes through
            const int lead = n1 / 2 - n0 / 2;
            e.win = tb0.win; e.shift = lead; e.wh = n0; e.wmode = 0;
            if (m.count > 0)
                for (int s = lane; s < lead; s += kWave) e.o[s] = right_half(R, e.Cp, s);
        } else {            // short/long: the long left half starts at n1/2 - n0/2
            e.win = tb0.win; e.shift = -(n1 / 2 - n0 / 2); e.wh = n0; e.wmode = 1;
        }
        if (same) {
            // step q overlaps with R[q] and R[C-1-q] only -- the two slots it replaces: in place.  Emit's four samples written out:
            // sample i meets right_half(i) * win[2C-1-i] (:9243-9250 long/long, :9270-9277 short/short)
            if (C == 512)
                overlap_same_size<2>(P, tb, R, e.o, m.count > 0, lane);   // n = 2048: the loop unrolled, addresses as immediates
            else
                overlap_same_size<0>(P, tb, R, e.o, m.count > 0, lane);
        } else {
            // a change of block size reads the old right half across lanes: all of it first, then the new one (two such blocks
            // per run of short blocks: their combine step runs twice)
            if (m.count > 0) {
#pragma unroll 1
                for (int q = lane; q < (C >> 1); q += kWave) {
                    const Quad z = combine(P, tb, q);
                    e(q, z.zl0);
                    e(C - 1 - q, z.zl1);
                    e(C + q, -z.zl1);
                    e(2 * C - 1 - q, -z.zl0);
                }
                sync();
            }
#pragma unroll 1
            for (int q = lane; q < (C >> 1); q += kWave) {
                const Quad z = combine(P, tb, q);
                R[q] = z.zr0;
                R[C - 1 - q] = z.zr1;
            }
        }
        sync();
        m = mnext;
    }
    if (last == nblocks) {
        const int n2 = (uniform(meta[nblocks - 1].W) ? bs1 : bs0) >> 1;
        for (int i = lane; i < n2; i += kWave) tl_out[i] = right_half(R, n2 >> 1, i);
    }
}

}  // namespace vsynth
}  // namespace ea1
