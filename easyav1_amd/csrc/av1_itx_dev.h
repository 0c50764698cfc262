// AV1 inverse transforms as compile-time-unrolled device code (gfx950).
//
// What it computes: AV1 spec section 7.13 / dav1d src/itx_1d.c + src/itx_tmpl.c (listed at
// /root/reference/cmake/dav1d.cmake:979,:1008; sources absent): integer butterfly
// networks with ONE Round2(.,12) per rotation, int16 saturation after every add/sub stage, so
// this is NOT a matrix product and does not belong on MFMA (SURVEY.md finding 9).
//
// How it is mapped: one lane owns one whole 1-D transform and keeps all N values in VGPRs; every
// array index below is a compile-time constant after unrolling, so the arrays are promoted to
// registers (check: .private_segment_fixed_size == 0 in the ISA).  The DCT is a template recursion
// on natural-order input (even half = half-length DCT of the even inputs, odd half = rotation /
// add-sub network), written independently of the oracle's slot-based loop formulation.
#pragma once
#include <hip/hip_runtime.h>

#include "ea1_av1_tables.inc"
#include "easyav1_amd_video.h"

namespace ea1 {
namespace itx {

__device__ static constexpr int kCos[65] = {EA1_COSPI_VALUES};

__device__ __forceinline__ int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
// Round2(w0 * a + w1 * b, 12); operands are int16-range so the sum stays inside int32
__device__ __forceinline__ int rot(int w0, int a, int w1, int b) { return (w0 * a + w1 * b + 2048) >> 12; }

__host__ __device__ constexpr int cbitrev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++)
        if (v & (1 << i)) r |= 1 << (bits - 1 - i);
    return r;
}
__host__ __device__ constexpr int clog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

// odd half of a 2M-point inverse DCT; o[j] is "slot M + j"
template <int M>
__device__ __forceinline__ void idct_odd(int (&o)[M])
{
    constexpr int L = clog2(M);
    constexpr int N = 2 * M;
    // input rotations: slot j holds input k = bitrev(M + j), its mirror holds N - k
#pragma unroll
    for (int j = 0; j < M / 2; j++) {
        const int k = cbitrev(M + j, L + 1);
        const int A = 64 * k / N;
        const int x = o[j], y = o[M - 1 - j];
        o[j] = rot(kCos[64 - A], x, -kCos[A], y);
        o[M - 1 - j] = rot(kCos[A], x, kCos[64 - A], y);
    }
#pragma unroll
    for (int t = 1; t <= L - 1; t++) {
        const int g = 1 << t;
#pragma unroll
        for (int q = 0; q < M / g; q++) {
#pragma unroll
            for (int i = 0; i < g / 2; i++) {
                const int lo = q * g + i, hi = q * g + g - 1 - i;
                const int a = o[lo], b = o[hi];
                if ((q & 1) == 0) {
                    o[lo] = clip16(a + b);
                    o[hi] = clip16(a - b);
                } else {
                    o[lo] = clip16(b - a);
                    o[hi] = clip16(b + a);
                }
            }
        }
        if (t == L - 1) {
#pragma unroll
            for (int lo = M / 4; lo < M / 2; lo++) {
                const int hi = M - 1 - lo;
                const int x = o[lo], y = o[hi];
                o[lo] = rot(-kCos[32], x, kCos[32], y);
                o[hi] = rot(kCos[32], x, kCos[32], y);
            }
        } else {
            const int B = 1 << (t + 1);
            const int nblocks = (M / 2) / B;
            const int bits = clog2(nblocks) + 1;
#pragma unroll
            for (int b = 0; b < nblocks; b++) {
                const int kb = 2 * cbitrev(b, bits) + 1;
                const int A = 16 * kb / nblocks;
#pragma unroll
                for (int i = 0; i < B / 4; i++) {
                    {
                        const int lo = b * B + B / 4 + i, hi = M - 1 - lo;
                        const int x = o[lo], y = o[hi];
                        o[lo] = rot(-kCos[A], x, kCos[64 - A], y);
                        o[hi] = rot(kCos[64 - A], x, kCos[A], y);
                    }
                    {
                        const int lo = b * B + B / 2 + i, hi = M - 1 - lo;
                        const int x = o[lo], y = o[hi];
                        o[lo] = rot(-kCos[64 - A], x, -kCos[A], y);
                        o[hi] = rot(-kCos[A], x, kCos[64 - A], y);
                    }
                }
            }
        }
    }
}

template <int N>
struct IDct {
    static __device__ __forceinline__ void run(const int (&in)[N], int (&out)[N])
    {
        constexpr int M = N / 2;
        constexpr int L = clog2(M);
        int ein[M], e[M], o[M];
#pragma unroll
        for (int i = 0; i < M; i++) ein[i] = in[2 * i];
        IDct<M>::run(ein, e);
#pragma unroll
        for (int j = 0; j < M; j++) o[j] = in[cbitrev(M + j, L + 1)];
        idct_odd<M>(o);
#pragma unroll
        for (int i = 0; i < M; i++) {
            out[i] = clip16(e[i] + o[M - 1 - i]);
            out[N - 1 - i] = clip16(e[i] - o[M - 1 - i]);
        }
    }
};
template <>
struct IDct<2> {
    static __device__ __forceinline__ void run(const int (&in)[2], int (&out)[2])
    {
        out[0] = rot(kCos[32], in[0], kCos[32], in[1]);
        out[1] = rot(kCos[32], in[0], -kCos[32], in[1]);
    }
};

__device__ __forceinline__ void iadst4(const int (&in)[4], int (&out)[4])
{
    const int x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    int s0 = EA1_SINPI_1_9 * x0;
    int s1 = EA1_SINPI_2_9 * x0;
    int s2 = EA1_SINPI_3_9 * x1;
    int s3 = EA1_SINPI_4_9 * x2;
    const int s4 = EA1_SINPI_1_9 * x2;
    const int s5 = EA1_SINPI_2_9 * x3;
    const int s6 = EA1_SINPI_4_9 * x3;
    const int b7 = x0 - x2 + x3;
    s0 = s0 + s3 + s5;
    s1 = s1 - s4 - s6;
    s3 = s2;
    s2 = EA1_SINPI_3_9 * b7;
    out[0] = (s0 + s3 + 2048) >> 12;
    out[1] = (s1 + s3 + 2048) >> 12;
    out[2] = (s2 + 2048) >> 12;
    out[3] = (s0 + s1 - s3 + 2048) >> 12;
}

// (lo, hi) <- (c[A] lo + c[64-A] hi, c[64-A] lo - c[A] hi)
#define EA1_RA(lo, hi, A)                                     \
    {                                                         \
        const int a_ = b[lo], b_ = b[hi];                     \
        b[lo] = rot(kCos[A], a_, kCos[64 - (A)], b_);         \
        b[hi] = rot(kCos[64 - (A)], a_, -kCos[A], b_);        \
    }
// (lo, hi) <- (-c[64-A] lo + c[A] hi, c[A] lo + c[64-A] hi)
#define EA1_RB(lo, hi, A)                                     \
    {                                                         \
        const int a_ = b[lo], b_ = b[hi];                     \
        b[lo] = rot(-kCos[64 - (A)], a_, kCos[A], b_);        \
        b[hi] = rot(kCos[A], a_, kCos[64 - (A)], b_);         \
    }
#define EA1_AS(lo, hi)                                        \
    {                                                         \
        const int a_ = b[lo], b_ = b[hi];                     \
        b[lo] = clip16(a_ + b_);                              \
        b[hi] = clip16(a_ - b_);                              \
    }

__device__ __forceinline__ void iadst8(const int (&in)[8], int (&out)[8])
{
    int b[8] = {in[7], in[0], in[5], in[2], in[3], in[4], in[1], in[6]};
    EA1_RA(0, 1, 4) EA1_RA(2, 3, 20) EA1_RA(4, 5, 36) EA1_RA(6, 7, 52)
    EA1_AS(0, 4) EA1_AS(1, 5) EA1_AS(2, 6) EA1_AS(3, 7)
    EA1_RA(4, 5, 16) EA1_RB(6, 7, 16)
    EA1_AS(0, 2) EA1_AS(1, 3) EA1_AS(4, 6) EA1_AS(5, 7)
    EA1_RA(2, 3, 32) EA1_RA(6, 7, 32)
    out[0] = b[0]; out[1] = -b[4]; out[2] = b[6]; out[3] = -b[2];
    out[4] = b[3]; out[5] = -b[7]; out[6] = b[5]; out[7] = -b[1];
}

__device__ __forceinline__ void iadst16(const int (&in)[16], int (&out)[16])
{
    int b[16] = {in[15], in[0], in[13], in[2], in[11], in[4], in[9], in[6],
                 in[7], in[8], in[5], in[10], in[3], in[12], in[1], in[14]};
    EA1_RA(0, 1, 2) EA1_RA(2, 3, 10) EA1_RA(4, 5, 18) EA1_RA(6, 7, 26)
    EA1_RA(8, 9, 34) EA1_RA(10, 11, 42) EA1_RA(12, 13, 50) EA1_RA(14, 15, 58)
    EA1_AS(0, 8) EA1_AS(1, 9) EA1_AS(2, 10) EA1_AS(3, 11) EA1_AS(4, 12) EA1_AS(5, 13) EA1_AS(6, 14) EA1_AS(7, 15)
    EA1_RA(8, 9, 8) EA1_RA(10, 11, 40) EA1_RB(12, 13, 8) EA1_RB(14, 15, 40)
    EA1_AS(0, 4) EA1_AS(1, 5) EA1_AS(2, 6) EA1_AS(3, 7) EA1_AS(8, 12) EA1_AS(9, 13) EA1_AS(10, 14) EA1_AS(11, 15)
    EA1_RA(4, 5, 16) EA1_RB(6, 7, 16) EA1_RA(12, 13, 16) EA1_RB(14, 15, 16)
    EA1_AS(0, 2) EA1_AS(1, 3) EA1_AS(4, 6) EA1_AS(5, 7) EA1_AS(8, 10) EA1_AS(9, 11) EA1_AS(12, 14) EA1_AS(13, 15)
    EA1_RA(2, 3, 32) EA1_RA(6, 7, 32) EA1_RA(10, 11, 32) EA1_RA(14, 15, 32)
    out[0] = b[0];   out[1] = -b[8];   out[2] = b[12];  out[3] = -b[4];
    out[4] = b[6];   out[5] = -b[14];  out[6] = b[10];  out[7] = -b[2];
    out[8] = b[3];   out[9] = -b[11];  out[10] = b[15]; out[11] = -b[7];
    out[12] = b[5];  out[13] = -b[13]; out[14] = b[9];  out[15] = -b[1];
}
#undef EA1_RA
#undef EA1_RB
#undef EA1_AS

template <int N>
__device__ __forceinline__ void iidentity(const int (&in)[N], int (&out)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (N == 4) out[i] = (in[i] * 5793 + 2048) >> 12;
        else if constexpr (N == 8) out[i] = in[i] * 2;
        else if constexpr (N == 16) out[i] = (in[i] * 11586 + 2048) >> 12;
        else out[i] = in[i] * 4;
    }
}

// inverse Walsh-Hadamard, 4 points (AV1 spec 7.13.2.10)
__device__ __forceinline__ void iwht4(int (&io)[4], int shift)
{
    int a = io[0] >> shift, c = io[1] >> shift, d = io[2] >> shift, b = io[3] >> shift;
    a += c;
    d -= b;
    const int e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= b;
    d += c;
    io[0] = a;
    io[1] = b;
    io[2] = c;
    io[3] = d;
}

// one 1-D inverse transform of runtime kind (EA1_TX_*) and compile-time length.
// NZ = number of leading inputs that can be non-zero (32 for the 64-point transforms).
template <int N>
__device__ __forceinline__ void inv_txfm_1d(int kind, const int (&in)[N], int (&out)[N])
{
    if constexpr (N == 64) {
        IDct<64>::run(in, out);
    } else if constexpr (N == 32) {
        if (kind == EA1_TX_IDENTITY) iidentity<32>(in, out);
        else IDct<32>::run(in, out);
    } else {
        if (kind == EA1_TX_DCT) {
            IDct<N>::run(in, out);
        } else if (kind == EA1_TX_IDENTITY) {
            iidentity<N>(in, out);
        } else {
            int t[N];
            if constexpr (N == 4) iadst4(in, t);
            else if constexpr (N == 8) iadst8(in, t);
            else iadst16(in, t);
            if (kind == EA1_TX_FLIPADST) {
#pragma unroll
                for (int i = 0; i < N; i++) out[i] = t[N - 1 - i];
            } else {
#pragma unroll
                for (int i = 0; i < N; i++) out[i] = t[i];
            }
        }
    }
}

__host__ __device__ constexpr int row_shift(int w, int h)
{
    // Transform_Row_Shift by shape (AV1 spec 7.13.3)
    const int lw = clog2(w), lh = clog2(h);
    if (lw == lh) return lw == 2 ? 0 : (lw == 3 ? 1 : 2);
    const int mn = lw < lh ? lw : lh, mx = lw < lh ? lh : lw;
    if (mx - mn == 1) return mn == 2 ? 0 : 1;      // 2:1: 4x8 -> 0, 8x16/16x32/32x64 -> 1
    return mn == 2 ? 1 : 2;                        // 4:1: 4x16 -> 1, 8x32/16x64 -> 2
}

// LDS needed by tx_block_add per transform block, in int16 units
template <int W, int H>
struct TxLds {
    static constexpr int SW = W < 32 ? W : 32, SH = H < 32 ? H : 32;
    static constexpr int CSTRIDE = SW + 2;  // +1 dword: row stride becomes odd in dwords, no bank conflicts
    static constexpr int TSTRIDE = W + 2;
    static constexpr int COEF = SH * CSTRIDE;
    static constexpr int TMP = SH * TSTRIDE;
    // after the column pass the same area holds the H x W residual (row stride RSTRIDE) for the
    // row-wise, 4-samples-per-lane add to the picture
    static constexpr int RSTRIDE = W + 4;
    static constexpr int RES = H * RSTRIDE;
    static constexpr int TOTAL = (COEF + TMP) > RES ? (COEF + TMP) : RES;
    static constexpr int THREADS = SH > W ? SH : W;  // lanes cooperating on one block
};

// Inverse transform + add for one W x H block, run by TxLds<W,H>::THREADS consecutive lanes
// (`sub` = lane index inside that group).  `lds` is this block's private LDS area.
//
// NZ: all non-zero coefficients lie in the top-left NZ x NZ corner (the parser knows this from the
// end-of-block position; dav1d uses eob the same way to skip zero rows).  Rows >= NZ are not
// transformed at all, and every 1-D transform sees literal zeros beyond NZ, which the compiler
// folds away -- a 32-point DCT with 8 live inputs is a fraction of the full network.  Only the
// NZ x NZ corner of the coefficient block is fetched.
//
// If `resid` is non-NULL the rounded residual is stored there (int16, row stride `rstride`) instead
// of being added to `dst`: used for intra blocks, whose prediction only exists once the wavefront
// reaches them (the transform itself has no such dependency and runs in the parallel phase).
// `corner`: the coefficient block is stored as its NW x NH corner alone (row stride NW, EA1_COEFS_CORNER) instead of
// inside the dense min(W,32) x min(H,32) block (row stride SW).
template <int W, int H, int NZ>
__device__ __forceinline__ void tx_block_add(int16_t *lds, const int16_t *__restrict__ coef, uint8_t *dst, int stride,
                                            int txtp, int sub, bool active, int16_t *resid = nullptr, int rstride = 0, bool corner = false)
{
    using T = TxLds<W, H>;
    constexpr int SW = T::SW, SH = T::SH;
    constexpr int NW = NZ < SW ? NZ : SW, NH = NZ < SH ? NZ : SH;   // live columns / rows
    const int cstride = corner ? NW : SW;
    int16_t *cl = lds;
    int16_t *tmp = lds + T::COEF;
    // stage the live corner (32-bit copies)
    if (active) {
        for (int i = sub; i < NW * NH / 2; i += T::THREADS) {
            const int y = (2 * i) / NW, x = (2 * i) % NW;
            *reinterpret_cast<uint32_t *>(&cl[y * T::CSTRIDE + x]) = *reinterpret_cast<const uint32_t *>(coef + y * cstride + x);
        }
    }
    __syncthreads();
    constexpr int shift = row_shift(W, H);
    constexpr bool rect2 = (W == 2 * H) || (H == 2 * W);
    bool wht = false;
    if constexpr (W == 4 && H == 4) wht = txtp == EA1_TXTP_WHT;   // lossless Walsh-Hadamard (7.13.2.10)
    if (wht) {
        if (active && sub < 4) {
            int t[4];
#pragma unroll
            for (int x = 0; x < 4; x++) t[x] = x < NW && sub < NH ? (int)cl[sub * T::CSTRIDE + x] : 0;
            iwht4(t, 2);
#pragma unroll
            for (int x = 0; x < 4; x++) tmp[sub * T::TSTRIDE + x] = (int16_t)t[x];
        }
    } else if (active && sub < NH) {
        int in[W], out[W];
#pragma unroll
        for (int x = 0; x < W; x++) {
            if (x < NW) {
                int c = cl[sub * T::CSTRIDE + x];
                if (rect2) c = (c * 181 + 128) >> 8;
                in[x] = c;
            } else {
                in[x] = 0;
            }
        }
        inv_txfm_1d<W>(txtp & 3, in, out);
#pragma unroll
        for (int x = 0; x < W; x++) tmp[sub * T::TSTRIDE + x] = (int16_t)clip16((out[x] + ((1 << shift) >> 1)) >> shift);
    }
    __syncthreads();
    int colres[H];
    if (active && sub < W) {
        int in[H];
#pragma unroll
        for (int y = 0; y < H; y++) in[y] = y < NH ? (int)tmp[y * T::TSTRIDE + sub] : 0;
        if (wht) {
            if constexpr (H == 4) {
#pragma unroll
                for (int y = 0; y < 4; y++) in[y] = (int)tmp[y * T::TSTRIDE + sub];
                iwht4(in, 0);
#pragma unroll
                for (int y = 0; y < 4; y++) colres[y] = in[y] * 16 - 8;   // undo the (x + 8) >> 4 below: no rounding shift
            }
        } else {
            inv_txfm_1d<H>((txtp >> 2) & 3, in, colres);
        }
    }
    __syncthreads();  // every column has been read: the area can be reused for the residual
    // residual block to LDS (column per lane), then a row-wise pass with 4 samples per lane:
    // one dword load + one dword store per 4 samples instead of 4 + 4 byte accesses
    int16_t *rl = lds;
    if (active && sub < W) {
#pragma unroll
        for (int y = 0; y < H; y++) rl[y * T::RSTRIDE + sub] = (int16_t)((colres[y] + 8) >> 4);
    }
    __syncthreads();
    if (active) {
        for (int i = sub; i < W * H / 4; i += T::THREADS) {
            const int y = i / (W / 4), g = i - y * (W / 4);
            const uint2 r = *reinterpret_cast<const uint2 *>(&rl[y * T::RSTRIDE + 4 * g]);
            if (resid) {
                *reinterpret_cast<uint2 *>(resid + y * rstride + 4 * g) = r;
            } else {
                uint32_t *d = reinterpret_cast<uint32_t *>(dst + y * stride + 4 * g);
                const uint32_t px = *d;
                const int r4[4] = {int16_t(r.x & 0xFFFF), int16_t(r.x >> 16), int16_t(r.y & 0xFFFF), int16_t(r.y >> 16)};
                uint32_t pk = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int v = int((px >> (8 * j)) & 255) + r4[j];
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    pk |= uint32_t(v) << (8 * j);
                }
                *d = pk;
            }
        }
    }
}

// run-time dispatch on the live-corner class nzl = log2(NZ) in 2..5 (NZ = 4, 8, 16, 32)
template <int W, int H>
__device__ __forceinline__ void tx_block_add_nz(int nzl, int16_t *lds, const int16_t *__restrict__ coef, uint8_t *dst,
                                               int stride, int txtp, int sub, bool active, int16_t *resid = nullptr,
                                               int rstride = 0, bool corner = false)
{
    // a corner class is worth a variant when it is smaller than the stored block in at least one direction; the
    // region read is then exactly the stated corner clipped to the block (what the IR promises: coefficients beyond
    // ea1_txblock.nz_log2 are not read -- the oracle masks the same region)
    constexpr int L = (W > H ? W : H) < 32 ? (W > H ? W : H) : 32;
    if constexpr (L > 4) {
        if (nzl <= 2) {
            tx_block_add<W, H, 4>(lds, coef, dst, stride, txtp, sub, active, resid, rstride, corner);
            return;
        }
    }
    if constexpr (L > 8) {
        if (nzl == 3) {
            tx_block_add<W, H, 8>(lds, coef, dst, stride, txtp, sub, active, resid, rstride, corner);
            return;
        }
    }
    if constexpr (L > 16) {
        if (nzl == 4) {
            tx_block_add<W, H, 16>(lds, coef, dst, stride, txtp, sub, active, resid, rstride, corner);
            return;
        }
    }
    tx_block_add<W, H, 32>(lds, coef, dst, stride, txtp, sub, active, resid, rstride, corner);
}

}  // namespace itx
}  // namespace ea1
