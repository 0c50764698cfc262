// Film grain synthesis (AV1 spec 7.18.3; dav1d src/filmgrain_tmpl.c + src/fg_apply_tmpl.c,
// /root/reference/cmake/dav1d.cmake:1005,:1004): SURVEY.md 8(a) row V10.
//
// Host part (grain_prepare, per picture): the sequential pieces -- the 16-bit LFSR, the grain
// templates with their auto-regressive filter, the scaling look-up tables and one random offset
// per 32x32 luma block -- packed into one blob per chain.
// GPU part (film_grain_kernel): one 256-lane workgroup per pair of 32x32 luma blocks and their
// 16x16 chroma blocks: 8 luma + 4 chroma samples per lane.  The specification builds 34-row noise stripes block by block
// and blends the two columns / rows in which neighbouring blocks overlap; here every sample
// evaluates that closed form directly: template value at the block's offset, blended with the
// left block's columns 32, 33 (chroma: 16) and -- for the first two rows (one row) of a stripe --
// with rows 32, 33 (16) of the stripe above, themselves blended horizontally.  Traffic: picture
// read once + written once (+ the luma row pair a chroma sample scales by: L2 hits), templates and
// tables are ~10 KB per chain and stay in L2.  HBM bound.
#include <algorithm>
#include <cstring>

#include "av1_dev.h"
#include "ea1_common.h"
#include "easyav1_amd_grain.h"
#include "film_grain.h"

namespace ea1 {

namespace {

struct Lfsr {
    uint32_t r;
    int get(int bits)
    {
        const uint32_t bit = ((r >> 0) ^ (r >> 1) ^ (r >> 3) ^ (r >> 12)) & 1;
        r = (r >> 1) | (bit << 15);
        return int((r >> (16 - bits)) & ((1u << bits) - 1));
    }
};

inline int rnd2(int v, int n) { return n ? (v + (1 << (n - 1))) >> n : v; }
inline int clampi_h(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }

void scaling_lut(int n, const uint8_t pts[][2], uint8_t *lut)
{
    memset(lut, 0, 256);
    if (n <= 0) return;
    for (int i = 0; i < pts[0][0]; i++) lut[i] = pts[0][1];
    for (int k = 0; k + 1 < n; k++) {
        const int dx = pts[k + 1][0] - pts[k][0], dy = pts[k + 1][1] - pts[k][1];
        const int delta = dy * ((65536 + (dx >> 1)) / dx);
        for (int x = 0; x < dx; x++) lut[pts[k][0] + x] = uint8_t(pts[k][1] + ((x * delta + 32768) >> 16));
    }
    for (int i = pts[n - 1][0]; i < 256; i++) lut[i] = pts[n - 1][1];
}

}  // namespace

int grain_validate(const ea1_film_grain &fg)
{
    if (fg.num_y_points > 14 || fg.num_uv_points[0] > 10 || fg.num_uv_points[1] > 10) return fail(EA1_EINVAL, "film grain: number of scaling points");
    auto increasing = [](int n, const uint8_t pts[][2]) {
        for (int i = 1; i < n; i++)
            if (pts[i][0] <= pts[i - 1][0]) return false;
        return true;
    };
    if (!increasing(fg.num_y_points, fg.point_y) || !increasing(fg.num_uv_points[0], fg.point_uv[0]) || !increasing(fg.num_uv_points[1], fg.point_uv[1]))
        return fail(EA1_EINVAL, "film grain: scaling point values must increase");
    if (fg.scaling_shift < 8 || fg.scaling_shift > 11 || fg.ar_coeff_lag > 3 || fg.ar_coeff_shift < 6 || fg.ar_coeff_shift > 9 || fg.grain_scale_shift > 3 ||
        fg.uv_offset[0] > 511 || fg.uv_offset[1] > 511)
        return fail(EA1_EINVAL, "film grain: parameter out of range");
    return EA1_OK;
}

size_t grain_blob_bytes(int w, int h) { return kGrainOffsets + size_t((h + 31) / 32) * size_t((w + 31) / 32); }

void grain_prepare(const ea1_film_grain &fg, const int16_t *gauss, int w, int h, uint8_t *blob)
{
    // ---- grain templates (7.18.3.3), 8-bit: values in [-128, 127] ----
    static thread_local int luma[73][82], chroma[2][38][44];
    Lfsr rng{fg.grain_seed};
    const int gshift = 12 - 8 + fg.grain_scale_shift, lag = fg.ar_coeff_lag, ashift = fg.ar_coeff_shift;
    for (int y = 0; y < 73; y++)
        for (int x = 0; x < 82; x++) luma[y][x] = fg.num_y_points ? rnd2(gauss[rng.get(11)], gshift) : 0;
    for (int y = 3; y < 73; y++)
        for (int x = 3; x < 79; x++) {
            int sum = 0;
            const int8_t *c = fg.ar_coeffs_y;
            for (int dr = -lag; dr <= 0; dr++)
                for (int dc = -lag; dc <= lag && (dr || dc); dc++) sum += *c++ * luma[y + dr][x + dc];
            luma[y][x] = clampi_h(luma[y][x] + rnd2(sum, ashift), -128, 127);
        }
    for (int p = 0; p < 2; p++) {
        rng.r = fg.grain_seed ^ (p ? 0x49d8u : 0xb524u);
        const bool on = fg.num_uv_points[p] || fg.chroma_scaling_from_luma;
        for (int y = 0; y < 38; y++)
            for (int x = 0; x < 44; x++) chroma[p][y][x] = on ? rnd2(gauss[rng.get(11)], gshift) : 0;
    }
    for (int y = 3; y < 38; y++)
        for (int x = 3; x < 41; x++) {
            int lavg = 0;
            if (fg.num_y_points) {
                const int lx = 2 * (x - 3) + 3, ly = 2 * (y - 3) + 3;
                lavg = rnd2(luma[ly][lx] + luma[ly][lx + 1] + luma[ly + 1][lx] + luma[ly + 1][lx + 1], 2);
            }
            for (int p = 0; p < 2; p++) {
                int sum = 0;
                const int8_t *c = fg.ar_coeffs_uv[p];
                for (int dr = -lag; dr <= 0; dr++)
                    for (int dc = -lag; dc <= lag && (dr || dc); dc++) sum += *c++ * chroma[p][y + dr][x + dc];
                sum += *c * lavg;   // the coefficient after the causal neighbourhood weighs the luma grain
                chroma[p][y][x] = clampi_h(chroma[p][y][x] + rnd2(sum, ashift), -128, 127);
            }
        }
    int8_t *tl = reinterpret_cast<int8_t *>(blob + kGrainLuma);
    for (int y = 0; y < 73; y++)
        for (int x = 0; x < 82; x++) tl[y * 82 + x] = int8_t(luma[y][x]);
    for (int p = 0; p < 2; p++) {
        int8_t *tc = reinterpret_cast<int8_t *>(blob + (p ? kGrainCr : kGrainCb));
        for (int y = 0; y < 38; y++)
            for (int x = 0; x < 44; x++) tc[y * 44 + x] = int8_t(chroma[p][y][x]);
    }
    // ---- scaling look-up tables (7.18.3.4) ----
    scaling_lut(fg.num_y_points, fg.point_y, blob + kGrainLut);
    for (int p = 0; p < 2; p++) {
        if (fg.chroma_scaling_from_luma) memcpy(blob + kGrainLut + 256 * (1 + p), blob + kGrainLut, 256);
        else scaling_lut(fg.num_uv_points[p], fg.point_uv[p], blob + kGrainLut + 256 * (1 + p));
    }
    // ---- one random byte per 32x32 luma block, the register re-seeded per 32-row stripe (7.18.3.5) ----
    const int nby = (h + 31) / 32, nbx = (w + 31) / 32;
    for (int s = 0; s < nby; s++) {
        rng.r = fg.grain_seed;
        rng.r ^= uint32_t((s * 37 + 178) & 255) << 8;
        rng.r ^= uint32_t((s * 173 + 105) & 255);
        for (int b = 0; b < nbx; b++) blob[kGrainOffsets + size_t(s) * nbx + b] = uint8_t(rng.get(8));
    }
}

// four consecutive bytes at an arbitrary byte address, as two aligned dword loads
__device__ __forceinline__ uint32_t load4_unaligned(const int8_t *p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
    return __builtin_amdgcn_alignbyte(q[1], q[0], uint32_t(a & 3));
}

// noise of the four samples (x .. x + 3, y) of a plane (SUB = 1: chroma), x a multiple of 4: closed
// form of the noise-stripe construction
template <int SUB>
__device__ __forceinline__ void grain_noise4(const int8_t *__restrict__ tmpl, const uint8_t *__restrict__ offs, int nbx, bool overlap, int x, int y,
                                             int (&out)[4])
{
    constexpr int BS = 32 >> SUB, TW = SUB ? 44 : 82, OV = 2 >> SUB;   // block size, template width, overlap width
    const int s = y / BS, i = y - s * BS, b = x / BS, j = x - b * BS;
    auto clip = [](int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); };
    auto blend = [&](int old, int cur, int k) {   // k: position inside the overlap
        const int v = SUB ? old * 23 + cur * 22 : (k == 0 ? old * 27 + cur * 17 : old * 17 + cur * 27);
        return clip((v + 16) >> 5);
    };
    // stripe `st`, row `row` (0 .. BS + OV - 1) at x .. x + 3, after the horizontal blend
    auto stripe = [&](int st, int row, int (&g)[4]) {
        const int r0 = offs[st * nbx + b];
        const int ox = SUB ? 6 + (r0 >> 4) : 9 + 2 * (r0 >> 4), oy = SUB ? 6 + (r0 & 15) : 9 + 2 * (r0 & 15);
        const uint32_t w = load4_unaligned(tmpl + (oy + row) * TW + ox + j);
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = int8_t(w >> (8 * k));
        if (overlap && b > 0 && j == 0) {
            const int r1 = offs[st * nbx + b - 1];
            const int px = SUB ? 6 + (r1 >> 4) : 9 + 2 * (r1 >> 4), py = SUB ? 6 + (r1 & 15) : 9 + 2 * (r1 & 15);
            const int8_t *q = tmpl + (py + row) * TW + px + BS;
#pragma unroll
            for (int k = 0; k < OV; k++) g[k] = blend(q[k], g[k], k);
        }
    };
    stripe(s, i, out);
    if (overlap && s > 0 && i < OV) {
        int up[4];
        stripe(s - 1, i + BS, up);
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = blend(up[k], out[k], i);
    }
}

__global__ void __launch_bounds__(256)
film_grain_kernel(const GrainView *__restrict__ views, TileGrid grid)
{
    __shared__ uint32_t lut32[192];   // the three scaling look-up tables of this picture
    const TilePos tp = xcd_tile(grid);
    if (!tp.valid) return;
    const GrainView &v = views[tp.z];
    if (!v.src[0]) return;
    const uint8_t *blob = v.blob;
    const uint8_t *offs = blob + kGrainOffsets;
    const int t = threadIdx.x;
    if (t < 192) lut32[t] = reinterpret_cast<const uint32_t *>(blob + kGrainLut)[t];
    __syncthreads();
    const uint8_t *lut_all = reinterpret_cast<const uint8_t *>(lut32);
    const bool overlap = v.flags & kGrainOverlap;
    const int rnd = 1 << (v.scaling_shift - 1);
    {   // luma: 32 rows x 8 groups of 8 samples (two 32x32 blocks side by side)
        const int x = tp.x * 64 + 8 * (t & 7), y = tp.y * 32 + (t >> 3);
        if (x < v.cw && y < v.ch) {
            const uint2 in = *reinterpret_cast<const uint2 *>(v.src[0] + size_t(y) * v.stride[0] + x);
            uint2 out = in;
            if ((v.flags & kGrainY) && x < v.w && y < v.h) {
                const int8_t *tmpl = reinterpret_cast<const int8_t *>(blob + kGrainLuma);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    int n[4];
                    grain_noise4<0>(tmpl, offs, v.nbx, overlap, x + 4 * half, y, n);
                    const uint32_t iw = half ? in.y : in.x;
                    uint32_t ow = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int orig = (iw >> (8 * k)) & 255;
                        const int val = orig + ((lut_all[orig] * n[k] + rnd) >> v.scaling_shift);
                        ow |= uint32_t(min(max(val, int(v.min_value)), int(v.max_luma))) << (8 * k);
                    }
                    const int nvis = v.w - (x + 4 * half);   // samples of this dword inside the visible picture
                    if (nvis < 4) ow = nvis <= 0 ? iw : (ow & (0xFFFFFFFFu >> (8 * (4 - nvis)))) | (iw & ~(0xFFFFFFFFu >> (8 * (4 - nvis))));
                    if (half) out.y = ow;
                    else out.x = ow;
                }
            }
            *reinterpret_cast<uint2 *>(v.dst[0] + size_t(y) * v.stride[0] + x) = out;
        }
    }
    {   // chroma: per plane 16 rows x 8 groups of 4 samples
        const int p = t >> 7, tt = t & 127;
        const int x = tp.x * 32 + 4 * (tt & 7), y = tp.y * 16 + (tt >> 3);
        const int cw = (v.w + 1) >> 1, ch = (v.h + 1) >> 1;   // visible chroma plane
        if (x < (v.cw >> 1) && y < (v.ch >> 1)) {
            const uint32_t in = *reinterpret_cast<const uint32_t *>(v.src[1 + p] + size_t(y) * v.stride[1 + p] + x);
            uint32_t out = in;
            if ((v.flags & (p ? kGrainCrOn : kGrainCbOn)) && x < cw && y < ch) {
                const uint8_t *lut = lut_all + 256 * (1 + p);
                // the un-noised luma samples 2x .. 2x + 7 of row 2y (the picture width is a multiple of 8)
                const uint2 lw = *reinterpret_cast<const uint2 *>(v.src[0] + size_t(2 * y) * v.stride[0] + 2 * x);
                int n[4];
                grain_noise4<1>(reinterpret_cast<const int8_t *>(blob + (p ? kGrainCr : kGrainCb)), offs, v.nbx, overlap, x, y, n);
                out = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int orig = (in >> (8 * k)) & 255;
                    const uint32_t pair = ((k < 2 ? lw.x : lw.y) >> (16 * (k & 1))) & 0xFFFF;
                    const int l0 = int(pair & 255), l1 = 2 * (x + k) + 1 < v.w ? int(pair >> 8) : l0;   // lumaNextX = Min(lumaX + 1, w - 1)
                    const int avg = (l0 + l1 + 1) >> 1;
                    int merged = avg;
                    if (!(v.flags & kGrainFromLuma)) {
                        const int combined = avg * v.uv_luma_mult[p] + orig * v.uv_mult[p];
                        merged = min(max((combined >> 6) + v.uv_offset[p], 0), 255);
                    }
                    const int val = orig + ((lut[merged] * n[k] + rnd) >> v.scaling_shift);
                    out |= uint32_t(min(max(val, int(v.min_value)), int(v.max_chroma))) << (8 * k);
                }
                const int nvis = cw - x;
                if (nvis < 4) out = (out & (0xFFFFFFFFu >> (8 * (4 - nvis)))) | (in & ~(0xFFFFFFFFu >> (8 * (4 - nvis))));
            }
            *reinterpret_cast<uint32_t *>(v.dst[1 + p] + size_t(y) * v.stride[1 + p] + x) = out;
        }
    }
}

void launch_film_grain(const GrainView *d_views, int nviews, int w, int h, hipStream_t st, Prof *prof, uint64_t frames)   // w, h: coded size
{
    const TileGrid grid = tile_grid((w + 63) / 64, (h + 31) / 32, nviews);   // a workgroup: 64 x 32 luma samples + 2 x (32 x 16) chroma
    // algorithmic bytes: picture read once + written once
    ProfScope ps(prof, "film_grain_kernel", frames * 3ull * uint64_t(w) * h, st);
    hipLaunchKernelGGL(film_grain_kernel, xcd_grid(grid), dim3(256), 0, st, d_views, grid);
}

}  // namespace ea1
