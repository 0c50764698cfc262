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

void grain_prepare(const ea1_film_grain &fg, const int16_t *gauss, int w, int h, int ssx, int ssy, uint8_t *blob)
{
    // ---- grain templates (7.18.3.3), 8-bit: values in [-128, 127]; a chroma template is 44 wide / 38 high where the planes are subsampled ----
    static thread_local int luma[73][82], chroma[2][73][82];
    const int ctw = ssx ? 44 : 82, cth = ssy ? 38 : 73;
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
        for (int y = 0; y < cth; y++)
            for (int x = 0; x < ctw; x++) chroma[p][y][x] = on ? rnd2(gauss[rng.get(11)], gshift) : 0;
    }
    for (int y = 3; y < cth; y++)
        for (int x = 3; x < ctw - 3; x++) {
            int lavg = 0;
            if (fg.num_y_points) {
                const int lx = ((x - 3) << ssx) + 3, ly = ((y - 3) << ssy) + 3;
                for (int i = 0; i <= ssy; i++)
                    for (int j = 0; j <= ssx; j++) lavg += luma[ly + i][lx + j];
                lavg = rnd2(lavg, ssx + ssy);
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
        for (int y = 0; y < cth; y++)
            for (int x = 0; x < ctw; x++) tc[y * ctw + x] = int8_t(chroma[p][y][x]);
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
template <int SUBX, int SUBY>
__device__ __forceinline__ void grain_noise4(const int8_t *__restrict__ tmpl, const uint8_t *__restrict__ offs, int nbx, bool overlap, int x, int y,
                                             int (&out)[4])
{
    // per direction: block size, overlap width (the weights: 27 / 17, 17 / 27 over two samples, 23 / 22 over one where the plane is subsampled)
    constexpr int BSX = 32 >> SUBX, BSY = 32 >> SUBY, TW = SUBX ? 44 : 82, OVX = 2 >> SUBX, OVY = 2 >> SUBY;
    const int s = y / BSY, i = y - s * BSY, b = x / BSX, j = x - b * BSX;
    auto clip = [](int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); };
    auto blend = [&](int sub, int old, int cur, int k) {   // k: position inside the overlap
        const int v = sub ? old * 23 + cur * 22 : (k == 0 ? old * 27 + cur * 17 : old * 17 + cur * 27);
        return clip((v + 16) >> 5);
    };
    // stripe `st`, row `row` (0 .. BSY + OVY - 1) at x .. x + 3, after the horizontal blend
    auto stripe = [&](int st, int row, int (&g)[4]) {
        const int r0 = offs[st * nbx + b];
        const int ox = SUBX ? 6 + (r0 >> 4) : 9 + 2 * (r0 >> 4), oy = SUBY ? 6 + (r0 & 15) : 9 + 2 * (r0 & 15);
        const uint32_t w = load4_unaligned(tmpl + (oy + row) * TW + ox + j);
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = int8_t(w >> (8 * k));
        if (overlap && b > 0 && j == 0) {
            const int r1 = offs[st * nbx + b - 1];
            const int px = SUBX ? 6 + (r1 >> 4) : 9 + 2 * (r1 >> 4), py = SUBY ? 6 + (r1 & 15) : 9 + 2 * (r1 & 15);
            const int8_t *q = tmpl + (py + row) * TW + px + BSX;
#pragma unroll
            for (int k = 0; k < OVX; k++) g[k] = blend(SUBX, q[k], g[k], k);
        }
    };
    stripe(s, i, out);
    if (overlap && s > 0 && i < OVY) {
        int up[4];
        stripe(s - 1, i + BSY, up);
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = blend(SUBY, up[k], out[k], i);
    }
}

// the four chroma samples (x .. x + 3, y) of plane 1 + p with their noise (7.18.3.5 add noise process); `in`: the samples before
template <int SUBX, int SUBY>
__device__ __forceinline__ uint32_t grain_chroma4(const GrainView &v, const uint8_t *lut_all, const uint8_t *offs, bool overlap, int rnd, int p, int x, int y, uint32_t in)
{
    const int cw = (v.w + SUBX) >> SUBX;   // visible chroma width
    const uint8_t *lut = lut_all + 256 * (1 + p);
    const uint8_t *lrow = v.src[0] + size_t(y << SUBY) * v.stride[0];   // the un-noised luma row the chroma row scales by
    int n[4];
    grain_noise4<SUBX, SUBY>(reinterpret_cast<const int8_t *>(v.blob + (p ? kGrainCr : kGrainCb)), offs, v.nbx, overlap, x, y, n);
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int orig = (in >> (8 * k)) & 255;
        const int lx = (x + k) << SUBX;
        const int l0 = lrow[lx < v.cw ? lx : v.cw - 1];   // (columns beyond the visible width are masked below; keep the read inside the row)
        const int l1 = SUBX ? lrow[lx + 1 < v.w ? lx + 1 : (v.w - 1 > 0 ? v.w - 1 : 0)] : l0;   // lumaNextX = Min(lumaX + 1, w - 1)
        const int avg = SUBX ? (l0 + l1 + 1) >> 1 : l0;
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
    return out;
}

// chroma planes of 4:4:4 / 4:2:2 pictures (the kernel below does the 4:2:0 ones beside the luma plane): a lane makes 4 samples of one
// row of one plane; grid.z = picture * 2 + plane
template <int SUBX>
__global__ void __launch_bounds__(256)
film_grain_chroma_kernel(const GrainView *__restrict__ views, TileGrid grid)
{
    __shared__ uint32_t lut32[192];
    const TilePos tp = xcd_tile(grid);
    if (!tp.valid) return;
    const GrainView &v = views[tp.z >> 1];
    if (!v.src[0]) return;
    const int p = tp.z & 1, t = threadIdx.x;
    if (t < 192) lut32[t] = reinterpret_cast<const uint32_t *>(v.blob + kGrainLut)[t];
    __syncthreads();
    const int x = tp.x * 64 + 4 * (t & 15), y = tp.y * 16 + (t >> 4);
    if (x >= (v.cw >> SUBX) || y >= v.ch) return;
    const uint32_t in = *reinterpret_cast<const uint32_t *>(v.src[1 + p] + size_t(y) * v.stride[1 + p] + x);
    uint32_t out = in;
    if ((v.flags & (p ? kGrainCrOn : kGrainCbOn)) && x < ((v.w + SUBX) >> SUBX) && y < v.h)
        out = grain_chroma4<SUBX, 0>(v, reinterpret_cast<const uint8_t *>(lut32), v.blob + kGrainOffsets, v.flags & kGrainOverlap, 1 << (v.scaling_shift - 1), p, x, y, in);
    *reinterpret_cast<uint32_t *>(v.dst[1 + p] + size_t(y) * v.stride[1 + p] + x) = out;
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
                    grain_noise4<0, 0>(tmpl, offs, v.nbx, overlap, x + 4 * half, y, n);
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
    if (v.ssx && v.ssy) {   // chroma (4:2:0; the other layouts: film_grain_chroma_kernel): per plane 16 rows x 8 groups of 4 samples
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
                grain_noise4<1, 1>(reinterpret_cast<const int8_t *>(blob + (p ? kGrainCr : kGrainCb)), offs, v.nbx, overlap, x, y, n);
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

void launch_film_grain(const GrainView *d_views, int nviews, int w, int h, int ssx, int ssy, hipStream_t st, Prof *prof, uint64_t frames)   // w, h: coded size
{
    const TileGrid grid = tile_grid((w + 63) / 64, (h + 31) / 32, nviews);   // a workgroup: 64 x 32 luma samples + 2 x (32 x 16) chroma
    // algorithmic bytes: picture read once + written once
    ProfScope ps(prof, "film_grain_kernel", frames * ((2ull * uint64_t(w) * h) + ((4ull * uint64_t(w) * h) >> (ssx + ssy))), st);
    hipLaunchKernelGGL(film_grain_kernel, xcd_grid(grid), dim3(256), 0, st, d_views, grid);
    if (ssx && ssy) return;
    // 4:4:4 / 4:2:2: the chroma planes on their own, a workgroup per 64 x 16 samples of one plane
    const TileGrid cgrid = tile_grid(((w >> ssx) + 63) / 64, (h + 15) / 16, 2 * nviews);
    if (ssx) hipLaunchKernelGGL(film_grain_chroma_kernel<1>, xcd_grid(cgrid), dim3(256), 0, st, d_views, cgrid);
    else hipLaunchKernelGGL(film_grain_chroma_kernel<0>, xcd_grid(cgrid), dim3(256), 0, st, d_views, cgrid);
}

}  // namespace ea1
