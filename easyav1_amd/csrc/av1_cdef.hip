// CDEF: constrained directional enhancement filter (AV1 spec 7.15; dav1d src/cdef_tmpl.c +
// cdef_apply_tmpl.c, /root/reference/cmake/dav1d.cmake:1003,:1002).
//
// Two launches:
//   cdef_dir_kernel     one lane per 8x8 luma block: direction + variance class (spec 7.15.2),
//                       2 bytes per block -> the only inter-kernel state.
//   cdef_filter_kernel  a 256-lane workgroup filters a 64x16 tile of one plane: the tile plus
//                       its 2-pixel halo (deblocked samples) is staged in LDS once; every lane
//                       produces 4 adjacent pixels and stores them with one 4-byte store.
// CDEF reads only deblocked samples and writes a separate picture (the unfiltered neighbours must
// stay visible), so traffic is read plane + write plane.  The filter is min/max/shift logic on
// 12 taps: VALU bound on LDS bytes, no contraction to speak of.
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const int8_t kCdefDir[8][2][2] = {EA1_CDEF_DIRECTIONS_VALUES};

__device__ __forceinline__ int cdef_constrain(int diff, int threshold, int shift)
{
    // shift = max(0, damping - floor_log2(threshold)); threshold != 0
    const int mag = diff < 0 ? -diff : diff;
    int v = threshold - (mag >> shift);
    v = v < 0 ? 0 : (v > mag ? mag : v);
    return diff < 0 ? -v : v;
}

__device__ __forceinline__ int flog2(int v) { return 31 - __clz(v); }

__global__ void __launch_bounds__(64)
cdef_dir_kernel(Batch bt, int in_second, int ux_first)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const uint8_t *__restrict__ luma = in_second ? fv.cdef.p[0] : fv.work.p[0];
    const int stride = fv.work.stride[0], w8 = fv.w8, sb_cols = fv.sb_cols;
    uint8_t *__restrict__ dir_var = fv.dir_var;
    const int ux = ux_first + blockIdx.x * 64 + threadIdx.x, uy = blockIdx.y;
    if (ux >= w8) return;
    const int u = uy * w8 + ux;
    const int idx = fv.desc.cdef_enable ? fv.cdef_idx[(uy >> 3) * sb_cols + (ux >> 3)] : 0xFF;
    if (idx == 0xFF || (fv.unit_flags[u] & 1)) {
        dir_var[2 * u] = 0;  // bit 7 clear: block is not filtered
        dir_var[2 * u + 1] = 0;
        return;
    }
    int p0[15], p1[11], p2[8], p3[11], p4[15], p5[11], p6[8], p7[11];
#pragma unroll
    for (int i = 0; i < 15; i++) p0[i] = p4[i] = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) p1[i] = p3[i] = p5[i] = p7[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) p2[i] = p6[i] = 0;
    const uint8_t *src = luma + size_t(uy * 8) * stride + ux * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint2 w = *reinterpret_cast<const uint2 *>(src + size_t(i) * stride);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x = int(((j < 4 ? w.x : w.y) >> (8 * (j & 3))) & 255) - 128;
            p0[i + j] += x;
            p1[i + j / 2] += x;
            p2[i] += x;
            p3[3 + i - j / 2] += x;
            p4[7 + i - j] += x;
            p5[3 - i / 2 + j] += x;
            p6[j] += x;
            p7[i / 2 + j] += x;
        }
    }
    int cost[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        cost[2] += p2[i] * p2[i];
        cost[6] += p6[i] * p6[i];
    }
    cost[2] *= 105;
    cost[6] *= 105;
    constexpr int div[9] = {EA1_CDEF_DIV_TABLE_VALUES};
#pragma unroll
    for (int i = 0; i < 7; i++) {
        cost[0] += (p0[i] * p0[i] + p0[14 - i] * p0[14 - i]) * div[i + 1];
        cost[4] += (p4[i] * p4[i] + p4[14 - i] * p4[14 - i]) * div[i + 1];
    }
    cost[0] += p0[7] * p0[7] * 105;
    cost[4] += p4[7] * p4[7] * 105;
#define EA1_ODD(c, P)                                                                   \
    {                                                                                   \
        int s = 0;                                                                      \
        _Pragma("unroll") for (int j = 0; j < 5; j++) s += P[3 + j] * P[3 + j];         \
        s *= 105;                                                                       \
        _Pragma("unroll") for (int j = 0; j < 3; j++) s += (P[j] * P[j] + P[10 - j] * P[10 - j]) * div[2 * j + 2]; \
        cost[c] = s;                                                                    \
    }
    EA1_ODD(1, p1) EA1_ODD(3, p3) EA1_ODD(5, p5) EA1_ODD(7, p7)
#undef EA1_ODD
    int best = 0, bd = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (cost[k] > best) {
            best = cost[k];
            bd = k;
        }
    int opp = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k == ((bd + 4) & 7)) opp = cost[k];
    const int var = (best - opp) >> 10;
    int cls = 0;  // 0: var == 0; else 1 + min(floor_log2(var >> 6), 12) (or 1 when var >> 6 == 0)
    if (var) {
        const int v6 = var >> 6;
        int vs = v6 ? flog2(v6) : 0;
        if (vs > 12) vs = 12;
        cls = 1 + vs;
    }
    dir_var[2 * u] = uint8_t(0x80 | bd);
    dir_var[2 * u + 1] = uint8_t(cls);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 splat(int v) { return s16x2{short(v), short(v)}; }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// constrain(p - x) on two samples at once (packed 16-bit VALU).  sign(d) * min(|d|, t) with
// t = max(0, thr - (|d| >> shift)) is clamp(d, -t, t): no sign extraction, and the max(0, .) is the
// saturation of an unsigned packed subtract (v_pk_sub_u16 clamp).
__device__ __forceinline__ s16x2 constrain2(s16x2 p, s16x2 x, s16x2 thr, s16x2 shift)
{
    const s16x2 d = p - x;
    const s16x2 mag = pmax(d, x - p);
    const u16x2 sh = __builtin_bit_cast(u16x2, mag) >> __builtin_bit_cast(u16x2, shift);
    const s16x2 t = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, thr), sh));
    return pmin(pmax(d, -t), t);
}

// UNIT = 8 (luma) or 4 (chroma): size of one CDEF block in this plane.
// Tile 64x16 + 2-sample halo staged as aligned dwords (LDS row = picture columns tx0-4 .. tx0+67).
// The 4 samples of a lane are two packed 16-bit pairs; a tap is two aligned LDS dwords + v_alignbyte
// + two v_perm unpacks.  Tiles that touch the picture border additionally replace taps outside the
// picture by the centre sample (= the spec's "tap not available", 7.15.2).
template <int UNIT>
__global__ void __launch_bounds__(256)
cdef_filter_kernel(Batch bt, int nplanes, int in_second, TileGrid grid)
{
    constexpr int TW = 64, TH = 16, LW = TW + 8, LWD = LW / 4;  // LDS row stride 72 bytes
    __shared__ uint32_t tile32[(TH + 4) * LWD];
    const TilePos tp = xcd_tile(grid);
    if (!tp.valid) return;
    const FrameView *fvp = bt.v[tp.z / nplanes];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    constexpr int is_chroma = UNIT == 4;
    const int pl = is_chroma ? 1 + tp.z % nplanes : 0;
    // input = deblocked picture (second buffer when the deblock stage ran), output = the other one
    const uint8_t *__restrict__ in = in_second ? fv.cdef.p[pl] : fv.work.p[pl];
    uint8_t *__restrict__ out = in_second ? fv.work.p[pl] : fv.cdef.p[pl];
    const int stride = fv.work.stride[pl], W = fv.work.pw(pl), H = fv.work.ph(pl), w8 = fv.w8, sb_cols = fv.sb_cols;
    const uint8_t *__restrict__ cdef_idx = fv.cdef_idx;
    const uint8_t *__restrict__ dir_var = fv.dir_var;
    const ea1_frame_desc &desc = fv.desc;
    const int tx0 = grid.x_origin + tp.x * TW, ty0 = tp.y * TH;   // x_origin: a multiple of 4 (aligned dword rows)
    if (tx0 >= W || ty0 >= H) return;
    const int tid = threadIdx.x;
    const bool interior = tx0 >= 4 && tx0 + TW + 4 <= W && ty0 >= 2 && ty0 + TH + 2 <= H;
    if (interior) {
        for (int i = tid; i < (TH + 4) * LWD; i += 256) {
            const int r = i / LWD, k = i - r * LWD;
            tile32[i] = *reinterpret_cast<const uint32_t *>(in + size_t(ty0 - 2 + r) * stride + tx0 - 4 + 4 * k);
        }
    } else {
        uint8_t *tb = reinterpret_cast<uint8_t *>(tile32);
        for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
            const int r = i / (TW + 4), c = i - r * (TW + 4);
            int y = ty0 - 2 + r, x = tx0 - 2 + c;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y);
            x = x < 0 ? 0 : (x >= W ? W - 1 : x);
            tb[r * LW + c + 2] = in[size_t(y) * stride + x];
        }
    }
    __syncthreads();
    // lane -> 4 samples.  A wave covers a 16x16 square (2x2 CDEF blocks for luma) rather than 4 full
    // rows of the tile: the per-block branches (skipped block, zero strengths, direction) then
    // diverge over 4 blocks per wave instead of 8.
    const int wv = tid >> 6, lw = tid & 63;
    const int ly = lw >> 2, lx = wv * 16 + (lw & 3) * 4;
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= H || x >= W) return;
    const int ux = x / UNIT, uy = y / UNIT;
    const int u = uy * w8 + ux;
    const uint8_t dv = dir_var[2 * u];
    const int cbase = (ly + 2) * LW + lx + 4;  // byte offset of the lane's first sample (4-aligned)
    const uint32_t centre = tile32[cbase >> 2];
    uint32_t result = centre;
    if (dv & 0x80) {
        const int idx = cdef_idx[(uy >> 3) * sb_cols + (ux >> 3)];  // units share the luma 8x8 grid
        const int str = is_chroma ? desc.cdef_uv_strength[idx] : desc.cdef_y_strength[idx];
        int pri = str >> 2, sec = str & 3;
        if (sec == 3) sec = 4;
        const int dir = pri ? (dv & 7) : 0;
        int damping = desc.cdef_damping;
        if (is_chroma) {
            damping -= 1;
        } else {
            const int cls = dir_var[2 * u + 1];
            pri = cls ? (pri * (4 + cls - 1) + 8) >> 4 : 0;
        }
        const int pri_shift = pri ? max(0, damping - flog2(pri)) : 0;
        const int sec_shift = sec ? max(0, damping - flog2(sec)) : 0;
        const int pt0 = (pri & 1) ? 3 : 4, pt1 = (pri & 1) ? 3 : 2;
        const int d1 = (dir + 2) & 7, d2 = (dir + 6) & 7;
        if (pri | sec) {
            const bool both = pri && sec;
            const s16x2 x01 = as_s16x2(__builtin_amdgcn_perm(0, centre, 0x0c010c00));
            const s16x2 x23 = as_s16x2(__builtin_amdgcn_perm(0, centre, 0x0c030c02));
            s16x2 sum01 = splat(0), sum23 = splat(0), mx01 = x01, mx23 = x23, mn01 = x01, mn23 = x23;
#pragma unroll
            for (int which = 0; which < 3; which++) {
                const int strength = which == 0 ? pri : sec;
                if (!strength) continue;
                const int dd = which == 0 ? dir : (which == 1 ? d2 : d1);
                const s16x2 thr = splat(strength), sh = splat(which == 0 ? pri_shift : sec_shift);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int w = which == 0 ? (k ? pt1 : pt0) : (k ? 1 : 2);
                    const int dy = kCdefDir[dd][k][0], dx = kCdefDir[dd][k][1];
#pragma unroll
                    for (int sgn = -1; sgn <= 1; sgn += 2) {
                        const int off = cbase + sgn * (dy * LW + dx);
                        const int q = off >> 2;
                        uint32_t tap = __builtin_amdgcn_alignbyte(tile32[q + 1], tile32[q], off & 3);
                        if (!interior) {
                            // spec 7.15.2 CdefAvailable: a tap outside the picture is skipped.  Replacing
                            // it by the centre sample is the same thing: difference 0, min/max unchanged.
                            const int yy = y + sgn * dy, xx = x + sgn * dx;   // first of the 4 samples
                            uint32_t keep = 0;
                            if (yy >= 0 && yy < H) {
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                    if (xx + j >= 0 && xx + j < W) keep |= 0xFFu << (8 * j);
                            }
                            tap = (tap & keep) | (centre & ~keep);
                        }
                        const s16x2 p01 = as_s16x2(__builtin_amdgcn_perm(0, tap, 0x0c010c00));
                        const s16x2 p23 = as_s16x2(__builtin_amdgcn_perm(0, tap, 0x0c030c02));
                        sum01 += splat(w) * constrain2(p01, x01, thr, sh);
                        sum23 += splat(w) * constrain2(p23, x23, thr, sh);
                        if (both) {
                            mx01 = pmax(mx01, p01);
                            mx23 = pmax(mx23, p23);
                            mn01 = pmin(mn01, p01);
                            mn23 = pmin(mn23, p23);
                        }
                    }
                }
            }
            s16x2 v01 = x01 + ((splat(8) + sum01 + (sum01 >> splat(15))) >> splat(4));
            s16x2 v23 = x23 + ((splat(8) + sum23 + (sum23 >> splat(15))) >> splat(4));
            if (both) {  // with a single tap class the result cannot leave [min, max]: no clip needed
                v01 = pmin(pmax(v01, mn01), mx01);
                v23 = pmin(pmax(v23, mn23), mx23);
            }
            result = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, v23), __builtin_bit_cast(uint32_t, v01), 0x06040200);
        }
    }
    *reinterpret_cast<uint32_t *>(out + size_t(y) * stride + x) = result;
}

void launch_cdef(const SubmitInfo &s, hipStream_t st, Prof *prof, int in_second)
{
    const int w8 = s.width / 8, h8 = s.height / 8, n = s.batch.n;
    uint64_t frames = 0;
    for (int c = 0; c < n; c++)
        if (s.stats[c]) frames++;
    if (!frames) return;
    const uint64_t ybytes = uint64_t(s.width) * s.height * uint64_t(s.win_x1 - s.win_x0) / uint64_t(s.width);   // of the window
    const int u0 = s.win_x0 / 8, u1 = (s.win_x1 + 7) / 8;
    (void)w8;
    {
        ProfScope ps(prof, "cdef_dir_kernel", frames * (ybytes + 2ull * (u1 - u0) * h8), st);
        hipLaunchKernelGGL(cdef_dir_kernel, dim3((u1 - u0 + 63) / 64, h8, n), dim3(64), 0, st, s.batch, in_second, u0);
    }
    {
        ProfScope ps(prof, "cdef_filter_kernel<8>", frames * 2 * ybytes, st);
        const TileGrid grid = tile_grid((s.win_x1 - s.win_x0 + 63) / 64, (s.height + 15) / 16, n, s.win_x0);
        hipLaunchKernelGGL((cdef_filter_kernel<8>), xcd_grid(grid), dim3(256), 0, st, s.batch, 1, in_second, grid);
    }
    ProfScope ps(prof, "cdef_filter_kernel<4>", frames * ybytes, st);
    const TileGrid grid = tile_grid(((s.win_x1 - s.win_x0) / 2 + 63) / 64, (s.height / 2 + 15) / 16, 2 * n, s.win_x0 / 2);
    hipLaunchKernelGGL((cdef_filter_kernel<4>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
}

}  // namespace ea1
