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
// CSX, CSY (UNIT = 4 only): subsampling of the chroma planes -- a CDEF block of theirs is (8 >> CSX) x (8 >> CSY) samples
template <int UNIT, int CSX = 1, int CSY = 1>
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
    // the 8x8 luma unit of the sample: a CDEF block of a chroma plane is (8 >> ssx) x (8 >> ssy) samples (UNIT = 4: the 4:2:0 name)
    const int ux = is_chroma ? x >> (3 - CSX) : x / UNIT, uy = is_chroma ? y >> (3 - CSY) : y / UNIT;
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
        // 7.15.1: chroma planes subsampled in one direction only (4:2:2) translate the luma direction (Cdef_Uv_Dir)
        const int dir = pri ? ((is_chroma && CSX && !CSY) ? int((0x66654207u >> (4 * (dv & 7))) & 7u) : (dv & 7)) : 0;
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

// ---- direction-uniform waves (round 3) --------------------------------------------------------------------------------------
// The kernel above pays for per-lane directions: tap offsets come out of a table per lane, every LDS address is computed, every
// alignment is a run-time operand.  Here a workgroup stages a 64x64 tile of the plane, sorts the tile's CDEF blocks by direction in
// LDS, and every wave then takes groups of blocks of ONE direction through code specialised for it: the 12 tap offsets are
// immediates of the LDS reads, the byte alignment of every tap is known at compile time (a quarter of the taps need no alignbyte at
// all), no table, no address arithmetic.  Blocks that are not filtered are copied.  Strengths stay per lane (a luma block's primary
// strength depends on its variance class; a chroma tile spans four superblocks).
//   UNIT 8: 64 blocks per tile, 16 lanes per block (4 samples each), 4 blocks per wave step
//   UNIT 4: 256 blocks per tile, 4 lanes per block, 16 blocks per wave step
__host__ __device__ constexpr bool is_luma_unit(int unit) { return unit == 8; }
struct CdefLaneParams {
    s16x2 pthr, psh, sthr, ssh, pw0, pw1;
};

template <int UNIT, int DIR, bool INTERIOR, int LW>
__device__ __forceinline__ uint32_t cdef_taps(const uint32_t *tile32, int cb4, uint32_t centre, const CdefLaneParams &lp, bool do_pri, bool do_sec, int x, int y, int W, int H)
{
    constexpr int8_t kDir[8][2][2] = {EA1_CDEF_DIRECTIONS_VALUES};
    const s16x2 x01 = as_s16x2(__builtin_amdgcn_perm(0, centre, 0x0c010c00));
    const s16x2 x23 = as_s16x2(__builtin_amdgcn_perm(0, centre, 0x0c030c02));
    s16x2 sum01 = splat(0), sum23 = splat(0), mx01 = x01, mx23 = x23, mn01 = x01, mn23 = x23;
    const bool both = do_pri && do_sec;   // wave-uniform
#pragma unroll
    for (int which = 0; which < 3; which++) {
        if (which == 0 ? !do_pri : !do_sec) continue;   // wave-uniform
        const int dd = which == 0 ? DIR : (which == 1 ? (DIR + 6) & 7 : (DIR + 2) & 7);
        const s16x2 thr = which == 0 ? lp.pthr : lp.sthr, sh = which == 0 ? lp.psh : lp.ssh;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const s16x2 w = which == 0 ? (k ? lp.pw1 : lp.pw0) : splat(k ? 1 : 2);
            const int dy = kDir[dd][k][0], dx = kDir[dd][k][1];
#pragma unroll
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                const int K = sgn * (dy * LW + dx);          // a constant after unrolling
                const int r = ((K % 4) + 4) % 4, a4 = (K - r) / 4;
                uint32_t tap = tile32[cb4 + a4];
                if (r) tap = __builtin_amdgcn_alignbyte(tile32[cb4 + a4 + 1], tap, r);
                if (!INTERIOR) {   // (a template argument: the interior variant is straight-line code, its LDS reads can all be in flight)
                    // spec 7.15.2 CdefAvailable: a tap outside the picture is skipped = replaced by the centre sample
                    const int yy = y + sgn * dy, xx = x + sgn * dx;
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
                sum01 += w * constrain2(p01, x01, thr, sh);
                sum23 += w * constrain2(p23, x23, thr, sh);
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
    // the range of the taps that were looked at: leaving out a class whose strength is zero narrows it, but the result of the other
    // class alone cannot leave the range of its own taps (every term moves x towards a tap by at most the distance to it) -- so a wave
    // step that runs one class only (no lane of it has both strengths) tracks no range at all
    if (both) {
        v01 = pmin(pmax(v01, mn01), mx01);
        v23 = pmin(pmax(v23, mn23), mx23);
    }
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, v23), __builtin_bit_cast(uint32_t, v01), 0x06040200);
}

template <int UNIT, int TW>
__global__ void __launch_bounds__(256)
cdef_sorted_kernel(Batch bt, int nplanes, int in_second, TileGrid grid)
{
    constexpr int TH = 64, LW = TW + 8, LWD = LW / 4;            // LDS row: picture columns tx0 - 4 .. tx0 + TW + 3
    constexpr int BPR = TW / UNIT, NB = BPR * (TH / UNIT);       // blocks per tile row / per tile
    constexpr int LPB = UNIT * UNIT / 4, BPW = 64 / LPB;          // lanes per block, blocks per wave step
    __shared__ uint32_t tile32[(TH + 4) * LWD];
    __shared__ uint16_t list[8][NB];     // blocks of the tile by direction: by << 8 | bx
    __shared__ uint16_t plain[NB];       // blocks that are copied
    __shared__ uint32_t count[9];        // per direction, [8] = plain
    __shared__ uint32_t count_r[8];      // per direction: the blocks of the tile's right half (they fill the list from its end)
    const TilePos tp = xcd_tile(grid);
    if (!tp.valid) return;
    const FrameView *fvp = bt.v[tp.z / nplanes];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    constexpr int is_chroma = UNIT == 4;
    const int pl = is_chroma ? 1 + tp.z % nplanes : 0;
    const uint8_t *__restrict__ in = in_second ? fv.cdef.p[pl] : fv.work.p[pl];
    uint8_t *__restrict__ out = in_second ? fv.work.p[pl] : fv.cdef.p[pl];
    const int stride = fv.work.stride[pl], W = fv.work.pw(pl), H = fv.work.ph(pl), w8 = fv.w8, sb_cols = fv.sb_cols;
    const uint8_t *__restrict__ cdef_idx = fv.cdef_idx;
    const uint8_t *__restrict__ dir_var = fv.dir_var;
    const ea1_frame_desc &desc = fv.desc;
    const int tx0 = grid.x_origin + tp.x * TW, ty0 = tp.y * TH;
    if (tx0 >= W || ty0 >= H) return;
    const int tid = threadIdx.x;
    if (tid < 9) count[tid] = 0;
    if (tid < 8) count_r[tid] = 0;
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
    // ---- the tile's blocks into the lists.  Order inside a list does not matter for the result; for speed the blocks of the tile's
    // left half come first and those of its right half last: the strengths are one pair per 64x64 superblock, so a wave step whose
    // blocks share a superblock runs -- or leaves out -- the primary and the secondary taps as a whole (at most one mixed step per list)
    constexpr int kHalf = (is_luma_unit(UNIT) ? 64 : 32) / UNIT;   // blocks per superblock row in this plane
    for (int b = tid; b < NB; b += 256) {
        const int bx = b % BPR, by = b / BPR;
        const int x = tx0 + bx * UNIT, y = ty0 + by * UNIT;
        if (x >= W || y >= H) continue;
        const int u = (y / UNIT) * w8 + x / UNIT;
        const uint8_t dv = dir_var[2 * u];
        int bucket = 8;
        if (dv & 0x80) {
            const int idx = cdef_idx[((y / UNIT) >> 3) * sb_cols + ((x / UNIT) >> 3)];
            const int str = is_chroma ? desc.cdef_uv_strength[idx] : desc.cdef_y_strength[idx];
            if (str) bucket = (str >> 2) ? (dv & 7) : 0;   // direction 0 when the primary strength is 0 (7.15.1)
        }
        if (bucket < 8 && bx >= kHalf) {
            const uint32_t at = atomicAdd(&count_r[bucket], 1u);
            list[bucket][NB - 1 - at] = uint16_t(by << 8 | bx);
        } else {
            const uint32_t at = atomicAdd(&count[bucket], 1u);
            (bucket == 8 ? plain : list[bucket])[at] = uint16_t(by << 8 | bx);
        }
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    const int sub = lane / LPB, ll = lane % LPB;                     // block of the wave step, lane inside the block
    const int row = is_chroma ? ll : ll >> 1, col = is_chroma ? 0 : (ll & 1) * 4;
    // ---- filtered blocks: wave steps of BPW blocks of one direction, dealt out to the waves in turn ----
    int step = 0;
#pragma unroll 1
    for (int dir = 0; dir < 8; dir++) {
        const int nl = int(count[dir]), n = nl + int(count_r[dir]);
#pragma unroll 1
        for (int g = 0; g * BPW < n; g++, step++) {
            if ((step & 3) != wv) continue;
            const int bi = g * BPW + sub;
            if (bi >= n) continue;                                   // idle lanes of a last, partial step
            const int code = list[dir][bi < nl ? bi : NB - 1 - (bi - nl)];
            const int bx = code & 255, by = code >> 8;
            const int lx = bx * UNIT + col, ly = by * UNIT + row;
            const int x = tx0 + lx, y = ty0 + ly;
            const int ux = x / UNIT, uy = y / UNIT, u = uy * w8 + ux;
            const int idx = cdef_idx[(uy >> 3) * sb_cols + (ux >> 3)];
            const int str = is_chroma ? desc.cdef_uv_strength[idx] : desc.cdef_y_strength[idx];
            int pri = str >> 2, sec = str & 3;
            if (sec == 3) sec = 4;
            int damping = desc.cdef_damping;
            if (is_chroma) {
                damping -= 1;
            } else {
                const int cls = dir_var[2 * u + 1];
                pri = cls ? (pri * (4 + cls - 1) + 8) >> 4 : 0;
            }
            CdefLaneParams lp;
            lp.pthr = splat(pri);
            lp.psh = splat(pri ? max(0, damping - flog2(pri)) : 0);
            lp.sthr = splat(sec);
            lp.ssh = splat(sec ? max(0, damping - flog2(sec)) : 0);
            lp.pw0 = splat((pri & 1) ? 3 : 4);
            lp.pw1 = splat((pri & 1) ? 3 : 2);
            const bool do_pri = __any(pri != 0), do_sec = __any(sec != 0);
            const int cb4 = ((ly + 2) * LW + lx + 4) >> 2;
            const uint32_t centre = tile32[cb4];
            uint32_t result;
#define EA1_CDEF_CASE(D, IN) case D: result = cdef_taps<UNIT, D, IN, LW>(tile32, cb4, centre, lp, do_pri, do_sec, x, y, W, H); break;
            if (interior) {   // workgroup-uniform
                switch (dir) {   // wave-uniform
                    EA1_CDEF_CASE(0, true) EA1_CDEF_CASE(1, true) EA1_CDEF_CASE(2, true) EA1_CDEF_CASE(3, true)
                    EA1_CDEF_CASE(4, true) EA1_CDEF_CASE(5, true) EA1_CDEF_CASE(6, true)
                default: result = cdef_taps<UNIT, 7, true, LW>(tile32, cb4, centre, lp, do_pri, do_sec, x, y, W, H); break;
                }
            } else {
                switch (dir) {
                    EA1_CDEF_CASE(0, false) EA1_CDEF_CASE(1, false) EA1_CDEF_CASE(2, false) EA1_CDEF_CASE(3, false)
                    EA1_CDEF_CASE(4, false) EA1_CDEF_CASE(5, false) EA1_CDEF_CASE(6, false)
                default: result = cdef_taps<UNIT, 7, false, LW>(tile32, cb4, centre, lp, do_pri, do_sec, x, y, W, H); break;
                }
            }
#undef EA1_CDEF_CASE
            *reinterpret_cast<uint32_t *>(out + size_t(y) * stride + x) = result;
        }
    }
    // ---- the other blocks: copied ----
    const int np = int(count[8]);
    for (int i = tid; i < np * LPB; i += 256) {
        const int code = plain[i / LPB], l2 = i % LPB;
        const int bx = code & 255, by = code >> 8;
        const int lx = bx * UNIT + (is_chroma ? 0 : (l2 & 1) * 4), ly = by * UNIT + (is_chroma ? l2 : l2 >> 1);
        *reinterpret_cast<uint32_t *>(out + size_t(ty0 + ly) * stride + tx0 + lx) = tile32[((ly + 2) * LW + lx + 4) >> 2];
    }
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
    static const bool v1_env = getenv("EA1_CDEF_V1") != nullptr;   // the round-2 kernel (per-lane directions), for before / after measurements (4:2:0)
    if (v1_env && s.ssx && s.ssy) {
        {
            ProfScope ps(prof, "cdef_filter_kernel<8>", frames * 2 * ybytes, st);
            const TileGrid grid = tile_grid((s.win_x1 - s.win_x0 + 63) / 64, (s.height + 15) / 16, n, s.win_x0);
            hipLaunchKernelGGL((cdef_filter_kernel<8>), xcd_grid(grid), dim3(256), 0, st, s.batch, 1, in_second, grid);
        }
        ProfScope ps(prof, "cdef_filter_kernel<4>", (frames * 4 * ybytes) >> (s.ssx + s.ssy), st);
        const TileGrid grid = tile_grid((((s.win_x1 - s.win_x0) >> s.ssx) + 63) / 64, ((s.height >> s.ssy) + 15) / 16, 2 * n, s.win_x0 >> s.ssx);
        hipLaunchKernelGGL((cdef_filter_kernel<4>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
        return;
    }
    static const int tw = getenv("EA1_CDEF_TW") ? atoi(getenv("EA1_CDEF_TW")) : 128;
    static const bool chroma_sorted_env = getenv("EA1_CDEF_CHROMA_SORTED") != nullptr;
    const bool chroma_sorted = chroma_sorted_env && s.ssx && s.ssy;   // (the sorted chroma variant knows 4x4 blocks only)
    const int cwin = (s.win_x1 - s.win_x0) >> s.ssx, cheight = s.height >> s.ssy, cx0 = s.win_x0 >> s.ssx;
    {
        ProfScope ps(prof, tw == 128 ? "cdef_sorted_kernel<8,128>" : "cdef_sorted_kernel<8,64>", frames * 2 * ybytes, st);
        const TileGrid grid = tile_grid((s.win_x1 - s.win_x0 + tw - 1) / tw, (s.height + 63) / 64, n, s.win_x0);
        if (tw == 128) hipLaunchKernelGGL((cdef_sorted_kernel<8, 128>), xcd_grid(grid), dim3(256), 0, st, s.batch, 1, in_second, grid);
        else hipLaunchKernelGGL((cdef_sorted_kernel<8, 64>), xcd_grid(grid), dim3(256), 0, st, s.batch, 1, in_second, grid);
    }
    // chroma: the per-lane-direction kernel stays (sorted: 510 us against 494 per 32 4K frames -- a 4x4 unit is 4 lanes, a wave step 16
    // units, and a quarter of the steps run part-empty)
    if (chroma_sorted) {
        ProfScope ps(prof, "cdef_sorted_kernel<4,64>", frames * ybytes, st);
        const TileGrid grid = tile_grid(((s.win_x1 - s.win_x0) / 2 + 63) / 64, (s.height / 2 + 63) / 64, 2 * n, s.win_x0 / 2);
        hipLaunchKernelGGL((cdef_sorted_kernel<4, 64>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
    } else {
        ProfScope ps(prof, s.ssx && s.ssy ? "cdef_filter_kernel<4,1,1>" : (s.ssx ? "cdef_filter_kernel<4,1,0>" : "cdef_filter_kernel<4,0,0>"), (frames * 4 * ybytes) >> (s.ssx + s.ssy), st);
        const TileGrid grid = tile_grid((cwin + 63) / 64, (cheight + 15) / 16, 2 * n, cx0);
        if (s.ssx && s.ssy) hipLaunchKernelGGL((cdef_filter_kernel<4, 1, 1>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
        else if (s.ssx) hipLaunchKernelGGL((cdef_filter_kernel<4, 1, 0>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
        else hipLaunchKernelGGL((cdef_filter_kernel<4, 0, 0>), xcd_grid(grid), dim3(256), 0, st, s.batch, 2, in_second, grid);
    }
}

}  // namespace ea1
