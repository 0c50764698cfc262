// Intra prediction for the intra wavefront (dav1d src/ipred_tmpl.c + recon_tmpl.c,
// /root/reference/cmake/dav1d.cmake:1007,:1014; AV1 spec 7.11.2).
//
// Intra blocks read reconstructed neighbours (left column, top row, top-left) of the CURRENT
// frame, so they form a dependency wavefront; inter blocks do not and are finished before it.
// Only the PREDICTION is serial: the residual of every intra block was already inverse-transformed
// into the chain's int16 residual plane by the parallel itx kernels, so a wavefront step is just
// "edges -> predict -> add residual -> clip".  The host sorts intra (block, plane) items by
// dependency level.
//
//   level 0 (no intra dependency; by far the widest level of an inter frame) : one plain launch,
//            one 64-lane workgroup per item, all chains.
//   levels 1.. (a key frame has hundreds of them, each a few dozen items wide): ONE launch with one
//            1024-lane workgroup per chain.  Its 16 waves share the items of a level and meet at a
//            workgroup barrier between levels: the whole wavefront of a chain stays on one CU, so
//            its loads see its own earlier stores through that CU's L1/L2 path with plain
//            workgroup-scope ordering -- no agent-scope fences, no L2 write-backs, no polling.
//            (A multi-CU version with per-chain arrival counters was measured 3x SLOWER: every
//            release fence is an L2 write-back behind megabytes of dirty lines left by the
//            prediction/transform kernels.)  Chains never wait for each other.
// Edges live in registers (lane l holds above[l] / left[l]) and are exchanged with wave shuffles,
// so an item needs no LDS and no barrier.
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const uint8_t kSmWeights[128] = {EA1_SM_WEIGHTS_VALUES};

__device__ __constant__ static const uint16_t kDrIntra[88] = {EA1_DR_INTRA_DERIVATIVE_VALUES};
__device__ __constant__ static const uint8_t kEdgeKernel[3][5] = {EA1_INTRA_EDGE_KERNEL_VALUES};
__device__ __constant__ static const int8_t kFilterTaps[5][8][7] = {EA1_FILTER_INTRA_TAPS_VALUES};
__device__ __constant__ static const uint8_t kModeAngle[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};
__device__ __constant__ static const uint8_t kIiWeights[128] = {EA1_II_WEIGHTS_1D_VALUES};

constexpr int kWaveWaves = 16;  // waves of the per-chain wavefront workgroup

// Per-wave LDS scratch of the predictors that need more than the edge registers:
//   directional modes : the two edges as int16 arrays (index -16 .. 159) for filtering / upsampling
//   filter intra      : the growing (h + 1) x (w + 1) prediction, bytes
constexpr int kEdgeN = 176, kEdgeOrg = 16;
constexpr int kFiStride = 36;
constexpr int kWaveLds = (2 * kEdgeN * 2 + 33 * kFiStride + 15) & ~15;   // bytes per wave

// orders this wave's LDS writes before its later LDS reads (lanes of a wave run in lock step; the
// fences keep the compiler from moving the accesses)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// an IntraItem held as the four dwords it was loaded as (keeps the prefetched items of the wave
// kernel in registers; a struct of bytes ends up in scratch)
struct Item {
    uint4 r;
    __device__ __forceinline__ int x() const { return r.x & 0xFFFF; }
    __device__ __forceinline__ int y() const { return r.x >> 16; }
    __device__ __forceinline__ int plane() const { return r.y & 0xFF; }
    __device__ __forceinline__ int wl2() const { return (r.y >> 8) & 15; }
    __device__ __forceinline__ int hl2() const { return (r.y >> 12) & 15; }
    __device__ __forceinline__ int mode() const { return (r.y >> 16) & 0xFF; }
    __device__ __forceinline__ int flags() const { return r.y >> 24; }
    __device__ __forceinline__ int angle_delta() const { return int8_t(r.z & 0xFF); }
    __device__ __forceinline__ int alpha() const { return int8_t((r.z >> 8) & 0xFF); }
    __device__ __forceinline__ int part() const { return (r.z >> 16) & 0xFF; }
    __device__ __forceinline__ int ii() const { return r.z >> 24; }
    __device__ __forceinline__ uint32_t aux() const { return r.w; }
};
__device__ __forceinline__ Item load_item(const IntraItem *p)
{
    Item it;
    it.r = *reinterpret_cast<const uint4 *>(p);
    return it;
}

struct Edges {
    int A, L, tl;  // lane l < w: above[l]; lane l < h: left[l]; tl: top-left (uniform)
};

// issue the edge loads of one item (nothing is consumed here, so several items can be in flight)
__device__ __forceinline__ Edges intra_load(const FrameView &fv, const Item it, int lane)
{
    const int pl = it.plane();
    const int w = 1 << it.wl2(), h = 1 << it.hl2();
    const bool have_left = it.flags() & kIIHaveLeft, have_top = it.flags() & kIIHaveTop;
    const int stride = fv.work.stride[pl];
    const uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    Edges e;
    e.A = 0;
    e.L = 0;
    // (AV1 spec 7.11.2 maxX / maxY: a block may reach beyond the TRUE coded plane, ea1_frame_desc.pad -- its edges repeat the last coded
    // sample out there; fv.work.pw / ph are the true coded size)
    const int max_x = fv.work.pw(pl) - 1, max_y = fv.work.ph(pl) - 1;
    if (lane < w) e.A = have_top ? f[size_t(y - 1) * stride + min(x + lane, max_x)] : (have_left ? f[size_t(y) * stride + x - 1] : 127);
    if (lane < h) e.L = have_left ? f[size_t(min(y + lane, max_y)) * stride + x - 1] : (have_top ? f[size_t(y - 1) * stride + x] : 129);
    if (have_top && have_left) e.tl = f[size_t(y - 1) * stride + x - 1];
    else if (have_top) e.tl = f[size_t(y - 1) * stride + x];
    else if (have_left) e.tl = f[size_t(y) * stride + x - 1];
    else e.tl = 128;
    return e;
}

// predict + add residual + store, one item by one wave; `lane` = lane id inside the wave.
// A lane owns 4 adjacent samples of a row (one dword store, one 8-byte residual load); the residual
// rows of the item (<= 4 per lane) are requested before the first one is consumed, so the loop is
// not serialised on load latency.  WL2 = log2 of the width (compile time: the lane -> sample map
// and the row loop are fixed); the height is a run-time value.  An item covers at most
// 4 * (64 / (w / 4)) rows; taller transform blocks arrive as several items (`part`).
// FULL also handles inter-intra items (AV1 spec 7.11.3.13): the intra prediction of the whole block
// is blended into the inter prediction already in the picture, with a smooth 1-D weight ramp or a
// wedge mask, before the residual is added.
template <int WL2, bool FULL>
__device__ __forceinline__ void intra_finish_w(const FrameView &fv, const Item it, const Edges e, int lane)
{
    constexpr int w = 1 << WL2;
    constexpr int gpr = w >> 2;                        // 4-sample groups per row
    constexpr int ystep = 64 / gpr;                    // rows covered by the wave per step
    constexpr int iters = WL2 <= 3 ? 1 : 4;            // row-steps per item (rows_per_part() on the host)
    const int hl2 = it.hl2(), h = 1 << hl2;
    const int row0 = it.part() * (ystep * iters);
    const int pl = it.plane();
    const int mode = it.mode();
    const int flags = it.flags();
    const bool have_left = flags & kIIHaveLeft, have_top = flags & kIIHaveTop, has_resid = flags & kIIResid;
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    const int A = e.A, L = e.L, tl = e.tl;
    int dc = 128;
    if (mode == EA1_DC_PRED) {
        const int sa = wave_sum(A), sl = wave_sum(L);  // lanes beyond w / h contribute 0
        if (have_top && have_left) dc = (sa + sl + ((w + h) >> 1)) / (w + h);
        else if (have_left) dc = (sl + (h >> 1)) >> hl2;
        else if (have_top) dc = (sa + (w >> 1)) >> WL2;
    }
    const int g = lane & (gpr - 1), y0 = lane / gpr;
    const int px = 4 * g;
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
    const uint8_t *wtx = kSmWeights + w, *wty = kSmWeights + h;
    int a[4], wx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = __shfl(A, px + j, 64);
        wx[j] = wtx[px + j];
    }
    const int a_last = __shfl(A, w - 1, 64), l_last = __shfl(L, h - 1, 64);
    const int ii_kind = FULL ? it.ii() : 0;
    const int ii_scale = 128 >> (WL2 > hl2 ? WL2 : hl2);        // 128 / max(w, h)
    const uint8_t *wmask = FULL ? fv.wedge + it.aux() : nullptr;  // wedge inter-intra: mask of the luma block
    const int mask_sl2 = it.alpha();                             // ... and log2 of its row stride
    uint2 rr[iters];
#pragma unroll
    for (int k = 0; k < iters; k++) {
        rr[k] = make_uint2(0, 0);
        const int i = row0 + y0 + k * ystep;
        if (has_resid && i < h) rr[k] = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
    }
#pragma unroll
    for (int k = 0; k < iters; k++) {
        if (k && row0 + k * ystep >= h) continue;   // uniform: this item has fewer row-steps
        const int i = row0 + y0 + k * ystep;
        const int l = __shfl(L, i & 63, 64);   // convergent: executed by every lane
        const int wy = wty[i & (h - 1)];
        uint32_t cur_px = 0;
        if constexpr (FULL) {
            if (ii_kind && i < h) cur_px = *reinterpret_cast<const uint32_t *>(dst + size_t(i) * stride);
        }
        const int r4[4] = {int16_t(rr[k].x & 0xFFFF), int16_t(rr[k].x >> 16), int16_t(rr[k].y & 0xFFFF), int16_t(rr[k].y >> 16)};
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int v;
            switch (mode) {
            case EA1_DC_PRED: v = dc; break;
            case EA1_V_PRED: v = a[j]; break;
            case EA1_H_PRED: v = l; break;
            case EA1_PAETH_PRED: {
                const int base = a[j] + l - tl;
                const int dl = abs(base - l), dt = abs(base - a[j]), dtl = abs(base - tl);
                v = (dl <= dt && dl <= dtl) ? l : (dt <= dtl ? a[j] : tl);
                break;
            }
            case EA1_SMOOTH_PRED: v = (wy * a[j] + (256 - wy) * l_last + wx[j] * l + (256 - wx[j]) * a_last + 256) >> 9; break;
            case EA1_SMOOTH_V_PRED: v = (wy * a[j] + (256 - wy) * l_last + 128) >> 8; break;
            default: v = (wx[j] * l + (256 - wx[j]) * a_last + 128) >> 8; break;   // SMOOTH_H
            }
            if constexpr (FULL) {
                if (ii_kind) {
                    int m;
                    if (ii_kind & 0x80) {
                        // 7.11.3.14: the luma mask, averaged over the luma samples a chroma sample covers
                        const int msx = pl ? fv.ssx : 0, msy = pl ? fv.ssy : 0;
                        const uint8_t *q = wmask + ((i << msy) << mask_sl2) + ((px + j) << msx);
                        if (!msx) m = q[0];
                        else if (!msy) m = (q[0] + q[1] + 1) >> 1;
                        else m = (q[0] + q[1] + q[1 << mask_sl2] + q[(1 << mask_sl2) + 1] + 2) >> 2;
                    } else {
                        const int kind = (ii_kind & 0x7F) - 1;
                        m = kind == EA1_II_V ? kIiWeights[i * ii_scale]
                                             : (kind == EA1_II_H ? kIiWeights[(px + j) * ii_scale]
                                                                 : (kind == EA1_II_SMOOTH ? kIiWeights[(i < px + j ? i : px + j) * ii_scale] : 32));
                    }
                    const int inter = (cur_px >> (8 * j)) & 255;
                    v = (m * v + (64 - m) * inter + 32) >> 6;
                }
            }
            v += r4[j];
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            pk |= uint32_t(v) << (8 * j);
        }
        if (i < h) *reinterpret_cast<uint32_t *>(dst + size_t(i) * stride) = pk;
    }
}

// ---- predictors beyond the edge-register path (AV1 spec 7.11.2.3 - 7.11.2.12, 7.11.4, 7.11.5) ----

// add the residual to 4 predicted samples and store them
__device__ __forceinline__ void store4(uint8_t *dst, const int (&v)[4], uint2 rr)
{
    const int r4[4] = {int16_t(rr.x & 0xFFFF), int16_t(rr.x >> 16), int16_t(rr.y & 0xFFFF), int16_t(rr.y >> 16)};
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int t = v[j] + r4[j];
        t = t < 0 ? 0 : (t > 255 ? 255 : t);
        pk |= uint32_t(t) << (8 * j);
    }
    *reinterpret_cast<uint32_t *>(dst) = pk;
}

// 7.11.2.9
__device__ __forceinline__ int edge_strength(int w, int h, int filter_type, int delta)
{
    const int d = delta < 0 ? -delta : delta, blk_wh = w + h;
    int s = 0;
    if (!filter_type) {
        if (blk_wh <= 8) s = d >= 56 ? 1 : 0;
        else if (blk_wh <= 16) s = d >= 40 ? 1 : 0;
        else if (blk_wh <= 24) s = d >= 32 ? 3 : (d >= 16 ? 2 : (d >= 8 ? 1 : 0));
        else if (blk_wh <= 32) s = d >= 32 ? 3 : (d >= 4 ? 2 : (d >= 1 ? 1 : 0));
        else s = d >= 1 ? 3 : 0;
    } else {
        if (blk_wh <= 8) s = d >= 64 ? 2 : (d >= 40 ? 1 : 0);
        else if (blk_wh <= 16) s = d >= 48 ? 2 : (d >= 20 ? 1 : 0);
        else if (blk_wh <= 24) s = d >= 4 ? 3 : 0;
        else s = d >= 1 ? 3 : 0;
    }
    return s;
}
// 7.11.2.10
__device__ __forceinline__ bool use_upsample(int w, int h, int filter_type, int delta)
{
    const int d = delta < 0 ? -delta : delta;
    if (d <= 0 || d >= 40) return false;
    return filter_type ? w + h <= 8 : w + h <= 16;
}
// 7.11.2.12 on an LDS edge: buf[-1 .. sz - 2], entries 0 .. sz - 2 are replaced
__device__ __forceinline__ void edge_filter_lds(int16_t *buf, int sz, int strength, int lane)
{
    int out[2];   // sz <= 129: at most two entries per lane
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = 1 + lane + 64 * t;
        int sum = 0;
        if (i < sz) {
#pragma unroll
            for (int j = 0; j < 5; j++) {
                int k = i - 2 + j;
                k = k < 0 ? 0 : (k > sz - 1 ? sz - 1 : k);
                sum += kEdgeKernel[strength - 1][j] * buf[k - 1];
            }
        }
        out[t] = (sum + 8) >> 4;
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = 1 + lane + 64 * t;
        if (i < sz) buf[i - 1] = int16_t(out[t]);
    }
    wave_lds_sync();
}
// 7.11.2.11 on an LDS edge: buf[-1 .. n - 1] -> buf[-2 .. 2n - 2]  (n <= 16)
__device__ __forceinline__ void edge_upsample_lds(int16_t *buf, int n, int lane)
{
    int d[4] = {0, 0, 0, 0};
    if (lane < n) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int idx = lane + k - 2;
            idx = idx < -1 ? -1 : (idx > n - 1 ? n - 1 : idx);
            d[k] = buf[idx];
        }
    }
    const int corner = buf[-1];
    wave_lds_sync();
    if (lane < n) {
        int sv = (-d[0] + 9 * d[1] + 9 * d[2] - d[3] + 8) >> 4;
        sv = sv < 0 ? 0 : (sv > 255 ? 255 : sv);
        buf[2 * lane - 1] = int16_t(sv);
        buf[2 * lane] = int16_t(d[2]);
    }
    if (lane == 0) buf[-2] = int16_t(corner);
    wave_lds_sync();
}

// directional prediction (7.11.2.4) of one item; edges are prepared in this wave's LDS area
template <int WL2>
__device__ __forceinline__ void intra_directional(const FrameView &fv, const Item it, int lane, uint8_t *wlds, int p_angle)
{
    constexpr int w = 1 << WL2, gpr = w >> 2, ystep = 64 / gpr, iters = WL2 <= 3 ? 1 : 4;
    int16_t *ab = reinterpret_cast<int16_t *>(wlds) + kEdgeOrg;
    int16_t *lf = ab + kEdgeN;
    const int hl2 = it.hl2(), h = 1 << hl2, n = w + h;
    const int pl = it.plane(), flags = it.flags();
    const bool have_left = flags & kIIHaveLeft, have_top = flags & kIIHaveTop;
    const bool have_tr = flags & kIIHaveTopRight, have_bl = flags & kIIHaveBottomLeft;
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    const int max_x = fv.work.pw(pl) - 1, max_y = fv.work.ph(pl) - 1;
    int tl;
    if (have_top && have_left) tl = f[size_t(y - 1) * stride + x - 1];
    else if (have_top) tl = f[size_t(y - 1) * stride + x];
    else if (have_left) tl = f[size_t(y) * stride + x - 1];
    else tl = 128;
    wave_lds_sync();   // the previous item of this wave is done with the area
    for (int i = lane; i < n; i += 64) {
        int a, l;
        if (!have_top) {
            a = have_left ? f[size_t(y) * stride + x - 1] : 127;
        } else {
            // at most w real samples above-right, then the last one repeats (see the oracle)
            int xx = x + (i < w ? i : (have_tr ? (i < 2 * w ? i : 2 * w - 1) : w - 1));
            xx = xx < max_x ? xx : max_x;
            a = f[size_t(y - 1) * stride + xx];
        }
        if (!have_left) {
            l = have_top ? f[size_t(y - 1) * stride + x] : 129;
        } else {
            int yy = y + (i < h ? i : (have_bl ? (i < 2 * h ? i : 2 * h - 1) : h - 1));
            yy = yy < max_y ? yy : max_y;
            l = f[size_t(yy) * stride + x - 1];
        }
        ab[i] = int16_t(a);
        lf[i] = int16_t(l);
    }
    if (lane == 0) {
        ab[-1] = int16_t(tl);
        lf[-1] = int16_t(tl);
    }
    wave_lds_sync();
    int up_a = 0, up_l = 0;
    if (flags & kIIEdgeFilter) {
        const int ft = (flags & kIISmoothEdge) ? 1 : 0;
        if (p_angle > 90 && p_angle < 180 && n >= 24) {
            const int v = (lf[0] * 5 + ab[-1] * 6 + ab[0] * 5 + 8) >> 4;
            wave_lds_sync();
            if (lane == 0) {
                ab[-1] = int16_t(v);
                lf[-1] = int16_t(v);
            }
            wave_lds_sync();
        }
        if (have_top) {
            const int st = edge_strength(w, h, ft, p_angle - 90);
            if (st) edge_filter_lds(ab, w + (p_angle < 90 ? h : 0) + 1, st, lane);
        }
        if (have_left) {
            const int st = edge_strength(w, h, ft, p_angle - 180);
            if (st) edge_filter_lds(lf, h + (p_angle > 180 ? w : 0) + 1, st, lane);
        }
        up_a = use_upsample(w, h, ft, p_angle - 90) ? 1 : 0;
        if (up_a) edge_upsample_lds(ab, w + (p_angle < 90 ? h : 0), lane);
        up_l = use_upsample(w, h, ft, p_angle - 180) ? 1 : 0;
        if (up_l) edge_upsample_lds(lf, h + (p_angle > 180 ? w : 0), lane);
    }
    int dx = 0, dy = 0;
    if (p_angle < 90) dx = kDrIntra[p_angle];
    else if (p_angle < 180) dx = kDrIntra[180 - p_angle];
    if (p_angle > 90 && p_angle < 180) dy = kDrIntra[p_angle - 90];
    else if (p_angle > 180) dy = kDrIntra[270 - p_angle];
    const int row0 = it.part() * (ystep * iters);
    const int g = lane & (gpr - 1), y0 = lane / gpr, px = 4 * g;
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
    const bool has_resid = flags & kIIResid;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        if (k && row0 + k * ystep >= h) continue;
        const int i = row0 + y0 + k * ystep;
        if (i >= h) continue;
        uint2 rr = make_uint2(0, 0);
        if (has_resid) rr = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
        int v[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = px + jj;
            if (p_angle < 90) {
                const int idx = (i + 1) * dx;
                const int base = (idx >> (6 - up_a)) + (j << up_a);
                const int shift = ((idx << up_a) >> 1) & 0x1F;
                const int max_base = (w + h - 1) << up_a;
                v[jj] = base < max_base ? (ab[base] * (32 - shift) + ab[base + 1] * shift + 16) >> 5 : ab[max_base];
            } else if (p_angle < 180) {
                int idx = (j << 6) - (i + 1) * dx;
                int base = idx >> (6 - up_a);
                if (base >= -(1 << up_a)) {
                    const int shift = ((idx * (1 << up_a)) >> 1) & 0x1F;
                    v[jj] = (ab[base] * (32 - shift) + ab[base + 1] * shift + 16) >> 5;
                } else {
                    idx = (i << 6) - (j + 1) * dy;
                    base = idx >> (6 - up_l);
                    const int shift = ((idx * (1 << up_l)) >> 1) & 0x1F;
                    v[jj] = (lf[base] * (32 - shift) + lf[base + 1] * shift + 16) >> 5;
                }
            } else {
                const int idx = (j + 1) * dy;
                const int base = (idx >> (6 - up_l)) + (i << up_l);
                const int shift = ((idx << up_l) >> 1) & 0x1F;
                v[jj] = (lf[base] * (32 - shift) + lf[base + 1] * shift + 16) >> 5;
            }
        }
        store4(dst + size_t(i) * stride, v, rr);
    }
}

// recursive filter intra (7.11.2.3): 4x2 cells along anti-diagonals, the prediction grows in LDS
template <int WL2>
__device__ __forceinline__ void intra_filter(const FrameView &fv, const Item it, const Edges e, int lane, uint8_t *wlds)
{
    constexpr int w = 1 << WL2, gpr = w >> 2, ystep = 64 / gpr, iters = WL2 <= 3 ? 1 : 4;
    uint8_t *pb = wlds + 2 * kEdgeN * 2;   // row r + 1, column c + 1 = prediction sample (r, c)
    const int hl2 = it.hl2(), h = 1 << hl2;
    const int pl = it.plane(), flags = it.flags();
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    const int fmode = it.mode() - kIntraFilter;
    wave_lds_sync();
    if (lane < w) pb[lane + 1] = uint8_t(e.A);
    if (lane < h) pb[(lane + 1) * kFiStride] = uint8_t(e.L);
    if (lane == 0) pb[0] = uint8_t(e.tl);
    wave_lds_sync();
    constexpr int cw = w / 4;
    const int ch = h / 2;
    const int cell = lane >> 3, o = lane & 7, i1 = o >> 2, j1 = o & 3;
    for (int d = 0; d < cw + ch - 1; d++) {
        const int i2min = d - (cw - 1) > 0 ? d - (cw - 1) : 0, i2max = d < ch - 1 ? d : ch - 1;
        const int i2 = i2min + cell, j4 = d - i2;
        const bool on = i2 <= i2max;
        int val = 0;
        if (on) {
            const uint8_t *c0 = pb + (2 * i2) * kFiStride + 4 * j4;
            const int p[7] = {c0[0], c0[1], c0[2], c0[3], c0[4], c0[kFiStride], c0[2 * kFiStride]};
            int pr = 0;
#pragma unroll
            for (int t = 0; t < 7; t++) pr += kFilterTaps[fmode][o][t] * p[t];
            pr = pr >= 0 ? (pr + 8) >> 4 : -((-pr + 8) >> 4);
            val = pr < 0 ? 0 : (pr > 255 ? 255 : pr);
        }
        wave_lds_sync();
        if (on) pb[(2 * i2 + 1 + i1) * kFiStride + 4 * j4 + 1 + j1] = uint8_t(val);
        wave_lds_sync();
    }
    const int g = lane & (gpr - 1), y0 = lane / gpr, px = 4 * g;
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        const int i = y0 + k * ystep;
        if (i >= h) continue;
        uint2 rr = make_uint2(0, 0);
        if (flags & kIIResid) rr = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
        const uint8_t *src = pb + (i + 1) * kFiStride + px + 1;
        const int v[4] = {src[0], src[1], src[2], src[3]};
        store4(dst + size_t(i) * stride, v, rr);
    }
}

// chroma from luma (7.11.5): DC prediction + alpha * (subsampled luma - its block average)
template <int WL2>
__device__ __forceinline__ void intra_cfl(const FrameView &fv, const Item it, const Edges e, int lane)
{
    constexpr int w = 1 << WL2, gpr = w >> 2, ystep = 64 / gpr, iters = WL2 <= 3 ? 1 : 4;
    const int hl2 = it.hl2(), h = 1 << hl2;
    const int pl = it.plane(), flags = it.flags();
    const bool have_left = flags & kIIHaveLeft, have_top = flags & kIIHaveTop;
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    int dc = 128;
    {
        const int sa = wave_sum(e.A), sl = wave_sum(e.L);
        if (have_top && have_left) dc = (sa + sl + ((w + h) >> 1)) / (w + h);
        else if (have_left) dc = (sl + (h >> 1)) >> hl2;
        else if (have_top) dc = (sa + (w >> 1)) >> WL2;
    }
    const int g = lane & (gpr - 1), y0 = lane / gpr, px = 4 * g;
    const uint8_t *luma = fv.work.p[0];
    const int ls = fv.work.stride[0];
    const int csx = fv.ssx, csy = fv.ssy;   // (uniform) 4:2:0: 2x2 luma samples per chroma sample, sum << 1; 4:2:2: 2x1, << 2; 4:4:4: << 3
    int L[iters][4];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        const int i = y0 + k * ystep;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) L[k][jj] = 0;
        if (i < h && (it.aux() || !(csx && csy))) {
            // a block that reaches beyond the coded frame: luma beyond MaxLumaW / MaxLumaH (aux) repeats the last coded column / row
            const int mlw = it.aux() ? int(it.aux() & 0xFFFF) : 1 << 30, mlh = it.aux() ? int(it.aux() >> 16) : 1 << 30;
            const int ly = min((y + i) << csy, mlh - 1 - csy);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int lx = min((x + px + jj) << csx, mlw - 1 - csx);
                const uint8_t *q = luma + size_t(ly) * ls + lx;
                int t = int(q[0]);
                if (csx) t += int(q[1]);
                if (csy) t += int(q[ls]) + int(q[ls + 1]);
                L[k][jj] = t << (3 - csx - csy);
                sum += L[k][jj];
            }
        } else if (i < h) {
            const uint8_t *r0 = luma + size_t(2 * (y + i)) * ls + 2 * (x + px);
            const uint2 a = *reinterpret_cast<const uint2 *>(r0), b = *reinterpret_cast<const uint2 *>(r0 + ls);
            const uint32_t wa[2] = {a.x, a.y}, wb[2] = {b.x, b.y};
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const uint32_t ua = wa[jj >> 1] >> (16 * (jj & 1)), ub = wb[jj >> 1] >> (16 * (jj & 1));
                const int t4 = int(ua & 255) + int((ua >> 8) & 255) + int(ub & 255) + int((ub >> 8) & 255);
                L[k][jj] = t4 << 1;
                sum += t4 << 1;
            }
        }
    }
    sum = wave_sum(sum);
    const int avg = (sum + ((1 << (WL2 + hl2)) >> 1)) >> (WL2 + hl2);
    const int alpha = it.alpha();
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        const int i = y0 + k * ystep;
        if (i >= h) continue;
        uint2 rr = make_uint2(0, 0);
        if (flags & kIIResid) rr = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
        int v[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int t = alpha * (L[k][jj] - avg);
            const int sc = t >= 0 ? (t + 32) >> 6 : -((-t + 32) >> 6);
            int q = dc + sc;
            v[jj] = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
        store4(dst + size_t(i) * stride, v, rr);
    }
}

// palette (7.11.4): colour index map -> colours
template <int WL2>
__device__ __forceinline__ void intra_palette(const FrameView &fv, const Item it, int lane)
{
    constexpr int w = 1 << WL2, gpr = w >> 2, ystep = 64 / gpr, iters = WL2 <= 3 ? 1 : 4;
    const int h = 1 << it.hl2();
    const int pl = it.plane(), flags = it.flags();
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x(), y = it.y();
    const PalRef pr = fv.pal_refs[it.aux()];
    const uint8_t *colors = fv.aux + pr.colors_off, *cmap = fv.aux + pr.map_off;
    const int row0 = it.part() * (ystep * iters);
    const int g = lane & (gpr - 1), y0 = lane / gpr, px = 4 * g;
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        if (k && row0 + k * ystep >= h) continue;
        const int i = row0 + y0 + k * ystep;
        if (i >= h) continue;
        uint2 rr = make_uint2(0, 0);
        if (flags & kIIResid) rr = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
        const uint8_t *m = cmap + size_t(i) * pr.map_stride + px;
        const int v[4] = {colors[m[0]], colors[m[1]], colors[m[2]], colors[m[3]]};
        store4(dst + size_t(i) * stride, v, rr);
    }
}

// intra block copy: the samples of the current picture displaced by the block vector (whole luma
// samples; a chroma position may be a half sample, interpolated bilinearly in two passes exactly
// like a single-reference inter prediction with the bilinear filter: Round2(., 3) then
// Round2(., 11)).  The source belongs to earlier wavefront levels (job creation checks that).
template <int WL2>
__device__ __forceinline__ void intra_bc(const FrameView &fv, const Item it, int lane)
{
    constexpr int w = 1 << WL2, gpr = w >> 2, ystep = 64 / gpr, iters = WL2 <= 3 ? 1 : 4;
    const int h = 1 << it.hl2();
    const int pl = it.plane(), flags = it.flags(), bsx = pl ? fv.ssx : 0, bsy = pl ? fv.ssy : 0;
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int W = fv.work.pw(pl), H = fv.work.ph(pl);
    const int x = it.x(), y = it.y();
    const int mvy = int16_t(it.aux() >> 16), mvx = int16_t(it.aux() & 0xFFFF);
    const int px16 = (x << 4) + (bsx ? mvx : mvx * 2), py16 = (y << 4) + (bsy ? mvy : mvy * 2);
    const int sx = px16 >> 4, sy = py16 >> 4, fx = px16 & 15, fy = py16 & 15;
    const int h1 = 8 * fx, h0 = 128 - h1, v1 = 8 * fy, v0 = 128 - v1;   // bilinear taps (sum 128)
    const int row0 = it.part() * (ystep * iters);
    const int g = lane & (gpr - 1), y0 = lane / gpr, px = 4 * g;
    uint8_t *dst = f + size_t(y) * stride + x + px;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x + px;
#pragma unroll
    for (int k = 0; k < iters; k++) {
        if (k && row0 + k * ystep >= h) continue;
        const int i = row0 + y0 + k * ystep;
        if (i >= h) continue;
        uint2 rr = make_uint2(0, 0);
        if (flags & kIIResid) rr = *reinterpret_cast<const uint2 *>(res + size_t(i) * stride);
        const uint8_t *r0 = f + size_t(min(max(sy + i, 0), H - 1)) * stride, *r1 = f + size_t(min(max(sy + i + 1, 0), H - 1)) * stride;
        int m0[5], m1[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int c = min(max(sx + px + j, 0), W - 1);
            m0[j] = r0[c];
            m1[j] = r1[c];
        }
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int a = (h0 * m0[j] + h1 * m0[j + 1] + 4) >> 3, b = (h0 * m1[j] + h1 * m1[j + 1] + 4) >> 3;
            v[j] = (v0 * a + v1 * b + 1024) >> 11;
        }
        store4(dst + size_t(i) * stride, v, rr);
    }
}

// FULL = false: only the edge-register predictors are compiled in (the job builder knows whether a
// frame needs more: JobStats::intra_special); keeps the common kernel lean in registers
template <int WL2, bool FULL>
__device__ __forceinline__ void intra_any(const FrameView &fv, const Item it, const Edges e, int lane, uint8_t *wlds)
{
    if constexpr (!FULL) {
        intra_finish_w<WL2, false>(fv, it, e, lane);
        return;
    }
    const int mode = it.mode();
    if (mode == EA1_UV_CFL_PRED) {
        if constexpr (WL2 <= 5) intra_cfl<WL2>(fv, it, e, lane);
        return;
    }
    if (mode == kIntraPalette) {
        intra_palette<WL2>(fv, it, lane);
        return;
    }
    if (mode == kIntraBC) {
        intra_bc<WL2>(fv, it, lane);
        return;
    }
    if (mode >= kIntraFilter) {
        if constexpr (WL2 <= 5) intra_filter<WL2>(fv, it, e, lane, wlds);
        return;
    }
    if (mode >= EA1_V_PRED && mode <= EA1_D67_PRED) {
        const int p_angle = kModeAngle[mode] + 3 * it.angle_delta();
        if (p_angle != 90 && p_angle != 180) {
            intra_directional<WL2>(fv, it, lane, wlds, p_angle);
            return;
        }
    }
    intra_finish_w<WL2, true>(fv, it, e, lane);
}

template <bool FULL>
__device__ __forceinline__ void intra_finish(const FrameView &fv, const Item it, const Edges e, int lane, uint8_t *wlds)
{
    switch (it.wl2()) {  // uniform per wave
    case 2: intra_any<2, FULL>(fv, it, e, lane, wlds); break;
    case 3: intra_any<3, FULL>(fv, it, e, lane, wlds); break;
    case 4: intra_any<4, FULL>(fv, it, e, lane, wlds); break;
    case 5: intra_any<5, FULL>(fv, it, e, lane, wlds); break;
    default: intra_any<6, FULL>(fv, it, e, lane, wlds); break;
    }
}

template <bool FULL>
__device__ __forceinline__ void intra_item(const FrameView &fv, const Item it, int lane, uint8_t *wlds)
{
    intra_finish<FULL>(fv, it, intra_load(fv, it, lane), lane, wlds);
}

// one level of every chain as a plain launch: one one-wave workgroup per item.  Used for the WIDE levels (launch_intra): level 0 of
// an inter frame, the body of a key frame -- spreading such a level over the whole chip beats walking it on one CU per chain
template <bool FULL>
__global__ void __launch_bounds__(64)
intra_level_kernel(Batch bt, int level)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || fvp->num_levels <= level) return;
    const FrameView &fv = *fvp;
    const uint32_t first = fv.level_start[level], last = fv.level_start[level + 1];
    if (blockIdx.x >= last - first) return;
    __shared__ __attribute__((aligned(16))) uint8_t wlds[FULL ? kWaveLds : 16];
    intra_item<FULL>(fv, load_item(&fv.intra_items[first + blockIdx.x]), threadIdx.x, wlds);
}

// a run of NARROW levels [first_level, end_level) (an inter frame's scattered intra blocks: dozens to hundreds of levels of a few
// items; the ramp at either end of a key frame): one 16-wave workgroup per chain, workgroup barrier between levels
template <bool FULL>
__global__ void __launch_bounds__(64 * kWaveWaves)
intra_wave_kernel(Batch bt, int first_level, int end_level)
{
    const FrameView *fvp = bt.v[blockIdx.x];
    if (!fvp || fvp->num_levels <= first_level) return;
    const FrameView &fv = *fvp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[FULL ? kWaveWaves * kWaveLds : 16];
    uint8_t *wlds = lds_all + (FULL ? wv * kWaveLds : 0);
    const int nlev = min(fv.num_levels, end_level);
    for (int lv = first_level; lv < nlev; lv++) {
        const uint32_t first = fv.level_start[lv], last = fv.level_start[lv + 1];
        // items of one level are independent; the next item's descriptor + edge loads are issued
        // before the current item is finished
        uint32_t idx = first + wv;
        Item cur;
        Edges ce;
        if (idx < last) {
            cur = load_item(&fv.intra_items[idx]);
            ce = intra_load(fv, cur, lane);
        }
        while (idx < last) {
            const uint32_t nidx = idx + kWaveWaves;
            Item nxt = cur;
            Edges ne = ce;
            if (nidx < last) {
                nxt = load_item(&fv.intra_items[nidx]);
                ne = intra_load(fv, nxt, lane);
            }
            intra_finish<FULL>(fv, cur, ce, lane, wlds);
            cur = nxt;
            ce = ne;
            idx = nidx;
        }
        __syncthreads();  // this CU's stores of level lv are ordered before any load of level lv + 1
    }
}

// (Round 3 tried the levels of a wide batch behind a grid barrier in device memory -- 1024 one-wave workgroups spinning on an arrival counter.
// Measured 3.6x SLOWER than a launch per level (63 ms against 17.6 ms for the ~900 levels of 32 4K key frames: a level boundary costs the
// cross-XCD coherence of the picture, which a kernel boundary pays once and a thousand waves' release / acquire fences a thousand times;
// profiles/r03_ab_intra_grid_barrier_bench.json), and a spin barrier needs every workgroup resident at once, which nothing guarantees when
// another context or stream holds CUs: a hang waiting to happen.  Removed in round 4; the measurement stays in profiles/.)

void launch_intra(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    int max_levels = 0, active = 0;
    uint64_t bytes = 0;
    bool full = false;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        active++;
        full |= s.stats[c]->intra_special;
        max_levels = std::max(max_levels, s.stats[c]->num_levels);
        // predicted pixels written once + int16 residual read where present
        bytes += s.stats[c]->intra_pixels + 2 * s.stats[c]->intra_resid_pixels;
    }
    if (!max_levels) return;
    // Every level is as wide as its widest chain.  WIDE levels (kWideLevel items or more: the body of a key frame, level 0 of an inter
    // frame) get a launch each -- one one-wave workgroup per item over the whole chip, ~19 us whatever the width; runs of NARROW levels are
    // walked by one workgroup per chain at ~3 us per level and 16 items.  (Until round 4 the choice was made once per batch, by the mean
    // width over all chains: a lone 4K key frame -- 663 levels of 584 items -- beside 15 inter frames went to ONE workgroup, 82 ms against
    // 10 ms for 16 key frames together; tools/dbg/real_levels.py, profiles/r04_z_real_levels.txt.)
    static const int wide = getenv("EA1_INTRA_WIDE") ? atoi(getenv("EA1_INTRA_WIDE")) : kWideLevel;   // (diagnostic knobs)
    static const int short_run = getenv("EA1_INTRA_SHORT_RUN") ? atoi(getenv("EA1_INTRA_SHORT_RUN")) : kShortRun;
    std::vector<int> width(size_t(max_levels), 0);
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        for (int k = 0; k < s.stats[c]->num_levels; k++)
            width[size_t(k)] = std::max(width[size_t(k)], int(s.stats[c]->level_start[k + 1] - s.stats[c]->level_start[k]));
    }
    (void)active;
    // the whole wavefront is timed as one entry
    ProfScope ps(prof, "intra_level_kernel+intra_wave_kernel", bytes, st);
    auto launch_level = [&](int k) {
        if (!width[size_t(k)]) return;
        if (full) hipLaunchKernelGGL(intra_level_kernel<true>, dim3(width[size_t(k)], 1, s.batch.n), dim3(64), 0, st, s.batch, k);
        else hipLaunchKernelGGL(intra_level_kernel<false>, dim3(width[size_t(k)], 1, s.batch.n), dim3(64), 0, st, s.batch, k);
    };
    int k = 0;
    while (k < max_levels) {
        if (width[size_t(k)] >= wide) {
            launch_level(k++);
            continue;
        }
        int e = k;
        while (e < max_levels && width[size_t(e)] < wide) e++;
        if (e - k < short_run) {   // a run this short is cheaper as launches than as one more kernel that has to start and drain
            while (k < e) launch_level(k++);
            continue;
        }
        if (full) hipLaunchKernelGGL(intra_wave_kernel<true>, dim3(s.batch.n), dim3(64 * kWaveWaves), 0, st, s.batch, k, e);
        else hipLaunchKernelGGL(intra_wave_kernel<false>, dim3(s.batch.n), dim3(64 * kWaveWaves), 0, st, s.batch, k, e);
        k = e;
    }
}

}  // namespace ea1
