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

constexpr int kWaveWaves = 16;  // waves of the per-chain wavefront workgroup

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
    if (lane < w) e.A = have_top ? f[size_t(y - 1) * stride + x + lane] : (have_left ? f[size_t(y) * stride + x - 1] : 127);
    if (lane < h) e.L = have_left ? f[size_t(y + lane) * stride + x - 1] : (have_top ? f[size_t(y - 1) * stride + x] : 129);
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
template <int WL2>
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
            v += r4[j];
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            pk |= uint32_t(v) << (8 * j);
        }
        if (i < h) *reinterpret_cast<uint32_t *>(dst + size_t(i) * stride) = pk;
    }
}

__device__ __forceinline__ void intra_finish(const FrameView &fv, const Item it, const Edges e, int lane)
{
    switch (it.wl2()) {  // uniform per wave
    case 2: intra_finish_w<2>(fv, it, e, lane); break;
    case 3: intra_finish_w<3>(fv, it, e, lane); break;
    case 4: intra_finish_w<4>(fv, it, e, lane); break;
    case 5: intra_finish_w<5>(fv, it, e, lane); break;
    default: intra_finish_w<6>(fv, it, e, lane); break;
    }
}

__device__ __forceinline__ void intra_item(const FrameView &fv, const Item it, int lane)
{
    intra_finish(fv, it, intra_load(fv, it, lane), lane);
}

// level 0 of every chain: one one-wave workgroup per item
__global__ void __launch_bounds__(64)
intra_level0_kernel(Batch bt)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || fvp->num_levels < 1) return;
    const FrameView &fv = *fvp;
    const uint32_t first = fv.level_start[0], last = fv.level_start[1];
    if (blockIdx.x >= last - first) return;
    intra_item(fv, load_item(&fv.intra_items[first + blockIdx.x]), threadIdx.x);
}

// levels 1..: one 16-wave workgroup per chain, workgroup barrier between levels
__global__ void __launch_bounds__(64 * kWaveWaves)
intra_wave_kernel(Batch bt)
{
    const FrameView *fvp = bt.v[blockIdx.x];
    if (!fvp || fvp->num_levels < 2) return;
    const FrameView &fv = *fvp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nlev = fv.num_levels;
    for (int lv = 1; lv < nlev; lv++) {
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
            intra_finish(fv, cur, ce, lane);
            cur = nxt;
            ce = ne;
            idx = nidx;
        }
        __syncthreads();  // this CU's stores of level lv are ordered before any load of level lv + 1
    }
}

void launch_intra(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    int max_levels = 0, max_items0 = 0;
    uint64_t bytes = 0;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        max_levels = std::max(max_levels, s.stats[c]->num_levels);
        max_items0 = std::max(max_items0, s.stats[c]->level0_items);
        // predicted pixels written once + int16 residual read where present
        bytes += s.stats[c]->intra_pixels + 2 * s.stats[c]->intra_resid_pixels;
    }
    if (!max_levels) return;
    // the whole wavefront is timed as one entry
    ProfScope ps(prof, "intra_level0_kernel+intra_wave_kernel", bytes, st);
    hipLaunchKernelGGL(intra_level0_kernel, dim3(max_items0, 1, s.batch.n), dim3(64), 0, st, s.batch);
    if (max_levels > 1) hipLaunchKernelGGL(intra_wave_kernel, dim3(s.batch.n), dim3(64 * kWaveWaves), 0, st, s.batch);
}

}  // namespace ea1
