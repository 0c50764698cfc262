// Intra prediction for the intra wavefront (dav1d src/ipred_tmpl.c + recon_tmpl.c,
// /root/reference/cmake/dav1d.cmake:1007,:1014; AV1 spec 7.11.2).
//
// Intra blocks read reconstructed neighbours (left column, top row, top-left) of the CURRENT
// frame, so they form a dependency wavefront; inter blocks do not and are finished before it.
// Only the PREDICTION is serial: the residual of every intra block was already inverse-transformed
// into the chain's int16 residual plane by the parallel itx kernels, so the wavefront step is
// just "edges -> predict -> add residual -> clip", a few microseconds per level.
// The host sorts intra (block, plane) items by dependency level; one launch per level covers that
// level of every chain (blockIdx.z), one 64-lane workgroup per item.
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const uint8_t kSmWeights[128] = {EA1_SM_WEIGHTS_VALUES};

__global__ void __launch_bounds__(64)
intra_level_kernel(Batch bt, int level)
{
    __shared__ int above[64], left[64];
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || level >= fvp->num_levels) return;
    const FrameView &fv = *fvp;
    const uint32_t first = fv.level_start[level], last = fv.level_start[level + 1];
    if (blockIdx.x >= last - first) return;
    const TxItem it = fv.intra_items[first + blockIdx.x];
    const ea1_block b = fv.blocks[it.blk];
    const int pl = it.plane, ss = pl ? 1 : 0;
    const int n = (1 << b.bwl) >> ss;
    const int mode = pl ? b.mode_uv : b.mode_y;
    const bool have_left = b.flags & EA1_BLK_HAVE_LEFT, have_top = b.flags & EA1_BLK_HAVE_TOP;
    const int stride = fv.work.stride[pl];
    uint8_t *f = fv.work.p[pl];
    const int x = it.x, y = it.y;
    const int lane = threadIdx.x;
    if (lane < n) {
        above[lane] = have_top ? f[size_t(y - 1) * stride + x + lane] : (have_left ? f[size_t(y) * stride + x - 1] : 127);
        left[lane] = have_left ? f[size_t(y + lane) * stride + x - 1] : (have_top ? f[size_t(y - 1) * stride + x] : 129);
    }
    int tl;
    if (have_top && have_left) tl = f[size_t(y - 1) * stride + x - 1];
    else if (have_top) tl = f[size_t(y - 1) * stride + x];
    else if (have_left) tl = f[size_t(y) * stride + x - 1];
    else tl = 128;
    __syncthreads();
    int dc = 128;
    if (mode == EA1_DC_PRED) {
        int sa = 0, sl = 0;
        for (int i = 0; i < n; i++) {
            sa += above[i];
            sl += left[i];
        }
        const int l2 = 31 - __clz(n);
        if (have_top && have_left) dc = (sa + sl + n) >> (l2 + 1);  // (sum + (w+h)/2) / (w+h), w == h
        else if (have_left) dc = (sl + (n >> 1)) >> l2;
        else if (have_top) dc = (sa + (n >> 1)) >> l2;
    }
    const int px = lane & (n - 1), y0 = lane / n, ystep = 64 / n;
    uint8_t *dst = f + size_t(y) * stride + x;
    const bool has_resid = it.coef_off != 0xFFFFFFFFu;
    const int16_t *res = fv.resid[pl] + size_t(y) * stride + x;
    const uint8_t *wt = kSmWeights + n;
    const int a = above[px], a_last = above[n - 1], l_last = left[n - 1], wx = wt[px];
    for (int i = y0; i < n; i += ystep) {
        int v;
        const int l = left[i];
        switch (mode) {
        case EA1_DC_PRED: v = dc; break;
        case EA1_V_PRED: v = a; break;
        case EA1_H_PRED: v = l; break;
        case EA1_PAETH_PRED: {
            const int base = a + l - tl;
            const int dl = abs(base - l), dt = abs(base - a), dtl = abs(base - tl);
            v = (dl <= dt && dl <= dtl) ? l : (dt <= dtl ? a : tl);
            break;
        }
        case EA1_SMOOTH_PRED: {
            const int wy = wt[i];
            v = (wy * a + (256 - wy) * l_last + wx * l + (256 - wx) * a_last + 256) >> 9;
            break;
        }
        case EA1_SMOOTH_V_PRED: {
            const int wy = wt[i];
            v = (wy * a + (256 - wy) * l_last + 128) >> 8;
            break;
        }
        default: v = (wx * l + (256 - wx) * a_last + 128) >> 8; break;
        }
        if (has_resid) {
            v += res[size_t(i) * stride + px];
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        dst[size_t(i) * stride + px] = uint8_t(v);
    }
}

void launch_intra(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    int max_levels = 0;
    uint64_t bytes = 0;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        max_levels = std::max(max_levels, s.stats[c]->num_levels);
        // predicted pixels written once + int16 residual read where present
        bytes += s.stats[c]->intra_pixels + 2 * s.stats[c]->intra_resid_pixels;
    }
    if (!max_levels) return;
    // the whole wavefront is timed as one entry (max_levels launches of intra_level_kernel)
    ProfScope ps(prof, "intra_level_kernel(all levels)", bytes, st);
    for (int lv = 0; lv < max_levels; lv++) {
        int max_items = 0;
        for (int c = 0; c < s.batch.n; c++) {
            if (!s.stats[c] || lv >= s.stats[c]->num_levels) continue;
            max_items = std::max(max_items, s.stats[c]->max_level_items);
        }
        hipLaunchKernelGGL(intra_level_kernel, dim3(max_items, 1, s.batch.n), dim3(64), 0, st, s.batch, lv);
    }
}

}  // namespace ea1
