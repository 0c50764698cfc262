// Intra prediction + residual for the intra wavefront (dav1d src/ipred_tmpl.c + recon_tmpl.c,
// /root/reference/cmake/dav1d.cmake:1007,:1014; AV1 spec 7.11.2).
//
// Intra blocks read reconstructed neighbours (left column, top row, top-left) of the CURRENT
// frame, so they form a dependency wavefront; inter blocks do not and are finished before it.
// The host sorts intra (block, plane) items by dependency level; one launch per level, one
// 64-lane workgroup per item: edges -> LDS, prediction written to the frame, then the inverse
// transform adds the residual in the same workgroup.
#include "av1_dev.h"
#include "av1_itx_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const uint8_t kSmWeights[128] = {EA1_SM_WEIGHTS_VALUES};

template <int N>
__device__ __forceinline__ void residual(int16_t *lds, const int16_t *coef, uint8_t *dst, int stride, int txtp, int lane)
{
    itx::tx_block_add<N, N>(lds, coef, dst, stride, txtp, lane, lane < itx::TxLds<N, N>::THREADS);
}

__global__ void __launch_bounds__(64)
intra_level_kernel(const TxItem *__restrict__ items, const ea1_block *__restrict__ blocks,
                   const int16_t *__restrict__ coefs, DevPic cur)
{
    __shared__ __attribute__((aligned(16))) int16_t lds[itx::TxLds<64, 64>::TOTAL];
    __shared__ int above[64], left[64];
    const TxItem it = items[blockIdx.x];
    const ea1_block b = blocks[it.blk];
    const int pl = it.plane, ss = pl ? 1 : 0;
    const int n = (1 << b.bwl) >> ss;
    const int mode = pl ? b.mode_uv : b.mode_y;
    const bool have_left = b.flags & EA1_BLK_HAVE_LEFT, have_top = b.flags & EA1_BLK_HAVE_TOP;
    const int stride = cur.stride[pl];
    uint8_t *f = cur.p[pl];
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
        dst[size_t(i) * stride + px] = uint8_t(v);
    }
    if (it.coef_off == 0xFFFFFFFFu) return;  // uniform
    __syncthreads();  // prediction visible to the lanes that add the residual (same workgroup)
    const int16_t *cf = coefs + it.coef_off;
    switch (n) {
    case 4: residual<4>(lds, cf, dst, stride, it.txtp, lane); break;
    case 8: residual<8>(lds, cf, dst, stride, it.txtp, lane); break;
    case 16: residual<16>(lds, cf, dst, stride, it.txtp, lane); break;
    case 32: residual<32>(lds, cf, dst, stride, it.txtp, lane); break;
    default: residual<64>(lds, cf, dst, stride, it.txtp, lane); break;
    }
}

void launch_intra_level(const JobDev &job, const DevPic &cur, int first_item, int count, hipStream_t st)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(intra_level_kernel, dim3(count), dim3(64), 0, st, job.intra_items + first_item, job.blocks,
                       job.coefs, cur);
}

}  // namespace ea1
