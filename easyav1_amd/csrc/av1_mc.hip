// Inter prediction: separable 8-tap sub-pel interpolation + compound average.
// Arithmetic: AV1 spec 7.11.3.4 / dav1d src/mc_tmpl.c put/prep_8tap + avg (listed at
// /root/reference/cmake/dav1d.cmake:1013): horizontal Round2(sum, 3), vertical Round2(sum, 11)
// for a single reference, Round2(sum, 7) per reference and Round2(a + b, 5) for the average.
// Each pass is one exact integer sum and one rounding, identical to dav1d's fused shortcuts.
//
// Mapping: one 64-lane workgroup per 8x8 luma unit (+ its two 4x4 chroma units).  The reference
// window (15x15 luma, 11x11 per chroma plane; coordinates clamped to the picture = edge
// emulation) is staged in LDS once, the horizontal pass writes a 15x8 / 11x4 intermediate tile
// to LDS, the vertical pass reads it back; one lane = one luma pixel.  Blocks larger than 8x8 are
// simply several units with the same motion vector (the filter is position independent).
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const int16_t kSubpel[5][16][8] = {EA1_SUBPEL_FILTERS_VALUES};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void __launch_bounds__(64)
inter_pred_kernel(Batch bt)
{
    __shared__ uint8_t win_y[15 * 16];
    __shared__ uint8_t win_c[2][11 * 12];
    __shared__ int16_t mid_y[15 * 8];
    __shared__ int16_t mid_c[2][11 * 4];

    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || !fvp->desc.is_inter) return;
    const FrameView &fv = *fvp;
    const int w8 = fv.w8;
    const DevPic *__restrict__ refs = fv.refs;
    const DevPic &cur = fv.work;
    const int u = blockIdx.x;
    const int x8 = u % w8, y8 = u / w8;
    const ea1_block b = fv.blocks[fv.blk_map[u]];
    if (!(b.flags & EA1_BLK_INTER)) return;  // uniform for the whole workgroup
    const int lane = threadIdx.x;
    const int nref = (b.flags & EA1_BLK_COMPOUND) ? 2 : 1;
    const int kind_h = b.mode_y & 3, kind_v = (b.mode_y >> 2) & 3;
    // luma blocks are >= 8 wide: 8-tap sets; chroma of an 8x8 block is 4 wide: 4-tap sets
    const bool c4 = b.bwl == 3;
    const int fih_c = c4 ? (kind_h == EA1_FILTER_SMOOTH ? 4 : 3) : kind_h;
    const int fiv_c = c4 ? (kind_v == EA1_FILTER_SMOOTH ? 4 : 3) : kind_v;
    const int rnd1 = nref == 2 ? 7 : 11;

    int acc_y = 0, acc_c = 0;
    for (int r = 0; r < nref; r++) {
        const DevPic ref = refs[b.ref[r]];
        const int mvy = b.mv[r][0], mvx = b.mv[r][1];
        const int px = ((x8 * 8) << 4) + 2 * mvx, py = ((y8 * 8) << 4) + 2 * mvy;
        const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
        const int pcx = ((x8 * 4) << 4) + mvx, pcy = ((y8 * 4) << 4) + mvy;
        const int icx = pcx >> 4, icy = pcy >> 4, fcx = pcx & 15, fcy = pcy & 15;
        if (r) __syncthreads();
        {
            const int W = ref.w, H = ref.h, stride = ref.stride[0];
            const uint8_t *src = ref.p[0];
            for (int i = lane; i < 225; i += 64) {
                const int rr = i / 15, cc = i - rr * 15;
                win_y[rr * 16 + cc] = src[size_t(clampi(iy - 3 + rr, 0, H - 1)) * stride + clampi(ix - 3 + cc, 0, W - 1)];
            }
            const int Wc = W >> 1, Hc = H >> 1;
            for (int i = lane; i < 242; i += 64) {
                const int pl = i >= 121, j = i - pl * 121;
                const int rr = j / 11, cc = j - rr * 11;
                win_c[pl][rr * 12 + cc] = ref.p[1 + pl][size_t(clampi(icy - 3 + rr, 0, Hc - 1)) * ref.stride[1 + pl] +
                                                       clampi(icx - 3 + cc, 0, Wc - 1)];
            }
        }
        __syncthreads();
        {
            const int16_t *fh = kSubpel[kind_h][fx];
            for (int i = lane; i < 120; i += 64) {
                const int rr = i >> 3, cc = i & 7;
                int s = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) s += fh[t] * win_y[rr * 16 + cc + t];
                mid_y[i] = int16_t((s + 4) >> 3);
            }
            const int16_t *fhc = kSubpel[fih_c][fcx];
            for (int i = lane; i < 88; i += 64) {
                const int pl = i >= 44, j = i - pl * 44;
                const int rr = j >> 2, cc = j & 3;
                int s = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) s += fhc[t] * win_c[pl][rr * 12 + cc + t];
                mid_c[pl][j] = int16_t((s + 4) >> 3);
            }
        }
        __syncthreads();
        {
            const int16_t *fv = kSubpel[kind_v][fy];
            const int rr = lane >> 3, cc = lane & 7;
            int s = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) s += fv[t] * mid_y[(rr + t) * 8 + cc];
            acc_y += (s + ((1 << rnd1) >> 1)) >> rnd1;
            if (lane < 32) {
                const int16_t *fvc = kSubpel[fiv_c][fcy];
                const int pl = lane >> 4, j = lane & 15;
                const int r2 = j >> 2, c2 = j & 3;
                int sc = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) sc += fvc[t] * mid_c[pl][(r2 + t) * 4 + c2];
                acc_c += (sc + ((1 << rnd1) >> 1)) >> rnd1;
            }
        }
    }
    if (nref == 2) {
        acc_y = (acc_y + 16) >> 5;
        acc_c = (acc_c + 16) >> 5;
    }
    {
        const int rr = lane >> 3, cc = lane & 7;
        cur.p[0][size_t(y8 * 8 + rr) * cur.stride[0] + x8 * 8 + cc] = uint8_t(clampi(acc_y, 0, 255));
        if (lane < 32) {
            const int pl = lane >> 4, j = lane & 15;
            cur.p[1 + pl][size_t(y8 * 4 + (j >> 2)) * cur.stride[1 + pl] + x8 * 4 + (j & 3)] = uint8_t(clampi(acc_c, 0, 255));
        }
    }
}

void launch_inter_pred(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    uint64_t bytes = 0;
    bool any = false;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c] || !s.stats[c]->desc.is_inter) continue;
        any = true;
        // algorithmic bytes: reference samples fetched once per prediction + predicted samples written
        bytes += (s.stats[c]->ref_luma_px + s.stats[c]->inter_luma_px) * 3 / 2;
    }
    if (!any) return;
    ProfScope ps(prof, "inter_pred_kernel", bytes, st);
    hipLaunchKernelGGL(inter_pred_kernel, dim3((s.width / 8) * (s.height / 8), 1, s.batch.n), dim3(64), 0, st, s.batch);
}

}  // namespace ea1
