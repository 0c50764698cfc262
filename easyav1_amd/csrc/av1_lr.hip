// Loop restoration: Wiener and self-guided filters (AV1 spec 7.17; dav1d
// src/looprestoration_tmpl.c + lr_apply_tmpl.c, /root/reference/cmake/dav1d.cmake:1011,:1012).
//
// Input is the CDEF output, except that rows outside the current 64-luma-row stripe (stripes are
// offset by 8 rows) come from the DEBLOCKED picture, at most 2 rows deep (spec 7.17.6
// "get_source_sample").  A workgroup restores one TW x TH tile that never straddles a stripe or
// restoration-unit boundary (TH = 8 luma / 4 chroma rows; boundaries are multiples of those), so
// stripe and unit parameters are uniform per tile.  The tile plus a 3-pixel halo is staged in LDS
// through the source-selection rule once; both filters then run entirely out of LDS:
//   Wiener: 7-tap horizontal -> int16 tile -> 7-tap vertical (exact integer FIR; a candidate for
//           MFMA-i8 only at much larger tiles, not worth it at 7 taps: VALU, HBM/L2 bound)
//   SGR   : box sums (r = 2 and/or 1) -> a/b surfaces -> weighted 3x3 -> projection
// Traffic: read plane (+ halo rows, + 4 of 64 rows twice from the deblocked picture), write plane.
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const int16_t kSgrParams[16][4] = {EA1_SGR_PARAMS_VALUES};

struct LrPlane {
    const uint8_t *deblocked, *cdef;
    uint8_t *out;
    int stride, W, H, ss;
    const ea1_lr_unit *units;
    int ucols, urows, unit_log2;
};

// PLANES = 1: luma (blockIdx.z = chain); PLANES = 2: both chroma planes (blockIdx.z = 2*chain + plane)
template <int TW, int TH, int NT, int PLANES>
__global__ void __launch_bounds__(NT)
lr_kernel(Batch bt, int cdef_ran)
{
    const FrameView *fvp = bt.v[blockIdx.z / PLANES];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const int p = PLANES == 1 ? 0 : 1 + int(blockIdx.z % PLANES);
    LrPlane pl;
    pl.deblocked = fv.work.p[p];
    pl.cdef = cdef_ran ? fv.cdef.p[p] : fv.work.p[p];
    pl.out = fv.dst.p[p];
    pl.stride = fv.work.stride[p];
    pl.W = fv.work.pw(p);
    pl.H = fv.work.ph(p);
    pl.ss = p ? 1 : 0;
    pl.units = fv.lr[p];
    pl.ucols = fv.lr_cols[p];
    pl.urows = fv.lr_rows[p];
    pl.unit_log2 = fv.desc.lr_unit_log2[p];
    const bool restored = fv.desc.lr_type[p] && pl.units;
    if (int(blockIdx.x) * TW >= pl.W || int(blockIdx.y) * TH >= pl.H) return;
    constexpr int SW = TW + 6, SH = TH + 6;    // source tile with halo 3
    constexpr int AW = TW + 2, AH = TH + 2;    // a/b surfaces
    __shared__ uint8_t src[SH * SW];
    __shared__ int32_t buf[4 * AH * AW];        // Wiener: int16 mid tile; SGR: A0, B0, A1, B1
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int W = pl.W, H = pl.H;
    // stripe of this tile (uniform): luma row of its first line
    const int luma_y = y0 << pl.ss;
    const int stripe = (luma_y + 8) >> 6;
    const int s_start = (stripe * 64 - 8) >> pl.ss;   // arithmetic shift: -8 >> 1 == -4
    const int s_end = s_start + (64 >> pl.ss) - 1;
    const int usz = 1 << pl.unit_log2;
    int urow = (y0 + (8 >> pl.ss)) >> pl.unit_log2;
    urow = urow > pl.urows - 1 ? pl.urows - 1 : urow;
    int ucol = x0 >> pl.unit_log2;
    ucol = ucol > pl.ucols - 1 ? pl.ucols - 1 : ucol;
    (void)usz;
    ea1_lr_unit unit;
    unit.type = EA1_LR_NONE;
    if (restored) unit = pl.units[urow * pl.ucols + ucol];

    for (int i = tid; i < SH * SW; i += NT) {
        const int r = i / SW, c = i - r * SW;
        int x = x0 - 3 + c, y = y0 - 3 + r;
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        const uint8_t *from = pl.cdef;
        if (y < s_start) {
            y = y > s_start - 2 ? y : s_start - 2;
            from = pl.deblocked;
        } else if (y > s_end) {
            y = y < s_end + 2 ? y : s_end + 2;
            from = pl.deblocked;
        }
        src[i] = from[size_t(y) * pl.stride + x];
    }
    __syncthreads();

    if (unit.type == EA1_LR_WIENER) {
        int16_t *mid = reinterpret_cast<int16_t *>(buf);  // SH x TW
        int hf[7], vf[7];
        hf[3] = vf[3] = 128;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            vf[i] = vf[6 - i] = unit.coef[0][i];
            vf[3] -= 2 * unit.coef[0][i];
            hf[i] = hf[6 - i] = unit.coef[1][i];
            hf[3] -= 2 * unit.coef[1][i];
        }
        for (int i = tid; i < SH * TW; i += NT) {
            const int r = i / TW, c = i - r * TW;
            int s = 0;
#pragma unroll
            for (int t = 0; t < 7; t++) s += hf[t] * src[r * SW + c + t];
            s = (s + 4) >> 3;
            s = s < -8192 ? -8192 : (s > 8191 ? 8191 : s);
            mid[i] = int16_t(s);
        }
        __syncthreads();
        for (int i = tid; i < TH * TW; i += NT) {
            const int r = i / TW, c = i - r * TW;
            if (y0 + r >= H || x0 + c >= W) continue;
            int s = 0;
#pragma unroll
            for (int t = 0; t < 7; t++) s += vf[t] * mid[(r + t) * TW + c];
            s = (s + 1024) >> 11;
            pl.out[size_t(y0 + r) * pl.stride + x0 + c] = uint8_t(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    } else if (unit.type == EA1_LR_SGRPROJ) {
        const int16_t *prm = kSgrParams[unit.sgr_set & 15];
        int32_t *A[2] = {buf, buf + 2 * AH * AW};
        int32_t *B[2] = {buf + AH * AW, buf + 3 * AH * AW};
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int r = prm[2 * pass], eps = prm[2 * pass + 1];
            if (!r) continue;  // uniform
            const int n = (2 * r + 1) * (2 * r + 1);
            const unsigned n2e = unsigned(n * n * eps);
            const unsigned sc = ((1u << 20) + n2e / 2) / n2e;
            const unsigned one_by_n = ((1u << 12) + n / 2) / n;
            for (int i = tid; i < AH * AW; i += NT) {
                const int ar = i / AW, ac = i - ar * AW;  // surface index (ar-1, ac-1) relative to the tile
                // pass 0 (r = 2) only ever reads odd rows (relative to the tile): skip the others
                if (pass == 0 && (ar & 1)) continue;      // ar - 1 even  <=>  ar odd
                unsigned a = 0, b = 0;
                // centre of the box in src coordinates: (ar - 1 + 3, ac - 1 + 3)
                for (int dy = -r; dy <= r; dy++)
                    for (int dx = -r; dx <= r; dx++) {
                        const unsigned c = src[(ar + 2 + dy) * SW + ac + 2 + dx];
                        a += c * c;
                        b += c;
                    }
                const int pv = int(a * n) - int(b * b);
                const unsigned p = pv < 0 ? 0u : unsigned(pv);
                const unsigned z = (p * sc + (1u << 19)) >> 20;   // < 2^32: p*sc <= 1.02e7 * 2^20/(625*22)
                unsigned a2;
                if (z >= 255) a2 = 256;
                else if (z == 0) a2 = 1;
                else a2 = ((z << 8) + z / 2) / (z + 1);
                const unsigned b2 = (256 - a2) * b * one_by_n;
                A[pass][i] = int(a2);
                B[pass][i] = int((b2 + (1u << 11)) >> 12);
            }
        }
        __syncthreads();
        const int w0 = unit.coef[0][0], w1 = unit.coef[0][1], w2 = 128 - w0 - w1;
        for (int i = tid; i < TH * TW; i += NT) {
            const int r = i / TW, c = i - r * TW;
            if (y0 + r >= H || x0 + c >= W) continue;
            const int px = src[(r + 3) * SW + c + 3];
            const int u = px << 4;
            int v = w1 * u;
            // pass 0 (r0)
            if (prm[0]) {
                int a = 0, b = 0;
                if (r & 1) {
                    // odd row: only its own a/b row, weights 5 6 5
                    const int base = (r + 1) * AW + c + 1;
                    a = 5 * A[0][base - 1] + 6 * A[0][base] + 5 * A[0][base + 1];
                    b = 5 * B[0][base - 1] + 6 * B[0][base] + 5 * B[0][base + 1];
                    v += w0 * ((a * px + b + (1 << 7)) >> 8);
                } else {
                    const int up = r * AW + c + 1, dn = (r + 2) * AW + c + 1;
                    a = 5 * (A[0][up - 1] + A[0][up + 1] + A[0][dn - 1] + A[0][dn + 1]) + 6 * (A[0][up] + A[0][dn]);
                    b = 5 * (B[0][up - 1] + B[0][up + 1] + B[0][dn - 1] + B[0][dn + 1]) + 6 * (B[0][up] + B[0][dn]);
                    v += w0 * ((a * px + b + (1 << 8)) >> 9);
                }
            } else {
                v += w0 * u;
            }
            if (prm[2]) {
                const int m = (r + 1) * AW + c + 1;
                const int a = 4 * (A[1][m] + A[1][m - 1] + A[1][m + 1] + A[1][m - AW] + A[1][m + AW]) +
                              3 * (A[1][m - AW - 1] + A[1][m - AW + 1] + A[1][m + AW - 1] + A[1][m + AW + 1]);
                const int b = 4 * (B[1][m] + B[1][m - 1] + B[1][m + 1] + B[1][m - AW] + B[1][m + AW]) +
                              3 * (B[1][m - AW - 1] + B[1][m - AW + 1] + B[1][m + AW - 1] + B[1][m + AW + 1]);
                v += w2 * ((a * px + b + (1 << 8)) >> 9);
            } else {
                v += w2 * u;
            }
            const int s = (v + (1 << 10)) >> 11;
            pl.out[size_t(y0 + r) * pl.stride + x0 + c] = uint8_t(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    } else {
        for (int i = tid; i < TH * TW; i += NT) {
            const int r = i / TW, c = i - r * TW;
            if (y0 + r >= H || x0 + c >= W) continue;
            pl.out[size_t(y0 + r) * pl.stride + x0 + c] = src[(r + 3) * SW + c + 3];
        }
    }
}

void launch_lr(const SubmitInfo &s, hipStream_t st, Prof *prof, int cdef_ran)
{
    const int n = s.batch.n;
    uint64_t frames = 0;
    for (int c = 0; c < n; c++)
        if (s.stats[c]) frames++;
    if (!frames) return;
    const uint64_t ybytes = uint64_t(s.width) * s.height;
    {
        ProfScope ps(prof, "lr_kernel<64,8,256,1>", frames * 2 * ybytes, st);
        hipLaunchKernelGGL((lr_kernel<64, 8, 256, 1>), dim3((s.width + 63) / 64, (s.height + 7) / 8, n), dim3(256), 0, st, s.batch, cdef_ran);
    }
    ProfScope ps(prof, "lr_kernel<32,4,128,2>", frames * ybytes, st);
    hipLaunchKernelGGL((lr_kernel<32, 4, 128, 2>), dim3((s.width / 2 + 31) / 32, (s.height / 2 + 3) / 4, 2 * n), dim3(128), 0, st,
                       s.batch, cdef_ran);
}

}  // namespace ea1
