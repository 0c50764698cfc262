// Inter prediction: separable 8-tap sub-pel interpolation + compound average.
// Arithmetic: AV1 spec 7.11.3.4 / dav1d src/mc_tmpl.c put/prep_8tap + avg (listed at
// /root/reference/cmake/dav1d.cmake:1013): horizontal Round2(sum, 3), vertical Round2(sum, 11)
// for a single reference, Round2(sum, 7) per reference and Round2(a + b, 5) for the average.
// Each pass is one exact integer sum and one rounding, identical to dav1d's fused shortcuts.
//
// Mapping: one 64-lane workgroup per 8x8 luma unit (+ its two 4x4 chroma units); blocks larger than
// 8x8 are simply several units with the same motion vector (the filter is position independent).
//   window  : 15x15 luma / 11x11 chroma reference samples, fetched as ALIGNED dwords (5 resp. 4 per
//             row) straight into LDS; windows that touch the picture border take a byte-wise path
//             with clamped coordinates (= dav1d's emu_edge)
//   H pass  : a lane produces 4 adjacent intermediates from 4 LDS dwords (byte-aligned with
//             v_alignbyte), 52 lanes cover luma + both chroma planes in one step
//   V pass  : a lane produces 4 adjacent outputs from 8 8-byte LDS reads and stores them as one dword
// Traffic: reference samples ~once (overlaps are L2 hits), prediction written once.
// The multiply-adds of both passes run as v_fma_f32: on gfx950 that instruction issues at the full rate, the integer
// v_mad_i32_i24 (and every dot / packed-16 alternative) at half of it (profiles/r02_a_valu_issue_rates.txt), and
// float intermediates need no 16-bit packing / unpacking around LDS.  Every value is an integer far below 2^24
// (H: |sum| <= 255 * sum|taps| < 2^16; V: |sum| < 2^13 * 2^8), so each product, each partial sum and each
// floor(x * 2^-k + 0.5) is exact: the results are the integer results, bit for bit (the parity tests run unchanged).
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const int16_t kSubpel[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};
// the same taps as floats (wave-uniform loads land in scalar registers, which v_fma_f32 takes directly)
__device__ __constant__ static const float kSubpelF[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};

// floor(v * 2^-SH + 0.5) = (v + (1 << (SH - 1))) >> SH for the integer v held in a float
template <int SH>
__device__ __forceinline__ float round_shift_f(float v)
{
    return __builtin_floorf(__builtin_fmaf(v, 1.0f / float(1 << SH), 0.5f));
}

// the 11 bytes b[0..10] of three dwords as floats (v_cvt_f32_ubyteN)
__device__ __forceinline__ void window11f(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, int o, float (&b)[12])
{
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, o);
    const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, o);
    const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, o);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        b[i] = float((w0 >> (8 * i)) & 255u);
        b[4 + i] = float((w1 >> (8 * i)) & 255u);
        b[8 + i] = float((w2 >> (8 * i)) & 255u);
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// see av1_lr.hip: keeps hipcc from selecting v_ashr_pk_u8_i32 (stale upper bits on gfx950)
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// 12 consecutive bytes starting at byte `o` (0..3) of the 16-byte sequence d0..d3
__device__ __forceinline__ void window12(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, int o, int (&b)[12])
{
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, o);
    const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, o);
    const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, o);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        b[i] = (w0 >> (8 * i)) & 255;
        b[4 + i] = (w1 >> (8 * i)) & 255;
        b[8 + i] = (w2 >> (8 * i)) & 255;
    }
}

// stage a (rows x 4*ND bytes) window whose first wanted sample is (wx0, wy0); returns the byte
// offset of that sample inside the staged rows
template <int ROWS, int ND, int NEED>
__device__ __forceinline__ int stage_window(uint32_t *lds, const uint8_t *__restrict__ src, int stride, int W, int H,
                                            int wx0, int wy0, int lane, int lane0, int nlanes)
{
    const bool inside = wx0 >= 0 && wy0 >= 0 && wx0 + NEED <= W && wy0 + ROWS <= H;
    if (inside) {
        const int ax = wx0 & ~3;
        for (int i = lane - lane0; i < ROWS * ND; i += nlanes) {
            if (i < 0) break;
            const int r = i / ND, k = i - r * ND;
            lds[i] = *reinterpret_cast<const uint32_t *>(src + size_t(wy0 + r) * stride + ax + 4 * k);
        }
        return wx0 - ax;
    }
    uint8_t *lb = reinterpret_cast<uint8_t *>(lds);
    for (int i = lane - lane0; i < ROWS * NEED; i += nlanes) {
        if (i < 0) break;
        const int r = i / NEED, c = i - r * NEED;
        lb[r * ND * 4 + c] = src[size_t(clampi(wy0 + r, 0, H - 1)) * stride + clampi(wx0 + c, 0, W - 1)];
    }
    return 0;
}

// One workgroup predicts one TS x TS luma tile (+ its two TS/2 chroma tiles) that shares a single
// motion-vector set: blocks of 32 and 64 are cut into 32x32 tiles (256 lanes), 16x16 and 8x8 blocks
// are one tile each (64 lanes).  Bigger tiles amortise the 7-sample filter apron (a 32x32 tile
// filters 39 rows for 32, an 8x8 unit 15 for 8) and keep every lane busy in both passes.
// EXT = true: the tile list with distance-weighted / wedge / difference-weighted compound (AV1 spec
// 7.11.3.12, 7.11.3.14, 7.11.3.15): both predictions are kept apart and blended with weights or
// with a per-sample mask; the luma mask of the tile is staged in LDS so that the chroma tasks can
// average its 2x2 cells.
template <int TS, int NT, int CLASS, bool EXT>
__global__ void __launch_bounds__(NT)
mc_tile_kernel(Batch bt)
{
    constexpr int CS = TS / 2;
    constexpr int GY = TS / 4, GC = CS / 4 > 0 ? CS / 4 : 1;
    constexpr int RY = TS + 7, RC = CS + 7;
    constexpr int NDY = (TS + 13) / 4, NDC = (CS + 13) / 4;
    constexpr int NHY = RY * GY, NHC = RC * GC;
    constexpr int NVY = TS * GY, NVC = CS * GC, NV = NVY + 2 * NVC;
    constexpr int VPT = (NV + NT - 1) / NT;
    __shared__ uint32_t win_y[RY * NDY];
    __shared__ uint32_t win_c[2][RC * NDC];
    __shared__ __attribute__((aligned(16))) float mid_y[RY * TS];
    __shared__ __attribute__((aligned(16))) float mid_c[2][RC * CS];

    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    if (fv.scaled_mask) return;   // frames with a scaled reference: mc_scaled_kernel
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private
    // L2), so workgroup b gets tile (b % 8) * ceil(n / 8) + b / 8: every XCD walks one contiguous
    // eighth of the (spatially ordered) tile list and neighbouring windows share that XCD's L2
    // instead of being fetched by all eight.  Placement only affects speed, never results.
    const int count = EXT ? fv.mc_ext_count[CLASS] : fv.mc_count[CLASS];
    const int per_xcd = (int(gridDim.x) + 7) >> 3;
    const int tile = int(blockIdx.x & 7) * per_xcd + int(blockIdx.x >> 3);
    if (tile >= count) return;
    const McItem *itp = (EXT ? fv.mc_ext_items[CLASS] : fv.mc_items[CLASS]) + tile;
    const uint4 raw = *reinterpret_cast<const uint4 *>(itp);
    uint4 raw2 = make_uint4(0, 0, 0, 0);
    if constexpr (EXT) raw2 = reinterpret_cast<const uint4 *>(itp)[1];
    const int comp_type = raw2.y & 0xFF, comp_arg = (raw2.y >> 8) & 0xFF;
    const int x8 = raw.x & 0xFFFF, y8 = raw.x >> 16;
    const bool compound = (raw.y >> 1) & 1;
    const int kind_h = (raw.y >> 8) & 3, kind_v = (raw.y >> 10) & 3;
    // filter set: bilinear is set 5; a 4-sample side of the predicted block takes the 4-tap sets
    // (only the chroma of blocks with an 8-sample luma side: item flags bit 2 / 3)
    const int fih_y = kind_h == EA1_FILTER_BILINEAR ? 5 : kind_h, fiv_y = kind_v == EA1_FILTER_BILINEAR ? 5 : kind_v;
    constexpr bool C4 = TS == 8;   // only tiles of the 8-sample class can have a 4-sample chroma side
    const int fih_c = kind_h == EA1_FILTER_BILINEAR ? 5 : ((C4 && (raw.y & 4)) ? (kind_h == EA1_FILTER_SMOOTH ? 4 : 3) : kind_h);
    const int fiv_c = kind_v == EA1_FILTER_BILINEAR ? 5 : ((C4 && (raw.y & 8)) ? (kind_v == EA1_FILTER_SMOOTH ? 4 : 3) : kind_v);
    const int nref = compound ? 2 : 1;
    const int tid = threadIdx.x;

    float acc[VPT][4], acc1[EXT ? VPT : 1][4];
#pragma unroll
    for (int k = 0; k < VPT; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = 0.0f;
#pragma unroll
    for (int k = 0; k < (EXT ? VPT : 1); k++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc1[k][j] = 0.0f;

    for (int r = 0; r < nref; r++) {
        const int slot = (raw.y >> (16 + 8 * r)) & 7;
        const uint32_t mvw = r ? raw.w : raw.z;
        const int mvy = int16_t(mvw & 0xFFFF), mvx = int16_t(mvw >> 16);
        const DevPic &ref = fv.slots[slot];
        const int px = ((x8 * 8) << 4) + 2 * mvx, py = ((y8 * 8) << 4) + 2 * mvy;
        const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
        const int pcx = ((x8 * 4) << 4) + mvx, pcy = ((y8 * 4) << 4) + mvy;
        const int icx = pcx >> 4, icy = pcy >> 4, fcx = pcx & 15, fcy = pcy & 15;
        if (r) __syncthreads();
        const int oy = stage_window<RY, NDY, RY>(win_y, ref.p[0], ref.stride[0], ref.w, ref.h, ix - 3, iy - 3, tid, 0, NT);
        const int ou = stage_window<RC, NDC, RC>(win_c[0], ref.p[1], ref.stride[1], ref.pw(1), ref.ph(1), icx - 3, icy - 3, tid, 0, NT);
        const int ov = stage_window<RC, NDC, RC>(win_c[1], ref.p[2], ref.stride[2], ref.pw(2), ref.ph(2), icx - 3, icy - 3, tid, 0, NT);
        __syncthreads();
        // horizontal pass: 4 intermediates per task.  Luma and chroma run as separate loops so that
        // the 8 taps are wave-uniform (scalar registers) instead of per-lane table loads.
        {
            float f[8];
#pragma unroll
            for (int t = 0; t < 8; t++) f[t] = kSubpelF[fih_y][fx][t];
            for (int h = tid; h < NHY; h += NT) {
                const int row = h / GY, g = h - row * GY;
                const uint32_t *rowp = win_y + row * NDY + g;
                float bb[12];
                window11f(rowp[0], rowp[1], rowp[2], rowp[3], oy, bb);
                float4 m;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float s = 0.0f;
#pragma unroll
                    for (int t = 0; t < 8; t++) s = __builtin_fmaf(f[t], bb[j + t], s);
                    (&m.x)[j] = round_shift_f<3>(s);
                }
                *reinterpret_cast<float4 *>(mid_y + row * TS + 4 * g) = m;
            }
#pragma unroll
            for (int t = 0; t < 8; t++) f[t] = kSubpelF[fih_c][fcx][t];
            for (int h = tid; h < 2 * NHC; h += NT) {
                const int pl = h >= NHC;
                const int hh = h - pl * NHC;
                const int row = hh / GC, g = hh - row * GC;
                const uint32_t *rowp = win_c[pl] + row * NDC + g;
                float bb[12];
                window11f(rowp[0], rowp[1], rowp[2], rowp[3], pl ? ov : ou, bb);
                float4 m;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float s = 0.0f;
#pragma unroll
                    for (int t = 0; t < 8; t++) s = __builtin_fmaf(f[t], bb[j + t], s);
                    (&m.x)[j] = round_shift_f<3>(s);
                }
                *reinterpret_cast<float4 *>(mid_c[pl] + row * CS + 4 * g) = m;
            }
        }
        __syncthreads();
        // vertical pass: 4 outputs per task, accumulated in the owning lane
        {
            float fl[8], fc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                fl[t] = kSubpelF[fiv_y][fy][t];
                fc[t] = kSubpelF[fiv_c][fcy][t];
            }
            // when the luma tasks fill whole passes over the workgroup (32x32 and 16x16 tiles), a pass is all luma
            // or all chroma and the taps stay in scalar registers; the 8x8 class mixes both in one wave
            constexpr bool kUniform = NVY % NT == 0;
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * NT;
                if (v >= NV) continue;
                const bool luma = kUniform ? (k * NT < NVY) : (v < NVY);
                int row, g, pl = 0;
                if (luma) {
                    row = v / GY;
                    g = v - row * GY;
                } else {
                    const int vc = v - NVY;
                    pl = vc >= NVC;
                    const int vv = vc - pl * NVC;
                    row = vv / GC;
                    g = vv - row * GC;
                }
                const float *col = luma ? mid_y + row * TS + 4 * g : mid_c[pl] + row * CS + 4 * g;
                const int rstride = luma ? TS : CS;
                float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const float4 w = *reinterpret_cast<const float4 *>(col + t * rstride);
                    const float ft = luma ? fl[t] : fc[t];
                    s[0] = __builtin_fmaf(ft, w.x, s[0]);
                    s[1] = __builtin_fmaf(ft, w.y, s[1]);
                    s[2] = __builtin_fmaf(ft, w.z, s[2]);
                    s[3] = __builtin_fmaf(ft, w.w, s[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float val = compound ? round_shift_f<7>(s[j]) : round_shift_f<11>(s[j]);
                    if (EXT && r) acc1[EXT ? k : 0][j] = val;
                    else acc[k][j] += val;
                }
            }
        }
    }
    const DevPic &cur = fv.work;
    uint8_t *lmask = reinterpret_cast<uint8_t *>(mid_y);   // EXT: luma mask of the tile, TS x TS bytes (mid_y holds 4 per sample)
    if constexpr (EXT) {
        if (comp_type >= EA1_COMP_WEDGE) {
            __syncthreads();   // the vertical pass is done with mid_y
            if (comp_type == EA1_COMP_WEDGE) {
                const uint8_t *wm = fv.wedge + raw2.x;
                const int ms = 1 << ((raw2.y >> 16) & 0xFF);
                for (int i = tid; i < TS * TS; i += NT) lmask[i] = wm[(i / TS) * ms + (i % TS)];
            } else {
#pragma unroll
                for (int k = 0; k < VPT; k++) {
                    const int v = tid + k * NT;
                    if (v >= NVY) continue;
                    const int row = v / GY, g = v - row * GY;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int d = int(acc[k][j]) - int(acc1[EXT ? k : 0][j]);
                        const int diff = ((d < 0 ? -d : d) + 8) >> 4;
                        const int m = clampi(38 + diff / 16, 0, 64);
                        lmask[row * TS + 4 * g + j] = uint8_t(comp_arg ? 64 - m : m);
                    }
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int v = tid + k * NT;
        if (v >= NV) continue;
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (!EXT) {
                // clamp to 0..255 and byte insert in one instruction (v_cvt_pk_u8_f32 saturates)
                const float val = compound ? round_shift_f<5>(acc[k][j]) : acc[k][j];
                pk = __builtin_amdgcn_cvt_pk_u8_f32(val, j, pk);
                continue;
            }
            int val;
            if constexpr (EXT) {
                const int a = int(acc[k][j]), b = int(acc1[EXT ? k : 0][j]);
                if (comp_type == EA1_COMP_DIST) {
                    val = (a * comp_arg + b * (16 - comp_arg) + 128) >> 8;
                } else if (comp_type == EA1_COMP_AVG) {
                    val = (a + b + 16) >> 5;
                } else {
                    int m;
                    if (v < NVY) {
                        const int row = v / GY, g = v - row * GY;
                        m = lmask[row * TS + 4 * g + j];
                    } else {
                        const int vc = v - NVY;
                        const int vv = vc >= NVC ? vc - NVC : vc;
                        const int row = vv / GC, g = vv - row * GC;
                        const uint8_t *q = lmask + (2 * row) * TS + 2 * (4 * g + j);
                        m = (q[0] + q[1] + q[TS] + q[TS + 1] + 2) >> 2;
                    }
                    val = (m * a + (64 - m) * b + 512) >> 10;
                }
            } else {
                val = 0;
            }
            val = opaque(val);
            pk |= uint32_t(clampi(val, 0, 255)) << (8 * j);
        }
        if (v < NVY) {
            const int row = v / GY, g = v - row * GY;
            *reinterpret_cast<uint32_t *>(cur.p[0] + size_t(y8 * 8 + row) * cur.stride[0] + x8 * 8 + 4 * g) = pk;
        } else {
            const int vc = v - NVY;
            const int pl = vc >= NVC;
            const int vv = vc - pl * NVC;
            const int row = vv / GC, g = vv - row * GC;
            *reinterpret_cast<uint32_t *>(cur.p[1 + pl] + size_t(y8 * 4 + row) * cur.stride[1 + pl] + x8 * 4 + 4 * g) = pk;
        }
    }
}

// ---- scaled references (AV1 spec 7.11.3.3 motion vector scaling, 7.11.3.4 block inter prediction) ----
// A frame that reads a reference of another size sends ALL its translated predictions here (with xScale = 1 << 14 the
// formulas below reduce to the unscaled ones, so its same-size references take the same path).  Positions are in
// 1/1024 sample units and step from the BLOCK's top-left sample: startX = Round2Signed(origX * xScale - (8 << 14), 8) + 32
// with origX = (x << 4) + ((2 mv) >> subX) + 8, step = Round2Signed(xScale, 4); sample (r, c) of the block filters the
// reference around column (startX + c * xStep) >> 10 with the sub-pel filter ((startX + c * xStep) >> 6) & 15, rows
// likewise from startY & 1023 -- the 8 intermediate rows of a sample are recomputed per sample (one lane = one sample:
// the tool is rare, the arithmetic is what matters), each Round2(., 3) as in the shared-intermediate form.
__device__ __forceinline__ int scaled_sample_sum(const DevPic &ref, int pl, int x0, int y0, int c, int r, int mvy, int mvx, int fih, int fiv, int xs, int ys);   // below, with the per-sample kernels

// `all`: every frame of the batch comes here, scaled references or not (4:4:4 / 4:2:2 contexts: mc_tile_kernel is the 4:2:0 kernel)
__global__ void __launch_bounds__(256)
mc_scaled_kernel(Batch bt, int all)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    if (!fv.scaled_mask && !all) return;
    const bool ext = blockIdx.y >= 3;   // the lists with distance-weighted / wedge / difference-weighted compound
    const int cls = ext ? blockIdx.y - 3 : blockIdx.y, TS = 8 << cls;
    if (int(blockIdx.x) >= (ext ? fv.mc_ext_count[cls] : fv.mc_count[cls])) return;
    const McItem it = (ext ? fv.mc_ext_items[cls] : fv.mc_items[cls])[blockIdx.x];
    const bool compound = (it.flags >> 1) & 1;
    const int kind_h = it.filt & 3, kind_v = (it.filt >> 2) & 3;
    const int nref = compound ? 2 : 1;
    const DevPic &cur = fv.work;
    const int csx = fv.ssx, csy = fv.ssy;
    const int nluma = TS * TS, nchroma = (TS >> csx) * (TS >> csy);
    for (int i = threadIdx.x; i < nluma + 2 * nchroma; i += 256) {
        int pl = 0, k = i;
        if (k >= nluma) {
            k -= nluma;
            pl = 1 + (k >= nchroma);
            k -= (pl - 1) * nchroma;
        }
        const int sx_ = pl ? csx : 0, sy_ = pl ? csy : 0, tw = TS >> sx_;
        const int ty = k / tw, tx = k - ty * tw;
        // this sample's offset from the block's corner, in the plane (positions step from the corner: item.ox8 / oy8)
        const int c = ((it.ox8 * 8) >> sx_) + tx, r = ((it.oy8 * 8) >> sy_) + ty;
        // filter sets as in the unscaled kernel: bilinear is set 5; a 4-sample side of the predicted block takes the 4-tap sets
        const bool w4 = pl && (it.flags & 4), h4 = pl && (it.flags & 8);
        const int fih = kind_h == EA1_FILTER_BILINEAR ? 5 : (w4 ? (kind_h == EA1_FILTER_SMOOTH ? 4 : 3) : kind_h);
        const int fiv = kind_v == EA1_FILTER_BILINEAR ? 5 : (h4 ? (kind_v == EA1_FILTER_SMOOTH ? 4 : 3) : kind_v);
        // one prediction of this sample (or of another sample of the block: the masks of chroma samples need luma ones)
        auto pred = [&](int q, int ppl, int pc, int pr, int pfih, int pfiv, int rnd) {
            const int slot = it.ref[q] & 7;
            const int psx = ppl ? csx : 0, psy = ppl ? csy : 0;
            const int sum = scaled_sample_sum(fv.slots[slot], ppl, (it.x8 * 8 - it.ox8 * 8) >> psx, (it.y8 * 8 - it.oy8 * 8) >> psy, pc, pr, it.mv[q][0], it.mv[q][1], pfih, pfiv,
                                              fv.ref_scale[slot][0], fv.ref_scale[slot][1]);
            return (sum + ((1 << rnd) >> 1)) >> rnd;
        };
        int val;
        if (!ext || it.comp_type == EA1_COMP_AVG) {
            int acc = 0;
            for (int q = 0; q < nref; q++) acc += pred(q, pl, c, r, fih, fiv, compound ? 7 : 11);
            val = compound ? (acc + 16) >> 5 : acc;
        } else {
            const int a = pred(0, pl, c, r, fih, fiv, 7), b = pred(1, pl, c, r, fih, fiv, 7);
            if (it.comp_type == EA1_COMP_DIST) {
                val = (a * it.comp_arg + b * (16 - it.comp_arg) + 128) >> 8;
            } else {
                // mask of a luma sample of the tile (tile coordinates): the wedge table, or from the two luma predictions
                const int lfih = kind_h == EA1_FILTER_BILINEAR ? 5 : kind_h, lfiv = kind_v == EA1_FILTER_BILINEAR ? 5 : kind_v;
                auto luma_mask = [&](int ly, int lx) {
                    if (it.comp_type == EA1_COMP_WEDGE) return int(fv.wedge[it.mask_off + (ly << it.mask_stride_log2) + lx]);
                    const int la = pred(0, 0, it.ox8 * 8 + lx, it.oy8 * 8 + ly, lfih, lfiv, 7), lb = pred(1, 0, it.ox8 * 8 + lx, it.oy8 * 8 + ly, lfih, lfiv, 7);
                    const int d = la - lb;
                    const int mm = clampi(38 + ((((d < 0 ? -d : d) + 8) >> 4) / 16), 0, 64);
                    return it.comp_arg ? 64 - mm : mm;
                };
                // (7.11.3.14: the luma mask averaged over the luma samples the chroma sample covers)
                const int m = !(sx_ | sy_) ? luma_mask(ty, tx)
                              : (!sy_ ? (luma_mask(ty, 2 * tx) + luma_mask(ty, 2 * tx + 1) + 1) >> 1
                                      : (luma_mask(2 * ty, 2 * tx) + luma_mask(2 * ty, 2 * tx + 1) + luma_mask(2 * ty + 1, 2 * tx) + luma_mask(2 * ty + 1, 2 * tx + 1) + 2) >> 2);
                val = (m * a + (64 - m) * b + 512) >> 10;
            }
        }
        val = opaque(val);
        const int x = (it.x8 * 8 >> sx_) + tx, y = (it.y8 * 8 >> sy_) + ty;
        cur.p[pl][size_t(y) * cur.stride[pl] + x] = uint8_t(clampi(val, 0, 255));
    }
}

template <int TS, int NT, int CLASS, bool EXT>
static void launch_mc_class(const SubmitInfo &s, hipStream_t st, Prof *prof, const char *name)
{
    int max_count = 0;
    uint64_t bytes = 0;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c] || !s.stats[c]->desc.is_inter) continue;
        const int cnt = EXT ? s.stats[c]->mc_ext_count[CLASS] : s.stats[c]->mc_count[CLASS];
        max_count = std::max(max_count, cnt);
        // algorithmic bytes: reference samples fetched once per prediction + predicted samples written
        // (per tile: 1.5 * TS^2 * (refs + 1); refs averaged over the frame, 2 for the extended list)
        const double rbar = EXT ? 2.0 : (s.stats[c]->inter_luma_px ? double(s.stats[c]->ref_luma_px) / double(s.stats[c]->inter_luma_px) : 1.0);
        bytes += uint64_t(1.5 * TS * TS * (rbar + 1.0) * cnt);
    }
    if (!max_count) return;
    ProfScope ps(prof, name, bytes, st);
    hipLaunchKernelGGL((mc_tile_kernel<TS, NT, CLASS, EXT>), dim3((max_count + 7) / 8 * 8, 1, s.batch.n), dim3(NT), 0, st, s.batch);
}

// ---- overlapped block motion compensation (AV1 spec 7.11.3.10) ----
// One workgroup per strip; every lane predicts single samples straight from the reference picture
// (two-pass 8-tap written out per sample: the strips are thin -- half a block at most -- and the
// tool is rare, so no staging) and blends them into the block's own prediction with the 1-D mask
// of the overlap length.
__device__ __constant__ static const uint8_t kObmcMask[62] = {EA1_OBMC_MASK_VALUES};

__device__ __forceinline__ int filter_set(int kind, int dim)
{
    if (kind == EA1_FILTER_BILINEAR) return 5;
    return dim <= 4 ? (kind == EA1_FILTER_SMOOTH ? 4 : 3) : kind;
}

__device__ __forceinline__ int round2_signed(int64_t v, int n)
{
    return v >= 0 ? int((v + (int64_t(1) << (n - 1))) >> n) : -int(((-v) + (int64_t(1) << (n - 1))) >> n);
}

// One sample of a prediction whose top-left sample is (x0, y0) of plane pl, at offset (c, r) from it, through a
// reference of another size (AV1 spec 7.11.3.3 / 7.11.3.4; xs, ys: scale factors in 1/16384): the sum of the vertical
// pass BEFORE its rounding (the caller rounds by 11, or by 7 for a compound half).
__device__ __forceinline__ int scaled_sample_sum(const DevPic &ref, int pl, int x0, int y0, int c, int r, int mvy, int mvx, int fih, int fiv, int xs, int ys)
{
    const int sx_ = pl ? ref.ssx : 0, sy_ = pl ? ref.ssy : 0;
    const int64_t orig_x = (int64_t(x0) << 4) + ((2 * mvx) >> sx_) + 8, orig_y = (int64_t(y0) << 4) + ((2 * mvy) >> sy_) + 8;
    const int start_x = round2_signed(orig_x * xs - (int64_t(8) << 14), 8) + 32;
    const int start_y = round2_signed(orig_y * ys - (int64_t(8) << 14), 8) + 32;
    const int step_x = round2_signed(xs, 4), step_y = round2_signed(ys, 4);
    const int px = start_x + step_x * c;
    const int py = (start_y & 1023) + step_y * r;
    const int col0 = (px >> 10) - 3, row0 = (start_y >> 10) + (py >> 10) - 3;
    const int16_t *fh = kSubpel[fih][(px >> 6) & 15], *fvt = kSubpel[fiv][(py >> 6) & 15];
    const int W = ref.pw(pl), H = ref.ph(pl), rs = ref.stride[pl];
    const uint8_t *plane = ref.p[pl];
    int sum = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint8_t *row = plane + size_t(clampi(row0 + t, 0, H - 1)) * rs;
        int hsum = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) hsum += fh[u] * row[clampi(col0 + u, 0, W - 1)];
        sum += fvt[t] * ((hsum + 4) >> 3);
    }
    return sum;
}

__global__ void __launch_bounds__(256)
obmc_kernel(Batch bt, int pass)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || int(blockIdx.x) >= fvp->obmc_count[pass]) return;
    const FrameView &fv = *fvp;
    const ObmcItem it = fv.obmc_items[pass][blockIdx.x];
    const DevPic &ref = fv.slots[it.ref];
    const int nplanes = (it.flags & 1) ? 3 : 1;
    for (int pl = 0; pl < nplanes; pl++) {
        const int sx_ = pl ? fv.ssx : 0, sy_ = pl ? fv.ssy : 0;
        const int w = it.w >> sx_, h = it.h >> sy_, x0 = it.x >> sx_, y0 = it.y >> sy_;
        const int W = ref.pw(pl), H = ref.ph(pl), rs = ref.stride[pl];
        const uint8_t *src = ref.p[pl];
        const int px = (x0 << 4) + (sx_ ? it.mv[1] : it.mv[1] * 2), py = (y0 << 4) + (sy_ ? it.mv[0] : it.mv[0] * 2);
        const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
        const int16_t *fh = kSubpel[filter_set(it.filt & 3, w)][fx];
        const int16_t *fvt = kSubpel[filter_set((it.filt >> 2) & 3, h)][fy];
        const uint8_t *om = kObmcMask + (pass == 0 ? h : w) - 2;
        uint8_t *dstp = fv.work.p[pl];
        const int ds = fv.work.stride[pl];
        const bool scaled = (fv.scaled_mask >> it.ref) & 1u;   // the neighbour's reference has another size
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int r = i / w, c = i - r * w;
            int acc = 0;
            if (scaled) {
                acc = scaled_sample_sum(ref, pl, x0, y0, c, r, it.mv[0], it.mv[1], filter_set(it.filt & 3, w), filter_set((it.filt >> 2) & 3, h),
                                        fv.ref_scale[it.ref][0], fv.ref_scale[it.ref][1]);
            } else {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int sy = clampi(iy + r + t - 3, 0, H - 1);
                    int sum = 0;
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += fh[u] * src[size_t(sy) * rs + clampi(ix + c + u - 3, 0, W - 1)];
                    acc += fvt[t] * ((sum + 4) >> 3);
                }
            }
            const int pred = clampi((acc + 1024) >> 11, 0, 255);
            const int m = pass == 0 ? om[r] : om[c];
            uint8_t *q = dstp + size_t(y0 + r) * ds + x0 + c;
            *q = uint8_t((m * *q + (64 - m) * pred + 32) >> 6);
        }
    }
}

// ---- small predictions of blocks with a 4-sample luma side (AV1 spec 7.11.3.1) ----
// one 64-lane workgroup per piece (<= 64 samples), one sample per lane, two-pass filter per sample
__global__ void __launch_bounds__(64)
mc_piece_kernel(Batch bt)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || int(blockIdx.x) >= fvp->piece_count) return;
    const FrameView &fv = *fvp;
    const PieceItem it = fv.piece_items[blockIdx.x];
    const int pl = it.plane, sx_ = pl ? fv.ssx : 0, sy_ = pl ? fv.ssy : 0;
    const DevPic &ref = fv.slots[it.ref];
    const int W = ref.pw(pl), H = ref.ph(pl), rs = ref.stride[pl];
    const uint8_t *src = ref.p[pl];
    const int w = it.w, h = it.h;
    const int px = (int(it.x) << 4) + (sx_ ? it.mv[1] : it.mv[1] * 2), py = (int(it.y) << 4) + (sy_ ? it.mv[0] : it.mv[0] * 2);
    const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
    const int16_t *fh = kSubpel[filter_set(it.filt & 3, w)][fx];
    const int16_t *fvt = kSubpel[filter_set((it.filt >> 2) & 3, h)][fy];
    const int i = threadIdx.x;
    if (i >= w * h) return;
    const int r = i / w, c = i - r * w;
    int acc = 0;
    if ((fv.scaled_mask >> it.ref) & 1u) {
        acc = scaled_sample_sum(ref, pl, it.x, it.y, c, r, it.mv[0], it.mv[1], filter_set(it.filt & 3, w), filter_set((it.filt >> 2) & 3, h),
                                fv.ref_scale[it.ref][0], fv.ref_scale[it.ref][1]);
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int sy = clampi(iy + r + t - 3, 0, H - 1);
            int sum = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) sum += fh[u] * src[size_t(sy) * rs + clampi(ix + c + u - 3, 0, W - 1)];
            acc += fvt[t] * ((sum + 4) >> 3);
        }
    }
    fv.work.p[pl][size_t(it.y + r) * fv.work.stride[pl] + it.x + c] = uint8_t(clampi((acc + 1024) >> 11, 0, 255));
}

// ---- warped motion (AV1 spec 7.11.3.5 block warp process; dav1d warp8x8 / warp8x8t) ----
// One 64-lane workgroup per item: a rectangle of at most 8x8 samples of one plane with one or two
// predictions.  A warped prediction filters 15 rows x 8 columns horizontally (two per lane, taps
// picked per sample from the position sx4 + alpha * column + beta * row), keeps the 15 x 8
// intermediates in LDS and filters them vertically, one output per lane (taps from
// sy4 + gamma * column + delta * row).  A translated prediction of the same block (matrix that
// failed the shear test, chroma smaller than 8x8, the other half of a compound pair) runs the
// per-sample two-pass filter of mc_piece_kernel.  The tool is rare (< 3 % of
// decode time), so the reference samples come straight from global memory with clamped
// coordinates; the filter table is the context's (caller-supplied) 193 x 8 table.
// Masked compound (wedge / difference-weighted) with a warped prediction: the luma rectangles run
// first (chroma_pass = 0) and leave the difference-weighted mask of their samples in the chain's
// second picture, which is idle during reconstruction; the chroma rectangles (chroma_pass = 1)
// average its 2x2 cells.  Wedge masks come from the context's table in both passes.
__global__ void __launch_bounds__(64)
mc_warp_kernel(Batch bt, int chroma_pass)
{
    __shared__ int mid[15 * 8];
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const int index = int(blockIdx.x) + (chroma_pass ? fv.warp_luma_count : 0);
    if (index >= (chroma_pass ? fv.warp_count : fv.warp_luma_count)) return;
    const WarpItem &it = fv.warp_items[index];
    const int pl = it.plane, sx_ = pl ? fv.ssx : 0, sy_ = pl ? fv.ssy : 0, w = it.w, h = it.h;
    const bool compound = it.flags & 1;
    const int nref = compound ? 2 : 1, rnd1 = compound ? 7 : 11;
    const int8_t *__restrict__ wf = fv.warp_filter;
    const int lane = threadIdx.x;
    const bool active = lane < w * h;
    const int r = active ? lane / w : 0, c = active ? lane - r * w : 0;
    int pred[2] = {0, 0};
    for (int k = 0; k < nref; k++) {
        const WarpRef &wr = it.r[k];
        const DevPic &ref = fv.slots[it.ref[k]];
        const int W = ref.pw(pl), H = ref.ph(pl), rs = ref.stride[pl];
        const uint8_t *__restrict__ src = ref.p[pl];
        if (it.flags & (2 << k)) {   // uniform
            const int ix4 = wr.ix4, iy4 = wr.iy4, sx4 = wr.sx4, sy4 = wr.sy4;
            if (k) __syncthreads();   // the first prediction is done with mid[]
            for (int e = lane; e < 15 * 8; e += 64) {
                const int i1 = e / 8 - 7, i2 = e % 8 - 4;
                const int sx = sx4 + wr.alpha * i2 + wr.beta * i1;
                const int8_t *taps = wf + (((sx + 512) >> 10) + 64) * 8;
                const uint8_t *row = src + size_t(clampi(iy4 + i1, 0, H - 1)) * rs;
                int sum = 0;
#pragma unroll
                for (int i3 = 0; i3 < 8; i3++) sum += taps[i3] * row[clampi(ix4 + i2 - 3 + i3, 0, W - 1)];
                mid[e] = (sum + 4) >> 3;
            }
            __syncthreads();
            const int i1 = r - 4, i2 = c - 4;
            const int sy = sy4 + wr.gamma * i2 + wr.delta * i1;
            const int8_t *taps = wf + (((sy + 512) >> 10) + 64) * 8;
            int sum = 0;
#pragma unroll
            for (int i3 = 0; i3 < 8; i3++) sum += taps[i3] * mid[(r + i3) * 8 + c];
            pred[k] = (sum + ((1 << rnd1) >> 1)) >> rnd1;
        } else if (active && ((fv.scaled_mask >> it.ref[k]) & 1u)) {
            // translated from a reference of another size (the other half of the pair warps, or the plane is too small to): positions step from the block's corner
            const int sum = scaled_sample_sum(ref, pl, int(it.x) - it.bx, int(it.y) - it.by, c + it.bx, r + it.by, wr.mv[0], wr.mv[1], it.fset & 15, it.fset >> 4,
                                              fv.ref_scale[it.ref[k]][0], fv.ref_scale[it.ref[k]][1]);
            pred[k] = (sum + ((1 << rnd1) >> 1)) >> rnd1;
        } else if (active) {
            const int px = ((int(it.x) + c) << 4) + (sx_ ? wr.mv[1] : wr.mv[1] * 2), py = ((int(it.y) + r) << 4) + (sy_ ? wr.mv[0] : wr.mv[0] * 2);
            const int ix = px >> 4, iy = py >> 4, fx = px & 15, fy = py & 15;
            const int16_t *fh = kSubpel[it.fset & 15][fx];
            const int16_t *fvt = kSubpel[it.fset >> 4][fy];
            int acc = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int sy = clampi(iy + t - 3, 0, H - 1);
                int sum = 0;
#pragma unroll
                for (int u = 0; u < 8; u++) sum += fh[u] * src[size_t(sy) * rs + clampi(ix + u - 3, 0, W - 1)];
                acc += fvt[t] * ((sum + 4) >> 3);
            }
            pred[k] = (acc + ((1 << rnd1) >> 1)) >> rnd1;
        }
    }
    if (!active) return;
    int val = pred[0];
    if (compound) {
        if (it.comp_type == EA1_COMP_DIST) {
            val = (pred[0] * it.comp_arg + pred[1] * (16 - it.comp_arg) + 128) >> 8;
        } else if (it.comp_type == EA1_COMP_AVG) {
            val = (pred[0] + pred[1] + 16) >> 5;
        } else {
            int m;
            if (it.comp_type == EA1_COMP_WEDGE) {
                const int ms = 1 << it.mask_stride_log2;
                const uint8_t *q = fv.wedge + it.mask_off + ((r << sy_) * ms) + (c << sx_);
                m = !sx_ ? q[0] : (!sy_ ? (q[0] + q[1] + 1) >> 1 : (q[0] + q[1] + q[ms] + q[ms + 1] + 2) >> 2);
            } else {
                uint8_t *scratch = fv.cdef.p[0];   // luma-sized, idle until the CDEF stage
                const int sst = fv.cdef.stride[0];
                if (pl == 0) {
                    const int d = pred[0] - pred[1];
                    const int diff = ((d < 0 ? -d : d) + 8) >> 4;
                    m = clampi(38 + diff / 16, 0, 64);
                    if (it.comp_arg) m = 64 - m;
                    scratch[size_t(it.y + r) * sst + it.x + c] = uint8_t(m);
                } else {
                    const uint8_t *q = scratch + size_t((it.y + r) << sy_) * sst + ((it.x + c) << sx_);
                    m = !sx_ ? q[0] : (!sy_ ? (q[0] + q[1] + 1) >> 1 : (q[0] + q[1] + q[sst] + q[sst + 1] + 2) >> 2);
                }
            }
            val = (m * pred[0] + (64 - m) * pred[1] + 512) >> 10;
        }
    }
    fv.work.p[pl][size_t(it.y + r) * fv.work.stride[pl] + it.x + c] = uint8_t(clampi(val, 0, 255));
}

void launch_inter_pred(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    {
        int max_count = 0;
        uint64_t bytes = 0;
        for (int c = 0; c < s.batch.n; c++) {
            if (!s.stats[c]) continue;
            max_count = std::max(max_count, s.stats[c]->piece_count);
            bytes += 2 * s.stats[c]->piece_px;
        }
        if (max_count) {
            ProfScope ps(prof, "mc_piece_kernel", bytes, st);
            hipLaunchKernelGGL(mc_piece_kernel, dim3(max_count, 1, s.batch.n), dim3(64), 0, st, s.batch);
        }
    }
    const bool layout_420 = s.ssx && s.ssy;
    if (layout_420) {
        launch_mc_class<32, 256, 2, false>(s, st, prof, "mc_tile_kernel<32,256,2,false>");
        launch_mc_class<16, 64, 1, false>(s, st, prof, "mc_tile_kernel<16,64,1,false>");
        launch_mc_class<8, 64, 0, false>(s, st, prof, "mc_tile_kernel<8,64,0,false>");
        launch_mc_class<32, 256, 2, true>(s, st, prof, "mc_tile_kernel<32,256,2,true>");
        launch_mc_class<16, 64, 1, true>(s, st, prof, "mc_tile_kernel<16,64,1,true>");
        launch_mc_class<8, 64, 0, true>(s, st, prof, "mc_tile_kernel<8,64,0,true>");
    }
    // frames that read a reference of another size -- and every frame of a 4:4:4 / 4:2:2 context: the tile kernels above are built around
    // the half-size chroma tile of 4:2:0, this one takes a sample at a time with the layout's own subsampling (with a scale of 1 its
    // arithmetic is that of the tile kernels)
    if (s.any_scaled || !layout_420) {
        int max_count = 0;
        uint64_t bytes = 0;
        for (int c = 0; c < s.batch.n; c++) {
            if (!s.stats[c] || !s.stats[c]->desc.is_inter) continue;
            for (int k = 0; k < 3; k++) {
                max_count = std::max(max_count, std::max(s.stats[c]->mc_count[k], s.stats[c]->mc_ext_count[k]));
                bytes += uint64_t(s.stats[c]->mc_count[k] + s.stats[c]->mc_ext_count[k]) * uint64_t(3 * (64 << (2 * k)));   // a prediction read + written, luma + chroma
            }
        }
        if (max_count) {
            ProfScope ps(prof, "mc_scaled_kernel", bytes, st);
            hipLaunchKernelGGL(mc_scaled_kernel, dim3(max_count, 6, s.batch.n), dim3(256), 0, st, s.batch, layout_420 ? 0 : 1);
        }
    }
    {
        int max_count = 0;
        uint64_t bytes = 0;
        for (int c = 0; c < s.batch.n; c++) {
            if (!s.stats[c]) continue;
            max_count = std::max(max_count, s.stats[c]->warp_count);
            bytes += 2 * s.stats[c]->warp_px;   // reference samples fetched + prediction written (single reference)
        }
        if (max_count) {
            ProfScope ps(prof, "mc_warp_kernel", bytes, st);
            int max_luma = 0, max_chroma = 0;
            for (int c = 0; c < s.batch.n; c++) {
                if (!s.stats[c]) continue;
                max_luma = std::max(max_luma, s.stats[c]->warp_luma_count);
                max_chroma = std::max(max_chroma, s.stats[c]->warp_count - s.stats[c]->warp_luma_count);
            }
            if (max_luma) hipLaunchKernelGGL(mc_warp_kernel, dim3(max_luma, 1, s.batch.n), dim3(64), 0, st, s.batch, 0);
            if (max_chroma) hipLaunchKernelGGL(mc_warp_kernel, dim3(max_chroma, 1, s.batch.n), dim3(64), 0, st, s.batch, 1);
        }
    }
    // OBMC: the strips from above first, then the strips from the left (they overlap in the
    // block's top-left corner and the specification applies them in this order)
    for (int pass = 0; pass < 2; pass++) {
        int max_count = 0;
        uint64_t bytes = 0;
        for (int c = 0; c < s.batch.n; c++) {
            if (!s.stats[c]) continue;
            max_count = std::max(max_count, s.stats[c]->obmc_count[pass]);
            bytes += s.stats[c]->obmc_px[pass] * 3;   // reference read + prediction read-modify-written
        }
        if (!max_count) continue;
        ProfScope ps(prof, "obmc_kernel", bytes, st);
        hipLaunchKernelGGL(obmc_kernel, dim3(max_count, 1, s.batch.n), dim3(256), 0, st, s.batch, pass);
    }
}

}  // namespace ea1
