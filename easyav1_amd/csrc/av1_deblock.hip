// Deblocking filter (AV1 spec 7.14; dav1d src/loopfilter_tmpl.c + lf_apply_tmpl.c + lf_mask.c,
// /root/reference/cmake/dav1d.cmake:1010,:1009,:980).
//
// All vertical edges of a plane are mutually independent (the transform-size rule keeps their
// read/modify footprints disjoint), and so are all horizontal edges; the only ordering is
// "vertical pass before horizontal pass" per plane.  So: one launch per (plane class, pass).
//   vertical edges  : a lane OWNS one transform-grid column group of a row (8 luma / 4 chroma
//                     pixels), evaluates the edge on its left (keeps the q side) and the edge on
//                     its right (keeps the p side) from the ORIGINAL samples and stores its own
//                     8/4 bytes once -> coalesced 8-byte (4-byte) stores, no write sharing.
//                     Because a lane reads its neighbours' samples, this pass is OUT OF PLACE
//                     (work picture -> second picture); an in-place version races at wave
//                     boundaries (caught by the 4K parity test: 4 samples off by one).
//   horizontal edges: a lane owns 4 adjacent columns of one edge; every row access is a 4-byte
//                     load/store, contiguous across the wave.
// Edge classes per 8x8 unit come from the job's edge_info map (built at hand-over like dav1d's
// lf_mask), levels are per frame.  Pure byte traffic: read the plane once and write it once per
// pass; integer VALU only.
#include "av1_dev.h"
#include "ea1_common.h"

namespace ea1 {

struct LfLimits {
    int limit, blimit, thresh;
};

__host__ __device__ static LfLimits make_limits(int lvl, int sharp)
{
    const int shift = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
    int limit = lvl >> shift;
    if (sharp > 0 && limit > 9 - sharp) limit = 9 - sharp;
    if (limit < 1) limit = 1;
    return LfLimits{limit, 2 * (lvl + 2) + limit, lvl >> 4};
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int c8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

// One sample line across an edge.  p[k] = p_k, q[k] = q_k.  NP = 7 for luma, 3 for chroma.
// cls: luma 1 -> filter size 4, 2 -> 8, 3 -> 16; chroma 1 -> 4, 2 -> 6 (AV1 spec 7.14.3).
template <bool CHROMA>
__device__ __forceinline__ void lf_line(int (&p)[CHROMA ? 3 : 7], int (&q)[CHROMA ? 3 : 7], int cls, const LfLimits lim)
{
    const bool wide = CHROMA ? cls == 2 : cls == 3;
    const int p0 = p[0], p1 = p[1], q0 = q[0], q1 = q[1];
    const bool hev = iabs(p1 - p0) > lim.thresh || iabs(q1 - q0) > lim.thresh;
    bool mask = iabs(p1 - p0) <= lim.limit && iabs(q1 - q0) <= lim.limit &&
                iabs(p0 - q0) * 2 + (iabs(p1 - q1) >> 1) <= lim.blimit;
    bool flat = false, flat2 = false;
    if constexpr (CHROMA) {
        if (wide) {
            mask = mask && iabs(p[2] - p1) <= lim.limit && iabs(q[2] - q1) <= lim.limit;
            flat = iabs(p1 - p0) <= 1 && iabs(q1 - q0) <= 1 && iabs(p[2] - p0) <= 1 && iabs(q[2] - q0) <= 1;
        }
    } else if (cls >= 2) {
        mask = mask && iabs(p[2] - p1) <= lim.limit && iabs(q[2] - q1) <= lim.limit &&
               iabs(p[3] - p[2]) <= lim.limit && iabs(q[3] - q[2]) <= lim.limit;
        flat = iabs(p1 - p0) <= 1 && iabs(q1 - q0) <= 1 && iabs(p[2] - p0) <= 1 && iabs(q[2] - q0) <= 1 &&
               iabs(p[3] - p0) <= 1 && iabs(q[3] - q0) <= 1;
        if (wide)
            flat2 = iabs(p[6] - p0) <= 1 && iabs(q[6] - q0) <= 1 && iabs(p[5] - p0) <= 1 && iabs(q[5] - q0) <= 1 &&
                    iabs(p[4] - p0) <= 1 && iabs(q[4] - q0) <= 1;
    }
    if (!mask) return;
    if (!flat) {
        // narrow filter (7.14.6.3)
        const int ps1 = p1 - 128, ps0 = p0 - 128, qs0 = q0 - 128, qs1 = q1 - 128;
        int f = hev ? c8(ps1 - qs1) : 0;
        f = c8(f + 3 * (qs0 - ps0));
        const int f1 = c8(f + 4) >> 3, f2 = c8(f + 3) >> 3;
        q[0] = c8(qs0 - f1) + 128;
        p[0] = c8(ps0 + f2) + 128;
        if (!hev) {
            const int f3 = (f1 + 1) >> 1;
            q[1] = c8(qs1 - f3) + 128;
            p[1] = c8(ps1 + f3) + 128;
        }
        return;
    }
    if constexpr (CHROMA) {
        // 6-tap (flat implies wide here)
        const int p2 = p[2], q2 = q[2];
        p[1] = (p2 * 3 + p1 * 2 + p0 * 2 + q0 + 4) >> 3;
        p[0] = (p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + 4) >> 3;
        q[0] = (p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + 4) >> 3;
        q[1] = (p0 + q0 * 2 + q1 * 2 + q2 * 3 + 4) >> 3;
    } else if (!flat2) {
        const int p2 = p[2], p3 = p[3], q2 = q[2], q3 = q[3];
        p[2] = (3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3;
        p[1] = (2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3;
        p[0] = (p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3;
        q[0] = (p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3;
        q[1] = (p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3;
        q[2] = (p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3;
    } else {
        // 14-tap: out_i = Round2(sum_{j=-6..6} tap(j) * F(clamp(i + j, -7, 6)), 4), tap = 2 for |j| <= 1
        int F[14];  // F[7 + k] = q_k, F[6 - k] = p_k
#pragma unroll
        for (int k = 0; k < 7; k++) {
            F[6 - k] = p[k];
            F[7 + k] = q[k];
        }
        int o[12];
#pragma unroll
        for (int i = -6; i < 6; i++) {
            int t = 0;
#pragma unroll
            for (int j = -6; j <= 6; j++) {
                int pos = i + j;
                pos = pos < -7 ? -7 : (pos > 6 ? 6 : pos);
                t += F[7 + pos] * ((j >= -1 && j <= 1) ? 2 : 1);
            }
            o[i + 6] = (t + 8) >> 4;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            p[k] = o[5 - k];
            q[k] = o[6 + k];
        }
    }
}

__device__ __forceinline__ void unpack8(uint2 w, int (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i] = (w.x >> (8 * i)) & 255;
        v[4 + i] = (w.y >> (8 * i)) & 255;
    }
}
__device__ __forceinline__ uint2 pack8(const int (&v)[8])
{
    uint2 w;
    w.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    w.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    return w;
}

// ---- luma, vertical edges: lane = (8-sample unit column, row) ----
// Every edge is evaluated ONCE, by the lane that owns the unit on its right; the p-side result
// travels to the owner of the left unit with a wave shuffle.  A wave therefore owns 63 units: lane
// 63 is a helper that only evaluates the edge between unit 62 and the next wave's first unit.
// A unit can also hold an edge in its middle (4-sample transforms): filter size 4, both sides
// inside the lane.  Transform alignment keeps the footprints of a unit's left, middle and right
// edges disjoint, so all three are evaluated on the original samples.
__global__ void __launch_bounds__(256)
deblock_luma_v_kernel(Batch bt, int ux_first)
{
    // (round 4) a lane takes the FOUR rows of its unit: edge classes, levels and limits are per 4x4 unit, so they are fetched and derived
    // once for four sample lines instead of once per line -- what made this pass cost twice the horizontal one per sample
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const uint8_t *__restrict__ plane = fv.work.p[0];
    uint8_t *__restrict__ dplane = fv.cdef.p[0];
    const int stride = fv.work.stride[0], w8 = fv.w8, H = fv.work.h;
    const int sharp = fv.desc.lf_sharpness;
    const int lane = threadIdx.x;
    const int ux = ux_first + blockIdx.x * 63 + lane;   // ux_first: first unit of the window (tile-column split), else 0
    const int y4 = blockIdx.y * 4 + threadIdx.y;        // the 4-row unit
    const int y = 4 * y4;
    if (y >= H) return;  // uniform per wave (one wave = one unit row; H is a multiple of 8)
    const bool valid = ux < w8;
    // the frame level of the pass gates it -- unless the levels are per block (delta_lf / segment features may raise a block above a
    // frame level of 0; dav1d's calc_lf_values only zeroes everything when BOTH luma levels are 0)
    const bool on = fv.desc.lf_per_block ? (fv.desc.lf_level[0] | fv.desc.lf_level[1]) != 0 : fv.desc.lf_level[0] != 0;
    // the two 4x4 units of this lane in unit row y4: left-edge classes in bits 0-1 of each byte
    const uint32_t e2 = on && valid ? *reinterpret_cast<const uint16_t *>(fv.luma_edge + size_t(y4) * (2 * w8) + 2 * ux) : 0;
    // level of an edge: this unit's block, or the block across the edge if that is 0 (7.14.4)
    int own_lvl = fv.desc.lf_level[0], lvlL = own_lvl;
    if (fv.desc.lf_per_block) {   // uniform
        // levels of the lane's two 4x4 units; an edge takes the level of the unit on its right, or of
        // the one on its left if that is 0
        const uint32_t l2 = valid ? *reinterpret_cast<const uint32_t *>(fv.lvl_y + size_t(y4) * (2 * w8) + 2 * ux) : 0;
        const int lv0 = l2 & 255, lv1 = (l2 >> 16) & 255;
        int left_lvl = __shfl_up(lv1, 1, 64);
        if (lane == 0 && ux > 0 && valid) left_lvl = fv.lvl_y[size_t(y4) * (2 * w8) + 2 * ux - 1] & 255;
        lvlL = lv0 ? lv0 : left_lvl;
        own_lvl = lv1 ? lv1 : lv0;
    }
    const int eL = lvlL ? e2 & 3 : 0, eM = own_lvl ? (e2 >> 8) & 3 : 0;   // eL is 0 at ux == 0
    const LfLimits lim = make_limits(lvlL, sharp), limM = make_limits(own_lvl, sharp);
    const int eR = __shfl_down(eL, 1, 64);
    const int nmodL = eL == 3 ? 6 : (eL == 2 ? 3 : 2), nmodR = eR == 3 ? 6 : (eR == 2 ? 3 : 2);
    const uint8_t *row0 = plane + size_t(y) * stride + ux * 8;
    uint2 own_w[4];
#pragma unroll
    for (int r = 0; r < 4; r++) own_w[r] = valid ? *reinterpret_cast<const uint2 *>(row0 + size_t(r) * stride) : make_uint2(0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint2 left_w;
        left_w.x = __shfl_up(own_w[r].x, 1, 64);
        left_w.y = __shfl_up(own_w[r].y, 1, 64);
        if (lane == 0 && eL) left_w = *reinterpret_cast<const uint2 *>(row0 + size_t(r) * stride - 8);
        int own[8], res[8], nb[8];
        unpack8(own_w[r], own);
        unpack8(left_w, nb);
#pragma unroll
        for (int i = 0; i < 8; i++) res[i] = own[i];
        if (eL) {
            int p[7], q[7];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                p[k] = nb[7 - k];
                q[k] = own[k];
            }
            lf_line<false>(p, q, eL, lim);
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k < nmodL) {
                    res[k] = q[k];
                    nb[7 - k] = p[k];
                }
        }
        if (eM) {   // edge between samples 3 and 4 of this unit: filter size 4 (modifies 2..5)
            int p[7] = {own[3], own[2], own[1], own[0], 0, 0, 0}, q[7] = {own[4], own[5], own[6], own[7], 0, 0, 0};
            lf_line<false>(p, q, 1, limM);
            res[3] = p[0];
            res[2] = p[1];
            res[4] = q[0];
            res[5] = q[1];
        }
        // hand the filtered p side (the left unit's samples) to the lane that owns that unit
        const uint2 pw = pack8(nb);
        uint2 from_right;
        from_right.x = __shfl_down(pw.x, 1, 64);
        from_right.y = __shfl_down(pw.y, 1, 64);
        if (lane < 63 && eR) {
            int rv[8];
            unpack8(from_right, rv);
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k < nmodR) res[7 - k] = rv[7 - k];
        }
        if (valid && lane < 63) *reinterpret_cast<uint2 *>(dplane + size_t(y + r) * stride + ux * 8) = pack8(res);
    }
}

// ---- luma, horizontal edges: lane = (4 columns, 4-row unit) ----
// in place: the footprints of different horizontal edges are disjoint (transform alignment)
// `step` = 2 when no frame of the batch has a luma transform edge on an odd 4-row unit (no
// 4-sample-high transforms): only the 8-row grid is visited then
__global__ void __launch_bounds__(256)
deblock_luma_h_kernel(Batch bt, int step, int x4_first)
{
    const FrameView *fvp = bt.v[blockIdx.z];
    if (!fvp || !(fvp->desc.lf_per_block ? (fvp->desc.lf_level[0] | fvp->desc.lf_level[1]) : fvp->desc.lf_level[1])) return;
    const FrameView &fv = *fvp;
    uint8_t *__restrict__ plane = fv.cdef.p[0];   // the vertical pass left its output here
    const int stride = fv.work.stride[0], W = fv.work.w, h4 = 2 * fv.h8, w4 = 2 * fv.w8;
    const int x = (x4_first + blockIdx.x * 64 + threadIdx.x) * 4;
    const int uy = (blockIdx.y * 4 + threadIdx.y) * step;
    if (x >= W || uy >= h4 || uy == 0) return;
    const int e = (fv.luma_edge[size_t(uy) * w4 + (x >> 2)] >> 2) & 3;
    if (!e) return;
    int lvl = fv.desc.lf_level[1];
    if (fv.desc.lf_per_block) {   // uniform
        lvl = (fv.lvl_y[size_t(uy) * w4 + (x >> 2)] >> 8) & 255;
        if (!lvl) lvl = (fv.lvl_y[size_t(uy - 1) * w4 + (x >> 2)] >> 8) & 255;
        if (!lvl) return;
    }
    const LfLimits lim = make_limits(lvl, fv.desc.lf_sharpness);
    uint8_t *base = plane + size_t(uy * 4) * stride + x;  // q0 row
    // rows read: filter size 4 -> p1..q1, 8 -> p3..q3, 16 -> p6..q6 (never outside the plane: a
    // class-c edge has transforms of at least that height on both sides)
    const int nrd = e == 3 ? 7 : (e == 2 ? 4 : 2);
    uint32_t rows[14];  // p6..p0, q0..q6
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const int d = k < 7 ? 7 - k : k - 6;   // distance from the edge, 1..7
        rows[k] = d <= nrd ? *reinterpret_cast<const uint32_t *>(base + ptrdiff_t(k - 7) * stride) : 0;
    }
    uint32_t out[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // p5..p0, q0..q5
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int p[7], q[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            p[k] = (rows[6 - k] >> (8 * c)) & 255;
            q[k] = (rows[7 + k] >> (8 * c)) & 255;
        }
        lf_line<false>(p, q, e, lim);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            out[5 - k] |= uint32_t(p[k]) << (8 * c);
            out[6 + k] |= uint32_t(q[k]) << (8 * c);
        }
    }
    const int nmod = e == 3 ? 6 : (e == 2 ? 3 : 2);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < nmod) {
            *reinterpret_cast<uint32_t *>(base + ptrdiff_t(-1 - k) * stride) = out[5 - k];
            *reinterpret_cast<uint32_t *>(base + ptrdiff_t(k) * stride) = out[6 + k];
        }
    }
}

// ---- chroma (both planes via blockIdx.z), vertical edges: lane = (PAIR of 4-sample units, row) ----
// same scheme as luma: a lane owns 8 samples, i.e. two chroma units; the edge on the pair's left is
// evaluated by this lane and its p side handed to the left owner by shuffle, the edge between the
// two units lies inside the lane (4- or 6-tap).  63 owned pairs per wave + 1 helper lane.  The
// footprints of neighbouring edges are disjoint (a 6-tap edge has 8-sample transforms on both
// sides), so every edge is evaluated on the original samples.
__global__ void __launch_bounds__(256)
deblock_chroma_v_kernel(Batch bt, int up_first)
{
    // (round 4: four rows per lane, like the luma pass)
    const FrameView *fvp = bt.v[blockIdx.z >> 1];
    const int pl = blockIdx.z & 1;
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const bool on = (fv.desc.lf_level[0] || fv.desc.lf_level[1]) && fv.desc.lf_level[2 + pl];
    const uint8_t *plane = fv.work.p[1 + pl];
    uint8_t *dplane = fv.cdef.p[1 + pl];
    const int stride = fv.work.stride[1 + pl], w8 = fv.cw4, Hc = fv.work.h >> fv.ssy;   // (w8: 4x4 units per chroma row -- W / 8 in 4:2:0)
    const int lane = threadIdx.x;
    const int up = up_first + blockIdx.x * 63 + lane;   // pair index; units 2 * up and 2 * up + 1
    const int y4 = blockIdx.y * 4 + threadIdx.y;        // the 4-row unit
    const int y = 4 * y4;
    if (y >= Hc) return;  // uniform per wave (Hc is a multiple of 4)
    const int ux = 2 * up;
    const bool valid = ux < w8, valid2 = ux + 1 < w8;
    int lvlL = fv.desc.lf_level[2 + pl], lvlM = lvlL;
    if (fv.desc.lf_per_block) {   // uniform
        const uint16_t *lrow = fv.lvl_c + size_t(y4) * w8;
        const int l0 = valid ? (lrow[ux] >> (8 * pl)) & 255 : 0, l1 = valid2 ? (lrow[ux + 1] >> (8 * pl)) & 255 : 0;
        int left_lvl = __shfl_up(l1, 1, 64);
        if (lane == 0 && ux > 0 && valid) left_lvl = (lrow[ux - 1] >> (8 * pl)) & 255;
        lvlL = l0 ? l0 : left_lvl;
        lvlM = l1 ? l1 : l0;
    }
    const LfLimits lim = make_limits(lvlL, fv.desc.lf_sharpness), limM = make_limits(lvlM, fv.desc.lf_sharpness);
    const uint8_t *erow = fv.chroma_edge + size_t(y4) * w8;
    const int eL = on && valid && lvlL ? erow[ux] & 3 : 0;   // 0 at ux == 0
    const int eM = on && valid2 && lvlM ? erow[ux + 1] & 3 : 0;
    const int eR = __shfl_down(eL, 1, 64);
    const uint8_t *row0 = plane + size_t(y) * stride + ux * 4;
    uint2 own_w[4];
#pragma unroll
    for (int r = 0; r < 4; r++) own_w[r] = valid ? *reinterpret_cast<const uint2 *>(row0 + size_t(r) * stride) : make_uint2(0, 0);   // the second half may be row padding: unused then
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t left_hi = __shfl_up(own_w[r].y, 1, 64);                // the 4 samples left of this pair
        if (lane == 0 && eL) left_hi = *reinterpret_cast<const uint32_t *>(row0 + size_t(r) * stride - 4);
        int own[8], res[8], nb[4];
        unpack8(own_w[r], own);
#pragma unroll
        for (int i = 0; i < 8; i++) res[i] = own[i];
#pragma unroll
        for (int i = 0; i < 4; i++) nb[i] = (left_hi >> (8 * i)) & 255;
        if (eL) {
            int p[3], q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                p[k] = nb[3 - k];
                q[k] = own[k];
            }
            lf_line<true>(p, q, eL, lim);
            res[0] = q[0];
            res[1] = q[1];
            nb[3] = p[0];
            nb[2] = p[1];
        }
        if (eM) {   // the edge between the pair's two units
            int p[3] = {own[3], own[2], own[1]}, q[3] = {own[4], own[5], own[6]};
            lf_line<true>(p, q, eM, limM);
            res[3] = p[0];
            res[2] = p[1];
            res[4] = q[0];
            res[5] = q[1];
        }
        const uint32_t pw = nb[0] | (nb[1] << 8) | (nb[2] << 16) | (nb[3] << 24);
        const uint32_t from_right = __shfl_down(pw, 1, 64);
        if (lane < 63 && eR) {
            res[6] = (from_right >> 16) & 255;
            res[7] = (from_right >> 24) & 255;
        }
        if (valid && lane < 63) {
            const uint2 o = pack8(res);
            uint8_t *d = dplane + size_t(y + r) * stride + ux * 4;
            if (valid2) *reinterpret_cast<uint2 *>(d) = o;
            else *reinterpret_cast<uint32_t *>(d) = o.x;
        }
    }
}

// ---- chroma, horizontal edges: lane = (4 columns = one unit, unit row) ----
__global__ void __launch_bounds__(256)
deblock_chroma_h_kernel(Batch bt, int ux_first)
{
    const FrameView *fvp = bt.v[blockIdx.z >> 1];
    const int pl = blockIdx.z & 1;
    if (!fvp || (!fvp->desc.lf_level[0] && !fvp->desc.lf_level[1]) || !fvp->desc.lf_level[2 + pl]) return;
    const FrameView &fv = *fvp;
    uint8_t *plane = fv.cdef.p[1 + pl];   // the vertical pass left its output here
    const int stride = fv.work.stride[1 + pl], w8 = fv.cw4, h8 = fv.ch4;   // (4x4 units of the chroma plane -- W / 8 x H / 8 in 4:2:0)
    const int ux = ux_first + blockIdx.x * 64 + threadIdx.x;
    const int uy = blockIdx.y * 4 + threadIdx.y;
    if (ux >= w8 || uy >= h8 || uy == 0) return;
    const int e = (fv.chroma_edge[size_t(uy) * w8 + ux] >> 2) & 3;
    if (!e) return;
    int lvl = fv.desc.lf_level[2 + pl];
    if (fv.desc.lf_per_block) {   // uniform
        lvl = (fv.lvl_c[size_t(uy) * w8 + ux] >> (8 * pl)) & 255;
        if (!lvl) lvl = (fv.lvl_c[size_t(uy - 1) * w8 + ux] >> (8 * pl)) & 255;
        if (!lvl) return;
    }
    const LfLimits lim = make_limits(lvl, fv.desc.lf_sharpness);
    uint8_t *base = plane + size_t(uy * 4) * stride + ux * 4;
    uint32_t rows[6];  // p2, p1, p0, q0, q1, q2
#pragma unroll
    for (int k = 0; k < 6; k++) rows[k] = *reinterpret_cast<const uint32_t *>(base + ptrdiff_t(k - 3) * stride);
    uint32_t out[4] = {0, 0, 0, 0};  // p1, p0, q0, q1
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int p[3], q[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p[k] = (rows[2 - k] >> (8 * c)) & 255;
            q[k] = (rows[3 + k] >> (8 * c)) & 255;
        }
        lf_line<true>(p, q, e, lim);
        out[0] |= uint32_t(p[1]) << (8 * c);
        out[1] |= uint32_t(p[0]) << (8 * c);
        out[2] |= uint32_t(q[0]) << (8 * c);
        out[3] |= uint32_t(q[1]) << (8 * c);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) *reinterpret_cast<uint32_t *>(base + ptrdiff_t(k - 2) * stride) = out[k];
}

void launch_deblock(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    // every pass reads and rewrites its plane once.  The vertical passes always run: they move the
    // picture from the work buffer to the second buffer (copy-through where deblocking is off).
    uint64_t yh = 0, cbh = 0, frames = 0;
    bool odd_rows = false;
    const uint64_t ybytes = uint64_t(s.width) * s.height;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        frames++;
        odd_rows |= s.stats[c]->luma_odd_row_edges;
        const ea1_frame_desc &d = s.stats[c]->desc;
        if (!d.lf_level[0] && !d.lf_level[1]) continue;  // spec 7.14.1: deblocking is off for the frame
        if (d.lf_level[1] || d.lf_per_block) yh += 2 * ybytes;
        cbh += ((2 * ybytes) >> (s.ssx + s.ssy)) * ((d.lf_level[2] ? 1 : 0) + (d.lf_level[3] ? 1 : 0));
    }
    if (!frames) return;
    const int h8 = s.height / 8, n = s.batch.n;
    // the window in 8-sample luma units (whole frame: 0 .. w8); the kernels bound their reads by the picture, the grid
    // only decides which columns are visited
    const int u0 = s.win_x0 / 8, u1 = (s.win_x1 + 7) / 8, nu = u1 - u0;
    const uint64_t share_num = uint64_t(s.win_x1 - s.win_x0), share_den = uint64_t(s.width);   // algorithmic bytes of the window
    auto part = [&](uint64_t bytes) { return bytes * share_num / share_den; };
    const dim3 blk(64, 4);
    {
        ProfScope ps(prof, "deblock_luma_v_kernel", part(frames * 2 * ybytes), st);
        hipLaunchKernelGGL(deblock_luma_v_kernel, dim3((nu + 62) / 63, (s.height / 4 + 3) / 4, n), blk, 0, st, s.batch, u0);   // a wave per row of 4x4 units
    }
    if (yh) {
        ProfScope ps(prof, "deblock_luma_h_kernel", part(yh), st);
        const int step = odd_rows ? 1 : 2;
        hipLaunchKernelGGL(deblock_luma_h_kernel, dim3((2 * nu + 63) / 64, (2 * h8 / step + 3) / 4, n), blk, 0, st, s.batch, step, 2 * u0);
    }
    // chroma: the window in 4x4 units of the chroma plane (4:2:0: the same numbers as the 8-sample luma units above)
    const int cu0 = (s.win_x0 >> s.ssx) / 4, cu1 = ((s.win_x1 >> s.ssx) + 3) / 4, ch4 = (s.height >> s.ssy) / 4;
    {
        ProfScope ps(prof, "deblock_chroma_v_kernel", part((frames * 4 * ybytes) >> (s.ssx + s.ssy)), st);
        const int p0 = cu0 / 2, p1 = (cu1 + 1) / 2;
        hipLaunchKernelGGL(deblock_chroma_v_kernel, dim3((p1 - p0 + 62) / 63, (ch4 + 3) / 4, 2 * n), blk, 0, st, s.batch, p0);
    }
    if (cbh) {
        ProfScope ps(prof, "deblock_chroma_h_kernel", part(cbh), st);
        hipLaunchKernelGGL(deblock_chroma_h_kernel, dim3((cu1 - cu0 + 63) / 64, (ch4 + 3) / 4, 2 * n), blk, 0, st, s.batch, cu0);
    }
}

}  // namespace ea1
