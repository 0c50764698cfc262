// Loop restoration: Wiener and self-guided filters (AV1 spec 7.17; dav1d
// src/looprestoration_tmpl.c + lr_apply_tmpl.c, /root/reference/cmake/dav1d.cmake:1011,:1012).
//
// Input is the CDEF output, except that rows outside the current 64-luma-row stripe (stripes are
// offset by 8 rows) come from the DEBLOCKED picture, at most 2 rows deep (spec 7.17.6
// "get_source_sample").  One workgroup restores one (stripe x TW columns) tile: restoration-unit
// rows only change at stripe boundaries and unit columns at multiples of TW, so stripe and unit
// parameters are uniform per tile and the vertical halo is paid once per 64 (32) rows.
//
// The tile plus a 3-sample halo is staged in LDS through the source-selection rule with aligned
// 4-byte loads; both filters then run out of LDS with 4 samples per lane and aligned 4/8/16-byte
// LDS accesses:
//   Wiener: 7-tap horizontal (3 dword reads -> 4 results) -> int16 tile -> 7-tap vertical on a
//           4x4 patch per lane (10 8-byte reads -> 16 results).  Exact integer FIR.
//   SGR   : separable box sums (horizontal 3/5-wide sums of c and c^2, then vertical) -> a/b
//           surfaces (uint16) -> weighted 3x3 per 4 samples -> projection.  The two passes
//           (r = 2, 1) reuse the same LDS; the pass-0 contribution stays in registers.
// Traffic: read plane once (+ 6/64 halo rows), write plane once.  Integer VALU, HBM/L2 bound.
#include "av1_dev.h"
#include "ea1_av1_tables.inc"
#include "ea1_common.h"

namespace ea1 {

__device__ __constant__ static const int16_t kSgrParams[16][4] = {EA1_SGR_PARAMS_VALUES};

// ROCm 7.2 / gfx950 code generation hazard: "clamp((x >> n), 0, 255)" of two values that are then
// packed into bytes is selected as v_ashr_pk_u8_i32, whose upper 16 result bits are NOT zero as the
// compiler assumes (they keep stale register contents); OR-ing the third byte on top corrupts it
// (seen as +1..+4 errors in every 4th column).  Hiding the shifted value from the pattern matcher
// keeps the plain shift + v_med3 + pack sequence.  (found by the bit-exact parity tests)
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// a2 = x_by_xplus1[min(z, 255)]: [derived] round(256 * z / (z + 1)) with the spec's end values
// ((z << 8) + z / 2) / (z + 1), 1 for z == 0 and 256 for z >= 255 -- a table instead of an integer
// division per sample (a variable 32-bit divide is ~40 VALU instructions)
struct XbyX1 {
    uint16_t v[256];
    constexpr XbyX1() : v{}
    {
        v[0] = 1;
        for (int z = 1; z < 255; z++) v[z] = uint16_t(((z << 8) + z / 2) / (z + 1));
        v[255] = 256;
    }
};
__device__ __constant__ static const XbyX1 kXbyX1{};

__device__ __forceinline__ void bytes12(uint32_t a, uint32_t b, uint32_t c, int (&v)[12])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i] = (a >> (8 * i)) & 255;
        v[4 + i] = (b >> (8 * i)) & 255;
        v[8 + i] = (c >> (8 * i)) & 255;
    }
}

// TW x TH: tile (TH = stripe height in this plane), NT threads, PLANES = 1 luma / 2 chroma.
// SGR = false builds a Wiener-only variant (no self-guided scratch in LDS: 14 KB instead of 52 KB
// for the 64x64 tile, i.e. 3x the resident workgroups); the launcher uses it when no restoration
// unit of the submit is self-guided in these planes.
// store 4 restored samples at (x, y); with CROP the bytes at or beyond the visible width W keep the CDEF output
template <bool CROP>
__device__ __forceinline__ void lr_store4(uint8_t *out, const uint8_t *cdef, int stride, int W, int x, int y, uint32_t pk)
{
    if (CROP && x + 3 >= W) {
        const uint32_t keep = 0xFFFFFFFFu >> (8 * (x + 4 - W));   // bytes with column < W (1..3 of them)
        const uint32_t cwd = *reinterpret_cast<const uint32_t *>(cdef + size_t(y) * stride + x);
        pk = (pk & keep) | (cwd & ~keep);
    }
    *reinterpret_cast<uint32_t *>(out + size_t(y) * stride + x) = pk;
}

// CROP: a frame of the batch has a visible size below its coded size (ea1_frame_desc.crop); the plain variant
// carries none of that handling
template <int TW, int TH, int NT, int PLANES, bool SGR, bool CROP>
__global__ void __launch_bounds__(NT)
lr_stripe_kernel(Batch bt, int deb_second, int cdef_second, TileGrid grid)
{
    constexpr int SWB = TW + 12;             // source tile row in bytes: picture x0-4 .. x0+TW+7
    constexpr int SWD = SWB / 4;             // dwords per source row
    constexpr int SH = TH + 6;               // source rows: picture y0-3 .. y0+TH+2
    constexpr int AW = TW + 4;               // a/b and box-sum row stride (elements), col jj = j + 1
    constexpr int GPT = (TW * TH / 4 + NT - 1) / NT;  // 4-sample groups owned by a lane
    __shared__ uint32_t src[SH * SWD];
    // Wiener: mid int16 [SH][TW].  SGR: uint2 [SH][AW] horizontal box sums of the radius in flight; a | b << 16 uint32 [TH+2][AW]
    constexpr int kWienerBytes = SH * TW * 2;
    constexpr int kSgrBytes = SH * AW * 8 + (TH + 2) * AW * 4;   // one radius at a time: {sum of squares, sum} rows + the a | b surface
    __shared__ __attribute__((aligned(16))) uint8_t scratch[(SGR && kSgrBytes > kWienerBytes) ? kSgrBytes : kWienerBytes];

    const TilePos tp = xcd_tile(grid);   // neighbouring tiles (shared halos) behind one L2
    if (!tp.valid) return;
    const FrameView *fvp = bt.v[tp.z / PLANES];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const int p = PLANES == 1 ? 0 : 1 + tp.z % PLANES;
    // PLANES = 2: the tile says the layout -- 32 x 32 is 4:2:0 (stripes of 64 >> 1 rows), 32 x 64 is 4:2:2, 64 x 64 is 4:4:4
    constexpr int sx_ = PLANES == 2 && TW == 32 ? 1 : 0, sy_ = PLANES == 2 && TH == 32 ? 1 : 0;
    // W x H: the VISIBLE plane (restoration clamps its reads there, AV1 spec 7.17 PlaneEndX / unitRows);
    // CW x CH: the coded plane (whole 8x8 luma units), which keeps the CDEF output beyond the visible part
    // (super-resolution, fv.up_vis_w != 0: the pictures restoration reads are the upscaled pair, upscaled_width visible samples in up_cw
    // stored ones; such submits always run the CROP variant)
    const bool sr = CROP && fv.up_vis_w != 0;
    const int CW = sr ? fv.up_cw >> sx_ : fv.work.pw(p), CH = fv.work.ph(p), stride = fv.work.stride[p];
    const int W = CROP ? ((sr ? fv.up_vis_w : fv.vis_w) + sx_) >> sx_ : CW, H = CROP ? (fv.vis_h + sy_) >> sy_ : CH;
    const int off = 8 >> sy_;
    const int x0 = grid.x_origin + tp.x * TW;   // x_origin: multiple of TW (a rank's strip starts on a superblock column)
    const int stripe = tp.y;
    const int s_start = stripe * TH - off, s_end = s_start + TH - 1;   // stripe rows in this plane
    const int y0 = s_start < 0 ? 0 : s_start;
    const int y1 = s_end + 1 > H ? H : s_end + 1;
    const int cy1 = s_end + 1 > CH ? CH : s_end + 1;
    if (x0 >= CW || y0 >= cy1) return;
    // which of the chain's two pictures holds the deblocked samples / the CDEF output
    const uint8_t *deblocked = sr ? fv.up_deb.p[p] : (deb_second ? fv.cdef.p[p] : fv.work.p[p]);
    const uint8_t *cdef = sr ? fv.up_cdef.p[p] : (cdef_second ? fv.cdef.p[p] : fv.work.p[p]);
    uint8_t *out = fv.dst.p[p];
    const int tid = threadIdx.x;
    if (CROP && (W < CW || H < CH)) {
        // rows and columns of the coded plane beyond the visible picture: CDEF output unchanged
        for (int i = tid; i < (cy1 - y0) * (TW / 4); i += NT) {
            const int r = i / (TW / 4), g = i - r * (TW / 4);
            const int x = x0 + 4 * g, y = y0 + r;
            // (a dword that straddles the visible width inside a visible row is written once, merged, by lr_store4 below)
            if (x < CW && (y >= H || x >= W)) *reinterpret_cast<uint32_t *>(out + size_t(y) * stride + x) = *reinterpret_cast<const uint32_t *>(cdef + size_t(y) * stride + x);
        }
        if (x0 >= W || y0 >= y1) return;   // uniform
    }
    const int th = y1 - y0;

    ea1_lr_unit unit;
    unit.type = EA1_LR_NONE;
    if (fv.desc.lr_type[p] && fv.lr[p]) {
        const int ul = fv.desc.lr_unit_log2[p];
        int urow = (y0 + off) >> ul, ucol = x0 >> ul;
        urow = urow > fv.lr_rows[p] - 1 ? fv.lr_rows[p] - 1 : urow;
        ucol = ucol > fv.lr_cols[p] - 1 ? fv.lr_cols[p] - 1 : ucol;
        unit = fv.lr[p][urow * fv.lr_cols[p] + ucol];
    }

    // ---- stage the source tile: row r <-> picture row y0 - 3 + r (through the stripe rule),
    //      dword k <-> picture columns x0 - 4 + 4k .. +3 (replicated outside the picture) ----
    for (int i = tid; i < (th + 6) * SWD; i += NT) {
        const int r = i / SWD, k = i - r * SWD;
        int y = y0 - 3 + r;
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        const uint8_t *from = cdef;
        if (y < s_start) {
            y = y > s_start - 2 ? y : s_start - 2;
            from = deblocked;
        } else if (y > s_end) {
            y = y < s_end + 2 ? y : s_end + 2;
            from = deblocked;
        }
        const uint8_t *row = from + size_t(y) * stride;
        const int x = x0 - 4 + 4 * k;
        uint32_t v;
        if (x >= 0 && x + 3 < W) {
            v = *reinterpret_cast<const uint32_t *>(row + x);
        } else if (!CROP || x < 0 || x >= W) {   // without crop W % 4 == 0: a dword is wholly in or out
            v = 0x01010101u * uint32_t(row[x < 0 ? 0 : W - 1]);
        } else {   // the dword straddles the right edge of the visible plane (width not a multiple of 4)
            v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v |= uint32_t(row[x + j < W ? x + j : W - 1]) << (8 * j);
        }
        src[i] = v;
    }
    __syncthreads();

    if (unit.type == EA1_LR_WIENER) {
        int16_t *mid = reinterpret_cast<int16_t *>(scratch);  // [SH][TW]
        int hf[7], vf[7];
        hf[3] = vf[3] = 128;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            vf[i] = vf[6 - i] = unit.coef[0][i];
            vf[3] -= 2 * unit.coef[0][i];
            hf[i] = hf[6 - i] = unit.coef[1][i];
            hf[3] -= 2 * unit.coef[1][i];
        }
        // horizontal: output col c of row r reads source bytes c+1 .. c+7
        for (int i = tid; i < (th + 6) * (TW / 4); i += NT) {
            const int r = i / (TW / 4), g = i - r * (TW / 4);
            int b[12];
            bytes12(src[r * SWD + g], src[r * SWD + g + 1], src[r * SWD + g + 2], b);
            int m[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int s = 0;
#pragma unroll
                for (int t = 0; t < 7; t++) s += hf[t] * b[j + 1 + t];
                s = (s + 4) >> 3;
                m[j] = s < -2048 ? -2048 : (s > 6143 ? 6143 : s);   // spec 7.17.4: Clip3(-offset, limit - offset, v), offset 2048, limit 8191
            }
            uint2 w;
            w.x = (uint32_t(m[0]) & 0xFFFF) | (uint32_t(m[1]) << 16);
            w.y = (uint32_t(m[2]) & 0xFFFF) | (uint32_t(m[3]) << 16);
            *reinterpret_cast<uint2 *>(&mid[r * TW + 4 * g]) = w;
        }
        __syncthreads();
        // vertical: a lane produces a 4x4 patch from 10 intermediate rows
        for (int i = tid; i < ((th + 3) / 4) * (TW / 4); i += NT) {   // th is a multiple of 4 unless the visible height is not
            const int seg = i / (TW / 4), g = i - seg * (TW / 4);
            if (x0 + 4 * g >= W) continue;
            int col[4][10];
#pragma unroll
            for (int rr = 0; rr < 10; rr++) {
                const int mr = !CROP || 4 * seg + rr < th + 5 ? 4 * seg + rr : th + 5;   // rows of a partial last segment: clamped, unused
                const uint2 w = *reinterpret_cast<const uint2 *>(&mid[mr * TW + 4 * g]);
                col[0][rr] = int16_t(w.x & 0xFFFF);
                col[1][rr] = int16_t(w.x >> 16);
                col[2][rr] = int16_t(w.y & 0xFFFF);
                col[3][rr] = int16_t(w.y >> 16);
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                uint32_t pk = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 7; t++) s += vf[t] * col[c][rr + t];
                    s = opaque((s + 1024) >> 11);
                    s = s < 0 ? 0 : (s > 255 ? 255 : s);
                    pk |= uint32_t(s) << (8 * c);
                }
                if (!CROP || 4 * seg + rr < th) lr_store4<CROP>(out, cdef, stride, W, x0 + 4 * g, y0 + 4 * seg + rr, pk);
            }
        }
    } else if (SGR && unit.type == EA1_LR_SGRPROJ) {
        // Self-guided filter, one radius at a time (r = 2 then r = 1; what pass 0 contributes stays in registers):
        //   A. horizontal box sums of c and c^2 with the byte dot product: a lane makes 4 columns of one row from
        //      three source dwords -- window = v_alignbyte, sums = v_dot4_u32_u8 against 0x01..01 and against itself
        //   B. vertical sums + the a / b surfaces: a lane walks down ONE column over a run of rows, box rows kept in
        //      registers (1.5 - 3 LDS reads per element instead of 2 (2r + 1)), a and b stored as one dword
        //   C. weighted 3x3 of the surfaces for 4 samples per lane, rows read as 16 + 8 byte accesses
        __shared__ uint16_t sgr_lut[256];
        sgr_lut[tid & 255] = kXbyX1.v[tid & 255];   // NT == 256; visible after the barriers below
        uint2 *HS = reinterpret_cast<uint2 *>(scratch);                        // [SH][AW] {sum of squares, sum} of the row's box
        uint32_t *AB = reinterpret_cast<uint32_t *>(scratch + SH * AW * 8);    // [TH+2][AW] a | b << 16
        const int16_t *prm = kSgrParams[unit.sgr_set & 15];
        const int w0 = unit.coef[0][0], w1 = unit.coef[0][1], w2 = 128 - w0 - w1;
        int acc[GPT][4];  // w1*u + w0*(flt0 or u) [+ w2*(flt1 or u)], per owned group
#pragma unroll
        for (int k = 0; k < GPT; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[k][j] = 0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int r = prm[2 * pass], eps = prm[2 * pass + 1];   // uniform for the workgroup
            const int wgt = pass == 0 ? w0 : w2;
            if (r) {
                const int n = (2 * r + 1) * (2 * r + 1);
                const unsigned sc = unsigned(eps);   // Sgr_Params holds the scale s of z = Round2(p * s, 20) itself (7.17.3); p * s + 2^19 < 2^32 for 8-bit samples
                const unsigned one_by_n = ((1u << 12) + n / 2) / n;
                __syncthreads();   // the source tile is staged / the previous pass has finished with HS and AB
                // ---- A: jj = 4q + j <-> column j - 1 of the tile <-> source byte jj + 3; bytes of the three dwords: 4q .. 4q + 11
                for (int i = tid; i < (th + 6) * (AW / 4); i += NT) {
                    const int rs = i / (AW / 4), q = i - rs * (AW / 4);
                    const uint32_t d0 = src[rs * SWD + q], d1 = src[rs * SWD + q + 1], d2 = src[rs * SWD + q + 2];
                    // window of output j = the four bytes from byte offset j + 2 (left neighbour, centre, right neighbour, second right)
                    const uint32_t wn[4] = {__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d1, d0, 3), d1, __builtin_amdgcn_alignbyte(d2, d1, 1)};
                    uint32_t sq[4], sm[4];
                    if (pass == 0) {   // 5 wide: the window plus the second left neighbour (byte offset j + 1)
                        const uint32_t l2[4] = {(d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24, d1 & 255u};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            sm[j] = __builtin_amdgcn_udot4(wn[j], 0x01010101u, l2[j], false);
                            sq[j] = __builtin_amdgcn_udot4(wn[j], wn[j], l2[j] * l2[j], false);
                        }
                    } else {           // 3 wide: the low three bytes of the window
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t w3 = wn[j] & 0x00FFFFFFu;
                            sm[j] = __builtin_amdgcn_udot4(w3, 0x00010101u, 0u, false);
                            sq[j] = __builtin_amdgcn_udot4(w3, w3, 0u, false);
                        }
                    }
                    uint4 *dst4 = reinterpret_cast<uint4 *>(&HS[rs * AW + 4 * q]);
                    dst4[0] = make_uint4(sq[0], sm[0], sq[1], sm[1]);
                    dst4[1] = make_uint4(sq[2], sm[2], sq[3], sm[3]);
                }
                __syncthreads();
                // ---- B: surface row ii = i + 1 for i in [-1, th], box centre = source row ii + 2.  Pass 0 only ever reads rows
                //      with even ii: only those are made.  A lane takes column jj and a run of RUN surface rows.
                auto ab_of = [&](unsigned a, unsigned b) {
                    const int pv = int(a * n) - int(b * b);
                    const unsigned pp = pv < 0 ? 0u : unsigned(pv);
                    const unsigned z = (pp * sc + (1u << 19)) >> 20;  // pp*sc < 2^32 for 8-bit samples
                    const unsigned a2 = sgr_lut[z > 255 ? 255 : z];
                    const unsigned b2 = (256 - a2) * b * one_by_n;
                    return a2 | (((b2 + (1u << 11)) >> 12) << 16);
                };
                if (pass == 0) {
                    constexpr int RUN = 3;                      // surface rows 2 ri of a run: 3 boxes of 5 rows share 9 source rows
                    const int nrows = (th + 3) / 2;             // ri = 0 .. nrows - 1 <-> ii = 2 ri
                    const int nruns = (nrows + RUN - 1) / RUN;
                    for (int item = tid; item < nruns * (TW + 2); item += NT) {
                        const int run = item / (TW + 2), jj = item - run * (TW + 2);
                        const int ri0 = run * RUN;
                        const uint2 *col = &HS[(2 * ri0) * AW + jj];   // source row of the first box's top row = ii + 2 - 2 = 2 ri0
                        uint2 v[2 * RUN + 3];
#pragma unroll
                        for (int t = 0; t < 2 * RUN + 3; t++) {
                            const int row = 2 * ri0 + t;
                            v[t] = row < th + 6 ? col[t * AW] : make_uint2(0u, 0u);
                        }
#pragma unroll
                        for (int o = 0; o < RUN; o++) {
                            if (ri0 + o >= nrows) break;
                            const unsigned a = v[2 * o].x + v[2 * o + 1].x + v[2 * o + 2].x + v[2 * o + 3].x + v[2 * o + 4].x;
                            const unsigned b = v[2 * o].y + v[2 * o + 1].y + v[2 * o + 2].y + v[2 * o + 3].y + v[2 * o + 4].y;
                            AB[(2 * (ri0 + o)) * AW + jj] = ab_of(a, b);
                        }
                    }
                } else {
                    constexpr int RUN = 6;                      // 6 boxes of 3 rows share 8 source rows
                    const int nrows = th + 2;
                    const int nruns = (nrows + RUN - 1) / RUN;
                    for (int item = tid; item < nruns * (TW + 2); item += NT) {
                        const int run = item / (TW + 2), jj = item - run * (TW + 2);
                        const int ii0 = run * RUN;
                        const uint2 *col = &HS[(ii0 + 1) * AW + jj];   // source row ii + 2 - 1
                        uint2 v[RUN + 2];
#pragma unroll
                        for (int t = 0; t < RUN + 2; t++) {
                            const int row = ii0 + 1 + t;
                            v[t] = row < th + 6 ? col[t * AW] : make_uint2(0u, 0u);
                        }
#pragma unroll
                        for (int o = 0; o < RUN; o++) {
                            if (ii0 + o >= nrows) break;
                            AB[(ii0 + o) * AW + jj] = ab_of(v[o].x + v[o + 1].x + v[o + 2].x, v[o].y + v[o + 1].y + v[o + 2].y);
                        }
                    }
                }
                __syncthreads();
            }
            // ---- C: weighted 3x3 + accumulate (also the pass-through terms when r == 0)
#pragma unroll
            for (int k = 0; k < GPT; k++) {
                const int gi = tid + k * NT;
                if (gi >= (th * TW) / 4) continue;
                const int i = gi / (TW / 4), g = gi - i * (TW / 4);
                const uint32_t cw = src[(i + 3) * SWD + g + 1];   // the 4 centre samples
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int px = (cw >> (8 * j)) & 255;
                    if (pass == 0) acc[k][j] = w1 * (px << 4);
                    if (!r) acc[k][j] += wgt * (px << 4);
                }
                if (!r) continue;
                // surface columns jj = 4g .. 4g + 5 of the rows around i (surface row ii = i + 1)
                int sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
#pragma unroll
                for (int dy = -1; dy <= 1; dy++) {
                    // pass 0 uses rows with odd (i + dy) only: i odd -> dy = 0, i even -> dy = +-1
                    if (pass == 0 && ((i + dy) & 1) == 0) continue;
                    const uint32_t *row = &AB[(i + 1 + dy) * AW + 4 * g];
                    const uint4 lo = *reinterpret_cast<const uint4 *>(row);
                    const uint2 hi = *reinterpret_cast<const uint2 *>(row + 4);
                    const uint32_t e[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
                    int av[6], bv[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) {
                        av[t] = int(e[t] & 0xFFFFu);
                        bv[t] = int(e[t] >> 16);
                    }
                    // pass 0: 5 6 5; pass 1: 4 4 4 on the centre row, 3 4 3 above and below
                    const int wc = pass == 0 ? 6 : 4;
                    const int ws = pass == 0 ? 5 : (dy == 0 ? 4 : 3);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        sa[j] += ws * (av[j] + av[j + 2]) + wc * av[j + 1];
                        sb[j] += ws * (bv[j] + bv[j + 2]) + wc * bv[j + 1];
                    }
                }
                const int sh = (pass == 0 && (i & 1)) ? 8 : 9;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int px = (cw >> (8 * j)) & 255;
                    const int flt = (sa[j] * px + sb[j] + (1 << (sh - 1))) >> sh;
                    acc[k][j] += wgt * flt;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < GPT; k++) {
            const int gi = tid + k * NT;
            if (gi >= (th * TW) / 4) continue;
            const int i = gi / (TW / 4), g = gi - i * (TW / 4);
            if (x0 + 4 * g >= W) continue;
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int s = opaque((acc[k][j] + (1 << 10)) >> 11);
                s = s < 0 ? 0 : (s > 255 ? 255 : s);
                pk |= uint32_t(s) << (8 * j);
            }
            lr_store4<CROP>(out, cdef, stride, W, x0 + 4 * g, y0 + i, pk);
        }
    } else {
        // unit (or plane) not restored: copy the CDEF output through
        for (int gi = tid; gi < (th * TW) / 4; gi += NT) {
            const int i = gi / (TW / 4), g = gi - i * (TW / 4);
            if (x0 + 4 * g >= W) continue;
            lr_store4<CROP>(out, cdef, stride, W, x0 + 4 * g, y0 + i, src[(i + 3) * SWD + g + 1]);
        }
    }
}

// Super-resolution (AV1 spec 7.16): the deblocked and the CDEF-filtered picture of a frame upscaled horizontally, 8 taps of the
// 64-phase Upscale_Filter around floor(position) with the source clamped to the coded plane.  A lane makes 4 adjacent output samples of
// one row of one picture; blockIdx.z = chain * 6 + picture * 3 + plane.  Positions in 1 / 16384 sample (SUPERRES_SCALE_BITS 14), phase =
// bits 8..13.  Far from the bound of anything: one 4K frame in a few microseconds.
__global__ void __launch_bounds__(256)
upscale_kernel(Batch bt, int deb_second, int cdef_second)
{
    const FrameView *fvp = bt.v[blockIdx.z / 6];
    if (!fvp || !fvp->up_vis_w) return;
    const FrameView &fv = *fvp;
    const int which = (blockIdx.z % 6) / 3, p = blockIdx.z % 3, ss = p ? fv.ssx : 0;
    const int down = (fv.vis_w + ss) >> ss, up = (fv.up_vis_w + ss) >> ss, last_x = fv.work.pw(p) - 1, rows = fv.work.ph(p);
    const int stride = fv.work.stride[p];
    const int y = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y >= rows || x0 >= (fv.up_cw >> ss)) return;
    const int second = which ? cdef_second : deb_second;
    const uint8_t *src = (second ? fv.cdef.p[p] : fv.work.p[p]) + size_t(y) * stride;
    uint8_t *dst = (which ? fv.up_cdef.p[p] : fv.up_deb.p[p]) + size_t(y) * stride;
    const int step = ((down << 14) + up / 2) / up;
    const int err = up * step - (down << 14);
    const int initial = ((-((up - down) << 13) + up / 2) / up + (1 << 7) - err / 2) & 16383;
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        int v = 0;
        if (x < up) {
            const int sx = -(1 << 14) + initial + x * step;
            const int px = sx >> 14;
            const int16_t *taps = fv.upscale_filter + ((sx & 16383) >> 8) * 8;
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int xx = px + k - 3;
                xx = xx < 0 ? 0 : (xx > last_x ? last_x : xx);
                sum += int(taps[k]) * int(src[xx]);
            }
            v = (sum + 64) >> 7;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        pk |= uint32_t(v) << (8 * j);
    }
    *reinterpret_cast<uint32_t *>(dst + x0) = pk;
}

void launch_upscale(const SubmitInfo &s, hipStream_t st, Prof *prof, int deb_second, int cdef_second)
{
    if (!s.up_cw) return;
    uint64_t frames = 0;
    for (int c = 0; c < s.batch.n; c++) frames += s.stats[c] && s.stats[c]->desc.upscaled_width;
    if (!frames) return;
    ProfScope ps(prof, "upscale_kernel", frames * uint64_t(s.width + s.up_cw) * s.height * 3, st);   // both pictures, 1.5 planes each (4:2:0): read + written
    hipLaunchKernelGGL(upscale_kernel, dim3((s.up_cw / 4 + 255) / 256, s.height, 6 * s.batch.n), dim3(256), 0, st, s.batch, deb_second, cdef_second);
}

void launch_lr(const SubmitInfo &s, hipStream_t st, Prof *prof, int deb_second, int cdef_second)
{
    const int n = s.batch.n;
    uint64_t frames = 0;
    bool sgr_y = false, sgr_c = false;
    for (int c = 0; c < n; c++) {
        if (!s.stats[c]) continue;
        frames++;
        sgr_y |= s.stats[c]->lr_has_sgr[0];
        sgr_c |= s.stats[c]->lr_has_sgr[1] || s.stats[c]->lr_has_sgr[2];
    }
    if (!frames) return;
    const int lx0 = s.up_cw ? 0 : s.lr_x0, lx1 = s.up_cw ? s.up_cw : s.lr_x1;   // super-resolution: restoration covers the upscaled width
    const uint64_t ybytes = uint64_t(lx1 - lx0) * s.height;   // of the window (whole frame: width x height)
    const int cth = 64 >> s.ssy, ctw = s.ssx ? 32 : 64;   // chroma tile: one stripe high, 32 or 64 samples wide
    const int ystripes = (s.height + 8 + 63) / 64, cstripes = ((s.height >> s.ssy) + (8 >> s.ssy) + cth - 1) / cth;
    const TileGrid grid_y = tile_grid((lx1 - lx0 + 63) / 64, ystripes, n, lx0),
                   grid_c = tile_grid((((lx1 - lx0) >> s.ssx) + ctw - 1) / ctw, cstripes, 2 * n, lx0 >> s.ssx);
    const dim3 gy = xcd_grid(grid_y), gc = xcd_grid(grid_c);
    bool crop = s.up_cw != 0;
    for (int c = 0; c < n; c++) crop |= s.stats[c] && s.stats[c]->desc.crop;
#define EA1_LR(TW_, TH_, PL_, SGR_, GRID_, DIM_)                                                                                         \
    do {                                                                                                                                 \
        if (crop) hipLaunchKernelGGL((lr_stripe_kernel<TW_, TH_, 256, PL_, SGR_, true>), DIM_, dim3(256), 0, st, s.batch, deb_second, cdef_second, GRID_);  \
        else hipLaunchKernelGGL((lr_stripe_kernel<TW_, TH_, 256, PL_, SGR_, false>), DIM_, dim3(256), 0, st, s.batch, deb_second, cdef_second, GRID_);     \
    } while (0)
    {
        // names as rocprofv3 prints them (template arguments: tile, lanes, planes, SGR, CROP)
        ProfScope ps(prof, sgr_y ? (crop ? "lr_stripe_kernel<64,64,256,1,true,true>" : "lr_stripe_kernel<64,64,256,1,true,false>")
                                 : (crop ? "lr_stripe_kernel<64,64,256,1,false,true>" : "lr_stripe_kernel<64,64,256,1,false,false>"), frames * 2 * ybytes, st);
        if (sgr_y) EA1_LR(64, 64, 1, true, grid_y, gy);
        else EA1_LR(64, 64, 1, false, grid_y, gy);
    }
    if (s.ssx && s.ssy) {
        ProfScope ps(prof, sgr_c ? (crop ? "lr_stripe_kernel<32,32,256,2,true,true>" : "lr_stripe_kernel<32,32,256,2,true,false>")
                                 : (crop ? "lr_stripe_kernel<32,32,256,2,false,true>" : "lr_stripe_kernel<32,32,256,2,false,false>"), frames * ybytes, st);
        if (sgr_c) EA1_LR(32, 32, 2, true, grid_c, gc);
        else EA1_LR(32, 32, 2, false, grid_c, gc);
    } else if (s.ssx) {   // 4:2:2: stripes of 64 chroma rows, half the luma width
        ProfScope ps(prof, "lr_stripe_kernel<32,64,256,2,...>", frames * 2 * ybytes, st);
        if (sgr_c) EA1_LR(32, 64, 2, true, grid_c, gc);
        else EA1_LR(32, 64, 2, false, grid_c, gc);
    } else {              // 4:4:4: the chroma planes have the luma plane's geometry
        ProfScope ps(prof, "lr_stripe_kernel<64,64,256,2,...>", frames * 4 * ybytes, st);
        if (sgr_c) EA1_LR(64, 64, 2, true, grid_c, gc);
        else EA1_LR(64, 64, 2, false, grid_c, gc);
    }
#undef EA1_LR
}

}  // namespace ea1
