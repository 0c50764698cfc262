// mc_mfma_probe -- the ONE real MFMA attempt on inter prediction (round-2 verdict 5b; SURVEY.md 0.9 / 8(a) V2: "each pass = one
// exact integer sum + one rounding => MFMA-i8-eligible").  Luma of 32x32 tiles, single reference, both passes of the separable 8-tap
// filter (AV1 spec 7.11.3.4) as int8 matrix products on v_mfma_i32_32x32x32_i8, ONE wave per tile:
//   H pass   Mid (64 x 32) = In (64 rows x 64 columns, samples - 128) x T (64 x 32), T[c][j] = fh[c - j - o] / 2: a banded Toeplitz
//            matrix (o = the window's byte alignment), built in registers from the 8 taps; 2 row blocks x 2 k-steps = 4 MFMAs.
//            (2 acc + 128 * 128 + 4) >> 3 = (acc + 8194) >> 2 is the specification's Round2(sum, 3)
//   V pass   Out (32 x 32) = G (32 x 64) x Mid (64 x 32), G[r][k] = fv[k - r] / 2.  The 14-bit intermediates are split into a signed
//            high byte and a low byte minus 128, two products each: Out = 256 (G x Hi) + G x Lo + 128 * 64 ...  The H pass's accumulators
//            ARE the V pass's B operand (column on the lane, rows in the registers: no LDS round trip, no lane movement); G is built
//            with its k order permuted to the accumulator's row order.  4 MFMAs.
// All taps are even (the halved form fits int8 for every phase but the integer position, which is a copy and not handled here).
// Compared with a float-FMA kernel of the product's design (256 lanes per tile, window in LDS, float intermediates in LDS) on the same
// tiles; both are checked bit for bit against a plain integer reference on the host.
//   hipcc --offload-arch=gfx950 -O3 -I include tools/mc_mfma_probe.hip -o tools/bin/mc_mfma_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ea1_av1_tables.inc"

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const int16_t hSubpel[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};
__device__ __constant__ static const int16_t kSubpel[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};
__device__ __constant__ static const float kSubpelF[6][16][8] = {EA1_SUBPEL_FILTERS_VALUES};

struct Tile {
    int x, y;      // top-left luma sample of the tile
    int ix, iy;    // integer part of the position in the reference (window starts at ix - 3, iy - 3)
    int fx, fy;    // 1/16 phases, 1..15
};

// ------------------------------------------------------------------------------------------------ MFMA variant: one wave per tile
__global__ void __launch_bounds__(64)
mc_mfma_kernel(const uint8_t *__restrict__ ref, int stride, uint8_t *__restrict__ out, const Tile *__restrict__ tiles, int ntiles, int filter)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[64 * 16];   // 64 rows x 64 bytes (rows 39.. and columns 46.. hold whatever)
    __shared__ __attribute__((aligned(16))) int8_t tapsH[128], tapsV[128];   // zero-padded halved taps: tap t at index 64 + t
    __shared__ __attribute__((aligned(16))) uint8_t tr[32 * 32];     // result transposed for dword stores
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    const Tile tl = tiles[t];
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    const int wx0 = tl.ix - 3, wy0 = tl.iy - 3, ax = wx0 & ~3, o = wx0 - ax;
    // window: 39 rows x 12 dwords (48 bytes); the other rows / columns of the 64 x 64 operand never meet a non-zero matrix entry
    for (int i = l; i < 39 * 12; i += 64) {
        const int row = i / 12, k = i - row * 12;
        win[row * 16 + k] = *reinterpret_cast<const uint32_t *>(ref + size_t(wy0 + row) * stride + ax + 4 * k) ^ 0x80808080u;   // samples - 128 as int8
    }
    for (int i = l; i < 128; i += 64) {
        const int tt = i - 64;
        tapsH[i] = tt >= 0 && tt < 8 ? int8_t(kSubpel[filter][tl.fx][tt] / 2) : int8_t(0);
        tapsV[i] = tt >= 0 && tt < 8 ? int8_t(kSubpel[filter][tl.fy][tt] / 2) : int8_t(0);
    }
    __syncthreads();
    // ---- H pass ----
    v16i acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) acc[mb] = v16i{};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        // B = T: lane (column j = r, half h), element e <-> k = 32 s + 16 h + e: fh[k - j - o]
        union {
            v4i v;
            int8_t b[16];
        } bt;
        const int base = 64 + 32 * s + 16 * h - r - o;
#pragma unroll
        for (int e = 0; e < 16; e++) bt.b[e] = tapsH[base + e];
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
            const v4i a = *reinterpret_cast<const v4i *>(&win[(32 * mb + r) * 16 + 8 * s + 4 * h]);   // A: row 32 mb + r, bytes 32 s + 16 h ..
            acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bt.v, acc[mb], 0, 0, 0);
        }
    }
    // Round2(sum, 3), then the split: mid = 256 hi + lo_u, as int8: hi (signed), lo_u - 128
    v4i bhi[2], blo[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        union {
            v4i v;
            uint8_t b[16];
        } hi, lo;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int mid = (acc[mb][reg] + 8194) >> 2;
            hi.b[reg] = uint8_t(mid >> 8);
            lo.b[reg] = uint8_t(mid) ^ 0x80;
        }
        bhi[mb] = hi.v;
        blo[mb] = lo.v;
    }
    // ---- V pass: A = G, lane (row r, half h), element j <-> k = 32 s + (j & 3) + 8 (j >> 2) + 4 h (the accumulator's row order) ----
    v16i ohi = v16i{}, olo = v16i{};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        union {
            v4i v;
            int8_t b[16];
        } g;
#pragma unroll
        for (int j = 0; j < 16; j++) g.b[j] = tapsV[64 + 32 * s + (j & 3) + 8 * (j >> 2) + 4 * h - r];
        ohi = __builtin_amdgcn_mfma_i32_32x32x32_i8(g.v, bhi[s], ohi, 0, 0, 0);
        olo = __builtin_amdgcn_mfma_i32_32x32x32_i8(g.v, blo[s], olo, 0, 0, 0);
    }
    // sum_v = 2 (256 ohi + olo + 128 * 64) ; Round2(sum_v, 11) = (256 ohi + olo + 8192 + 512) >> 10
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        int v = (256 * ohi[reg] + olo[reg] + 8704) >> 10;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        tr[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = uint8_t(v);
    }
    __syncthreads();
    for (int i = l; i < 32 * 8; i += 64) {
        const int row = i >> 3, k = i & 7;
        *reinterpret_cast<uint32_t *>(out + size_t(tl.y + row) * stride + tl.x + 4 * k) = reinterpret_cast<const uint32_t *>(tr)[row * 8 + k];
    }
}

// ---- the same with the glue tightened: operands built from aligned dword reads + v_alignbyte, biases folded into the accumulators'
// initial values, bytes packed with v_perm, and the V pass computed TRANSPOSED (Out^T = Mid^T x G^T: the H accumulators serve as the
// A operand unchanged) so that a lane ends up with 4 x 4 consecutive samples of one output row: dword stores, no LDS transpose ----
__device__ __forceinline__ uint32_t lds_unaligned_dword(const int8_t *base, int idx)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base) + (idx >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], idx & 3);
}

template <int FLAGS>   // 1: T from dword reads, 2: bytes packed with v_perm, 4: G from dword reads, 8: transposed V pass + direct stores
__global__ void __launch_bounds__(64)
mc_mfma2_kernel(const uint8_t *__restrict__ ref, int stride, uint8_t *__restrict__ out, const Tile *__restrict__ tiles, int ntiles, int filter)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[64 * 16];
    __shared__ __attribute__((aligned(16))) int8_t tapsH[136], tapsV[136];
    __shared__ __attribute__((aligned(16))) uint8_t tr[32 * 32];
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    const Tile tl = tiles[t];
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    const int wx0 = tl.ix - 3, wy0 = tl.iy - 3, ax = wx0 & ~3, o = wx0 - ax;
    for (int i = l; i < 39 * 12; i += 64) {
        const int row = i / 12, k = i - row * 12;
        win[row * 16 + k] = (FLAGS & 16) ? uint32_t(i * 2654435761u) : *reinterpret_cast<const uint32_t *>(ref + size_t(wy0 + row) * stride + ax + 4 * k) ^ 0x80808080u;   // 16: no global reads (timing only)
    }
    for (int i = l; i < 136; i += 64) {
        const int tt = i - 64;
        tapsH[i] = tt >= 0 && tt < 8 ? int8_t(kSubpel[filter][tl.fx][tt] / 2) : int8_t(0);
        tapsV[i] = tt >= 0 && tt < 8 ? int8_t(kSubpel[filter][tl.fy][tt] / 2) : int8_t(0);
    }
    __syncthreads();
    v16i acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) acc[mb][reg] = 8194;   // (acc + 8194) >> 2 = Round2(sum, 3)
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int base = 64 + 32 * s + 16 * h - r - o;
        union {
            v4i v;
            int8_t b[16];
        } bt;
        if (FLAGS & 1) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(tapsH) + (base >> 2);
            const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
            const int sh = base & 3;
            bt.v[0] = int(__builtin_amdgcn_alignbyte(d1, d0, sh));
            bt.v[1] = int(__builtin_amdgcn_alignbyte(d2, d1, sh));
            bt.v[2] = int(__builtin_amdgcn_alignbyte(d3, d2, sh));
            bt.v[3] = int(__builtin_amdgcn_alignbyte(d4, d3, sh));
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) bt.b[e] = tapsH[base + e];
        }
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
            const v4i a = *reinterpret_cast<const v4i *>(&win[(32 * mb + r) * 16 + 8 * s + 4 * h]);
            acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bt.v, acc[mb], 0, 0, 0);
        }
    }
    v4i ahi[2], alo[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        if (FLAGS & 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t m0 = uint32_t(acc[mb][4 * q] >> 2), m1 = uint32_t(acc[mb][4 * q + 1] >> 2), m2 = uint32_t(acc[mb][4 * q + 2] >> 2), m3 = uint32_t(acc[mb][4 * q + 3] >> 2);
                // v_perm_b32(hi, lo, sel): selector bytes 0..3 pick from lo, 4..7 from hi
                const uint32_t lo01 = __builtin_amdgcn_perm(m1, m0, 0x0c0c0400), lo23 = __builtin_amdgcn_perm(m3, m2, 0x04000c0c);
                const uint32_t hi01 = __builtin_amdgcn_perm(m1, m0, 0x0c0c0501), hi23 = __builtin_amdgcn_perm(m3, m2, 0x05010c0c);
                alo[mb][q] = int((lo01 | lo23) ^ 0x80808080u);
                ahi[mb][q] = int(hi01 | hi23);
            }
        } else {
            union {
                v4i v;
                uint8_t b[16];
            } hi, lo;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int mid = acc[mb][reg] >> 2;
                hi.b[reg] = uint8_t(mid >> 8);
                lo.b[reg] = uint8_t(mid) ^ 0x80;
            }
            ahi[mb] = hi.v;
            alo[mb] = lo.v;
        }
    }
    v16i ohi, olo;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        ohi[reg] = 0;
        olo[reg] = 8704;   // (256 ohi + olo + 8704) >> 10 = Round2(sum_v, 11)
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        union {
            v4i v;
            int8_t b[16];
        } g;
        if (FLAGS & 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) g.v[q] = int(lds_unaligned_dword(tapsV, 64 + 32 * s + 8 * q + 4 * h - r));
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) g.b[j] = tapsV[64 + 32 * s + (j & 3) + 8 * (j >> 2) + 4 * h - r];
        }
        if (FLAGS & 8) {
            // transposed: D[j][r] = sum_k Mid[k][j] G[r][k]; A = the H result (lane: column j, elements in accumulator row order), B = G
            ohi = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[s], g.v, ohi, 0, 0, 0);
            olo = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[s], g.v, olo, 0, 0, 0);
        } else {
            ohi = __builtin_amdgcn_mfma_i32_32x32x32_i8(g.v, ahi[s], ohi, 0, 0, 0);
            olo = __builtin_amdgcn_mfma_i32_32x32x32_i8(g.v, alo[s], olo, 0, 0, 0);
        }
    }
    if (FLAGS & 8) {
        // lane (r, h): output row r, columns 8 q + 4 h + (0..3) in registers 4 q .. 4 q + 3
        uint8_t *orow = out + size_t(tl.y + r) * stride + tl.x + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int v = (256 * ohi[4 * q + e] + olo[4 * q + e]) >> 10;
                asm volatile("" : "+v"(v));   // keeps hipcc from selecting v_ashr_pk_u8_i32, whose upper result bits are stale on gfx950
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                pk |= uint32_t(v) << (8 * e);
            }
            if (!(FLAGS & 32) || pk == 0x12345678u) *reinterpret_cast<uint32_t *>(orow + 8 * q) = pk;   // 32: (practically) no stores (timing only)
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            int v = (256 * ohi[reg] + olo[reg]) >> 10;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            tr[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = uint8_t(v);
        }
        __syncthreads();
        for (int i = l; i < 32 * 8; i += 64) {
            const int row = i >> 3, k = i & 7;
            *reinterpret_cast<uint32_t *>(out + size_t(tl.y + row) * stride + tl.x + 4 * k) = reinterpret_cast<const uint32_t *>(tr)[row * 8 + k];
        }
    }
}

// ------------------------------------------------------------------------------------------------ float-FMA variant (the product's design)
template <int SH>
__device__ __forceinline__ float round_shift_f(float v)
{
    return __builtin_floorf(__builtin_fmaf(v, 1.0f / float(1 << SH), 0.5f));
}

__global__ void __launch_bounds__(256)
mc_fma_kernel(const uint8_t *__restrict__ ref, int stride, uint8_t *__restrict__ out, const Tile *__restrict__ tiles, int ntiles, int filter)
{
    constexpr int TS = 32, RY = TS + 7, NDY = (TS + 13) / 4, GY = TS / 4;
    __shared__ uint32_t win[RY * NDY];
    __shared__ __attribute__((aligned(16))) float mid[RY * TS];
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    const Tile tl = tiles[t];
    const int tid = threadIdx.x;
    const int wx0 = tl.ix - 3, wy0 = tl.iy - 3, ax = wx0 & ~3, o = wx0 - ax;
    for (int i = tid; i < RY * NDY; i += 256) {
        const int row = i / NDY, k = i - row * NDY;
        win[i] = *reinterpret_cast<const uint32_t *>(ref + size_t(wy0 + row) * stride + ax + 4 * k);
    }
    __syncthreads();
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = kSubpelF[filter][tl.fx][k];
    for (int hh = tid; hh < RY * GY; hh += 256) {
        const int row = hh / GY, g = hh - row * GY;
        const uint32_t *rowp = win + row * NDY + g;
        const uint32_t w0 = __builtin_amdgcn_alignbyte(rowp[1], rowp[0], o), w1 = __builtin_amdgcn_alignbyte(rowp[2], rowp[1], o), w2 = __builtin_amdgcn_alignbyte(rowp[3], rowp[2], o);
        float bb[12];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            bb[i] = float((w0 >> (8 * i)) & 255u);
            bb[4 + i] = float((w1 >> (8 * i)) & 255u);
            bb[8 + i] = float((w2 >> (8 * i)) & 255u);
        }
        float4 m;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) s = __builtin_fmaf(f[k], bb[j + k], s);
            (&m.x)[j] = round_shift_f<3>(s);
        }
        *reinterpret_cast<float4 *>(mid + row * TS + 4 * g) = m;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = kSubpelF[filter][tl.fy][k];
    {
        const int row = tid / GY, g = tid - row * GY;   // 32 rows x 8 groups = 256 lanes
        const float *col = mid + row * TS + 4 * g;
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 w = *reinterpret_cast<const float4 *>(col + k * TS);
            s[0] = __builtin_fmaf(f[k], w.x, s[0]);
            s[1] = __builtin_fmaf(f[k], w.y, s[1]);
            s[2] = __builtin_fmaf(f[k], w.z, s[2]);
            s[3] = __builtin_fmaf(f[k], w.w, s[3]);
        }
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) pk = __builtin_amdgcn_cvt_pk_u8_f32(round_shift_f<11>(s[j]), j, pk);
        *reinterpret_cast<uint32_t *>(out + size_t(tl.y + row) * stride + tl.x + 4 * g) = pk;
    }
}

int main()
{
    const int W = 3840, H = 2160, stride = 3840, filter = 0;
    std::vector<uint8_t> ref(size_t(stride) * H), want(size_t(stride) * H, 0), got(size_t(stride) * H);
    srand(3);
    for (auto &v : ref) v = uint8_t(rand());
    for (int i = 0; i < 4096; i++) ref[size_t(rand() % H) * stride + rand() % W] = (i & 1) ? 255 : 0;   // extremes
    std::vector<Tile> tiles;
    for (int y = 0; y < H - 31; y += 32)
        for (int x = 0; x < W; x += 32) {
            Tile t;
            t.x = x;
            t.y = y;
            // the tile's own position plus a motion vector of up to 64 samples, as the graded streams have: neighbouring
            // tiles read neighbouring windows (uniformly random windows make BOTH kernels wait for HBM: 3.5 TB/s of 128-byte sectors)
            t.ix = x + rand() % 129 - 64;
            t.iy = y + rand() % 129 - 64;
            t.ix = t.ix < 8 ? 8 : (t.ix > W - 96 ? W - 96 : t.ix);
            t.iy = t.iy < 8 ? 8 : (t.iy > H - 96 ? H - 96 : t.iy);
            t.fx = 1 + rand() % 15;
            t.fy = 1 + rand() % 15;
            tiles.push_back(t);
        }
    const int n1 = int(tiles.size());
    const int copies = 16;   // the product launches one frame of 32 chains at once: enough tiles to fill every SIMD many times over
    for (int c = 1; c < copies; c++)
        for (int i = 0; i < n1; i++) tiles.push_back(tiles[size_t(i)]);   // (the copies write the same samples again)
    const int n = int(tiles.size());
    for (size_t ti = 0; ti < size_t(n1); ti++) {   // the specification's arithmetic, plain
        const Tile &t = tiles[ti];
        static int mid[39][32];
        for (int r = 0; r < 39; r++)
            for (int c = 0; c < 32; c++) {
                int s = 0;
                for (int k = 0; k < 8; k++) s += hSubpel[filter][t.fx][k] * ref[size_t(t.iy - 3 + r) * stride + t.ix - 3 + c + k];
                mid[r][c] = (s + 4) >> 3;
            }
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                int s = 0;
                for (int k = 0; k < 8; k++) s += hSubpel[filter][t.fy][k] * mid[r + k][c];
                const int v = (s + 1024) >> 11;
                want[size_t(t.y + r) * stride + t.x + c] = uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
    uint8_t *dref, *dout;
    Tile *dt;
    CK(hipMalloc(&dref, ref.size()));
    CK(hipMalloc(&dout, ref.size()));
    CK(hipMalloc(&dt, n * sizeof(Tile)));
    CK(hipMemcpy(dref, ref.data(), ref.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tiles.data(), n * sizeof(Tile), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 50;
    for (int variant = 0; variant < 11; variant++) {
        CK(hipMemset(dout, 0, ref.size()));
        float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < (pass ? reps : 2); k++) {
                if (variant == 0) hipLaunchKernelGGL(mc_mfma_kernel, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 2) hipLaunchKernelGGL(mc_mfma2_kernel<0>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 3) hipLaunchKernelGGL(mc_mfma2_kernel<1>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 4) hipLaunchKernelGGL(mc_mfma2_kernel<2>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 5) hipLaunchKernelGGL(mc_mfma2_kernel<4>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 6) hipLaunchKernelGGL(mc_mfma2_kernel<8>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 7) hipLaunchKernelGGL(mc_mfma2_kernel<15>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 8) hipLaunchKernelGGL(mc_mfma2_kernel<15 + 16>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 9) hipLaunchKernelGGL(mc_mfma2_kernel<15 + 32>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else if (variant == 10) hipLaunchKernelGGL(mc_mfma2_kernel<15 + 48>, dim3(n), dim3(64), 0, 0, dref, stride, dout, dt, n, filter);
                else hipLaunchKernelGGL(mc_fma_kernel, dim3(n), dim3(256), 0, 0, dref, stride, dout, dt, n, filter);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(got.data(), dout, ref.size(), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (const Tile &t : tiles)
            for (int r = 0; r < 32; r++)
                for (int c = 0; c < 32; c++) bad += got[size_t(t.y + r) * stride + t.x + c] != want[size_t(t.y + r) * stride + t.x + c];
        if (bad && variant == 6) {
            int shown = 0;
            for (const Tile &t : tiles) {
                int tb = 0;
                for (int r = 0; r < 32; r++)
                    for (int c = 0; c < 32; c++) tb += got[size_t(t.y + r) * stride + t.x + c] != want[size_t(t.y + r) * stride + t.x + c];
                if (!tb) continue;
                printf("tile at (%d,%d) fx %d fy %d ix %d (o = %d): %d wrong; map of wrong samples (rows down):\n", t.x, t.y, t.fx, t.fy, t.ix, (t.ix - 3) & 3, tb);
                for (int r = 0; r < 32; r++) {
                    for (int c = 0; c < 32; c++) putchar(got[size_t(t.y + r) * stride + t.x + c] != want[size_t(t.y + r) * stride + t.x + c] ? 'X' : '.');
                    printf("   got %3d want %3d\n", got[size_t(t.y + r) * stride + t.x + 5], want[size_t(t.y + r) * stride + t.x + 5]);
                }
                if (++shown == 2) break;
            }
        }
        printf("%-34s %8.1f us per launch of %d 32x32 luma tiles (one reference), %zu samples differ from the integer reference\n",
               variant == 0 ? "int8 MFMA, one wave per tile:" : (variant == 1 ? "float FMA, 256 lanes per tile:" : (variant == 2 ? "int8 MFMA v2 flags 0:" : (variant == 3 ? "int8 MFMA v2 flag T dwords:" : (variant == 4 ? "int8 MFMA v2 flag perm pack:" : (variant == 5 ? "int8 MFMA v2 flag G dwords:" : (variant == 6 ? "int8 MFMA v2 flag transposed V:" : (variant == 7 ? "int8 MFMA v2 all flags:" : (variant == 8 ? "  ... without the global reads:" : (variant == 9 ? "  ... without the stores:" : "  ... without reads and stores:"))))))))), ms * 1e3 / reps, n, bad);
    }
    return 0;
}
