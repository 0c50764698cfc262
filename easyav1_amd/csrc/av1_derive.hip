// Hand-over of a parsed frame: what the in-loop filters need to know about the frame, derived ON THE
// GPU from the block and transform-block records that were uploaded anyway (the GPU analogue of
// dav1d's lf_mask.c, /root/reference/cmake/dav1d.cmake:980):
//   block map        index of the block covering each 4x4 luma unit
//   transform maps   index of the transform block covering each 4x4-sample unit, per plane
//   luma_edge / chroma_edge   deblocking edge class of every unit's left and top edge (AV1 spec 7.14.2/3:
//                    an edge exists where a transform block starts, except inside a skipped inter block;
//                    the filter length follows the smaller of the two transform sizes across the edge)
//   lvl_y / lvl_c    deblocking levels of the unit's block
//   unit_flags       per 8x8: every block in it is skipped (CDEF leaves it alone)
//   transform items / inter tiles   the two big work lists of a frame (one entry per coded transform block, one
//                    per prediction tile): the host counts them while it validates, the entries are written here
// All of it is data-parallel over blocks, transform blocks and units; on the host it cost half of
// ea1_job_create.  One launch serves every job of a hand-over batch (blockIdx.y = job).  The records have been validated by the host before these kernels are enqueued
// (positions inside the frame, blocks tile the frame, transform blocks tile their block), so the
// scatters stay inside the maps.  Runs on the context's upload stream, behind the copy of the records.
#include "av1_dev.h"
#include "ea1_common.h"

namespace ea1 {

__global__ void __launch_bounds__(256)
derive_block_map_kernel(DeriveBatch batch)
{
    const DeriveArgs &a = *batch.a[blockIdx.y];
    const uint32_t bi = blockIdx.x * 256 + threadIdx.x;
    // the list cursors and the zero coefficients of the job: cleared here, ahead of every kernel that uses them
    if (bi < uint32_t(kDeriveCursors)) a.cursors[bi] = 0;
    if (bi < kZeroCoefs / 2) reinterpret_cast<uint32_t *>(a.coefs)[bi] = 0;
    if (bi >= a.num_blocks) return;
    const ea1_block b = a.blocks[bi];
    const int n4w = 1 << (b.bwl - 2), n4h = 1 << (b.bhl - 2);
    uint32_t *row = a.map + size_t(b.y4) * a.w4 + b.x4;
    for (int j = 0; j < n4h; j++, row += a.w4)
        for (int i = 0; i < n4w; i++) row[i] = bi;
}

__global__ void __launch_bounds__(256)
derive_tx_map_kernel(DeriveBatch batch)
{
    const DeriveArgs &a = *batch.a[blockIdx.y];
    const uint32_t ti = blockIdx.x * 256 + threadIdx.x;
    if (ti >= a.num_tx) return;
    const ea1_txblock t = a.txs[ti];
    if (t.plane > 1) return;   // U and V share one transform grid (checked by the host): only U's map is made
    const int n4w = 1 << ((t.size & 15) - 2), n4h = 1 << ((t.size >> 4) - 2);
    const int mw = t.plane ? a.cw4 : a.w4;
    uint32_t *row = a.txmap[t.plane] + size_t(t.y4) * mw + t.x4;
    for (int j = 0; j < n4h; j++, row += mw)
        for (int i = 0; i < n4w; i++) row[i] = ti;
}

// offset of a wedge mask in the context's table (wedge.cpp: sizes 8x8, 8x16, 8x32, 16x8, ..., 32 masks each)
__device__ __forceinline__ uint32_t wedge_offset_dev(int wl2, int hl2, int sign, int index)
{
    uint32_t off = 0;
    for (int a = 3; a <= 5; a++)
        for (int b = 3; b <= 5; b++) {
            if (a == wl2 && b == hl2) return off + uint32_t((sign * 16 + index) << (wl2 + hl2));
            off += 32u << (a + b);
        }
    return 0;
}

// One lane per block: the transform items of its coded transform blocks and its prediction tiles, placed into the
// buckets the host has sized (the same rules, see job_prepare in av1_recon.cpp).  A frame's 10^5 items go to a few
// dozen lists: atomics on the global cursors alone serialise in the L2 (370 us for a 4K frame).  So a workgroup
// counts its items per list in LDS first, takes ONE range per list from the global cursor, and hands out the places
// inside it with LDS atomics.
// entries of a coded block in the corner layout (EA1_COEFS_DIAGONAL frames: nz_log2 = class - 2 | D << 2); even
__device__ __forceinline__ uint32_t corner_entries(const ea1_txblock &t)
{
    const int n = 4 << (t.nz_log2 & 3);
    const int w = 1 << (t.size & 15), h = 1 << (t.size >> 4);
    return uint32_t((w < n ? w : n) * (h < n ? h : n));
}

struct ItemOf {
    int bucket;     // cursor index, -1: no item
    TxItem it;
};

__device__ __forceinline__ ItemOf tx_item_of(const DeriveArgs &a, const ea1_txblock &t, bool inter, bool interintra, const bool (&plane_has)[3])
{
    ItemOf r;
    r.bucket = -1;
    const bool has = t.coef_off != EA1_NO_COEFS;
    // inter-intra blocks get their residual through the residual plane: their transform blocks without
    // coefficients clear it when another one of the block and plane has some (zero coefficients, smallest corner)
    const bool zero_fill = interintra && !has && plane_has[t.plane];
    if (!has && !zero_fill) return r;
    const int wl = t.size & 15, hl = t.size >> 4;
    TxItem it;
    memset(&it, 0, sizeof it);
    it.coef_off = has ? a.coef_rebase + t.coef_off : 0;
    it.x = uint16_t(t.x4 * 4);
    it.y = uint16_t(t.y4 * 4);
    it.plane = t.plane;
    it.txtp = t.txtp;
    it.intra = (inter && !interintra) ? 0 : 1;
    it.corner = (has && a.corner_layout) ? 1 : 0;
    const bool hinted = has && t.nz_log2 >= 2 && t.nz_log2 <= 5;
    int bucket = 0;
    if (has && a.diagonal_layout) {
        // packed diagonals: corner class in the low bits, last coded diagonal above; the corner-layout place of the
        // block is handed out by the caller (coef_off), the packed source travels in src24
        it.nz_log2 = uint8_t(2 + (t.nz_log2 & 3));
        bucket = t.nz_log2 & 3;
        it.corner = uint8_t(1 + (t.nz_log2 >> 2));
        const uint32_t src = (t.coef_off - a.packed_lo) >> 1;
        it.src24[0] = uint8_t(src);
        it.src24[1] = uint8_t(src >> 8);
        it.src24[2] = uint8_t(src >> 16);
    } else if (hinted) {
        it.nz_log2 = t.nz_log2;   // the parser's own bound (it knows the end-of-block position)
        bucket = t.nz_log2 - 2;
    } else {
        // no hint (dense layout only): extent of the non-zero coefficients from the uploaded block itself
        int ext = 0;
        const int sw = wl < 5 ? 1 << wl : 32, sh = hl < 5 ? 1 << hl : 32;
        const int16_t *c = a.coefs + it.coef_off;
        for (int yy = 0; has && yy < sh; yy++)
            for (int xx = sw - 1; xx >= 0; xx--)
                if (c[yy * sw + xx]) {
                    const int m = xx > yy ? xx : yy;
                    ext = ext > m ? ext : m;
                    break;
                }
        it.nz_log2 = uint8_t(ext < 4 ? 2 : (ext < 8 ? 3 : (ext < 16 ? 4 : 5)));
    }
    r.bucket = tx_size_index(wl, hl) * 4 + bucket;
    r.it = it;
    return r;
}

__global__ void __launch_bounds__(256)
derive_items_kernel(DeriveBatch batch)
{
    __shared__ uint32_t cnt[kDeriveCursors], base[kDeriveCursors];
    const DeriveArgs &a = *batch.a[blockIdx.y];
    for (int i = threadIdx.x; i < kDeriveCursors; i += 256) cnt[i] = 0;
    __syncthreads();
    const uint32_t bi = blockIdx.x * 256 + threadIdx.x;
    ea1_block b;
    memset(&b, 0, sizeof b);
    bool mine = bi < a.num_blocks;
    if (mine) {
        b = a.blocks[bi];
        mine = b.x4 * 4 >= a.strip_x0 && b.x4 * 4 < a.strip_x1;   // else: another rank's block
    }
    const int x = b.x4 * 4, y = b.y4 * 4, bw = 1 << b.bwl, bh = 1 << b.bhl;
    const bool inter = (b.flags & EA1_BLK_INTER) != 0;
    const bool interintra = inter && (b.flags & EA1_BLK_INTERINTRA);
    const bool sub8 = b.bwl == 2 || b.bhl == 2;
    // inter tiles (blocks flagged for warped motion are the host's: whether the warp is usable is decided there)
    const bool tiles = mine && inter && !sub8 && !(b.flags & EA1_BLK_WARP);
    const bool comp = (b.flags & EA1_BLK_COMPOUND) != 0;
    const bool ext = comp && b.comp_type != EA1_COMP_AVG;
    const int mn = bw < bh ? bw : bh;
    const int ts = mn < 32 ? mn : 32;
    const int mc_cls = ts == 8 ? 0 : (ts == 16 ? 1 : 2);
    const int ntiles = (bw / ts) * (bh / ts);
    const int mc_cursor = kTxSizes * 4 + (ext ? 3 : 0) + mc_cls;
    bool plane_has[3] = {false, false, false};
    if (mine && interintra)
        for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++) {
            const ea1_txblock t = a.txs[ti];
            if (t.coef_off != EA1_NO_COEFS) plane_has[t.plane] = true;
        }
    // pass 1: how many items of each list this workgroup makes
    if (tiles) atomicAdd(&cnt[mc_cursor], uint32_t(ntiles));
    if (mine)
        for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++) {
            const ea1_txblock t = a.txs[ti];
            const bool has = t.coef_off != EA1_NO_COEFS;
            if (!has && !(interintra && plane_has[t.plane])) continue;
            const bool hinted = has && t.nz_log2 >= 2 && t.nz_log2 <= 5;
            const int bk = (has && a.diagonal_layout) ? (t.nz_log2 & 3) : (hinted ? t.nz_log2 - 2 : 0);
            atomicAdd(&cnt[tx_size_index(t.size & 15, t.size >> 4) * 4 + bk], 1u);
            if (has && a.diagonal_layout) atomicAdd(&cnt[kCoefCursor], corner_entries(t));
        }
    __syncthreads();
    // one range per list for the workgroup
    for (int i = threadIdx.x; i < kDeriveCursors; i += 256) {
        const uint32_t n = cnt[i];
        base[i] = n ? atomicAdd(&a.cursors[i], n) : 0;
        cnt[i] = 0;
    }
    __syncthreads();
    // pass 2: the items
    if (tiles) {
        McItem mi;
        memset(&mi, 0, sizeof mi);
        mi.flags = uint8_t((comp ? 2 : 0) | ((bw >> a.ssx) == 4 ? 4 : 0) | ((bh >> a.ssy) == 4 ? 8 : 0));
        mi.filt = b.mode_y;
        mi.ref[0] = b.ref[0];
        mi.ref[1] = b.ref[1];
        mi.mv[0][0] = b.mv[0][0];
        mi.mv[0][1] = b.mv[0][1];
        mi.mv[1][0] = b.mv[1][0];
        mi.mv[1][1] = b.mv[1][1];
        mi.comp_type = ext ? b.comp_type : uint8_t(EA1_COMP_AVG);
        mi.comp_arg = uint8_t(b.comp_type == EA1_COMP_WEDGE ? 0 : b.comp_arg);
        mi.mask_stride_log2 = b.bwl;
        McItem *list = ext ? a.mc_ext_items[mc_cls] : a.mc_items[mc_cls];
        uint32_t pos = (ext ? a.mc_ext_base[mc_cls] : a.mc_base[mc_cls]) + base[mc_cursor] + atomicAdd(&cnt[mc_cursor], uint32_t(ntiles));
        const uint32_t wedge0 = (ext && b.comp_type == EA1_COMP_WEDGE) ? wedge_offset_dev(b.bwl, b.bhl, (b.comp_arg >> 4) & 1, b.comp_arg & 15) : 0;
        for (int ty = 0; ty < bh; ty += ts)
            for (int tx_ = 0; tx_ < bw; tx_ += ts) {
                mi.x8 = uint16_t((x + tx_) / 8);
                mi.y8 = uint16_t((y + ty) / 8);
                mi.ox8 = uint8_t(tx_ / 8);
                mi.oy8 = uint8_t(ty / 8);
                if (ext && b.comp_type == EA1_COMP_WEDGE) mi.mask_off = wedge0 + uint32_t(ty * bw + tx_);
                list[pos++] = mi;
            }
    }
    if (mine)
        for (uint32_t ti = b.first_tx; ti < b.first_tx + b.n_tx; ti++) {
            const ea1_txblock t = a.txs[ti];
            ItemOf io = tx_item_of(a, t, inter, interintra, plane_has);
            if (io.bucket < 0) continue;
            if (a.diagonal_layout && t.coef_off != EA1_NO_COEFS)   // its corner-layout place behind the zeros
                io.it.coef_off = kZeroCoefs + base[kCoefCursor] + atomicAdd(&cnt[kCoefCursor], corner_entries(t));
            const int cls = io.bucket >> 2, bk = io.bucket & 3;
            const uint32_t pos = a.tx_base[cls][bk] + base[io.bucket] + atomicAdd(&cnt[io.bucket], 1u);
            a.tx_items[cls][pos] = io.it;
        }
}

// geometry of a block in a plane (see include/easyav1_amd_video.h): in a subsampled direction the chroma of a block with a
// 4-sample side starts at the 8-aligned luma position and covers at least 4 chroma samples
__device__ __forceinline__ void block_origin(const ea1_block &b, int pl, int ssx, int ssy, int &x, int &y)
{
    if (pl == 0) {
        x = b.x4 * 4;
        y = b.y4 * 4;
    } else {
        x = ((b.x4 & ~ssx) * 4) >> ssx;
        y = ((b.y4 & ~ssy) * 4) >> ssy;
    }
}

// one lane per unit: luma units first (w4 * h4; the lanes of the even units also make unit_flags of their 8x8), then chroma units
// (cw4 * ch4, with lvl_c)
__global__ void __launch_bounds__(256)
derive_edges_kernel(DeriveBatch batch)
{
    const DeriveArgs &a = *batch.a[blockIdx.y];
    const int nl = a.w4 * a.h4, nc = a.cw4 * a.ch4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nl + nc) return;
    const int pl = i >= nl ? 1 : 0;
    const int u = pl ? i - nl : i;
    const int mw = pl ? a.cw4 : a.w4;
    const int uy = u / mw, ux = u - uy * mw;
    const int ssx = a.ssx, ssy = a.ssy;
    const uint32_t *txm = a.txmap[pl];
    const ea1_txblock t = a.txs[txm[u]];
    // chroma units belong to the block that owns the chroma of their luma area: the last (odd) 4x4 luma unit in a subsampled direction
    const ea1_block b = a.blocks[pl ? a.map[size_t((uy << ssy) + ssy) * a.w4 + (ux << ssx) + ssx] : a.map[size_t(uy) * a.w4 + ux]];
    const bool inner_off = (b.flags & EA1_BLK_SKIP) && (b.flags & EA1_BLK_INTER);
    int bx, by;
    block_origin(b, pl, ssx, ssy, bx, by);
    uint32_t e = 0;
    // AV1 spec 7.14.2 "onScreen" (dav1d lf_apply: columns / rows up to w4 / h4 = (size + 3) >> 2): a unit whose luma position lies at or
    // beyond the frame's true size is not filtered in either pass -- matters when crop >= 4, since CDEF of visible samples reads there
    const bool on_screen = (pl ? (4 * ux) << ssx : 4 * ux) < a.vis_w && (pl ? (4 * uy) << ssy : 4 * uy) < a.vis_h;
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {   // 0: left edge (vertical), 1: top edge
        const int pos = dir ? uy : ux;
        if (!on_screen || pos == 0 || (dir ? t.y4 : t.x4) != pos) continue;
        const bool block_edge = (dir ? by : bx) == pos * 4;
        if (!block_edge && inner_off) continue;
        const ea1_txblock pt = a.txs[txm[dir ? u - mw : u - 1]];
        const int cur = dir ? t.size >> 4 : t.size & 15, prev = dir ? pt.size >> 4 : pt.size & 15;
        const int m = cur < prev ? cur : prev;   // log2 of the smaller transform dimension
        const int cls = pl ? (m >= 3 ? 2 : 1) : (m >= 4 ? 3 : (m == 3 ? 2 : 1));
        e |= uint32_t(cls) << (2 * dir);
    }
    const bool per_block = a.lf_per_block != 0;
    if (pl == 0) {
        a.luma_edge[u] = uint8_t(e);
        a.lvl_y[u] = per_block ? uint16_t(b.lf_level[0] | (b.lf_level[1] << 8)) : uint16_t(a.lf_level[0] | (a.lf_level[1] << 8));
        if (!((ux | uy) & 1)) {   // (w4, h4 are even)
            bool all_skip = true;
#pragma unroll
            for (int k = 0; k < 4; k++) all_skip &= (a.blocks[a.map[size_t(uy + (k >> 1)) * a.w4 + ux + (k & 1)]].flags & EA1_BLK_SKIP) != 0;
            a.unit_flags[size_t(uy >> 1) * a.w8 + (ux >> 1)] = uint8_t(all_skip ? 1 : 0);
        }
    } else {
        a.chroma_edge[u] = uint8_t(e);
        a.lvl_c[u] = per_block ? uint16_t(b.lf_level[2] | (b.lf_level[3] << 8)) : uint16_t(a.lf_level[2] | (a.lf_level[3] << 8));
    }
}

// EA1_COEFS_DIAGONAL: the packed diagonals of every coded block into its corner-layout place.  16 lanes per
// transform item (a lane makes two adjacent entries per step), grid.y = transform size, grid.z = job.
template <int K>
__device__ __forceinline__ void expand_class(const DeriveArgs &a, uint32_t item, int l)
{
    constexpr int wl = kTxLog2[K][0], hl = kTxLog2[K][1];
    constexpr int SW = wl < 5 ? 1 << wl : 32, SH = hl < 5 ? 1 << hl : 32;
    if (item >= a.tx_count[K]) return;
    TxItem *ip = a.tx_items[K] + item;
    const TxItem it = *ip;
    if (!it.corner) return;   // zero-fill item: no coefficients of its own
    const int D = it.corner - 1, n = 1 << it.nz_log2;
    const int cw = SW < n ? SW : n, ch = SH < n ? SH : n;
    const int16_t *src = a.packed + 2 * (uint32_t(it.src24[0]) | (uint32_t(it.src24[1]) << 8) | (uint32_t(it.src24[2]) << 16));
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.coefs + it.coef_off);
    for (int o = l; o < cw * ch / 2; o += 16) {
        const int y = (2 * o) / cw, x = (2 * o) - y * cw;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int d = x + k + y;
            if (d <= D) {
                const int y0 = d - (SW - 1) > 0 ? d - (SW - 1) : 0;
                v |= uint32_t(uint16_t(src[diag_before(SW, SH, d) + (y - y0)])) << (16 * k);
            }
        }
        dst[o] = v;
    }
    if (l == 0) ip->corner = 1;   // from here on: an ordinary corner-layout item
}

__global__ void __launch_bounds__(256)
expand_coefs_kernel(DeriveBatch batch)
{
    const DeriveArgs &a = *batch.a[blockIdx.z];
    if (!a.diagonal_layout) return;
    const uint32_t item = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    switch (blockIdx.y) {
#define EA1_CASE(K) case K: expand_class<K>(a, item, l); break;
        EA1_CASE(0) EA1_CASE(1) EA1_CASE(2) EA1_CASE(3) EA1_CASE(4) EA1_CASE(5) EA1_CASE(6) EA1_CASE(7) EA1_CASE(8) EA1_CASE(9)
        EA1_CASE(10) EA1_CASE(11) EA1_CASE(12) EA1_CASE(13) EA1_CASE(14) EA1_CASE(15) EA1_CASE(16) EA1_CASE(17) EA1_CASE(18)
#undef EA1_CASE
    default: break;
    }
}

void launch_derive(const DeriveBatch &b, uint32_t max_blocks, uint32_t max_tx, int max_units, uint32_t max_items, hipStream_t st)
{
    if (b.n <= 0) return;
    // the block-map kernel also clears the cursors and the zero coefficients: at least one workgroup of it per job
    const uint32_t first = max_blocks > kZeroCoefs / 2 ? max_blocks : kZeroCoefs / 2;
    hipLaunchKernelGGL(derive_block_map_kernel, dim3((first + 255) / 256, b.n), dim3(256), 0, st, b);
    if (max_tx) hipLaunchKernelGGL(derive_tx_map_kernel, dim3((max_tx + 255) / 256, b.n), dim3(256), 0, st, b);
    if (max_blocks) hipLaunchKernelGGL(derive_items_kernel, dim3((max_blocks + 255) / 256, b.n), dim3(256), 0, st, b);
    if (max_items) hipLaunchKernelGGL(expand_coefs_kernel, dim3((max_items + 15) / 16, kTxSizes, b.n), dim3(256), 0, st, b);
    hipLaunchKernelGGL(derive_edges_kernel, dim3((max_units + 255) / 256, b.n), dim3(256), 0, st, b);
}

}  // namespace ea1
