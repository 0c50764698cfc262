// Hand-over of a parsed frame: what the in-loop filters need to know about the frame, derived ON THE
// GPU from the block and transform-block records that were uploaded anyway (the GPU analogue of
// dav1d's lf_mask.c, /root/reference/cmake/dav1d.cmake:980):
//   block map        index of the block covering each 4x4 luma unit
//   transform maps   index of the transform block covering each 4x4-sample unit, per plane
//   luma_edge / chroma_edge   deblocking edge class of every unit's left and top edge (AV1 spec 7.14.2/3:
//                    an edge exists where a transform block starts, except inside a skipped inter block;
//                    the filter length follows the smaller of the two transform sizes across the edge)
//   lvl_y / lvl_c    deblocking levels of the unit's block
//   unit_flags       per 8x8: every block in it is skipped (CDEF leaves it alone) / its chroma owner is inter
// All of it is data-parallel over blocks, transform blocks and units; on the host it cost a quarter of
// ea1_job_create.  The records have been validated by the host before these kernels are enqueued
// (positions inside the frame, blocks tile the frame, transform blocks tile their block), so the
// scatters stay inside the maps.  Runs on the context's upload stream, behind the copy of the records.
#include "av1_dev.h"
#include "ea1_common.h"

namespace ea1 {

__global__ void __launch_bounds__(256)
derive_block_map_kernel(DeriveArgs a)
{
    const uint32_t bi = blockIdx.x * 256 + threadIdx.x;
    if (bi >= a.num_blocks) return;
    const ea1_block b = a.blocks[bi];
    const int n4w = 1 << (b.bwl - 2), n4h = 1 << (b.bhl - 2);
    uint32_t *row = a.map + size_t(b.y4) * a.w4 + b.x4;
    for (int j = 0; j < n4h; j++, row += a.w4)
        for (int i = 0; i < n4w; i++) row[i] = bi;
}

__global__ void __launch_bounds__(256)
derive_tx_map_kernel(DeriveArgs a)
{
    const uint32_t ti = blockIdx.x * 256 + threadIdx.x;
    if (ti >= a.num_tx) return;
    const ea1_txblock t = a.txs[ti];
    if (t.plane > 1) return;   // U and V share one transform grid (checked by the host): only U's map is made
    const int n4w = 1 << ((t.size & 15) - 2), n4h = 1 << ((t.size >> 4) - 2);
    const int mw = t.plane ? a.w8 : a.w4;
    uint32_t *row = a.txmap[t.plane] + size_t(t.y4) * mw + t.x4;
    for (int j = 0; j < n4h; j++, row += mw)
        for (int i = 0; i < n4w; i++) row[i] = ti;
}

// 4:2:0 geometry of a block in a plane (see include/easyav1_amd_video.h): chroma of a block with a 4-sample
// side starts at the 8-aligned luma position and covers at least 4x4 chroma samples
__device__ __forceinline__ void block_origin(const ea1_block &b, int pl, int &x, int &y)
{
    if (pl == 0) {
        x = b.x4 * 4;
        y = b.y4 * 4;
    } else {
        x = (b.x4 & ~1) * 2;
        y = (b.y4 & ~1) * 2;
    }
}

// one lane per unit: luma units first (w4 * h4), then chroma units (w8 * h8, which also carry unit_flags and lvl_c)
__global__ void __launch_bounds__(256)
derive_edges_kernel(DeriveArgs a)
{
    const int nl = a.w4 * a.h4, nc = a.w8 * a.h8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nl + nc) return;
    const int pl = i >= nl ? 1 : 0;
    const int u = pl ? i - nl : i;
    const int mw = pl ? a.w8 : a.w4;
    const int uy = u / mw, ux = u - uy * mw;
    const uint32_t *txm = a.txmap[pl];
    const ea1_txblock t = a.txs[txm[u]];
    // chroma units belong to the block that owns the chroma of the 8x8 luma area
    const ea1_block b = a.blocks[pl ? a.map[size_t(2 * uy + 1) * a.w4 + 2 * ux + 1] : a.map[size_t(uy) * a.w4 + ux]];
    const bool inner_off = (b.flags & EA1_BLK_SKIP) && (b.flags & EA1_BLK_INTER);
    int bx, by;
    block_origin(b, pl, bx, by);
    uint32_t e = 0;
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {   // 0: left edge (vertical), 1: top edge
        const int pos = dir ? uy : ux;
        if (pos == 0 || (dir ? t.y4 : t.x4) != pos) continue;
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
    } else {
        a.chroma_edge[u] = uint8_t(e);
        a.lvl_c[u] = per_block ? uint16_t(b.lf_level[2] | (b.lf_level[3] << 8)) : uint16_t(a.lf_level[2] | (a.lf_level[3] << 8));
        bool all_skip = true;
#pragma unroll
        for (int k = 0; k < 4; k++) all_skip &= (a.blocks[a.map[size_t(2 * uy + (k >> 1)) * a.w4 + 2 * ux + (k & 1)]].flags & EA1_BLK_SKIP) != 0;
        a.unit_flags[u] = uint8_t((all_skip ? 1 : 0) | ((b.flags & EA1_BLK_INTER) ? 2 : 0));
    }
}

void launch_derive(const DeriveArgs &a, hipStream_t st)
{
    if (a.num_blocks) hipLaunchKernelGGL(derive_block_map_kernel, dim3((a.num_blocks + 255) / 256), dim3(256), 0, st, a);
    if (a.num_tx) hipLaunchKernelGGL(derive_tx_map_kernel, dim3((a.num_tx + 255) / 256), dim3(256), 0, st, a);
    const int units = a.w4 * a.h4 + a.w8 * a.h8;
    hipLaunchKernelGGL(derive_edges_kernel, dim3((units + 255) / 256), dim3(256), 0, st, a);
}

}  // namespace ea1
