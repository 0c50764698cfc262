// Tile-column split of one frame over several GPUs (BASELINE configs[3]): the
// device side of the two exchanges.  A rank reconstructs the blocks of its own strip of tile
// columns; before the in-loop filters it needs 16 luma (8 chroma) columns of its neighbours'
// reconstruction next to its strip, after them every rank needs everybody's finished strip for
// the next frame's inter prediction.  Both exchanges move COLUMN RANGES of pitched pictures, which
// the collective (RCCL all-gather, driven by the host side through torch.distributed) wants as
// contiguous buffers: this kernel packs / unpacks them.  HBM bound, trivially: 2 x bytes moved.
#include <cstdlib>

#include "av1_dev.h"
#include "ea1_common.h"

namespace ea1 {

// buffer layout: Y rows of `w` bytes, then U rows of w/2, then V rows; rows = plane height.
// A lane moves 4 bytes (w is a multiple of 8, x a multiple of 8: dword-aligned in every plane).
__global__ void __launch_bounds__(256)
copy_columns_kernel(DevPic pic, int x, int w, uint8_t *buf, int to_picture)
{
    const int pl = blockIdx.z;
    const int ss = pl ? 1 : 0;
    const int px = x >> ss, pw_ = w >> ss, rows = pic.h >> ss;
    const int dwords = pw_ >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * dwords) return;
    const int r = i / dwords, c = i - r * dwords;
    const size_t plane_off = pl == 0 ? 0 : size_t(w) * pic.h + (pl == 2 ? size_t(w >> 1) * (pic.h >> 1) : 0);
    uint32_t *b = reinterpret_cast<uint32_t *>(buf + plane_off + size_t(r) * pw_) + c;
    uint32_t *p = reinterpret_cast<uint32_t *>(pic.p[pl] + size_t(r) * pic.stride[pl] + px) + c;
    if (to_picture) *p = *b;
    else *b = *p;
}

// ---- hand-over: pinned host memory pulled into device memory by a kernel ----------------------------------
// One launch moves up to kPullSegs separate buffers (the staged records and the coefficient array of several
// frames).  Measured on this box (tools/zero_copy_probe.hip, profiles/r02_g_zero_copy_probe.txt): 55 GB/s whatever
// the buffer size, against 27 GB/s (0.5 MB buffers) and 42 GB/s (1.6 MB) for back-to-back hipMemcpyAsync calls,
// whose fixed cost per copy is what small frames pay.  A workgroup moves 64 KB chunks, 16 bytes per lane and
// four loads in flight per lane; a fixed, small grid strides over the chunks so that the copy takes few wave
// slots away from the reconstruction kernels running beside it.
__global__ void __launch_bounds__(256)
pull_kernel(PullArgs a)
{
    constexpr uint32_t kChunk = 4096;   // 16-byte units per workgroup step
    for (uint32_t blk = blockIdx.x; blk < a.total_chunks; blk += gridDim.x) {
        int s = 0;
        while (s + 1 < a.n && a.seg[s + 1].first_chunk <= blk) s++;
        const PullSeg sg = a.seg[s];
        const uint32_t n16 = uint32_t(sg.bytes >> 4);
        const uint32_t begin = (blk - sg.first_chunk) * kChunk;
        const uint32_t end = begin + kChunk < n16 ? begin + kChunk : n16;
        const uint4 *src = static_cast<const uint4 *>(sg.src);
        uint4 *dst = static_cast<uint4 *>(sg.dst);
        for (uint32_t i = begin + threadIdx.x; i < end; i += 1024) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + 256 * k < end) v[k] = src[i + 256 * k];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + 256 * k < end) dst[i + 256 * k] = v[k];
        }
        // a tail of less than 16 bytes (coefficient arrays: an even number of bytes), by the segment's last chunk
        const uint32_t tail = uint32_t(sg.bytes & 15);
        if (tail && end == n16 && threadIdx.x < tail / 2)
            reinterpret_cast<uint16_t *>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint16_t *>(src + n16)[threadIdx.x];
    }
}

void launch_pull(PullArgs &a, hipStream_t st)
{
    if (a.n <= 0) return;
    uint32_t chunks = 0;
    for (int k = 0; k < a.n; k++) {
        a.seg[k].first_chunk = chunks;
        chunks += uint32_t(((a.seg[k].bytes >> 4) + 4095) / 4096);
        if (!(a.seg[k].bytes >> 4)) chunks += 1;   // a segment that is all tail still needs its chunk
    }
    a.total_chunks = chunks;
    static const uint32_t max_grid = getenv("EA1_PULL_GRID") ? uint32_t(atoi(getenv("EA1_PULL_GRID"))) : 192u;
    const uint32_t grid = chunks < max_grid ? chunks : max_grid;
    hipLaunchKernelGGL(pull_kernel, dim3(grid), dim3(256), 0, st, a);
}

size_t columns_bytes(int w, int h) { return size_t(w) * h * 3 / 2; }

void launch_copy_columns(const DevPic &pic, int x, int w, uint8_t *buf, bool to_picture, hipStream_t st)
{
    const int items = pic.h * (w / 4);   // luma has the most
    hipLaunchKernelGGL(copy_columns_kernel, dim3((items + 255) / 256, 1, 3), dim3(256), 0, st, pic, x, w, buf, to_picture ? 1 : 0);
}

}  // namespace ea1
