// Tile-column split of one frame over several GPUs (BASELINE configs[3]): the
// device side of the two exchanges.  A rank reconstructs the blocks of its own strip of tile
// columns; before the in-loop filters it needs 16 luma (8 chroma) columns of its neighbours'
// reconstruction next to its strip, after them every rank needs everybody's finished strip for
// the next frame's inter prediction.  Both exchanges move COLUMN RANGES of pitched pictures, which
// the collective (RCCL all-gather, driven by the host side through torch.distributed) wants as
// contiguous buffers: this kernel packs / unpacks them.  HBM bound, trivially: 2 x bytes moved.
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

size_t columns_bytes(int w, int h) { return size_t(w) * h * 3 / 2; }

void launch_copy_columns(const DevPic &pic, int x, int w, uint8_t *buf, bool to_picture, hipStream_t st)
{
    const int items = pic.h * (w / 4);   // luma has the most
    hipLaunchKernelGGL(copy_columns_kernel, dim3((items + 255) / 256, 1, 3), dim3(256), 0, st, pic, x, w, buf, to_picture ? 1 : 0);
}

}  // namespace ea1
