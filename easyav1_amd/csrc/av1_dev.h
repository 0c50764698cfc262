// Internal device-side view of one parsed frame and of the decoder context, shared by the AV1
// kernels (av1_*.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "easyav1_amd_video.h"

namespace ea1 {

// One picture in HBM: three planes in one allocation, row stride a multiple of 128 bytes so that
// every row start is aligned for 16-byte vector access.
struct DevPic {
    uint8_t *p[3];
    int stride[3];
    int w, h;  // luma size
    __host__ __device__ int pw(int pl) const { return pl ? w >> 1 : w; }
    __host__ __device__ int ph(int pl) const { return pl ? h >> 1 : h; }
};

// one (block, plane) transform to run; built on the host at job creation
struct TxItem {
    uint32_t blk;       // block index
    uint32_t coef_off;  // int16 index of this plane's coefficients
    uint16_t x, y;      // position in the plane, pixels
    uint8_t plane;
    uint8_t txtp;
    uint8_t pad[2];
};
static_assert(sizeof(TxItem) == 16, "TxItem layout");

// per 8x8-luma-unit deblocking edge classes (GPU analogue of dav1d's lf_mask.c output)
//   bits 0-1 luma left edge   0 none, 1 -> 8-tap class, 2 -> 16-tap class
//   bits 2-3 luma top edge
//   bits 4-5 chroma left edge 0 none, 1 -> 4-tap, 2 -> 6-tap
//   bits 6-7 chroma top edge
// plus bit flags needed by CDEF in a second byte map
struct JobDev {
    ea1_frame_desc desc;
    const ea1_block *blocks;
    const int16_t *coefs;
    const uint32_t *blk_map;   // [h8][w8] block index per 8x8 luma unit
    const uint8_t *edge_info;  // [h8][w8]
    const uint8_t *unit_flags; // [h8][w8] bit0 = block has skip flag, bit1 = inter
    const uint8_t *cdef_idx;   // [sb_rows][sb_cols]
    const ea1_lr_unit *lr[3];
    int lr_cols[3], lr_rows[3];
    // transform work lists of the inter blocks, one per square size class log2 n = 2..6
    const TxItem *tx_items[5];
    int tx_count[5];
    // intra wavefront: items (all planes) sorted by level
    const TxItem *intra_items;
    const uint32_t *level_start;  // host copy is kept in the job, this is informational
    int num_levels;
    int w8, h8, sb_cols, sb_rows;
    // algorithmic-byte accounting (host side only; SURVEY.md 8(d) per-kernel figures)
    uint64_t tx_coefs[5], tx_pixels[5];   // inter transform lists: coefficient positions / pixels touched
    uint64_t intra_coefs, intra_pixels;   // intra items
    uint64_t inter_luma_px, ref_luma_px;  // predicted luma pixels / luma reference pixels fetched
};

// Optional per-kernel timing with HIP events on the launch stream (off by default; bench.py's
// roofline figure).  Every launcher brackets its kernels with begin()/end().
struct Prof {
    virtual void begin(const char *kernel, uint64_t algorithmic_bytes, hipStream_t st) = 0;
    virtual void end(hipStream_t st) = 0;
    virtual ~Prof() = default;
};
struct ProfScope {
    Prof *p;
    hipStream_t st;
    ProfScope(Prof *p_, const char *kernel, uint64_t bytes, hipStream_t st_) : p(p_), st(st_)
    {
        if (p) p->begin(kernel, bytes, st);
    }
    ~ProfScope()
    {
        if (p) p->end(st);
    }
};

// kernel launchers (one per translation unit); all asynchronous on `st`
void launch_inter_pred(const JobDev &job, const DevPic *refs /* device array[8] */, const DevPic &cur, hipStream_t st, Prof *prof);
void launch_itx_inter(const JobDev &job, const DevPic &cur, hipStream_t st, Prof *prof);
void launch_intra_level(const JobDev &job, const DevPic &cur, int first_item, int count, hipStream_t st);
void launch_deblock(const JobDev &job, const DevPic &cur, hipStream_t st, Prof *prof);
void launch_cdef(const JobDev &job, const DevPic &in, const DevPic &out, uint8_t *dir_var /* 2 bytes per unit */,
                 hipStream_t st, Prof *prof);
void launch_lr(const JobDev &job, const DevPic &deblocked, const DevPic &cdef, const DevPic &out, hipStream_t st, Prof *prof);
int launch_dsp_itx_add(uint8_t *plane, size_t stride, int w, int h, const int16_t *d_coefs, const uint32_t *d_pos,
                       const uint8_t *d_txtp, int count, hipStream_t st);

}  // namespace ea1
