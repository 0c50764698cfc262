// film grain: shared between the context code (av1_recon.cpp) and the kernel file (film_grain.hip)
#pragma once
#include "av1_dev.h"
#include "easyav1_amd_grain.h"

namespace ea1 {

// layout of one chain's blob (bytes): templates as int8, three look-up tables, one random byte per
// 32x32 luma block ([stripe][block])
constexpr size_t kGrainLuma = 0;                 // 73 x 82
constexpr size_t kGrainCb = 5988;                // 38 x 44 (4:2:0), 73 x 44 (4:2:2), 73 x 82 (4:4:4)
constexpr size_t kGrainCr = kGrainCb + 5988;
constexpr size_t kGrainLut = kGrainCr + 5988;    // 3 x 256
constexpr size_t kGrainOffsets = kGrainLut + 768;

enum : uint8_t { kGrainY = 1, kGrainCbOn = 2, kGrainCrOn = 4, kGrainOverlap = 8, kGrainFromLuma = 16 };

// one picture to grain: lives in device memory next to the blobs, one per chain
struct GrainView {
    const uint8_t *src[3];   // src[0] == nullptr: chain idle
    uint8_t *dst[3];
    const uint8_t *blob;
    int stride[3];
    int w, h, nbx;           // VISIBLE luma size (the grain is synthesised for it), 32x32 blocks per stripe
    int cw, ch;              // coded luma size (whole 8x8 units): samples beyond the visible part are copied
    int16_t uv_mult[2], uv_luma_mult[2], uv_offset[2];   // already minus 128 / 128 / 256
    uint8_t flags, scaling_shift, min_value, max_luma, max_chroma, ssx, ssy, pad[1];   // ssx, ssy: chroma subsampling of the pictures
};

int grain_validate(const ea1_film_grain &fg);
size_t grain_blob_bytes(int w, int h);
void grain_prepare(const ea1_film_grain &fg, const int16_t *gauss, int w, int h, int ssx, int ssy, uint8_t *blob);
void launch_film_grain(const GrainView *d_views, int nviews, int w, int h, int ssx, int ssy, hipStream_t st, Prof *prof, uint64_t frames);

}  // namespace ea1
