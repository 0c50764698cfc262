// Wedge masks for masked compound and inter-intra prediction (AV1 spec 7.11.3.11; dav1d
// src/wedge.c, /root/reference/cmake/dav1d.cmake:995).  Built once per decoder context on the host
// and uploaded: 9 block sizes (8/16/32 squared) x 2 signs x 16 wedges, one byte per luma sample.
#include <cstring>
#include <vector>

#include "av1_dev.h"
#include "ea1_av1_tables.inc"

namespace ea1 {

namespace {

enum { kHor = 0, kVer = 1, kO27 = 2, kO63 = 3, kO117 = 4, kO153 = 5 };

struct Master {
    uint8_t m[6][64][64];
    Master()
    {
        static const uint8_t odd[64] = {EA1_WEDGE_MASTER_ODD_VALUES};
        static const uint8_t even[64] = {EA1_WEDGE_MASTER_EVEN_VALUES};
        static const uint8_t vert[64] = {EA1_WEDGE_MASTER_VERTICAL_VALUES};
        auto clip = [](int v) { return v < 0 ? 0 : (v > 63 ? 63 : v); };
        // oblique 63: the profile slides one sample to the left every two rows
        for (int j = 0; j < 64; j++) {
            int shift = 16;
            for (int i = 0; i < 64; i += 2) {
                m[kO63][i][j] = even[clip(j - shift)];
                shift--;
                m[kO63][i + 1][j] = odd[clip(j - shift)];
                m[kVer][i][j] = m[kVer][i + 1][j] = vert[j];
            }
        }
        // the other directions are transposes / mirrors of the two above
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 64; j++) {
                const uint8_t v = m[kO63][i][j];
                m[kO27][j][i] = v;
                m[kO117][i][63 - j] = uint8_t(64 - v);
                m[kO153][63 - j][i] = uint8_t(64 - v);
                m[kHor][j][i] = m[kVer][i][j];
            }
    }
};

}  // namespace

uint32_t wedge_mask_offset(int wl2, int hl2, int sign, int index)
{
    // sizes in (wl2, hl2) order 8x8, 8x16, 8x32, 16x8, ...; every size holds 32 masks of w * h bytes
    uint32_t off = 0;
    for (int a = 3; a <= 5; a++)
        for (int b = 3; b <= 5; b++) {
            if (a == wl2 && b == hl2) return off + uint32_t((sign * 16 + index) << (wl2 + hl2));
            off += 32u << (a + b);
        }
    return 0;
}

uint32_t wedge_table_bytes() { return wedge_mask_offset(5, 5, 0, 0) + (32u << 10); }

void build_wedge_table(uint8_t *out)
{
    static const Master master;
    static const uint8_t codebook[3][16][3] = {EA1_WEDGE_CODEBOOK_VALUES};
    for (int wl2 = 3; wl2 <= 5; wl2++)
        for (int hl2 = 3; hl2 <= 5; hl2++) {
            const int w = 1 << wl2, h = 1 << hl2;
            for (int idx = 0; idx < 16; idx++) {
                const uint8_t *cb = codebook[h > w ? 0 : (h < w ? 1 : 2)][idx];
                const int dir = cb[0], xo = 32 - ((cb[1] * w) >> 3), yo = 32 - ((cb[2] * h) >> 3);
                // sign convention: the mask whose first row + first column average below 32 is stored flipped
                int sum = 0;
                for (int i = 0; i < w; i++) sum += master.m[dir][yo][xo + i];
                for (int i = 1; i < h; i++) sum += master.m[dir][yo + i][xo];
                const int flip = (sum + (w + h - 1) / 2) / (w + h - 1) < 32;
                for (int sign = 0; sign < 2; sign++) {
                    uint8_t *dst = out + wedge_mask_offset(wl2, hl2, sign, idx);
                    for (int i = 0; i < h; i++)
                        for (int j = 0; j < w; j++) {
                            const int v = master.m[dir][yo + i][xo + j];
                            dst[i * w + j] = uint8_t(flip == sign ? v : 64 - v);
                        }
                }
            }
        }
}

}  // namespace ea1
