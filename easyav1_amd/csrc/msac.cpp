// Symbol decoder: AV1 specification 8.2 (initialisation 8.2.2, boolean / literal 8.2.4-5, symbol
// decoding with CDF adaptation 8.2.6); dav1d src/msac.c.  Table-free part of the bitstream front
// end, see include/easyav1_amd_obu.h.  Host C++ by design (BASELINE.json north_star keeps symbol
// parsing on the CPU).
#include <cstring>

#include "ea1_common.h"
#include "easyav1_amd_obu.h"

namespace {

// f(n) on the tile bytes, most significant bit first; past the end the stream reads as zeros
inline uint32_t take_bits(ea1_msac *s, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        const size_t byte = s->pos_bits >> 3;
        const uint32_t bit = byte < s->size ? (s->data[byte] >> (7 - (s->pos_bits & 7))) & 1u : 0u;
        v = (v << 1) | bit;
        s->pos_bits++;
    }
    return v;
}

inline int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }

// one decision of read_symbol: the partition of the current range by an N-symbol CDF
inline int decode(ea1_msac *s, const uint16_t *cdf, int n)
{
    uint32_t cur = s->range, prev;
    int symbol = -1;
    do {
        symbol++;
        prev = cur;
        const uint32_t f = (1u << 15) - cdf[symbol];
        cur = (((s->range >> 8) * (f >> 6)) >> 1) + 4u * uint32_t(n - symbol - 1);
    } while (s->value < cur);
    s->range = prev - cur;
    s->value -= cur;
    // renormalisation: bring the range back to 16 bits, pulling as many bits as the tile still has
    const int bits = 15 - floor_log2(s->range);
    s->range <<= bits;
    const int64_t avail = s->max_bits > 0 ? s->max_bits : 0;
    const int num_bits = int(bits < avail ? bits : avail);
    const uint32_t new_data = take_bits(s, num_bits);
    const uint32_t padded = new_data << (bits - num_bits);
    s->value = padded ^ (((s->value + 1) << bits) - 1);
    s->max_bits -= bits;
    return symbol;
}

}  // namespace

extern "C" int ea1_msac_init(ea1_msac *s, const uint8_t *data, size_t size, int disable_cdf_update)
{
    if (!s || !data || !size) return ea1::fail(EA1_EINVAL, "ea1_msac_init: bad arguments");
    memset(s, 0, sizeof *s);
    s->data = data;
    s->size = size;
    s->disable_update = disable_cdf_update != 0;
    const int num_bits = size * 8 < 15 ? int(size * 8) : 15;
    const uint32_t buf = take_bits(s, num_bits);
    s->value = ((1u << 15) - 1) ^ (buf << (15 - num_bits));
    s->range = 1u << 15;
    s->max_bits = int64_t(size) * 8 - 15;
    return EA1_OK;
}

extern "C" int ea1_msac_symbol(ea1_msac *s, uint16_t *cdf, int n)
{
    const int symbol = decode(s, cdf, n);
    if (!s->disable_update) {
        // 8.2.6: move every cumulative value towards the decoded symbol's step function
        const int rate = 3 + (cdf[n] > 15) + (cdf[n] > 31) + (floor_log2(uint32_t(n)) < 2 ? floor_log2(uint32_t(n)) : 2);
        uint32_t tmp = 0;
        for (int i = 0; i < n - 1; i++) {
            if (i == symbol) tmp = 1u << 15;
            if (tmp < cdf[i]) cdf[i] = uint16_t(cdf[i] - ((cdf[i] - tmp) >> rate));
            else cdf[i] = uint16_t(cdf[i] + ((tmp - cdf[i]) >> rate));
        }
        cdf[n] = uint16_t(cdf[n] + (cdf[n] < 32));
    }
    return symbol;
}

extern "C" int ea1_msac_bool(ea1_msac *s)
{
    static const uint16_t half[3] = {1u << 14, 1u << 15, 0};
    return decode(s, half, 2);
}

extern "C" uint32_t ea1_msac_literal(ea1_msac *s, int nbits)
{
    uint32_t x = 0;
    for (int i = 0; i < nbits; i++) x = 2 * x + uint32_t(ea1_msac_bool(s));
    return x;
}

extern "C" int64_t ea1_msac_bits_left(const ea1_msac *s) { return s ? s->max_bits : 0; }
