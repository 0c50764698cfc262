// bit reader of the OBU parsers: f(n), su(n), ns(n), uvlc(), leb128 (AV1 specification 4.10)
#pragma once
#include <cstddef>
#include <cstdint>

namespace ea1 {

struct BitReader {
    const uint8_t *p;
    size_t size, pos = 0;   // pos in bits
    bool overrun = false;
    uint32_t f(int n)   // most significant bit first; n <= 32
    {
        uint32_t v = 0;
        for (int i = 0; i < n; i++) {
            if ((pos >> 3) >= size) {
                overrun = true;
                return 0;
            }
            v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1);
            pos++;
        }
        return v;
    }
    int su(int n)   // signed, n bits including the sign
    {
        const int v = int(f(n)), sign_mask = 1 << (n - 1);
        return (v & sign_mask) ? v - 2 * sign_mask : v;
    }
    uint32_t ns(uint32_t n)   // non-symmetric unsigned value below n
    {
        int w = 0;
        for (uint32_t x = n; x; x >>= 1) w++;
        const uint32_t m = (1u << w) - n, v = f(w - 1);
        if (v < m) return v;
        return (v << 1) - m + f(1);
    }
    uint32_t uvlc()
    {
        int zeros = 0;
        while (!overrun && !f(1))
            if (++zeros >= 32) return 0xFFFFFFFFu;
        if (zeros >= 32) return 0xFFFFFFFFu;
        return zeros ? f(zeros) + ((1u << zeros) - 1) : 0;
    }
};

}  // namespace ea1
