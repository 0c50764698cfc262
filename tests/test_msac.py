"""The symbol decoder (AV1 specification 8.2) against an independent ENCODER written for this test in
exact integer arithmetic (Python big integers: no renormalisation buffer, no carry propagation to get
wrong): random alphabets of 2..16 symbols with random CDFs, adaptation on and off, booleans and
literals mixed in.  Parity with dav1d's msac.c is UNPINNED (dav1d absent); the engine is table-free."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def obu():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import obu
    return obu


class Encoder:
    """Arithmetic encoder matching the decoder's partition rule.  The decoder sees the complement of the
    bit string; in bit-string space symbol s of an N-symbol CDF owns [R - prev_s, R - cur_s) of the current
    range R, with cur_k = ((R >> 8) * ((32768 - cdf[k]) >> 6) >> 1) + 4 (N - k - 1) and prev_s = cur_{s-1} (R for
    s = 0)."""

    def __init__(self):
        self.low, self.rng, self.shift = 0, 1 << 15, 0

    def encode(self, s, cdf, n):
        cur = lambda k: (((self.rng >> 8) * ((32768 - int(cdf[k])) >> 6)) >> 1) + 4 * (n - k - 1)
        prev = self.rng if s == 0 else cur(s - 1)
        c = cur(s)
        self.low += self.rng - prev
        self.rng = prev - c
        assert self.rng > 0
        d = 15 - (self.rng.bit_length() - 1)
        self.low <<= d
        self.rng <<= d
        self.shift += d

    def bool(self, b):
        self.encode(b, [1 << 14, 1 << 15], 2)

    def literal(self, v, n):
        for i in range(n - 1, -1, -1):
            self.bool((v >> i) & 1)

    def finish(self):
        """shortest bit string (zero padded by the decoder) inside [low, low + rng)"""
        width = 15 + self.shift
        t = width
        while t > 0:
            b = -(-self.low >> t) << t      # ceil to a multiple of 2^t
            if b < self.low + self.rng:
                break
            t -= 1
        b = -(-self.low >> t) << t
        assert self.low <= b < self.low + self.rng and b < (1 << width)
        nbits = max(width - t, 1)
        value = b >> t
        nbytes = (nbits + 7) // 8
        return (value << (nbytes * 8 - nbits)).to_bytes(nbytes, "big")


def adapt(cdf, n, symbol):
    rate = 3 + (cdf[n] > 15) + (cdf[n] > 31) + min(n.bit_length() - 1, 2)
    tmp = 0
    for i in range(n - 1):
        if i == symbol:
            tmp = 1 << 15
        if tmp < cdf[i]:
            cdf[i] -= (cdf[i] - tmp) >> rate
        else:
            cdf[i] += (tmp - cdf[i]) >> rate
    cdf[n] += cdf[n] < 32


def random_cdf(rng, n):
    """N increasing cumulative values ending at 32768, steps of at least 128 (every symbol codable), + counter"""
    cuts = np.sort(rng.choice(np.arange(1, 32768 // 128), n - 1, replace=False)) * 128
    return [int(c) for c in cuts] + [32768, 0]


@pytest.mark.parametrize("disable_update", [False, True])
def test_round_trip_of_random_symbol_streams(obu, disable_update):
    rng = np.random.default_rng(7 + disable_update)
    for trial in range(60):
        nctx = int(rng.integers(1, 6))
        sizes = [int(rng.integers(2, 17)) for _ in range(nctx)]
        enc_cdfs = [random_cdf(rng, n) for n in sizes]
        dec_cdfs = [np.array(c, np.uint16) for c in enc_cdfs]
        enc = Encoder()
        script = []
        for _ in range(int(rng.integers(1, 400))):
            kind = rng.integers(10)
            if kind == 0:
                b = int(rng.integers(2))
                enc.bool(b)
                script.append(("bool", b))
            elif kind == 1:
                n = int(rng.integers(1, 17))
                v = int(rng.integers(1 << n))
                enc.literal(v, n)
                script.append(("lit", n, v))
            else:
                c = int(rng.integers(nctx))
                n = sizes[c]
                cdf = enc_cdfs[c]
                # draw the symbol from the CDF itself most of the time (adaptation then sharpens it), sometimes uniformly
                if rng.integers(4):
                    s = int(np.searchsorted(np.array(cdf[:n]), rng.integers(32768), side="right"))
                else:
                    s = int(rng.integers(n))
                enc.encode(s, cdf, n)
                if not disable_update:
                    adapt(cdf, n, s)
                script.append(("sym", c, s))
        data = enc.finish()
        dec = obu.SymbolDecoder(data, disable_update)
        for step in script:
            if step[0] == "bool":
                assert dec.bool() == step[1]
            elif step[0] == "lit":
                assert dec.literal(step[1]) == step[2]
            else:
                assert dec.symbol(dec_cdfs[step[1]]) == step[2], (trial, step)
        for c in range(nctx):   # the adapted contexts agree with the encoder's
            assert dec_cdfs[c].tolist() == enc_cdfs[c]
        assert dec.bits_left() <= len(data) * 8 - 15


def test_adaptation_moves_towards_the_symbols(obu):
    """a uniform 4-symbol CDF fed the same symbol converges on it; the counter saturates at 32 and the rate
    slows down 4 -> 5 -> 6 (3 + min(log2 N, 2) = 5 for N = 4, + 1 after 16, + 1 after 32 symbols)"""
    enc = Encoder()
    cdf = [8192, 16384, 24576, 32768, 0]
    ref = list(cdf)
    for _ in range(200):
        enc.encode(2, ref, 4)
        adapt(ref, 4, 2)
    d = obu.SymbolDecoder(enc.finish())
    arr = np.array(cdf, np.uint16)
    steps = []
    for _ in range(200):
        before = int(arr[1])
        assert d.symbol(arr) == 2
        steps.append(before - int(arr[1]))
    assert arr[4] == 32 and arr.tolist() == ref
    assert arr[1] < 16384 // 4 and arr[2] > 32768 - 16384 // 4   # cdf[1] -> 0, cdf[2] -> 32768: symbol 2 takes the range
    assert all(a >= b for a, b in zip(steps, steps[1:])) or steps[16] < steps[15]   # the steps shrink, with a drop when the rate changes
    assert steps[0] == 16384 >> 5 and steps[17] == int(ref_step(17)) and steps[40] <= steps[17]


def ref_step(k):
    cdf = [8192, 16384, 24576, 32768, 0]
    for _ in range(k):
        adapt(cdf, 4, 2)
    before = cdf[1]
    adapt(cdf, 4, 2)
    return before - cdf[1]


def test_short_and_exhausted_buffers(obu):
    """a one-byte tile, and decoding far past the end (zero padding): defined results, no reads out of bounds"""
    d = obu.SymbolDecoder(b"\\x80")
    vals = [d.bool() for _ in range(64)]
    assert set(vals) <= {0, 1} and d.bits_left() < 0
    d = obu.SymbolDecoder(b"\\xff\\xff\\xff")
    cdf = np.array([4096, 8192, 32768, 0], np.uint16)
    for _ in range(1000):
        assert 0 <= d.symbol(cdf) < 3
