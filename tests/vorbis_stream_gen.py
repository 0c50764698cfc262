"""Random but VALID Vorbis I setups for differential tests of the packet front end against the reference
decoder (there is no Vorbis encoder in this environment).

Every codebook gets a complete prefix tree, so ANY bit string is a decodable sequence of codewords:
audio packets can then simply be random bytes behind a valid (type, mode) prefix.  Both decoders
read the same bits; the streams exercise codebook forms (ordered / sparse, lookup types 0 / 1 / 2,
sequence flag), floor 1 layouts, residue types 0 / 1 / 2 with cascades, channel coupling, submaps,
1..6 channels and every end-of-packet position."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, value, n):
        for i in range(n):
            self.bits.append((int(value) >> i) & 1)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(sum(b[i + k] << k for k in range(8)) for i in range(0, len(b), 8))


def ilog(v):
    r = 0
    while v:
        r += 1
        v >>= 1
    return r


def tree_lengths(rng, n, max_len=14):
    """codeword lengths of a random COMPLETE binary tree with n leaves"""
    leaves = [0]
    while len(leaves) < n:
        cand = [i for i, d in enumerate(leaves) if d < max_len]
        i = cand[rng.integers(len(cand))]
        d = leaves.pop(i)
        leaves += [d + 1, d + 1]
    rng.shuffle(leaves)
    return leaves


def pack_float(mant, exp, neg=False):
    """Vorbis packed float: value = mant * 2^exp"""
    return (0x80000000 if neg else 0) | ((exp + 788) << 21) | mant


def write_book(w, rng, dim, used_entries, maptype, ordered=False, sparse=False):
    lens = tree_lengths(rng, used_entries)
    if ordered:
        lens = sorted(lens)
    elif sparse:   # sprinkle unused entries
        out = []
        for L in lens:
            while rng.random() < 0.2:
                out.append(0)
            out.append(L)
        lens = out
    entries = len(lens)
    if maptype == 1:   # entries must reach quantvals ** dim distinct vectors: keep it well formed
        pass
    w.write(0x564342, 24)
    w.write(dim, 16)
    w.write(entries, 24)
    if ordered:
        w.write(1, 1)
        w.write(lens[0] - 1, 5)
        i, length = 0, lens[0]
        while i < entries:
            num = sum(1 for L in lens[i:] if L == length)
            w.write(num, ilog(entries - i))
            i += num
            length += 1
    else:
        w.write(0, 1)
        w.write(1 if sparse else 0, 1)
        for L in lens:
            if sparse:
                w.write(1 if L else 0, 1)
                if not L:
                    continue
            w.write(L - 1, 5)
    w.write(maptype, 4)
    if maptype:
        w.write(pack_float(int(rng.integers(1, 64)), int(rng.integers(-8, -2)), bool(rng.integers(2))), 32)   # minimum
        w.write(pack_float(int(rng.integers(1, 32)), int(rng.integers(-7, -3))), 32)                        # delta
        qbits = int(rng.integers(2, 6))
        w.write(qbits - 1, 4)
        w.write(int(rng.integers(2)), 1)   # sequence flag
        if maptype == 1:
            vals = 1
            while (vals + 1) ** dim <= entries:
                vals += 1
            nq = vals
        else:
            nq = entries * dim
        for _ in range(nq):
            w.write(int(rng.integers(1 << qbits)), qbits)
    return entries


def make_stream(seed, channels=2, bs=(256, 2048)):
    """-> (id_header, comment_header, setup_header, nmodes, mode_is_long[])"""
    rng = np.random.default_rng(seed)
    idw = BitWriter()
    idw.write(0, 32)
    idw.write(channels, 8)
    idw.write(48000, 32)
    for _ in range(3):
        idw.write(0, 32)
    idw.write(ilog(bs[0]) - 1, 4)
    idw.write(ilog(bs[1]) - 1, 4)
    idw.write(1, 1)
    id_header = b"\x01vorbis" + idw.bytes()
    cw = BitWriter()
    vendor = b"easyav1-amd tests"
    cw.write(len(vendor), 32)
    for c in vendor:
        cw.write(c, 8)
    cw.write(0, 32)
    cw.write(1, 1)
    comment_header = b"\x03vorbis" + cw.bytes()

    w = BitWriter()
    books = []   # (kind, dim, entries): 'scalar' (lookup 0) or 'vq'

    def add_book(kind, dim, used, **kw):
        maptype = 0 if kind == "scalar" else int(rng.integers(1, 3))
        books.append([kind, dim, 0])
        return len(books) - 1, (dim, used, maptype, kw)

    plan = []
    # floor books: scalar value books + master (class) books
    value_books = []
    for _ in range(4):
        idx, spec = add_book("scalar", 1, int(rng.integers(8, 80)), ordered=bool(rng.integers(2)), sparse=bool(rng.integers(2)))
        plan.append(spec)
        value_books.append(idx)
    master_books = []
    for _ in range(2):
        idx, spec = add_book("scalar", int(rng.integers(1, 4)), int(rng.integers(4, 40)))
        plan.append(spec)
        master_books.append(idx)
    # residue books
    res_specs = []
    for which in range(2):
        grouping = int(rng.choice([4, 8, 16, 32]))
        parts = int(rng.integers(2, 6))
        pdim = int(rng.integers(1, 4))
        phrase, spec = add_book("scalar", pdim, parts ** pdim + int(rng.integers(0, 3)), sparse=bool(rng.integers(2)))
        plan.append(spec)
        cascade, stage_books = [], []
        for _ in range(parts):
            c = int(rng.integers(0, 8)) | (int(rng.integers(0, 2)) << 3)
            cascade.append(c)
            row = []
            for k in range(8):
                if c & (1 << k):
                    dim = int(rng.choice([d for d in (1, 2, 4) if grouping % d == 0]))
                    idx, sp = add_book("vq", dim, int(rng.integers(3, 60)), ordered=bool(rng.integers(2)))
                    plan.append(sp)
                    row.append(idx)
            stage_books.append(row)
        res_specs.append(dict(type=int(rng.integers(0, 3)), grouping=grouping, parts=parts, phrase=phrase, cascade=cascade,
                              stage_books=stage_books))
    w.write(len(plan) - 1, 8)
    for (dim, used, maptype, kw) in plan:
        write_book(w, rng, dim, used, maptype, **kw)
    w.write(0, 6)
    w.write(0, 16)
    # two floors (one per block size)
    w.write(1, 6)
    for which in range(2):
        w.write(1, 16)
        nparts = int(rng.integers(1, 5))
        nclass = int(rng.integers(1, 4))
        pclass = [int(rng.integers(nclass)) for _ in range(nparts)]
        pclass[0] = nclass - 1   # the highest class must occur (it defines how many classes are read)
        w.write(nparts, 5)
        for c in pclass:
            w.write(c, 4)
        dims = []
        for c in range(nclass):
            dim = int(rng.integers(1, 6))
            subs = int(rng.integers(0, 3))
            dims.append(dim)
            w.write(dim - 1, 3)
            w.write(subs, 2)
            if subs:
                w.write(master_books[int(rng.integers(len(master_books)))], 8)
            for _ in range(1 << subs):
                w.write(0 if rng.random() < 0.2 else value_books[int(rng.integers(len(value_books)))] + 1, 8)
        w.write(int(rng.integers(0, 4)), 2)    # multiplier - 1
        rangebits = ilog(bs[which] // 2) - 1
        w.write(rangebits, 4)
        count = sum(dims[c] for c in pclass)
        xs = rng.choice(np.arange(1, 1 << rangebits), size=count, replace=False)
        for x in xs:
            w.write(int(x), rangebits)
    # two residues
    w.write(1, 6)
    for which, r in enumerate(res_specs):
        w.write(r["type"], 16)
        half = bs[which] // 2
        limit = half * (channels if r["type"] == 2 else 1)
        w.write(0, 24)
        w.write(int(rng.integers(limit // 2, limit + 1)), 24)
        w.write(r["grouping"] - 1, 24)
        w.write(r["parts"] - 1, 6)
        w.write(r["phrase"], 8)
        for c in r["cascade"]:
            w.write(c & 7, 3)
            if c >> 3:
                w.write(1, 1)
                w.write(c >> 3, 5)
            else:
                w.write(0, 1)
        for row in r["stage_books"]:
            for idx in row:
                w.write(idx, 8)
    # two mappings
    w.write(1, 6)
    for which in range(2):
        w.write(0, 16)
        submaps = int(rng.integers(1, 3)) if channels > 1 else 1
        if submaps > 1:
            w.write(1, 1)
            w.write(submaps - 1, 4)
        else:
            w.write(0, 1)
        steps = int(rng.integers(0, 3)) if channels > 1 else 0
        if steps:
            w.write(1, 1)
            w.write(steps - 1, 8)
            for _ in range(steps):
                m, a = rng.choice(channels, size=2, replace=False)
                w.write(int(m), ilog(channels - 1))
                w.write(int(a), ilog(channels - 1))
        else:
            w.write(0, 1)
        w.write(0, 2)
        if submaps > 1:
            for _ in range(channels):
                w.write(int(rng.integers(submaps)), 4)
        for _ in range(submaps):
            w.write(0, 8)
            w.write(which, 8)                      # floor matching the block size of the modes that use this mapping
            w.write(which, 8)
    # modes: short ones on mapping 0, long ones on mapping 1
    nmodes = int(rng.integers(2, 5))
    flags = [0, 1] + [int(rng.integers(2)) for _ in range(nmodes - 2)]
    w.write(nmodes - 1, 6)
    for f in flags:
        w.write(f, 1)
        w.write(0, 16)
        w.write(0, 16)
        w.write(f, 8)
    w.write(1, 1)
    setup_header = b"\x05vorbis" + w.bytes()
    return id_header, comment_header, setup_header, flags


def random_packets(seed, flags, count=40):
    """audio packets: a valid (type = 0, mode) prefix followed by random bits of random length"""
    rng = np.random.default_rng(seed + 7919)
    modebits = ilog(len(flags) - 1)
    out = []
    for _ in range(count):
        w = BitWriter()
        w.write(0, 1)
        w.write(int(rng.integers(len(flags))), modebits)
        body = rng.integers(0, 256, size=int(rng.integers(1, 400)), dtype=np.uint8)
        for b in body:
            w.write(int(b), 8)
        out.append(w.bytes())
    return out
