// Vorbis I packet front end (include/easyav1_amd_vorbis.h): header parsing, codebooks, floor 1,
// residues 0/1/2, channel coupling -> one spectrum per channel, then the GPU synthesis of
// vorbis_synth.hip.  Host code by design (BASELINE.json keeps bitstream parsing on the CPU).
//
// Written from the Vorbis I specification (codeword assignment 3.2.1 and lookup1_values 9.2.3 as the
// text defines them: a leftmost-free-leaf tree walk and an exact integer root -- not libvorbis's
// marker-array / float-estimate routines); the residue loops follow libvorbis's order of float
// additions because that order is what makes the PCM bit-identical.  The reference implementation
// it must agree with bit for bit is libvorbis as vendored in /root/reference/ext/minivorbis/minivorbis.h:
//   setup header      _vorbis_unpack_books :14424, vorbis_staticbook_unpack :18031
//   VQ values         _book_unquantize :18554 (double arithmetic, one rounding to float per value)
//   packet            vorbis_synthesis :12786, mapping0_inverse :17729
//   floor 1           floor1_unpack :14952, floor1_look :15016, floor1_inverse1 :15791, _inverse2 :15877
//   residue           res0_unpack :16336, _01inverse :16788, res2_inverse :16947,
//                     vorbis_book_decodevs_add / decodev_add / decodevv_add :18259-18340
// Float results depend on the ORDER of the additions into the residue vector, which is therefore kept
// exactly as the specification (and libvorbis) defines it: stage by stage, partition by partition.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_vorbis.h"

namespace {

int ilog(uint32_t v)
{
    int r = 0;
    for (; v; r++) v >>= 1;
    return r;
}

// LSb-first bit reader with libogg's end-of-packet rule: a read that would pass the end fails and
// so does every read after it
struct Bits {
    const uint8_t *p;
    size_t nbits, pos = 0;
    bool eop = false;
    Bits(const uint8_t *data, size_t bytes) : p(data), nbits(bytes * 8) {}
    long read(int n)
    {
        if (eop || pos + size_t(n) > nbits) {
            eop = true;
            pos = nbits;
            return -1;
        }
        uint32_t v = 0;
        for (int i = 0; i < n; i++, pos++) v |= uint32_t((p[pos >> 3] >> (pos & 7)) & 1) << i;
        return long(v);
    }
};

// 32-bit packed float of the setup header: 21-bit mantissa, 10-bit exponent biased by 788, sign
float float32_unpack(uint32_t v)
{
    double mant = double(v & 0x1fffff);
    if (v & 0x80000000u) mant = -mant;
    int e = int((v & 0x7fe00000u) >> 21) - 20 - 768;
    e = e > 63 ? 63 : (e < -63 ? -63 : e);
    return float(std::ldexp(mant, e));
}

struct Book {
    int dim = 0, entries = 0, used = 0, maptype = 0;
    std::vector<uint8_t> len;
    std::vector<float> val;       // entries * dim (maptype 1, 2)
    std::vector<int32_t> tree;    // node i: tree[2i + bit] = next node (> 0), ~entry (< 0) or 0 (nothing)
    // scalar decode: walk the tree bit by bit; -1 at end of packet or on an unassigned codeword
    long decode(Bits &b) const
    {
        if (!used) return -1;
        int node = 0;
        for (;;) {
            const long bit = b.read(1);
            if (bit < 0) return -1;
            const int32_t nx = tree[size_t(2 * node + bit)];
            if (nx < 0) return ~nx;
            if (nx == 0) return -1;
            node = nx;
        }
    }
};

// Vorbis I 9.2.3 lookup1_values: "the greatest integer value for which [return_value] to the power of
// [codebook_dimensions] is less than or equal to [codebook_entries]".  Exact integers: bisection on v
// with a power that stops as soon as it exceeds `entries` (entries < 2^24, so nothing overflows).
long lookup1_values(long entries, long dim)
{
    if (entries < 1 || dim < 1) return 0;   // a book without entries has no value vector
    auto power_fits = [&](long v) {   // v^dim <= entries ?
        if (v <= 1) return true;      // 0 and 1 (entries >= 1)
        long p = 1;
        for (long k = 0; k < dim; k++) {   // p doubles at least: leaves after at most 25 rounds
            if (p > entries / v) return false;
            p *= v;
        }
        return true;
    };
    long lo = 0, hi = entries;   // invariant: lo^dim <= entries; hi = entries fits only when dim == 1
    while (lo < hi) {
        const long mid = lo + (hi - lo + 1) / 2;
        if (power_fits(mid)) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Vorbis I 3.2.1: the codewords are not transmitted, only their lengths; entry after entry takes "the
// lowest valued unused binary Huffman codeword" of its length, which in the decision tree (bit 0 =
// left branch, first bit at the root) is the LEFTMOST leaf position of that depth that is still free.
// The tree is built directly that way.  `room[n]` = depth below node n of the nearest unoccupied
// position (kNoRoom: every path under n ends in a leaf), so finding the position costs one walk of at
// most 32 steps and a codeword that would collide with an assigned one (3.2.1: an overspecified tree)
// shows up as "no room at that depth".  An underspecified tree -- positions left over once all
// entries are placed -- is an error as well, except for the one-entry book, whose single codeword
// "0" was accepted after the fact by the reference decoder and is accepted here too.
bool build_tree(Book &bk)
{
    constexpr int kNoRoom = 1 << 20;
    bk.tree.assign(2, 0);
    std::vector<int> room(1, 1);   // both branches of the root are unoccupied: nearest position 1 below
    auto slot_room = [&](int32_t s) { return s == 0 ? 0 : (s < 0 ? kNoRoom : room[size_t(s)]); };
    int path[33];
    int single_len = 0;
    for (int i = 0; i < bk.entries; i++) {
        const int L = bk.len[size_t(i)];
        if (!L) continue;
        if (L > 32 || room[0] > L) return false;   // no codeword of that length is left
        int node = 0;
        for (int depth = L; depth >= 1; depth--) {   // `depth` levels below `node` is where the leaf goes
            path[L - depth] = node;
            const int bit = slot_room(bk.tree[size_t(2 * node)]) <= depth - 1 ? 0 : 1;
            const size_t at = size_t(2 * node + bit);
            if (depth == 1) {
                bk.tree[at] = ~int32_t(i);
            } else if (bk.tree[at] == 0) {
                const int32_t fresh = int32_t(room.size());
                bk.tree[at] = fresh;
                bk.tree.resize(bk.tree.size() + 2, 0);
                room.push_back(1);
                node = fresh;
            } else {
                node = bk.tree[at];
            }
        }
        for (int k = L - 1; k >= 0; k--) {   // nearest free position of every node on the way back up
            const int n = path[k];
            const int r = std::min(slot_room(bk.tree[size_t(2 * n)]), slot_room(bk.tree[size_t(2 * n + 1)]));
            room[size_t(n)] = r >= kNoRoom ? kNoRoom : r + 1;
        }
        single_len = L;
        bk.used++;
    }
    if (bk.used == 0) return true;                       // a book nobody can decode from; using it ends the packet
    if (bk.used == 1 && single_len == 1) return true;    // the one-entry book: codeword "0"
    return room[0] >= kNoRoom;                           // every position taken: fully specified tree
}

bool unpack_book(Bits &b, Book &bk)
{
    if (b.read(24) != 0x564342) return false;
    bk.dim = int(b.read(16));
    bk.entries = int(b.read(24));
    if (bk.entries < 0 || bk.dim < 0) return false;
    if (ilog(uint32_t(bk.dim)) + ilog(uint32_t(bk.entries)) > 24) return false;
    bk.len.assign(size_t(bk.entries), 0);
    const long ordered = b.read(1);
    if (ordered == 0) {
        const long sparse = b.read(1);
        for (int i = 0; i < bk.entries; i++) {
            if (sparse) {
                const long flag = b.read(1);
                if (flag < 0) return false;
                if (!flag) continue;
            }
            const long n = b.read(5);
            if (n < 0) return false;
            bk.len[size_t(i)] = uint8_t(n + 1);
        }
    } else if (ordered == 1) {
        long length = b.read(5) + 1;
        if (length == 0) return false;
        for (int i = 0; i < bk.entries;) {
            const long num = b.read(ilog(uint32_t(bk.entries - i)));
            if (num < 0 || length > 32 || num > bk.entries - i || (num > 0 && ((num - 1) >> (length - 1)) > 1)) return false;
            for (long j = 0; j < num; j++, i++) bk.len[size_t(i)] = uint8_t(length);
            length++;
        }
    } else {
        return false;
    }
    bk.maptype = int(b.read(4));
    if (bk.maptype == 1 || bk.maptype == 2) {
        const uint32_t qmin = uint32_t(b.read(32)), qdelta = uint32_t(b.read(32));
        const int qbits = int(b.read(4)) + 1;
        const long seq = b.read(1);
        if (seq < 0) return false;
        const long nq = bk.maptype == 1 ? (bk.dim == 0 ? 0 : lookup1_values(bk.entries, bk.dim)) : long(bk.entries) * bk.dim;
        std::vector<uint32_t> q(size_t(nq), 0u);
        for (long i = 0; i < nq; i++) {
            const long v = b.read(qbits);
            if (v < 0) return false;
            q[size_t(i)] = uint32_t(v);
        }
        // value = |q| * delta + min + last, evaluated in double, rounded to float once per value
        const float mindel = float32_unpack(qmin), delta = float32_unpack(qdelta);
        bk.val.assign(size_t(bk.entries) * size_t(bk.dim), 0.f);
        for (int j = 0; j < bk.entries; j++) {
            float last = 0.f;
            long div = 1;
            for (int k = 0; k < bk.dim; k++) {
                const uint32_t qv = bk.maptype == 1 ? q[size_t((j / div) % nq)] : q[size_t(j) * size_t(bk.dim) + size_t(k)];
                const float val = float(std::fabs(double(float(qv))) * double(delta) + double(mindel) + double(last));
                if (seq) last = val;
                bk.val[size_t(j) * size_t(bk.dim) + size_t(k)] = val;
                if (bk.maptype == 1) div *= nq;
            }
        }
    } else if (bk.maptype != 0) {
        return false;
    }
    if (b.eop) return false;
    return build_tree(bk);
}

struct Floor1 {
    int partitions = 0, partclass[32] = {0}, class_dim[16] = {0}, class_subs[16] = {0}, class_book[16] = {0};
    int class_subbook[16][8];
    int mult = 1, posts = 0, quant_q = 256;
    int postlist[65 + 2] = {0};
    int order[65 + 2] = {0};       // post indices sorted by position
    int lo[65] = {0}, hi[65] = {0};
};

bool unpack_floor1(Bits &b, Floor1 &f, int nbooks)
{
    f.partitions = int(b.read(5));
    int maxclass = -1;
    for (int j = 0; j < f.partitions; j++) {
        f.partclass[j] = int(b.read(4));
        if (f.partclass[j] < 0) return false;
        if (f.partclass[j] > maxclass) maxclass = f.partclass[j];
    }
    for (int j = 0; j <= maxclass; j++) {
        f.class_dim[j] = int(b.read(3)) + 1;
        f.class_subs[j] = int(b.read(2));
        if (f.class_subs[j] < 0) return false;
        if (f.class_subs[j]) f.class_book[j] = int(b.read(8));
        if (f.class_book[j] < 0 || f.class_book[j] >= nbooks) return false;
        for (int k = 0; k < (1 << f.class_subs[j]); k++) {
            f.class_subbook[j][k] = int(b.read(8)) - 1;
            if (f.class_subbook[j][k] < -1 || f.class_subbook[j][k] >= nbooks) return false;
        }
    }
    f.mult = int(b.read(2)) + 1;
    const int rangebits = int(b.read(4));
    if (rangebits < 0) return false;
    int count = 0;
    for (int j = 0, k = 0; j < f.partitions; j++) {
        count += f.class_dim[f.partclass[j]];
        if (count > 63) return false;
        for (; k < count; k++) {
            const int t = f.postlist[k + 2] = int(b.read(rangebits));
            if (t < 0 || t >= (1 << rangebits)) return false;
        }
    }
    f.postlist[0] = 0;
    f.postlist[1] = 1 << rangebits;
    f.posts = count + 2;
    for (int i = 0; i < f.posts; i++) f.order[i] = i;
    for (int i = 1; i < f.posts; i++)   // insertion sort by position (positions are distinct, checked next)
        for (int j = i; j > 0 && f.postlist[f.order[j - 1]] > f.postlist[f.order[j]]; j--) std::swap(f.order[j - 1], f.order[j]);
    for (int i = 1; i < f.posts; i++)
        if (f.postlist[f.order[i - 1]] == f.postlist[f.order[i]]) return false;
    static const int qq[4] = {256, 128, 86, 64};
    f.quant_q = qq[f.mult - 1];
    for (int i = 0; i < f.posts - 2; i++) {   // nearest already-known posts below / above each post
        int lo = 0, hi = 1, lx = 0, hx = f.postlist[1];
        const int cx = f.postlist[i + 2];
        for (int j = 0; j < i + 2; j++) {
            const int x = f.postlist[j];
            if (x > lx && x < cx) {
                lo = j;
                lx = x;
            }
            if (x < hx && x > cx) {
                hi = j;
                hx = x;
            }
        }
        f.lo[i] = lo;
        f.hi[i] = hi;
    }
    return !b.eop;
}

struct Residue {
    int type = 0, begin = 0, end = 0, grouping = 1, partitions = 1, groupbook = 0, stages = 0, partvals = 1;
    int cascade[64] = {0};
    int books[64][8];          // book index per (partition class, stage), -1 = none
};

bool unpack_residue(Bits &b, Residue &r, const std::vector<Book> &books)
{
    r.begin = int(b.read(24));
    r.end = int(b.read(24));
    r.grouping = int(b.read(24)) + 1;
    r.partitions = int(b.read(6)) + 1;
    r.groupbook = int(b.read(8));
    if (r.groupbook < 0 || r.groupbook >= int(books.size())) return false;
    for (int j = 0; j < r.partitions; j++) {
        int cascade = int(b.read(3));
        const long cflag = b.read(1);
        if (cflag < 0) return false;
        if (cflag) {
            const long c = b.read(5);
            if (c < 0) return false;
            cascade |= int(c) << 3;
        }
        r.cascade[j] = cascade;
    }
    r.stages = 0;
    for (int j = 0; j < r.partitions; j++) {
        if (ilog(uint32_t(r.cascade[j])) > r.stages) r.stages = ilog(uint32_t(r.cascade[j]));
        for (int k = 0; k < 8; k++) {
            r.books[j][k] = -1;
            if (r.cascade[j] & (1 << k)) {
                const int bk = int(b.read(8));
                if (bk < 0 || bk >= int(books.size()) || books[size_t(bk)].maptype == 0) return false;
                r.books[j][k] = bk;
            }
        }
    }
    const Book &gb = books[size_t(r.groupbook)];
    if (gb.dim < 1) return false;
    r.partvals = 1;
    for (int d = 0; d < gb.dim; d++) {
        r.partvals *= r.partitions;
        if (r.partvals > gb.entries) return false;
    }
    return !b.eop;
}

struct Mapping {
    int submaps = 1, coupling_steps = 0;
    int mag[256], ang[256];
    int chmux[256] = {0};
    int floorsub[16] = {0}, ressub[16] = {0};
};

struct Mode {
    int blockflag = 0, mapping = 0;
};

}  // namespace

struct ea1_vorbis_decoder {
    int channels = 0, bs[2] = {0, 0}, device = -1;
    long rate = 0;
    std::vector<Book> books;
    std::vector<Floor1> floors;
    std::vector<Residue> residues;
    std::vector<Mapping> maps;
    std::vector<Mode> modes;
    int modebits = 0;
    // GPU synthesis
    ea1_vorbis_synth *synth = nullptr;
    float *h_spectra = nullptr, *d_spectra = nullptr, *d_pcm = nullptr;   // pinned host / device staging
    size_t cap_blocks = 0;
    std::vector<float> scratch;   // host-only decoders: nothing
};

namespace {

const float *db_table()
{
    // Vorbis I 10.1 floor1_inverse_dB_table.  The specification prints the values with 8 significant
    // digits; they are exp(0.11512925 * dB) with dB = (i - 255) * 140 / 256 evaluated in double and
    // printed as %.7e -- printing and re-parsing reproduces all 256 floats bit for bit
    // (tests/test_vorbis_front.py checks them against the reference build).
    static float table[256];
    static bool ready = false;
    if (!ready) {
        for (int i = 0; i < 256; i++) {
            char buf[32];
            snprintf(buf, sizeof buf, "%.7e", std::exp((double(i - 255) * 0.546875) * 0.11512925));
            table[i] = strtof(buf, nullptr);
        }
        ready = true;
    }
    return table;
}

int render_point(int x0, int x1, int y0, int y1, int x)
{
    y0 &= 0x7fff;
    y1 &= 0x7fff;
    const int dy = y1 - y0, adx = x1 - x0, ady = dy < 0 ? -dy : dy;
    const int off = ady * (x - x0) / adx;
    return dy < 0 ? y0 - off : y0 + off;
}

// multiply d[x0 .. min(x1, n)) by the dB line from (x0, y0) to (x1, y1) (integer line, 7.2.4)
void render_line(int n, int x0, int x1, int y0, int y1, float *d, const float *db)
{
    const int dy = y1 - y0, adx = x1 - x0;
    int ady = dy < 0 ? -dy : dy;
    const int base = dy / adx, sy = dy < 0 ? base - 1 : base + 1;
    int x = x0, y = y0, err = 0;
    ady -= (base < 0 ? -base : base) * adx;
    if (n > x1) n = x1;
    if (x < n) d[x] *= db[y];
    while (++x < n) {
        err += ady;
        if (err >= adx) {
            err -= adx;
            y += sy;
        } else {
            y += base;
        }
        d[x] *= db[y];
    }
}

// floor 1 packet decode: fit[] = post amplitudes, bit 15 set = post not coded (interpolated)
bool floor1_decode(const ea1_vorbis_decoder &dec, const Floor1 &f, Bits &b, int *fit)
{
    if (b.read(1) != 1) return false;
    const int bits = ilog(uint32_t(f.quant_q - 1));
    fit[0] = int(b.read(bits));
    fit[1] = int(b.read(bits));
    for (int i = 0, j = 2; i < f.partitions; i++) {
        const int cls = f.partclass[i], cdim = f.class_dim[cls], csubbits = f.class_subs[cls], csub = 1 << csubbits;
        int cval = 0;
        if (csubbits) {
            cval = int(dec.books[size_t(f.class_book[cls])].decode(b));
            if (cval == -1) return false;
        }
        for (int k = 0; k < cdim; k++) {
            const int book = f.class_subbook[cls][cval & (csub - 1)];
            cval >>= csubbits;
            if (book >= 0) {
                if ((fit[j + k] = int(dec.books[size_t(book)].decode(b))) == -1) return false;
            } else {
                fit[j + k] = 0;
            }
        }
        j += cdim;
    }
    if (b.eop) return false;
    for (int i = 2; i < f.posts; i++) {
        const int predicted = render_point(f.postlist[f.lo[i - 2]], f.postlist[f.hi[i - 2]], fit[f.lo[i - 2]], fit[f.hi[i - 2]], f.postlist[i]);
        const int hiroom = f.quant_q - predicted, loroom = predicted;
        const int room = (hiroom < loroom ? hiroom : loroom) * 2;   // may be negative on corrupt input (libvorbis shifts; same value)
        int val = fit[i];
        if (val) {
            if (val >= room) val = hiroom > loroom ? val - loroom : -1 - (val - hiroom);
            else val = (val & 1) ? -((val + 1) >> 1) : val >> 1;
            fit[i] = (val + predicted) & 0x7fff;
            fit[f.lo[i - 2]] &= 0x7fff;
            fit[f.hi[i - 2]] &= 0x7fff;
        } else {
            fit[i] = predicted | 0x8000;
        }
    }
    return true;
}

void floor1_apply(const Floor1 &f, const int *fit, int n, float *out)
{
    const float *db = db_table();
    int hx = 0, lx = 0;
    int ly = fit[0] * f.mult;
    ly = ly < 0 ? 0 : (ly > 255 ? 255 : ly);
    for (int j = 1; j < f.posts; j++) {
        const int cur = f.order[j];
        int hy = fit[cur] & 0x7fff;
        if (hy == fit[cur]) {
            hx = f.postlist[cur];
            hy *= f.mult;
            hy = hy < 0 ? 0 : (hy > 255 ? 255 : hy);
            render_line(n, lx, hx, ly, hy, out, db);
            lx = hx;
            ly = hy;
        }
    }
    for (int j = hx; j < n; j++) out[j] *= db[ly];
}

// residue types 0 and 1: vectors[] = the submap's channels that carry a residue
void residue01_decode(const ea1_vorbis_decoder &dec, const Residue &r, Bits &b, float **vec, int ch, int half)
{
    const Book &phrase = dec.books[size_t(r.groupbook)];
    const int spp = r.grouping, ppw = phrase.dim;
    const int end = r.end < half ? r.end : half;
    const int n = end - r.begin;
    if (n <= 0 || ch <= 0) return;
    const int partvals = n / spp, partwords = (partvals + ppw - 1) / ppw;
    std::vector<int> cls(size_t(ch) * size_t(partwords) * size_t(ppw), 0);
    for (int s = 0; s < r.stages; s++) {
        for (int i = 0, l = 0; i < partvals; l++) {
            if (s == 0) {
                for (int j = 0; j < ch; j++) {
                    long temp = phrase.decode(b);
                    if (temp == -1 || temp >= r.partvals) return;
                    // the phrase value is a base-`partitions` number, most significant digit first
                    long mult = r.partvals / r.partitions;
                    for (int k = 0; k < ppw; k++) {
                        const long deco = temp / mult;
                        temp -= deco * mult;
                        mult /= r.partitions;
                        cls[(size_t(j) * size_t(partwords) + size_t(l)) * size_t(ppw) + size_t(k)] = int(deco);
                    }
                }
            }
            for (int k = 0; k < ppw && i < partvals; k++, i++)
                for (int j = 0; j < ch; j++) {
                    const int c = cls[(size_t(j) * size_t(partwords) + size_t(l)) * size_t(ppw) + size_t(k)];
                    if (!(r.cascade[c] & (1 << s)) || r.books[c][s] < 0) continue;
                    const Book &bk = dec.books[size_t(r.books[c][s])];
                    if (!bk.used) continue;
                    float *a = vec[j] + r.begin + i * spp;
                    if (r.type == 0) {
                        // type 0: the vector components are interleaved with stride n / dim
                        const int step = spp / bk.dim;
                        std::vector<const float *> t(size_t(step), nullptr);
                        for (int q = 0; q < step; q++) {
                            const long e = bk.decode(b);
                            if (e == -1) return;
                            t[size_t(q)] = bk.val.data() + size_t(e) * size_t(bk.dim);
                        }
                        for (int d = 0, o = 0; d < bk.dim; d++, o += step)
                            for (int q = 0; o + q < spp && q < step; q++) a[o + q] += t[size_t(q)][d];
                    } else {
                        for (int q = 0; q < spp;) {
                            const long e = bk.decode(b);
                            if (e == -1) return;
                            const float *t = bk.val.data() + size_t(e) * size_t(bk.dim);
                            for (int d = 0; q < spp && d < bk.dim;) a[q++] += t[d++];
                        }
                    }
                }
        }
    }
}

// residue type 2: the channels of the submap interleaved into one vector
void residue2_decode(const ea1_vorbis_decoder &dec, const Residue &r, Bits &b, float **vec, const int *nonzero, int ch, int half)
{
    const Book &phrase = dec.books[size_t(r.groupbook)];
    const int spp = r.grouping, ppw = phrase.dim;
    const int max = half * ch;
    const int end = r.end < max ? r.end : max;
    const int n = end - r.begin;
    if (n <= 0 || ch <= 0) return;
    bool any = false;
    for (int i = 0; i < ch; i++) any |= nonzero[i] != 0;
    if (!any) return;
    const int partvals = n / spp, partwords = (partvals + ppw - 1) / ppw;
    std::vector<int> cls(size_t(partwords) * size_t(ppw), 0);
    for (int s = 0; s < r.stages; s++) {
        for (int i = 0, l = 0; i < partvals; l++) {
            if (s == 0) {
                long temp = phrase.decode(b);
                if (temp == -1 || temp >= r.partvals) return;
                long mult = r.partvals / r.partitions;
                for (int k = 0; k < ppw; k++) {
                    const long deco = temp / mult;
                    temp -= deco * mult;
                    mult /= r.partitions;
                    cls[size_t(l) * size_t(ppw) + size_t(k)] = int(deco);
                }
            }
            for (int k = 0; k < ppw && i < partvals; k++, i++) {
                const int c = cls[size_t(l) * size_t(ppw) + size_t(k)];
                if (!(r.cascade[c] & (1 << s)) || r.books[c][s] < 0) continue;
                const Book &bk = dec.books[size_t(r.books[c][s])];
                if (!bk.used) continue;
                const long offset = long(i) * spp + r.begin;
                int chptr = 0;
                const long m = (offset + spp) / ch;
                for (long pos = offset / ch; pos < m;) {
                    const long e = bk.decode(b);
                    if (e == -1) return;
                    const float *t = bk.val.data() + size_t(e) * size_t(bk.dim);
                    for (int d = 0; pos < m && d < bk.dim; d++) {
                        vec[chptr++][pos] += t[d];
                        if (chptr == ch) {
                            chptr = 0;
                            pos++;
                        }
                    }
                }
            }
        }
    }
}

bool parse_setup(ea1_vorbis_decoder &d, const uint8_t *p, size_t n)
{
    if (n < 7 || p[0] != 5 || memcmp(p + 1, "vorbis", 6)) return false;
    Bits b(p + 7, n - 7);
    const int nbooks = int(b.read(8)) + 1;
    if (nbooks <= 0) return false;
    d.books.resize(size_t(nbooks));
    for (Book &bk : d.books)
        if (!unpack_book(b, bk)) return false;
    const int times = int(b.read(6)) + 1;
    for (int i = 0; i < times; i++)
        if (b.read(16) != 0) return false;
    const int nfloors = int(b.read(6)) + 1;
    if (nfloors <= 0) return false;
    d.floors.resize(size_t(nfloors));
    for (Floor1 &f : d.floors) {
        if (b.read(16) != 1) return false;   // floor type 0 is not supported (see the header)
        if (!unpack_floor1(b, f, nbooks)) return false;
    }
    const int nres = int(b.read(6)) + 1;
    if (nres <= 0) return false;
    d.residues.resize(size_t(nres));
    for (Residue &r : d.residues) {
        r.type = int(b.read(16));
        if (r.type < 0 || r.type > 2) return false;
        if (!unpack_residue(b, r, d.books)) return false;
    }
    const int nmaps = int(b.read(6)) + 1;
    if (nmaps <= 0) return false;
    d.maps.resize(size_t(nmaps));
    for (Mapping &m : d.maps) {
        if (b.read(16) != 0) return false;
        long f = b.read(1);
        if (f < 0) return false;
        m.submaps = f ? int(b.read(4)) + 1 : 1;
        f = b.read(1);
        if (f < 0) return false;
        if (f) {
            m.coupling_steps = int(b.read(8)) + 1;
            for (int i = 0; i < m.coupling_steps; i++) {
                const int bits = ilog(uint32_t(d.channels - 1));
                m.mag[i] = int(b.read(bits));
                m.ang[i] = int(b.read(bits));
                if (m.mag[i] < 0 || m.ang[i] < 0 || m.mag[i] == m.ang[i] || m.mag[i] >= d.channels || m.ang[i] >= d.channels) return false;
            }
        }
        if (b.read(2) != 0) return false;
        if (m.submaps > 1)
            for (int i = 0; i < d.channels; i++) {
                m.chmux[i] = int(b.read(4));
                if (m.chmux[i] < 0 || m.chmux[i] >= m.submaps) return false;
            }
        for (int i = 0; i < m.submaps; i++) {
            b.read(8);
            m.floorsub[i] = int(b.read(8));
            m.ressub[i] = int(b.read(8));
            if (m.floorsub[i] < 0 || m.floorsub[i] >= nfloors || m.ressub[i] < 0 || m.ressub[i] >= nres) return false;
        }
    }
    const int nmodes = int(b.read(6)) + 1;
    if (nmodes <= 0) return false;
    d.modes.resize(size_t(nmodes));
    for (Mode &m : d.modes) {
        m.blockflag = int(b.read(1));
        const long wt = b.read(16), tt = b.read(16);
        m.mapping = int(b.read(8));
        if (wt != 0 || tt != 0 || m.mapping < 0 || m.mapping >= nmaps) return false;
    }
    if (b.read(1) != 1) return false;
    d.modebits = ilog(uint32_t(nmodes - 1));
    if (getenv("EA1_VORBIS_DEBUG")) {
        fprintf(stderr, "vorbis setup: %d books, %d floors, %d residues, %d maps, %d modes:", nbooks, nfloors, nres, nmaps, nmodes);
        for (const Mode &m : d.modes) fprintf(stderr, " (flag %d map %d)", m.blockflag, m.mapping);
        for (const Mapping &m : d.maps) fprintf(stderr, " map[sub %d cpl %d floor %d res %d]", m.submaps, m.coupling_steps, m.floorsub[0], m.ressub[0]);
        for (const Residue &r : d.residues) fprintf(stderr, " res[type %d %d..%d grp %d parts %d book %d stages %d]", r.type, r.begin, r.end, r.grouping, r.partitions, r.groupbook, r.stages);
        for (const Floor1 &f : d.floors) fprintf(stderr, " floor[parts %d posts %d mult %d]", f.partitions, f.posts, f.mult);
        fprintf(stderr, "\n");
    }
    return true;
}

// one audio packet -> spectra rows (n/2 floats each, row stride `stride`); false = not decodable
bool packet_spectra(const ea1_vorbis_decoder &d, const uint8_t *packet, size_t size, float *spectra, size_t stride, int *W_out)
{
    Bits b(packet, size);
    if (b.read(1) != 0) return false;
    const long mode = b.read(d.modebits);
    if (mode < 0 || mode >= long(d.modes.size())) return false;
    const int W = d.modes[size_t(mode)].blockflag;
    if (W) {
        b.read(1);
        if (b.read(1) < 0) return false;
    }
    const Mapping &map = d.maps[size_t(d.modes[size_t(mode)].mapping)];
    const int n = d.bs[W], half = n / 2, ch = d.channels;
    std::vector<int> fit(size_t(ch) * 67, 0), nonzero(size_t(ch), 0);
    std::vector<char> has_floor(size_t(ch), 0);
    for (int i = 0; i < ch; i++) {
        const Floor1 &f = d.floors[size_t(map.floorsub[map.chmux[i]])];
        has_floor[size_t(i)] = floor1_decode(d, f, b, &fit[size_t(i) * 67]) ? 1 : 0;
        nonzero[size_t(i)] = has_floor[size_t(i)];
        memset(spectra + size_t(i) * stride, 0, sizeof(float) * size_t(half));
    }
    for (int i = 0; i < map.coupling_steps; i++)
        if (nonzero[size_t(map.mag[i])] || nonzero[size_t(map.ang[i])]) nonzero[size_t(map.mag[i])] = nonzero[size_t(map.ang[i])] = 1;
    for (int sm = 0; sm < map.submaps; sm++) {
        std::vector<float *> vec;
        std::vector<int> nz;
        for (int j = 0; j < ch; j++)
            if (map.chmux[j] == sm) {
                vec.push_back(spectra + size_t(j) * stride);
                nz.push_back(nonzero[size_t(j)]);
            }
        const Residue &r = d.residues[size_t(map.ressub[sm])];
        if (r.type == 2) {
            residue2_decode(d, r, b, vec.data(), nz.data(), int(vec.size()), half);
        } else {
            std::vector<float *> used;
            for (size_t j = 0; j < vec.size(); j++)
                if (nz[j]) used.push_back(vec[j]);
            residue01_decode(d, r, b, used.data(), int(used.size()), half);
        }
    }
    for (int i = map.coupling_steps - 1; i >= 0; i--) {
        float *pm = spectra + size_t(map.mag[i]) * stride, *pa = spectra + size_t(map.ang[i]) * stride;
        for (int j = 0; j < half; j++) {
            const float mag = pm[j], ang = pa[j];
            if (mag > 0) {
                if (ang > 0) {
                    pm[j] = mag;
                    pa[j] = mag - ang;
                } else {
                    pa[j] = mag;
                    pm[j] = mag + ang;
                }
            } else {
                if (ang > 0) {
                    pm[j] = mag;
                    pa[j] = mag + ang;
                } else {
                    pa[j] = mag;
                    pm[j] = mag - ang;
                }
            }
        }
    }
    for (int i = 0; i < ch; i++) {
        float *row = spectra + size_t(i) * stride;
        if (has_floor[size_t(i)]) floor1_apply(d.floors[size_t(map.floorsub[map.chmux[i]])], &fit[size_t(i) * 67], half, row);
        else memset(row, 0, sizeof(float) * size_t(half));
    }
    *W_out = W;
    return true;
}

void release_gpu(ea1_vorbis_decoder *d)
{
    if (d->device < 0) return;
    ea1::DeviceGuard g(d->device);
    if (d->h_spectra) (void)hipHostFree(d->h_spectra);
    if (d->d_spectra) (void)hipFree(d->d_spectra);
    if (d->d_pcm) (void)hipFree(d->d_pcm);
    d->h_spectra = d->d_spectra = d->d_pcm = nullptr;
    d->cap_blocks = 0;
}

}  // namespace

extern "C" const float *ea1_vorbis_floor1_db_table(void) { return db_table(); }

extern "C" int ea1_vorbis_decoder_create(ea1_vorbis_decoder **out, const uint8_t *id, size_t id_size, const uint8_t *comment,
                                         size_t comment_size, const uint8_t *setup, size_t setup_size, int device)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_create: out is NULL");
    *out = nullptr;
    if (!id || !setup) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_create: NULL header");
    (void)comment;
    (void)comment_size;   // the comment header carries no decoding parameters
    auto *d = new (std::nothrow) ea1_vorbis_decoder();
    if (!d) return ea1::fail(EA1_ENOMEM, "out of memory");
    // identification header (Vorbis I 4.2.2)
    bool ok = id_size >= 30 && id[0] == 1 && !memcmp(id + 1, "vorbis", 6);
    if (ok) {
        Bits b(id + 7, id_size - 7);
        ok = b.read(32) == 0;
        d->channels = int(b.read(8));
        d->rate = b.read(32);
        b.read(32);
        b.read(32);
        b.read(32);
        const long b0 = b.read(4), b1 = b.read(4);
        ok = ok && b0 >= 6 && b1 >= b0 && b1 <= 13 && d->channels >= 1 && d->rate >= 1 && b.read(1) == 1;
        d->bs[0] = 1 << b0;
        d->bs[1] = 1 << b1;
    }
    if (!ok) {
        delete d;
        return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_create: bad identification header");
    }
    if (!parse_setup(*d, setup, setup_size)) {
        delete d;
        return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_create: bad or unsupported setup header");
    }
    d->device = device;
    if (device >= 0) {
        const int rc = ea1_vorbis_synth_create(&d->synth, d->channels, d->bs[0], d->bs[1], device);
        if (rc != EA1_OK) {
            delete d;
            return rc;
        }
    }
    *out = d;
    return EA1_OK;
}

extern "C" void ea1_vorbis_decoder_destroy(ea1_vorbis_decoder *d)
{
    if (!d) return;
    release_gpu(d);
    if (d->synth) ea1_vorbis_synth_destroy(d->synth);
    delete d;
}

extern "C" int ea1_vorbis_decoder_info(const ea1_vorbis_decoder *d, int *channels, long *rate, int *bs0, int *bs1)
{
    if (!d) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_info: NULL handle");
    if (channels) *channels = d->channels;
    if (rate) *rate = d->rate;
    if (bs0) *bs0 = d->bs[0];
    if (bs1) *bs1 = d->bs[1];
    return EA1_OK;
}

extern "C" int ea1_vorbis_decoder_spectra(ea1_vorbis_decoder *d, const uint8_t *packet, size_t size, float *spectra,
                                          size_t stride, int *long_block)
{
    if (!d || !packet || !spectra || !long_block || stride < size_t(d->bs[1] / 2))
        return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_spectra: bad arguments");
    if (!packet_spectra(*d, packet, size, spectra, stride, long_block)) return ea1::fail(EA1_EINVAL, "not a decodable Vorbis audio packet");
    return EA1_OK;
}

extern "C" int ea1_vorbis_decoder_restart(ea1_vorbis_decoder *d)
{
    if (!d) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_restart: NULL handle");
    return d->synth ? ea1_vorbis_synth_restart(d->synth) : EA1_OK;
}

static int decode_packets(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes, int npackets, float *pcm, size_t pcm_stride,
                          int *counts, int64_t *total, uint8_t *accepted, void *stream);

extern "C" int ea1_vorbis_decoder_decode(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes, int npackets,
                                         float *pcm, size_t pcm_stride, int *counts, int64_t *total, void *stream)
{
    return decode_packets(d, packets, sizes, npackets, pcm, pcm_stride, counts, total, nullptr, stream);
}

extern "C" int ea1_vorbis_decoder_decode_checked(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes, int npackets,
                                                 float *pcm, size_t pcm_stride, int *counts, int64_t *total, uint8_t *accepted, void *stream)
{
    if (!accepted && npackets > 0) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_decode_checked: accepted is NULL");
    return decode_packets(d, packets, sizes, npackets, pcm, pcm_stride, counts, total, accepted, stream);
}

static int decode_packets(ea1_vorbis_decoder *d, const uint8_t *const *packets, const size_t *sizes, int npackets, float *pcm, size_t pcm_stride,
                          int *counts, int64_t *total, uint8_t *accepted, void *stream)
{
    if (!d || !packets || !sizes || npackets < 0 || (npackets && !pcm)) return ea1::fail(EA1_EINVAL, "ea1_vorbis_decoder_decode: bad arguments");
    if (!d->synth) return ea1::fail(EA1_ESTATE, "ea1_vorbis_decoder_decode: host-only decoder (created with device < 0)");
    if (total) *total = 0;
    for (int k = 0; counts && k < npackets; k++) counts[k] = 0;
    for (int k = 0; accepted && k < npackets; k++) accepted[k] = 0;
    if (!npackets) return EA1_OK;
    ea1::DeviceGuard g(d->device);
    if (!g.ok) return ea1::fail(EA1_ENODEV, "hipSetDevice(%d) failed", d->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t half1 = size_t(d->bs[1] / 2), row = size_t(d->channels) * half1;
    if (size_t(npackets) > d->cap_blocks) {
        release_gpu(d);
        const size_t cap = size_t(npackets) + 64;
        if (hipHostMalloc(reinterpret_cast<void **>(&d->h_spectra), cap * row * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&d->d_spectra), cap * row * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&d->d_pcm), cap * row * sizeof(float)) != hipSuccess) {
            release_gpu(d);
            return ea1::fail(EA1_ENOMEM, "ea1_vorbis_decoder_decode: staging allocation failed");
        }
        d->cap_blocks = cap;
    }
    // host stage: packets -> spectra (undecodable packets are dropped, as easyav1 does)
    std::vector<uint8_t> flags;
    std::vector<int> owner;
    for (int k = 0; k < npackets; k++) {
        int W = 0;
        if (!packets[k] || !packet_spectra(*d, packets[k], sizes[k], d->h_spectra + flags.size() * row, half1, &W)) continue;
        flags.push_back(uint8_t(W));
        owner.push_back(k);
        if (accepted) accepted[k] = 1;
    }
    if (flags.empty()) return EA1_OK;
    const int nb = int(flags.size());
    const size_t stride = size_t(nb) * half1;   // upper bound of the samples per channel
    EA1_HIP(hipMemcpyAsync(d->d_spectra, d->h_spectra, size_t(nb) * row * sizeof(float), hipMemcpyHostToDevice, st));
    std::vector<int> bc(size_t(nb), 0);
    int64_t tot = 0;
    const int rc = ea1_vorbis_synth_decode(d->synth, d->d_spectra, flags.data(), nb, d->d_pcm, stride, bc.data(), &tot, st);
    if (rc != EA1_OK) return rc;
    if (size_t(tot) > pcm_stride) return ea1::fail(EA1_ECAPACITY, "ea1_vorbis_decoder_decode: %lld samples do not fit pcm_stride %zu", (long long)tot, pcm_stride);
    if (tot)
        EA1_HIP(hipMemcpy2DAsync(pcm, pcm_stride * sizeof(float), d->d_pcm, stride * sizeof(float), size_t(tot) * sizeof(float), size_t(d->channels),
                                 hipMemcpyDeviceToHost, st));
    EA1_HIP(hipStreamSynchronize(st));
    for (int i = 0; counts && i < nb; i++) counts[owner[size_t(i)]] = bc[size_t(i)];
    if (total) *total = tot;
    return EA1_OK;
}
