// A parsed frame as the payload of a private OBU_METADATA (include/easyav1_amd_ir_packet.h): the
// vehicle that lets container files drive the GPU reconstruction path through easyav1's public
// interface (the decode_video path of /root/reference/src/easyav1.c:2360-2428) while the
// block-level AV1 symbol parser does not exist.  Host code; the bytes come from a file, so the
// parser trusts nothing: sizes are checked here, contents by ea1_job_create.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_ir_packet.h"

namespace {

constexpr uint8_t kTypeLeb[2] = {uint8_t((EA1_IR_METADATA_TYPE & 0x7F) | 0x80), uint8_t(EA1_IR_METADATA_TYPE >> 7)};
constexpr char kMagic[4] = {'E', 'A', '1', 'F'};
constexpr size_t kFixed = 2 + 4 + 4 + 4 * 4 + 4 * 4;   // type, magic, version, struct sizes, num_sb + lr_count[3]

struct Sizes {
    size_t blocks, txs, coefs, aux, cdef, lr[3];
    size_t total() const { return blocks + txs + coefs + aux + cdef + lr[0] + lr[1] + lr[2]; }
};

Sizes sizes_of(const ea1_frame_desc &d, uint32_t num_sb, const uint32_t lr_count[3])
{
    Sizes s;
    s.blocks = size_t(d.num_blocks) * sizeof(ea1_block);
    s.txs = size_t(d.num_tx) * sizeof(ea1_txblock);
    s.coefs = size_t(d.num_coefs) * sizeof(int16_t);
    s.aux = d.num_aux;
    s.cdef = num_sb;
    for (int p = 0; p < 3; p++) s.lr[p] = size_t(lr_count[p]) * sizeof(ea1_lr_unit);
    return s;
}

void put32(uint8_t *&p, uint32_t v)
{
    for (int k = 0; k < 4; k++) *p++ = uint8_t(v >> (8 * k));
}
uint32_t get32(const uint8_t *&p)
{
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) v |= uint32_t(*p++) << (8 * k);
    return v;
}

}  // namespace

struct ea1_ir_storage {
    std::vector<uint64_t> mem;   // 8-byte aligned backing of every array
};

extern "C" size_t ea1_ir_packet_bytes(const ea1_ir_frame *f)
{
    if (!f) return 0;
    return kFixed + sizeof(ea1_frame_desc) + sizes_of(f->desc, f->num_sb, f->lr_count).total() + 1;
}

extern "C" int ea1_ir_packet_write(const ea1_ir_frame *f, uint8_t *out, size_t cap)
{
    if (!f || !out) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_write: NULL argument");
    const size_t need = ea1_ir_packet_bytes(f);
    if (cap < need) return ea1::fail(EA1_ECAPACITY, "ea1_ir_packet_write: %zu bytes needed, %zu given", need, cap);
    const Sizes s = sizes_of(f->desc, f->num_sb, f->lr_count);
    if ((s.blocks && !f->blocks) || (s.txs && !f->txs) || (s.coefs && !f->coefs) || (s.aux && !f->aux) || (s.cdef && !f->cdef_idx) ||
        (s.lr[0] && !f->lr[0]) || (s.lr[1] && !f->lr[1]) || (s.lr[2] && !f->lr[2]))
        return ea1::fail(EA1_EINVAL, "ea1_ir_packet_write: a counted array is NULL");
    uint8_t *p = out;
    *p++ = kTypeLeb[0];
    *p++ = kTypeLeb[1];
    memcpy(p, kMagic, 4);
    p += 4;
    put32(p, EA1_IR_VERSION);
    put32(p, sizeof(ea1_frame_desc));
    put32(p, sizeof(ea1_block));
    put32(p, sizeof(ea1_txblock));
    put32(p, sizeof(ea1_lr_unit));
    put32(p, f->num_sb);
    for (int k = 0; k < 3; k++) put32(p, f->lr_count[k]);
    memcpy(p, &f->desc, sizeof f->desc);
    p += sizeof f->desc;
    auto put = [&](const void *src, size_t n) {
        if (n) memcpy(p, src, n);
        p += n;
    };
    put(f->blocks, s.blocks);
    put(f->txs, s.txs);
    put(f->coefs, s.coefs);
    put(f->aux, s.aux);
    put(f->cdef_idx, s.cdef);
    for (int k = 0; k < 3; k++) put(f->lr[k], s.lr[k]);
    *p++ = 0x80;
    return EA1_OK;
}

extern "C" int ea1_ir_packet_probe(const uint8_t *payload, size_t size)
{
    return payload && size >= kFixed && payload[0] == kTypeLeb[0] && payload[1] == kTypeLeb[1] && !memcmp(payload + 2, kMagic, 4);
}

// the checks and the copy of both parse calls: `mem` = 8-byte aligned words to fill (NULL: only compute `words`)
static int parse_common(const uint8_t *payload, size_t size, ea1_ir_frame *out, uint64_t *mem, size_t cap_words, size_t *words_out)
{
    if (!ea1_ir_packet_probe(payload, size)) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: not a parsed-frame payload");
    const uint8_t *p = payload + 6;
    if (get32(p) != EA1_IR_VERSION) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: unknown version");
    if (get32(p) != sizeof(ea1_frame_desc) || get32(p) != sizeof(ea1_block) || get32(p) != sizeof(ea1_txblock) || get32(p) != sizeof(ea1_lr_unit))
        return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: structure sizes differ from this build");
    memset(out, 0, sizeof *out);
    out->num_sb = get32(p);
    for (int k = 0; k < 3; k++) out->lr_count[k] = get32(p);
    if (size < kFixed + sizeof(ea1_frame_desc)) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: truncated");
    memcpy(&out->desc, p, sizeof out->desc);
    p += sizeof out->desc;
    const Sizes s = sizes_of(out->desc, out->num_sb, out->lr_count);
    const size_t left = size - (kFixed + sizeof(ea1_frame_desc));
    // every addend is below 2^32 * 48, the sum cannot wrap a 64-bit size_t
    if (s.total() > left) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: the counts describe %zu bytes, the payload holds %zu", s.total(), left);
    const size_t parts[8] = {s.blocks, s.txs, s.coefs, s.aux, s.cdef, s.lr[0], s.lr[1], s.lr[2]};
    size_t words = 0;
    for (size_t n : parts) words += (n + 7) / 8;
    *words_out = words;
    if (!mem) return EA1_OK;
    if (words > cap_words) return ea1::fail(EA1_ECAPACITY, "ea1_ir_packet_parse_into: %zu bytes needed, %zu given", words * 8, cap_words * 8);
    uint64_t *w = mem;
    const void *at[8];
    for (int k = 0; k < 8; k++) {
        at[k] = parts[k] ? w : nullptr;
        if (parts[k]) {
            w[(parts[k] + 7) / 8 - 1] = 0;   // the padding of the last word
            memcpy(w, p, parts[k]);
        }
        p += parts[k];
        w += (parts[k] + 7) / 8;
    }
    out->blocks = static_cast<const ea1_block *>(at[0]);
    out->txs = static_cast<const ea1_txblock *>(at[1]);
    out->coefs = static_cast<const int16_t *>(at[2]);
    out->aux = static_cast<const uint8_t *>(at[3]);
    out->cdef_idx = static_cast<const uint8_t *>(at[4]);
    for (int k = 0; k < 3; k++) out->lr[k] = static_cast<const ea1_lr_unit *>(at[5 + k]);
    return EA1_OK;
}

extern "C" int ea1_ir_packet_parse(const uint8_t *payload, size_t size, ea1_ir_storage **storage, ea1_ir_frame *out)
{
    if (!payload || !storage || !out) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse: NULL argument");
    *storage = nullptr;
    size_t words = 0;
    int rc = parse_common(payload, size, out, nullptr, 0, &words);
    if (rc != EA1_OK) return rc;
    auto *st = new (std::nothrow) ea1_ir_storage();
    if (!st) return ea1::fail(EA1_ENOMEM, "out of memory");
    try {
        st->mem.assign(words + 1, 0);
    } catch (...) {
        delete st;
        return ea1::fail(EA1_ENOMEM, "out of memory");
    }
    rc = parse_common(payload, size, out, st->mem.data(), words + 1, &words);
    if (rc != EA1_OK) {
        delete st;
        return rc;
    }
    *storage = st;
    return EA1_OK;
}

extern "C" int ea1_ir_packet_peek(const uint8_t *payload, size_t size, ea1_frame_desc *desc)
{
    if (!payload || !desc) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_peek: NULL argument");
    ea1_ir_frame f;
    size_t words = 0;
    const int rc = parse_common(payload, size, &f, nullptr, 0, &words);
    if (rc == EA1_OK) *desc = f.desc;
    return rc;
}

extern "C" int ea1_ir_packet_parse_into(const uint8_t *payload, size_t size, void *mem, size_t cap, ea1_ir_frame *out)
{
    if (!payload || !mem || !out || (reinterpret_cast<uintptr_t>(mem) & 7)) return ea1::fail(EA1_EINVAL, "ea1_ir_packet_parse_into: NULL or unaligned argument");
    size_t words = 0;
    return parse_common(payload, size, out, static_cast<uint64_t *>(mem), cap / 8, &words);
}

extern "C" void ea1_ir_storage_destroy(ea1_ir_storage *s) { delete s; }
