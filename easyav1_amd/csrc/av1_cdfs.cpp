// The table blob of the block-level AV1 parser: layout, loading, checks.  See av1_cdfs.h.
#include "av1_cdfs.h"

#include <new>
#include <string>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_obu.h"

namespace ea1 {
namespace av1 {

#define EA1_AV1_MODE_SECTION(name, dims, n) {#name, offsetof(ModeCdfs, name), int(sizeof(ModeCdfs::name) / (sizeof(uint16_t) * ((n) + 1))), n},
#define EA1_AV1_COEF_SECTION(name, dims, n) {#name, offsetof(CoefCdfs, name), int(sizeof(CoefCdfs::name) / (sizeof(uint16_t) * ((n) + 1))), n},
const CdfSection kModeSections[] = {EA1_AV1_MODE_CDFS(EA1_AV1_MODE_SECTION)};
const int kNumModeSections = int(sizeof kModeSections / sizeof kModeSections[0]);
const CdfSection kCoefSections[] = {EA1_AV1_COEF_CDFS(EA1_AV1_COEF_SECTION)};
const int kNumCoefSections = int(sizeof kCoefSections / sizeof kCoefSections[0]);

namespace {
struct ScanShape {
    int wl2, hl2;
    bool one_d;   // has Mrow / Mcol variants (both sides at most 16)
};
// order: the transform size numbering of easyav1_amd_video.h without the sizes that have a 64-sample side
const ScanShape kScanShapes[] = {{2, 2, true},  {3, 3, true},  {4, 4, true},  {5, 5, false}, {2, 3, true},  {3, 2, true},  {3, 4, true},
                                 {4, 3, true},  {4, 5, false}, {5, 4, false}, {2, 4, true},  {4, 2, true},  {3, 5, false}, {5, 3, false}};
std::vector<ScanSection> make_scan_sections()
{
    static std::vector<std::string> names;
    std::vector<ScanSection> out;
    size_t off = 0;
    static const char *const kind_name[3] = {"Default_Scan", "Mrow_Scan", "Mcol_Scan"};
    names.reserve(64);
    for (int kind = 0; kind < 3; kind++)
        for (const ScanShape &s : kScanShapes) {
            if (kind && !s.one_d) continue;
            names.push_back(std::string(kind_name[kind]) + "_" + std::to_string(1 << s.wl2) + "x" + std::to_string(1 << s.hl2));
            out.push_back({nullptr, kind, s.wl2, s.hl2, off});
            off += size_t(1) << (s.wl2 + s.hl2);
        }
    for (size_t i = 0; i < out.size(); i++) out[i].name = names[i].c_str();
    return out;
}
const std::vector<ScanSection> &scan_sections()
{
    static const std::vector<ScanSection> s = make_scan_sections();
    return s;
}
}  // namespace
const ScanSection *scan_section_table() { return scan_sections().data(); }
int scan_section_count() { return int(scan_sections().size()); }

static void reset(void *base, const CdfSection *sec, int nsec)
{
    for (int s = 0; s < nsec; s++) {
        uint16_t *p = reinterpret_cast<uint16_t *>(static_cast<char *>(base) + sec[s].offset);
        for (int i = 0; i < sec[s].count; i++) p[size_t(i) * (sec[s].symbols + 1) + sec[s].symbols] = 0;
    }
}
void reset_counters(Cdfs *c)
{
    reset(&c->m, kModeSections, kNumModeSections);
    reset(&c->c, kCoefSections, kNumCoefSections);
}

}  // namespace av1
}  // namespace ea1

using namespace ea1::av1;

struct ea1_av1_tables {
    Tables t;
};

namespace {
constexpr uint32_t kMagic = 0x54314145u;   // "EA1T"
constexpr uint32_t kVersion = 1;
constexpr size_t kHeaderU16 = 8;           // magic, version, total u16 count, reserved

struct Flat {   // one section of the blob in file order
    const char *name;
    int kind;         // 0 CDFs, 1 scan, 2 quantiser look-up
    size_t offset;    // in uint16 units from the start of the blob
    size_t count;     // uint16 units
    int symbols, w, h;
    int where;        // CDFs: -1 = mode, 0..3 = coefficient class; scan: index into the scan sections; q: 0 dc, 1 ac
    int index;        // section index inside its table
};
std::vector<Flat> make_layout()
{
    std::vector<Flat> v;
    size_t off = kHeaderU16;
    for (int s = 0; s < kNumModeSections; s++) {
        const size_t n = size_t(kModeSections[s].count) * kModeSections[s].symbols;
        v.push_back({kModeSections[s].name, 0, off, n, kModeSections[s].symbols, 0, 0, -1, s});
        off += n;
    }
    static std::vector<std::string> names;
    names.reserve(size_t(4 * kNumCoefSections));
    for (int q = 0; q < 4; q++)
        for (int s = 0; s < kNumCoefSections; s++) {
            const size_t n = size_t(kCoefSections[s].count) * kCoefSections[s].symbols;
            names.push_back(std::string(kCoefSections[s].name) + "_q" + std::to_string(q));
            v.push_back({nullptr, 0, off, n, kCoefSections[s].symbols, 0, 0, q, s});
            off += n;
        }
    size_t k = 0;
    for (Flat &f : v)
        if (!f.name) f.name = names[k++].c_str();
    for (int s = 0; s < scan_section_count(); s++) {
        const ScanSection &sc = scan_section_table()[s];
        const size_t n = size_t(1) << (sc.wl2 + sc.hl2);
        v.push_back({sc.name, 1, off, n, 0, 1 << sc.wl2, 1 << sc.hl2, s, s});
        off += n;
    }
    v.push_back({"Dc_Qlookup_8bit", 2, off, 256, 0, 0, 0, 0, 0});
    off += 256;
    v.push_back({"Ac_Qlookup_8bit", 2, off, 256, 0, 0, 0, 1, 0});
    off += 256;
    {
        static std::vector<std::string> qnames;
        qnames.reserve(15 * 2 * 14);
        const size_t first = v.size();
        for (int level = 0; level < 15; level++)
            for (int pl = 0; pl < 2; pl++) {
                size_t at = 0;
                for (const ScanShape &sh : kScanShapes) {
                    const size_t n = size_t(1) << (sh.wl2 + sh.hl2);
                    qnames.push_back("Quantizer_Matrix_L" + std::to_string(level) + (pl ? "_chroma_" : "_luma_") + std::to_string(1 << sh.wl2) + "x" + std::to_string(1 << sh.hl2));
                    v.push_back({nullptr, 4, off, n, 0, 1 << sh.wl2, 1 << sh.hl2, level * 2 + pl, int(at)});
                    off += n;
                    at += n;
                }
            }
        for (size_t i = first; i < v.size(); i++) v[i].name = qnames[i - first].c_str();
    }
    v.push_back({"Gaussian_Sequence", 3, off, 2048, 0, 0, 0, 0, 0});   // int16 values, two's complement in the uint16 units
    off += 2048;
    v.push_back({"Upscale_Filter", 5, off, 64 * 8, 0, 8, 64, 0, 0});   // [64 phases][8 taps], int16 values like the Gaussian sequence
    off += 64 * 8;
    return v;
}
const std::vector<Flat> &layout()
{
    static const std::vector<Flat> v = make_layout();
    return v;
}
size_t blob_u16() { return layout().back().offset + layout().back().count; }
}  // namespace

extern "C" size_t ea1_av1_tables_blob_size(void) { return blob_u16() * 2; }
extern "C" int ea1_av1_tables_num_sections(void) { return int(layout().size()); }
extern "C" int ea1_av1_tables_section(int index, ea1_av1_tables_section_info *out)
{
    if (index < 0 || index >= int(layout().size()) || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_tables_section: index %d", index);
    const Flat &f = layout()[size_t(index)];
    out->name = f.name;
    out->kind = f.kind;
    out->offset = f.offset * 2;
    out->count = uint32_t(f.count);
    out->symbols = f.symbols;
    out->width = f.w;
    out->height = f.h;
    return EA1_OK;
}
extern "C" int ea1_av1_tables_blob_header(void *blob, size_t size)
{
    if (!blob || size < kHeaderU16 * 2) return ea1::fail(EA1_EINVAL, "ea1_av1_tables_blob_header: buffer of %zu bytes", size);
    uint32_t h[4] = {kMagic, kVersion, uint32_t(blob_u16()), 0};
    memcpy(blob, h, sizeof h);
    return EA1_OK;
}

extern "C" int ea1_av1_tables_load(const void *blob, size_t size, ea1_av1_tables **out)
{
    if (!blob || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: NULL argument");
    if (size != blob_u16() * 2) return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: %zu bytes, the layout of this library has %zu", size, blob_u16() * 2);
    uint32_t h[4];
    memcpy(h, blob, sizeof h);
    if (h[0] != kMagic || h[1] != kVersion || h[2] != blob_u16()) return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: not a table blob of version %u", kVersion);
    const uint16_t *src = static_cast<const uint16_t *>(blob);   // the caller's buffer may be unaligned: copy out below with memcpy
    ea1_av1_tables *t = new (std::nothrow) ea1_av1_tables;
    if (!t) return ea1::fail(EA1_ENOMEM, "ea1_av1_tables_load: out of memory");
    memset(&t->t, 0, sizeof t->t);
    std::vector<uint16_t> v;
    for (const Flat &f : layout()) {
        v.resize(f.count);
        memcpy(v.data(), reinterpret_cast<const char *>(src) + f.offset * 2, f.count * 2);
        if (f.kind == 0) {
            const CdfSection &sec = f.where < 0 ? kModeSections[f.index] : kCoefSections[f.index];
            char *base = f.where < 0 ? reinterpret_cast<char *>(&t->t.mode) : reinterpret_cast<char *>(&t->t.coef[f.where]);
            uint16_t *dst = reinterpret_cast<uint16_t *>(base + sec.offset);
            for (int i = 0; i < sec.count; i++) {
                const uint16_t *c = v.data() + size_t(i) * sec.symbols;
                for (int k = 0; k < sec.symbols; k++) {
                    // every symbol must keep a codable share: the decoder's partition floors probabilities at 4 / 32768 per symbol,
                    // but a CDF that DEcreases is a broken table
                    if ((k && c[k] < c[k - 1]) || c[k] > 32768 || (k == sec.symbols - 1 && c[k] != 32768)) {
                        delete t;
                        return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: %s[%d] is not a cumulative distribution ending at 32768", f.name, i);
                    }
                    dst[size_t(i) * (sec.symbols + 1) + k] = c[k];
                }
                dst[size_t(i) * (sec.symbols + 1) + sec.symbols] = 0;
            }
        } else if (f.kind == 1) {
            const ScanSection &sc = scan_section_table()[f.where];
            std::vector<uint8_t> seen(f.count, 0);
            for (size_t i = 0; i < f.count; i++) {
                if (v[i] >= f.count || seen[v[i]]) {
                    delete t;
                    return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: %s is not a permutation of its %zu positions", f.name, f.count);
                }
                seen[v[i]] = 1;
            }
            memcpy(t->t.scans + sc.offset, v.data(), f.count * 2);
            const int idx = ea1_tx_size_index(sc.wl2, sc.hl2);
            t->t.scan[sc.kind][idx] = t->t.scans + sc.offset;
        } else if (f.kind == 4) {
            // one size of one quantiser matrix: 8-bit weights (32 = neutral), `where` = level * 2 + plane type, `index` = offset in it
            uint8_t *dst = &t->t.qmatrix[f.where >> 1][f.where & 1][f.index];
            for (size_t i = 0; i < f.count; i++) {
                if (v[i] == 0 || v[i] > 255) {
                    delete t;
                    return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: %s[%zu] = %u (1..255)", f.name, i, v[i]);
                }
                dst[i] = uint8_t(v[i]);
            }
            int wl = 0, hl = 0;
            while ((1 << wl) < f.w) wl++;
            while ((1 << hl) < f.h) hl++;
            t->t.qm[f.where >> 1][f.where & 1][ea1_tx_size_index(wl, hl)] = dst;
        } else if (f.kind == 3) {
            memcpy(t->t.gaussian, v.data(), 2048 * 2);
        } else if (f.kind == 5) {
            // the super-resolution taps (7.16): 7-bit filters -- every row must sum to 128 or be all zero (a blob without the table)
            for (int ph = 0; ph < 64; ph++) {
                int sum = 0;
                bool any = false;
                for (int k = 0; k < 8; k++) {
                    const int16_t tap = int16_t(v[size_t(ph) * 8 + k]);
                    if (tap < -128 || tap > 255) {
                        delete t;
                        return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: Upscale_Filter[%d][%d] = %d", ph, k, tap);
                    }
                    t->t.upscale[ph][k] = tap;
                    sum += tap;
                    any |= tap != 0;
                }
                if (any && sum != 128) {
                    delete t;
                    return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: Upscale_Filter[%d] sums to %d, not 128", ph, sum);
                }
            }
        } else {
            int16_t *dst = f.where ? t->t.ac_q : t->t.dc_q;
            for (int i = 0; i < 256; i++) {
                if (v[size_t(i)] == 0 || v[size_t(i)] > 0x7FFF) {
                    delete t;
                    return ea1::fail(EA1_EINVAL, "ea1_av1_tables_load: %s[%d] = %u", f.name, i, v[size_t(i)]);
                }
                dst[i] = int16_t(v[size_t(i)]);
            }
        }
    }
    for (int level = 0; level < 15; level++)
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t **q = t->t.qm[level][pl];
            q[ea1_tx_size_index(6, 6)] = q[ea1_tx_size_index(6, 5)] = q[ea1_tx_size_index(5, 6)] = q[ea1_tx_size_index(5, 5)];
            q[ea1_tx_size_index(4, 6)] = q[ea1_tx_size_index(4, 5)];
            q[ea1_tx_size_index(6, 4)] = q[ea1_tx_size_index(5, 4)];
        }
    *out = t;
    return EA1_OK;
}
extern "C" void ea1_av1_tables_destroy(ea1_av1_tables *t) { delete t; }
extern "C" const int16_t *ea1_av1_tables_gaussian_sequence(const ea1_av1_tables *t) { return t ? t->t.gaussian : nullptr; }
extern "C" const int16_t *ea1_av1_tables_upscale_filter(const ea1_av1_tables *t) { return t ? &t->t.upscale[0][0] : nullptr; }

namespace ea1 {
namespace av1 {
const Tables *tables_of(const ea1_av1_tables *t) { return t ? &t->t : nullptr; }
}  // namespace av1
}  // namespace ea1
