// Mutation fuzzer for the host-side parsers of untrusted input (OBU / sequence header / frame header /
// tile group, WebM reader, Vorbis header + packet front end), built with AddressSanitizer +
// UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on this pool):
//   python tools/fuzz/make_corpus.py /tmp/ea1_corpus && bash tools/fuzz/run.sh /tmp/ea1_corpus 2000
// Every corpus file is parsed as is and then `iterations` times with random byte flips, truncations,
// insertions and splices.  Any sanitizer report or crash fails the run; return codes are free.
#include <dirent.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "easyav1_amd_obu.h"
#include "easyav1_amd_vorbis.h"
#include "easyav1_amd_webm.h"

namespace {

uint64_t rng_state = 0x1234567;
uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return uint32_t(rng_state >> 16);
}

std::vector<std::string> list_dir(const std::string &d)
{
    std::vector<std::string> out;
    if (DIR *dir = opendir(d.c_str())) {
        while (dirent *e = readdir(dir))
            if (e->d_name[0] != '.') out.push_back(d + "/" + e->d_name);
        closedir(dir);
    }
    return out;
}

std::vector<uint8_t> read_file(const std::string &p)
{
    std::vector<uint8_t> v;
    if (FILE *f = fopen(p.c_str(), "rb")) {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        fclose(f);
    }
    return v;
}

void mutate(std::vector<uint8_t> &v)
{
    const int kinds = 1 + int(rnd() % 4);
    for (int k = 0; k < kinds && !v.empty(); k++) {
        switch (rnd() % 5) {
        case 0: v[rnd() % v.size()] = uint8_t(rnd()); break;
        case 1: v[rnd() % v.size()] ^= uint8_t(1u << (rnd() % 8)); break;
        case 2: v.resize(rnd() % (v.size() + 1)); break;
        case 3: v.insert(v.begin() + rnd() % (v.size() + 1), uint8_t(rnd() % 3 ? rnd() : 0xFF)); break;
        default: {
            const size_t a = rnd() % v.size(), n = rnd() % 9;
            for (size_t i = 0; i < n && a + i < v.size(); i++) v[a + i] = uint8_t(rnd() % 2 ? 0xFF : 0x00);
        }
        }
    }
}

uint32_t rd32(const std::vector<uint8_t> &v, size_t &pos)
{
    uint32_t x = 0;
    if (pos + 4 <= v.size()) memcpy(&x, &v[pos], 4);
    pos += 4;
    return x;
}

void run_seq(std::vector<uint8_t> d)
{
    ea1_av1_sequence_header h;
    (void)ea1_av1_parse_sequence_header(d.data(), d.size(), &h);
    {   // the frame assembler over the same bytes (temporal unit: delimiter, sequence header, OBU_FRAME with junk)
        ea1_av1_parser *p = nullptr;
        if (ea1_av1_parser_create(&p) == EA1_OK) {
            static ea1_av1_tile tiles[4096];
            ea1_av1_frame_header fh;
            size_t pos = 0;
            for (int guard = 0; guard < 64 && pos < d.size(); guard++) {
                size_t used = 0;
                int count = 0;
                const int rc = ea1_av1_parser_next_frame(p, d.data() + pos, d.size() - pos, &used, &fh, tiles, 4096, &count);
                if (rc == EA1_OK) {
                    volatile uint8_t sink = 0;
                    for (int t = 0; t < count; t++)
                        for (size_t i = 0; i < tiles[t].size; i++) sink ^= d[pos + tiles[t].offset + i];
                    (void)ea1_av1_parser_end_frame(p, &fh);
                }
                if (rc != EA1_OK || !used) break;
                pos += used;
            }
            ea1_av1_parser_destroy(p);
        }
    }
    size_t pos = 0;
    while (pos < d.size()) {
        ea1_obu o;
        if (ea1_av1_next_obu(d.data() + pos, d.size() - pos, &o) != EA1_OK || !o.total_size) break;
        volatile uint8_t sink = 0;
        for (size_t i = 0; i < o.payload_size; i++) sink ^= o.payload[i];   // the payload range must be readable
        pos += o.total_size;
    }
}

// file: u32 seq size, seq struct, u32 frame count, {u32 size, payload + tile bytes}...
void run_frames(const std::vector<uint8_t> &file, bool fuzz)
{
    size_t pos = 0;
    const uint32_t ss = rd32(file, pos);
    if (ss != sizeof(ea1_av1_sequence_header) || pos + ss > file.size()) return;
    ea1_av1_sequence_header seq;
    memcpy(&seq, &file[pos], ss);
    pos += ss;
    const uint32_t nframes = rd32(file, pos);
    ea1_av1_parser *p = nullptr;
    if (ea1_av1_parser_create(&p) != EA1_OK) return;
    if (ea1_av1_parser_set_sequence_header(p, &seq) == EA1_OK) {
        for (uint32_t f = 0; f < nframes && pos < file.size(); f++) {
            const uint32_t n = rd32(file, pos);
            if (pos + n > file.size()) break;
            std::vector<uint8_t> payload(file.begin() + long(pos), file.begin() + long(pos + n));
            pos += n;
            if (fuzz && rnd() % 2) mutate(payload);
            ea1_av1_frame_header h;
            if (payload.empty()) continue;
            if (ea1_av1_parser_frame_header(p, payload.data(), payload.size(), int(rnd() % 8), int(rnd() % 4), &h) == EA1_OK) {
                if (!h.show_existing_frame) {
                    const size_t off = (h.header_bits + 7) / 8;
                    if (off <= payload.size()) {
                        std::vector<ea1_av1_tile> tiles(4096);
                        int count = 0;
                        if (ea1_av1_parse_tile_group(&h, payload.data() + off, payload.size() - off, tiles.data(), int(tiles.size()), &count) == EA1_OK) {
                            volatile uint8_t sink = 0;
                            for (int t = 0; t < count; t++)
                                for (size_t i = 0; i < tiles[size_t(t)].size; i++) sink ^= payload[off + tiles[size_t(t)].offset + i];
                        }
                    }
                    ea1_frame_desc desc;
                    ea1_film_grain fg;
                    (void)ea1_av1_frame_header_to_desc(&h, &seq, &desc);
                    (void)ea1_av1_frame_header_film_grain(&h, &seq, &fg);
                }
                (void)ea1_av1_parser_end_frame(p, &h);
            }
        }
    }
    ea1_av1_parser_destroy(p);
}

void run_webm(const std::vector<uint8_t> &d)
{
    ea1_webm *w = nullptr;
    if (d.empty() || ea1_webm_open_memory(&w, d.data(), d.size()) != EA1_OK) return;
    uint32_t ntracks = 0;
    (void)ea1_webm_track_count(w, &ntracks);
    for (uint32_t t = 0; t < ntracks; t++) {
        ea1_webm_track tr;
        if (ea1_webm_track_info(w, t, &tr) != EA1_OK) continue;
        for (uint32_t i = 0; i < tr.codec_data_count; i++) {
            const uint8_t *cd = nullptr;
            size_t cs = 0;
            if (ea1_webm_codec_data(w, t, i, &cd, &cs) == EA1_OK) {
                volatile uint8_t sink = 0;
                for (size_t k = 0; k < cs; k++) sink ^= cd[k];
            }
        }
    }
    uint64_t v = 0;
    (void)ea1_webm_duration(w, &v);
    (void)ea1_webm_tstamp_scale(w, &v);
    static ea1_webm_packet pk;
    for (int guard = 0; guard < 100000; guard++) {
        if (ea1_webm_read_packet(w, &pk) <= 0) break;
        volatile uint8_t sink = 0;
        for (uint32_t f = 0; f < pk.count; f++)
            for (size_t k = 0; k < pk.size[f]; k++) sink ^= pk.data[f][k];
    }
    (void)ea1_webm_rewind(w);
    ea1_webm_destroy(w);
}

// file: u32 item count, {u32 size, bytes}: three headers, then audio packets
void run_vorbis(const std::vector<uint8_t> &file, bool fuzz)
{
    size_t pos = 0;
    const uint32_t n = rd32(file, pos);
    std::vector<std::vector<uint8_t>> items;
    for (uint32_t i = 0; i < n && pos < file.size(); i++) {
        const uint32_t sz = rd32(file, pos);
        if (pos + sz > file.size()) break;
        items.emplace_back(file.begin() + long(pos), file.begin() + long(pos + sz));
        pos += sz;
    }
    if (items.size() < 3) return;
    if (fuzz) {
        const size_t which = rnd() % items.size();
        mutate(items[which]);
        if (rnd() % 4 == 0) mutate(items[rnd() % 3]);
    }
    for (auto &it : items)
        if (it.empty()) it.push_back(0);
    ea1_vorbis_decoder *d = nullptr;
    if (ea1_vorbis_decoder_create(&d, items[0].data(), items[0].size(), items[1].data(), items[1].size(), items[2].data(), items[2].size(), -1) != EA1_OK)
        return;
    int ch = 0, b0 = 0, b1 = 0;
    long rate = 0;
    if (ea1_vorbis_decoder_info(d, &ch, &rate, &b0, &b1) == EA1_OK && ch > 0 && b1 > 0) {
        std::vector<float> spectra(size_t(ch) * size_t(b1 / 2));
        for (size_t i = 3; i < items.size(); i++) {
            int long_block = 0;
            (void)ea1_vorbis_decoder_spectra(d, items[i].data(), items[i].size(), spectra.data(), size_t(b1 / 2), &long_block);
        }
    }
    ea1_vorbis_decoder_destroy(d);
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s corpus_dir [iterations per file]\n", argv[0]);
        return 2;
    }
    const std::string root = argv[1];
    const int iters = argc > 2 ? atoi(argv[2]) : 500;
    size_t runs = 0;
    for (const std::string &f : list_dir(root + "/seq")) {
        const std::vector<uint8_t> d = read_file(f);
        run_seq(d);
        for (int i = 0; i < iters; i++, runs++) {
            std::vector<uint8_t> m = d;
            mutate(m);
            if (!m.empty()) run_seq(m);
        }
    }
    for (const std::string &f : list_dir(root + "/frame")) {
        const std::vector<uint8_t> d = read_file(f);
        run_frames(d, false);
        for (int i = 0; i < iters; i++, runs++) run_frames(d, true);
    }
    for (const std::string &f : list_dir(root + "/webm")) {
        const std::vector<uint8_t> d = read_file(f);
        run_webm(d);
        for (int i = 0; i < iters; i++, runs++) {
            std::vector<uint8_t> m = d;
            mutate(m);
            run_webm(m);
        }
    }
    for (const std::string &f : list_dir(root + "/vorbis")) {
        const std::vector<uint8_t> d = read_file(f);
        run_vorbis(d, false);
        for (int i = 0; i < iters; i++, runs++) run_vorbis(d, true);
    }
    printf("fuzz_parsers: %zu mutated inputs, no sanitizer report\n", runs);
    return 0;
}
