// Mutation fuzzer for the block-level AV1 parser (easyav1_amd/csrc/av1_tile_syntax.h and its frame driver): the one host routine that
// walks megabytes of untrusted, entropy-coded bytes.  ASan + UBSan on the CPU:
//   python tools/fuzz/make_av1_corpus.py /tmp/ea1_av1_corpus && bash tools/fuzz/run.sh /tmp/ea1_av1_corpus 300 av1
// A corpus file: u32 size + the table blob, then temporal units as u32 size + bytes.  Every stream is parsed as is (must succeed) and
// then `iterations` times with mutations of one or more of its units -- header bytes and tile bytes alike; frames after a mutated
// one run on whatever state it left.  Any sanitizer report or crash fails the run; return codes are free.
#include <dirent.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "easyav1_amd_obu.h"

namespace {
uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return uint32_t(rng_state >> 16);
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
        switch (rnd() % 6) {
        case 0: v[rnd() % v.size()] = uint8_t(rnd()); break;
        case 1: v[rnd() % v.size()] ^= uint8_t(1u << (rnd() % 8)); break;
        case 2: v.resize(rnd() % (v.size() + 1)); break;
        case 3: v.insert(v.begin() + rnd() % (v.size() + 1), uint8_t(rnd() % 3 ? rnd() : 0xFF)); break;
        case 4: {   // a burst inside the second half: tile data
            const size_t a = v.size() / 2 + rnd() % (v.size() - v.size() / 2), n = 1 + rnd() % 16;
            for (size_t i = 0; i < n && a + i < v.size(); i++) v[a + i] = uint8_t(rnd());
            break;
        }
        default: {
            const size_t a = rnd() % v.size(), n = rnd() % 9;
            for (size_t i = 0; i < n && a + i < v.size(); i++) v[a + i] = uint8_t(rnd() % 2 ? 0xFF : 0x00);
        }
        }
    }
}
struct Stream {
    std::vector<uint8_t> blob;
    std::vector<std::vector<uint8_t>> units;
};
bool load(const std::string &path, Stream *s)
{
    const std::vector<uint8_t> d = read_file(path);
    size_t pos = 0;
    auto take = [&](std::vector<uint8_t> *out) {
        if (pos + 4 > d.size()) return false;
        uint32_t n;
        memcpy(&n, &d[pos], 4);
        pos += 4;
        if (pos + n > d.size()) return false;
        out->assign(d.begin() + long(pos), d.begin() + long(pos + n));
        pos += n;
        return true;
    };
    if (!take(&s->blob)) return false;
    std::vector<uint8_t> u;
    while (take(&u)) s->units.push_back(u);
    return !s->units.empty();
}
// parse every frame of every unit; returns the number of frames that parsed
int run(const ea1_av1_tables *t, const std::vector<std::vector<uint8_t>> &units)
{
    ea1_av1_frame_parser *p = nullptr;
    if (ea1_av1_frame_parser_create(t, &p) != EA1_OK) abort();
    int ok = 0;
    for (const auto &u : units) {
        size_t pos = 0;
        for (int guard = 0; pos < u.size() && guard < 64; guard++) {
            ea1_av1_parsed out;
            const int rc = ea1_av1_frame_parser_parse(p, u.data() + pos, u.size() - pos, &out);
            if (rc != EA1_OK) break;
            ok++;
            // touch what came out: every array must be as long as the description says
            uint64_t sum = 0;
            for (uint32_t i = 0; i < out.frame.desc.num_blocks; i++) sum += out.frame.blocks[i].x4;
            for (uint32_t i = 0; i < out.frame.desc.num_tx; i++) sum += out.frame.txs[i].x4;
            for (uint32_t i = 0; i < out.frame.desc.num_coefs; i++) sum += uint16_t(out.frame.coefs[i]);
            for (uint32_t i = 0; i < out.frame.desc.num_aux; i++) sum += out.frame.aux[i];
            for (uint32_t i = 0; i < out.frame.num_sb; i++) sum += out.frame.cdef_idx[i];
            for (int pl = 0; pl < 3; pl++)
                for (uint32_t i = 0; i < out.frame.lr_count[pl]; i++) sum += out.frame.lr[pl][i].type;
            if (sum == 0x123456789ull) puts("");
            if (!out.consumed) break;
            pos += out.consumed;
        }
    }
    ea1_av1_frame_parser_destroy(p);
    return ok;
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int iterations = argc > 2 ? atoi(argv[2]) : 200;
    std::vector<std::string> files;
    if (DIR *dir = opendir(argv[1])) {
        while (dirent *e = readdir(dir))
            if (e->d_name[0] != '.') files.push_back(std::string(argv[1]) + "/" + e->d_name);
        closedir(dir);
    }
    long long total = 0, parsed = 0;
    for (const std::string &path : files) {
        Stream s;
        if (!load(path, &s)) continue;
        ea1_av1_tables *t = nullptr;
        if (ea1_av1_tables_load(s.blob.data(), s.blob.size(), &t) != EA1_OK) {
            fprintf(stderr, "%s: bad table blob: %s\n", path.c_str(), ea1_last_error());
            return 1;
        }
        int frames = 0;
        for (const auto &u : s.units) frames += 1;
        if (run(t, s.units) < frames) {
            fprintf(stderr, "%s: the unmutated stream does not parse: %s\n", path.c_str(), ea1_last_error());
            return 1;
        }
        for (int it = 0; it < iterations; it++) {
            std::vector<std::vector<uint8_t>> m = s.units;
            const int nmut = 1 + int(rnd() % 2);
            for (int k = 0; k < nmut; k++) mutate(m[rnd() % m.size()]);
            parsed += run(t, m);
            total++;
        }
        // a broken table blob must be refused or harmless, too
        for (int it = 0; it < 20; it++) {
            std::vector<uint8_t> b = s.blob;
            b[16 + rnd() % (b.size() - 16)] ^= uint8_t(1u << (rnd() % 8));
            ea1_av1_tables *t2 = nullptr;
            if (ea1_av1_tables_load(b.data(), b.size(), &t2) == EA1_OK) {
                run(t2, s.units);
                ea1_av1_tables_destroy(t2);
            }
        }
        ea1_av1_tables_destroy(t);
    }
    printf("%zu streams, %lld mutated runs, %lld frames parsed in them\n", files.size(), total, parsed);
    return 0;
}
