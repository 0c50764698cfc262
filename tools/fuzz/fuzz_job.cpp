// Mutation fuzzer for the host part of ea1_job_create (validation + schedule derivation: the code that
// stands between an untrusted parsed frame and the GPU kernels), AddressSanitizer + UBSan on the CPU:
//   bash tools/fuzz/run.sh <corpus dir> <iterations> job
// Seeds are frames of the library's own synthetic generator (every tool incl. warped motion / intra
// block copy / crop); each iteration corrupts one field class of a copy, as tests/test_video_gpu.py does
// on the GPU.  Without a device the call ends in EA1_ENOMEM / EA1_EHIP after all host phases have run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "easyav1_amd_synth.h"
#include "easyav1_amd_video.h"

extern "C" int ea1_fuzz_job_create(uint32_t width, uint32_t height, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                   const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3]);
extern "C" int ea1_fuzz_job_create_layout(uint32_t width, uint32_t height, int layout, const ea1_frame_desc *desc, const ea1_block *blocks, const ea1_txblock *txs,
                                          const int16_t *coefs, const uint8_t *aux, const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3]);
#include "easyav1_amd_ir_packet.h"
extern "C" int ea1_fuzz_job_prepare_layout(uint32_t width, uint32_t height, int layout, const ea1_ir_frame *frame);

// with -DEA1_FUZZ_ORACLE the frames that pass validation also go through the CPU oracle (same sanitizers): whatever the
// library accepts, the checker must be able to decode
#ifdef EA1_FUZZ_ORACLE
extern "C" int ea1o_decode_frame(const ea1_frame_desc *d, const ea1_block *blocks, const ea1_txblock *txs, const int16_t *coefs, const uint8_t *aux,
                                 const uint8_t *cdef_idx, const ea1_lr_unit *const lr[3], uint8_t *const refs[EA1_NUM_REF_SLOTS][3], unsigned stages,
                                 uint8_t *const out_recon[3], uint8_t *const out_deblock[3], uint8_t *const out_cdef[3], uint8_t *const out_final[3]);
extern "C" void ea1o_set_warp_filter(const int8_t *f);
extern "C" void ea1o_set_chroma_format(int ssx, int ssy);
#endif

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return uint32_t(rng_state >> 16);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    struct Cfg {
        uint32_t w, h, tiles, features, frame;
        uint8_t cr, cb;
    };
    const Cfg cfgs[] = {{192, 128, 1, 0x7FFF | 0x10000, 1, 0, 0}, {192, 128, 1, 0x7FFF | 0x20000, 0, 0, 0}, {200, 120, 2, 0x7FFF | 0x10000, 2, 3, 5},
                        {64, 64, 1, 0x7FFF, 1, 7, 7},           {320, 192, 2, 0x8000, 0, 0, 0},          {448, 256, 4, 0x7FFF | 0x30000, 3, 1, 2},
                        // the packed layouts: corner, diagonal (lossless 4x4 blocks too)
                        {192, 128, 1, 0x7FFF | 0x80000, 1, 0, 0},  {192, 128, 2, 0x7FFF | 0x100000, 1, 0, 0}, {128, 64, 1, 0x8000 | 0x100000, 0, 0, 0}};
    size_t runs = 0, accepted_host = 0;
#ifdef EA1_FUZZ_ORACLE
    {
        int8_t wf[193 * 8];
        ea1_synth_warp_filter(wf);
        ea1o_set_warp_filter(wf);
    }
#endif
    for (const Cfg &c : cfgs) {
        ea1_synth_params p;
        ea1_synth_default_params(&p, c.w, c.h, c.tiles, 5);
        p.features = c.features;
        p.gop = 4;
        p.crop_right = c.cr;
        p.crop_bottom = c.cb;
        ea1_synth_frame *f = nullptr;
        if (ea1_synth_frame_create(&p, c.frame, &f) != EA1_OK) return 3;
        const ea1_frame_desc &d0 = f->desc;
        for (int it = 0; it <= iters; it++, runs++) {
            ea1_frame_desc d = d0;
            std::vector<ea1_block> blocks(f->blocks, f->blocks + d0.num_blocks);
            std::vector<ea1_txblock> txs(f->txs, f->txs + d0.num_tx);
            std::vector<int16_t> coefs(f->coefs, f->coefs + d0.num_coefs);
            std::vector<uint8_t> aux(f->aux, f->aux + d0.num_aux);
            std::vector<uint8_t> cdef(f->cdef_idx, f->cdef_idx + f->num_sb);
            std::vector<ea1_lr_unit> lr[3];
            for (int pl = 0; pl < 3; pl++) lr[pl].assign(f->lr[pl], f->lr[pl] + f->lr_count[pl]);
            if (it) {   // iteration 0: the valid frame
                const int nmut = 1 + int(rnd() % 3);
                for (int m = 0; m < nmut; m++) switch (rnd() % 6) {
                    case 0: if (!blocks.empty()) reinterpret_cast<uint8_t *>(blocks.data())[rnd() % (blocks.size() * sizeof(ea1_block))] = uint8_t(rnd()); break;
                    case 1: if (!txs.empty()) reinterpret_cast<uint8_t *>(txs.data())[rnd() % (txs.size() * sizeof(ea1_txblock))] = uint8_t(rnd()); break;
                    case 2: for (int k = 0; k < 64 && !coefs.empty(); k++) coefs[rnd() % coefs.size()] = (rnd() & 1) ? 32767 : -32768; break;
                    case 3: for (int k = 0; k < 16 && !aux.empty(); k++) aux[rnd() % aux.size()] = uint8_t(rnd()); break;
                    case 4:
                        if (!cdef.empty()) cdef[rnd() % cdef.size()] = uint8_t(rnd());
                        for (int pl = 0; pl < 3; pl++)
                            if (!lr[pl].empty()) reinterpret_cast<uint8_t *>(lr[pl].data())[rnd() % (lr[pl].size() * sizeof(ea1_lr_unit))] = uint8_t(rnd());
                        break;
                    default: reinterpret_cast<uint8_t *>(&d)[8 + rnd() % 36] = uint8_t(rnd()); break;   // header bytes, not the sizes / counts
                    }
            }
            const ea1_lr_unit *lrp[3] = {lr[0].empty() ? nullptr : lr[0].data(), lr[1].empty() ? nullptr : lr[1].data(), lr[2].empty() ? nullptr : lr[2].data()};
            const int rc = ea1_fuzz_job_create(c.w, c.h, &d, blocks.data(), txs.data(), coefs.empty() ? nullptr : coefs.data(), aux.empty() ? nullptr : aux.data(),
                                               cdef.data(), lrp);
            accepted_host += rc != EA1_EINVAL;
#ifdef EA1_FUZZ_ORACLE
            if (rc != EA1_EINVAL && d.dst_slot < 8) {
                static std::vector<uint8_t> ry, ru, rv, oy, ou, ov;
                ry.assign(size_t(c.w) * c.h, 100);
                ru.assign(size_t(c.w) * c.h / 4, 120);
                rv.assign(size_t(c.w) * c.h / 4, 140);
                oy.resize(ry.size());
                ou.resize(ru.size());
                ov.resize(rv.size());
                uint8_t *refs[EA1_NUM_REF_SLOTS][3];
                for (auto &r : refs) {
                    r[0] = ry.data();
                    r[1] = ru.data();
                    r[2] = rv.data();
                }
                uint8_t *out[3] = {oy.data(), ou.data(), ov.data()};
                (void)ea1o_decode_frame(&d, blocks.data(), txs.data(), coefs.empty() ? nullptr : coefs.data(), aux.empty() ? nullptr : aux.data(), cdef.data(), lrp, refs,
                                        EA1_STAGE_ALL, nullptr, nullptr, nullptr, out);
            }
#endif
            if (!it && rc == EA1_EINVAL) {
                fprintf(stderr, "the valid seed frame was rejected: %s\n", ea1_last_error());
                return 4;
            }
        }
        ea1_synth_frame_destroy(f);
    }
    // ---- the other chroma layouts (round 5): parsed frames of generated 4:4:4 / 4:2:2 streams as hand-over packets (tools/fuzz/make_layout_seeds.py), a context of the
    //      frame's own layout -- and, every eighth iteration, of another one (every record of the frame is then in the wrong geometry)
    size_t layout_runs = 0;
    if (argc > 2) {
        FILE *fp = fopen(argv[2], "rb");
        if (!fp) {
            fprintf(stderr, "cannot read %s\n", argv[2]);
            return 5;
        }
        std::vector<uint8_t> all;
        uint8_t buf[65536];
        for (size_t n; (n = fread(buf, 1, sizeof buf, fp)) > 0;) all.insert(all.end(), buf, buf + n);
        fclose(fp);
        for (size_t pos = 0; pos + 4 <= all.size();) {
            uint32_t len;
            memcpy(&len, &all[pos], 4);
            pos += 4;
            if (pos + len > all.size()) return 6;
            std::vector<uint8_t> pk(all.begin() + long(pos), all.begin() + long(pos + len));   // (own allocation: the parse hands out pointers into it)
            pos += len;
            ea1_ir_frame f;
            ea1_ir_storage *storage = nullptr;
            if (ea1_ir_packet_parse(pk.data(), pk.size(), &storage, &f) != EA1_OK) return 7;
            const ea1_frame_desc d0 = f.desc;
            for (int it = 0; it <= iters; it++, runs++, layout_runs++) {
                ea1_frame_desc d = d0;
                std::vector<ea1_block> blocks(f.blocks, f.blocks + d0.num_blocks);
                std::vector<ea1_txblock> txs(f.txs, f.txs + d0.num_tx);
                std::vector<int16_t> coefs(f.coefs, f.coefs + d0.num_coefs);
                std::vector<uint8_t> aux(f.aux, f.aux + d0.num_aux);
                std::vector<uint8_t> cdef(f.cdef_idx, f.cdef_idx + f.num_sb);
                std::vector<ea1_lr_unit> lr[3];
                for (int pl = 0; pl < 3; pl++) lr[pl].assign(f.lr[pl], f.lr[pl] + f.lr_count[pl]);
                int layout = d0.layout;
                if (it) {
                    const int nmut = 1 + int(rnd() % 3);
                    for (int m = 0; m < nmut; m++) switch (rnd() % 6) {
                        case 0: if (!blocks.empty()) reinterpret_cast<uint8_t *>(blocks.data())[rnd() % (blocks.size() * sizeof(ea1_block))] = uint8_t(rnd()); break;
                        case 1: if (!txs.empty()) reinterpret_cast<uint8_t *>(txs.data())[rnd() % (txs.size() * sizeof(ea1_txblock))] = uint8_t(rnd()); break;
                        case 2: for (int k = 0; k < 64 && !coefs.empty(); k++) coefs[rnd() % coefs.size()] = (rnd() & 1) ? 32767 : -32768; break;
                        case 3: for (int k = 0; k < 16 && !aux.empty(); k++) aux[rnd() % aux.size()] = uint8_t(rnd()); break;
                        case 4:
                            if (!cdef.empty()) cdef[rnd() % cdef.size()] = uint8_t(rnd());
                            for (int pl = 0; pl < 3; pl++)
                                if (!lr[pl].empty()) reinterpret_cast<uint8_t *>(lr[pl].data())[rnd() % (lr[pl].size() * sizeof(ea1_lr_unit))] = uint8_t(rnd());
                            break;
                        default: reinterpret_cast<uint8_t *>(&d)[8 + rnd() % 36] = uint8_t(rnd()); break;
                        }
                    if ((it & 7) == 0) {
                        layout = int(rnd() % 3);
                        d.layout = uint8_t(layout);
                    }
                }
                // (through ea1_job_prepare: a mutated header may ask for more restoration units / CDEF indices than the arrays hold -- the frame says how many it has)
                ea1_ir_frame mf = f;
                mf.desc = d;
                mf.blocks = blocks.data(), mf.txs = txs.data(), mf.coefs = coefs.empty() ? nullptr : coefs.data(), mf.aux = aux.empty() ? nullptr : aux.data();
                mf.cdef_idx = cdef.empty() ? nullptr : cdef.data();
                for (int pl = 0; pl < 3; pl++) mf.lr[pl] = lr[pl].empty() ? nullptr : lr[pl].data();
                const int rc = ea1_fuzz_job_prepare_layout(d0.width, d0.height, layout, &mf);
                accepted_host += rc != EA1_EINVAL;
#ifdef EA1_FUZZ_ORACLE
                if (rc != EA1_EINVAL && rc != EA1_ENOTSUP && d.dst_slot < 8 && !d.upscaled_width) {   // whatever the library accepts, the checker must be able to decode
                    const int sx = layout == EA1_LAYOUT_444 ? 0 : 1, sy = layout == EA1_LAYOUT_420 ? 1 : 0;
                    ea1o_set_chroma_format(sx, sy);
                    static std::vector<uint8_t> ry, ru, rv, oy, ou, ov;
                    ry.assign(size_t(d0.width) * d0.height, 100);
                    ru.assign(size_t(d0.width >> sx) * (d0.height >> sy), 120);
                    rv.assign(ru.size(), 140);
                    oy.resize(ry.size());
                    ou.resize(ru.size());
                    ov.resize(rv.size());
                    uint8_t *refs[EA1_NUM_REF_SLOTS][3];
                    for (auto &r : refs) {
                        r[0] = ry.data();
                        r[1] = ru.data();
                        r[2] = rv.data();
                    }
                    uint8_t *out[3] = {oy.data(), ou.data(), ov.data()};
                    (void)ea1o_decode_frame(&d, blocks.data(), txs.data(), mf.coefs, mf.aux, mf.cdef_idx, mf.lr, refs, EA1_STAGE_ALL, nullptr, nullptr, nullptr, out);
                    ea1o_set_chroma_format(1, 1);
                }
#endif
                if (!it && rc == EA1_EINVAL) {
                    fprintf(stderr, "the valid seed frame (layout %d) was rejected: %s\n", layout, ea1_last_error());
                    return 4;
                }
            }
            ea1_ir_storage_destroy(storage);
        }
    }
    printf("fuzz_job: %zu frames (%zu of them in 4:4:4 / 4:2:2 contexts; %zu passed validation), no sanitizer report\n", runs, layout_runs, accepted_host);
    return 0;
}
