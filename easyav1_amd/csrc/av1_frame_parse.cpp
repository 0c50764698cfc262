// ea1_av1_frame_parser: the block-level AV1 parse behind the C ABI (include/easyav1_amd_obu.h).  One temporal unit in, one
// ea1_ir_frame out.  See av1_tile_syntax.h for what is parsed and av1_frame_driver.h for the frame-level bookkeeping.
#include <chrono>
#include <cstdlib>
#include <new>

#include "av1_frame_driver.h"

using namespace ea1::av1;

struct ea1_av1_frame_parser {
    FrameDriver d;
    std::vector<ea1_av1_tile> tiles;
};

extern "C" int ea1_av1_frame_parser_create(const ea1_av1_tables *tables, ea1_av1_frame_parser **out)
{
    if (!tables || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_create: NULL argument");
    ea1_av1_frame_parser *p = new (std::nothrow) ea1_av1_frame_parser;
    if (!p) return ea1::fail(EA1_ENOMEM, "ea1_av1_frame_parser_create: out of memory");
    p->d.tab = tables_of(tables);
    const int rc = ea1_av1_parser_create(&p->d.hp);
    if (rc != EA1_OK) {
        delete p;
        return rc;
    }
    p->tiles.resize(size_t(EA1_MAX_TILE_COLS) * EA1_MAX_TILE_ROWS);
    *out = p;
    return EA1_OK;
}

extern "C" void ea1_av1_frame_parser_destroy(ea1_av1_frame_parser *p)
{
    if (!p) return;
    ea1_av1_parser_destroy(p->d.hp);
    delete p;
}

extern "C" void ea1_av1_frame_parser_reset(ea1_av1_frame_parser *p)
{
    if (!p) return;
    // the header parser keeps its sequence header; its slots are overwritten by the key frame that must follow
    p->d.reset_slots();
}

extern "C" int ea1_av1_frame_parser_parse(ea1_av1_frame_parser *p, const uint8_t *data, size_t size, ea1_av1_parsed *out)
{
    if (!p || !data || !out) return ea1::fail(EA1_EINVAL, "ea1_av1_frame_parser_parse: NULL argument");
    const auto t0 = std::chrono::steady_clock::now();
    memset(out, 0, sizeof *out);
    size_t consumed = 0;
    int count = 0;
    int rc = ea1_av1_parser_next_frame(p->d.hp, data, size, &consumed, &out->header, p->tiles.data(), int(p->tiles.size()), &count);
    if (rc != EA1_OK) return rc;
    const ea1_av1_frame_header &hdr = out->header;
    out->consumed = consumed;
    if (hdr.show_existing_frame) {
        rc = p->d.show_existing(hdr, &out->frame);
        if (rc != EA1_OK) return rc;
        out->show_existing = 1;
        out->show_frame = 1;
        out->has_film_grain = ea1_av1_frame_header_film_grain(&hdr, &p->d.seq, &out->film_grain) == EA1_OK;
        out->parse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return EA1_OK;
    }
    rc = p->d.begin_frame(hdr);
    if (rc != EA1_OK) return rc;
    static const bool slow = getenv("EA1_AV1_SLOW_MSAC") != nullptr;
    for (int k = 0; k < count; k++) {
        const ea1_av1_tile &t = p->tiles[size_t(k)];
        if (slow) {
            SlowMsacReader io;
            rc = io.init(data + t.offset, t.size, hdr.disable_cdf_update);
            if (rc == EA1_OK) rc = p->d.run_tile(io, t, t.tile_num == hdr.context_update_tile_id);
        } else {
            MsacReader io;
            rc = io.init(data + t.offset, t.size, hdr.disable_cdf_update);
            if (rc == EA1_OK) rc = p->d.run_tile(io, t, t.tile_num == hdr.context_update_tile_id);
        }
        if (rc != EA1_OK) return rc;
    }
    rc = p->d.end_frame(hdr, &out->frame);
    if (rc != EA1_OK) return rc;
    out->show_frame = hdr.show_frame;
    out->has_film_grain = ea1_av1_frame_header_film_grain(&hdr, &p->d.seq, &out->film_grain) == EA1_OK;
    out->parse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return EA1_OK;
}
