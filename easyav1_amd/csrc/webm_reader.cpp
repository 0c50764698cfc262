// WebM (Matroska subset) reader, see include/easyav1_amd_webm.h.  Host code.
// Written from the Matroska / EBML specifications; the behaviour it must reproduce is nestegg's as
// easyav1 uses it (/root/reference/ext/nestegg/nestegg.c: nestegg_init :2172, ne_read_block :1490,
// nestegg_read_packet :2888, nestegg_track_codec_data :2529).
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <string>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_webm.h"

namespace {

enum : uint32_t {
    kEbml = 0x1A45DFA3, kDocType = 0x4282, kDocTypeReadVersion = 0x4285, kEbmlReadVersion = 0x42F7,
    kSegment = 0x18538067, kSeekHead = 0x114D9B74, kInfo = 0x1549A966, kTimecodeScale = 0x2AD7B1, kDuration = 0x4489,
    kTracks = 0x1654AE6B, kTrackEntry = 0xAE, kTrackNumber = 0xD7, kTrackType = 0x83, kCodecId = 0x86, kCodecPrivate = 0x63A2,
    kDefaultDuration = 0x23E383, kCodecDelay = 0x56AA, kVideo = 0xE0, kPixelWidth = 0xB0, kPixelHeight = 0xBA, kAudio = 0xE1, kSamplingFreq = 0xB5,
    kChannels = 0x9F, kCluster = 0x1F43B675, kTimecode = 0xE7, kSimpleBlock = 0xA3, kBlockGroup = 0xA0, kBlock = 0xA1,
    kBlockDuration = 0x9B, kReferenceBlock = 0xFB, kCues = 0x1C53BB6B, kSeek = 0x4DBB, kSeekId = 0x53AB, kSeekPosition = 0x53AC,
    kCuePoint = 0xBB, kCueTime = 0xB3, kCueTrackPositions = 0xB7, kCueTrack = 0xF7, kCueClusterPosition = 0xF1, kTags = 0x1254C367, kChapters = 0x1043A770, kAttachments = 0x1941A469,
};
const uint64_t kUnknown = ~uint64_t(0);

// bytes that are about to be read: a reader over a lazily filled mapping asks its owner to bring them in first (ea1_webm_open_paged)
struct Pager {
    void (*fn)(void *, size_t, size_t) = nullptr;
    void *ud = nullptr;
    void need(size_t off, size_t len, size_t total) const
    {
        if (fn && off < total) fn(ud, off, len < total - off ? len : total - off);
    }
};

struct Reader {
    const uint8_t *p;
    size_t n, pos = 0;
    bool ok = true;
    const Pager *pg = nullptr;
    void touch(size_t len) const
    {
        if (pg) pg->need(pos, len, n);
    }
    // element ID: 1..4 bytes, length marker kept
    uint32_t id()
    {
        if (pos >= n) return ok = false, 0;
        touch(12);   // an ID and the size behind it
        const uint8_t b = p[pos];
        int len = b & 0x80 ? 1 : (b & 0x40 ? 2 : (b & 0x20 ? 3 : (b & 0x10 ? 4 : 0)));
        if (!len || pos + size_t(len) > n) return ok = false, 0;
        uint32_t v = 0;
        for (int i = 0; i < len; i++) v = (v << 8) | p[pos++];
        return v;
    }
    // data size / vint: 1..8 bytes, marker removed; all value bits set = unknown size
    uint64_t vint(int *length = nullptr)
    {
        if (pos >= n) return ok = false, 0;
        touch(8);
        const uint8_t b = p[pos];
        int len = 1;
        while (len <= 8 && !(b & (0x80 >> (len - 1)))) len++;
        if (len > 8 || pos + size_t(len) > n) return ok = false, 0;
        uint64_t v = b & (0xFFu >> len);
        bool all = v == (0xFFu >> len);
        pos++;
        for (int i = 1; i < len; i++) {
            all = all && p[pos] == 0xFF;
            v = (v << 8) | p[pos++];
        }
        if (length) *length = len;
        return all ? kUnknown : v;
    }
    uint64_t uint_of(uint64_t size)
    {
        if (size > 8 || pos + size > n) return ok = false, 0;
        touch(size_t(size));
        uint64_t v = 0;
        for (uint64_t i = 0; i < size; i++) v = (v << 8) | p[pos++];
        return v;
    }
    double float_of(uint64_t size)
    {
        if (size == 4) {
            const uint32_t u = uint32_t(uint_of(4));
            float f;
            memcpy(&f, &u, 4);
            return f;
        }
        if (size == 8) {
            const uint64_t u = uint_of(8);
            double d;
            memcpy(&d, &u, 8);
            return d;
        }
        ok = false;
        return 0;
    }
};

struct Track {
    ea1_webm_track info;
    const uint8_t *priv = nullptr;
    size_t priv_size = 0;
    // a reader over a lazily filled mapping keeps a COPY of the CodecPrivate: the owner gives pages behind the reading position back to the system
    // (easyav1_api.cpp drop_read_pages) and a later track re-open (easyav1_update_settings) must not find zeros there (round-4 advisor).  Shared: Track
    // objects are copied into the reader's vector, the pointers into the copy stay valid
    std::shared_ptr<std::vector<uint8_t>> priv_own;
    std::vector<std::pair<const uint8_t *, size_t>> items;
};

}  // namespace

struct ea1_webm {
    const uint8_t *p = nullptr;
    size_t n = 0;
    uint64_t tc_scale = 1000000;
    double duration = -1;            // in timecode units
    std::vector<Track> tracks;
    size_t segment_end = 0, first_cluster = 0, segment_start = 0;
    Pager pager;
    // Cues (seeking): where the element is (from the SeekHead, or met before the first Cluster), read on the first seek
    size_t cues_at = 0;
    bool cues_loaded = false;
    struct Cue { uint64_t time; uint64_t track; size_t cluster; };   // time in timecode units, cluster = absolute offset
    std::vector<Cue> cues;
    // read state
    size_t pos = 0;
    bool have_tc = false;
    uint64_t cluster_tc = 0;
};

namespace {

int codec_of(const std::string &id)
{
    if (id == "V_VP8") return EA1_WEBM_CODEC_VP8;
    if (id == "V_VP9") return EA1_WEBM_CODEC_VP9;
    if (id == "V_AV1") return EA1_WEBM_CODEC_AV1;
    if (id == "A_VORBIS") return EA1_WEBM_CODEC_VORBIS;
    if (id == "A_OPUS") return EA1_WEBM_CODEC_OPUS;
    return EA1_WEBM_CODEC_UNKNOWN;
}

bool parse_track(Reader &r, size_t end, Track &t)
{
    memset(&t.info, 0, sizeof t.info);
    t.info.type = EA1_WEBM_TRACK_UNKNOWN;
    t.info.codec = EA1_WEBM_CODEC_UNKNOWN;
    while (r.ok && r.pos < end) {
        const uint32_t id = r.id();
        const uint64_t size = r.vint();
        if (!r.ok || size == kUnknown || r.pos + size > end) return false;
        const size_t next = r.pos + size_t(size);
        switch (id) {
        case kTrackNumber: t.info.number = r.uint_of(size); break;
        case kTrackType: {
            const uint64_t ty = r.uint_of(size);
            t.info.type = ty == 1 ? EA1_WEBM_TRACK_VIDEO : (ty == 2 ? EA1_WEBM_TRACK_AUDIO : EA1_WEBM_TRACK_UNKNOWN);
            break;
        }
        case kCodecId:
            r.touch(size_t(size));
            t.info.codec = codec_of(std::string(reinterpret_cast<const char *>(r.p + r.pos), size_t(size)));
            break;
        case kCodecPrivate:
            r.touch(size_t(size));
            t.priv = r.p + r.pos;
            t.priv_size = size_t(size);
            if (r.pg && r.pos + size_t(size) <= r.n) {
                t.priv_own = std::make_shared<std::vector<uint8_t>>(r.p + r.pos, r.p + r.pos + size_t(size));
                t.priv = t.priv_own->data();
            }
            break;
        case kDefaultDuration: t.info.default_duration = r.uint_of(size); break;
        case kCodecDelay: t.info.codec_delay = r.uint_of(size); break;
        case kVideo:
        case kAudio:
            while (r.ok && r.pos < next) {
                const uint32_t cid = r.id();
                const uint64_t cs = r.vint();
                if (!r.ok || cs == kUnknown || r.pos + cs > next) return false;
                const size_t cnext = r.pos + size_t(cs);
                if (cid == kPixelWidth) t.info.width = uint32_t(r.uint_of(cs));
                else if (cid == kPixelHeight) t.info.height = uint32_t(r.uint_of(cs));
                else if (cid == kSamplingFreq) t.info.rate = r.float_of(cs);
                else if (cid == kChannels) t.info.channels = uint32_t(r.uint_of(cs));
                r.pos = cnext;
            }
            break;
        default: break;
        }
        r.pos = next;
    }
    return r.ok;
}

}  // namespace

namespace {

// Xiph lacing of a Vorbis CodecPrivate: count - 1, then the sizes of all but the last header
bool split_vorbis_private(Track &t)
{
    const uint8_t *p = t.priv;
    const size_t n = t.priv_size;
    if (n < 1) return false;
    const unsigned count = unsigned(p[0]) + 1;
    if (count != 3) return false;
    size_t pos = 1, sizes[3] = {0, 0, 0}, sum = 0;
    for (unsigned i = 0; i + 1 < count; i++) {
        for (;;) {
            if (pos >= n) return false;
            const uint8_t b = p[pos++];
            sizes[i] += b;
            if (b != 255) break;
        }
        sum += sizes[i];
    }
    if (pos + sum > n) return false;
    sizes[2] = n - pos - sum;
    for (unsigned i = 0; i < count; i++) {
        t.items.emplace_back(p + pos, sizes[i]);
        pos += sizes[i];
    }
    return true;
}

int parse_headers(ea1_webm &w)
{
    Reader r{w.p, w.n};
    r.pg = &w.pager;
    if (r.id() != kEbml) return ea1::fail(EA1_EINVAL, "ea1_webm_open: not an EBML file");
    uint64_t hs = r.vint();
    if (!r.ok || hs == kUnknown || r.pos + hs > w.n) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad EBML header");
    {
        const size_t end = r.pos + size_t(hs);
        std::string doctype = "matroska";
        uint64_t read_version = 1, ebml_read = 1;
        while (r.ok && r.pos < end) {
            const uint32_t id = r.id();
            const uint64_t sz = r.vint();
            if (!r.ok || sz == kUnknown || r.pos + sz > end) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad EBML header");
            const size_t next = r.pos + size_t(sz);
            if (id == kDocType) {
                r.touch(size_t(sz));
                doctype.assign(reinterpret_cast<const char *>(r.p + r.pos), size_t(sz));
            }
            else if (id == kDocTypeReadVersion) read_version = r.uint_of(sz);
            else if (id == kEbmlReadVersion) ebml_read = r.uint_of(sz);
            r.pos = next;
        }
        while (!doctype.empty() && doctype.back() == 0) doctype.pop_back();
        if (doctype != "webm" && doctype != "matroska") return ea1::fail(EA1_EINVAL, "ea1_webm_open: DocType '%s'", doctype.c_str());
        if (ebml_read != 1 || read_version > 2) return ea1::fail(EA1_EINVAL, "ea1_webm_open: unsupported EBML / DocType read version");
    }
    if (r.id() != kSegment) return ea1::fail(EA1_EINVAL, "ea1_webm_open: no Segment");
    const uint64_t ss = r.vint();
    if (!r.ok) return ea1::fail(EA1_EINVAL, "ea1_webm_open: truncated Segment");
    w.segment_end = (ss == kUnknown || r.pos + ss > w.n) ? w.n : r.pos + size_t(ss);
    w.segment_start = r.pos;
    bool have_tracks = false;
    while (r.pos < w.segment_end) {
        const size_t at = r.pos;
        const uint32_t id = r.id();
        const uint64_t sz = r.vint();
        if (!r.ok) break;
        if (id == kCluster) {
            w.first_cluster = at;
            break;
        }
        if (sz == kUnknown || r.pos + sz > w.segment_end) return ea1::fail(EA1_EINVAL, "ea1_webm_open: element %#x leaves the Segment", id);
        const size_t next = r.pos + size_t(sz);
        if (id == kCues) {
            w.cues_at = at;
        } else if (id == kSeekHead) {
            // where the Cues are (seeking; nestegg reads the SeekHead for the same purpose: /root/reference/ext/nestegg/nestegg.c:1861-2025)
            while (r.ok && r.pos < next) {
                const uint32_t cid = r.id();
                const uint64_t cs = r.vint();
                if (!r.ok || cs == kUnknown || r.pos + cs > next) break;
                const size_t cn = r.pos + size_t(cs);
                if (cid == kSeek) {
                    uint64_t seek_id = 0, seek_pos = 0;
                    while (r.ok && r.pos < cn) {
                        const uint32_t eid = r.id();
                        const uint64_t es = r.vint();
                        if (!r.ok || es == kUnknown || r.pos + es > cn) break;
                        const size_t en = r.pos + size_t(es);
                        if (eid == kSeekId) seek_id = r.uint_of(es);
                        else if (eid == kSeekPosition) seek_pos = r.uint_of(es);
                        r.pos = en;
                    }
                    if (seek_id == kCues && !w.cues_at && w.segment_start + seek_pos < w.segment_end) w.cues_at = w.segment_start + size_t(seek_pos);
                }
                r.pos = cn;
            }
            r.ok = true;   // (a damaged SeekHead only costs the index)
        }
        if (id == kInfo) {
            while (r.ok && r.pos < next) {
                const uint32_t cid = r.id();
                const uint64_t cs = r.vint();
                if (!r.ok || cs == kUnknown || r.pos + cs > next) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad Info");
                const size_t cn = r.pos + size_t(cs);
                if (cid == kTimecodeScale) w.tc_scale = r.uint_of(cs);
                else if (cid == kDuration) w.duration = r.float_of(cs);
                r.pos = cn;
            }
        } else if (id == kTracks) {
            while (r.ok && r.pos < next) {
                const uint32_t cid = r.id();
                const uint64_t cs = r.vint();
                if (!r.ok || cs == kUnknown || r.pos + cs > next) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad Tracks");
                const size_t cn = r.pos + size_t(cs);
                if (cid == kTrackEntry) {
                    Track t;
                    if (!parse_track(r, cn, t)) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad TrackEntry");
                    if (t.info.codec == EA1_WEBM_CODEC_VORBIS) {
                        if (!split_vorbis_private(t)) return ea1::fail(EA1_EINVAL, "ea1_webm_open: bad Vorbis CodecPrivate");
                    } else if (t.priv) {
                        t.items.emplace_back(t.priv, t.priv_size);
                    }
                    t.info.codec_data_count = uint32_t(t.items.size());
                    w.tracks.push_back(t);
                    have_tracks = true;
                }
                r.pos = cn;
            }
        }
        r.pos = next;
    }
    if (!have_tracks) return ea1::fail(EA1_EINVAL, "ea1_webm_open: no Tracks before the first Cluster");
    if (!w.tc_scale) return ea1::fail(EA1_EINVAL, "ea1_webm_open: TimecodeScale 0");
    if (!w.first_cluster) return ea1::fail(EA1_EINVAL, "ea1_webm_open: no Cluster (nestegg_init needs the first one too)");
    w.pos = w.first_cluster;
    return EA1_OK;
}

// a Block / SimpleBlock body: track, timecode, flags, lacing -> frames (AV1 spec-independent; Matroska 'Block')
int parse_block(ea1_webm &w, bool simple, size_t at, size_t size, ea1_webm_packet *out)
{
    w.pager.need(at, size, w.n);
    Reader r{w.p, at + size, at};
    int len = 0;
    const uint64_t number = r.vint(&len);
    if (!r.ok || number == 0 || number == kUnknown) return -1;
    if (r.pos + 3 > r.n) return -1;
    const int16_t tc = int16_t((uint16_t(r.p[r.pos]) << 8) | r.p[r.pos + 1]);
    const uint8_t flags = r.p[r.pos + 2];
    r.pos += 3;
    const int lacing = (flags & 0x06) >> 1;
    uint32_t frames = 1;
    if (lacing) {
        if (r.pos >= r.n) return -1;
        frames = uint32_t(r.p[r.pos++]) + 1;
    }
    size_t sizes[EA1_WEBM_MAX_FRAMES];
    const size_t end = at + size;
    if (lacing == 0) {
        sizes[0] = end - r.pos;
    } else if (lacing == 1) {          // Xiph: sizes of all but the last frame as runs of 255 + remainder
        if (frames == 1) return -1;
        size_t sum = 0;
        for (uint32_t i = 0; i + 1 < frames; i++) {
            size_t v = 0;
            for (;;) {
                if (r.pos >= end) return -1;
                const uint8_t b = r.p[r.pos++];
                v += b;
                if (b != 255) break;
            }
            sizes[i] = v;
            sum += v;
        }
        if (r.pos + sum > end) return -1;
        sizes[frames - 1] = end - r.pos - sum;
    } else if (lacing == 2) {          // fixed
        if ((end - r.pos) % frames) return -1;
        for (uint32_t i = 0; i < frames; i++) sizes[i] = (end - r.pos) / frames;
    } else {                           // EBML: first size as a vint, then signed vint differences
        if (frames == 1) return -1;
        int l = 0;
        uint64_t v = r.vint(&l);
        if (!r.ok || v == kUnknown) return -1;
        sizes[0] = size_t(v);
        size_t sum = sizes[0];
        for (uint32_t i = 1; i + 1 < frames; i++) {
            const uint64_t raw = r.vint(&l);
            if (!r.ok) return -1;
            const int64_t bias = (int64_t(1) << (7 * l - 1)) - 1;
            const int64_t diff = (raw == kUnknown ? ((int64_t(1) << (7 * l)) - 1) : int64_t(raw)) - bias;
            const int64_t nv = int64_t(sizes[i - 1]) + diff;
            if (nv < 0) return -1;
            sizes[i] = size_t(nv);
            sum += sizes[i];
        }
        if (r.pos + sum > end) return -1;
        sizes[frames - 1] = end - r.pos - sum;
    }
    uint32_t track = 0;
    for (; track < w.tracks.size(); track++)
        if (w.tracks[track].info.number == number) break;
    if (track == w.tracks.size()) return -1;
    if (!w.have_tc) return -1;
    int64_t abs_tc = int64_t(tc) + int64_t(w.cluster_tc);
    if (abs_tc < 0) abs_tc = 0;
    out->track = track;
    out->tstamp = uint64_t(abs_tc) * w.tc_scale;
    out->duration = 0;
    out->keyframe = simple ? ((flags & 0x80) ? 1 : 0) : -1;
    out->count = frames;
    size_t pos = r.pos;
    for (uint32_t i = 0; i < frames; i++) {
        if (pos + sizes[i] > end) return -1;
        out->data[i] = w.p + pos;
        out->size[i] = sizes[i];
        pos += sizes[i];
    }
    return 1;
}

}  // namespace

extern "C" int ea1_webm_open_memory(ea1_webm **out, const uint8_t *data, size_t size)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_webm_open_memory: out is NULL");
    *out = nullptr;
    if (!data || !size) return ea1::fail(EA1_EINVAL, "ea1_webm_open_memory: empty input");
    auto *w = new (std::nothrow) ea1_webm();
    if (!w) return ea1::fail(EA1_ENOMEM, "out of memory");
    w->p = data;
    w->n = size;
    const int rc = parse_headers(*w);
    if (rc != EA1_OK) {
        delete w;
        return rc;
    }
    *out = w;
    return EA1_OK;
}

extern "C" int ea1_webm_open_paged(ea1_webm **out, const uint8_t *data, size_t size, void (*need)(void *userdata, size_t offset, size_t length), void *userdata)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_webm_open_paged: out is NULL");
    *out = nullptr;
    if (!data || !size || !need) return ea1::fail(EA1_EINVAL, "ea1_webm_open_paged: empty input");
    auto *w = new (std::nothrow) ea1_webm();
    if (!w) return ea1::fail(EA1_ENOMEM, "out of memory");
    w->p = data;
    w->n = size;
    w->pager.fn = need;
    w->pager.ud = userdata;
    const int rc = parse_headers(*w);
    if (rc != EA1_OK) {
        delete w;
        return rc;
    }
    *out = w;
    return EA1_OK;
}

namespace {
// the Cues element: CuePoint { CueTime, CueTrackPositions { CueTrack, CueClusterPosition } ... }
void load_cues(ea1_webm &w)
{
    w.cues_loaded = true;
    if (!w.cues_at) return;
    Reader r{w.p, w.segment_end, w.cues_at};
    r.pg = &w.pager;
    if (r.id() != kCues) return;
    const uint64_t sz = r.vint();
    if (!r.ok || sz == kUnknown || r.pos + sz > w.segment_end) return;
    const size_t end = r.pos + size_t(sz);
    w.pager.need(r.pos, size_t(sz), w.n);
    while (r.ok && r.pos < end) {
        const uint32_t id = r.id();
        const uint64_t cs = r.vint();
        if (!r.ok || cs == kUnknown || r.pos + cs > end) break;
        const size_t cn = r.pos + size_t(cs);
        if (id == kCuePoint) {
            uint64_t time = 0;
            bool have_time = false;
            while (r.ok && r.pos < cn) {
                const uint32_t pid = r.id();
                const uint64_t ps = r.vint();
                if (!r.ok || ps == kUnknown || r.pos + ps > cn) break;
                const size_t pn = r.pos + size_t(ps);
                if (pid == kCueTime) {
                    time = r.uint_of(ps);
                    have_time = true;
                } else if (pid == kCueTrackPositions && have_time) {
                    uint64_t track = 0, cpos = ~uint64_t(0);
                    while (r.ok && r.pos < pn) {
                        const uint32_t tid = r.id();
                        const uint64_t ts = r.vint();
                        if (!r.ok || ts == kUnknown || r.pos + ts > pn) break;
                        const size_t tn = r.pos + size_t(ts);
                        if (tid == kCueTrack) track = r.uint_of(ts);
                        else if (tid == kCueClusterPosition) cpos = r.uint_of(ts);
                        r.pos = tn;
                    }
                    if (cpos != ~uint64_t(0) && w.segment_start + cpos < w.segment_end) w.cues.push_back({time, track, w.segment_start + size_t(cpos)});
                }
                r.pos = pn;
            }
        }
        r.pos = cn;
    }
}
}  // namespace

extern "C" int ea1_webm_has_cues(ea1_webm *w)
{
    if (!w) return 0;
    if (!w->cues_loaded) load_cues(*w);
    return w->cues.empty() ? 0 : 1;
}

extern "C" int ea1_webm_seek(ea1_webm *w, uint32_t track, uint64_t ns)
{
    if (!w || track >= w->tracks.size()) return ea1::fail(EA1_EINVAL, "ea1_webm_seek: bad arguments");
    if (!w->cues_loaded) load_cues(*w);
    const uint64_t number = w->tracks[track].info.number;
    const ea1_webm::Cue *best = nullptr;
    for (const ea1_webm::Cue &c : w->cues) {
        if (c.track != number || c.time * w->tc_scale > ns) continue;
        if (!best || c.time >= best->time) best = &c;
    }
    if (!best) return ea1::fail(EA1_ENOENT, "ea1_webm_seek: no cue point of track %u at or before %llu ns", track, (unsigned long long)ns);
    // the position must hold a Cluster (a damaged index must not send the packet reader into the middle of an element)
    Reader r{w->p, w->segment_end, best->cluster};
    r.pg = &w->pager;
    if (r.id() != kCluster) return ea1::fail(EA1_EINVAL, "ea1_webm_seek: the cue point does not name a Cluster");
    w->pos = best->cluster;
    w->have_tc = false;
    return EA1_OK;
}

extern "C" void ea1_webm_destroy(ea1_webm *w) { delete w; }

extern "C" int ea1_webm_track_count(const ea1_webm *w, uint32_t *count)
{
    if (!w || !count) return ea1::fail(EA1_EINVAL, "ea1_webm_track_count: NULL argument");
    *count = uint32_t(w->tracks.size());
    return EA1_OK;
}

extern "C" int ea1_webm_track_info(const ea1_webm *w, uint32_t track, ea1_webm_track *out)
{
    if (!w || !out || track >= w->tracks.size()) return ea1::fail(EA1_EINVAL, "ea1_webm_track_info: bad arguments");
    *out = w->tracks[track].info;
    return EA1_OK;
}

extern "C" int ea1_webm_codec_data(const ea1_webm *w, uint32_t track, uint32_t item, const uint8_t **data, size_t *size)
{
    if (!w || !data || !size || track >= w->tracks.size() || item >= w->tracks[track].items.size())
        return ea1::fail(EA1_EINVAL, "ea1_webm_codec_data: bad arguments");
    *data = w->tracks[track].items[item].first;
    *size = w->tracks[track].items[item].second;
    return EA1_OK;
}

extern "C" int ea1_webm_duration(const ea1_webm *w, uint64_t *ns)
{
    if (!w || !ns) return ea1::fail(EA1_EINVAL, "ea1_webm_duration: NULL argument");
    if (w->duration < 0) return ea1::fail(EA1_ESTATE, "ea1_webm_duration: the Segment has no Duration");
    *ns = uint64_t(w->duration * double(w->tc_scale));
    return EA1_OK;
}

extern "C" int ea1_webm_tstamp_scale(const ea1_webm *w, uint64_t *scale)
{
    if (!w || !scale) return ea1::fail(EA1_EINVAL, "ea1_webm_tstamp_scale: NULL argument");
    *scale = w->tc_scale;
    return EA1_OK;
}

extern "C" int ea1_webm_rewind(ea1_webm *w)
{
    if (!w) return ea1::fail(EA1_EINVAL, "ea1_webm_rewind: NULL handle");
    w->pos = w->first_cluster;
    w->have_tc = false;
    return EA1_OK;
}

extern "C" int ea1_webm_read_packet(ea1_webm *w, ea1_webm_packet *out)
{
    if (!w || !out) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: NULL argument");
    // The packet stream is read FLAT, as nestegg does: a Cluster header only (re)sets the cluster
    // timecode, blocks and groups are taken wherever they come, everything else is skipped.  That is
    // what makes unknown-size Clusters / Segments work without tracking nesting.  Anything that runs
    // past the end of the data is the end of the stream (0); malformed content inside it is an error.
    for (;;) {
        if (w->pos >= w->segment_end) return 0;
        Reader r{w->p, w->segment_end, w->pos};
        r.pg = &w->pager;
        uint32_t id = r.id();
        uint64_t sz = r.vint();
        if (!r.ok) return 0;
        if (id == kCluster) {
            id = r.id();
            sz = r.vint();
            if (!r.ok) return 0;
            if (id == 0xBF) {   // CRC-32 may precede the Timecode
                if (sz == kUnknown || r.pos + sz > w->segment_end) return 0;
                r.pos += size_t(sz);
                id = r.id();
                sz = r.vint();
                if (!r.ok) return 0;
            }
            if (id != kTimecode) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: a Cluster must start with its Timecode");
            if (sz == kUnknown || r.pos + sz > w->segment_end) return 0;
            if (sz > 8) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: Timecode of %llu bytes", (unsigned long long)sz);
            w->cluster_tc = r.uint_of(sz);
            w->have_tc = true;
            w->pos = r.pos;
            continue;
        }
        if (sz == kUnknown) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: element %#x of unknown size", id);
        if (r.pos + sz > w->segment_end) return 0;
        const size_t body = r.pos, next = r.pos + size_t(sz);
        w->pos = next;
        if (id == kSimpleBlock) {
            if (parse_block(*w, true, body, size_t(sz), out) != 1) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: malformed SimpleBlock");
            return 1;
        }
        if (id == kBlockGroup) {
            w->pager.need(body, size_t(sz), w->n);
            Reader g{w->p, next, body};
            bool got = false, has_ref = false;
            uint64_t dur = 0;
            while (g.ok && g.pos < next) {
                const uint32_t cid = g.id();
                const uint64_t cs = g.vint();
                if (!g.ok || cs == kUnknown || g.pos + cs > next) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: malformed BlockGroup");
                const size_t cn = g.pos + size_t(cs);
                if (cid == kBlock) {
                    if (parse_block(*w, false, g.pos, size_t(cs), out) != 1) return ea1::fail(EA1_EINVAL, "ea1_webm_read_packet: malformed Block");
                    got = true;
                } else if (cid == kBlockDuration) {
                    dur = g.uint_of(cs) * w->tc_scale;
                } else if (cid == kReferenceBlock) {
                    has_ref = true;   // the block depends on another one: not a keyframe
                }
                g.pos = cn;
            }
            if (got) {
                out->duration = dur;
                if (has_ref) out->keyframe = 0;
                return 1;
            }
        }
        // Cues, Tags, Void, CRC-32, unknown elements: skipped
    }
}
