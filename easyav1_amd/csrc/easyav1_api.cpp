// libeasyav1.so: the 34 entry points of easyav1's public interface (include/easyav1.h, binary
// compatible with /root/reference/src/easyav1.h:449-822) on top of this repository's C ABI.
//
//   demux   ea1_webm_*            (easyav1_amd_webm.h)    for nestegg, src/easyav1.c:1279-1506, :1904-2017
//   audio   ea1_vorbis_decoder_*  (easyav1_amd_vorbis.h)  for vorbis_synthesis / blockin / pcmout, :2434-2499:
//           host packet front end -> GPU IMDCT + window + overlap-add, several packets per launch
//           (look-ahead), PCM into the 4096-sample hand-off buffer (ea1_audio_ring_*, :2501-2532, :3604)
//   video   track facts from the container, OBU / sequence / frame headers through ea1_av1_parser_* (easyav1_amd_obu.h); pictures come from
//           the GPU reconstruction path (easyav1_video_sink.cpp) for a plain AV1 bitstream -- the block-level symbol parser of this
//           repository (ea1_av1_frame_parser_*, the specification's tables shipped beside the library in data/av1_spec_tables.bin) runs on
//           the sink's helper threads, one parser per GOP chain -- and for packets that carry frames already parsed
//           (easyav1_amd_ir_packet.h).  8-bit 4:2:0 / 4:0:0; anything else is EASYAV1_STATUS_DECODER_ERROR with the reason.
//
// What stands where the reference has its decoder thread with the 10-packet prefetch ring (src/easyav1.c:1813-1824, :2141): the sink's
// pipeline (helpers, GOP chains, pictures parked in HBM).  Seeks position the demuxer by the Cues when the file has them (ea1_webm_seek,
// src/easyav1.c:2942, :3026) and decode forward from the last key frame inside the seek (:2961-3216).
// Behaviour that callers can observe is kept: return codes, sticky errors (:2625), the order in
// which packets of the two tracks are taken (:2103-2131), timestamps in ms, ownership (:3990-4039), the settings' branches (:3852-3949).
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <algorithm>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "easyav1.h"
#include "easyav1_amd_audio.h"
#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_obu.h"
#include "easyav1_amd_vorbis.h"
#include "easyav1_amd_webm.h"
#include "easyav1_video_sink.h"

namespace {

constexpr unsigned kVideoPrefetch = 10;      // packets kept ahead of the position (src/easyav1.c:40)
constexpr size_t kMaxLookaheadPackets = 8192;    // the look-ahead of long GOPs (want_more_packets): at most this many packets ...
constexpr size_t kMaxLookaheadBytes = size_t(768) << 20;   // ... and bytes queued
constexpr uint64_t kMaxLookaheadPictures = uint64_t(128) << 30;   // ... and bytes of the pictures they decode to (GOPs decoded ahead wait in HBM: 288 GB on an MI355X)
constexpr int kFetchBurst = 32;              // packets read per call while a longer look-ahead fills (sync_queues)
constexpr int kAudioLookahead = 16;          // audio packets decoded per GPU launch
constexpr easyav1_timestamp kSkipMs = 1000;  // decode_until skips ahead beyond this lag (src/easyav1.c:42)

const easyav1_settings kDefaults = {
    EASYAV1_TRUE, EASYAV1_TRUE, EASYAV1_TRUE, EASYAV1_TRUE, EASYAV1_FALSE, {nullptr, nullptr, nullptr}, 0, 0, EASYAV1_FALSE, 0,
    EASYAV1_LOG_LEVEL_WARNING};

struct QueuedPacket {
    ea1_webm_packet pk;
    easyav1_timestamp ts;   // ms, audio offset applied
    bool keyframe;
    uint64_t first_chunk;   // audio: number of its first frame among the kept audio frames of the track
    bool done;              // video: already handed to the parser / the GPU path (packets are worked on ahead of the position)
    bool has_ticket;        // video: it carried parsed frames; last_ticket = the sink's ticket of the last of them
    uint64_t last_ticket;
    unsigned frames;        // video: parsed frames it carried
};

struct DecodedChunk {
    std::vector<float> pcm;   // channels x count, planar
    int count;
    bool accepted;
};

}  // namespace

struct easyav1_t {
    easyav1_settings settings;
    int status = EASYAV1_STATUS_OK;   // easyav1_status, negative detail codes included
    std::mutex info;                  // position + status, which the playback thread shares with the caller
    // Everything else the playback thread and the caller both reach -- the packet queues, the video sink, the audio ring and its
    // `pending` flag, the frame structs the getters hand out -- is guarded by `state`: every entry point that touches it holds the
    // lock, the playback thread holds it around each decode step and each seek (the reference: its io / decoder / info mutexes,
    // src/easyav1.c:3525-3548).  Recursive: callbacks run under it on the playback thread and may call the getters.
    std::recursive_mutex state;
    easyav1_timestamp position = 0, duration = 0;
    uint64_t time_scale = 0;
    // input
    const uint8_t *data = nullptr;
    size_t size = 0;
    // file / custom stream input: a mapping of the stream's size that is filled THROUGH THE CALLER'S CALLBACKS, page by page, when the demuxer
    // gets there (the reference reads through them as it goes, src/easyav1.c:717-848); pages that are in stay in (queued packets point into them)
    struct LazyFile {
        uint8_t *base = nullptr;
        size_t size = 0;
        easyav1_stream st{};
        std::vector<uint8_t> have;       // per page
        std::mutex mu;                   // the packet reader and the audio look-ahead reader share it
        bool failed = false;
        uint64_t bytes_read = 0, reads = 0;
        size_t dropped = 0;              // pages below this one have been given back (drop_read_pages)
        uint64_t pages_dropped = 0;
        static constexpr size_t kPage = 64 * 1024;
        static constexpr size_t kKeepFront = 8;   // the first pages stay: headers, codec private data the decoders were opened from
    } lazy;
    FILE *close_on_destroy = nullptr;
    ea1_webm *webm = nullptr, *ahead = nullptr;   // packet reader; audio look-ahead reader over the same bytes
    uint32_t num_tracks = 0, video_tracks = 0, audio_tracks = 0;
    // queues (src/easyav1.c:1752-2132)
    std::deque<QueuedPacket> vq, aq;
    size_t vq_keys = 0, vq_bytes = 0;   // packets of vq the container calls key frames / payload bytes queued (the look-ahead of long GOPs, want_more_packets)
    bool all_fetched = false, synced = false;
    bool fetch_failed = false;    // the demuxer could not read on (a failed read through the caller's callbacks, a broken element): the packets in front of it
                                  // are still decoded, then the stream ends in EASYAV1_STATUS_DECODER_ERROR instead of FINISHED
    uint64_t chunks_queued = 0;       // kept audio frames handed to the queue so far
    easyav1_timestamp skip_before = 0;   // after a seek: packets before this time are dropped by both readers ...
    easyav1_timestamp video_from = 0;    // ... except video packets from the last key frame at or before the target: decoded, their pictures discarded
    struct {
        bool active = false;
        uint32_t track = 0, width = 0, height = 0, fps = 0;
        uint64_t processed = 0;
        bool any_decoded = false;         // a video packet has had its turn since the start / the last seek ...
        easyav1_timestamp decoded_ts = 0; // ... the last one's timestamp: pictures of packets worked on AHEAD of that are not shown yet
        ea1_av1_parser *parser = nullptr;
        bool have_seq = false, warned = false;
        ea1_av1_sequence_header seq;      // the newest one READ (the pipeline works ahead of the picture that is shown) ...
        std::vector<ea1_av1_sequence_header> seqs;   // ... every distinct one of the track, numbered: a packet is submitted under its header's number
        uint32_t seq_tag = 0, sink_tag = 0;          //     (video_sink_set_tag) and its picture is described from that header, as dav1d's pictures carry theirs
        easyav1_video_frame frame;
        ea1::VideoSink *sink = nullptr;   // GPU reconstruction of parsed-frame packets (created on first use)
        // a sequence header the sink's context cannot hold (another chroma layout, frames beyond its size; dav1d re-allocates at such a key frame): its
        // packet is held back until its turn has come (`barrier`: nothing behind it is queued either), then the sink retires -- it keeps the pictures
        // nobody took yet -- and a new one is made for the new sequence
        ea1::VideoSink *old_sink = nullptr;
        bool barrier = false;
    } video;
    struct {
        bool active = false;
        uint32_t track = 0, channels = 0, rate = 0;
        int64_t offset = 0;           // ms added to audio timestamps (settings + codec delay)
        uint64_t codec_delay = 0;     // ns, the track's CodecDelay
        ea1_vorbis_decoder *dec = nullptr;
        ea1_audio_ring *ring = nullptr;
        bool pending = false;         // samples wait in the ring (has_samples_in_buffer)
        easyav1_audio_frame frame;
        std::vector<const float *> planes;
        std::deque<DecodedChunk> cache;   // look-ahead results, cache[0] is chunk number cache_first
        uint64_t cache_first = 0;
        bool ahead_done = false;
    } audio;
    struct {
        std::thread thread;
        std::atomic<bool> active{false}, pause{false}, seek_requested{false};
        std::atomic<uint64_t> seek_to{0};
        std::mutex mutex;
    } playback;
};

namespace {

void logf(const easyav1_t *e, easyav1_log_level_t level, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
void logf(const easyav1_t *e, easyav1_log_level_t level, const char *fmt, ...)
{
    if ((e ? e->settings.log_level : kDefaults.log_level) < level) return;
    static const char *names[] = {"error", "warning", "info"};
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "easyav1 (%s): ", names[level]);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

bool is_error(int s) { return s <= 0; }   // ERROR and every detail code (src/easyav1.c EASYAV1_STATUS_IS_ERROR)

void set_error(easyav1_t *e, easyav1_status code, const char *what)
{
    logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", what);
    if (e) {
        std::lock_guard<std::mutex> g(e->info);
        e->status = code;
    }
}

easyav1_timestamp to_ms(const easyav1_t *e, uint64_t ns) { return ns / e->time_scale; }

// ---- tracks ------------------------------------------------------------------------------------
void close_audio(easyav1_t *e)
{
    if (e->audio.dec) ea1_vorbis_decoder_destroy(e->audio.dec);
    if (e->audio.ring) ea1_audio_ring_destroy(e->audio.ring);
    e->audio.dec = nullptr;
    e->audio.ring = nullptr;
    e->audio.cache.clear();
    e->audio.active = false;
    e->audio.pending = false;
}

void close_video(easyav1_t *e)
{
    if (e->video.sink) ea1::video_sink_destroy(e->video.sink);
    if (e->video.old_sink) ea1::video_sink_destroy(e->video.old_sink);
    if (e->video.parser) ea1_av1_parser_destroy(e->video.parser);
    e->video.sink = e->video.old_sink = nullptr;
    e->video.barrier = false;
    e->video.sink_tag = 0;   // (a new sink starts at tag 0)
    e->video.parser = nullptr;
    e->video.active = false;
}

void fill_frame_properties(easyav1_t *e, const ea1_av1_sequence_header &s);
void take_sequence_header(easyav1_t *e, const ea1_av1_sequence_header &seq);
int open_reader(easyav1_t *e, ea1_webm **out);   // a demuxer over the input: the caller's memory, or the lazily filled mapping of a stream

// a sequence header among the OBUs of `data`, if there is one (malformed bytes: nothing is taken)
void scan_sequence_header(easyav1_t *e, const uint8_t *data, size_t size)
{
    for (size_t pos = 0; pos < size;) {
        ea1_obu obu;
        if (ea1_av1_next_obu(data + pos, size - pos, &obu) != EA1_OK) return;
        if (obu.type == 1) {
            ea1_av1_sequence_header seq;
            memset(&seq, 0, sizeof seq);
            if (ea1_av1_parse_sequence_header(data + pos, obu.total_size, &seq) == EA1_OK) take_sequence_header(e, seq);
            return;
        }
        pos += obu.total_size;
    }
}

bool open_video(easyav1_t *e, uint32_t track, const ea1_webm_track &ti)
{
    if (!ti.default_duration) {   // the reference asks nestegg for it and fails without (src/easyav1.c:1394)
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get video track frame duration.");
        return false;
    }
    if (ea1_av1_parser_create(&e->video.parser) != EA1_OK) {
        set_error(e, EASYAV1_STATUS_OUT_OF_MEMORY, "Failed to initialize the AV1 header parser.");
        return false;
    }
    e->video.fps = unsigned(1000 * e->time_scale / ti.default_duration);
    e->video.width = ti.width;
    e->video.height = ti.height;
    e->video.track = track;
    e->video.active = true;
    e->video.have_seq = false;
    e->video.seqs.clear();
    e->video.seq_tag = 0;
    memset(&e->video.frame, 0, sizeof e->video.frame);
    // the track's CodecPrivate (AV1CodecConfigurationRecord: 4 bytes, then configOBUs) usually carries the sequence header: known
    // before the first packet, and after a seek that lands behind the packet that repeated it (the reference looks for it from
    // the start of the file in that case: SEEKING_FOR_SQHDR, src/easyav1.c:2601-2607)
    const uint8_t *priv = nullptr;
    size_t priv_size = 0;
    if (ti.codec_data_count >= 1 && ea1_webm_codec_data(e->webm, track, 0, &priv, &priv_size) == EA1_OK && priv_size > 4) scan_sequence_header(e, priv + 4, priv_size - 4);
    logf(e, EASYAV1_LOG_LEVEL_INFO, "Video initialized. Size: %ux%u, %u FPS.", e->video.width, e->video.height, e->video.fps);
    return true;
}

bool make_audio_buffer(easyav1_t *e)
{
    if (e->audio.ring) ea1_audio_ring_destroy(e->audio.ring);
    e->audio.ring = nullptr;
    if (ea1_audio_ring_create(&e->audio.ring, int(e->audio.channels), e->settings.interlace_audio ? 1 : 0) != EA1_OK) {
        set_error(e, EASYAV1_STATUS_OUT_OF_MEMORY, "Failed to allocate audio buffer.");
        return false;
    }
    memset(&e->audio.frame, 0, sizeof e->audio.frame);
    e->audio.frame.channels = e->audio.channels;
    e->audio.planes.assign(e->audio.channels, nullptr);
    e->audio.pending = false;
    return true;
}

bool open_audio(easyav1_t *e, uint32_t track, const ea1_webm_track &ti)
{
    if (ti.codec_data_count != 3) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Vorbis data should always have 3 headers.");
        return false;
    }
    const uint8_t *hdr[3];
    size_t hsz[3];
    for (uint32_t k = 0; k < 3; k++)
        if (ea1_webm_codec_data(e->webm, track, k, &hdr[k], &hsz[k]) != EA1_OK) {
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get audio codec header data.");
            return false;
        }
    const int rc = ea1_vorbis_decoder_create(&e->audio.dec, hdr[0], hsz[0], hdr[1], hsz[1], hdr[2], hsz[2], 0);
    if (rc != EA1_OK) {
        // no CPU fallback: without the GPU library / device the audio track cannot be decoded
        logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, rc == EA1_EINVAL ? "Failed to process audio codec headers." : "Failed to initialize vorbis synthesis.");
        return false;
    }
    e->audio.channels = ti.channels;
    e->audio.rate = unsigned(ti.rate);
    e->audio.track = track;
    e->audio.codec_delay = ti.codec_delay;
    e->audio.offset = e->settings.audio_offset_time + int64_t(to_ms(e, ti.codec_delay));   // src/easyav1.c:1495-1496
    if (!make_audio_buffer(e)) return false;
    e->audio.active = true;
    logf(e, EASYAV1_LOG_LEVEL_INFO, "Audio initialized. Channels: %u, sample rate: %uhz.", e->audio.channels, e->audio.rate);
    return true;
}

// track discovery in file order: the n-th video / audio track the settings ask for (src/easyav1.c:1279-1384)
bool open_tracks(easyav1_t *e)
{
    if (ea1_webm_track_count(e->webm, &e->num_tracks) != EA1_OK) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get track count");
        return false;
    }
    e->video_tracks = e->audio_tracks = 0;
    int vtrack = -1, atrack = -1;
    ea1_webm_track vinfo, ainfo;
    for (uint32_t t = 0; t < e->num_tracks; t++) {
        ea1_webm_track ti;
        if (ea1_webm_track_info(e->webm, t, &ti) != EA1_OK) {
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get track type.");
            return false;
        }
        if (ti.type == EA1_WEBM_TRACK_UNKNOWN) {
            logf(e, EASYAV1_LOG_LEVEL_WARNING, "Unknown track type found, ignoring.");
            continue;
        }
        const bool video = ti.type == EA1_WEBM_TRACK_VIDEO;
        uint32_t &seen = video ? e->video_tracks : e->audio_tracks;
        const bool wanted = (video ? e->settings.enable_video : e->settings.enable_audio) && (video ? vtrack : atrack) < 0 &&
                            seen == (video ? e->settings.video_track : e->settings.audio_track);
        seen++;
        if (!wanted) continue;
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Found requested %s track %u at webm track %u.", video ? "video" : "audio", seen - 1, t);
        if (ti.codec != (video ? EA1_WEBM_CODEC_AV1 : EA1_WEBM_CODEC_VORBIS)) {
            logf(e, EASYAV1_LOG_LEVEL_WARNING, video ? "Unsupported video codec found. Only AV1 codec is supported. Not displaying video."
                                                     : "Unsupported audio codec found. Only vorbis codec is supported. Not playing audio.");
            continue;
        }
        (video ? vtrack : atrack) = int(t);
        (video ? vinfo : ainfo) = ti;
    }
    if (vtrack >= 0 && !open_video(e, uint32_t(vtrack), vinfo)) return false;
    if (atrack >= 0 && !open_audio(e, uint32_t(atrack), ainfo)) return false;
    logf(e, EASYAV1_LOG_LEVEL_INFO, "Total video tracks: %u", e->video_tracks);
    logf(e, EASYAV1_LOG_LEVEL_INFO, "Total audio tracks: %u", e->audio_tracks);
    return true;
}

// ---- packets -----------------------------------------------------------------------------------
// the rule by which an audio packet enters the queue and with which timestamp (src/easyav1.c:1960-1972);
// the look-ahead reader applies the same rule so that both see the same sequence of audio frames
bool keep_audio_packet(const easyav1_t *e, easyav1_timestamp &ts)
{
    const int64_t off = e->audio.offset;
    if (off < 0 && uint64_t(-off) > ts) return false;
    if (off > 0 && ts + uint64_t(off) > e->duration) return false;
    ts += uint64_t(off);
    return true;
}

// ea1_webm_read_packet over the input of this handle.  With stream input (file / custom callbacks) the demuxer reads a mapping whose pages
// page_in() fills through the caller's callbacks; a failed seek or read leaves zeros there, which the demuxer may take for the end of the
// file or hand on as payload.  The reference turns a failed read into nestegg_read_packet < 0 and EASYAV1_STATUS_DECODER_ERROR
// (src/easyav1.c:1908-1913): so does this -- whatever the demuxer made of the zeros (round-4 advisor).
int read_packet_checked(easyav1_t *e, ea1_webm *w, ea1_webm_packet *pk)
{
    const int r = ea1_webm_read_packet(w, pk);
    if (e->lazy.base && __atomic_load_n(&e->lazy.failed, __ATOMIC_ACQUIRE)) return -1;
    return r;
}

// reads one packet of the file; queues it when it belongs to an active track.  false = nothing queued
bool fetch_packet(easyav1_t *e)
{
    QueuedPacket q;
    const int r = read_packet_checked(e, e->webm, &q.pk);
    if (r < 0) {
        // The reference reports it at once -- ten packets ahead of the position (src/easyav1.c:1908-1913).  This reader runs hundreds of packets
        // ahead (the GOP chains' look-ahead): reporting at once would throw away seconds of pictures that are already on their way, so the
        // failure ends the look-ahead and becomes the stream's status when the queues have run dry (end_of_packets)
        e->fetch_failed = true;
        e->all_fetched = true;
        return false;
    }
    if (r == 0) {
        e->all_fetched = true;
        return false;
    }
    const bool video = e->video.active && q.pk.track == e->video.track, audio = e->audio.active && q.pk.track == e->audio.track;
    if (!video && !audio) return false;
    q.ts = to_ms(e, q.pk.tstamp);
    if (audio && !keep_audio_packet(e, q.ts)) return false;
    if (q.ts < (video ? e->video_from : e->skip_before)) return false;   // a seek is in effect: this packet lies before its target
    q.keyframe = q.pk.keyframe == 1;
    q.first_chunk = 0;
    q.done = false;
    q.has_ticket = false;
    q.last_ticket = 0;
    q.frames = 0;
    if (audio) {
        q.first_chunk = e->chunks_queued;
        e->chunks_queued += q.pk.count;
    }
    (video ? e->vq : e->aq).push_back(q);
    if (video) {
        e->vq_keys += q.keyframe ? 1 : 0;
        for (uint32_t k = 0; k < q.pk.count; k++) e->vq_bytes += q.pk.size[k];
    }
    return true;
}

// what the REFERENCE reads ahead before it chooses the next packet (src/easyav1.c:2026-2072): something in a queue, ten video packets, and -- with
// the audio moved back -- audio up to the position.  These are never cut short by the burst limit of sync_queues (round-4 advisor: a video packet
// could be taken ahead of an audio packet with a lower shifted timestamp that had not been read yet).
bool reference_wants_more(const easyav1_t *e)
{
    if (e->vq.empty() && e->aq.empty()) return true;
    if (e->video.active && e->vq.size() <= size_t(kVideoPrefetch)) return true;
    if (e->audio.active && e->audio.offset < 0 && (e->aq.empty() || e->aq.back().ts < e->position)) return true;
    return false;
}

bool want_more_packets(const easyav1_t *e)
{
    if (e->vq.empty() && e->aq.empty()) return true;
    // the reference's decoder thread works 10 packets ahead; the GPU sink wants several GOPs in flight (easyav1_video_sink.h)
    if (e->video.active && e->vq.size() <= size_t(e->video.sink ? ea1::video_sink_lookahead(e->video.sink) : kVideoPrefetch)) return true;
    // The GPU sink decodes closed GOPs side by side: a look-ahead counted in packets holds too few of them when GOPs are long (a key frame
    // every 5 s at 60 frames/s: 3 GOPs in 1 024 packets).  Read on until every chain has a GOP to work on -- within a budget of packets and bytes.
    static const bool fixed_lookahead = getenv("EASYAV1_AMD_FIXED_LOOKAHEAD") != nullptr;   // (diagnostics: the packet count only)
    if (e->video.active && e->video.sink && !fixed_lookahead) {
        const size_t chains = size_t(ea1::video_sink_chains(e->video.sink));
        const uint64_t picture = uint64_t(e->video.width) * e->video.height * 3 / 2;
        if (chains > 1 && e->vq_keys < 2 * chains && e->vq.size() < kMaxLookaheadPackets && e->vq_bytes < kMaxLookaheadBytes &&
            uint64_t(e->vq.size()) * picture < kMaxLookaheadPictures)
            return true;
    }
    // audio is read ahead only when its timestamps were moved back (src/easyav1.c:2056-2072)
    if (e->audio.active && e->audio.offset < 0 && (e->aq.empty() || e->aq.back().ts < e->position)) return true;
    return false;
}

// both queues are empty and nothing more will be read: the end of the stream, or the failed read that cut it short
easyav1_status end_of_packets(easyav1_t *e)
{
    if (e->fetch_failed) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to read packet.");
        return EASYAV1_STATUS_ERROR;
    }
    std::lock_guard<std::mutex> g(e->info);
    e->status = EASYAV1_STATUS_FINISHED;
    return EASYAV1_STATUS_OK;
}

easyav1_status sync_queues(easyav1_t *e)
{
    if (e->status == EASYAV1_STATUS_FINISHED) return EASYAV1_STATUS_FINISHED;
    if (e->synced || e->all_fetched) return EASYAV1_STATUS_OK;
    // The GPU sink's look-ahead is hundreds of packets (several GOP chains): it fills over several calls, kFetchBurst packets at a time, and
    // while the caller waits for a picture anyway (decode_packet) -- read in one go before the first picture it cost a 4K stream 230 ms
    // (profiles/r04_af_first_frame_trace.txt).  The reference's 10 packets fit one burst.
    // (counted in packets QUEUED: after a seek without Cues thousands of packets before the target are read and dropped)
    const size_t before = e->vq.size() + e->aq.size();
    while (!e->all_fetched && want_more_packets(e) && (e->vq.size() + e->aq.size() < before + size_t(kFetchBurst) || reference_wants_more(e)))
        if (!fetch_packet(e) && is_error(e->status)) return EASYAV1_STATUS_ERROR;
    e->synced = e->all_fetched || !want_more_packets(e);
    if (e->vq.empty() && e->aq.empty() && e->all_fetched) return end_of_packets(e);
    return EASYAV1_STATUS_OK;
}

// the packet to decode next: the earlier of the two queue heads, video first on a tie (src/easyav1.c:2103-2131)
easyav1_status decode_packet(easyav1_t *e, QueuedPacket &q, bool is_video);
easyav1_status submit_video_packet(easyav1_t *e, QueuedPacket &q, bool its_turn = false);

QueuedPacket *next_packet(easyav1_t *e, bool *is_video)
{
    if (e->status == EASYAV1_STATUS_FINISHED || sync_queues(e) != EASYAV1_STATUS_OK) return nullptr;
    if (e->vq.empty() && e->aq.empty()) return nullptr;
    // the prefetched video packets go to the decoder now: their pictures are reconstructed on the GPU while the
    // caller is still busy with earlier ones (the reference's decoder thread runs ahead the same way)
    for (QueuedPacket &q : e->vq)
        if (!q.done && submit_video_packet(e, q) == EASYAV1_STATUS_ERROR) return nullptr;
    *is_video = !e->vq.empty() && (e->aq.empty() || e->vq.front().ts <= e->aq.front().ts);
    return *is_video ? &e->vq.front() : &e->aq.front();
}

// Stream input: the pages of the mapping that lie before every queued packet go back to the system (the reference reads through small
// buffers: its memory does not grow with the file).  Whoever needs them again -- a seek backwards -- reads them again through the callbacks.
void drop_read_pages(easyav1_t *e)
{
    auto &lz = e->lazy;
    if (!lz.base) return;
    constexpr size_t P = easyav1_t::LazyFile::kPage;
    const uint8_t *low = nullptr;
    for (const std::deque<QueuedPacket> *q : {&e->vq, &e->aq})
        if (!q->empty())
            for (uint32_t k = 0; k < q->front().pk.count; k++)
                if (!low || q->front().pk.data[k] < low) low = q->front().pk.data[k];
    if (!low || low < lz.base || low >= lz.base + lz.size) return;
    const size_t upto = size_t(low - lz.base) / P;                     // pages [0, upto) hold nothing that is queued
    const size_t from = std::max(lz.dropped, easyav1_t::LazyFile::kKeepFront);
    if (upto < from + 16) return;                                      // a megabyte at a time
    std::lock_guard<std::mutex> g(lz.mu);
    for (size_t k = from; k < upto; k++) {
        lz.pages_dropped += lz.have[k] ? 1 : 0;
        __atomic_store_n(&lz.have[k], uint8_t(0), __ATOMIC_RELEASE);
    }
    (void)madvise(lz.base + from * P, (upto - from) * P, MADV_DONTNEED);
    lz.dropped = upto;
}

void release_packet(easyav1_t *e, bool is_video)
{
    if (is_video) {
        const QueuedPacket &q = e->vq.front();
        e->vq_keys -= q.keyframe ? 1 : 0;
        for (uint32_t k = 0; k < q.pk.count; k++) e->vq_bytes -= q.pk.size[k];
    }
    (is_video ? e->vq : e->aq).pop_front();
    drop_read_pages(e);
    if (!e->all_fetched) {
        e->synced = false;
    } else if (e->vq.empty() && e->aq.empty()) {
        (void)end_of_packets(e);
    }
}

// ---- audio ---------------------------------------------------------------------------------------
// Decodes the next kAudioLookahead kept audio frames of the track in ONE call of the GPU path
// (a launch per packet would be launch latency and nothing else: a 2048-sample block is 8 KiB).
// The look-ahead reader walks the same bytes as the packet reader, later packets only.
bool fill_audio_cache(easyav1_t *e)
{
    std::vector<const uint8_t *> ptrs;
    std::vector<size_t> sizes;
    while (!e->audio.ahead_done && int(ptrs.size()) < kAudioLookahead) {
        ea1_webm_packet pk;
        const int r = read_packet_checked(e, e->ahead, &pk);
        if (r < 0) {
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to read packet.");
            return false;
        }
        if (r == 0) {
            e->audio.ahead_done = true;
            break;
        }
        if (pk.track != e->audio.track) continue;
        easyav1_timestamp ts = to_ms(e, pk.tstamp);
        if (!keep_audio_packet(e, ts) || ts < e->skip_before) continue;
        for (uint32_t k = 0; k < pk.count; k++) {
            ptrs.push_back(pk.data[k]);
            sizes.push_back(pk.size[k]);
        }
    }
    if (ptrs.empty()) return true;
    const int n = int(ptrs.size());
    int b0 = 0, b1 = 0, ch = 0;
    long rate = 0;
    ea1_vorbis_decoder_info(e->audio.dec, &ch, &rate, &b0, &b1);
    const size_t cap = size_t(n) * size_t(b1 / 2) + 1;
    std::vector<float> pcm(size_t(ch) * cap);
    std::vector<int> counts(size_t(n), 0);
    std::vector<uint8_t> accepted(size_t(n), 0);
    int64_t total = 0;
    if (ea1_vorbis_decoder_decode_checked(e->audio.dec, ptrs.data(), sizes.data(), n, pcm.data(), cap, counts.data(), &total, accepted.data(), nullptr) != EA1_OK) {
        logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to process audio packet.");
        return false;
    }
    size_t at = 0;
    for (int k = 0; k < n; k++) {
        DecodedChunk c;
        c.count = counts[size_t(k)];
        c.accepted = accepted[size_t(k)] != 0;
        c.pcm.resize(size_t(ch) * size_t(c.count));
        for (int cc = 0; cc < ch; cc++) memcpy(c.pcm.data() + size_t(cc) * c.count, pcm.data() + size_t(cc) * cap + at, size_t(c.count) * sizeof(float));
        at += size_t(c.count);
        e->audio.cache.push_back(std::move(c));
    }
    return true;
}

// one audio frame of a packet: src/easyav1.c:2434-2499 with the PCM taken from the look-ahead cache
easyav1_status decode_audio_chunk(easyav1_t *e, const QueuedPacket &q, uint64_t chunk)
{
    if (!e->audio.pending) {
        ea1_audio_ring_begin_packet(e->audio.ring);
        e->audio.frame.samples = 0;
        e->audio.frame.timestamp = q.ts;
    }
    while (e->audio.cache_first + e->audio.cache.size() <= chunk) {
        const size_t before = e->audio.cache.size();
        if (!fill_audio_cache(e)) return EASYAV1_STATUS_ERROR;
        if (e->audio.cache.size() == before) {   // the readers disagree: cannot happen on an unchanged buffer
            set_error(e, EASYAV1_STATUS_INVALID_STATE, "Audio look-ahead lost track of the packet sequence.");
            return EASYAV1_STATUS_ERROR;
        }
    }
    while (e->audio.cache_first < chunk) {
        e->audio.cache.pop_front();
        e->audio.cache_first++;
    }
    const DecodedChunk &c = e->audio.cache.front();
    if (!c.accepted) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to process audio packet.");
        return EASYAV1_STATUS_ERROR;
    }
    if (c.count > 0) {
        ea1_audio_ring_push(e->audio.ring, c.pcm.data(), size_t(c.count), c.count);
        e->audio.pending = true;
    }
    return EASYAV1_STATUS_OK;
}

// ---- video ---------------------------------------------------------------------------------------
// a sequence header read from the track: the newest one, and its number among the distinct ones seen (an encoder repeats its header with every key frame)
void take_sequence_header(easyav1_t *e, const ea1_av1_sequence_header &seq)
{
    auto &v = e->video;
    uint32_t k = 0;
    while (k < v.seqs.size() && memcmp(&v.seqs[k], &seq, sizeof seq)) k++;
    if (k == v.seqs.size()) v.seqs.push_back(seq);
    v.seq_tag = k;
    if (!v.have_seq) fill_frame_properties(e, seq);   // (until the first picture is there)
    v.seq = seq;
    v.have_seq = true;
}

void fill_frame_properties(easyav1_t *e, const ea1_av1_sequence_header &s)
{
    auto &p = e->video.frame.properties;
    static const easyav1_pixel_layout layouts[4] = {EASYAV1_PIXEL_LAYOUT_YUV400, EASYAV1_PIXEL_LAYOUT_YUV420, EASYAV1_PIXEL_LAYOUT_YUV422, EASYAV1_PIXEL_LAYOUT_YUV444};
    p.pixel_layout = s.layout < 4 ? layouts[s.layout] : EASYAV1_PIXEL_LAYOUT_UNKNOWN;
    p.bits_per_color = s.hbd == 0 ? EASYAV1_BITS_PER_COLOR_8 : (s.hbd == 1 ? EASYAV1_BITS_PER_COLOR_10 : (s.hbd == 2 ? EASYAV1_BITS_PER_COLOR_12 : EASYAV1_BITS_PER_COLOR_UNKNOWN));
    p.color_space = s.color_range ? EASYAV1_COLOR_SPACE_FULL : EASYAV1_COLOR_SPACE_LIMITED;
    // the three colour description codes share their numbering with the AV1 syntax (src/easyav1.c:3340-3490 maps them one to one)
    p.color_primaries = s.color_primaries <= 12 && s.color_primaries != 3 ? easyav1_color_primaries(s.color_primaries < 3 ? s.color_primaries : s.color_primaries - 1) : EASYAV1_COLOR_PRIMARIES_UNSPECIFIED;
    p.transfer_characteristics = s.transfer_characteristics <= 18 && s.transfer_characteristics != 3 ? easyav1_transfer_characteristics(s.transfer_characteristics < 3 ? s.transfer_characteristics : s.transfer_characteristics - 1)
                                                                                                       : EASYAV1_TRANSFER_CHARACTERISTICS_UNSPECIFIED;
    static const easyav1_matrix_coefficients mats[15] = {
        EASYAV1_MATRIX_COEFFICIENTS_IDENTITY, EASYAV1_MATRIX_COEFFICIENTS_BT709, EASYAV1_MATRIX_COEFFICIENTS_UNKNOWN, EASYAV1_MATRIX_COEFFICIENTS_UNSPECIFIED,
        EASYAV1_MATRIX_COEFFICIENTS_FCC, EASYAV1_MATRIX_COEFFICIENTS_BT470BG, EASYAV1_MATRIX_COEFFICIENTS_BT601, EASYAV1_MATRIX_COEFFICIENTS_SMPTE240,
        EASYAV1_MATRIX_COEFFICIENTS_SMPTE_YCGCO, EASYAV1_MATRIX_COEFFICIENTS_BT2020_NCL, EASYAV1_MATRIX_COEFFICIENTS_BT2020_CL, EASYAV1_MATRIX_COEFFICIENTS_SMPTE2085,
        EASYAV1_MATRIX_COEFFICIENTS_CHROMATICITY_NCL, EASYAV1_MATRIX_COEFFICIENTS_CHROMATICITY_CL, EASYAV1_MATRIX_COEFFICIENTS_ICTCP};
    p.matrix_coefficients = s.matrix_coefficients < 15 ? mats[s.matrix_coefficients] : EASYAV1_MATRIX_COEFFICIENTS_UNSPECIFIED;
    p.chroma_sample_position = s.chroma_sample_position == 1 ? EASYAV1_CHROMA_SAMPLE_POSITION_VERTICAL
                                                              : (s.chroma_sample_position == 2 ? EASYAV1_CHROMA_SAMPLE_POSITION_COLOCATED : EASYAV1_CHROMA_SAMPLE_POSITION_UNKNOWN);
}

// one temporal unit: headers through the OBU parser; parsed-frame payloads QUEUED on the GPU path (the sink's threads take it
// from there; decode_packet waits for the result when the packet's turn has come)
easyav1_status submit_video_chunk(easyav1_t *e, QueuedPacket &q, const uint8_t *data, size_t size)
{
    size_t pos = 0;
    int plain_frames = 0;
    while (pos < size) {
        ea1_obu obu;
        if (ea1_av1_next_obu(data + pos, size - pos, &obu) != EA1_OK) {
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to decode video packet: malformed OBU.");
            return EASYAV1_STATUS_ERROR;
        }
        if (obu.type == 1) {   // OBU_SEQUENCE_HEADER
            ea1_av1_sequence_header seq;
            memset(&seq, 0, sizeof seq);
            if (ea1_av1_parse_sequence_header(data + pos, obu.total_size, &seq) != EA1_OK) {   // takes whole OBUs
                set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to decode video packet: bad sequence header.");
                return EASYAV1_STATUS_ERROR;
            }
            take_sequence_header(e, seq);
        } else if (obu.type == 5 && ea1::is_parsed_frame_metadata(obu.payload, obu.payload_size)) {
            if (!e->video.sink && ea1::video_sink_create(&e->video.sink, e->video.width, e->video.height) != EA1_OK) {
                logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
                set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to initialize the GPU reconstruction path.");
                return EASYAV1_STATUS_ERROR;
            }
            // frames between the key frame a seek went back to and its target are decoded for their successors' sake only
            if (e->video.sink_tag != e->video.seq_tag) ea1::video_sink_set_tag(e->video.sink, e->video.sink_tag = e->video.seq_tag);
            if (ea1::video_sink_submit(e->video.sink, obu.payload, obu.payload_size, q.ts, q.keyframe && q.frames == 0, q.ts < e->skip_before, &q.last_ticket) != EA1_OK) {
                logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
                set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to reconstruct a parsed frame.");
                return EASYAV1_STATUS_ERROR;
            }
            q.has_ticket = true;
            q.frames++;
        } else if (obu.type == 3 || obu.type == 6) {   // OBU_FRAME_HEADER / OBU_FRAME
            plain_frames++;
        }
        pos += obu.total_size;
    }
    // a plain AV1 frame: the whole unit goes to the sink's block-level parser when its tables were given (EASYAV1_AMD_AV1_TABLES:
    // the library owns no default CDFs / scan orders / quantiser tables, include/easyav1_amd_obu.h)
    if (plain_frames && q.frames == 0) {
        if (!e->video.sink && ea1::video_sink_create(&e->video.sink, e->video.width, e->video.height) != EA1_OK) {
            logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to initialize the GPU reconstruction path.");
            return EASYAV1_STATUS_ERROR;
        }
        if (ea1::video_sink_parses_av1(e->video.sink)) {
            if (e->video.sink_tag != e->video.seq_tag) ea1::video_sink_set_tag(e->video.sink, e->video.sink_tag = e->video.seq_tag);
            if (ea1::video_sink_submit_av1(e->video.sink, data, size, q.ts, q.keyframe, q.ts < e->skip_before, &q.last_ticket) != EA1_OK) {
                logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
                set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to queue an AV1 frame.");
                return EASYAV1_STATUS_ERROR;
            }
            q.has_ticket = true;
            q.frames++;
        } else if (!e->video.warned) {
            e->video.warned = true;
            logf(e, EASYAV1_LOG_LEVEL_WARNING, "AV1 tile data needs the specification's entropy tables, which this library does not own: name a table "
                                               "blob in EASYAV1_AMD_AV1_TABLES (include/easyav1_amd_obu.h).  The track's headers are read, its pictures are not produced.");
        }
    }
    return EASYAV1_STATUS_OK;
}

int context_layout(const ea1_av1_sequence_header &seq)
{
    return seq.layout == EA1_LAYOUT_I444 ? EA1_LAYOUT_444 : (seq.layout == EA1_LAYOUT_I422 ? EA1_LAYOUT_422 : EA1_LAYOUT_420);
}

// does the packet bring a sequence header that the sink's context cannot hold?  (plain AV1 only: parsed-frame packets name their size themselves)
bool needs_another_context(easyav1_t *e, const QueuedPacket &q, ea1_av1_sequence_header *seq)
{
    if (!e->video.sink || !ea1::video_sink_parses_av1(e->video.sink)) return false;
    for (uint32_t k = 0; k < q.pk.count; k++) {
        const uint8_t *data = q.pk.data[k];
        const size_t size = q.pk.size[k];
        for (size_t pos = 0; pos < size;) {
            ea1_obu obu;
            if (ea1_av1_next_obu(data + pos, size - pos, &obu) != EA1_OK) return false;   // (the submission reports it)
            if (obu.type == 1) {
                memset(seq, 0, sizeof *seq);
                if (ea1_av1_parse_sequence_header(data + pos, obu.total_size, seq) != EA1_OK) return false;
                if (!seq->hbd && !ea1::video_sink_fits(e->video.sink, seq->max_width, seq->max_height, context_layout(*seq))) return true;
            }
            pos += obu.total_size;
        }
    }
    return false;
}

easyav1_status submit_video_packet(easyav1_t *e, QueuedPacket &q, bool its_turn)
{
    if (q.done) return EASYAV1_STATUS_OK;
    if (e->video.barrier && !its_turn) return EASYAV1_STATUS_OK;   // (packets are queued in order: nothing passes the one that waits)
    ea1_av1_sequence_header seq;
    if (needs_another_context(e, q, &seq)) {
        if (!its_turn) {
            e->video.barrier = true;
            return EASYAV1_STATUS_OK;
        }
        // every packet in front of this one has been waited for: the sink is idle and holds only pictures nobody took yet
        logf(e, EASYAV1_LOG_LEVEL_INFO, "A sequence header of %ux%u (layout %d) does not fit the decoder context: a new one is made.", unsigned(seq.max_width), unsigned(seq.max_height), int(seq.layout));
        if (e->video.old_sink) ea1::video_sink_destroy(e->video.old_sink);
        e->video.old_sink = e->video.sink;
        e->video.sink = nullptr;
        e->video.sink_tag = 0;   // (a new sink starts at tag 0)
        if (ea1::video_sink_create(&e->video.sink, e->video.width, e->video.height) != EA1_OK) {
            logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to initialize the GPU reconstruction path.");
            return EASYAV1_STATUS_ERROR;
        }
        ea1::video_sink_declare(e->video.sink, seq.max_width, seq.max_height, context_layout(seq));
    }
    e->video.barrier = false;
    q.done = true;
    for (uint32_t k = 0; k < q.pk.count; k++)
        if (submit_video_chunk(e, q, q.pk.data[k], q.pk.size[k]) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
    return EASYAV1_STATUS_OK;
}

easyav1_status decode_packet(easyav1_t *e, QueuedPacket &q, bool is_video)
{
    if (is_video) {
        if (submit_video_packet(e, q, true) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
        // the packet's turn has come: wait until its pictures are there, as the reference waits for its decoder thread
        // (src/easyav1.c:2568-2585 "while (packet->decoded == EASYAV1_FALSE)")
        if (q.has_ticket) {
            // until its pictures are there the look-ahead goes on filling (q stays valid: a deque keeps references across push_back)
            while (!e->all_fetched && want_more_packets(e) && !ea1::video_sink_ready(e->video.sink, q.last_ticket)) {
                for (int k = 0; k < 8 && !e->all_fetched && want_more_packets(e); k++)
                    if (!fetch_packet(e) && is_error(e->status)) return EASYAV1_STATUS_ERROR;
                for (QueuedPacket &p : e->vq)
                    if (!p.done && submit_video_packet(e, p) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
            }
            if (ea1::video_sink_wait(e->video.sink, q.last_ticket) != EA1_OK) {
                logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
                set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to reconstruct a parsed frame.");
                return EASYAV1_STATUS_ERROR;
            }
            e->video.processed += q.frames;
        }
        e->video.any_decoded = true;
        e->video.decoded_ts = q.ts;
        return EASYAV1_STATUS_OK;
    }
    for (uint32_t k = 0; k < q.pk.count; k++)
        if (decode_audio_chunk(e, q, q.first_chunk + k) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
    return EASYAV1_STATUS_OK;
}

void run_callbacks(easyav1_t *e)
{
    if (e->settings.callbacks.video)
        if (const easyav1_video_frame *f = easyav1_get_video_frame(e)) e->settings.callbacks.video(f, e->settings.callbacks.userdata);
    if (e->settings.callbacks.audio)
        if (const easyav1_audio_frame *f = easyav1_get_audio_frame(e)) e->settings.callbacks.audio(f, e->settings.callbacks.userdata);
}

// ---- seeking: back to the first cluster, everything before the target is dropped ----------------
easyav1_status do_seek(easyav1_t *e, easyav1_timestamp ts, bool fast)
{
    if (is_error(e->status)) return EASYAV1_STATUS_ERROR;
    if (ts > e->duration) ts = e->duration;
    // where reading starts again: the Cluster of the last cue point at or before the target when the file has Cues (nestegg_track_seek,
    // src/easyav1.c:2942,:3026 -- O(1) clusters touched), else the first cluster (the block headers are walked from there)
    const uint32_t cue_track = e->video.active ? e->video.track : e->audio.track;
    // -> 1: positioned by a cue point, 2: at the first cluster, 0: failed
    auto reposition = [&](ea1_webm *w, bool by_cues, easyav1_timestamp to) {
        if (!w) return 2;
        if (by_cues && (e->video.active || e->audio.active) && ea1_webm_seek(w, cue_track, uint64_t(to) * 1000000ull) == EA1_OK) return 1;
        return ea1_webm_rewind(w) == EA1_OK ? 2 : 0;
    };
    int how = reposition(e->webm, true, ts);
    if (!how) {
        set_error(e, EASYAV1_STATUS_IO_ERROR, "Failed to seek.");
        return EASYAV1_STATUS_ERROR;
    }
    e->vq.clear();
    e->aq.clear();
    e->video.barrier = false;   // (the packet that waited is gone with the queue)
    e->vq_keys = e->vq_bytes = 0;
    e->lazy.dropped = 0;   // (pages before the new position may be read again: they are candidates again)
    e->all_fetched = e->synced = false;
    e->fetch_failed = false;   // (a seek reads again: the failure, if it lasts, is met again)
    e->chunks_queued = 0;
    e->video_from = ts;
    if (e->video.active) {
        // pictures need their references: decoding restarts at the last key frame at or before the target (the reference finds it
        // through the Cues and decodes forward, src/easyav1.c:2961-3216; this reader walks the blocks from the cue point's cluster, or from the
        // first cluster when the file has no Cues -- or when no key frame lies between the cue point and the target: an index that names a cluster
        // without one, round-4 advisor; the reference steps back a cue point at a time, :3133-3150)
        easyav1_timestamp key = 0;
        bool found = false;
        for (int attempt = 0; attempt < 2 && !found; attempt++) {
            if (attempt == 1) {
                if (how != 1) break;          // the scan started at the first cluster already
                how = reposition(e->webm, false, ts);
                if (!how) {
                    set_error(e, EASYAV1_STATUS_IO_ERROR, "Failed to seek.");
                    return EASYAV1_STATUS_ERROR;
                }
            }
            for (;;) {
                ea1_webm_packet pk;
                const int r = read_packet_checked(e, e->webm, &pk);
                if (r < 0) {
                    set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to read packet.");
                    return EASYAV1_STATUS_ERROR;
                }
                if (r == 0) break;
                const easyav1_timestamp pts = to_ms(e, pk.tstamp);
                if (pts > ts) break;
                if (pk.track == e->video.track && pk.keyframe == 1) {
                    key = pts;
                    found = true;
                }
                if (pk.track == e->video.track && !e->video.have_seq)
                    for (uint32_t k = 0; k < pk.count && !e->video.have_seq; k++) scan_sequence_header(e, pk.data[k], pk.size[k]);
            }
        }
        if (found) e->video_from = key;
        // fast seeking (settings.use_fast_seeking, src/easyav1.c:3106-3111): playing resumes AT that key frame instead of the frames
        // up to the target being decoded and thrown away
        if (found && fast) ts = key;
        // back to where reading starts: by the cue point again when the key frame was found behind it, else from the first cluster
        if (!reposition(e->webm, how == 1, found ? key : ts)) {
            set_error(e, EASYAV1_STATUS_IO_ERROR, "Failed to seek.");
            return EASYAV1_STATUS_ERROR;
        }
    }
    // the look-ahead reader of the audio starts where the demuxer does (the same cue point / the first cluster: both drop what lies before the target)
    if (!reposition(e->ahead, how == 1, e->video.active ? e->video_from : ts)) {
        set_error(e, EASYAV1_STATUS_IO_ERROR, "Failed to seek.");
        return EASYAV1_STATUS_ERROR;
    }
    e->skip_before = ts;
    if (e->video_from > ts) e->video_from = ts;
    e->audio.cache.clear();
    e->audio.cache_first = 0;
    e->audio.ahead_done = false;
    e->audio.pending = false;
    if (e->audio.dec) ea1_vorbis_decoder_restart(e->audio.dec);   // vorbis_synthesis_restart, src/easyav1.c:3077
    if (e->audio.ring) ea1_audio_ring_get(e->audio.ring, nullptr, nullptr);
    if (e->video.sink) ea1::video_sink_flush(e->video.sink);
    if (e->video.old_sink) ea1::video_sink_destroy(e->video.old_sink);   // (a retired sink holds pictures of before the seek only)
    e->video.old_sink = nullptr;
    {
        std::lock_guard<std::mutex> g(e->info);
        e->position = ts;
        e->status = EASYAV1_STATUS_OK;
    }
    // catch up from the key frame to the target inside the seek, as the reference does (SEEKING_FOR_TIMESTAMP decodes up to the
    // target before the seek returns, src/easyav1.c:3140-3216): the pictures on the way are reconstructed and thrown away
    for (;;) {
        if (sync_queues(e) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
        if (e->status == EASYAV1_STATUS_FINISHED || e->vq.empty() || e->vq.front().ts >= ts) break;
        for (QueuedPacket &q : e->vq)
            if (!q.done && submit_video_packet(e, q) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
        const uint64_t before = e->video.processed;
        if (decode_packet(e, e->vq.front(), true) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
        e->video.processed = before;   // not frames the caller asked for
        release_packet(e, true);
    }
    e->video.any_decoded = false;
    return EASYAV1_STATUS_OK;
}

easyav1_status decode_until(easyav1_t *e, easyav1_timestamp ts)
{
    if (is_error(e->status)) return EASYAV1_STATUS_ERROR;
    if (e->status == EASYAV1_STATUS_FINISHED) return EASYAV1_STATUS_FINISHED;
    if (ts <= e->position) return EASYAV1_STATUS_OK;
    if (e->settings.skip_unprocessed_frames && ts - e->position > kSkipMs && !e->playback.seek_requested.load()) {
        // too far behind (src/easyav1.c:2693-2713): when a key frame lies between the position and the target, go on from the last
        // such key frame (a fast seek) instead of decoding everything in between; without video: straight to the target
        bool jump = !e->video.active && !e->audio.active;
        if (e->video.active) {
            easyav1_timestamp key = 0;
            bool found = false;
            ea1_webm *scan = nullptr;
            if (open_reader(e, &scan) == EA1_OK) {
                // (with Cues: from the last cue point at or before the target instead of from the first cluster)
                (void)ea1_webm_seek(scan, e->video.track, uint64_t(ts) * 1000000ull);
                for (;;) {
                    ea1_webm_packet pk;
                    if (read_packet_checked(e, scan, &pk) <= 0) break;   // (a failed read: the decode that follows meets it and reports it)
                    const easyav1_timestamp pts = to_ms(e, pk.tstamp);
                    if (pts > ts) break;
                    if (pk.track == e->video.track && pk.keyframe == 1 && pts > e->position) {
                        key = pts;
                        found = true;
                    }
                }
                ea1_webm_destroy(scan);
            }
            (void)key;
            jump = found;
        }
        if (jump && do_seek(e, ts, true) == EASYAV1_STATUS_ERROR) return EASYAV1_STATUS_ERROR;
    }
    easyav1_status status = easyav1_status(e->status);
    while (status == EASYAV1_STATUS_OK) {
        bool is_video = false;
        QueuedPacket *q = next_packet(e, &is_video);
        if (e->status == EASYAV1_STATUS_FINISHED) {
            status = EASYAV1_STATUS_FINISHED;
            break;
        }
        if (!q) return EASYAV1_STATUS_ERROR;
        if (e->playback.pause.load() || e->playback.seek_requested.load()) return status;
        if (q->ts >= ts) break;
        {
            std::lock_guard<std::mutex> g(e->info);
            e->position = q->ts;
        }
        status = decode_packet(e, *q, is_video);
        release_packet(e, is_video);
    }
    if (status == EASYAV1_STATUS_OK) {
        std::lock_guard<std::mutex> g(e->info);
        e->position = ts;
    }
    if (status != EASYAV1_STATUS_ERROR) run_callbacks(e);
    return status;
}

uint64_t ticks_ms()
{
    return uint64_t(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}

// wall-clock paced decoding (src/easyav1.c:2805-2853)
void playback_loop(easyav1_t *e)
{
    uint64_t last = ticks_ms(), now = last;
    for (;;) {
        {
            std::lock_guard<std::mutex> g(e->playback.mutex);
            std::lock_guard<std::recursive_mutex> st(e->state);
            if (!e->playback.active.load() || e->playback.pause.load()) break;
            if (decode_until(e, e->position + (now - last)) == EASYAV1_STATUS_ERROR) break;
        }
        if (now == last) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        last = now;
        now = ticks_ms();
        if (e->playback.seek_requested.exchange(false)) {
            std::lock_guard<std::recursive_mutex> st(e->state);
            if (do_seek(e, e->playback.seek_to.load(), e->settings.use_fast_seeking != EASYAV1_FALSE) == EASYAV1_STATUS_ERROR) break;
            last = now = ticks_ms();
        }
    }
}

// reads a caller-supplied stream into memory: the reader of this repository works on a flat buffer
// the size of the stream and a mapping of that size, nothing read yet
bool map_stream(easyav1_t *e, const easyav1_stream &st)
{
    if (st.seek_func(0, SEEK_END, st.userdata) != 0) return false;
    const int64_t n = st.tell_func(st.userdata);
    if (n <= 0 || st.seek_func(0, SEEK_SET, st.userdata) != 0) return false;
    void *m = mmap(nullptr, size_t(n), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return false;
    e->lazy.base = static_cast<uint8_t *>(m);
    e->lazy.size = size_t(n);
    e->lazy.st = st;
    e->lazy.have.assign((size_t(n) + easyav1_t::LazyFile::kPage - 1) / easyav1_t::LazyFile::kPage, 0);
    return true;
}

// ea1_webm_open_paged's callback: bring [off, off + len) in -- the missing pages of the range, neighbouring ones with one read
void page_in(void *ud, size_t off, size_t len)
{
    auto &lz = static_cast<easyav1_t *>(ud)->lazy;
    constexpr size_t P = easyav1_t::LazyFile::kPage;
    if (!len || off >= lz.size) return;
    const size_t first = off / P, last = std::min(lz.size - 1, off + len - 1) / P;
    {
        bool all = true;   // (the common case, without the lock: pages only ever appear)
        for (size_t k = first; k <= last && all; k++) all = __atomic_load_n(&lz.have[k], __ATOMIC_ACQUIRE) != 0;
        if (all) return;
    }
    std::lock_guard<std::mutex> g(lz.mu);
    for (size_t k = first; k <= last;) {
        if (lz.have[k]) {
            k++;
            continue;
        }
        size_t k1 = k;
        while (k1 <= last && !lz.have[k1]) k1++;
        const size_t a = k * P, b = std::min(lz.size, k1 * P);
        if (lz.failed || lz.st.seek_func(int64_t(a), SEEK_SET, lz.st.userdata) != 0 || lz.st.read_func(lz.base + a, b - a, lz.st.userdata) != 1) {
            __atomic_store_n(&lz.failed, true, __ATOMIC_RELEASE);   // the bytes stay zero; read_packet_checked() turns whatever the demuxer makes of them into an error
            memset(lz.base + a, 0, b - a);
        } else {
            lz.bytes_read += b - a, lz.reads++;
        }
        for (size_t j = k; j < k1; j++) __atomic_store_n(&lz.have[j], uint8_t(1), __ATOMIC_RELEASE);
        k = k1;
    }
}

int open_reader(easyav1_t *e, ea1_webm **out)
{
    return e->lazy.base ? ea1_webm_open_paged(out, e->data, e->size, page_in, e) : ea1_webm_open_memory(out, e->data, e->size);
}

// The GPU path is opened inside easyav1_init_*, where the reference opens dav1d (src/easyav1.c:1405): the HIP runtime (~200 ms when this is
// the process's first use of it), the decoder context, and -- for a packet in parsed form -- pinned buffers.  A plain AV1 track is opened
// for the largest frame its sequence header allows.  A failure here is not fatal: the first decode call reports it.
void open_gpu_path_early(easyav1_t *e)
{
    if (!e->video.active || e->vq.empty() || e->video.sink) return;
    const QueuedPacket &q = e->vq.front();
    if (!q.pk.count) return;
    const uint8_t *data = q.pk.data[0];
    const size_t size = q.pk.size[0];
    bool plain = false;
    for (size_t pos = 0; pos < size;) {
        ea1_obu obu;
        if (ea1_av1_next_obu(data + pos, size - pos, &obu) != EA1_OK) return;
        if (obu.type == 3 || obu.type == 6) plain = true;   // OBU_FRAME_HEADER / OBU_FRAME
        if (obu.type == 5 && ea1::is_parsed_frame_metadata(obu.payload, obu.payload_size)) {
            ea1_frame_desc d;
            if (ea1_ir_packet_peek(obu.payload, obu.payload_size, &d) != EA1_OK) return;
            if (ea1::video_sink_create(&e->video.sink, e->video.width, e->video.height) != EA1_OK) return;
            (void)ea1::video_sink_open(e->video.sink, d.width, d.height, obu.payload_size, d.layout);
            return;
        }
        pos += obu.total_size;
    }
    if (plain && e->video.have_seq) {
        if (ea1::video_sink_create(&e->video.sink, e->video.width, e->video.height) != EA1_OK) return;
        if (!ea1::video_sink_parses_av1(e->video.sink)) return;
        // (the track's FIRST sequence header: the packets read ahead may have brought another one already)
        const ea1_av1_sequence_header first_seq = e->video.seqs.empty() ? e->video.seq : e->video.seqs[0];
        const int layout = first_seq.layout == EA1_LAYOUT_I444 ? EA1_LAYOUT_444 : (first_seq.layout == EA1_LAYOUT_I422 ? EA1_LAYOUT_422 : EA1_LAYOUT_420);
        ea1::video_sink_declare(e->video.sink, first_seq.max_width, first_seq.max_height, layout);
        // the packets read so far go to the sink's parser threads now: they parse the first key frame while this thread makes the context
        for (QueuedPacket &p : e->vq)
            if (!p.done && submit_video_packet(e, p) == EASYAV1_STATUS_ERROR) return;
        (void)ea1::video_sink_open(e->video.sink, (first_seq.max_width + 7u) & ~7u, (first_seq.max_height + 7u) & ~7u, 0, layout);
    }
}

easyav1_t *open_common(easyav1_t *e)
{
    if (open_reader(e, &e->webm) != EA1_OK) {
        logf(e, EASYAV1_LOG_LEVEL_ERROR, "%s", ea1_last_error());
        set_error(e, EASYAV1_STATUS_INVALID_STATE, "Failed to initialize webm context");
        easyav1_destroy(&e);
        return nullptr;
    }
    uint64_t dur_ns = 0;
    if (ea1_webm_duration(e->webm, &dur_ns) != EA1_OK) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get duration");
        easyav1_destroy(&e);
        return nullptr;
    }
    if (ea1_webm_tstamp_scale(e->webm, &e->time_scale) != EA1_OK || e->time_scale == 0) {
        set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Time scale is 0.");
        easyav1_destroy(&e);
        return nullptr;
    }
    e->duration = to_ms(e, dur_ns);
    logf(e, EASYAV1_LOG_LEVEL_INFO, "File duration: %llu minutes and %llu seconds.", (unsigned long long)(e->duration / 60000), (unsigned long long)((e->duration / 1000) % 60));
    if (!open_tracks(e) || (e->audio.active && open_reader(e, &e->ahead) != EA1_OK) || sync_queues(e) != EASYAV1_STATUS_OK) {
        easyav1_destroy(&e);
        return nullptr;
    }
    open_gpu_path_early(e);
    return e;
}

easyav1_t *new_instance(const easyav1_settings *settings)
{
    auto *e = new (std::nothrow) easyav1_t();
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Failed to allocate memory for the easyav1 handle");
        return nullptr;
    }
    e->settings = settings ? *settings : kDefaults;
    if (!settings) logf(e, EASYAV1_LOG_LEVEL_INFO, "No settings provided, using default settings.");
    return e;
}

}  // namespace

// ================================================================================================
extern "C" {

easyav1_settings easyav1_default_settings(void) { return kDefaults; }

easyav1_t *easyav1_init_from_custom_stream(const easyav1_stream *stream, const easyav1_settings *settings)
{
    if (!stream || !stream->read_func || !stream->seek_func || !stream->tell_func) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Stream is NULL or missing read, seek or tell functions");
        return nullptr;
    }
    easyav1_t *e = new_instance(settings);
    if (!e) return nullptr;
    if (!map_stream(e, *stream)) {
        set_error(e, EASYAV1_STATUS_IO_ERROR, "Failed to read the stream");
        easyav1_destroy(&e);
        return nullptr;
    }
    e->data = e->lazy.base;
    e->size = e->lazy.size;
    return open_common(e);
}

easyav1_t *easyav1_init_from_memory(const void *data, size_t size, const easyav1_settings *settings)
{
    if (!data || !size) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Data is NULL or size is 0");
        return nullptr;
    }
    easyav1_t *e = new_instance(settings);
    if (!e) return nullptr;
    e->data = static_cast<const uint8_t *>(data);   // borrowed: the caller keeps it alive (src/easyav1.c:1658)
    e->size = size;
    e = open_common(e);
    if (!e) logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Failed to create easyav1 handle from memory stream");
    return e;
}

easyav1_t *easyav1_init_from_file(FILE *f, const easyav1_settings *settings)
{
    if (!f) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "File handle is NULL");
        return nullptr;
    }
    const bool close_it = settings ? settings->close_handle_on_destroy != EASYAV1_FALSE : kDefaults.close_handle_on_destroy != EASYAV1_FALSE;
    easyav1_stream st;
    st.read_func = [](void *buf, size_t n, void *u) -> int {
        const size_t r = fread(buf, 1, n, static_cast<FILE *>(u));
        return r == n ? 1 : (feof(static_cast<FILE *>(u)) ? 0 : -1);
    };
    st.seek_func = [](int64_t off, int origin, void *u) -> int { return fseek(static_cast<FILE *>(u), long(off), origin); };
    st.tell_func = [](void *u) -> int64_t { return ftell(static_cast<FILE *>(u)); };
    st.userdata = f;
    easyav1_t *e = easyav1_init_from_custom_stream(&st, settings);
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Failed to create easyav1 structure from file handle");
        if (close_it) fclose(f);
        return nullptr;
    }
    if (close_it) e->close_on_destroy = f;
    return e;
}

easyav1_t *easyav1_init_from_filename(const char *filename, const easyav1_settings *settings)
{
    if (!filename) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Filename is NULL");
        return nullptr;
    }
    FILE *f = fopen(filename, "rb");
    if (!f) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Failed to open file %s", filename);
        return nullptr;
    }
    easyav1_settings s = settings ? *settings : kDefaults;
    s.close_handle_on_destroy = EASYAV1_FALSE;   // the file is ours either way: closed exactly once below / in destroy
    easyav1_t *e = easyav1_init_from_file(f, &s);
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_ERROR, "Failed to create easyav1 structure from file %s", filename);
        fclose(f);
        return nullptr;
    }
    if (settings) e->settings.close_handle_on_destroy = settings->close_handle_on_destroy;
    e->close_on_destroy = f;
    return e;
}

easyav1_status easyav1_decode_next(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    if (is_error(e->status)) return EASYAV1_STATUS_ERROR;
    if (e->playback.active.load()) {
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Cannot manually decode when using playback functions.");
        return EASYAV1_STATUS_OK;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    bool is_video = false;
    QueuedPacket *q = next_packet(e, &is_video);
    if (e->status == EASYAV1_STATUS_FINISHED) return EASYAV1_STATUS_FINISHED;
    if (!q) return EASYAV1_STATUS_ERROR;
    {
        std::lock_guard<std::mutex> g(e->info);
        e->position = q->ts;
    }
    const easyav1_status status = decode_packet(e, *q, is_video);
    release_packet(e, is_video);
    if (status == EASYAV1_STATUS_OK) run_callbacks(e);
    return status;
}

easyav1_status easyav1_decode_until(easyav1_t *e, easyav1_timestamp timestamp)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    if (e->playback.active.load()) {
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Cannot manually decode when using playback functions.");
        return EASYAV1_STATUS_OK;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    return decode_until(e, timestamp);
}

easyav1_status easyav1_decode_for(easyav1_t *e, easyav1_timestamp time)
{
    if (!e) {   // the reference dereferences the handle before its check (src/easyav1.c:2791); a NULL handle is an error here
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    return easyav1_decode_until(e, easyav1_get_current_timestamp(e) + time);
}

easyav1_status easyav1_play(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    if (is_error(e->status)) return EASYAV1_STATUS_ERROR;
    if (e->status == EASYAV1_STATUS_FINISHED) return EASYAV1_STATUS_FINISHED;
    if (e->playback.active.load()) {
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Playback is already active.");
        return EASYAV1_STATUS_OK;
    }
    e->playback.pause = false;
    e->playback.seek_requested = false;
    e->playback.active = true;
    try {
        e->playback.thread = std::thread(playback_loop, e);
    } catch (...) {
        logf(e, EASYAV1_LOG_LEVEL_ERROR, "Failed to create playback thread.");
        e->playback.active = false;
        return EASYAV1_STATUS_ERROR;
    }
    return EASYAV1_STATUS_OK;
}

void easyav1_stop(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return;
    }
    if (!e->playback.active.load()) {
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Playback is not active.");
        return;
    }
    e->playback.pause = true;
    if (e->playback.thread.joinable()) e->playback.thread.join();
    e->playback.active = false;
    e->playback.pause = false;
    e->playback.seek_requested = false;
}

easyav1_status easyav1_seek_to_timestamp(easyav1_t *e, easyav1_timestamp timestamp)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    if (e->playback.active.load()) {   // the playback thread performs it between two decode steps
        e->playback.seek_to = timestamp;
        e->playback.seek_requested = true;
        return EASYAV1_STATUS_OK;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    return do_seek(e, timestamp, e->settings.use_fast_seeking != EASYAV1_FALSE);
}

easyav1_status easyav1_seek_forward(easyav1_t *e, easyav1_timestamp time)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    return easyav1_seek_to_timestamp(e, easyav1_get_current_timestamp(e) + time);
}

easyav1_status easyav1_seek_backward(easyav1_t *e, easyav1_timestamp time)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    const easyav1_timestamp pos = easyav1_get_current_timestamp(e);
    return easyav1_seek_to_timestamp(e, pos - (time > pos ? pos : time));
}

easyav1_status easyav1_get_status(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    std::lock_guard<std::mutex> g(e->info);
    return is_error(e->status) ? EASYAV1_STATUS_ERROR : easyav1_status(e->status);
}

easyav1_timestamp easyav1_get_current_timestamp(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return 0;
    }
    std::lock_guard<std::mutex> g(e->info);
    return e->position;
}

#define EA1_GETTER(expr, fallback)                                        \
    if (!e || !e->webm) {                                                 \
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");       \
        return fallback;                                                  \
    }                                                                     \
    return expr;

easyav1_bool easyav1_has_video_track(const easyav1_t *e) { EA1_GETTER(e->video.active ? EASYAV1_TRUE : EASYAV1_FALSE, EASYAV1_FALSE) }
easyav1_bool easyav1_has_audio_track(const easyav1_t *e) { EA1_GETTER(e->audio.active ? EASYAV1_TRUE : EASYAV1_FALSE, EASYAV1_FALSE) }
unsigned int easyav1_get_video_width(const easyav1_t *e) { EA1_GETTER(e->video.active ? e->video.width : 0, 0) }
unsigned int easyav1_get_video_height(const easyav1_t *e) { EA1_GETTER(e->video.active ? e->video.height : 0, 0) }
unsigned int easyav1_get_video_fps(const easyav1_t *e) { EA1_GETTER(e->video.active ? e->video.fps : 0, 0) }
unsigned int easyav1_get_audio_channels(const easyav1_t *e) { EA1_GETTER(e->audio.active ? e->audio.channels : 0, 0) }
unsigned int easyav1_get_audio_sample_rate(const easyav1_t *e) { EA1_GETTER(e->audio.active ? e->audio.rate : 0, 0) }
easyav1_timestamp easyav1_get_duration(const easyav1_t *e) { EA1_GETTER(e->duration, 0) }
#undef EA1_GETTER
unsigned int easyav1_get_total_video_tracks(const easyav1_t *e) { return e ? e->video_tracks : 0; }
unsigned int easyav1_get_total_audio_tracks(const easyav1_t *e) { return e ? e->audio_tracks : 0; }

easyav1_bool easyav1_is_finished(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return EASYAV1_FALSE;
    }
    std::lock_guard<std::mutex> g(e->info);
    return e->status == EASYAV1_STATUS_FINISHED ? EASYAV1_TRUE : EASYAV1_FALSE;
}

easyav1_bool easyav1_has_video_frame(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return EASYAV1_FALSE;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    if (!e->video.sink || !e->video.any_decoded) return EASYAV1_FALSE;
    const easyav1_timestamp due = std::min(easyav1_get_current_timestamp(e), e->video.decoded_ts);
    if (e->video.old_sink && ea1::video_sink_has_frame(e->video.old_sink, due)) return EASYAV1_TRUE;
    return ea1::video_sink_has_frame(e->video.sink, due) ? EASYAV1_TRUE : EASYAV1_FALSE;
}

const easyav1_video_frame *easyav1_get_video_frame(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return nullptr;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    if (!e->video.sink || !e->video.have_seq || !e->video.any_decoded) return nullptr;
    ea1::SinkFrame f;
    const easyav1_timestamp due = std::min(easyav1_get_current_timestamp(e), e->video.decoded_ts);
    // a retired sink (its sequence header needed another context) hands out its pictures first; it goes when none is left
    bool taken = e->video.old_sink && ea1::video_sink_take(e->video.old_sink, due, &f);
    if (!taken && e->video.old_sink && !ea1::video_sink_has_frame(e->video.old_sink, UINT64_MAX)) {
        ea1::video_sink_destroy(e->video.old_sink);
        e->video.old_sink = nullptr;
    }
    if (!taken && !ea1::video_sink_take(e->video.sink, due, &f)) return nullptr;
    easyav1_video_frame &out = e->video.frame;
    for (int pl = 0; pl < 3; pl++) {
        out.data[pl] = f.data[pl];
        out.stride[pl] = f.stride[pl];
    }
    // the picture's own sequence header (numbered when its packet was submitted): layout, colour description, 4:0:0
    const ea1_av1_sequence_header &sq = f.tag < e->video.seqs.size() ? e->video.seqs[f.tag] : e->video.seq;
    fill_frame_properties(e, sq);
    if (sq.monochrome) {   // 4:0:0: no chroma planes (dav1d: data[1] = data[2] = NULL, stride[1] = 0; src/easyav1.c:3557-3563 hands them on)
        out.data[1] = out.data[2] = nullptr;
        out.stride[1] = out.stride[2] = 0;
    }
    out.properties.width = f.width;
    out.properties.height = f.height;
    out.timestamp = f.timestamp;
    return &out;
}

uint64_t easyav1_get_total_video_frames_processed(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return 0;
    }
    std::lock_guard<std::recursive_mutex> st(const_cast<easyav1_t *>(e)->state);
    return e->video.processed;
}

easyav1_bool easyav1_is_audio_buffer_filled(const easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return EASYAV1_FALSE;
    }
    std::lock_guard<std::recursive_mutex> st(const_cast<easyav1_t *>(e)->state);
    return e->audio.ring && ea1_audio_ring_is_filled(e->audio.ring) ? EASYAV1_TRUE : EASYAV1_FALSE;
}

const easyav1_audio_frame *easyav1_get_audio_frame(easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return nullptr;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    if (!e->audio.ring || !e->audio.pending) return nullptr;
    e->audio.pending = false;
    const float *buf = nullptr;
    size_t bytes = 0;
    const unsigned n = ea1_audio_ring_get(e->audio.ring, &buf, &bytes);
    if (!n) return nullptr;
    easyav1_audio_frame &f = e->audio.frame;
    f.samples = n;
    f.bytes = bytes;
    if (e->settings.interlace_audio) {
        f.pcm.interlaced = buf;
    } else {
        for (unsigned c = 0; c < e->audio.channels; c++) e->audio.planes[c] = buf + size_t(c) * EA1_AUDIO_BUFFER_SIZE;
        f.pcm.deinterlaced = e->audio.planes.data();
    }
    return &f;
}

easyav1_settings easyav1_get_current_settings(const easyav1_t *e)
{
    if (!e) {
        logf(nullptr, EASYAV1_LOG_LEVEL_WARNING, "Handle is NULL");
        return kDefaults;
    }
    return e->settings;
}

// the n-th track of a kind, picked again while the file plays (change_track, src/easyav1.c:3795-3850): a video track of another codec
// leaves the video off; an audio track of another codec is passed over and the NEXT audio track is tried (the reference's loop does not
// count the refused one); a track that is not there switches the kind off with a warning -- none of these is an error
static easyav1_status change_track(easyav1_t *e, bool video, unsigned track_id)
{
    unsigned current = 0;
    for (uint32_t t = 0; t < e->num_tracks; t++) {
        ea1_webm_track ti;
        if (ea1_webm_track_info(e->webm, t, &ti) != EA1_OK) {
            set_error(e, EASYAV1_STATUS_DECODER_ERROR, "Failed to get track type.");
            return EASYAV1_STATUS_ERROR;
        }
        if (ti.type != (video ? EA1_WEBM_TRACK_VIDEO : EA1_WEBM_TRACK_AUDIO)) continue;
        if (current != track_id) {
            current++;
            continue;
        }
        logf(e, EASYAV1_LOG_LEVEL_INFO, "Found requested %s track %u at webm track %u.", video ? "video" : "audio", current, t);
        if (video) {
            if (ti.codec != EA1_WEBM_CODEC_AV1) {
                logf(e, EASYAV1_LOG_LEVEL_WARNING, "Unsupported video codec found. Only AV1 codec is supported. Not displaying video.");
                return EASYAV1_STATUS_OK;
            }
            return open_video(e, t, ti) ? EASYAV1_STATUS_OK : EASYAV1_STATUS_ERROR;
        }
        if (ti.codec != EA1_WEBM_CODEC_VORBIS) {
            logf(e, EASYAV1_LOG_LEVEL_WARNING, "Unsupported audio codec found. Only vorbis codec is supported. Not playing audio.");
            continue;
        }
        return open_audio(e, t, ti) ? EASYAV1_STATUS_OK : EASYAV1_STATUS_ERROR;
    }
    logf(e, EASYAV1_LOG_LEVEL_WARNING, "Track was not found, disabling %s.", video ? "video" : "audio");
    return EASYAV1_STATUS_OK;
}

// src/easyav1.c:3852-3949, branch for branch.  What follows from that text and is kept: a kind that is switched off and on again WITHOUT naming
// another track stays off (change_track is only called when the track index changes); interlace_audio and audio_offset_time are looked at only
// when the audio track settings did not change; every change ends in a SLOW seek to the current position, which restarts the audio decoder
// (vorbis_synthesis_restart) and decodes the video forward from the last key frame.
easyav1_status easyav1_update_settings(easyav1_t *e, const easyav1_settings *settings)
{
    if (!e || !settings) {
        logf(e, EASYAV1_LOG_LEVEL_WARNING, e ? "Settings are NULL" : "Handle is NULL");
        return EASYAV1_STATUS_ERROR;
    }
    std::lock_guard<std::recursive_mutex> st(e->state);
    const easyav1_settings old = e->settings;
    e->settings = *settings;
    bool must_seek = false;
    easyav1_status status = EASYAV1_STATUS_OK;
    if (settings->enable_audio != old.enable_audio || settings->audio_track != old.audio_track) {
        must_seek = true;
        if (old.enable_audio == EASYAV1_TRUE) close_audio(e);
        if (settings->enable_audio == EASYAV1_TRUE && settings->audio_track != old.audio_track) status = change_track(e, false, settings->audio_track);
        if (status == EASYAV1_STATUS_OK && e->audio.active && !e->ahead && open_reader(e, &e->ahead) != EA1_OK) status = EASYAV1_STATUS_ERROR;
    } else if (settings->enable_audio == EASYAV1_TRUE && settings->interlace_audio != old.interlace_audio) {
        // (prepare_audio_buffer of the reference allocates whether or not a track is active; here the ring belongs to an open track)
        if (e->audio.active && !make_audio_buffer(e)) status = EASYAV1_STATUS_ERROR;
        must_seek = true;
    } else if (settings->enable_audio == EASYAV1_TRUE && settings->audio_offset_time != old.audio_offset_time) {
        e->audio.offset = settings->audio_offset_time + int64_t(to_ms(e, e->audio.codec_delay));   // src/easyav1.c:3909-3910
        must_seek = true;
    }
    if (settings->enable_video != old.enable_video || settings->video_track != old.video_track) {
        must_seek = true;
        if (old.enable_video == EASYAV1_TRUE) close_video(e);
        if (settings->enable_video == EASYAV1_TRUE && settings->video_track != old.video_track && status == EASYAV1_STATUS_OK)
            status = change_track(e, true, settings->video_track);
    }
    if (status != EASYAV1_STATUS_OK) return status;
    if (!must_seek) return EASYAV1_STATUS_OK;
    logf(e, EASYAV1_LOG_LEVEL_INFO, "Settings changed, seeking to timestamp %llu.", (unsigned long long)e->position);
    // (slow seeking whatever the setting says: the position must stay where it is, src/easyav1.c:3937-3945; the reference bumps its position by
    // one to get past its own "already there" test -- do_seek here has no such test)
    const easyav1_bool fast = e->settings.use_fast_seeking;
    e->settings.use_fast_seeking = EASYAV1_FALSE;
    const easyav1_status st_seek = easyav1_seek_to_timestamp(e, easyav1_get_current_timestamp(e));
    e->settings.use_fast_seeking = fast;
    return st_seek;
}

void easyav1_destroy(easyav1_t **handle)
{
    if (!handle || !*handle) {
        logf(nullptr, EASYAV1_LOG_LEVEL_INFO, "Handle is NULL");
        return;
    }
    easyav1_t *e = *handle;
    if (e->playback.active.load()) easyav1_stop(e);
    close_video(e);
    close_audio(e);
    if (e->ahead) ea1_webm_destroy(e->ahead);
    if (e->webm) ea1_webm_destroy(e->webm);
    if (e->close_on_destroy) fclose(e->close_on_destroy);
    if (e->lazy.base) munmap(e->lazy.base, e->lazy.size);
    if (getenv("EASYAV1_AMD_STATS") && e->lazy.base)
        fprintf(stderr, "[easyav1 stats] input: %llu of %zu bytes read through the stream callbacks in %llu reads, %llu pages of 64 KiB given back behind the reader\n",
                (unsigned long long)e->lazy.bytes_read, e->lazy.size, (unsigned long long)e->lazy.reads, (unsigned long long)e->lazy.pages_dropped);
    delete e;
    *handle = nullptr;
}

}  // extern "C"
