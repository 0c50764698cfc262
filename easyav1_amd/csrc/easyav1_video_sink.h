// Video side of libeasyav1.so: parsed frames that arrive inside the container (private
// OBU_METADATA, include/easyav1_amd_ir_packet.h) -> GPU reconstruction (easyav1_amd_video.h) ->
// pictures in pinned host memory, handed out in presentation order.  Stands where the reference
// has dav1d_send_data / dav1d_get_picture, its decoder thread and its decoded-frame ring
// (/root/reference/src/easyav1.c:2271-2333, :2360-2428, :2133-2178).  Internal to the shim, not part of the C ABI.
//
// A pipeline: video_sink_submit only queues a packet (the shim queues the packets it has prefetched, as the
// reference's decoder thread runs up to 10 packets ahead of the position, src/easyav1.c:1813-1824 -- here further:
// video_sink_lookahead packets, enough for several GOPs);
//   helper threads   parse the payload into pinned memory and run the host half of the hand-over (ea1_job_prepare)
//   one issuer       commits jobs (copies + derivation), submits the reconstruction -- one frame of every GOP chain
//                    per launch: closed GOPs are independent, a key-frame packet starts the next chain in turn --
//                    on one HIP stream and the copies of the finished pictures to pinned host memory on a second one
//   the caller       video_sink_wait(ticket) = "this packet is decoded" (what decode_packet of the reference waits
//                    for, src/easyav1.c:2568-2585), video_sink_take = dav1d_get_picture + the frame ring
// All calls below come from one thread at a time (the shim holds its state lock); the sink's own threads never
// touch what those calls hand out.
#pragma once
#include <cstddef>
#include <cstdint>

namespace ea1 {

struct VideoSink;

struct SinkFrame {
    const uint8_t *data[3];   // host pointers, valid until the next video_sink_take / flush / destroy
    size_t stride[3];
    unsigned width, height;   // visible size
    uint64_t timestamp;
    uint32_t tag;             // video_sink_set_tag's value when the picture's packet was submitted
};

bool is_parsed_frame_metadata(const uint8_t *payload, size_t size);
// device: EASYAV1_AMD_DEVICE (default 0); helper threads: EASYAV1_AMD_THREADS (default min(8, hardware threads); 16 with the AV1
// tables); GOP chains in flight: EASYAV1_AMD_GOPS (default 16 up to 1080p, 8 above -- 16 with the AV1 tables; 1 = one frame per
// launch); packets read ahead: EASYAV1_AMD_LOOKAHEAD
int video_sink_create(VideoSink **out, unsigned width, unsigned height);
// the decoder context, the pinned picture ring and the parse storage at once (what opening a decoder costs; without this call the
// first frames make them as they go): frame_w x frame_h = size of the first frame, payload_bytes = size of its packet
// layout: EA1_LAYOUT_* of the stream (ea1_frame_desc.layout of its frames)
int video_sink_open(VideoSink *s, unsigned frame_w, unsigned frame_h, size_t payload_bytes, int layout);
// layout and largest frame size of the track's sequence header, named BEFORE packets are submitted: the helpers work on several GOPs at once, and
// without this the first frame one of them happens to finish decides what the context is made for
void video_sink_declare(VideoSink *s, unsigned max_w, unsigned max_h, int layout);
// the context this sink has, or will make from what was declared, holds frames of this layout up to max_w x max_h (a new sequence header inside a track:
// one that does not fit is decoded by a sink of its own, made when the pictures in front of it are out)
bool video_sink_fits(VideoSink *s, unsigned max_w, unsigned max_h, int layout);
// how many video packets the shim should keep queued here ahead of the position
int video_sink_lookahead(const VideoSink *s);
// GOP chains this sink (all its device contexts together) decodes side by side
int video_sink_chains(const VideoSink *s);
void video_sink_destroy(VideoSink *s);
// queues one parsed frame; `payload` must stay valid until the frame has been waited for, flushed or the sink destroyed
// (the shim's packets live as long as the container bytes).  discard: the frame is reconstructed (later frames
// predict from it) but no picture is delivered (frames between a key frame and a seek target).
// keyframe: the container flags the packet as a key frame -- it starts a GOP, which the next chain in turn decodes.
int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp, bool keyframe, bool discard, uint64_t *ticket);
// a number of the caller's that travels with every packet submitted from now on and comes back with its picture (SinkFrame.tag): the shim numbers
// the sequence headers of a track with it -- the pipeline works far ahead of the picture that is shown, and a picture's properties are those of
// ITS sequence (the reference reads them off the picture: src/easyav1.c:3287-3504)
void video_sink_set_tag(VideoSink *s, uint32_t tag);
// A plain AV1 temporal unit (the whole container packet) instead of a parsed-frame payload: the sink's block-level parser
// (include/easyav1_amd_obu.h, tables from the file EASYAV1_AMD_AV1_TABLES names) turns it into the same hand-over.  The symbol
// parse of a stream is serial -- one helper at a time, packets in order -- everything behind it is the pipeline above.
// video_sink_parses_av1: the tables were given.  EA1_ENOTSUP from video_sink_wait: the stream uses what the parser does not cover.
bool video_sink_parses_av1(const VideoSink *s);
int video_sink_submit_av1(VideoSink *s, const uint8_t *unit, size_t size, uint64_t timestamp, bool keyframe, bool discard, uint64_t *ticket);
// frames parsed so far and the host seconds their symbol parse took (one thread): the Amdahl number of the path
void video_sink_av1_stats(const VideoSink *s, uint64_t *frames, double *seconds);
// blocks until that frame is reconstructed and its picture has landed in host memory; an error of this or an
// earlier frame is returned here (text in ea1_last_error())
int video_sink_wait(VideoSink *s, uint64_t ticket);
// video_sink_wait(ticket) would return at once (done, failed, or gone): the caller may spend the time until then feeding more packets
bool video_sink_ready(VideoSink *s, uint64_t ticket);
// the oldest finished picture has a timestamp <= position
bool video_sink_has_frame(VideoSink *s, uint64_t position);
// hands that picture out; false when none is due
bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out);
void video_sink_flush(VideoSink *s);

}  // namespace ea1
