// Video side of libeasyav1.so: parsed frames that arrive inside the container (private
// OBU_METADATA, include/easyav1_amd_ir_packet.h) -> GPU reconstruction (easyav1_amd_video.h) ->
// pictures in pinned host memory, handed out in presentation order.  Stands where the reference
// has dav1d_send_data / dav1d_get_picture plus its decoded-frame ring
// (/root/reference/src/easyav1.c:2360-2428, :2133-2178).  Internal to the shim, not part of the C ABI.
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
};

bool is_parsed_frame_metadata(const uint8_t *payload, size_t size);
int video_sink_create(VideoSink **out, unsigned width, unsigned height);
void video_sink_destroy(VideoSink *s);
// reconstructs asynchronously; the picture becomes available to video_sink_take in submission order
int video_sink_submit(VideoSink *s, const uint8_t *payload, size_t size, uint64_t timestamp);
// the oldest waiting picture has a timestamp <= position
bool video_sink_has_frame(const VideoSink *s, uint64_t position);
// hands that picture out (waits for its copy to land); false when none is due
bool video_sink_take(VideoSink *s, uint64_t position, SinkFrame *out);
void video_sink_flush(VideoSink *s);

}  // namespace ea1
