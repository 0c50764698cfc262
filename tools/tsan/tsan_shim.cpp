// ThreadSanitizer scenarios for the host shim behind include/easyav1.h (see tools/tsan/run.sh and device_stub.cpp): the player pattern of
// the reference's tools/easyav1_player.c -- easyav1_play() and polling getters from the caller's thread (:910-912, :1302), seeks while playing
// (:807-811, :902), settings changed while playing (:367), destroy while playing (:1372) -- on a real plain-AV1 WebM file.  Exit code 0 and
// no ThreadSanitizer report = pass.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "easyav1.h"

static std::vector<uint8_t> slurp_file(const char *path)
{
    std::vector<uint8_t> v;
    FILE *fp = fopen(path, "rb");
    if (!fp) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(fp);
    return v;
}

static easyav1_t *open_file(const std::vector<uint8_t> &data)
{
    easyav1_settings st = easyav1_default_settings();
    st.enable_audio = EASYAV1_FALSE;
    st.log_level = EASYAV1_LOG_LEVEL_ERROR;
    return easyav1_init_from_memory(data.data(), data.size(), &st);
}
static void nap(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::vector<uint8_t> data = slurp_file(argv[1]);
    if (data.empty()) return 3;
    unsigned long frames = 0;
    // 1. decode loop of the benchmark tool (caller thread + helpers + stage thread)
    {
        easyav1_t *e = open_file(data);
        if (!e) return 4;
        while (easyav1_decode_next(e) == EASYAV1_STATUS_OK)
            if (easyav1_has_video_frame(e) && easyav1_get_video_frame(e)) frames++;
        if (easyav1_get_status(e) == EASYAV1_STATUS_ERROR) return 5;
        easyav1_destroy(&e);
    }
    // 2. play, then poll the getters while the playback thread decodes
    {
        easyav1_t *e = open_file(data);
        if (!e || easyav1_play(e) != EASYAV1_STATUS_OK) return 6;
        for (int i = 0; i < 400 && !easyav1_is_finished(e); i++) {
            if (easyav1_has_video_frame(e)) {
                const easyav1_video_frame *f = easyav1_get_video_frame(e);
                if (f) frames += static_cast<const uint8_t *>(f->data[0])[0] != 0;
            }
            (void)easyav1_get_current_timestamp(e);
            (void)easyav1_get_total_video_frames_processed(e);
            nap(1);
        }
        easyav1_stop(e);
        easyav1_destroy(&e);
    }
    // 3. a seek storm while playing
    {
        easyav1_t *e = open_file(data);
        if (!e || easyav1_play(e) != EASYAV1_STATUS_OK) return 7;
        const easyav1_timestamp dur = easyav1_get_duration(e);
        for (int i = 0; i < 40; i++) {
            easyav1_seek_to_timestamp(e, dur ? (easyav1_timestamp(i) * 7919u) % dur : 0);
            if (easyav1_has_video_frame(e)) (void)easyav1_get_video_frame(e);
            if (i % 5 == 0) easyav1_seek_forward(e, 40);
            if (i % 7 == 0) easyav1_seek_backward(e, 80);
            nap(2);
        }
        easyav1_stop(e);
        easyav1_destroy(&e);
    }
    // 4. settings updated while playing
    {
        easyav1_t *e = open_file(data);
        if (!e || easyav1_play(e) != EASYAV1_STATUS_OK) return 8;
        for (int i = 0; i < 12; i++) {
            easyav1_settings st = easyav1_get_current_settings(e);
            st.skip_unprocessed_frames = i & 1 ? EASYAV1_TRUE : EASYAV1_FALSE;
            st.use_fast_seeking = i & 2 ? EASYAV1_TRUE : EASYAV1_FALSE;
            easyav1_update_settings(e, &st);
            if (easyav1_has_video_frame(e)) (void)easyav1_get_video_frame(e);
            nap(3);
        }
        easyav1_stop(e);
        easyav1_destroy(&e);
    }
    // 5. destroy while the playback thread runs
    for (int k = 0; k < 3; k++) {
        easyav1_t *e = open_file(data);
        if (!e || easyav1_play(e) != EASYAV1_STATUS_OK) return 9;
        nap(5 + 10 * k);
        easyav1_destroy(&e);
        if (e) return 10;
    }
    // 6. GOPs dealt over two device contexts (EASYAV1_AMD_DEVICES=0,0: the dealer of easyav1_video_sink.cpp): as many pictures as one sink
    //    gives, in stream order
    {
        unsigned long one = 0, two = 0;
        for (int pass = 0; pass < 2; pass++) {
            if (pass) setenv("EASYAV1_AMD_DEVICES", "0,0", 1);
            easyav1_t *e = open_file(data);
            if (!e) return 11;
            easyav1_timestamp last = 0;
            bool first = true;
            while (easyav1_decode_next(e) == EASYAV1_STATUS_OK)
                if (easyav1_has_video_frame(e)) {
                    const easyav1_video_frame *f = easyav1_get_video_frame(e);
                    if (!f) continue;
                    if (!first && f->timestamp <= last) return 12;   // out of order
                    last = f->timestamp, first = false;
                    (pass ? two : one)++;
                }
            if (easyav1_get_status(e) == EASYAV1_STATUS_ERROR) return 13;
            easyav1_destroy(&e);
        }
        unsetenv("EASYAV1_AMD_DEVICES");
        if (one != two || !one) return 14;
        frames += two;
    }
    printf("tsan_shim: %lu pictures taken, all scenarios done\n", frames);
    return 0;
}
