#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include "sampler.h"
#include "easyav1_amd_video.h"
#include "easyav1_amd_ir_packet.h"
extern "C" int ea1_fuzz_job_prepare_layout(uint32_t width, uint32_t height, int layout, const ea1_ir_frame *frame);
int main(int argc, char **argv)
{
    FILE *fp = fopen(argv[1], "rb");
    int reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<uint8_t> all;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) all.insert(all.end(), buf, buf + n);
    if (getenv("PROF")) prof_start();
    size_t pos = 0; int k = 0;
    double total = 0;
    while (pos + 4 <= all.size()) {
        uint32_t len; memcpy(&len, &all[pos], 4); pos += 4;
        ea1_ir_frame f; ea1_ir_storage *st = nullptr;
        if (ea1_ir_packet_parse(&all[pos], len, &st, &f) != EA1_OK) { printf("parse failed\n"); return 1; }
        pos += len;
        double best = 1e9; int rc = 0;
        for (int r = 0; r < reps; r++) {
            auto t = std::chrono::steady_clock::now();
            rc = ea1_fuzz_job_prepare_layout(f.desc.width, f.desc.height, f.desc.layout, &f);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
            if (ms < best) best = ms;
        }
        printf("frame %d: rc %d, %.3f ms (blocks %u, tx %u)\n", k++, rc, best, f.desc.num_blocks, f.desc.num_tx);
        total += best;
    }
    if (getenv("PROF")) prof_dump(getenv("PROF"));
    printf("total %.3f ms\n", total);
}
