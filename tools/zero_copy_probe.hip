// Host link, kernel-driven: ONE launch that pulls many separate pinned host buffers into device memory
// (a list of {dst, src, bytes}), against the same bytes moved by back-to-back hipMemcpyAsync calls.
//   hipcc --offload-arch=gfx950 -O3 tools/zero_copy_probe.hip -o tools/bin/zero_copy_probe && tools/bin/zero_copy_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct Seg {
    uint4 *dst;
    const uint4 *src;
    uint32_t n16;   // 16-byte units
    uint32_t first_block;
};

// workgroup w copies the w-th chunk of kChunk 16-byte units; segs are found by a short scan of first_block
constexpr uint32_t kChunk = 4096;   // 64 KB per workgroup
__global__ void __launch_bounds__(256) pull_kernel(const Seg *segs, int nsegs)
{
    int s = 0;
    while (s + 1 < nsegs && segs[s + 1].first_block <= blockIdx.x) s++;
    const Seg sg = segs[s];
    const uint32_t begin = (blockIdx.x - sg.first_block) * kChunk;
    const uint32_t end = begin + kChunk < sg.n16 ? begin + kChunk : sg.n16;
    for (uint32_t i = begin + threadIdx.x; i < end; i += 1024) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + 256 * k < end) v[k] = sg.src[i + 256 * k];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + 256 * k < end) sg.dst[i + 256 * k] = v[k];
    }
}

int main()
{
    const size_t sizes[] = {64 << 10, 512 << 10, 1600 << 10, 6400 << 10};
    hipStream_t st;
    hipStreamCreate(&st);
    for (size_t sz : sizes) {
        const int n = int((512u << 20) / sz) > 256 ? 256 : int((512u << 20) / sz);
        std::vector<void *> h(n), d(n);
        for (int k = 0; k < n; k++) {
            hipHostMalloc(&h[k], sz, hipHostMallocDefault);
            hipMalloc(&d[k], sz);
            memset(h[k], k + 1, sz);
        }
        std::vector<Seg> segs(n);
        uint32_t blocks = 0;
        for (int k = 0; k < n; k++) {
            segs[k] = Seg{static_cast<uint4 *>(d[k]), static_cast<const uint4 *>(h[k]), uint32_t(sz / 16), blocks};
            blocks += uint32_t((sz / 16 + kChunk - 1) / kChunk);
        }
        Seg *dsegs;
        hipMalloc(&dsegs, n * sizeof(Seg));
        hipMemcpy(dsegs, segs.data(), n * sizeof(Seg), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            double best = 0;
            for (int rep = 0; rep < 4; rep++) {
                hipStreamSynchronize(st);
                const auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) {
                    for (int k = 0; k < n; k++) hipMemcpyAsync(d[k], h[k], sz, hipMemcpyHostToDevice, st);
                } else {
                    hipLaunchKernelGGL(pull_kernel, dim3(blocks), dim3(256), 0, st, dsegs, n);
                }
                hipStreamSynchronize(st);
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                best = std::max(best, double(n) * sz / s / 1e9);
            }
            printf("%5zu KB x %3d buffers: %-28s %6.1f GB/s\n", sz >> 10, n, mode == 0 ? "hipMemcpyAsync back to back" : "one pull kernel", best);
        }
        uint8_t probe = 0;
        hipMemcpy(&probe, static_cast<uint8_t *>(d[n - 1]) + sz - 1, 1, hipMemcpyDeviceToHost);
        if (probe != uint8_t(n)) printf("  MISMATCH %u\n", probe);
        for (int k = 0; k < n; k++) {
            hipHostFree(h[k]);
            hipFree(d[k]);
        }
        hipFree(dsegs);
    }
    return 0;
}
