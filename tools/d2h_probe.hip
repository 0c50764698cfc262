// d2h_probe -- why do device-to-host picture copies that run BESIDE kernels reach ~30 GB/s on this box when the same copies on
// an idle device reach ~47 (profiles/r02_b_host_link_probe.txt; bench.py with_d2h)?  Measures 12.5 MB "pictures" (one 4K frame in
// device layout) going to pinned host memory
//   by the DMA engines (hipMemcpyAsync): one stream, a high-priority stream, the picture split over 2 / 4 streams
//   by a kernel that stores to the pinned buffer (device -> host writes over the link), grids of 8 .. 256 workgroups
// each alone, beside a VALU-bound kernel (no memory traffic) and beside an HBM-streaming kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/d2h_probe.hip -o tools/bin/d2h_probe ; HSA_ENABLE_SDMA=0 repeats it with blit kernels
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__global__ void valu_busy(float *out, int iters)
{
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = 0; i < iters; i++) {
        a = a * b + 0.5f;
        b = b * 0.9999f + a * 1e-9f;
    }
    if (a == 12345.678f) out[0] = a + b;   // never true: keeps the loop
}

__global__ void hbm_busy(float4 *buf, size_t n, int rounds)
{
    for (int r = 0; r < rounds; r++)
        for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
            float4 v = buf[i];
            v.x += 1.0f;
            buf[i] = v;
        }
}

// stores 16 bytes per lane per step into host memory: whole 64-byte (wave: 1 KB) write bursts over the link
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void push_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
    u32x4 *d = reinterpret_cast<u32x4 *>(dst);
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x)
        __builtin_nontemporal_store(s[i], &d[i]);
}

using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

int main()
{
    const size_t pic = 12503040, npics = 32, rounds = 6;
    uint8_t *dev = nullptr, *host = nullptr;
    float *sink = nullptr;
    float4 *stream_buf = nullptr;
    const size_t stream_n = (256u << 20) / 16;
    CK(hipMalloc(&dev, pic * npics));
    CK(hipHostMalloc(&host, pic * npics * 2, hipHostMallocDefault));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&stream_buf, stream_n * 16));
    CK(hipMemset(dev, 1, pic * npics));
    CK(hipMemset(stream_buf, 0, stream_n * 16));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t busy_st, cp[4], cp_hi;
    CK(hipStreamCreateWithFlags(&busy_st, hipStreamNonBlocking));
    for (auto &s : cp) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&cp_hi, hipStreamNonBlocking, hi));
    printf("stream priorities: lowest %d, highest %d; SDMA %s\n", lo, hi, getenv("HSA_ENABLE_SDMA") ? getenv("HSA_ENABLE_SDMA") : "default (on)");

    enum { ALONE, VALU, HBM };
    const char *beside_name[3] = {"alone", "beside a VALU-bound kernel", "beside an HBM-streaming kernel"};
    auto launch_busy = [&](int kind) {
        if (kind == VALU) hipLaunchKernelGGL(valu_busy, dim3(4096), dim3(256), 0, busy_st, sink, 3000000);
        if (kind == HBM) hipLaunchKernelGGL(hbm_busy, dim3(2048), dim3(256), 0, busy_st, stream_buf, stream_n, 1500);
    };
    struct Variant {
        const char *name;
        int kind;   // 0 DMA one stream, 1 DMA high priority, 2 DMA split in 2, 3 DMA split in 4, 10 + log2(grid): push kernel
    };
    const std::vector<Variant> variants = {{"DMA, one stream", 0}, {"DMA, high-priority stream", 1}, {"DMA, picture split over 2 streams", 2}, {"DMA, picture split over 4 streams", 3},
                                           {"push kernel, 8 workgroups", 13}, {"push kernel, 32 workgroups", 15}, {"push kernel, 64 workgroups", 16}, {"push kernel, 256 workgroups", 18},
                                           {"push kernel, 64 workgroups, high-priority stream", 116}};
    for (int beside = ALONE; beside <= HBM; beside++) {
        for (const Variant &v : variants) {
            CK(hipDeviceSynchronize());
            // the busy kernel runs long enough to cover the copies (it is stopped being waited for: its time is not the point)
            launch_busy(beside);
            const Clock::time_point t0 = Clock::now();
            for (size_t r = 0; r < rounds; r++)
                for (size_t k = 0; k < npics; k++) {
                    const uint8_t *s = dev + k * pic;
                    uint8_t *d = host + ((r & 1) * npics + k) * pic;
                    if (v.kind == 0) CK(hipMemcpyAsync(d, s, pic, hipMemcpyDeviceToHost, cp[0]));
                    else if (v.kind == 1) CK(hipMemcpyAsync(d, s, pic, hipMemcpyDeviceToHost, cp_hi));
                    else if (v.kind == 2 || v.kind == 3) {
                        const int parts = v.kind == 2 ? 2 : 4;
                        const size_t chunk = (pic / parts + 4095) & ~size_t(4095);
                        for (int p = 0; p < parts; p++) {
                            const size_t o = p * chunk, n = o + chunk > pic ? pic - o : chunk;
                            CK(hipMemcpyAsync(d + o, s + o, n, hipMemcpyDeviceToHost, cp[p]));
                        }
                    } else {
                        const int grid = 1 << ((v.kind % 100) - 10);
                        hipLaunchKernelGGL(push_kernel, dim3(grid), dim3(256), 0, v.kind >= 100 ? cp_hi : cp[0], reinterpret_cast<const uint4 *>(s), reinterpret_cast<uint4 *>(d), pic / 16);
                    }
                }
            for (auto &s : cp) CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(cp_hi));
            const double dt = secs(t0);
            const bool still_busy = beside != ALONE && hipStreamQuery(busy_st) == hipErrorNotReady;
            (void)hipGetLastError();
            printf("%-52s %-32s %6.2f GB/s%s\n", v.name, beside_name[beside], rounds * npics * pic / dt / 1e9,
                   beside != ALONE && !still_busy ? "   (the busy kernel ended before the copies: figure partly un-contended)" : "");
            fflush(stdout);
            CK(hipDeviceSynchronize());
        }
    }
    return 0;
}
