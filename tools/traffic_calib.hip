// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// reconstruction kernels use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports 1/2 of the bytes of a
// wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Every kernel touches each byte of a 1 GiB buffer (4x the
// Infinity Cache) exactly once, so bytes per launch = 2^30; run once per counter:
//   hipcc --offload-arch=gfx950 -O3 tools/traffic_calib.hip -o /tmp/traffic_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- /tmp/traffic_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- /tmp/traffic_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t kBytes = size_t(1) << 30;

template <class T>
__global__ void __launch_bounds__(256) rd_stream(const T *__restrict__ src, uint32_t *out, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;   // keeps the loads alive, (almost) never stores
}

// a 1920-byte-pitch picture read as 64-byte row segments: one workgroup per 64 x 16 tile, 16 lanes
// per row, one dword per lane (the staging pattern of the in-loop filter kernels, without halos)
__global__ void __launch_bounds__(256) rd_tile64(const uint8_t *__restrict__ src, uint32_t *out, int rows)
{
    const int tile_x = blockIdx.x % 30, tile_y = blockIdx.x / 30;
    const int r = threadIdx.x / 16, c = threadIdx.x % 16;
    const size_t y = size_t(tile_y) * 16 + r;
    uint32_t acc = 0;
    if (y < size_t(rows)) acc = *reinterpret_cast<const uint32_t *>(src + y * 1920 + tile_x * 64 + c * 4);
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <class T>
__global__ void __launch_bounds__(256) wr_stream(T *__restrict__ dst, size_t n, uint32_t seed)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
        T v;
        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = seed + uint32_t(i);
        dst[i] = v;
    }
}

// the partial-line stores of the 8x8 inter tiles: 8 rows x 8 bytes per tile, two lanes per row
__global__ void __launch_bounds__(64) wr_tile8(uint8_t *__restrict__ dst, int rows, uint32_t seed)
{
    const int tile_x = blockIdx.x % 240, tile_y = blockIdx.x / 240;
    const int r = threadIdx.x / 2, c = threadIdx.x % 2;
    const size_t y = size_t(tile_y) * 8 + r;
    if (threadIdx.x < 16 && y < size_t(rows)) *reinterpret_cast<uint32_t *>(dst + y * 1920 + tile_x * 8 + c * 4) = seed + threadIdx.x;
}

int main()
{
    uint8_t *buf = nullptr;
    uint32_t *out = nullptr;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 1 << 22) != hipSuccess) return 1;
    hipMemset(buf, 1, kBytes);
    const int rows = int(kBytes / 1920);   // 559240 rows of 1920 bytes
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(rd_stream<uint32_t>, dim3(16384), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(buf), out, kBytes / 4);
        hipLaunchKernelGGL(rd_stream<uint2>, dim3(16384), dim3(256), 0, 0, reinterpret_cast<const uint2 *>(buf), out, kBytes / 8);
        hipLaunchKernelGGL(rd_stream<uint4>, dim3(16384), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), out, kBytes / 16);
        hipLaunchKernelGGL(rd_tile64, dim3(30 * ((rows + 15) / 16)), dim3(256), 0, 0, buf, out, rows);
        hipLaunchKernelGGL(wr_stream<uint32_t>, dim3(16384), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(buf), kBytes / 4, 7u + rep);
        hipLaunchKernelGGL(wr_stream<uint4>, dim3(16384), dim3(256), 0, 0, reinterpret_cast<uint4 *>(buf), kBytes / 16, 9u + rep);
        hipLaunchKernelGGL(wr_tile8, dim3(240 * ((rows + 7) / 8)), dim3(64), 0, 0, buf, rows, 11u + rep);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("bytes per launch: streams %zu, rd_tile64 %zu, wr_tile8 %zu\n", kBytes, size_t(rows) * 1920, size_t(rows) * 1920);
    hipFree(buf);
    hipFree(out);
    return 0;
}
