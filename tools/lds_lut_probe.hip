// What a per-lane byte lookup in an LDS table costs on gfx950, alone and beside half-rate vector work: the question behind
// "CDEF's constrain() as a table".  8 independent chains per lane, 8 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_lut_probe.hip -o /tmp/lds_lut_probe && /tmp/lds_lut_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define ITERS 1024

// SPREAD: index mask of the walk (511: all 128 dwords of a 512-byte table; 31: 8 dwords -- differences near zero);
// VOPS: half-rate vector instructions per lookup beside it (v_pk_max_i16 on an independent register)
template <int SPREAD, int VOPS, bool LOOKUP>
__global__ void __launch_bounds__(256) k_probe(uint32_t *out, const uint8_t *table, uint32_t seed)
{
    __shared__ uint8_t lut[4][512];
    for (int i = threadIdx.x; i < 2048; i += 256) (&lut[0][0])[i] = table[i];
    __syncthreads();
    uint32_t a[8], v[8];
    for (int k = 0; k < 8; k++) {
        a[k] = threadIdx.x * (2 * k + 3) + seed;
        v[k] = a[k] * 77 + 5;
    }
    const uint32_t base = (threadIdx.x >> 4 & 3) * 512;   // a table per 16-lane group, as a block of 8x8 samples has its strength
    const uint32_t c = seed | 0x00030001u;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (LOOKUP) {
                const uint32_t idx = (a[k] & SPREAD) + base;
                a[k] = a[k] * 5 + (&lut[0][0])[idx] + 1;   // (v_mad_u32_u24 / lshl_add: one more half-rate op, in every variant)
            } else {
                a[k] = a[k] * 5 + 1;
            }
#pragma unroll
            for (int j = 0; j < VOPS; j++) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[k]) : "v"(c));
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; k++) r ^= a[k] ^ v[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename K>
static double run(K kern, uint32_t *out, const uint8_t *table, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, table, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, table, 7u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int cus = 256, blocks = cus * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    uint32_t *out;
    uint8_t *table, host[2048];
    srand(1);
    for (int i = 0; i < 2048; i++) host[i] = uint8_t(rand());
    hipMalloc(&out, size_t(blocks) * 256 * 4);
    hipMalloc(&table, 2048);
    hipMemcpy(table, host, 2048, hipMemcpyHostToDevice);
    const double ghz = 2.4;
    auto report = [&](const char *name, double ms, int vops, bool lookup) {
        // SIMD cycles per (wave, chain step): 8 waves per SIMD, ITERS x 8 steps each
        const double cyc = ms * 1e-3 * ghz * 1e9 / (8.0 * ITERS * 8);
        printf("%-44s %8.3f ms  %6.2f SIMD cycles per step (%d half-rate ops%s)\n", name, ms, cyc, vops + 1, lookup ? " + 1 lookup" : "");
    };
#define RUN(S, V, L, NAME) report(NAME, run(k_probe<S, V, L>, out, table, blocks), V, L)
    RUN(511, 0, false, "no lookup, 0 extra ops");
    RUN(511, 4, false, "no lookup, 4 extra ops");
    RUN(511, 8, false, "no lookup, 8 extra ops");
    RUN(511, 0, true, "lookup over 512 bytes, 0 extra ops");
    RUN(127, 0, true, "lookup over 128 bytes, 0 extra ops");
    RUN(31, 0, true, "lookup over 32 bytes, 0 extra ops");
    RUN(511, 2, true, "lookup over 512 bytes, 2 extra ops");
    RUN(511, 4, true, "lookup over 512 bytes, 4 extra ops");
    RUN(511, 6, true, "lookup over 512 bytes, 6 extra ops");
    RUN(511, 8, true, "lookup over 512 bytes, 8 extra ops");
    RUN(511, 12, true, "lookup over 512 bytes, 12 extra ops");
    RUN(127, 4, true, "lookup over 128 bytes, 4 extra ops");
    RUN(127, 8, true, "lookup over 128 bytes, 8 extra ops");
    RUN(31, 4, true, "lookup over 32 bytes, 4 extra ops");
    RUN(31, 8, true, "lookup over 32 bytes, 8 extra ops");
    return 0;
}
