// Which lane holds which element of the operands of v_mfma_i32_32x32x32_i8 (gfx950)?  The guide gives the bf16 maps and says "other
// dtypes: check the map with exact integer data".  One wave multiplies random int8 matrices under candidate maps; the map under which
// the result equals the host product is the hardware's.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_i8_probe.hip -o tools/bin/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ int kmap(int cand, int h, int j)
{
    if (cand == 0) return 16 * h + j;                           // natural: 16 consecutive k per lane half
    if (cand == 1) return 8 * h + (j & 7) + 16 * (j >> 3);      // two K=16 halves, each split 8 | 8 over the lane halves
    return 4 * h + (j & 3) + 8 * (j >> 2);                      // four K=8 quarters
}

__global__ void probe(const int8_t *A, const int8_t *B, int *D, int cand)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v4i v; int8_t b[16]; } a, b;
    for (int j = 0; j < 16; j++) {
        const int k = kmap(cand, h, j);
        a.b[j] = A[r * 32 + k];      // A[row r][k]
        b.b[j] = B[k * 32 + r];      // B[k][col r]
    }
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}

int main()
{
    int8_t hA[1024], hB[1024];
    int want[1024], got[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) {
        hA[i] = int8_t(rand() % 255 - 127);
        hB[i] = int8_t(rand() % 255 - 127);
    }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0;
            for (int k = 0; k < 32; k++) s += int(hA[i * 32 + k]) * int(hB[k * 32 + j]);
            want[i * 32 + j] = s;
        }
    int8_t *dA, *dB;
    int *dD;
    hipMalloc(&dA, 1024);
    hipMalloc(&dB, 1024);
    hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    for (int cand = 0; cand < 3; cand++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, cand);
        hipMemcpy(got, dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; i++) bad += got[i] != want[i];
        printf("candidate %d (%s): %d of 1024 results differ\n", cand, cand == 0 ? "k = 16 h + j" : (cand == 1 ? "k = 8 h + (j & 7) + 16 (j >> 3)" : "k = 4 h + (j & 3) + 8 (j >> 2)"), bad);
    }
    return 0;
}
