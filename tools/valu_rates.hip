// Micro-benchmark: issue rate of the integer VALU / LDS instructions the reconstruction kernels lean
// on (gfx950).  hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define BODY(NAME, STMT)                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed)                    \
    {                                                                                            \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        uint32_t b = seed | 1, c = seed * 9 + 5;                                                 \
        for (int i = 0; i < ITERS; i++) {                                                        \
            STMT(a0) STMT(a1) STMT(a2) STMT(a3) STMT(a4) STMT(a5) STMT(a6) STMT(a7)              \
        }                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
    }
#define S_ADD(x) x = x + b;
#define S_MAD24(x) x = __umul24(x & 0xFFFFFF, b & 0xFFF) + c;
#define S_MULLO(x) x = x * b + c;
#define S_MULHI(x) x = __umulhi(x, b) + c;
#define S_PKADD(x) x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, x) + __builtin_bit_cast(s16x2, b));
#define S_PKMAX(x) x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, c)));
#define S_PKMAD(x) x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, x) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c));
#define S_DOT4(x) x = (uint32_t)__builtin_amdgcn_sdot4((int)x, (int)b, (int)c, false);
#define S_DOT2(x) x = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, b), (int)c, false);
#define S_PERM(x) x = __builtin_amdgcn_perm(x, b, 0x05010400u);
#define S_ALIGN(x) x = __builtin_amdgcn_alignbyte(x, b, 1);
#define S_BFE(x) x = ((x >> 8) & 255) + c;
#define S_MED3(x) x = (uint32_t)min(max((int)x, (int)b), (int)c);
#define S_MAD64(x) x = (uint32_t)(((uint64_t)x * b + c) >> 7);
#define S_SAD(x) x = __builtin_amdgcn_sad_u8(x, b, c);
BODY(k_add, S_ADD) BODY(k_mad24, S_MAD24) BODY(k_mullo, S_MULLO) BODY(k_mulhi, S_MULHI) BODY(k_pkadd, S_PKADD) BODY(k_pkmax, S_PKMAX)
BODY(k_pkmad, S_PKMAD) BODY(k_dot4, S_DOT4) BODY(k_dot2, S_DOT2) BODY(k_perm, S_PERM) BODY(k_align, S_ALIGN) BODY(k_bfe, S_BFE)
BODY(k_med3, S_MED3) BODY(k_mad64, S_MAD64) BODY(k_sad, S_SAD)
__global__ void __launch_bounds__(256) k_lds(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t l[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) l[i] = i * seed;
    __syncthreads();
    uint32_t acc = 0, idx = threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc += l[(idx + 33 * k) & 1023];
        idx += acc & 1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class K> static void run(const char *name, K kern, double ops_per_iter)
{
    uint32_t *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_insts = double(blocks) * 4 * ITERS * ops_per_iter;            // per-wave instructions
    const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
    printf("%-10s %8.3f ms  %7.1f wave-instr/us/SIMD  => %5.2f cycles per wave instruction at 2.4 GHz\n", name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
    hipFree(d);
}
int main()
{
    run("add", k_add, 8); run("mad_u24", k_mad24, 8 * 3); run("mul_lo", k_mullo, 8); run("mul_hi", k_mulhi, 8 * 2); run("pk_add16", k_pkadd, 8);
    run("pk_max16", k_pkmax, 8); run("pk_mad16", k_pkmad, 8); run("dot4_i8", k_dot4, 8); run("dot2_i16", k_dot2, 8); run("perm", k_perm, 8);
    run("alignbyte", k_align, 8); run("bfe+add", k_bfe, 8 * 2); run("med3", k_med3, 8); run("mad64", k_mad64, 8); run("sad_u8", k_sad, 8);
    run("lds_b32", k_lds, 8);
    return 0;
}
