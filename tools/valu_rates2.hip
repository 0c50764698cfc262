// Issue rate of the exact VALU instructions the byte-crunching kernels are made of (gfx950), pinned with inline
// asm so that the compiler cannot fold or re-select them.  8 independent dependency chains per lane, 8 waves per
// SIMD, every CU busy: what one SIMD sustains per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates2.hip -o /tmp/valu_rates2 && /tmp/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define K3(NAME, ASM)                                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed)                              \
    {                                                                                                      \
        uint32_t a[8];                                                                                     \
        for (int k = 0; k < 8; k++) a[k] = threadIdx.x * (2 * k + 3) + seed;                               \
        uint32_t b = seed | 0x00030001u, c = seed * 9 + 0x00050005u;                                       \
        for (int i = 0; i < ITERS; i++) {                                                                  \
            _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
        }                                                                                                  \
        uint32_t r = 0;                                                                                    \
        for (int k = 0; k < 8; k++) r ^= a[k];                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                           \
    }
K3(k_add, "v_add_u32 %0, %0, %1")
K3(k_and, "v_and_b32 %0, %0, %1")
K3(k_lshr, "v_lshrrev_b32 %0, 1, %0")
K3(k_max, "v_max_i32 %0, %0, %1")
K3(k_min, "v_min_u32 %0, %0, %1")
K3(k_add3, "v_add3_u32 %0, %0, %1, %2")
K3(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
K3(k_andor, "v_and_or_b32 %0, %0, %1, %2")
K3(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
K3(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
K3(k_perm, "v_perm_b32 %0, %0, %1, %2")
K3(k_align, "v_alignbyte_b32 %0, %0, %1, 1")
K3(k_alignbit, "v_alignbit_b32 %0, %0, %1, 8")
K3(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
K3(k_madi24, "v_mad_i32_i24 %0, %0, %1, %2")
K3(k_mul24, "v_mul_u32_u24 %0, %0, %1")
K3(k_mullo, "v_mul_lo_u32 %0, %0, %1")
K3(k_med3, "v_med3_i32 %0, %0, %1, %2")
K3(k_max3, "v_max3_i32 %0, %0, %1, %2")
K3(k_min3, "v_min3_i32 %0, %0, %1, %2")
K3(k_sad, "v_sad_u8 %0, %0, %1, %2")
K3(k_msad, "v_msad_u8 %0, %0, %1, %2")
K3(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
K3(k_dot4, "v_dot4_i32_i8 %0, %0, %1, %2")
K3(k_dot4u, "v_dot4_u32_u8 %0, %0, %1, %2")
K3(k_dot2, "v_dot2_i32_i16 %0, %0, %1, %2")
K3(k_pkadd, "v_pk_add_u16 %0, %0, %1")
K3(k_pkaddi, "v_pk_add_i16 %0, %0, %1")
K3(k_pksub, "v_pk_sub_i16 %0, %0, %1")
K3(k_pksubc, "v_pk_sub_u16 %0, %0, %1 clamp")
K3(k_pkmax, "v_pk_max_i16 %0, %0, %1")
K3(k_pkmin, "v_pk_min_i16 %0, %0, %1")
K3(k_pkmaxu, "v_pk_max_u16 %0, %0, %1")
K3(k_pklshr, "v_pk_lshrrev_b16 %0, %1, %0")
K3(k_pkashr, "v_pk_ashrrev_i16 %0, %1, %0")
K3(k_pklshl, "v_pk_lshlrev_b16 %0, %1, %0")
K3(k_pkmul, "v_pk_mul_lo_u16 %0, %0, %1")
K3(k_pkmad, "v_pk_mad_i16 %0, %0, %1, %2")
K3(k_pkmadu, "v_pk_mad_u16 %0, %0, %1, %2")
K3(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K3(k_mov, "v_mov_b32 %0, %1")
K3(k_sub16, "v_sub_u16 %0, %0, %1")
K3(k_max16, "v_max_i16 %0, %0, %1")
K3(k_mad16, "v_mad_u16 %0, %0, %1, %2")
K3(k_cvtpk, "v_cvt_pk_u8_f32 %0, %0, %1, %2")
K3(k_fma, "v_fma_f32 %0, %0, %1, %2")
K3(k_pkfma, "v_pk_fma_f16 %0, %0, %1, %2")
K3(k_or3, "v_or3_b32 %0, %0, %1, %2")
K3(k_lshlor, "v_lshl_or_b32 %0, %0, 8, %1")
K3(k_addlshl, "v_add_lshl_u32 %0, %0, %1, 1")
K3(k_subrev, "v_sub_u32 %0, %1, %0")
K3(k_ashr, "v_ashrrev_i32 %0, 4, %0")
K3(k_cvtub0, "v_cvt_f32_ubyte0 %0, %0")
K3(k_cvtub2, "v_cvt_f32_ubyte2 %0, %0")
K3(k_cvtf32u32, "v_cvt_f32_u32 %0, %0")
K3(k_cvti32f32, "v_cvt_i32_f32 %0, %0")
K3(k_floor, "v_floor_f32 %0, %0")
K3(k_maxf, "v_max_f32 %0, %0, %1")
K3(k_minf, "v_min_f32 %0, %0, %1")
K3(k_med3f, "v_med3_f32 %0, %0, %1, %2")
K3(k_mulf, "v_mul_f32 %0, %0, %1")
K3(k_addf, "v_add_f32 %0, %0, %1")
K3(k_fmac, "v_fmac_f32 %0, %1, %2")
K3(k_orsdwa, "v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
K3(k_mul24sdwa, "v_mul_i32_i24_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
K3(k_xor, "v_xor_b32 %0, %0, %1")
K3(k_lshl, "v_lshlrev_b32 %0, 1, %0")
template <class K> static void run(const char *name, K kern)
{
    uint32_t *d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_insts = double(blocks) * 4 * ITERS * 8;
    const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
    printf("%-22s %8.3f ms  => %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, 2400.0 / per_simd_per_us);
    hipFree(d);
}
#define R(k) run(#k, k)
int main()
{
    R(k_add); R(k_and); R(k_lshr); R(k_ashr); R(k_subrev); R(k_max); R(k_min); R(k_mov); R(k_cndmask);
    R(k_perm); R(k_align); R(k_alignbit);
    R(k_mad24); R(k_madi24); R(k_mul24); R(k_mullo); R(k_med3); R(k_max3); R(k_min3); R(k_sad); R(k_msad); R(k_lerp);
    R(k_dot4); R(k_dot4u); R(k_dot2);
    R(k_pkadd); R(k_pkaddi); R(k_pksub); R(k_pksubc); R(k_pkmax); R(k_pkmin); R(k_pkmaxu); R(k_pklshr); R(k_pkashr); R(k_pklshl);
    R(k_pkmul); R(k_pkmad); R(k_pkmadu);
    R(k_sub16); R(k_max16); R(k_mad16); R(k_cvtpk); R(k_fma); R(k_pkfma);
    R(k_add3); R(k_lshladd); R(k_andor); R(k_or3); R(k_lshlor); R(k_addlshl); R(k_bfe); R(k_bfi); R(k_xor); R(k_lshl);
    R(k_cvtub0); R(k_cvtub2); R(k_cvtf32u32); R(k_cvti32f32); R(k_floor); R(k_maxf); R(k_minf); R(k_med3f); R(k_mulf); R(k_addf); R(k_fmac); R(k_orsdwa); R(k_mul24sdwa);
    return 0;
}
