// valu_bench.hip -- issue cost of the integer VALU instructions sk_scan_main is made of (gfx950).
// Every kernel runs a long unrolled chain mix of ONE instruction on 8 independent registers,
// 16 waves per CU on all CUs; reports wave-instructions per cycle per SIMD (peak 0.5 = 2 cyc/instr).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define ITERS 2048

#define BODY8(ASM) \
    asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b)); \
    asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b)); \
    asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b)); \
    asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b));

#define KERNEL(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t a, uint32_t b) { \
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    for (int i = 0; i < ITERS; i++) { BODY8(ASM) BODY8(ASM) BODY8(ASM) BODY8(ASM) } \
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345) out[0] = r0; }

KERNEL(k_add,      "v_add_u32 %0, %0, %1")
KERNEL(k_xor,      "v_xor_b32 %0, %0, %1")
KERNEL(k_min,      "v_min_u32 %0, %0, %1")
KERNEL(k_lshl_or,  "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 2")
KERNEL(k_bfe,      "v_bfe_u32 %0, %0, 3, 7")
KERNEL(k_perm,     "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_u24,  "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_add3,     "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL(k_xad,      "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_bfrev,    "v_bfrev_b32 %0, %0")
KERNEL(k_bitop3,   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp,      "v_cmp_ne_u32 vcc, %0, %1")
KERNEL(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_pk_mul16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pk_add16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_min16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_and,      "v_and_b32 %0, %0, %1")
KERNEL(k_or,       "v_or_b32 %0, %0, %1")
KERNEL(k_lshl,     "v_lshlrev_b32 %0, 2, %0")
KERNEL(k_lshr,     "v_lshrrev_b32 %0, 2, %0")
KERNEL(k_sub,      "v_sub_u32 %0, %0, %1")
KERNEL(k_not,      "v_not_b32 %0, %0")
KERNEL(k_min3,     "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_andor,    "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_min_f32,  "v_min_f32 %0, %0, %1")
KERNEL(k_max_f32,  "v_max_f32 %0, %0, %1")
KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_max_u32,  "v_max_u32 %0, %0, %1")
KERNEL(k_min_i32,  "v_min_i32 %0, %0, %1")
KERNEL(k_cnd_e64,  "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL(k_cmp_e64,  "v_cmp_ne_u32_e64 s[10:11], %0, %1")
KERNEL(k_cmp_f32,  "v_cmp_lt_f32_e64 s[10:11], %0, %1")
KERNEL(k_mov,      "v_mov_b32 %0, %1")
KERNEL(k_or3,      "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_bfi,      "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_addlshl,  "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL(k_bfe_i32,  "v_bfe_i32 %0, %0, 3, 1")
KERNEL(k_ashr,     "v_ashrrev_i32 %0, 2, %0")
KERNEL(k_lshl_v,   "v_lshlrev_b32 %0, %1, %0")
KERNEL(k_sdwa,     "v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_cvt,      "v_cvt_f32_u32 %0, %0")
KERNEL(k_mul_f32,  "v_mul_f32 %0, %0, %1")
KERNEL(k_add_f32,  "v_add_f32 %0, %0, %1")
KERNEL(k_fma,      "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_dot4,     "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_dot4acc,  "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_sad_u8,   "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_msad_u8,  "v_msad_u8 %0, %0, %1, %2")
KERNEL(k_and_k,    "v_and_b32 %0, 0x7070707, %0")
KERNEL(k_lshl1,    "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_add_self, "v_add_u32 %0, %0, %0")
KERNEL(k_mul_i24,  "v_mul_i32_i24 %0, %0, %1")


template <typename K> int run(const char *name, K k, uint32_t *out, int regs64)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 4;                      // 16 waves per CU
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double winstr = (double)blocks * 4 * ITERS * 32;            // wave-instructions
    const double per_simd_per_us = winstr / 1024.0 / (ms * 1e3);
    printf("%-18s %8.3f ms   %7.1f wave-instr/us/SIMD  = %5.2f cycles/instr at 2.4 GHz\n", name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
    return 0;
}

int main()
{
    uint32_t *out; CK(hipMalloc(&out, 64));
#define R(k) if (run(#k, k, out, 0)) return 1;
    R(k_add) R(k_xor) R(k_min) R(k_lshl_or) R(k_alignbit) R(k_bfe) R(k_perm) R(k_mul_lo) R(k_mul_u24) R(k_mad_u24) R(k_mul_hi)
    R(k_add3) R(k_lshl_add) R(k_xad) R(k_bfrev) R(k_bitop3) R(k_cndmask) R(k_cmp) R(k_mbcnt)
    R(k_pk_mul16) R(k_pk_add16) R(k_pk_min16) R(k_fma) R(k_and) R(k_or) R(k_lshl) R(k_lshr) R(k_sub) R(k_not) R(k_min3) R(k_andor)
    R(k_min_f32) R(k_max_f32) R(k_min3_f32) R(k_max_u32) R(k_min_i32) R(k_cnd_e64) R(k_cmp_e64) R(k_cmp_f32) R(k_mov) R(k_or3) R(k_bfi) R(k_addlshl) R(k_bfe_i32) R(k_ashr) R(k_lshl_v) R(k_sdwa) R(k_cvt) R(k_mul_f32) R(k_add_f32)
    R(k_dot4) R(k_dot4acc) R(k_sad_u8) R(k_msad_u8) R(k_and_k) R(k_lshl1) R(k_add_self) R(k_mul_i24)
    return 0;
}
