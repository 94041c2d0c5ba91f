// Micro-benchmark: VALU issue rate per SIMD for integer / packed / fp32 ops on gfx950.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed | 1;
    unsigned long long q0 = a0 * 0x100000001ull, q1 = a1 * 0x100000003ull, q2 = a2 * 0x100000005ull, q3 = a3 * 0x100000007ull;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_and_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_lshlrev_b32 %2, 1, %2\n v_or_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_lshrrev_b32 %6, 1, %6\n v_or_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_sub_i16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_lshlrev_b16 %4, 1, %4\n v_pk_add_u16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %1\n v_perm_b32 %1, %1, %8, %2\n v_perm_b32 %2, %2, %8, %3\n v_perm_b32 %3, %3, %8, %4\n v_perm_b32 %4, %4, %8, %5\n v_perm_b32 %5, %5, %8, %6\n v_perm_b32 %6, %6, %8, %7\n v_perm_b32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_sub_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_sub_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_sub_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_sub_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 6) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_bfe_u32 %2, %2, 3, 8\n v_lshl_or_b32 %3, %3, 2, %8\n v_or3_b32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_alignbit_b32 %6, %6, %8, 8\n v_bfi_b32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 8) { REP16(asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b));) }
        if (OP == 9) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_eq_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_eq_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 10) { REP16(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cmp_eq_u32_e64 s[24:25], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[24:25]\n v_cmp_eq_u32_e64 s[26:27], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27");) }
        if (OP == 11) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 12) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %1, %8\n v_alignbyte_b32 %1, %1, %2, %8\n v_alignbyte_b32 %2, %2, %3, %8\n v_alignbyte_b32 %3, %3, %4, %8\n v_alignbyte_b32 %4, %4, %5, %8\n v_alignbyte_b32 %5, %5, %6, %8\n v_alignbyte_b32 %6, %6, %7, %8\n v_alignbyte_b32 %7, %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 13) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 14) { REP16(asm volatile("v_lshrrev_b32 %0, %8, %0\n v_lshrrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_sub_u32 %4, %4, %8\n v_subrev_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}

template <int OP>
void run(const char* name, uint32_t* d, int wg_per_cu) {
    const int iters = 2000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves (256 threads = 4 waves = 1 per SIMD per WG), each iters*128 instrs
    const double inst_per_simd = (double)wg_per_cu * iters * 128.0;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wg_per_cu, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 8}) {
        run<0>("v_add_u32", d, w); run<1>("and/xor/shift/or", d, w); run<2>("v_fma_f32", d, w); run<3>("v_pk_*16", d, w);
        run<4>("v_perm_b32", d, w); run<5>("sdwa add/sub", d, w); run<6>("v_mov_dpp row_shr", d, w); run<7>("vop3 mix", d, w);
        run<8>("v_lsh*_b64", d, w); run<9>("v_cmp e32 + cndmask e32", d, w); run<10>("v_cmp e64 + cndmask e64", d, w); run<11>("mul24/mad24/mul_lo mix", d, w);
        run<12>("v_alignbyte vgpr", d, w); run<13>("v_mul_lo_u32", d, w); run<14>("vop2 shifts/sub/minmax", d, w);
    }
    return 0;
}
