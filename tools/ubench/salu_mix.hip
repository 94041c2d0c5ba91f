// Does scalar work ride for free beside vector work on gfx950?  8 waves per SIMD, independent chains.
// A: 8 fast-class VALU per block; B: A + 8 SALU interleaved; C: 8 SALU only; D: A + 16 SALU; E: 8 slow-class VALU + 8 SALU;
// F: A + 4 taken-free s_cbranch (never taken) ; G: 8 slow VALU
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define V8 "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
#define W8 "v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8\n"
#define S8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 3\n s_xor_b32 s22, s22, s20\n s_and_b32 s23, s23, s21\n s_add_u32 s24, s24, 5\n s_lshr_b32 s25, s25, 1\n s_or_b32 s26, s26, s24\n s_add_u32 s27, s27, 7\n"
#define MIX "v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 3\n v_add_u32 %2, %2, %8\n s_xor_b32 s22, s22, s20\n v_add_u32 %3, %3, %8\n s_and_b32 s23, s23, s21\n v_add_u32 %4, %4, %8\n s_add_u32 s24, s24, 5\n v_add_u32 %5, %5, %8\n s_lshr_b32 s25, s25, 1\n v_add_u32 %6, %6, %8\n s_or_b32 s26, s26, s24\n v_add_u32 %7, %7, %8\n s_add_u32 s27, s27, 7\n"
#define WMIX "v_perm_b32 %0, %0, %8, %8\n s_add_u32 s20, s20, 1\n v_perm_b32 %1, %1, %8, %8\n s_add_u32 s21, s21, 3\n v_perm_b32 %2, %2, %8, %8\n s_xor_b32 s22, s22, s20\n v_perm_b32 %3, %3, %8, %8\n s_and_b32 s23, s23, s21\n v_perm_b32 %4, %4, %8, %8\n s_add_u32 s24, s24, 5\n v_perm_b32 %5, %5, %8, %8\n s_lshr_b32 s25, s25, 1\n v_perm_b32 %6, %6, %8, %8\n s_or_b32 s26, s26, s24\n v_perm_b32 %7, %7, %8, %8\n s_add_u32 s27, s27, 7\n"
#define BR4 "s_cmp_eq_u32 s20, s21\n s_cbranch_scc1 1f\n1:\n s_cmp_eq_u32 s22, s23\n s_cbranch_scc1 2f\n2:\n"
#define CLOB : "s20","s21","s22","s23","s24","s25","s26","s27","scc"
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = (seed & 7) | 1;
  asm volatile("s_mov_b32 s20, 1\n s_mov_b32 s21, 2\n s_mov_b32 s22, 3\n s_mov_b32 s23, 4\n s_mov_b32 s24, 5\n s_mov_b32 s25, 6\n s_mov_b32 s26, 7\n s_mov_b32 s27, 8" ::CLOB);
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile(V8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 1) { REP16(asm volatile(MIX : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 2) { REP16(asm volatile(S8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 3) { REP16(asm volatile(MIX S8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 4) { REP16(asm volatile(WMIX : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 5) { REP16(asm volatile(V8 BR4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 6) { REP16(asm volatile(W8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
    if (OP == 7) { REP16(asm volatile(V8 S8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) CLOB);) }
  }
  uint32_t s; asm volatile("s_add_u32 %0, s20, s27" : "=s"(s) ::);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s;
}
template <int OP> void run(const char* name, uint32_t* d, int wg) {
  const int iters = 1000, blocks = 256 * wg; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u); (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD %d  %-44s %.3f ms  -> %.2f ns per 8-instruction block per SIMD\n", wg, name, ms, ms * 1e6 / ((double)wg * iters * 16));
}
int main() { uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int wg : {2, 8}) {
    run<0>("A: 8 fast VALU", d, wg); run<2>("C: 8 SALU", d, wg); run<1>("B: 8 fast VALU + 8 SALU interleaved", d, wg); run<7>("B': 8 fast VALU then 8 SALU", d, wg);
    run<3>("D: 8 fast VALU + 16 SALU", d, wg); run<6>("G: 8 slow VALU", d, wg); run<4>("E: 8 slow VALU + 8 SALU interleaved", d, wg); run<5>("F: 8 fast VALU + 2 cmp + 2 untaken branches", d, wg);
  }
  return 0; }
