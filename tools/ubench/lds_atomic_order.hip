// In what order does gfx950's LDS apply the lanes of ONE atomic instruction that hit the same address?
// If it is ascending lane order, ds_max_rtn_u32 of (lane + 1) hands every lane the highest LOWER lane on its address -- the QOI-compat encoder's
// "nearest earlier pixel of the row in my index slot" in one instruction, where it now keeps a 64-bit lane mask per slot (ds_or_b64, ds_read_b64,
// a clear, and 64-bit mask arithmetic).  The property tested, over random and adversarial address patterns, with other waves hammering the
// same LDS: for every lane, the returned pre-op value == max stamp over the lower lanes with the same address (else the cell's old value);
// and for plain ds_write_b32 of 64 lanes: the highest lane on an address wins.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
template <class T>
__global__ __launch_bounds__(512) void k(const uint8_t* slots, uint32_t* bad, int rounds, int nslots_pow) {
  __shared__ T cell[8][64];
  __shared__ uint32_t wcell[8][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t nbad = 0, nbadw = 0;
  cell[w][lane] = 0; wcell[w][lane] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int r = 0; r < rounds; ++r) {
    const uint32_t h = slots[((size_t)blockIdx.x * rounds + r) * 512 + threadIdx.x] & ((1u << nslots_pow) - 1u);
    const uint32_t base = (uint32_t)r * 64u;
    const bool act = (slots[((size_t)blockIdx.x * rounds + r) * 512 + (threadIdx.x ^ 1)] & 0x80) == 0 || nslots_pow == 0;   // (some lanes sit out)
    uint32_t prev = 0;
    // (T = 64 bits: the stamp in the high half, a payload in the low half, as the encoder's stamp << 32 | pixel)
    if (act) prev = (uint32_t)(__hip_atomic_fetch_max(&cell[w][h], sizeof(T) == 8 ? (T)((unsigned long long)(base + lane + 1u) << 32 | (lane * 2654435761u)) : (T)(base + lane + 1u),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> (sizeof(T) == 8 ? 32 : 0));
    if (act) wcell[w][h] = base + lane + 1u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t fin = wcell[w][h], finc = (uint32_t)(cell[w][h] >> (sizeof(T) == 8 ? 32 : 0));
    // expected: the highest lower active lane with my slot
    uint32_t want = 0, last = 0;
    for (int j = 0; j < 64; ++j) {
      const uint32_t hj = __shfl(h, j), aj = __shfl((int)act, j);
      if (aj && hj == h) { if (j < lane) want = base + j + 1u; last = base + j + 1u; }
    }
    if (act) {
      if (want ? prev != want : prev > base) ++nbad;          // (no lower lane of this round: an older round's stamp)
      if (fin != last) ++nbadw;
      if (finc != last) ++nbad;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (nbad) atomicAdd(&bad[0], nbad);
  if (nbadw) atomicAdd(&bad[1], nbadw);
}
int main() {
  const int blocks = 1024, rounds = 64;
  const size_t n = (size_t)blocks * rounds * 512;
  uint8_t* h = (uint8_t*)malloc(n); uint8_t* d; uint32_t* bad; (void)hipMalloc(&d, n); (void)hipMalloc(&bad, 8);
  for (int pat = 0; pat < 6; ++pat) {
    srand(1234 + pat);
    for (size_t i = 0; i < n; ++i) {
      uint8_t v = (uint8_t)(rand() >> 7);
      if (pat == 3) v = (uint8_t)(((i & 63) * 5) & 63);          // strided
      if (pat == 4) v = (uint8_t)((i & 63) >> 2);                // runs of four lanes
      if (pat == 5) v = (uint8_t)(((i & 63) & 1) ? 7 : (rand() & 63));
      h[i] = v;
    }
    for (int pw : {6, 3, 1, 0}) {
      (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 8);
      hipLaunchKernelGGL(k<uint32_t>, dim3(blocks), dim3(512), 0, 0, d, bad, rounds, pw);
      uint32_t b[2]; (void)hipMemcpy(b, bad, 8, hipMemcpyDeviceToHost);
      (void)hipMemset(bad, 0, 8);
      hipLaunchKernelGGL(k<unsigned long long>, dim3(blocks), dim3(512), 0, 0, d, bad, rounds, pw);
      uint32_t b64[2]; (void)hipMemcpy(b64, bad, 8, hipMemcpyDeviceToHost);
      b[0] += b64[0]; b[1] += b64[1];
      printf("pattern %d, %2d slots: %zu atomics of 32 and as many of 64 bits, ds_max_rtn out of order %u, plain-write winner not the highest lane %u\n", pat, 1 << pw, n, b[0], b[1]);
    }
  }
  return 0;
}
