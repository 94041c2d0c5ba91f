// Does global_load_lds_dwordx4 (gfx950) accept global addresses that are not 16-byte aligned?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_align.hip -o tools/bin/lds_dma_align
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef const __attribute__((address_space(1))) uint8_t* gcptr8;
typedef __attribute__((address_space(3))) void* lptr;
__global__ void k(const uint8_t* src, uint8_t* out, int off, int half) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) buf[i] = 0xee;
    __syncthreads();
    gcptr8 p = (gcptr8)src + off;
    __builtin_amdgcn_global_load_lds(p + lane * 16, (lptr)&buf[0], 16, 0, 0);
    if (!half || lane < 32) __builtin_amdgcn_global_load_lds(p + 1024 + lane * 16, (lptr)&buf[1024], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = buf[i];
}
int main() {
    uint8_t h[4096 + 64], o[2048], *d, *dout;
    for (int i = 0; i < (int)sizeof h; ++i) h[i] = (uint8_t)(i * 7 + (i >> 8));
    hipMalloc(&d, sizeof h); hipMalloc(&dout, 2048);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int half = 0; half < 2; ++half)
    for (int off : {0, 16, 4, 8, 12, 1, 2, 3, 5, 17}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, off, half);
        if (hipDeviceSynchronize() != hipSuccess) { printf("off %d: launch failed\n", off); return 1; }
        hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost);
        int bad = 0, n = half ? 1536 : 2048;
        for (int i = 0; i < n; ++i) bad += o[i] != h[off + i];
        int untouched = 0; for (int i = n; i < 2048; ++i) untouched += o[i] == 0xee;
        printf("half=%d off=%2d: %d mismatching bytes of %d, tail untouched %d/%d\n", half, off, bad, n, untouched, 2048 - n);
        bad_total += bad;
    }
    return bad_total != 0;
}
