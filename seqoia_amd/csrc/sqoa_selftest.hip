// sqoa_selftest.hip -- does THIS device's LDS apply the lanes of one instruction in ascending lane order?
//
// The default QOI-compat kernels rest on that order (sqoa_encode.hip, SQOA_ENC_Q_RTN = 3: one `ds_max_rtn_u64` per row hands every pixel the
// most recent earlier writer of its index slot, seqoia.h:563-572; sqoa_decode_qoi.hip, TABW: the resolver's index update is a plain
// `ds_write_b32` of which the highest lane on a slot stays, seqoia.h:785-787).  gfx950 behaves that way (tools/ubench/lds_atomic_order.hip:
// 8 x 10^8 atomics, none out of order) but the ISA manual does not promise it, and a library that guessed wrong would write VALID-LOOKING
// streams with wrong INDEX ops -- silently: the consumer runs no parity test.  So the library carries BOTH forms of those kernels -- the
// order-free ones (per-slot 64-bit lane masks and `ds_max_u64`: round 4's shipping code) as `sqoa::fb::encode_kernel` (this file's sibling build
// of sqoa_encode.hip, -DSQOA_LDS_ORDER_FB) and `decode_qoi_kernel<.., TABW = false>` -- and asks the device, once per device and process, when the
// first plan is created: the probe below, a few hundred microseconds.  Only a device that passes gets the fast forms.
//
//   SQOA_HIP_LDS_ORDER=0   force the order-free kernels (the GPU tests run the whole QOI-compat parity set this way as well)
//   SQOA_HIP_LDS_ORDER=1   trust the order without asking
//
// The reference has no counterpart: it is one thread (seqoia.h:563-582, :785-787).
#include <atomic>
#include <stdlib.h>

#include "sqoa_common.h"

namespace sqoa {

SQOA_DEV uint32_t st_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Eight waves on one workgroup's LDS, as in the kernels.  Per round every lane picks a slot (pattern by `pat`: random over 64 / 8 / 2 slots,
// strided, runs of four lanes, every other lane on one slot, all on one) and -- unless it sits the round out -- does what the kernels do:
//   * an atomic max WITH RETURN of a stamp that grows with round and lane, 32 bits wide and 64 (stamp << 32 | payload): the value handed back
//     must be the stamp of the highest LOWER active lane on the slot, or where there is none an older round's;
//   * a plain 32-bit write of the stamp: what the slot holds afterwards must be the HIGHEST active lane's.
// bad[0] counts returns out of order, bad[1] plain-write winners that are not the highest lane, bad[2] lanes checked (a probe that checked
// nothing has proved nothing).
__global__ __launch_bounds__(512) void lds_order_probe_kernel(uint32_t* __restrict__ bad, int rounds) {
    __shared__ uint32_t cell32[8][64];
    __shared__ unsigned long long cell64[8][64];
    __shared__ uint32_t wcell[8][64];
    const int w = threadIdx.x >> 6, lane = lane_id();
    uint32_t nbad = 0, nbadw = 0, nchk = 0;
    cell32[w][lane] = 0; cell64[w][lane] = 0; wcell[w][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < rounds; ++r) {
        const uint32_t seed = st_mix32((blockIdx.x * 8u + (uint32_t)w) * 0x9E3779B9u + (uint32_t)r);
        const uint32_t rnd = st_mix32(seed ^ ((uint32_t)lane * 0x85EBCA6Bu + 0x01234567u));
        const uint32_t pat = (blockIdx.x + (uint32_t)r) % 7u;
        uint32_t h;
        switch (pat) {
        case 0: h = rnd & 63u; break;
        case 1: h = rnd & 7u; break;
        case 2: h = rnd & 1u; break;
        case 3: h = ((uint32_t)lane * 5u) & 63u; break;
        case 4: h = (uint32_t)lane >> 2; break;
        case 5: h = (lane & 1) ? 7u : (rnd & 63u); break;
        default: h = 11u; break;
        }
        const bool act = pat == 6u || ((rnd >> 8) & 3u) != 0u;   // (a quarter of the lanes sit out)
        const uint32_t base = (uint32_t)r * 64u, stamp = base + (uint32_t)lane + 1u;
        uint32_t prev32 = 0, prev64 = 0;
        if (act) {
            prev32 = __hip_atomic_fetch_max(&cell32[w][h], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            prev64 = (uint32_t)(__hip_atomic_fetch_max(&cell64[w][h], (unsigned long long)stamp << 32 | (rnd | 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 32);
            __hip_atomic_store(&wcell[w][h], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t fin = __hip_atomic_load(&wcell[w][h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        // expected: the highest lower active lane on my slot; the highest active lane on it
        uint32_t want = 0, last = 0;
        for (int j = 0; j < 64; ++j) {
            const uint32_t hj = bcast_lane(h, j), aj = bcast_lane(act ? 1u : 0u, j);
            if (aj && hj == h) { if (j < lane) want = base + (uint32_t)j + 1u; last = base + (uint32_t)j + 1u; }
        }
        if (act) {
            ++nchk;
            if (want ? prev32 != want : prev32 > base) ++nbad;   // (no lower lane of this round: an older round's stamp, or none)
            if (want ? prev64 != want : prev64 > base) ++nbad;
            if (fin != last) ++nbadw;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(&bad[0], nbad);
    if (nbadw) atomicAdd(&bad[1], nbadw);
    atomicAdd(&bad[2], nchk);
}

}  // namespace sqoa

namespace {
constexpr int kMaxDev = 64;
// per device: 0 = not asked yet, 1 = the order holds (probe passed), 2 = it does not, or could not be asked: the order-free kernels
std::atomic<int> g_probe[kMaxDev];
// per device, what the launchers read: 0 = nobody has decided yet (the order-free kernels), 1 = the fast forms, 2 = the order-free kernels.
// Set by sqoa_lds_order_refresh (every plan creation: the environment knob is read there like the plan's other knobs).
std::atomic<int> g_mode[kMaxDev];
}  // namespace

// Runs the probe on `device` (which must be current); 1 = passed.
static int run_probe(void) {
    uint32_t* d_bad = nullptr;
    hipStream_t s = nullptr;
    uint32_t h[3] = {1u, 1u, 0u};
    bool ok = hipMalloc((void**)&d_bad, sizeof h) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
              hipMemsetAsync(d_bad, 0, sizeof h, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(sqoa::lds_order_probe_kernel, dim3(512), dim3(512), 0, s, d_bad, 28);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_bad, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    if (s) hipStreamDestroy(s);
    if (d_bad) hipFree(d_bad);
    return ok && h[0] == 0u && h[1] == 0u && h[2] != 0u;
}

// Called with `device` current, when a plan is created or refilled: decides which form of the QOI-compat kernels launches on this device take
// from now on.  Returns 1 = the fast forms (the LDS's lane order holds, or SQOA_HIP_LDS_ORDER=1 says so), 0 = the order-free ones.
extern "C" int sqoa_lds_order_refresh(int device) {
    if (device < 0 || device >= kMaxDev) return 0;
    int fb;
    const char* k = getenv("SQOA_HIP_LDS_ORDER");
    if (k && k[0] == '0') fb = 1;
    else if (k && k[0] == '1') fb = 0;
    else {
        int st = g_probe[device].load(std::memory_order_acquire);
        if (st == 0) {   // (two threads may both ask: the answer is the same)
            st = run_probe() ? 1 : 2;
            g_probe[device].store(st, std::memory_order_release);
        }
        fb = st != 1;
    }
    g_mode[device].store(fb ? 2 : 1, std::memory_order_release);
    return !fb;
}

// The launchers' question (sqoa_encode.hip, sqoa_decode_qoi.hip): the order-free kernels on the calling thread's current device?
// A device nobody has asked about yet (no plan was ever created on it: not a path the library has) gets them.
extern "C" int sqoa_lds_order_fb(void) {
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDev) return 1;
    return g_mode[device].load(std::memory_order_acquire) != 1;
}

// C-ABI (include/sqoa_hip.h): which form launches on `device` take now -- 1 = the forms that rest on the LDS's lane order (the device passed the
// probe, or SQOA_HIP_LDS_ORDER=1), 0 = the order-free ones; -1 = no such device.  Asks the device if nobody has yet.
extern "C" int sqoa_hip_lds_order(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= kMaxDev) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    return sqoa_lds_order_refresh(device);
}

// The probe's raw counts, for the tests and tools: out[0] = atomic returns out of order, out[1] = plain-write winners that are not the highest
// lane, out[2] = lanes checked.  0 on success.
extern "C" int sqoa_hip_lds_order_probe(int device, unsigned int* out) {
    int ndev = 0;
    if (!out || hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    uint32_t* d_bad = nullptr;
    if (hipMalloc((void**)&d_bad, 12) != hipSuccess) return -1;
    int rc = -1;
    if (hipMemset(d_bad, 0, 12) == hipSuccess) {
        hipLaunchKernelGGL(sqoa::lds_order_probe_kernel, dim3(2048), dim3(512), 0, 0, d_bad, 64);
        if (hipGetLastError() == hipSuccess && hipMemcpy(out, d_bad, 12, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    hipFree(d_bad);
    return rc;
}
