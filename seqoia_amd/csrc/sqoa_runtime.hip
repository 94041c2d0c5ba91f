// sqoa_runtime.hip -- C-ABI batch layer of libsqoa_mi355x.so (include/sqoa_hip.h):
// plans (job tables resident in HBM), launches, timing with HIP events, device
// memory helpers, and the synthetic-input / digest utility kernels.
//
// The reference has no counterpart for this file: it encodes one image per call
// on the host (seqoia.h:363, :374).  Argument validation mirrors the
// reference's own (seqoia.h:465-480) so a job the reference would refuse is
// reported as SQOA_HIP_ST_BADARG instead of being launched.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sqoa_common.h"

extern "C" hipError_t sqoa_launch_encode(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                         sqoa_hip_result* d_results, int cus, hipStream_t stream);
extern "C" hipError_t sqoa_launch_encode_small(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                               sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_wave(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int shape, sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_standby(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, int retry, sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int shape, int n_b, int shape_b, int b_launched, void* join_event, int wave_only, int skip_standby, const sqoa_dp_tables* dp,
                                         sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_encode_strips(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const sqoa_st_tables* st, hipStream_t stream);
extern "C" uint32_t sqoa_dp_chunks(uint32_t src_size, int lb);
extern "C" uint32_t sqoa_dp_per_wg(unsigned long long total_chunks, unsigned wgs, size_t* row_bytes);
extern "C" size_t sqoa_dq_state_bytes(void);
extern "C" hipError_t sqoa_launch_dp_classify(const sqoa_hip_job* d_jobs, const int* d_ids, int n, void* d_word, hipStream_t stream);
extern "C" int sqoa_lds_order_refresh(int device);   // sqoa_selftest.hip

// A stream decoded by one workgroup runs at one compute unit's instruction issue (r = about 1.2 MB of stream per ms) however
// empty the chip is; the three-pass decode costs SQOA_DP_COST x the instructions but spreads them over every CU (R = about
// 660 MB / ms for the chip; tools/time_few.py: 1024x1024 images, one 0.039 against 1.99 ms, 64: 0.54 against 2.16 ms, 256:
// 1.96 against 2.21 ms, 320: 2.43 / 2.54, 512: 3.87 / 2.57, 1024: 7.70 against 3.69 ms).  The plan sorts the streams by size
// and sends the k largest through the passes, k minimising
//       SQOA_DP_COST * bytes(first k) + max(SQOA_DP_SOLO * bytes(stream k), bytes(rest))        [time x R]
// i.e. the passes, then the one-workgroup-per-stream kernel bounded by its largest stream or by the chip.
#ifndef SQOA_DP_COST
#define SQOA_DP_COST 2.1
#endif
#ifndef SQOA_DP_SOLO
#define SQOA_DP_SOLO (550.0 * 1.25)   // R / r, x 1.25: a stream that shares its CU with other streams is slower than alone
#endif
// ... and where the others will take the balanced kernel (sqoa_decode_bal.hip; round 5: 1024 streams of 1 Mpixel 2.59 ms chained, 5.96 in
// three passes; 256 streams, one per compute unit, 1.37 against 1.56): the passes cost 2.3 x, a lone stream 2.3 x 256 x 1.37 / 1.56
// Lists that go mostly to THREE channels: the balanced kernel takes three-channel jobs (12-byte stores), but which kernel wins is a matter of
// content the plan cannot see -- class mixed to three channels 3.10 -> 2.25 ms with it, class photo 2.78 -> 3.19 (ops of even length: LDS bank
// conflicts of the balanced walk's table look-ups; to four channels the same: 2.71 against 3.19, and class mixed 2.98 against 2.57).  Photographs
// are what three-channel lists usually hold: they stay with decode_wave_kernel unless SQOA_HIP_DEC_BAL=1 says otherwise; three-channel jobs
// inside a list that takes the balanced kernel for its four-channel bytes ride along in the one launch (BASELINE configs[4]: 10.50 -> 10.34 ms).
#ifndef SQOA_DEC_BAL_RGB
#define SQOA_DEC_BAL_RGB 0
#endif
#ifndef SQOA_DP_COST_BAL
#define SQOA_DP_COST_BAL 2.3
#define SQOA_DP_SOLO_BAL 520.0
#endif
// The same for the encoder (sqoa_strips.hip): an image alone runs at r = 1.8 Mpixel / ms, the chip at R = 670 Mpixel / ms;
// cut into strips an image costs 1.8 x as much of the chip's time (the cut search, a prologue per strip, one more copy of the
// stream).  tools/time_few_encode.py, 1024x1024 images: one 0.038 against 0.587 ms, 64: 0.21 / 0.63, 256: 0.68 / 0.67,
// 1024: 2.70 / 1.52 ms; one 3840x2160 image 0.060 against 4.56 ms.
#ifndef SQOA_ST_COST
#define SQOA_ST_COST 1.8
#endif
#ifndef SQOA_ST_SOLO
#define SQOA_ST_SOLO (385.0 * 1.25)
#endif
// QOI-compat images (the colour index enters every strip through two more small kernels, sqoa_strips.hip).  The QOI-compat
// encoder is the slower kernel: alone an image runs at 0.755 Mpixel / ms, the chip at 290; cut into strips a full chip's worth of
// images costs 1.83 x (tools/time_single_qoi.py, 1024 x 1024 images: one 0.069 against 1.33 ms, 16: 0.14 / 1.34, 64: 0.45 / 1.36,
// 256: 1.64 / 1.37, 1024: 6.47 / 3.53 ms; one 3840 x 2160 image 0.138 against 10.4 ms).
#ifndef SQOA_STQ_COST
#define SQOA_STQ_COST 1.85
#endif
#ifndef SQOA_STQ_SOLO
#define SQOA_STQ_SOLO 385.0
#endif
// Strips: 16 384 pixels each (32 sub-tiles, four per wave) when that fills the chip; a lone image gets at least one strip per
// compute unit, down to 4096 pixels (8 sub-tiles: one per wave) -- a strip's sub-tiles are one serial chain of position
// hand-overs, and the other CUs are idle anyway (one 1024 x 1024 image: 64 strips 0.036 ms, 256 strips 0.024 ms); never more
// than SQOA_ST_WGS strips per launch (two rounds of workgroups on the chip: a 3840 x 2160 image in 2025 strips took 0.071 ms,
// in 506 strips 0.056 ms -- every strip costs a cut search, a prologue and a piece of the pack).
#ifndef SQOA_ST_MIN_PX
#define SQOA_ST_MIN_PX 4096u
#endif
#ifndef SQOA_ST_PX
#define SQOA_ST_PX 16384u
#endif
#ifndef SQOA_ST_WGS
#define SQOA_ST_WGS 2048u
#endif
// Encode: images of at most SQOA_ENC_SMALL_PX pixels get one wave each when a plan holds at least SQOA_ENC_SMALL_MIN of them
// (sqoa_encode.hip, sqoa_launch_encode_small; a few small images are done sooner by eight waves each).
#ifndef SQOA_ENC_SMALL_PX
#define SQOA_ENC_SMALL_PX 16384ull
#endif
#ifndef SQOA_ENC_SMALL_MIN
#define SQOA_ENC_SMALL_MIN 512u
#endif
#ifndef SQOA_DP_MIN_BYTES
#define SQOA_DP_MIN_BYTES (16u << 10)
#endif

static thread_local char g_err[256] = "";

static int fail(int code, const char* what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

#define HIP_TRY(expr, code)                                   \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return fail(code, #expr, e_);   \
    } while (0)

struct sqoa_hip_plan {
    int device = 0;
    int cus = 256;                      // compute units of `device` (asked when the plan is filled): the launchers' shape choices
    int kind = 0;
    int n = 0, cap = 0;
    sqoa_hip_job* d_jobs = nullptr;
    sqoa_hip_result* d_results = nullptr;
    uint32_t* d_word = nullptr;         // one scratch word (the framing hint of the multi-pass streams, plan_fill)
    int* d_ids = nullptr;               // job indices grouped by kernel variant
    int n_sqoa = 0, n_qoi = 0;          // encode: counts per variant; decode: n_sqoa = all valid jobs
    int n_sqoa_small = 0, n_qoi_small = 0;   // encode: how many at the end of each list are small images that get one wave each
    int small = 0;                      // decode: many short streams (more images than the chip holds large workgroups)
    int shape = 0;                      // decode, the streams that get a workgroup each: the wave decoder's shape (sqoa_launch_decode_wave) -- bit 0 = many short
                                        // streams, bit 1 = 12 stream bytes per lane, bit 2 = position-major stores
    int shape_nobal = 0;                // decode: the shape the same streams get WITHOUT the balanced kernel -- taken from the launch after the one whose result
                                        // rows say that the content is of the kind that kernel is slow on (row_content_hint, sqoa_common.h)
    bool content_switched = false;
    std::vector<sqoa_hip_job> h_jobs;   // decode: the job table and the id list as uploaded (sqoa_hip_plan_results re-partitions the list by content)
    std::vector<int> h_ids;
    int n_multi = 0;                    // decode: the first n_multi ids take the three chip-wide passes
    int n_chain_b = 0, shape_b = 0;     // decode: the last n_chain_b of the streams that get a workgroup each are launched apart, with a shape of their own
                                        // (a plan of mixed output formats: its four-channel streams take the balanced kernel, the others what suits them)
    // decode, the two kernels behind the wave decoder (QOI-compat / mono / 1-2 channels / SQOA_OP_REF): launched with every launch until a launch's
    // result rows -- read by sqoa_hip_plan_results -- say that no job needed them; from then on left out, and sqoa_hip_plan_results runs them itself
    // for a launch whose rows come back SLOWPATH after all (the streams in the buffers have changed)
    bool lazy_skip = false, skipped_last = false, launched = false;
    int wave_only = 0;                  // decode: every job carries SQOA_HIP_JOB_WAVEONLY -- the kernels behind the SQOA wave decoder are not launched
    sqoa_dp_tables dp = {nullptr, 0, nullptr, 0, 16, nullptr, nullptr, nullptr, 1, 256, 8};   // decode: the few large streams that take the three-pass route
    size_t dp_images_cap = 0, dp_flags_cap = 0, dp_wgs_cap = 0, dp_tab_cap = 0, dp_qstate_cap = 0;   // allocated rows
    sqoa_st_tables st = {nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr};   // encode: the few large images cut into strips
    int n_cut_sqoa = 0, n_cut_qoi = 0;   // how many at the front of the SQOA / QOI-compat id list they are
    size_t st_images_cap = 0, st_strips_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_done = nullptr;       // recorded behind the kernels of the most recent launch, on whatever stream it used
    hipStream_t side = nullptr;         // decode plans that launch a part of their streams apart (n_chain_b): that part runs here, forked from and joined to the launch's stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool pending = false;               // ev_done has been recorded since the last wait
};

// Wait for the most recent launch of the plan, whichever stream it went to (a launch captured into a graph records
// nothing: whoever replays the graph synchronises the replay).
static hipError_t plan_wait(sqoa_hip_plan* p) {
    if (p->pending) {
        hipError_t e = hipEventSynchronize(p->ev_done);
        if (e != hipSuccess) return e;
        p->pending = false;
    }
    return hipStreamSynchronize(p->stream);
}

extern "C" {

const char* sqoa_hip_last_error(void) { return g_err; }

int sqoa_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// w*h*(ch+1) + 14 + 8 (reference seqoia.h:487-489) + 1 (start byte, seqoia.h:513)
static size_t encode_raw_bound(const sqoa_desc* d) {
    if (!d || d->width == 0 || d->height == 0 || d->channels < 1 || d->channels > 6) return 0;
    const unsigned ch = (d->channels < 3 ? 1u : 3u) + ((d->channels & 1) ? 0u : 1u);
    return (size_t)d->width * d->height * (ch + 1) + 14 + 8 + 1;
}

size_t sqoa_hip_encode_bound(const sqoa_desc* d) { return (encode_raw_bound(d) + 255) & ~(size_t)255; }

int sqoa_hip_malloc(int device, size_t bytes, unsigned long long* dptr) {
    if (!dptr) return fail(SQOA_HIP_EINVAL, "sqoa_hip_malloc: null out pointer");
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 1), SQOA_HIP_ENOMEM);
    *dptr = (unsigned long long)(uintptr_t)p;
    return 0;
}

int sqoa_hip_free(int device, unsigned long long dptr) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipFree((void*)(uintptr_t)dptr), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_memcpy_h2d(int device, unsigned long long dst, const void* src, size_t bytes) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipMemcpy((void*)(uintptr_t)dst, src, bytes, hipMemcpyHostToDevice), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_memcpy_d2h(int device, void* dst, unsigned long long src, size_t bytes) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipMemcpy(dst, (const void*)(uintptr_t)src, bytes, hipMemcpyDeviceToHost), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_host_alloc(size_t bytes, void** p) {
    if (!p) return fail(SQOA_HIP_EINVAL, "sqoa_hip_host_alloc: null out pointer");
    HIP_TRY(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault), SQOA_HIP_ENOMEM);
    return 0;
}

int sqoa_hip_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_stream_create(int device, void** stream) {
    if (!stream) return fail(SQOA_HIP_EINVAL, "sqoa_hip_stream_create: null out pointer");
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), SQOA_HIP_ENOMEM);
    *stream = (void*)s;
    return 0;
}

int sqoa_hip_stream_destroy(int device, void* stream) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_stream_sync(int device, void* stream) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream), SQOA_HIP_ELAUNCH);
    return 0;
}

int sqoa_hip_memcpy_h2d_async(int device, unsigned long long dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipMemcpyAsync((void*)(uintptr_t)dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), SQOA_HIP_EINVAL);
    return 0;
}

int sqoa_hip_memcpy_d2h_async(int device, void* dst, unsigned long long src, size_t bytes, void* stream) {
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    HIP_TRY(hipMemcpyAsync(dst, (const void*)(uintptr_t)src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), SQOA_HIP_EINVAL);
    return 0;
}

// The reference's argument checks for sqoa_encode (seqoia.h:465-480).
static bool encode_job_ok(const sqoa_hip_job& j) {
    const sqoa_desc& d = j.desc;
    if (!j.src || !j.dst || d.width == 0 || d.height == 0 || d.channels < 1 || d.channels > 6 || d.colorspace > 1) return false;
    if (d.height >= 400000000u / d.width) return false;
    if (d.channels < 3 && d.qoi_compat) return false;
    if (j.reserved & ~(SQOA_HIP_JOB_NOHEADER | SQOA_HIP_JOB_NOEND | SQOA_HIP_JOB_PREV)) return false;
    if (j.reserved && d.qoi_compat) return false;  // strip jobs: SQOA-native only (the colour index would have to cross the cut)
    return true;
}

// Every second run of `cus` ids reversed, from `first` on.  The dispatcher hands workgroups to the compute units round-robin,
// so of a batch that fits the chip at once (four streams per CU) CU k gets ranks k, C + k, 2C + k, 3C + k of a sorted list:
// CU 0 would get the largest stream of every quarter.  Boustrophedon order evens the CUs' loads out (headline decode: max / mean
// CU load 1.026 -> 1.001, 3.74 -> 3.70 ms).
static void snake(std::vector<int>& v, size_t first, int cus, size_t last = (size_t)-1) {
    if (last > v.size()) last = v.size();
    for (size_t a = first + cus; a < last; a += 2 * (size_t)cus) std::reverse(v.begin() + a, v.begin() + std::min(last, a + cus));
}

// Dispatch order of a decode plan's streams and the route each takes: pure host arithmetic, no device needed.
// order[0 .. *n_multipass) = the streams decoded in three chip-wide passes (largest first), order[*n_multipass .. n) = the ones
// that get a workgroup each.  `mode`: 0 = the cost model above, -1 = no stream multi-pass, 1 = every stream.
static int decode_route(const unsigned int* src_sizes, int n, int cus, int mode, double dp_cost, double dp_solo, int* order, int* n_multipass);
int sqoa_hip_decode_route(const unsigned int* src_sizes, int n, int cus, int mode, int* order, int* n_multipass) {
    return decode_route(src_sizes, n, cus, mode, SQOA_DP_COST, SQOA_DP_SOLO, order, n_multipass);
}
// (`dp_cost`, `dp_solo`: the model's two constants -- the passes' cost per stream byte and a lone stream's on its one compute unit, in
// units of the one-workgroup-per-stream kernel's cost per byte on a full chip -- for the kernel that will decode the others: plan_fill)
static int decode_route(const unsigned int* src_sizes, int n, int cus, int mode, double dp_cost, double dp_solo, int* order, int* n_multipass) {
    if (!src_sizes || !order || !n_multipass || n < 0 || cus <= 0) return fail(SQOA_HIP_EINVAL, "sqoa_hip_decode_route: bad arguments");
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = i;
    // one workgroup per stream, dispatched in id order: the largest first, so that a batch of mixed sizes (BASELINE.json
    // configs[4]) does not end with one screenshot-sized stream running alone
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return src_sizes[a] > src_sizes[b]; });
    double bytes = 0;
    for (int i = 0; i < n; ++i) bytes += src_sizes[i];
    size_t k_best = 0;
    if (mode > 0) k_best = (size_t)n;
    else if (mode == 0) {
        double best = 0, head = 0;
        for (size_t k = 0; k <= (size_t)n; ++k) {
            const double bk = k < (size_t)n ? (double)src_sizes[ids[k]] : 0.0;
            const double rest = bytes - head;
            const double cost = dp_cost * head + (dp_solo * bk > rest ? dp_solo * bk : rest);
            if (k == 0 || cost < best) { best = cost; k_best = k; }
            if (k == (size_t)n || src_sizes[ids[k]] < SQOA_DP_MIN_BYTES) break;
            head += bk;
        }
    }
    snake(ids, k_best, cus);
    memcpy(order, ids.data(), sizeof(int) * (size_t)n);
    *n_multipass = (int)k_best;
    return 0;
}

// The same choice for an encode plan's SQOA-native images: order[0 .. *n_strip_images) = the images cut into strips (largest
// first; strips[k] = how many for order[k]), the rest one workgroup each.  `pixels[i]` = width * height.
static int encode_route(const unsigned long long* pixels, int n, int cus, int mode, double st_cost, double st_solo, int* order, int* strips, int* n_strip_images);
int sqoa_hip_encode_route(const unsigned long long* pixels, int n, int cus, int mode, int* order, int* strips, int* n_strip_images) {
    return encode_route(pixels, n, cus, mode, SQOA_ST_COST, SQOA_ST_SOLO, order, strips, n_strip_images);
}
static int encode_route(const unsigned long long* pixels, int n, int cus, int mode, double st_cost, double st_solo, int* order, int* strips, int* n_strip_images) {
    if (!pixels || !order || !strips || !n_strip_images || n < 0 || cus <= 0) return fail(SQOA_HIP_EINVAL, "sqoa_hip_encode_route: bad arguments");
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = i;
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return pixels[a] > pixels[b]; });
    double all = 0;
    for (int i = 0; i < n; ++i) all += (double)pixels[i];
    size_t k_best = 0;
    double head_best = 0;
    if (mode >= 0) {
        double best = 0, head = 0;
        for (size_t k = 0; k <= (size_t)n; ++k) {
            const double bk = k < (size_t)n ? (double)pixels[ids[k]] : 0.0;
            const double rest = all - head;
            const double cost = st_cost * head + (st_solo * bk > rest ? st_solo * bk : rest);
            if (mode > 0 || k == 0 || cost < best) { best = cost; k_best = k; head_best = head; }
            if (k == (size_t)n || pixels[ids[k]] < 2ull * SQOA_ST_PX || pixels[ids[k]] > 0xffffffffull) break;
            head += bk;
        }
    }
    // pixels per strip (see SQOA_ST_PX above)
    double want = head_best / SQOA_ST_PX;
    if (want < cus) want = cus;
    if (want > SQOA_ST_WGS) want = SQOA_ST_WGS;
    double per = head_best / want;
    if (per < SQOA_ST_MIN_PX) per = SQOA_ST_MIN_PX;
    for (size_t k = 0; k < (size_t)n; ++k) {
        double s = k < k_best ? (double)pixels[ids[k]] / per : 0.0;
        strips[k] = k < k_best ? (s < 2.0 ? 2 : (int)s) : 0;
    }
    snake(ids, k_best, cus);
    memcpy(order, ids.data(), sizeof(int) * (size_t)n);
    *n_strip_images = (int)k_best;
    return 0;
}

// Which shape of the wave decoder a part of a decode plan's id list gets (ids[a .. b) of `jobs`): what the plan can see without reading a stream --
// output channels, image shapes, stream sizes -- and the knobs (plan_fill has the reasons; sqoa_hip_plan_results calls it again when it splits a
// list by what the launches saw of the content).  bit 0 many short streams, bit 1 twelve stream bytes per lane, bit 2 position-major stores,
// bit 3 the balanced kernel, bit 4 its three-channel instance.
static int decode_shape_of(const sqoa_hip_job* jobs, const int* ids, size_t a, size_t b, int small, bool allow_bal) {
    if (a >= b) return 0;
    const char* kw = getenv("SQOA_HIP_DEC_WIDE");
    const char* ks = getenv("SQOA_HIP_DEC_SWEEP");
    const char* kb = getenv("SQOA_HIP_DEC_BAL");
    unsigned long long by = 0, by4 = 0, px = 0;
    bool shaped = true, all4 = true, any3 = false;
    for (size_t k = a; k < b; ++k) {
        const sqoa_hip_job& j = jobs[ids[k]];
        by += j.src_size;
        px += (unsigned long long)j.desc.width * j.desc.height;
        if (j.desc.channels == 4) by4 += j.src_size;
        // what the balanced kernel takes: four channels to a dword-aligned address, or three (round 5: 12-byte stores at any byte address)
        if (!(j.desc.channels == 3 || (j.desc.channels == 4 && !(j.dst & 3ull)))) all4 = false;
        if (j.desc.channels != 4) any3 = true;
        if (!j.desc.width || !j.desc.height || j.desc.channels < 3) shaped = false;
    }
    const bool roomy = by / (b - a) >= (12u << 10);
    int wide = roomy && 4u * by4 >= 3u * by, sweep = shaped && roomy && by < px;
    if (kw) wide = kw[0] == '1' ? 1 : kw[0] == '0' ? 0 : wide;
    if (ks) sweep = ks[0] == '1' ? 1 : ks[0] == '0' ? 0 : sweep;
    //  * the balanced kernel (sqoa_decode_bal.hip: ops dealt out evenly over the lanes, 1024-byte chunks) where the wide shape
    //    would be picked for streams that are many chunks long and not run-heavy (SQOA_HIP_DEC_BAL = 0 / 1 forces it)
    //    -- a kernel for four channels to dword-aligned addresses and (its RGB instance, bit 4) for three channels, so only for lists
    //    whose every job is one of the two; a list to three channels alone keeps the wave decoder (SQOA_DEC_BAL_RGB above)
    int bal = (wide || (roomy && SQOA_DEC_BAL_RGB && all4)) && !sweep && !small;
    if (kb) bal = kb[0] == '1' ? 1 : kb[0] == '0' ? 0 : bal;
    bal = bal && all4 && allow_bal;
    return small | wide << 1 | sweep << 2 | bal << 3 | (bal && any3 ? 1 : 0) << 4;   // (bit 4: the balanced kernel's instance that takes three-channel jobs)
}

// Classify jobs, fill the pre-launch result rows and the per-variant id lists, upload all three.
static int plan_fill(sqoa_hip_plan* p, const sqoa_hip_job* jobs, int n) {
    std::vector<sqoa_hip_result> init(n ? n : 1);
    std::vector<int> ids_a, ids_b;
    for (int i = 0; i < n; ++i) {
        memset(&init[i], 0, sizeof(sqoa_hip_result));
        if (p->kind == SQOA_HIP_ENCODE) {
            if (!encode_job_ok(jobs[i])) { init[i].status = SQOA_HIP_ST_BADARG; continue; }
            if (jobs[i].dst_cap < encode_raw_bound(&jobs[i].desc)) { init[i].status = SQOA_HIP_ST_OVERFLOW; continue; }
            (jobs[i].desc.qoi_compat ? ids_b : ids_a).push_back(i);
        } else {
            // sqoa_decode's argument checks (seqoia.h:662-668); header checks happen in the kernel
            if (!jobs[i].src || !jobs[i].dst || jobs[i].desc.channels > 4 || jobs[i].src_size < 22 || (jobs[i].reserved & ~SQOA_HIP_JOB_WAVEONLY)) {
                init[i].status = SQOA_HIP_ST_BADARG;
                continue;
            }
            ids_a.push_back(i);
        }
    }
    // whatever fails below, the plan must not keep the previous fill's route tables beside the new id list
    p->st.n_images = p->st.n_strips = p->st.n_images_qoi = p->st.n_strips_qoi = 0;
    p->n_cut_sqoa = p->n_cut_qoi = 0;
    p->dp.n_images = p->dp.n_wgs = 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || cus <= 0) cus = 256;
    p->cus = cus;
    p->dp.cus = cus;
    // Which form of the QOI-compat kernels launches on this device take (the device is asked ONCE per process, here, before anything could be
    // capturing a stream; SQOA_HIP_LDS_ORDER=0 / 1 is read with every fill like the plan's other knobs): sqoa_selftest.hip
    (void)sqoa_lds_order_refresh(p->device);
    typedef sqoa::dp_image dp_image_row;   // (sqoa_common.h: the kernels read the very same structs)
    typedef sqoa::dp_wg dp_wg_row;
    std::vector<dp_image_row> dp_images;
    std::vector<dp_wg_row> dp_wgs;
    size_t dp_rows = 0, dp_row_bytes = 32;
    int n_multi_dec = 0;
    typedef sqoa::st_image st_image_row;
    typedef sqoa::st_strip st_strip_row;
    std::vector<st_image_row> st_images;
    std::vector<st_strip_row> st_strips;
    size_t st_ws = 0, st_images_sqoa = 0, st_strips_sqoa = 0;   // (the QOI-compat images' rows lie behind the SQOA-native ones')
    int n_cut_a = 0, n_cut_b = 0;
    if (p->kind == SQOA_HIP_ENCODE) {
        // largest image first (see sqoa_hip_decode_route), then boustrophedon; the QOI-compat list as it is
        auto npx = [&](int i) { return (unsigned long long)jobs[i].desc.width * jobs[i].desc.height; };
        // Many SMALL images get one wave each instead of eight (sqoa_launch_encode_small): they sit at the end of a list sorted
        // by size, and each part of the list is put into boustrophedon order by itself.
        auto split_small = [&](std::vector<int>& v, size_t first) -> int {
            std::stable_sort(v.begin() + first, v.end(), [&](int a, int b) { return npx(a) > npx(b); });
            size_t cut = v.size();
            while (cut > first && npx(v[cut - 1]) <= SQOA_ENC_SMALL_PX) --cut;
            if (v.size() - cut < SQOA_ENC_SMALL_MIN) cut = v.size();
            snake(v, first, cus, cut);
            snake(v, cut, cus);
            return (int)(v.size() - cut);
        };
        // SQOA_HIP_STRIPS=0 / =1 in the environment: never / every image that is large enough (tests and measurements).  A
        // caller that sends strip jobs of its own has done the cutting already.  The images cut come first in their list,
        // SQOA-native ones first in the tables, QOI-compat ones behind them (sqoa_launch_encode_strips).
        const char* knob = getenv("SQOA_HIP_STRIPS");
        const int knob_mode = knob && knob[0] == '1' ? 1 : knob && knob[0] == '0' ? -1 : 0;
        auto route_list = [&](std::vector<int>& ids, bool qoi, int* n_small) -> int {
            *n_small = 0;
            if (ids.empty()) return 0;
            int mode = knob_mode;
            std::vector<unsigned long long> px(ids.size());
            for (size_t k = 0; k < ids.size(); ++k) { px[k] = npx(ids[k]); if (jobs[ids[k]].reserved) mode = -1; }
            std::vector<int> order(ids.size()), nst(ids.size());
            int n_cut = 0;
            int rc = encode_route(px.data(), (int)px.size(), cus, mode, qoi ? SQOA_STQ_COST : SQOA_ST_COST, qoi ? SQOA_STQ_SOLO : SQOA_ST_SOLO,
                                  order.data(), nst.data(), &n_cut);
            if (rc) return -1;
            std::vector<int> sorted(ids.size());
            for (size_t k = 0; k < ids.size(); ++k) sorted[k] = ids[order[k]];
            ids.swap(sorted);
            for (int k = 0; k < n_cut; ++k) {
                const int i = ids[k];
                const uint32_t ch = (jobs[i].desc.channels < 3 ? 1u : 3u) + ((jobs[i].desc.channels & 1) ? 0u : 1u);
                const uint32_t first = (uint32_t)st_strips.size(), ns = (uint32_t)nst[k];
                const uint32_t img = (uint32_t)st_images.size();
                st_images.push_back({i, first, ns, (uint32_t)npx(i), (unsigned long long)st_ws});   // (.ws: offset, the base is added below)
                for (uint32_t q = 0; q < ns; ++q) st_strips.push_back({img, q});
                st_ws += (((size_t)npx(i) * (ch + 1) + 255) & ~(size_t)255) + (size_t)ns * 512 + 512;
            }
            *n_small = split_small(ids, (size_t)n_cut);
            return n_cut;
        };
        const int cut_a = route_list(ids_a, false, &p->n_sqoa_small);
        if (cut_a < 0) return SQOA_HIP_EINVAL;
        st_images_sqoa = st_images.size(); st_strips_sqoa = st_strips.size();
        const int cut_b = route_list(ids_b, true, &p->n_qoi_small);
        if (cut_b < 0) return SQOA_HIP_EINVAL;
        n_cut_a = cut_a; n_cut_b = cut_b;
    } else if (!ids_a.empty()) {
        unsigned long long bytes = 0;
        std::vector<unsigned int> sizes(ids_a.size());
        for (size_t k = 0; k < ids_a.size(); ++k) { sizes[k] = jobs[ids_a[k]].src_size; bytes += sizes[k]; }
        p->small = ids_a.size() > 2048 && bytes / ids_a.size() < (256u << 10);
        p->wave_only = 1;
        for (int i : ids_a) if (!(jobs[i].reserved & SQOA_HIP_JOB_WAVEONLY)) p->wave_only = 0;
        // SQOA_HIP_MULTIPASS=0 / =1 in the environment: never / every stream (tests and measurements; the results are the same)
        const char* knob = getenv("SQOA_HIP_MULTIPASS");
        const int mode = knob && knob[0] == '1' ? 1 : knob && knob[0] == '0' ? -1 : 0;
        std::vector<int> order(ids_a.size());
        int n_multi = 0;
        // (streams that will take the balanced kernel: it is 12 % faster on a full chip and a third faster with a compute unit to itself than the
        // kernel the model's constants were fitted to: tools/time_few.py, 256 streams of 1 Mpixel 1.37 ms chained against 1.56 in three passes,
        // 1024 of them 2.59 against 5.96.  The SAME predicate as shape_of below -- over the whole list, the multi-pass prefix is not known yet --,
        // so that a list the balanced kernel will not take -- to three channels alone, run-heavy, mostly not four-channel bytes -- is not
        // routed with constants fitted to a kernel it does not run)
        bool bal_cand = false;
        {
            unsigned long long by4 = 0, px = 0;
            bool shaped = true, all4 = true;
            for (int i : ids_a) {
                const sqoa_hip_job& j = jobs[i];
                px += (unsigned long long)j.desc.width * j.desc.height;
                if (j.desc.channels == 4) by4 += j.src_size;
                if (!(j.desc.channels == 3 || (j.desc.channels == 4 && !(j.dst & 3ull)))) all4 = false;
                if (!j.desc.width || !j.desc.height || j.desc.channels < 3) shaped = false;
            }
            const bool roomy = bytes / ids_a.size() >= (12u << 10);
            int wide = roomy && 4u * by4 >= 3u * bytes, sweep = shaped && roomy && bytes < px;
            if (const char* kw = getenv("SQOA_HIP_DEC_WIDE")) wide = kw[0] == '1' ? 1 : kw[0] == '0' ? 0 : wide;
            if (const char* ks = getenv("SQOA_HIP_DEC_SWEEP")) sweep = ks[0] == '1' ? 1 : ks[0] == '0' ? 0 : sweep;
            int bal = (wide || (roomy && SQOA_DEC_BAL_RGB && all4)) && !sweep && !p->small;
            if (const char* kb = getenv("SQOA_HIP_DEC_BAL")) bal = kb[0] == '1' ? 1 : kb[0] == '0' ? 0 : bal;
            bal_cand = bal && all4;
        }
        int rc = decode_route(sizes.data(), (int)sizes.size(), cus, mode, bal_cand ? SQOA_DP_COST_BAL : SQOA_DP_COST, bal_cand ? SQOA_DP_SOLO_BAL : SQOA_DP_SOLO, order.data(), &n_multi);
        if (rc) return rc;
        std::vector<int> sorted(ids_a.size());
        for (size_t k = 0; k < ids_a.size(); ++k) sorted[k] = ids_a[order[k]];
        ids_a.swap(sorted);
        n_multi_dec = n_multi;   // (their tables: below, once the job table is on the device)
        // The streams that get a workgroup each: which shape of the wave decoder (sqoa_launch_decode_wave)?  What the plan can see
        // without reading a stream is what the jobs say -- output channels, image shape -- and the streams' sizes:
        //  * 12 stream bytes per lane instead of 8 ("wide") when the streams that go to FOUR channels are three quarters of the launch's
        //    bytes or more (to three channels the byte stores are the bound and the wide shape loses 9 % where to four it gains as much:
        //    BASELINE configs[4]'s rank shard, 83 % four-channel bytes, 12.11 -> 11.38 ms) and the streams are not tiny (a few kilobytes are a handful of chunks
        //    either way, and fewer, larger ones leave waves idle: 4096 icons of 128 x 128, 4 KB each, 0.130 -> 0.140 ms; 16 KB: 0.516 -> 0.508);
        //  * the kernel with the position-major store path ("sweep") for RUN-heavy streams to 3 or 4 channels: under one stream byte
        //    per pixel over the launch, where the jobs state the shapes they expect.
        // (A plan of mixed channel counts as two launches, the four-channel jobs first with their own shape, was measured on BASELINE
        // configs[4]'s rank shard -- 557 RGB photos among 7 635 RGBA images -- and lost, 12.11 -> 12.37 ms: the second launch's streams
        // no longer share the chip with the first's.  One launch, one shape.)
        // (SQOA_HIP_DEC_WIDE / SQOA_HIP_DEC_SWEEP = 0 / 1 in the environment force either: tests and measurements; the results are the same.)
        {
            const size_t c0 = (size_t)n_multi;
            // what the balanced kernel takes: four channels to a dword-aligned address, or three (round 5: 12-byte stores at any byte address)
            auto four = [&](int i) { return jobs[i].desc.channels == 3 || (jobs[i].desc.channels == 4 && !(jobs[i].dst & 3ull)); };
            auto shape_of = [&](size_t a, size_t b, bool allow_bal = true) -> int { return decode_shape_of(jobs, ids_a.data(), a, b, p->small, allow_bal); };
            p->shape = shape_of(c0, ids_a.size());
            p->shape_nobal = shape_of(c0, ids_a.size(), false);
            p->n_chain_b = 0; p->shape_b = 0;
            // A list of mixed output formats (BASELINE configs[4]: 557 RGB photos to three channels among 7 635 RGBA images): the balanced kernel
            // is 12 % faster on the four-channel streams, which are most of such a list -- they go first, in a launch of their own, and the
            // others behind them with the shape that suits THEM, on a side stream so that the two launches share the chip (one behind the
            // other on one stream, the second launch's few long streams would have the chip to themselves: measured in round 4, lost).
            // Where the four-channel streams are less than half of the bytes, or would not take the balanced kernel alone: one launch as before.
            // (SQOA_HIP_DEC_SPLIT = 0 / 1 forces it.)
            if (!(p->shape & 8) && ids_a.size() - c0 >= 2) {
                std::vector<int> a4, rest;
                unsigned long long by4 = 0, by = 0;
                for (size_t k = c0; k < ids_a.size(); ++k) {
                    (four(ids_a[k]) ? a4 : rest).push_back(ids_a[k]);
                    by += jobs[ids_a[k]].src_size;
                    if (four(ids_a[k])) by4 += jobs[ids_a[k]].src_size;
                }
                int split = !a4.empty() && !rest.empty() && 2 * by4 >= by;
                if (const char* kp = getenv("SQOA_HIP_DEC_SPLIT")) split = kp[0] == '1' ? (!a4.empty() && !rest.empty()) : kp[0] == '0' ? 0 : split;
                if (split) {
                    std::vector<int> keep(ids_a.begin() + (long)c0, ids_a.end());
                    std::copy(a4.begin(), a4.end(), ids_a.begin() + (long)c0);            // (both halves keep the largest-first order)
                    std::copy(rest.begin(), rest.end(), ids_a.begin() + (long)(c0 + a4.size()));
                    const int sa = shape_of(c0, c0 + a4.size());
                    if (sa & 8) {
                        p->shape = sa;
                        p->shape_nobal = shape_of(c0, c0 + a4.size(), false);
                        p->shape_b = shape_of(c0 + a4.size(), ids_a.size());
                        p->n_chain_b = (int)rest.size();
                    } else std::copy(keep.begin(), keep.end(), ids_a.begin() + (long)c0);   // (the four-channel streams alone would not take it either)
                }
            }
        }
    }
    p->lazy_skip = p->skipped_last = p->launched = false;
    p->content_switched = false;
    p->n = n;
    p->n_sqoa = (int)ids_a.size();
    p->n_qoi = (int)ids_b.size();
    ids_a.insert(ids_a.end(), ids_b.begin(), ids_b.end());
    if (p->kind == SQOA_HIP_DECODE) { p->h_jobs.assign(jobs, jobs + n); p->h_ids = ids_a; p->n_multi = n_multi_dec; }
    if (n == 0) return 0;
    hipError_t e;
    if ((e = hipMemcpy(p->d_jobs, jobs, sizeof(sqoa_hip_job) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
        // (the result rows as this fill leaves them -- BADARG / OVERFLOW rows of refused jobs, zeros elsewhere --, ONCE per fill and straight from
        // the host: a pageable-host-to-device hipMemcpy has returned only when the copy is done, which a device-to-device one need not)
        (e = hipMemcpy(p->d_results, init.data(), sizeof(sqoa_hip_result) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
        (!ids_a.empty() && (e = hipMemcpy(p->d_ids, ids_a.data(), sizeof(int) * ids_a.size(), hipMemcpyHostToDevice)) != hipSuccess))
        return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan: job table upload", e);
    p->dp.any_qoi = 0;
    if (n_multi_dec > 0) {
        // Which framing do the multi-pass streams use?  One byte of each (byte 14), read where it lies: the streams of a plan's few
        // LARGE jobs must be readable device memory when the plan is created or updated (sqoa_hip.h says so), and what a producer
        // on another stream has not written yet reads as whatever is there -- which is why this is only a hint.  It picks the shape
        // of the tables and spares plans without QOI-compat streams the speculative pass's three launches; a stream it misses
        // stays SLOWPATH and goes to the one-workgroup kernel behind.
        if (!p->wave_only) {   // (SQOA_HIP_JOB_WAVEONLY: the caller has parsed the headers -- SQOA colour streams)
            uint32_t any = 1;
            p->dp.any_qoi = 1;
            if (sqoa_launch_dp_classify(p->d_jobs, p->d_ids, n_multi_dec, p->d_word, p->stream) == hipSuccess &&
                hipMemcpyAsync(&any, p->d_word, sizeof any, hipMemcpyDeviceToHost, p->stream) == hipSuccess &&
                hipStreamSynchronize(p->stream) == hipSuccess)
                p->dp.any_qoi = any != 0;
            else (void)hipGetLastError();
        }
        // About 4096 workgroups a pass (four rounds on the chip) -- but ONE SQOA stream alone: about 1024, one round (one 3840 x 2160
        // stream 0.103 -> 0.091 ms: every workgroup of a pass pays a prologue and the ramp of its chains; the same number of chunks
        // as 8 x 1 Mpixel the same way: 0.103 -> 0.106 ms, hence one stream only).  Not where the speculative pass will read the
        // rows: its warm-up is one row, thrown away (0.26 -> 0.49 ms for that stream in QOI-compat).
        // 768-byte chunks (12 stream bytes per lane) where no QOI-compat stream is among them -- the speculative pass reads the rows with
        // 512-byte chunk numbers -- and the streams go to four channels (to three the wide shape loses: sqoa_launch_decode_wave);
        // SQOA_HIP_DEC_WIDE=0 / 1 forces it where it may be forced
        {
            bool all4 = true;
            for (int k = 0; k < n_multi_dec; ++k) if (jobs[ids_a[k]].desc.channels != 4) all4 = false;
            int wide = all4;
            if (const char* kw = getenv("SQOA_HIP_DEC_WIDE")) wide = kw[0] == '1' ? 1 : kw[0] == '0' ? 0 : wide;
            p->dp.lb = (wide && !p->dp.any_qoi) ? 12 : 8;
        }
        unsigned long long total_chunks = 0;
        for (int k = 0; k < n_multi_dec; ++k) total_chunks += sqoa_dp_chunks(jobs[ids_a[k]].src_size, p->dp.lb);
        const uint32_t per = sqoa_dp_per_wg(total_chunks, !p->dp.any_qoi && n_multi_dec == 1 ? 1024u : 4096u, &dp_row_bytes);
        p->dp.per = per;
        for (int k = 0; k < n_multi_dec; ++k) {
            const int i = ids_a[k];
            const uint32_t nc = sqoa_dp_chunks(jobs[i].src_size, p->dp.lb), nw = (nc + per - 1) / per;
            if (dp_wgs.size() + nw > 0x7fffffffull) break;   // (the rest of the prefix gets a workgroup per stream: the launch splits the id list at dp.n_images)
            const uint32_t img = (uint32_t)dp_images.size();
            dp_images.push_back({i, (uint32_t)dp_rows, nc, nw});
            for (uint32_t c0 = 0; c0 < nc; c0 += per) dp_wgs.push_back({img, c0});
            dp_rows += nw;
        }
    }
    p->dp.n_images = (int)dp_images.size();
    p->dp.n_wgs = (int)dp_wgs.size();
    auto grow = [&](void** d, size_t* cap, size_t rows, size_t row_bytes) -> hipError_t {
        if (rows <= *cap) return hipSuccess;
        if (*d) { hipError_t e2 = hipFree(*d); *d = nullptr; *cap = 0; if (e2 != hipSuccess) return e2; }
        hipError_t e2 = hipMalloc(d, rows * row_bytes);
        if (e2 == hipSuccess) *cap = rows;
        return e2;
    };
    p->st.n_images = (int)st_images.size();
    p->st.n_strips = (int)st_strips.size();
    p->st.n_images_qoi = (int)(st_images.size() - st_images_sqoa);
    p->st.n_strips_qoi = (int)(st_strips.size() - st_strips_sqoa);
    p->n_cut_sqoa = n_cut_a;
    p->n_cut_qoi = n_cut_b;
    if (!st_images.empty()) {
        const size_t ni = st_images.size(), ns = st_strips.size();
        e = grow(&p->st.d_images, &p->st_images_cap, ni, sizeof(st_image_row));
        if (e == hipSuccess && ns > p->st_strips_cap) {   // the per-strip tables share one capacity
            void** tabs[8] = {&p->st.d_strips, &p->st.d_cuts, &p->st.d_jobs, &p->st.d_ids, &p->st.d_results, &p->st.d_offsets, &p->st.d_qtab, &p->st.d_qtin};
            const size_t row[8] = {sizeof(st_strip_row), sizeof(uint32_t), sizeof(sqoa_hip_job), sizeof(int), sizeof(sqoa_hip_result), sizeof(unsigned long long),
                                   64 * sizeof(unsigned long long), 64 * sizeof(uint32_t)};
            p->st_strips_cap = 0;
            for (int t = 0; t < 8 && e == hipSuccess; ++t) {
                if (*tabs[t]) { hipFree(*tabs[t]); *tabs[t] = nullptr; }
                e = hipMalloc(tabs[t], (ns + 1) * row[t]);
            }
            if (e == hipSuccess) p->st_strips_cap = ns;
        }

        if (e == hipSuccess) e = grow(&p->st.d_ws, &p->st.ws_bytes, st_ws, 1);
        if (e != hipSuccess) {
            // no memory for the strips' workspace (5 bytes per pixel of the images cut): not an error -- the images stay where they
            // are in the id list, its largest entries, and get a workgroup each like the rest
            (void)hipGetLastError();
            p->st.n_images = p->st.n_strips = p->st.n_images_qoi = p->st.n_strips_qoi = 0;
            p->n_cut_sqoa = p->n_cut_qoi = 0;
            st_images.clear();
        }
        if (st_images.empty()) goto strips_done;
        for (auto& r : st_images) r.ws += (unsigned long long)(uintptr_t)p->st.d_ws;
        if ((e = hipMemcpy(p->st.d_images, st_images.data(), sizeof(st_image_row) * ni, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->st.d_strips, st_strips.data(), sizeof(st_strip_row) * ns, hipMemcpyHostToDevice)) != hipSuccess) {
            p->st.n_images = p->st.n_strips = p->st.n_images_qoi = p->st.n_strips_qoi = 0;
            p->n_cut_sqoa = p->n_cut_qoi = 0;
            return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan: tables of the strip encode, upload", e);
        }
    }
strips_done:
    if (!dp_images.empty()) {
        if ((e = grow(&p->dp.d_flags, &p->dp_flags_cap, dp_images.size(), sizeof(uint32_t))) != hipSuccess ||
            (e = grow(&p->dp.d_images, &p->dp_images_cap, dp_images.size(), sizeof(dp_image_row))) != hipSuccess ||
            (e = grow(&p->dp.d_wgs, &p->dp_wgs_cap, dp_wgs.size(), sizeof(dp_wg_row))) != hipSuccess ||
            (e = grow(&p->dp.d_tab, &p->dp_tab_cap, dp_rows, dp_row_bytes)) != hipSuccess ||
            (e = grow(&p->dp.d_qstate, &p->dp_qstate_cap, dp_rows, sqoa_dq_state_bytes())) != hipSuccess) {
            // (no memory for the tables: not an error -- the streams are decoded one workgroup each like the rest)
            (void)hipGetLastError();
            p->dp.n_images = p->dp.n_wgs = 0;
            return 0;
        }
        if ((e = hipMemcpy(p->dp.d_images, dp_images.data(), sizeof(dp_image_row) * dp_images.size(), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->dp.d_wgs, dp_wgs.data(), sizeof(dp_wg_row) * dp_wgs.size(), hipMemcpyHostToDevice)) != hipSuccess) {
            p->dp.n_images = p->dp.n_wgs = 0;
            return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan: tables of the multi-pass decode, upload", e);
        }
    }
    return 0;
}

int sqoa_hip_plan_create(int device, int kind, const sqoa_hip_job* jobs, int n, sqoa_hip_plan** out) {
    if (!out || !jobs || n < 0 || (kind != SQOA_HIP_ENCODE && kind != SQOA_HIP_DECODE))
        return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(SQOA_HIP_ENOGPU, "sqoa_hip_plan_create: no usable HIP device (libsqoa_mi355x has no CPU fallback)");
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);

    sqoa_hip_plan* p = new sqoa_hip_plan;
    p->device = device;
    p->kind = kind;
    p->cap = n;
    const size_t jb = sizeof(sqoa_hip_job) * (size_t)(n ? n : 1), rb = sizeof(sqoa_hip_result) * (size_t)(n ? n : 1);
    hipError_t e;
    if ((e = hipMalloc((void**)&p->d_jobs, jb)) != hipSuccess || (e = hipMalloc((void**)&p->d_results, rb)) != hipSuccess ||
        (e = hipMalloc((void**)&p->d_ids, sizeof(int) * (size_t)(n ? n : 1))) != hipSuccess ||
        (e = hipMalloc((void**)&p->d_word, 256)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&p->ev0)) != hipSuccess ||
        (e = hipEventCreate(&p->ev1)) != hipSuccess || (e = hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming)) != hipSuccess ||
        (kind == SQOA_HIP_DECODE && ((e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking)) != hipSuccess ||
                                     (e = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming)) != hipSuccess))) {
        sqoa_hip_plan_destroy(p);
        return fail(SQOA_HIP_ENOMEM, "sqoa_hip_plan_create: allocation", e);
    }
    int rc = plan_fill(p, jobs, n);
    if (rc) { sqoa_hip_plan_destroy(p); return rc; }
    *out = p;
    return 0;
}

int sqoa_hip_plan_update(sqoa_hip_plan* p, const sqoa_hip_job* jobs, int n) {
    if (!p || !jobs || n < 0 || n > p->cap) return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan_update: bad arguments");
    HIP_TRY(hipSetDevice(p->device), SQOA_HIP_ENOGPU);
    HIP_TRY(plan_wait(p), SQOA_HIP_ELAUNCH);  // the job table may still be read by a launch on the caller's stream
    return plan_fill(p, jobs, n);
}

void sqoa_hip_plan_destroy(sqoa_hip_plan* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) plan_wait(p);
    if (p->ev_done) hipEventDestroy(p->ev_done);
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    if (p->ev_join) hipEventDestroy(p->ev_join);
    if (p->side) hipStreamDestroy(p->side);
    if (p->d_jobs) hipFree(p->d_jobs);
    if (p->d_results) hipFree(p->d_results);
    if (p->d_word) hipFree(p->d_word);
    if (p->d_ids) hipFree(p->d_ids);
    for (void* d : {p->st.d_images, p->st.d_strips, p->st.d_cuts, p->st.d_jobs, p->st.d_ids, p->st.d_results, p->st.d_offsets, p->st.d_ws, p->st.d_qtab, p->st.d_qtin})
        if (d) hipFree(d);
    if (p->dp.d_images) hipFree(p->dp.d_images);
    if (p->dp.d_wgs) hipFree(p->dp.d_wgs);
    if (p->dp.d_tab) hipFree(p->dp.d_tab);
    if (p->dp.d_flags) hipFree(p->dp.d_flags);
    if (p->dp.d_qstate) hipFree(p->dp.d_qstate);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}

static int plan_enqueue(sqoa_hip_plan* p, hipStream_t s) {
    if (p->n == 0) return 0;
    HIP_TRY(hipSetDevice(p->device), SQOA_HIP_ENOGPU);  // the calling thread's current device may be another one
    // (no copy of the initial result rows in front of the kernels: the rows of refused jobs were written when the plan was filled
    // and nothing touches them, every other row is reset by the first kernel that takes its job -- encode_kernel, the first
    // workgroup of strip_pack_kernel, decode_wave_kernel in its one-workgroup shape or as the first workgroup of pass 1)
    if (p->kind == SQOA_HIP_ENCODE) {
        HIP_TRY(sqoa_launch_encode_strips(p->d_jobs, p->d_results, &p->st, s), SQOA_HIP_ELAUNCH);
        // each id list: [images cut into strips (above) | one workgroup each | many small ones, one wave each]
        const int* ida = p->d_ids; const int* idb = p->d_ids + p->n_sqoa;
        HIP_TRY(sqoa_launch_encode(p->d_jobs, ida + p->n_cut_sqoa, p->n_sqoa - p->n_cut_sqoa - p->n_sqoa_small, 0, p->d_results, p->cus, s), SQOA_HIP_ELAUNCH);
        HIP_TRY(sqoa_launch_encode_small(p->d_jobs, ida + p->n_sqoa - p->n_sqoa_small, p->n_sqoa_small, 0, p->d_results, s), SQOA_HIP_ELAUNCH);
        HIP_TRY(sqoa_launch_encode(p->d_jobs, idb + p->n_cut_qoi, p->n_qoi - p->n_cut_qoi - p->n_qoi_small, 1, p->d_results, p->cus, s), SQOA_HIP_ELAUNCH);
        HIP_TRY(sqoa_launch_encode_small(p->d_jobs, idb + p->n_qoi - p->n_qoi_small, p->n_qoi_small, 1, p->d_results, s), SQOA_HIP_ELAUNCH);
    } else {
        // The part of the streams that is launched apart (a list of mixed output formats: plan_fill) goes to the side stream, forked off in
        // front of everything and joined behind the wave decoders: its few hundred workgroups share the chip with the others' instead of
        // following them (an event fork / join: capturable into a graph like the rest of a launch).
        const bool fork = p->n_chain_b > 0 && p->side;
        if (fork) {
            HIP_TRY(hipEventRecord(p->ev_fork, s), SQOA_HIP_ELAUNCH);
            HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0), SQOA_HIP_ELAUNCH);
            HIP_TRY(sqoa_launch_decode_wave(p->d_jobs, p->d_ids + p->n_sqoa - p->n_chain_b, p->n_chain_b, p->shape_b, p->d_results, p->side), SQOA_HIP_ELAUNCH);
            HIP_TRY(hipEventRecord(p->ev_join, p->side), SQOA_HIP_ELAUNCH);
        }
        // The lazy skip leaves out the kernels BEHIND the wave decoder and nothing else (the speculative QOI-compat pass keeps running: the streams
        // it accepts never mark their rows).  Not while the stream is being captured: a graph outlives the plan's skip state -- replayed over
        // streams that need the stand-by kernels after all it would have no one to run them, and sqoa_hip_plan_results could not know.
        bool skip = p->lazy_skip && !p->wave_only;
        if (skip) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
            if (cap != hipStreamCaptureStatusNone) skip = false;
        }
        HIP_TRY(sqoa_launch_decode(p->d_jobs, p->d_ids, p->n_sqoa, p->shape, p->n_chain_b, p->shape_b, fork ? 1 : 0, fork ? (void*)p->ev_join : nullptr, p->wave_only ? 1 : 0, skip ? 1 : 0, &p->dp,
                                   p->d_results, s), SQOA_HIP_ELAUNCH);
        p->skipped_last = skip;
        p->launched = true;
    }
    return 0;
}

int sqoa_hip_plan_launch(sqoa_hip_plan* p, void* stream) {
    if (!p) return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan_launch: null plan");
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    if (p->n == 0) return 0;
    int rc = plan_enqueue(p, s);
    if (rc) return rc;
    // results / update / destroy wait for THIS launch, not just for the plan's own stream
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    if (cap == hipStreamCaptureStatusNone) {
        HIP_TRY(hipEventRecord(p->ev_done, s), SQOA_HIP_ELAUNCH);
        p->pending = true;
    }
    return 0;
}

void* sqoa_hip_plan_stream(sqoa_hip_plan* p) { return p ? (void*)p->stream : nullptr; }

// Diagnostic: the kernel shape the plan's NEXT launch takes for the streams that get a workgroup each (decode plans; bit 0 many short streams,
// bit 1 twelve stream bytes per lane, bit 2 position-major stores, bit 3 the balanced kernel, bit 4 its three-channel instance; bit 8: the next
// launch leaves the kernels behind the wave decoder out -- the lazy skip; bit 9: a part of the list is launched apart, with a shape of its own), -1 for no plan.
int sqoa_hip_plan_shape(const sqoa_hip_plan* p) { return p ? (p->shape | ((p->lazy_skip && !p->wave_only) ? 0x100 : 0) | (p->n_chain_b > 0 ? 0x200 : 0)) : -1; }

int sqoa_hip_plan_results(sqoa_hip_plan* p, sqoa_hip_result* out, int n) {
    if (!p || !out || n < 0 || n > p->n) return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan_results: bad arguments");
    HIP_TRY(hipSetDevice(p->device), SQOA_HIP_ENOGPU);
    HIP_TRY(plan_wait(p), SQOA_HIP_ELAUNCH);
    if (n) HIP_TRY(hipMemcpy(out, p->d_results, sizeof(sqoa_hip_result) * (size_t)n, hipMemcpyDeviceToHost), SQOA_HIP_ELAUNCH);
    if (p->kind == SQOA_HIP_DECODE && !p->wave_only && p->launched && n) {
        static const bool lazy = [] { const char* k = getenv("SQOA_HIP_DEC_LAZY"); return !(k && k[0] == '0'); }();
        bool slow = false, used = false;
        for (int i = 0; i < n; ++i) {
            slow = slow || out[i].status == SQOA_HIP_ST_SLOWPATH;
            used = used || (reinterpret_cast<const unsigned char*>(&out[i].desc)[11] & 1u) != 0;   // (bit 0: stand-by used; bits 1..7: the content hint)
        }
        if (slow) {
            // a row still SLOWPATH on a plan that is not WAVEONLY: the launch left the kernels behind the wave decoder out (the lazy skip) and a
            // stream needed them after all -- or the rows come from a launch this call cannot see into (a graph captured earlier): now, then
            // the rows once more.  (Behind a launch that did carry them no row is SLOWPATH: the fallback kernel takes whatever is left.)
            HIP_TRY(sqoa_launch_decode_standby(p->d_jobs, p->d_ids, p->n_sqoa, p->small, ((p->shape | (p->n_chain_b > 0 ? p->shape_b : 0)) & 8) ? 1 : 0, p->d_results, p->stream), SQOA_HIP_ELAUNCH);
            HIP_TRY(hipStreamSynchronize(p->stream), SQOA_HIP_ELAUNCH);
            HIP_TRY(hipMemcpy(out, p->d_results, sizeof(sqoa_hip_result) * (size_t)n, hipMemcpyDeviceToHost), SQOA_HIP_ELAUNCH);
            p->lazy_skip = false; p->skipped_last = false;
        } else if (n == p->n && lazy) p->lazy_skip = !used;   // (a full view of the launch: no job took them -- the next launch does without)
        // What the balanced kernel saw of the streams' CONTENT (row_content_hint, sqoa_common.h): where the jobs that make most of the plan's pixels
        // are streams it is the slower kernel on -- 448 op starts or more per 1024-byte chunk: ops of two bytes throughout, photographs, every
        // lane owns exactly eight ops and its LDS accesses fall on a quarter of the banks (class photo to four channels: 3.05 ms against
        // decode_wave_kernel's 2.71) -- or streams with more op starts in a chunk than it has slots (ramps, flat images: they
        // went through the retry), the plan's next launches take the shape the same list gets without it.  Decided ONCE per fill, from a full
        // view of a launch's rows; SQOA_HIP_DEC_CONTENT=0 switches it off, a forced SQOA_HIP_DEC_BAL is not overruled.
        if (n == p->n && (p->shape & 8) && !p->content_switched) {
            const char* kc = getenv("SQOA_HIP_DEC_CONTENT");
            const bool content = !(kc && kc[0] == '0');
            // (the balanced part of the id list: behind the multi-pass prefix, in front of a part that was split off when the plan was filled)
            const size_t c0 = (size_t)p->dp.n_images, c1 = (size_t)(p->n_sqoa - p->n_chain_b);
            auto slow_job = [&](int i) {
                const unsigned hint = reinterpret_cast<const unsigned char*>(&out[i].desc)[11] >> 1;
                // (127: the slot overflow.  Jobs to FOUR channels only: what they move to is decode_wave_kernel at twelve bytes a lane with dword
                // stores; a three-channel job would move to its eight-byte shape with byte stores, and BASELINE configs[4]'s rank shard -- 557
                // three-channel photographs among 7 635 four-channel images -- decodes in 10.28 ms with them in the balanced kernel's
                // three-channel instance and in 10.5 ms with them split off: measured, round 6)
                return hint >= 1u + 448u / 4u && out[i].status == SQOA_HIP_ST_OK && p->h_jobs[i].desc.channels == 4;
            };
            unsigned long long px_said = 0, px_slow = 0;
            size_t n_slow = 0;
            for (size_t k = c0; k < c1 && k < p->h_ids.size(); ++k) {
                const int i = p->h_ids[k];
                const unsigned hint = reinterpret_cast<const unsigned char*>(&out[i].desc)[11] >> 1;
                if (!hint || out[i].status != SQOA_HIP_ST_OK) continue;
                const unsigned long long w = (unsigned long long)out[i].desc.width * out[i].desc.height;
                px_said += w;
                if (slow_job(i)) { px_slow += w; ++n_slow; }
            }
            const char* kb = getenv("SQOA_HIP_DEC_BAL");
            const bool forced = kb && (kb[0] == '0' || kb[0] == '1');   // (as decode_shape_of reads it: set to anything else is not set)
            if (content && !forced && px_said && px_slow) {
                bool whole = 2 * px_slow >= px_said;
                // A list of BOTH kinds -- a suite of screenshots and photographs: BASELINE configs[4] -- is split the way plan_fill splits a list of mixed
                // output formats: the streams the balanced kernel is slow on join (or become) the part that is launched beside it, on the plan's side
                // stream, with the shape THEY get without it.  From a twentieth of the pixels on; below that the second launch is not worth its ramp.
                // (1 024 x 1 Mpixel RGBA, photographs among screenshots, tools/time_mixed_content.py: 128 photographs 2.63 -> 2.56 ms, 256: 2.61 -> 2.46,
                // 400: 2.73 -> 2.47.)
                if (!whole && 20 * px_slow >= px_said && p->side && c1 - c0 - n_slow >= 1) {
                    std::vector<int> ids = p->h_ids, keep, moved;
                    for (size_t k = c0; k < c1; ++k) (slow_job(ids[k]) ? moved : keep).push_back(ids[k]);
                    // (they go BEHIND the ones that stay and IN FRONT of a part that was split off when the plan was filled, which keeps its place:
                    // a graph captured earlier still launches the balanced kernel over [c0, c1) and finds nothing there it cannot take)
                    std::copy(keep.begin(), keep.end(), ids.begin() + (long)c0);
                    std::copy(moved.begin(), moved.end(), ids.begin() + (long)(c0 + keep.size()));
                    moved.insert(moved.end(), ids.begin() + (long)c1, ids.begin() + p->n_sqoa);
                    const int sa = decode_shape_of(p->h_jobs.data(), ids.data(), c0, c0 + keep.size(), p->small, true);
                    if ((sa & 8) && hipMemcpy(p->d_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice) == hipSuccess) {
                        p->h_ids.swap(ids);
                        p->shape = sa;
                        p->shape_nobal = decode_shape_of(p->h_jobs.data(), p->h_ids.data(), c0, c0 + keep.size(), p->small, false);
                        p->shape_b = decode_shape_of(p->h_jobs.data(), p->h_ids.data(), c0 + keep.size(), (size_t)p->n_sqoa, p->small, false);
                        p->n_chain_b = (int)moved.size();
                        p->content_switched = true;
                    } else {
                        (void)hipGetLastError();
                        whole = true;   // (what is left would not take the balanced kernel by itself, or the list could not be uploaded: the whole list, then)
                    }
                }
                if (whole && !p->content_switched) {
                    p->shape = p->shape_nobal;
                    p->content_switched = true;
                }
            }
        }
    }
    // (the stand-by marker lives in the pad byte behind sqoa_desc's three flags: the plan's business, not the caller's)
    for (int i = 0; i < n; ++i) reinterpret_cast<unsigned char*>(&out[i].desc)[11] = 0;
    return 0;
}

int sqoa_hip_plan_time(sqoa_hip_plan* p, int iters, float* ms_total) {
    if (!p || iters <= 0 || !ms_total) return fail(SQOA_HIP_EINVAL, "sqoa_hip_plan_time: bad arguments");
    HIP_TRY(hipSetDevice(p->device), SQOA_HIP_ENOGPU);
    HIP_TRY(plan_wait(p), SQOA_HIP_ELAUNCH);   // a launch on the caller's stream may still be using the result rows and workspaces
    HIP_TRY(hipEventRecord(p->ev0, p->stream), SQOA_HIP_ELAUNCH);
    for (int i = 0; i < iters; ++i) {
        int rc = plan_enqueue(p, p->stream);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(p->ev1, p->stream), SQOA_HIP_ELAUNCH);
    HIP_TRY(hipEventSynchronize(p->ev1), SQOA_HIP_ELAUNCH);
    HIP_TRY(hipEventElapsedTime(ms_total, p->ev0, p->ev1), SQOA_HIP_ELAUNCH);
    return 0;
}

}  // extern "C"

// ---- utility kernels ------------------------------------------------------------------------

namespace sqoa {

SQOA_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// SURVEY.md Appendix C generator; one thread per pixel, grid.y = image.
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ dst, size_t stride, uint32_t w, uint32_t h,
                                                    uint32_t channels, int cls_kind, uint32_t seed, uint32_t img0) {
    const uint32_t img = img0 + blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t x = i % w, y = i / w;
    const uint32_t s0 = mix32(seed * 0x9E3779B9u + img);
    const uint32_t n = mix32(s0 ^ (i * 0x85EBCA6Bu + 0x01234567u));
    const uint32_t m = mix32(s0 ^ ((i >> 1) * 0x85EBCA6Bu + 0x089ABCDEu));
    uint32_t t, cls;
    if (cls_kind == 0) {
        t = mix32(s0 ^ ((y >> 6) * 73856093u) ^ ((x >> 6) * 19349663u));
        cls = t & 7u;
    } else if (cls_kind == 1) {
        t = mix32(s0 ^ ((y >> 5) * 73856093u) ^ ((x >> 5) * 19349663u));
        if ((t & 15u) < 15u) {
            const uint32_t k = (t & 15u) < 8u ? 0u : (t >> 4) & 7u;
            t = mix32(s0 + 0x1000u + k);
            cls = 0;
        } else cls = 4;
    } else { t = s0; cls = 3; }
    const uint32_t br = (t >> 8) & 255u, bg = (t >> 16) & 255u, bb = t >> 24;
    uint32_t r, g, b, a = 255;
    switch (cls) {
    case 0: r = br; g = bg; b = bb; break;
    case 1: r = br + (x >> 1); g = bg + (x >> 1) + (y >> 2); b = bb + (y >> 1); break;
    case 2: r = br + (x >> 1) + (n & 3u) - 2u; g = bg + (x >> 1) + ((n >> 2) & 3u) - 2u; b = bb + (y >> 1) + ((n >> 4) & 3u) - 2u; break;
    case 3: {
        const uint32_t dd = (n >> 8) % 25u;
        r = br + (x >> 1) + dd + ((n >> 16) & 7u) - 16u;
        g = bg + (y >> 1) + dd - 12u;
        b = bb + ((x + y) >> 2) + dd + ((n >> 20) & 7u) - 16u;
        break;
    }
    case 4: { const uint32_t q = mix32(t + (m & 15u)); r = q & 255u; g = (q >> 8) & 255u; b = (q >> 16) & 255u; break; }
    case 5: r = n & 255u; g = (n >> 8) & 255u; b = (n >> 16) & 255u; break;
    case 6: r = n & 255u; g = (n >> 8) & 255u; b = (n >> 16) & 255u; a = n >> 24; break;
    default: r = br; g = bg; b = bb; a = 255u - ((x & 63u) << 1); break;
    }
    uint8_t* o = dst + (size_t)blockIdx.y * stride + (size_t)i * channels;
    if (channels == 4) *reinterpret_cast<uint32_t*>(o) = (r & 255u) | (g & 255u) << 8 | (b & 255u) << 16 | a << 24;
    else { o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b; }
}

// offsets[i] = sum of len[0..i): one workgroup walks the table in tiles of 1024 with a DPP/LDS block scan
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const sqoa_hip_result* __restrict__ res, int n, unsigned long long* __restrict__ offsets) {
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + t;
        const unsigned long long v = i < n && res[i].status == 0 && res[i].len > 0 ? (unsigned long long)res[i].len : 0ull;
        unsigned long long x = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[wv] = x;
        __syncthreads();
        unsigned long long pre = s_carry;
        for (int k = 0; k < wv; ++k) pre += s_wave[k];
        if (i < n) offsets[i] = pre + x - v;
        __syncthreads();
        if (t == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (t == 0) offsets[n] = s_carry;
}

// one workgroup per stream: 16-byte loads from the (256-byte aligned) slot, unaligned 16-byte stores into the blob
__global__ __launch_bounds__(256) void pack_copy_kernel(const sqoa_hip_job* __restrict__ jobs, const sqoa_hip_result* __restrict__ res,
                                                        const unsigned long long* __restrict__ offsets, uint8_t* __restrict__ blob,
                                                        unsigned long long blob_cap) {
    const int i = blockIdx.x;
    const unsigned long long o0 = offsets[i], o1 = offsets[i + 1];
    if (o1 <= o0 || o1 > blob_cap) return;
    const uint32_t len = (uint32_t)(o1 - o0);
    gcptr8 src = (gcptr8)jobs[i].dst;
    gptr8 dst = (gptr8)(blob + o0);
    if ((jobs[i].dst & 15ull) == 0) {
        typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
        typedef __attribute__((address_space(1))) u32x4_unaligned* gptr128u;
        const uint32_t nv = len >> 4;
        for (uint32_t k = threadIdx.x; k < nv; k += blockDim.x) ((gptr128u)dst)[k] = ((gcptr128)src)[k];
        for (uint32_t k = (nv << 4) + threadIdx.x; k < len; k += blockDim.x) dst[k] = src[k];
    } else {
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) dst[k] = src[k];
    }
}

// One lane per buffer: FNV-1a is a serial chain; this is a checking aid, not a hot path.
__global__ void fnv_kernel(const unsigned long long* __restrict__ ptrs, const uint32_t* __restrict__ lens, int n,
                           unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(ptrs[i]);
    const uint32_t len = lens[i];
    unsigned long long h = 0xcbf29ce484222325ull;
    uint32_t k = 0;
    for (; k < len && ((uintptr_t)(p + k) & 15u); ++k) h = (h ^ p[k]) * 0x100000001b3ull;
    for (; k + 16u <= len; k += 16u) {  // one 16-byte load per 16 steps of the chain
        const uint4 q = *reinterpret_cast<const uint4*>(p + k);
        const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) h = (h ^ ((d[j >> 2] >> (8 * (j & 3))) & 0xffu)) * 0x100000001b3ull;
    }
    for (; k < len; ++k) h = (h ^ p[k]) * 0x100000001b3ull;
    out[i] = h;
}

}  // namespace sqoa

extern "C" hipError_t sqoa_launch_pack_offsets(const sqoa_hip_result* d_res, int n, unsigned long long* d_offsets, hipStream_t stream) {
    hipLaunchKernelGGL(sqoa::pack_offsets_kernel, dim3(1), dim3(1024), 0, stream, d_res, n, d_offsets);
    return hipGetLastError();
}

extern "C" {

static int run_batch(int device, int kind, const sqoa_hip_job* jobs, int n, sqoa_hip_result* results) {
    if (n < 0 || (n > 0 && (!jobs || !results))) return fail(SQOA_HIP_EINVAL, "sqoa_hip_*_batch: bad arguments");
    if (n == 0) return 0;
    sqoa_hip_plan* p = nullptr;
    int rc = sqoa_hip_plan_create(device, kind, jobs, n, &p);
    if (rc) return rc;
    rc = sqoa_hip_plan_launch(p, nullptr);
    if (!rc) rc = sqoa_hip_plan_results(p, results, n);
    sqoa_hip_plan_destroy(p);
    return rc;
}
int sqoa_hip_encode_batch(int device, const sqoa_hip_job* jobs, int n, sqoa_hip_result* results) { return run_batch(device, SQOA_HIP_ENCODE, jobs, n, results); }
int sqoa_hip_decode_batch(int device, const sqoa_hip_job* jobs, int n, sqoa_hip_result* results) { return run_batch(device, SQOA_HIP_DECODE, jobs, n, results); }

int sqoa_hip_synth(int device, unsigned long long dst, size_t stride, unsigned w, unsigned h, unsigned channels,
                   int cls, unsigned seed, unsigned img0, int n) {
    if (!dst || !w || !h || (channels != 3 && channels != 4) || cls < 0 || cls > 2 || n < 0 || (size_t)w * h > 0xffffffffu)
        return fail(SQOA_HIP_EINVAL, "sqoa_hip_synth: bad arguments");
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    const unsigned blocks = (unsigned)(((size_t)w * h + 255) / 256);
    for (int done = 0; done < n; done += 32768) {
        const int m = n - done < 32768 ? n - done : 32768;
        hipLaunchKernelGGL(sqoa::synth_kernel, dim3(blocks, m), dim3(256), 0, 0, (uint8_t*)(uintptr_t)dst + (size_t)done * stride, stride, w, h,
                           channels, cls, seed, img0 + done);
    }
    HIP_TRY(hipGetLastError(), SQOA_HIP_ELAUNCH);
    HIP_TRY(hipDeviceSynchronize(), SQOA_HIP_ELAUNCH);
    return 0;
}

int sqoa_hip_pack(sqoa_hip_plan* p, unsigned long long d_blob, size_t blob_cap, unsigned long long d_offsets,
                  unsigned long long* total, void* stream) {
    if (!p || p->kind != SQOA_HIP_ENCODE || !d_blob || !d_offsets) return fail(SQOA_HIP_EINVAL, "sqoa_hip_pack: bad arguments");
    HIP_TRY(hipSetDevice(p->device), SQOA_HIP_ENOGPU);
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    unsigned long long* off = (unsigned long long*)(uintptr_t)d_offsets;
    hipLaunchKernelGGL(sqoa::pack_offsets_kernel, dim3(1), dim3(1024), 0, s, p->d_results, p->n, off);
    if (p->n) hipLaunchKernelGGL(sqoa::pack_copy_kernel, dim3(p->n), dim3(256), 0, s, p->d_jobs, p->d_results, off, (uint8_t*)(uintptr_t)d_blob,
                                 (unsigned long long)blob_cap);
    HIP_TRY(hipGetLastError(), SQOA_HIP_ELAUNCH);
    if (total) {
        HIP_TRY(hipMemcpyAsync(total, off + p->n, sizeof(unsigned long long), hipMemcpyDeviceToHost, s), SQOA_HIP_ELAUNCH);
        HIP_TRY(hipStreamSynchronize(s), SQOA_HIP_ELAUNCH);
        if (*total > blob_cap) return fail(SQOA_HIP_ENOMEM, "sqoa_hip_pack: blob too small");
    }
    return 0;
}

int sqoa_hip_fnv1a64(int device, const unsigned long long* dptrs, const unsigned int* lens, int n, unsigned long long* digests) {
    if (!dptrs || !lens || !digests || n < 0) return fail(SQOA_HIP_EINVAL, "sqoa_hip_fnv1a64: bad arguments");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device), SQOA_HIP_ENOGPU);
    // one allocation: [n pointers | n digests | n lengths]
    unsigned long long* d_p = nullptr;
    HIP_TRY(hipMalloc((void**)&d_p, 20 * (size_t)n), SQOA_HIP_ENOMEM);
    unsigned long long* d_o = d_p + n;
    uint32_t* d_l = (uint32_t*)(d_o + n);
    hipError_t e = hipMemcpy(d_p, dptrs, 8 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_l, lens, 4 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sqoa::fnv_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_p, d_l, n, d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(digests, d_o, 8 * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(d_p);
    HIP_TRY(e, SQOA_HIP_ELAUNCH);
    return 0;
}

}  // extern "C"
