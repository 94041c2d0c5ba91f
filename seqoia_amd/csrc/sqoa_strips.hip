// sqoa_strips.hip -- few large images through the encoder with the whole chip (gfx950).
//
// One image is one workgroup, i.e. one compute unit's instruction issue: 0.575 ms per Mpixel however idle the other 255 CUs
// are (a lone 3840x2160 image: 4.5 ms).  The encoder already takes STRIP jobs (sqoa_hip.h: a stretch of an image's pixels,
// no header, no end marker, the pixel in front as predecessor); a strip must start at a pixel that differs from its
// predecessor so that no run (seqoia.h:544-561) crosses the cut.  A plan's pixels are in device memory, so the cuts are found
// there, by three small kernels around the unchanged encoder (the legacy sqoa_encode is a plan of one job and gets the same):
//
//   strip_cut_kernel   one wave per nominal cut: the first pixel at or behind it that differs from its predecessor, searched
//                      up to the next nominal cut (none: the stretch is the inside of one run and joins the strip in front);
//   strip_job_kernel   one wave per image: the cuts that exist become strip jobs (source stretch, flags, a slot of the plan's
//                      workspace), the others a skipped row (id -1);
//   encode_kernel      over the strip jobs;
//   strip_pack_kernel  one workgroup per strip: its stream goes behind the image's header (seqoia.h:497-514) and the streams
//                      of the strips in front of it, at the image's own dst; the first one writes header and result row.
#include "sqoa_common.h"

namespace sqoa {

// (st_image, st_strip: sqoa_common.h)
constexpr uint32_t ST_NONE = 0xffffffffu;

SQOA_DEV uint32_t st_nominal(uint32_t npx, uint32_t k, uint32_t s) { return (uint32_t)((unsigned long long)npx * k / s); }
SQOA_DEV int st_stride(const sqoa_desc& d) { return (d.channels < 3 ? 1 : 3) + ((d.channels & 1) ? 0 : 1); }   // seqoia.h:486

__global__ __launch_bounds__(64) void strip_cut_kernel(const sqoa_hip_job* __restrict__ jobs, const st_image* __restrict__ images,
                                                       const st_strip* __restrict__ strips, uint32_t* __restrict__ cuts) {
    const int lane = lane_id();
    const st_strip me = strips[blockIdx.x];
    const st_image im = images[me.img];
    uint32_t cut = 0;
    if (me.k) {
        const sqoa_hip_job job = jobs[im.jid];
        const uint32_t ch = (uint32_t)st_stride(job.desc);
        gcptr8 src = (gcptr8)job.src;
        const uint32_t lo = st_nominal(im.npx, me.k, im.nstrips), hi = st_nominal(im.npx, me.k + 1, im.nstrips);
        cut = ST_NONE;
        for (uint32_t base = lo ? lo : 1u; base < hi; base += 64) {   // (pixel 0 has no predecessor in the image)
            const uint32_t p = base + lane;
            bool differs = false;
            if (p < hi) {
                gcptr8 a = src + (size_t)p * ch;
                for (uint32_t b = 0; b < ch; ++b) differs |= a[b] != a[(ptrdiff_t)b - (ptrdiff_t)ch];
            }
            const unsigned long long m = __ballot(differs);
            if (m) { cut = base + (uint32_t)__builtin_ctzll(m); break; }
        }
    }
    if (lane == 0) cuts[im.first + me.k] = cut;
}

__global__ __launch_bounds__(64) void strip_job_kernel(const sqoa_hip_job* __restrict__ jobs, const st_image* __restrict__ images,
                                                       const uint32_t* __restrict__ cuts, sqoa_hip_job* __restrict__ sjobs, int* __restrict__ sids,
                                                       sqoa_hip_result* __restrict__ sres) {
    const int lane = lane_id();
    const st_image im = images[blockIdx.x];
    const sqoa_hip_job job = jobs[im.jid];
    const uint32_t ch = (uint32_t)st_stride(job.desc);
    for (uint32_t k = lane; k < im.nstrips; k += 64) {
        const uint32_t row = im.first + k, cut = cuts[row];
        sres[row].len = 0;      // the rows as they must read before the encoder runs (a skipped row stays like this)
        sres[row].status = SQOA_HIP_ST_OK;
        if (cut == ST_NONE) { sids[row] = -1; continue; }
        uint32_t nx = k + 1;
        while (nx < im.nstrips && cuts[im.first + nx] == ST_NONE) ++nx;
        const uint32_t end = nx < im.nstrips ? cuts[im.first + nx] : im.npx;
        sqoa_hip_job s;
        s.src = job.src + (unsigned long long)cut * ch;
        // slots in the workspace: worst case ch + 1 bytes per pixel (seqoia.h:487-489) behind the strips in front, 256-aligned
        s.dst = im.ws + (((unsigned long long)cut * (ch + 1) + 255ull) & ~255ull) + (unsigned long long)k * 512ull;
        s.src_size = 0;
        s.dst_cap = (end - cut) * (ch + 1) + 256u;
        s.desc = job.desc;
        s.desc.width = end - cut;
        s.desc.height = 1;
        s.reserved = SQOA_HIP_JOB_NOHEADER | (cut ? SQOA_HIP_JOB_PREV : 0u) | (end < im.npx ? SQOA_HIP_JOB_NOEND : 0u);
        sjobs[row] = s;
        sids[row] = (int)row;
    }
}

// `offsets`: exclusive prefix sums of the strip rows' stream lengths (pack_offsets_kernel over the strips' result rows)
__global__ __launch_bounds__(256) void strip_pack_kernel(const sqoa_hip_job* __restrict__ jobs, const st_image* __restrict__ images,
                                                         const st_strip* __restrict__ strips, const sqoa_hip_job* __restrict__ sjobs,
                                                         const sqoa_hip_result* __restrict__ sres, const unsigned long long* __restrict__ offsets,
                                                         sqoa_hip_result* __restrict__ results) {
    const st_strip me = strips[blockIdx.x];
    const st_image im = images[me.img];
    const sqoa_hip_job job = jobs[im.jid];
    const uint32_t row = im.first + me.k;
    gptr8 dst = (gptr8)job.dst;
    constexpr uint32_t kHead = kHeader + 1;   // header + start byte (seqoia.h:497-514)
    if (me.k == 0) {
        __shared__ int s_bad;
        if (threadIdx.x == 0) s_bad = 0;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < im.nstrips; k += blockDim.x)
            if (sres[im.first + k].status != SQOA_HIP_ST_OK) s_bad = 1;
        __syncthreads();
        if (threadIdx.x < kHead) {
            const uint32_t t = threadIdx.x, word = t < 4 ? 0x53716f61u : t < 8 ? job.desc.width : job.desc.height;
            dst[t] = t < 12 ? (uint8_t)(word >> (24 - 8 * (t & 3))) : t == 12 ? (uint8_t)st_stride(job.desc) : t == 13 ? job.desc.colorspace : (uint8_t)0x31;
        }
        if (threadIdx.x == 0) {
            const unsigned long long total = kHead + offsets[im.first + im.nstrips] - offsets[im.first];
            results[im.jid].len = s_bad || total > job.dst_cap ? 0 : (int)total;
            results[im.jid].status = s_bad ? SQOA_HIP_ST_INTERNAL : total > job.dst_cap ? SQOA_HIP_ST_OVERFLOW : SQOA_HIP_ST_OK;
        }
    }
    const unsigned long long o0 = offsets[row], o1 = offsets[row + 1];
    if (o1 <= o0) return;   // a skipped row
    const unsigned long long at = kHead + (o0 - offsets[im.first]);
    const uint32_t len = (uint32_t)(o1 - o0);
    if (at + len > job.dst_cap) return;
    gcptr8 src = (gcptr8)sjobs[row].dst;   // 256-byte aligned
    typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
    typedef __attribute__((address_space(1))) u32x4_unaligned* gptr128u;
    const uint32_t nv = len >> 4;
    for (uint32_t i = threadIdx.x; i < nv; i += blockDim.x) ((gptr128u)(dst + at))[i] = ((gcptr128)src)[i];
    for (uint32_t i = (nv << 4) + threadIdx.x; i < len; i += blockDim.x) dst[at + i] = src[i];
}

}  // namespace sqoa

extern "C" hipError_t sqoa_launch_encode(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi, sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_pack_offsets(const sqoa_hip_result* d_res, int n, unsigned long long* d_offsets, hipStream_t stream);

extern "C" hipError_t sqoa_launch_encode_strips(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const sqoa_st_tables* st, hipStream_t stream) {
    using namespace sqoa;
    if (!st || st->n_images <= 0 || st->n_strips <= 0) return hipSuccess;
    const st_image* im = (const st_image*)st->d_images;
    const st_strip* ss = (const st_strip*)st->d_strips;
    sqoa_hip_job* sjobs = (sqoa_hip_job*)st->d_jobs;
    sqoa_hip_result* sres = (sqoa_hip_result*)st->d_results;
    hipLaunchKernelGGL(strip_cut_kernel, dim3(st->n_strips), dim3(64), 0, stream, d_jobs, im, ss, (uint32_t*)st->d_cuts);
    hipLaunchKernelGGL(strip_job_kernel, dim3(st->n_images), dim3(64), 0, stream, d_jobs, im, (const uint32_t*)st->d_cuts, sjobs, (int*)st->d_ids, sres);
    hipError_t e = sqoa_launch_encode(sjobs, (const int*)st->d_ids, st->n_strips, 0, sres, stream);
    if (e != hipSuccess) return e;
    e = sqoa_launch_pack_offsets(sres, st->n_strips, (unsigned long long*)st->d_offsets, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(strip_pack_kernel, dim3(st->n_strips), dim3(256), 0, stream, d_jobs, im, ss, sjobs, sres, (const unsigned long long*)st->d_offsets, d_results);
    return hipGetLastError();
}


