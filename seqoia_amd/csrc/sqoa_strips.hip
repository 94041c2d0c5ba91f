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
//                      of the strips in front of it (it adds their lengths up itself), at the image's own dst; the first one writes
//                      header and result row.
//
// QOI-compat images (qoi_compat = 1) carry one more piece of state across a cut: the 64-entry colour index (seqoia.h:563-582).
// It only ever holds the LAST non-run pixel that hashed to each slot (a hit leaves the slot as it is, a miss overwrites it,
// seqoia.h:566-571; run pixels never touch it, seqoia.h:544-550), so a strip's effect on it is "overwrite the slots I wrote"
// whatever it held before -- and that composes in strip order.  Two more small kernels in front of the encoder:
//
//   strip_qtab_kernel  one workgroup per strip: the last non-run pixel per slot inside the strip (ds_max_u64 of position | value);
//   strip_qscan_kernel one workgroup per image: the strips' tables composed in order, the index entering each written out (64 dwords per strip).
//
// encode_kernel<8, true> then starts every strip from its entering index instead of from zeros (seqoia.h:518).
#include "sqoa_common.h"

namespace sqoa {

// (st_image, st_strip: sqoa_common.h)
constexpr uint32_t ST_NONE = 0xffffffffu;

SQOA_DEV uint32_t st_nominal(uint32_t npx, uint32_t k, uint32_t s) { return (uint32_t)((unsigned long long)npx * k / s); }
SQOA_DEV int st_stride(const sqoa_desc& d) { return (d.channels < 3 ? 1 : 3) + ((d.channels & 1) ? 0 : 1); }   // seqoia.h:486

// The strip jobs of one image from its cuts (one wave, lane = strip row modulo 64).  The cuts are copied to LDS first, all loads in flight
// together: read one by one where they lie, with the search for the next cut that exists, the loop is a chain of L2 round trips.
constexpr uint32_t ST_CUTS_LDS = 2048;   // = SQOA_ST_WGS (sqoa_runtime.hip): no launch holds more strip rows
SQOA_DEV void strip_build_jobs(const sqoa_hip_job& job, const st_image& im, const uint32_t* cuts, sqoa_hip_job* sjobs, int* sids, sqoa_hip_result* sres, int lane) {
    __shared__ uint32_t s_cuts[ST_CUTS_LDS];
    const uint32_t ch = (uint32_t)st_stride(job.desc);
    const bool in_lds = im.nstrips <= ST_CUTS_LDS;
    if (in_lds) {
        for (uint32_t k0 = 0; k0 < im.nstrips; k0 += 512) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) { const uint32_t k = k0 + 64u * u + (uint32_t)lane; v[u] = k < im.nstrips ? cuts[im.first + k] : ST_NONE; }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) { const uint32_t k = k0 + 64u * u + (uint32_t)lane; if (k < im.nstrips) s_cuts[k] = v[u]; }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    auto cut_of = [&](uint32_t k) { return in_lds ? s_cuts[k] : cuts[im.first + k]; };
    for (uint32_t k = lane; k < im.nstrips; k += 64) {
        const uint32_t row = im.first + k, cut = cut_of(k);
        sres[row].len = 0;      // the rows as they must read before the encoder runs (a skipped row stays like this)
        sres[row].status = SQOA_HIP_ST_OK;
        if (cut == ST_NONE) { sids[row] = -1; continue; }
        uint32_t nx = k + 1;
        while (nx < im.nstrips && cut_of(nx) == ST_NONE) ++nx;
        const uint32_t end = nx < im.nstrips ? cut_of(nx) : im.npx;
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

// One wave per nominal cut.  (Round 6 tried to have the wave that finishes an image's LAST cut build the strip jobs on the spot -- a tick per image
// counted with agent-scope acquire / release atomics -- to save strip_job_kernel's launch: byte-exact, ten minutes of soak, and SLOWER, 16 us against
// 4.8 + 5.4 for the two launches: 506 waves releasing through one counter means 506 write-backs of the L2 in a row on a part with eight of them.)
__global__ __launch_bounds__(64) void strip_cut_kernel(const sqoa_hip_job* __restrict__ jobs, const st_image* __restrict__ images,
                                                       const st_strip* __restrict__ strips, uint32_t* __restrict__ cuts) {
    const int lane = lane_id();
    const st_strip me = strips[blockIdx.x];
    const st_image im = images[me.img];
    const sqoa_hip_job job = jobs[im.jid];
    uint32_t cut = 0;
    if (me.k) {
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
    const st_image im = images[blockIdx.x];
    strip_build_jobs(jobs[im.jid], im, cuts, sjobs, sids, sres, lane_id());
}

// QOI-compat strips: per slot of the colour index, the last pixel of the strip that differs from its predecessor and hashes to
// the slot, as (pixel position in the strip + 1) << 32 | value; 0 = the strip does not write the slot.  One workgroup per strip row.
// (Round 6: eight pixels a thread and pass, their loads issued together -- the loop of one pixel a pass was one memory latency per
// 256 pixels, 30 us for the 506 strips of a lone 3840 x 2160 image, a quarter of that image's whole encode.)
__global__ __launch_bounds__(256) void strip_qtab_kernel(const sqoa_hip_job* __restrict__ sjobs, const int* __restrict__ sids, uint32_t row0,
                                                         unsigned long long* __restrict__ qtab) {
    __shared__ unsigned long long s_last[64];
    const uint32_t row = row0 + blockIdx.x;
    if (threadIdx.x < 64) s_last[threadIdx.x] = 0ull;
    __syncthreads();
    if (sids[row] >= 0) {
        const sqoa_hip_job job = sjobs[row];
        const int ch = st_stride(job.desc);
        gcptr8 src = (gcptr8)job.src;
        const uint32_t n = job.desc.width * job.desc.height;
        // the strip's first pixel: its predecessor is the pixel in front of the strip, or (0,0,0,255) at the image's start (seqoia.h:521-525)
        const uint32_t pv0 = (job.reserved & SQOA_HIP_JOB_PREV) ? load_px(src - ch, 0, ch) : kPxInit;
        if (ch == 4 && !(job.src & 3ull)) {
            // four pixels a load (RGBA): a strip starts wherever its cut fell, so the first `head` pixels up to the next 16-byte boundary go one by
            // one, then quads, four loads in flight; the pixel in front of a quad is the last of the quad before -- one more dword, from a line
            // the neighbouring thread fetches; the last one to three pixels one by one again
            constexpr int U4 = 4;
            gcptr32 s32 = (gcptr32)src;
            const uint32_t head0 = (uint32_t)((16u - (uint32_t)(job.src & 15ull)) & 15u) >> 2, head = head0 < n ? head0 : n;
            const uint32_t nq = (n - head) >> 2;
            auto one = [&](uint32_t p) {
                const uint32_t px1 = s32[p], pv1 = p ? s32[p - 1u] : pv0;
                if (px1 != pv1) atomicMax(&s_last[qoi_hash(px1)], (unsigned long long)(p + 1u) << 32 | px1);
            };
            if (threadIdx.x < head) one(threadIdx.x);
            for (uint32_t p = head + (nq << 2) + threadIdx.x; p < n; p += 256u) one(p);
            gcptr128 q128 = (gcptr128)(s32 + head);
            for (uint32_t q0 = threadIdx.x; q0 < nq; q0 += 256u * U4) {
                u32x4 v[U4];
                uint32_t pq[U4];
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const uint32_t qi = q0 + 256u * (uint32_t)u, p = head + 4u * qi;
                    v[u] = qi < nq ? q128[qi] : u32x4{0u, 0u, 0u, 0u};
                    pq[u] = qi < nq ? (p ? s32[p - 1u] : pv0) : 0u;
                }
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const uint32_t qi = q0 + 256u * (uint32_t)u, p = head + 4u * qi;
                    if (qi >= nq) continue;
                    const uint32_t px4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    uint32_t prev = pq[u];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (px4[i] != prev) atomicMax(&s_last[qoi_hash(px4[i])], (unsigned long long)(p + (uint32_t)i + 1u) << 32 | px4[i]);
                        prev = px4[i];
                    }
                }
            }
        } else {
        constexpr int U = 8;
        for (uint32_t p0 = threadIdx.x; p0 < n; p0 += 256u * U) {
            uint32_t px[U], pv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t p = p0 + 256u * (uint32_t)u;
                px[u] = p < n ? load_px(src, p, ch) : 0u;
                pv[u] = p < n ? (p ? load_px(src, p - 1, ch) : pv0) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t p = p0 + 256u * (uint32_t)u;
                if (p < n && px[u] != pv[u]) atomicMax(&s_last[qoi_hash(px[u])], (unsigned long long)(p + 1u) << 32 | px[u]);
            }
        }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) qtab[(size_t)row * 64 + threadIdx.x] = s_last[threadIdx.x];
}

// The colour index entering every strip of an image: the strips' tables composed by overwrite, in strip order, from all zeros
// (seqoia.h:518).  One workgroup of 16 waves per image, lane = slot.  (Round 6: one wave walking all strips, four loads in flight, took 38 us
// for 506 strips -- a load latency per four strips.)  Wave v takes the strips [v S / 16, (v + 1) S / 16): first what its stretch leaves in
// every slot (the last strip of the stretch that writes it), to LDS; then, starting from what the stretches in front leave, the table
// entering each of its strips.  Eight loads in flight both times.
__global__ __launch_bounds__(1024) void strip_qscan_kernel(const st_image* __restrict__ images, uint32_t img0, const unsigned long long* __restrict__ qtab,
                                                           uint32_t* __restrict__ qtin) {
    constexpr uint32_t WV = 16, U = 8;
    __shared__ unsigned long long s_seg[WV][64];   // stamp (1 = the stretch writes the slot) << 32 | value
    const st_image im = images[img0 + blockIdx.x];
    const int lane = lane_id();
    const uint32_t v = (uint32_t)bcast_first(threadIdx.x >> 6);
    const uint32_t k0 = (uint32_t)((unsigned long long)im.nstrips * v / WV), k1 = (uint32_t)((unsigned long long)im.nstrips * (v + 1) / WV);
    const unsigned long long* q = qtab + (size_t)im.first * 64 + lane;
    unsigned long long mine = 0;
    for (uint32_t k = k0; k < k1; k += U) {
        unsigned long long e[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) e[u] = k + u < k1 ? q[(size_t)(k + u) * 64] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) if (e[u] >> 32) mine = 1ull << 32 | (uint32_t)e[u];
    }
    s_seg[v][lane] = mine;
    __syncthreads();
    uint32_t t = 0;
    for (uint32_t w2 = 0; w2 < v; ++w2) { const unsigned long long e = s_seg[w2][lane]; if (e >> 32) t = (uint32_t)e; }
    uint32_t* o = qtin + (size_t)im.first * 64 + lane;
    for (uint32_t k = k0; k < k1; k += U) {
        unsigned long long e[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) e[u] = k + u < k1 ? q[(size_t)(k + u) * 64] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            if (k + u < k1) o[(size_t)(k + u) * 64] = t;
            if (e[u] >> 32) t = (uint32_t)e[u];
        }
    }
}

// Where a strip's stream goes: behind the header and the streams of the image's strips in front of it -- the sum of THEIR lengths, which every
// workgroup adds up for itself from the strips' result rows (round 6: a block reduction over at most 2 048 rows instead of a one-workgroup prefix-sum
// kernel in front: one launch less in a chain of launches that a lone image's encode consists of).
__global__ __launch_bounds__(256) void strip_pack_kernel(const sqoa_hip_job* __restrict__ jobs, const st_image* __restrict__ images,
                                                         const st_strip* __restrict__ strips, const sqoa_hip_job* __restrict__ sjobs,
                                                         const sqoa_hip_result* __restrict__ sres, sqoa_hip_result* __restrict__ results) {
    __shared__ unsigned long long s_part[2][4];
    __shared__ int s_bad;
    const st_strip me = strips[blockIdx.x];
    const st_image im = images[me.img];
    const sqoa_hip_job job = jobs[im.jid];
    const uint32_t row = im.first + me.k;
    gptr8 dst = (gptr8)job.dst;
    const bool qoi = job.desc.qoi_compat != 0;
    const uint32_t kHead = qoi ? kHeader : kHeader + 1;   // header [+ start byte, SQOA] (seqoia.h:497-514)
    // bytes of the strips in front of mine; the image's first strip also adds up all of them (the image's length) and looks at every status
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const uint32_t upto = me.k == 0 ? im.nstrips : me.k;
    unsigned long long part = 0;
    int bad = 0;
    for (uint32_t k = threadIdx.x; k < upto; k += blockDim.x) {
        const sqoa_hip_result r = sres[im.first + k];
        if (r.status == SQOA_HIP_ST_OK && r.len > 0) part += (unsigned long long)r.len;
        if (r.status != SQOA_HIP_ST_OK) bad = 1;
    }
    for (int d = 32; d; d >>= 1) part += __shfl_down(part, d);
    if ((threadIdx.x & 63) == 0) s_part[0][threadIdx.x >> 6] = part;
    if (bad) s_bad = 1;
    __syncthreads();
    const unsigned long long sum = s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3];
    if (me.k == 0) {
        if (threadIdx.x < kHead) {
            const uint32_t t = threadIdx.x, word = t < 4 ? (qoi ? 0x716f6966u : 0x53716f61u) : t < 8 ? job.desc.width : job.desc.height;
            dst[t] = t < 12 ? (uint8_t)(word >> (24 - 8 * (t & 3))) : t == 12 ? (uint8_t)st_stride(job.desc) : t == 13 ? job.desc.colorspace : (uint8_t)0x31;
        }
        if (threadIdx.x == 0) {
            const unsigned long long total = kHead + sum;
            results[im.jid].len = s_bad || total > job.dst_cap ? 0 : (int)total;
            results[im.jid].status = s_bad ? SQOA_HIP_ST_INTERNAL : total > job.dst_cap ? SQOA_HIP_ST_OVERFLOW : SQOA_HIP_ST_OK;
        }
    }
    const sqoa_hip_result mine = sres[row];
    const unsigned long long o0 = me.k == 0 ? 0ull : sum;
    const unsigned long long o1 = o0 + (mine.status == SQOA_HIP_ST_OK && mine.len > 0 ? (unsigned long long)mine.len : 0ull);
    if (o1 <= o0) return;   // a skipped row
    const unsigned long long at = kHead + o0;
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

extern "C" hipError_t sqoa_launch_encode_tab(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi, sqoa_hip_result* d_results,
                                             const uint32_t* d_qtab_in, int cus, hipStream_t stream);
extern "C" hipError_t sqoa_launch_pack_offsets(const sqoa_hip_result* d_res, int n, unsigned long long* d_offsets, hipStream_t stream);

// The images of the tables: SQOA-native ones first (strip rows [0, n_strips - n_strips_qoi)), QOI-compat ones behind them.
extern "C" hipError_t sqoa_launch_encode_strips(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const sqoa_st_tables* st, hipStream_t stream) {
    using namespace sqoa;
    if (!st || st->n_images <= 0 || st->n_strips <= 0) return hipSuccess;
    const st_image* im = (const st_image*)st->d_images;
    const st_strip* ss = (const st_strip*)st->d_strips;
    sqoa_hip_job* sjobs = (sqoa_hip_job*)st->d_jobs;
    sqoa_hip_result* sres = (sqoa_hip_result*)st->d_results;
    const int* sids = (const int*)st->d_ids;
    const int ns_q = st->n_strips_qoi, ns_s = st->n_strips - ns_q, ni_q = st->n_images_qoi, ni_s = st->n_images - ni_q;
    hipLaunchKernelGGL(strip_cut_kernel, dim3(st->n_strips), dim3(64), 0, stream, d_jobs, im, ss, (uint32_t*)st->d_cuts);
    hipLaunchKernelGGL(strip_job_kernel, dim3(st->n_images), dim3(64), 0, stream, d_jobs, im, (const uint32_t*)st->d_cuts, sjobs, (int*)st->d_ids, sres);
    hipError_t e = sqoa_launch_encode_tab(sjobs, sids, ns_s, 0, sres, nullptr, 0, stream);
    if (e != hipSuccess) return e;
    if (ns_q > 0) {
        hipLaunchKernelGGL(strip_qtab_kernel, dim3(ns_q), dim3(256), 0, stream, sjobs, sids, (uint32_t)ns_s, (unsigned long long*)st->d_qtab);
        hipLaunchKernelGGL(strip_qscan_kernel, dim3(ni_q), dim3(1024), 0, stream, im, (uint32_t)ni_s, (const unsigned long long*)st->d_qtab, (uint32_t*)st->d_qtin);
        // (the kernel indexes the entering tables by job id = strip row)
        e = sqoa_launch_encode_tab(sjobs, sids + ns_s, ns_q, 1, sres, (const uint32_t*)st->d_qtin, 0, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(strip_pack_kernel, dim3(st->n_strips), dim3(256), 0, stream, d_jobs, im, ss, sjobs, sres, d_results);
    return hipGetLastError();
}
