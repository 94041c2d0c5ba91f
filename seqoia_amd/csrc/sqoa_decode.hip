// sqoa_decode.hip -- SQOA / QOI batch decoder for gfx950 (MI355X).
//
// Replaces the reference decoder loop (seqoia.h:722-806) and its header handling
// (seqoia.h:672-707).  Two device paths share one job table:
//
//   * decode_serial_kernel: one image per LANE, the full op set exactly as the
//     reference parses it, including the crafted-input cases (SQOA_OP_REF with
//     the SQOA_NEXT cursor quirk seqoia.h:418/:729-738, mono streams with a
//     128-entry index, exhausted streams).  It is the correctness net for
//     whatever the wave-parallel path declines, and the whole path for tiny
//     images, where 64 independent images per wave is the right shape anyway.
//   * decode_wave_kernel (sqoa_decode_wave.hip): the wave-parallel path for
//     colour streams without REF ops.
#include "sqoa_common.h"

namespace sqoa {

struct Hdr {
    uint32_t width, height;
    uint8_t channels, colorspace, qoi;
    bool ok;
};

// Header parse + validation, reference seqoia.h:672-688 (+ :705-707).  `size` >= 22 is
// guaranteed by the plan.  Writes the header fields to `res` the way sqoa_decode fills *desc.
SQOA_DEV Hdr parse_header(gcptr8 b, sqoa_hip_result* res, bool write_res) {
    Hdr h;
    const uint32_t magic = (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3];
    h.width = (uint32_t)b[4] << 24 | (uint32_t)b[5] << 16 | (uint32_t)b[6] << 8 | b[7];
    h.height = (uint32_t)b[8] << 24 | (uint32_t)b[9] << 16 | (uint32_t)b[10] << 8 | b[11];
    h.channels = b[12];
    h.colorspace = b[13];
    h.qoi = b[14] != 0x31;
    if (write_res) {
        res->desc.width = h.width;
        res->desc.height = h.height;
        res->desc.channels = h.channels;
        res->desc.colorspace = h.colorspace;
        res->desc.qoi_compat = h.qoi;
    }
    h.ok = !(h.width == 0 || h.height == 0 || h.channels < 1 || h.channels > 6 || h.colorspace > 1 ||
             !(magic == 0x716f6966u || magic == 0x53716f61u) || (magic == 0x716f6966u && !h.qoi) ||
             h.height >= 400000000u / h.width);
    return h;
}

// Cursor with the reference's SQOA_NEXT rule (seqoia.h:418): sitting on `ref` jumps to refp+1
// WITHOUT advancing; otherwise post-increment.
struct Cursor {
    int p, ref, refp;
    SQOA_DEV int next() {
        if (p == ref) { p = refp + 1; return p; }
        return p++;
    }
};

constexpr int SERIAL_LANES = 64;

// One image per lane.  table layout [slot][lane] so equal slots across lanes never conflict.
__global__ __launch_bounds__(SERIAL_LANES) void decode_serial_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                                     const int* __restrict__ job_ids, int n,
                                                                     sqoa_hip_result* __restrict__ results, int only_slowpath) {
    __shared__ uint32_t s_table[128][SERIAL_LANES];
    const int t = threadIdx.x;
    const int slot = blockIdx.x * SERIAL_LANES + t;
    if (slot >= n) return;
    const int jid = job_ids[slot];
    const sqoa_hip_job job = jobs[jid];
    sqoa_hip_result* res = &results[jid];
    if (only_slowpath) {  // second launch: only what the wave-parallel kernel declined
        if (res->status != SQOA_HIP_ST_SLOWPATH) return;
        res->status = SQOA_HIP_ST_OK;
    }
    gcptr8 bytes = (gcptr8)job.src;
    gptr8 out = (gptr8)job.dst;
    const int size = (int)job.src_size;

    const Hdr h = parse_header(bytes, res, true);
    if (!h.ok) { res->status = SQOA_HIP_ST_BADARG; return; }
    const int col = h.channels < 3 ? 1 : 3;
    const uint32_t index_size = col == 1 ? 128u : 64u;  // seqoia.h:690-697
    int channels = job.desc.channels;
    bool add_alpha = (channels & 1) == 0;
    if (channels == 0) { add_alpha = (h.channels & 1) == 0; channels = col + (add_alpha ? 1 : 0); }  // seqoia.h:699-702
    const bool qoi = h.qoi;
    const uint32_t N = h.width * h.height;
    if ((unsigned long long)N * channels > job.dst_cap) { res->status = SQOA_HIP_ST_OVERFLOW; return; }

    Cursor c{kHeader + (qoi ? 0 : 1), -1, 0};
    for (int i = 0; i < 128; ++i) s_table[i][t] = 0;
    uint32_t r = 0, g = 0, b = 0, a = 255;
    int run = 0;
    const int chunks_len = size - 8;
    // Fast mode: while no SQOA_OP_REF has been met the cursor only moves forward, so one unaligned 8-byte load
    // per op (op byte, <= 4 data bytes, ALPHA peek) replaces up to six dependent byte loads.  The first REF
    // switches to the byte-exact cursor below for the rest of the stream (seqoia.h:418, :729-738).
    bool exact = false;
    for (uint32_t i = 0; i < N; ++i) {
        if (run > 0) run--;
        else if (c.p < chunks_len && !exact) {
            unsigned long long q;
            {
                typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
                typedef const __attribute__((address_space(1))) u64_unaligned* gq;
                gcptr8 a8 = bytes + c.p;
                if (c.p + 8 <= size) q = *(gq)a8;
                else { q = 0; for (int k = 0; k < 8 && c.p + k < size; ++k) q |= (unsigned long long)a8[k] << (8 * k); }
            }
            const int b1 = (int)(q & 0xff);
            if (!qoi && b1 < 0x60) { exact = true; --i; continue; }  // REF: redo this pixel with the exact cursor
            int used = 1;
            if (b1 >= 0xfe) {  // seqoia.h:740-752
                if (col == 3) { r = (q >> 8) & 255u; g = (q >> 16) & 255u; b = (q >> 24) & 255u; used = 4; }
                else { g = (q >> 8) & 255u; used = 2; }
                if (b1 == 0xff) { a = (q >> (8 * used)) & 255u; used++; }
            } else if (qoi && (uint32_t)b1 < index_size) {  // seqoia.h:753-755
                const uint32_t v = s_table[b1][t];
                r = v & 255u; g = (v >> 8) & 255u; b = (v >> 16) & 255u; a = v >> 24;
            } else if (qoi && (b1 & 0xc0) == 0x40) {  // seqoia.h:756-760
                r = (r + ((b1 >> 4) & 3) - 2) & 255u;
                g = (g + ((b1 >> 2) & 3) - 2) & 255u;
                b = (b + (b1 & 3) - 2) & 255u;
            } else if ((b1 & 0xc0) == 0x80) {  // seqoia.h:761-769
                const int vg = (b1 & 0x3f) - 32;
                g = (g + vg) & 255u;
                if (col == 3) {
                    const int b2 = (int)((q >> 8) & 255u);
                    r = (r + vg - 8 + ((b2 >> 4) & 15)) & 255u;
                    b = (b + vg - 8 + (b2 & 15)) & 255u;
                    used = 2;
                }
            } else if (!qoi && b1 == 0xfd) run = 511;  // seqoia.h:770-772
            else run = b1 & 0x3f;                        // seqoia.h:773-775
            if (!qoi && col == 3) {  // ALPHA suffix, seqoia.h:777-783
                const int pk = (int)((q >> (8 * used)) & 255u);
                if (pk >= 0x60 && pk < 0x80) { a = (a + (pk & 31) - 16) & 255u; used++; }
            }
            c.p += used;
            if (qoi) s_table[(r * 3 + g * 5 + b * 7 + a * 11) % index_size][t] = r | g << 8 | b << 16 | a << 24;  // seqoia.h:785-787
        }
        else if (c.p < chunks_len) {
            int b1 = bytes[c.next()];
            if (!qoi && b1 < 0x60) {  // SQOA_OP_REF, seqoia.h:729-738
                c.refp = c.p;
                c.ref = c.p - (b1 & 31);
                c.p = c.ref - 2 - (b1 >> 5);
                if (c.p < 0) { res->status = SQOA_HIP_ST_BADARG; return; }
                b1 = bytes[c.p++];
            }
            if (b1 >= 0xfe) {  // seqoia.h:740-752
                if (col == 3) { r = bytes[c.next()]; g = bytes[c.next()]; b = bytes[c.next()]; }
                else g = bytes[c.next()];
                if (b1 == 0xff) a = bytes[c.next()];
            } else if (qoi && (uint32_t)b1 < index_size) {  // seqoia.h:753-755
                const uint32_t v = s_table[b1][t];
                r = v & 255u; g = (v >> 8) & 255u; b = (v >> 16) & 255u; a = v >> 24;
            } else if (qoi && (b1 & 0xc0) == 0x40) {  // seqoia.h:756-760
                r = (r + ((b1 >> 4) & 3) - 2) & 255u;
                g = (g + ((b1 >> 2) & 3) - 2) & 255u;
                b = (b + (b1 & 3) - 2) & 255u;
            } else if ((b1 & 0xc0) == 0x80) {  // seqoia.h:761-769
                const int vg = (b1 & 0x3f) - 32;
                g = (g + vg) & 255u;
                if (col == 3) {
                    const int b2 = bytes[c.next()];
                    r = (r + vg - 8 + ((b2 >> 4) & 15)) & 255u;
                    b = (b + vg - 8 + (b2 & 15)) & 255u;
                }
            } else if (!qoi && b1 == 0xfd) run = 511;  // seqoia.h:770-772
            else run = b1 & 0x3f;                        // seqoia.h:773-775
            if (!qoi && col == 3) {  // ALPHA suffix, seqoia.h:777-783
                const int pk = bytes[c.p];
                if (pk >= 0x60 && pk < 0x80) {
                    const int s = bytes[c.next()];
                    a = (a + (s & 31) - 16) & 255u;
                }
            }
            if (qoi) s_table[(r * 3 + g * 5 + b * 7 + a * 11) % index_size][t] = r | g << 8 | b << 16 | a << 24;  // seqoia.h:785-787
        }
        gptr8 o = out + (size_t)i * channels;  // seqoia.h:790-805
        if (channels == 4 && (job.dst & 3ull) == 0) {
            const uint32_t rr = col == 3 ? r : g, bb = col == 3 ? b : g;
            ((gptr32)out)[i] = rr | g << 8 | bb << 16 | a << 24;
            continue;
        }
        if (channels >= 3 && col == 3) { o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b; }
        else { o[0] = (uint8_t)g; if (channels >= 3) { o[1] = (uint8_t)g; o[2] = (uint8_t)g; } }
        if (add_alpha) o[channels - 1] = (uint8_t)a;
    }
    res->len = (int)(N * channels);
}

}  // namespace sqoa

extern "C" hipError_t sqoa_launch_decode_wave(const sqoa_hip_job* d_jobs, const int* d_ids, int n,
                                              sqoa_hip_result* d_results, hipStream_t stream);

// Wave-parallel kernel first; whatever it marks SQOA_HIP_ST_SLOWPATH goes through the exact serial kernel.
extern "C" hipError_t sqoa_launch_decode(const sqoa_hip_job* d_jobs, const int* d_ids, int n,
                                         sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipError_t e = sqoa_launch_decode_wave(d_jobs, d_ids, n, d_results, stream);
    if (e != hipSuccess) return e;
    const int blocks = (n + sqoa::SERIAL_LANES - 1) / sqoa::SERIAL_LANES;
    hipLaunchKernelGGL(sqoa::decode_serial_kernel, dim3(blocks), dim3(sqoa::SERIAL_LANES), 0, stream, d_jobs, d_ids, n, d_results, 1);
    return hipGetLastError();
}
