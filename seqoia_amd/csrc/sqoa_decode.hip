// sqoa_decode.hip -- the decoders behind the wave-parallel SQOA path, for gfx950 (MI355X).
//
// Replaces the reference decoder loop (seqoia.h:722-806) and its header handling (seqoia.h:672-707) for every
// stream sqoa_decode_wave.hip declines (SQOA_HIP_ST_SLOWPATH).  One workgroup of two waves per image:
//
//   * the record pipeline (QOI-compat colour streams; SQOA mono streams; any output channel count).  QOI_OP_INDEX makes every later pixel depend on
//     the VALUES of all earlier ones (the slot is a hash of the value, seqoia.h:785-787), so pixel values cannot be
//     scanned like SQOA's.  Everything else still runs across the wave, 512 stream bytes at a time: cooperative
//     staging in LDS, the parser as a scan of transition functions, every lane decoding the ops that start in its 8
//     bytes into (transform, index slot, pixel count) records laid out in stream order (wave 0).  Wave 1 runs the
//     reference's recurrence over the records, 64 ops at a time: pixel values between two INDEX ops by a segmented
//     scan of the transforms, index updates by per-slot stamps, one serial look-up per INDEX op; then it stores the
//     pixels and expands the runs.  The two waves are coupled by double-buffered record arrays in LDS.  Parallelism
//     across images comes from the batch: one workgroup per image, thousands of workgroups.
//   * decode_exact: the complete op set exactly as the reference parses it, including the crafted-input cases
//     (SQOA_OP_REF with the SQOA_NEXT cursor quirk seqoia.h:418/:729-738, mono streams in QOI framing with their
//     128-entry index, SQOA colour streams to 1/2 channels, exhausted streams), one lane per image.  It is the correctness net for whatever the two
//     wave-parallel decoders decline.
#include <type_traits>

#include "sqoa_decode_parts.h"

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

// ---- exact decoder, one lane ---------------------------------------------------------------------------------
// `table`: 128 dwords of LDS private to the calling lane.
SQOA_DEV void decode_exact(const sqoa_hip_job& job, const Hdr& h, sqoa_hip_result* res, uint32_t* table) {
    gcptr8 bytes = (gcptr8)job.src;
    gptr8 out = (gptr8)job.dst;
    const int size = (int)job.src_size;
    const int col = h.channels < 3 ? 1 : 3;
    const uint32_t index_size = col == 1 ? 128u : 64u;  // seqoia.h:690-697
    int channels = job.desc.channels;
    bool add_alpha = (channels & 1) == 0;
    if (channels == 0) { add_alpha = (h.channels & 1) == 0; channels = col + (add_alpha ? 1 : 0); }  // seqoia.h:699-702
    const bool qoi = h.qoi;
    const uint32_t N = h.width * h.height;
    if ((unsigned long long)N * channels > job.dst_cap) { res->status = SQOA_HIP_ST_OVERFLOW; return; }

    Cursor c{kHeader + (qoi ? 0 : 1), -1, 0};
    for (int i = 0; i < 128; ++i) table[i] = 0;
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
                const uint32_t v = table[b1];
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
            if (qoi) table[(r * 3 + g * 5 + b * 7 + a * 11) % index_size] = r | g << 8 | b << 16 | a << 24;  // seqoia.h:785-787
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
                const uint32_t v = table[b1];
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
            if (qoi) table[(r * 3 + g * 5 + b * 7 + a * 11) % index_size] = r | g << 8 | b << 16 | a << 24;  // seqoia.h:785-787
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

// ---- QOI-compat colour streams, one wave per image -------------------------------------------------------------
constexpr int QD_LB = 8;                          // stream bytes per lane
constexpr int QD_CHUNK = 64 * QD_LB;              // stream bytes per step
constexpr int QD_LDS_DW = QD_CHUNK / 4 + 8;       // staged dwords: chunk + look-ahead + misalignment
constexpr int QD_NLD = (QD_LDS_DW + 63) / 64;

struct QoiLds {
    uint32_t bytes[QD_LDS_DW];
    uint32_t ftab[256];               // byte -> low half of its parser transition function
    // per op, in stream order, double-buffered between the two waves: the transform's value bytes and
    // 1 = INDEX | absolute-channel code (2 = rgb, 4 = rgba) | slot << 3 | pixel count << 9
    uint32_t rec_v[2][QD_CHUNK + 64];
    uint16_t rec_m[2][QD_CHUNK + 64];  // (15 bits: QOI runs are <= 62 pixels)
    uint32_t nrec[2];                 // ops in each buffer
    uint32_t ready[2], freed[2];      // chunk number + 1 that filled / released each buffer
    uint32_t stop;                    // the image is complete: the front wave may leave
    uint32_t stamp[64];               // per index slot: stream-order number + 1 of its last writer
};

// One step of the segmented transform scan: `hf` (all ones / zero) marks lanes whose segment head lies at or below them
// within the distance covered so far; such lanes keep their transform, the others prepend the one arriving from below.
template <int CTRL, int ROWMASK>
SQOA_DEV Xf seg_scan_step(Xf T, uint32_t& hf) {
    Xf P;
    P.v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v, CTRL, ROWMASK, 0xf, false);
    P.abs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.abs, CTRL, ROWMASK, 0xf, false);
    const uint32_t pf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hf, CTRL, ROWMASK, 0xf, false);
    const Xf C = xf_then(P, T);
    const Xf R{(T.v & hf) | (C.v & ~hf), (T.abs & hf) | (C.abs & ~hf)};
    hf |= pf;
    return R;
}

// record meta -> byte mask of the absolute channels
SQOA_DEV uint32_t qd_amask(uint32_t m) { return (m & 2u) ? 0x00ffffffu : (m & 4u) ? 0xffffffffu : 0u; }

constexpr uint32_t QD_SPIN_MAX = 1u << 24;

// Wait until *flag reaches `want` (flags only grow) or the image is declared complete; false = leave.
template <int SLEEP>
SQOA_DEV bool qd_wait(uint32_t* flag, uint32_t want, uint32_t* stop, uint32_t& bad) {
    for (uint32_t spin = 0; spin < QD_SPIN_MAX; ++spin) {
        if (bcast_first(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= want) return true;
        if (bcast_first(__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return false;
        __builtin_amdgcn_s_sleep(SLEEP);
    }
    bad = 1;
    return false;
}

// Front wave: stage, parse, decode ops into stream-order records (everything that does not depend on pixel values).
SQOA_DEV void qoi_front(const sqoa_hip_job& job, bool mono, QoiLds& L, int lane, uint32_t& bad) {
    gcptr8 src = (gcptr8)job.src;
    // (any byte address: offsets are relative to the dword-aligned address at or in front of the stream)
    const uint32_t soff = (uint32_t)(job.src & 3ull);
    src -= soff;
    const uint32_t size = job.src_size + soff;
    const uint32_t p0 = kHeader + (mono ? 1u : 0u) + soff;  // first chunk byte (QOI: right after the header; SQOA: after the start byte, seqoia.h:705)
    const uint32_t chunks_len = size - 8;       // seqoia.h:721
    const uint32_t area = chunks_len > p0 ? chunks_len - p0 : 0;
    const uint32_t nchunks = area ? (area + QD_CHUNK - 1) / QD_CHUNK : 1;

    uint32_t stg[QD_NLD];
    auto load_chunk = [&](uint32_t c, uint32_t (&r)[QD_NLD]) {
        const uint32_t d0 = (p0 + c * QD_CHUNK) >> 2;
#pragma unroll
        for (int k = 0; k < QD_NLD; ++k) {
            const uint32_t d = d0 + lane + 64 * k;
            r[k] = (k < QD_NLD - 1 || lane < QD_LDS_DW - 64 * (QD_NLD - 1)) && d * 4u < size ? ((gcptr32)src)[d] : 0u;
        }
    };
    load_chunk(0, stg);
    uint32_t st_in = ST_A;

    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t buf = c & 1u;
#pragma unroll
        for (int k = 0; k < QD_NLD - 1; ++k) L.bytes[lane + 64 * k] = stg[k];
        if (lane < QD_LDS_DW - 64 * (QD_NLD - 1)) L.bytes[lane + 64 * (QD_NLD - 1)] = stg[QD_NLD - 1];
        if (c + 1 < nchunks) load_chunk(c + 1, stg);
        const uint32_t gs = p0 + c * QD_CHUNK;
        const uint32_t my0 = gs + QD_LB * lane;
        const uint32_t nlive = my0 >= chunks_len ? 0u : (chunks_len - my0 > (uint32_t)QD_LB ? (uint32_t)QD_LB : chunks_len - my0);
        const uint32_t loff = (gs & 3u) + QD_LB * lane;

        // ---- parse: the state in front of every lane's first byte ----
        const unsigned long long own = lds_read8(L.bytes, loff);
        Fn F{ID_LO, ID_HI};
#pragma unroll
        for (int k = 0; k < QD_LB; ++k) {
            const uint32_t b = (uint32_t)(own >> (8 * k)) & 0xffu;
            const uint32_t flo = (uint32_t)k < nlive ? L.ftab[b] : (ST_X | ST_X << 8 | 0x02000000u);
            F = compose(Fn{flo, 0x07060403u}, F);
        }
        const Fn Fi = wave_compose_scan(F);
        const Fn G{bcast_lane(Fi.lo, 63), bcast_lane(Fi.hi, 63)};
        const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
        const uint32_t lst = apply(Fx, st_in);
        st_in = bcast_first(apply(G, st_in));

        // ---- walk: my ops, in order ----
        uint32_t pos = lst >= 2u && lst <= 5u ? lst - 1u : (lst == ST_X ? (uint32_t)QD_LB : 0u);
        uint32_t rv[QD_LB], rm[QD_LB];
        uint32_t nops = 0, iters = QD_LB;
        bool ref = false;
#pragma unroll
        for (int i = 0; i < QD_LB; ++i) {
            const bool act = pos < nlive;
            if (__ballot(act) == 0ull) { iters = i; break; }
            const unsigned long long q = lds_read8(L.bytes, loff + (pos < (uint32_t)QD_LB ? pos : (uint32_t)QD_LB));
            const uint32_t b0 = (uint32_t)q & 0xffu, b1 = ((uint32_t)q >> 8) & 0xffu;
            uint32_t v = 0, m = 0, cnt = 1, len = 1;
            if (mono) {
                // SQOA mono / mono+alpha streams: only g and a exist (r = b = 0 throughout), no ALPHA suffix, no colour index
                if (b0 >= 0xfeu) {                   // seqoia.h:740-752 with col == 1: fe g / ff g a
                    v = (b1 << 8) | (b0 == 0xffu ? (uint32_t)(q >> 16) << 24 : 0u);
                    m = b0 == 0xffu ? 4u : 2u;
                    len = b0 - 0xfcu;
                } else if (b0 < 0x60u) {             // SQOA_OP_REF (seqoia.h:729-738): the exact decoder takes the image
                    if (act) ref = true;
                } else if ((b0 & 0xc0u) == 0x80u) {  // seqoia.h:761-764: one byte
                    v = (((b0 & 0x3fu) - 32u) & 0xffu) << 8;
                } else cnt = b0 == 0xfdu ? 512u : (b0 & 0x3fu) + 1u;  // seqoia.h:770-775
            } else if (b0 >= 0xfeu) {                // seqoia.h:740-752
                v = (uint32_t)(q >> 8) & (b0 == 0xffu ? 0xffffffffu : 0x00ffffffu);
                m = b0 == 0xffu ? 4u : 2u;
                len = b0 - 0xfau;
            } else if (b0 < 0x40u) {                 // seqoia.h:753-755
                m = 1u | b0 << 3;
            } else if (b0 < 0x80u) {                 // seqoia.h:756-760
                v = ((((b0 >> 4) & 3u) - 2u) & 0xffu) | ((((b0 >> 2) & 3u) - 2u) & 0xffu) << 8 | (((b0 & 3u) - 2u) & 0xffu) << 16;
            } else if (b0 < 0xc0u) {                 // seqoia.h:761-769
                const uint32_t vg = (b0 & 0x3fu) - 32u;
                v = ((vg - 8u + (b1 >> 4)) & 0xffu) | (vg & 0xffu) << 8 | ((vg - 8u + (b1 & 15u)) & 0xffu) << 16;
                len = 2;
            } else cnt = (b0 & 0x3fu) + 1u;          // seqoia.h:773-775
            rv[i] = v;
            rm[i] = act ? (m | cnt << (mono ? 3 : 9)) : 0u;  // QOI: slot in bits 3-8, runs <= 62; mono: no slot, runs <= 512
            nops += act ? 1u : 0u;
            pos += act ? len : 0u;
        }

        iters = bcast_first(iters);
        if (__ballot(ref) != 0ull) {  // hand the whole image to decode_exact
            if (lane == 0) __hip_atomic_store(&L.stop, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        // ---- records in stream order, handed to the back wave ----
        const uint32_t oinc = wave_incl_scan(nops);
        const uint32_t T = bcast_lane(oinc, 63);
        const uint32_t obase = oinc - nops;
        if (c >= 2u && !qd_wait<16>(&L.freed[buf], c - 1u, &L.stop, bad)) return;  // (polls rarely: it runs ahead of the back wave)  // chunk c-2 has left this buffer
#pragma unroll
        for (int i = 0; i < QD_LB; ++i) {
            if ((uint32_t)i >= iters) break;
            if (rm[i] >> 3) { L.rec_v[buf][obase + i] = rv[i]; L.rec_m[buf][obase + i] = (uint16_t)rm[i]; }
        }
        if (lane == 0) L.nrec[buf] = T;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&L.ready[buf], c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (bcast_first(__hip_atomic_load(&L.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return;
    }
}

// Back wave: the recurrence of seqoia.h:740-787 over the records, 64 ops (one per lane) at a time, and the pixels.
// bytes a pixel is stored as (seqoia.h:790-805): colour r g b [a] / g [a]; mono g g g [a] / g [a]
SQOA_DEV uint32_t out_px(uint32_t px, bool mono) { return mono ? (((px >> 8) & 0xffu) * 0x010101u) | (px & 0xff000000u) : px; }
SQOA_DEV void put_small(gptr8 out, uint32_t P, int oc, uint32_t px) {  // 1 / 2 output channels: g [a]
    gptr8 o = out + (size_t)P * (unsigned)oc;
    o[0] = (uint8_t)(px >> 8);
    if (oc == 2) o[1] = (uint8_t)(px >> 24);
}

SQOA_DEV void qoi_back(const sqoa_hip_job& job, const Hdr& h, bool mono, sqoa_hip_result* res, QoiLds& L, int lane, uint32_t& bad) {
    gptr8 out = (gptr8)job.dst;
    const uint32_t size = job.src_size;
    const uint32_t N = h.width * h.height;
    const int oc = (int)(job.desc.channels ? job.desc.channels : (mono ? 1u : 3u) + ((h.channels & 1u) ? 0u : 1u));
    const uint32_t p0 = kHeader + (mono ? 1u : 0u), chunks_len = size - 8;
    const uint32_t area = chunks_len > p0 ? chunks_len - p0 : 0;
    const uint32_t nchunks = area ? (area + QD_CHUNK - 1) / QD_CHUNK : 1;
    const bool dw_out = oc == 4 && (job.dst & 3ull) == 0;

    uint32_t tabv = 0;   // the colour index, slot = lane (seqoia.h:712, :785-787)
    uint32_t seen = 0;   // stamp of the writer tabv holds
    uint32_t opno = 0;   // stream-order number of the current group's first op
    uint32_t pos_in = 0;
    uint32_t pxv = kPxInit;  // the current pixel, the same in every lane (seqoia.h:716-719)

    for (uint32_t c = 0; c < nchunks && pos_in < N; ++c) {
        const uint32_t buf = c & 1u;
        if (!qd_wait<1>(&L.ready[buf], c + 1u, &L.stop, bad)) break;
        const uint32_t T = bcast_first(L.nrec[buf]);
        // Between two QOI_OP_INDEX ops pixel values do not depend on the colour index: they are one segmented scan of
        // the ops' transforms (segments start behind every INDEX op) applied to the pixel the segment starts from.
        // "index[hash] = px" for a whole segment is one ds_max of stream-order stamps per op (the last writer of a slot
        // wins, as in the reference's in-order stores) and one fetch by the lane that owns the slot; what stays serial
        // is one look-up per INDEX op.  The index itself is one VGPR, lane = slot.
        for (uint32_t g = 0; g < T; g += 64, opno += 64) {
            const uint32_t ng = T - g < 64u ? T - g : 64u;
            const bool live = (uint32_t)lane < ng;
            const uint32_t m = live ? L.rec_m[buf][g + lane] : 0u;
            const bool isx = m & 1u;
            Xf S{(live && !isx) ? L.rec_v[buf][g + lane] : 0u, isx ? 0u : qd_amask(m)};
            uint32_t hf = shift_up1(isx ? ~0u : 0u, ~0u);  // segment heads: the group's first op and every op behind an INDEX op
            S = seg_scan_step<DPP_ROW_SHR + 1, 0xf>(S, hf);
            S = seg_scan_step<DPP_ROW_SHR + 2, 0xf>(S, hf);
            S = seg_scan_step<DPP_ROW_SHR + 4, 0xf>(S, hf);
            S = seg_scan_step<DPP_ROW_SHR + 8, 0xf>(S, hf);
            S = seg_scan_step<DPP_ROW_BCAST15, 0xa>(S, hf);
            S = seg_scan_step<DPP_ROW_BCAST31, 0xc>(S, hf);
            unsigned long long idxmask = __ballot(isx);
            uint32_t s0 = 0, resv = 0;
            for (;;) {
                const uint32_t j = idxmask ? (uint32_t)__builtin_ctzll(idxmask) : ng;  // next INDEX op, or the end of the group
                const uint32_t pxc = add8(pxv & ~S.abs, S.v);                            // valid in lanes [s0, j)
                const uint32_t hv = qoi_hash(pxc);
                const bool inseg = (uint32_t)lane >= s0 && (uint32_t)lane < j;
                if (inseg) resv = pxc;
                if (j > s0) {
                    if (inseg) atomicMax(&L.stamp[hv], opno + (uint32_t)lane + 1u);
                    const uint32_t w = L.stamp[lane];
                    const uint32_t from = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((w - opno - 1u) << 2), (int)pxc);
                    if (w > seen) { tabv = from; seen = w; }
                    pxv = bcast_lane(pxc, (int)(j - 1u));
                }
                if (j >= ng) break;
                pxv = bcast_lane(tabv, (int)((bcast_lane(m, (int)j) >> 3) & 63u));    // seqoia.h:753-755
                tabv = (uint32_t)lane == qoi_hash(pxv) ? pxv : tabv;
                if ((uint32_t)lane == j) resv = pxv;
                idxmask &= idxmask - 1ull;
                s0 = j + 1u;
                if (s0 >= ng) break;
            }
            // ---- pixels of these 64 ops: literal ops lane-local, runs by the whole wave (seqoia.h:723-725, :790-805) ----
            const uint32_t cnt = mono ? m >> 3 : m >> 9;
            const uint32_t pinc = wave_incl_scan(cnt);
            const uint32_t pxo = out_px(resv, mono);
            const uint32_t P = pos_in + pinc - cnt;
            const bool any_long = __ballot(cnt > 2u) != 0ull;
            if (dw_out) {
                if (cnt >= 1u && cnt <= 2u && P < N) ((gptr32)out)[P] = pxo;
                if (cnt == 2u && P + 1u < N) ((gptr32)out)[P + 1u] = pxo;
            } else if (oc >= 3) {
                for (uint32_t k = 0; k < cnt && k < 2u && cnt <= 2u && P + k < N; ++k) {
                    gptr8 o = out + (size_t)(P + k) * (unsigned)oc;
                    o[0] = (uint8_t)pxo; o[1] = (uint8_t)(pxo >> 8); o[2] = (uint8_t)(pxo >> 16);
                    if (oc == 4) o[3] = (uint8_t)(pxo >> 24);
                }
            } else {
                for (uint32_t k = 0; k < cnt && k < 2u && cnt <= 2u && P + k < N; ++k) put_small(out, P + k, oc, resv);
            }
            if (any_long) {
                unsigned long long runs = __ballot(cnt > 2u && P < N);
                while (runs) {
                    const int l = __builtin_ctzll(runs);
                    runs &= runs - 1ull;
                    const uint32_t rp = bcast_lane(P, l), rvv = bcast_lane(pxo, l), rraw = bcast_lane(resv, l);
                    uint32_t rc = bcast_lane(cnt, l);
                    if (rp + rc > N) rc = N - rp;
                    if (dw_out) fill_pixels<4>(out, rp, rc, rvv, lane);
                    else if (oc == 3) fill_pixels<3>(out, rp, rc, rvv, lane);
                    else if (oc == 4)
                        for (uint32_t k = lane; k < rc; k += 64) {
                            gptr8 o = out + (size_t)(rp + k) * 4;
                            o[0] = (uint8_t)rvv; o[1] = (uint8_t)(rvv >> 8); o[2] = (uint8_t)(rvv >> 16); o[3] = (uint8_t)(rvv >> 24);
                        }
                    else
                        for (uint32_t k = lane; k < rc; k += 64) put_small(out, rp + k, oc, rraw);
                }
            }
            const uint32_t ptot = bcast_lane(pinc, 63);
            pos_in = pos_in + ptot > N ? N : pos_in + ptot;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&L.freed[buf], c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (lane == 0) __hip_atomic_fetch_max(&L.stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // stream exhausted: the last pixel repeats (seqoia.h:726)
    const bool aborted = bcast_first(__hip_atomic_load(&L.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 2u;
    if (pos_in < N && !bad && !aborted) {
        const uint32_t pxo = out_px(pxv, mono);
        if (dw_out) fill_pixels<4>(out, pos_in, N - pos_in, pxo, lane);
        else if (oc == 3) fill_pixels<3>(out, pos_in, N - pos_in, pxo, lane);
        else if (oc == 4)
            for (uint32_t k = pos_in + lane; k < N; k += 64) {
                gptr8 o = out + (size_t)k * 4;
                o[0] = (uint8_t)pxo; o[1] = (uint8_t)(pxo >> 8); o[2] = (uint8_t)(pxo >> 16); o[3] = (uint8_t)(pxo >> 24);
            }
        else
            for (uint32_t k = pos_in + lane; k < N; k += 64) put_small(out, k, oc, pxv);
    }
    if (lane == 0) res->len = (int)(N * (uint32_t)oc);
}

// One image per workgroup of two waves.
__global__ __launch_bounds__(128) void decode_fallback_kernel(const sqoa_hip_job* __restrict__ jobs, const int* __restrict__ job_ids,
                                                             sqoa_hip_result* __restrict__ results, int only_slowpath) {
    __shared__ QoiLds L;
    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    const int jid = job_ids[blockIdx.x];
    sqoa_hip_result* res = &results[jid];
    if (only_slowpath && __builtin_amdgcn_readfirstlane(res->status) != SQOA_HIP_ST_SLOWPATH) return;  // only what the wave-parallel kernel declined: one scalar load otherwise
    const sqoa_hip_job job = jobs[jid];
    __syncthreads();                                                    // every lane has read the status
    if (only_slowpath && threadIdx.x == 0) { res->status = SQOA_HIP_ST_OK; row_mark_standby(res, 1); }

    const Hdr h = parse_header((gcptr8)job.src, res, threadIdx.x == 0);
    const bool shape_ok = (job.desc.width == 0 || job.desc.width == h.width) && (job.desc.height == 0 || job.desc.height == h.height);
    if (!h.ok || !shape_ok) { if (threadIdx.x == 0) res->status = SQOA_HIP_ST_BADARG; return; }
    const uint32_t oc = job.desc.channels ? job.desc.channels : (h.channels < 3 ? 1u : 3u) + ((h.channels & 1u) ? 0u : 1u);
    // what the record pipeline takes: QOI-compat colour streams and SQOA mono streams, any output channel count
    const bool mono = !h.qoi && h.channels < 3;
    const bool wave_ok = ((h.qoi && h.channels >= 3) || mono) && (unsigned long long)h.width * h.height * oc <= job.dst_cap;
    if (!wave_ok) {
        if (threadIdx.x == 0) decode_exact(job, h, res, &L.rec_v[0][0]);  // (the record buffers are free: 128 dwords of colour index)
        return;
    }
    // wave 0 turns bytes into op records, wave 1 turns records into pixels (seqoia.h:722-806)
    if (threadIdx.x < 64) { L.stamp[lane] = 0; if (lane < 2) { L.ready[lane] = 0; L.freed[lane] = 0; L.nrec[lane] = 0; } if (lane == 0) L.stop = 0; }
    for (uint32_t b0 = threadIdx.x; b0 < 256u; b0 += 128) {  // op lengths -- QOI: fe 4, ff 5, 10xxxxxx 2; SQOA mono: fe 2, ff 3; all others 1
        const uint32_t len = mono ? (b0 >= 0xfeu ? b0 - 0xfcu : 1u) : (b0 >= 0xfeu ? b0 - 0xfau : ((b0 & 0xc0u) == 0x80u ? 2u : 1u));
        const uint32_t opn = len == 1u ? (uint32_t)ST_A : len;
        L.ftab[b0] = opn | opn << 8 | 0x02000000u;
    }
    __syncthreads();
    uint32_t bad = 0;
    if (w == 0) qoi_front(job, mono, L, lane, bad);
    else qoi_back(job, h, mono, res, L, lane, bad);
    __syncthreads();
    if (L.stop == 2u) {  // a REF op in a mono stream: redo the image exactly
        __syncthreads();
        if (threadIdx.x == 0) decode_exact(job, h, res, &L.rec_v[0][0]);
        return;
    }
    if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
}

}  // namespace sqoa

extern "C" hipError_t sqoa_launch_decode_wave(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int shape,
                                              sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_qoi(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_retry(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, sqoa_hip_result* d_results, hipStream_t stream);

extern "C" hipError_t sqoa_launch_decode_qoi_spec(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const void* d_images, int n_images,
                                                  const void* d_wgs, int n_wgs, const void* d_tab, const void* d_flags, void* d_qstate, unsigned per,
                                                  int cus, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_passes(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const void* d_images, int n_images,
                                                const void* d_wgs, int n_wgs, void* d_tab, void* d_flags, unsigned per, int scans, int lb, hipStream_t stream);

// The kernels behind the wave decoder, over every job of the plan: what it marked SQOA_HIP_ST_SLOWPATH goes to the chunk-parallel QOI-compat
// kernel (colour streams to 3 / 4 channels) and whatever that leaves marked (mono, 1- / 2-channel output, SQOA_OP_REF) to the fallback kernel; a
// job that is not marked costs each of them one scalar load.
extern "C" hipError_t sqoa_launch_decode_standby(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, int retry, sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    // (`retry`: the launch in front was the balanced kernel, which hands on a stream with more op starts in a chunk than it has slots: the wave
    // decoder first, over the streams so marked)
    if (retry) e = sqoa_launch_decode_retry(d_jobs, d_ids, n, small, d_results, stream);
    if (e != hipSuccess) return e;
#if !defined(SQOA_QDEC) || SQOA_QDEC
    e = sqoa_launch_decode_qoi(d_jobs, d_ids, n, small, d_results, stream);
    if (e != hipSuccess) return e;
#endif
    hipLaunchKernelGGL(sqoa::decode_fallback_kernel, dim3(n), dim3(128), 0, stream, d_jobs, d_ids, d_results, 1);
    return hipGetLastError();
}

// Wave-parallel SQOA kernel first (the first `n_multi` ids: few large streams, decoded in three chip-wide passes; the others:
// one workgroup per stream); what it marks SQOA_HIP_ST_SLOWPATH goes to the chunk-parallel QOI-compat kernel (colour
// streams to 3 / 4 channels) and whatever that leaves marked (mono, 1- / 2-channel output, SQOA_OP_REF) to the fallback kernel.
// `shape`: the wave decoder's shape for the streams that get a workgroup each -- bit 0 = many short streams, bit 1 = twelve stream bytes per
// lane, bit 2 = the position-major store path (sqoa_launch_decode_wave; the plan decides, sqoa_runtime.hip)
// `n_b`, `shape_b`: the last n_b of the streams that get a workgroup each are launched apart from the others, with a shape of their own (the plan
// splits lists of mixed output formats: sqoa_runtime.hip) -- here, behind the others, or (`b_launched`) by the caller on a side stream, whose
// `join_event` this stream waits for before the kernels that look at every job's status
// `wave_only`: every job is SQOA_HIP_JOB_WAVEONLY -- neither the speculative QOI-compat pass nor the kernels behind the wave decoder are launched.
// `skip_standby`: the plan's lazy skip (sqoa_runtime.hip) -- ONLY the kernels behind the wave decoder (retry, QOI-compat, fallback) are left out;
// the speculative pass of the multi-pass QOI-compat streams and the scans it reads are part of every launch (a stream it accepts never marks its
// row as "stand-by used", so a plan of such streams would otherwise have lost its decode with the skip).
extern "C" hipError_t sqoa_launch_decode(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int shape, int n_b, int shape_b, int b_launched, void* join_event, int wave_only, int skip_standby, const sqoa_dp_tables* dp,
                                         sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int small = shape & 1;
    const int n_multi = dp ? dp->n_images : 0;
    hipError_t e = hipSuccess;
    // (the scan kernels between the passes: only where the speculative pass reads what they leave, or where the rows are too many
    // for every workgroup to compose them by itself)
#ifndef SQOA_DP_FUSE_ROWS
#define SQOA_DP_FUSE_ROWS 1024   /* (a lone 1 Mpixel stream, 670 rows: 0.0355 -> 0.032 ms; 2594 rows, one 3840 x 2160 stream: equal; 64 x 1 Mpixel, 5184 rows: 0.517 -> 0.531 ms) */
#endif
    const int scans = n_multi && ((!wave_only && dp->any_qoi) || dp->n_wgs > SQOA_DP_FUSE_ROWS);
    if (n_multi) e = sqoa_launch_decode_passes(d_jobs, d_results, dp->d_images, dp->n_images, dp->d_wgs, dp->n_wgs, dp->d_tab, dp->d_flags, dp->per, scans, dp->lb, stream);
    if (e != hipSuccess) return e;
    // the QOI-compat streams among them: speculative chip-wide pass + check (sqoa_decode_qoi.hip); what fails the check stays SLOWPATH
    if (n_multi && !wave_only && dp->any_qoi && dp->lb != 12) e = sqoa_launch_decode_qoi_spec(d_jobs, d_results, dp->d_images, dp->n_images, dp->d_wgs, dp->n_wgs, dp->d_tab, dp->d_flags,
                                                               dp->d_qstate, dp->per, dp->cus, stream);
    if (e != hipSuccess) return e;
    e = sqoa_launch_decode_wave(d_jobs, d_ids + n_multi, n - n_multi - n_b, shape, d_results, stream);
    if (e != hipSuccess) return e;
    if (n_b > 0 && !b_launched) e = sqoa_launch_decode_wave(d_jobs, d_ids + n - n_b, n_b, shape_b, d_results, stream);
    if (e != hipSuccess) return e;
    if (n_b > 0 && b_launched && join_event) e = hipStreamWaitEvent(stream, (hipEvent_t)join_event, 0);
    if (e != hipSuccess) return e;
    if (wave_only || skip_standby) return hipSuccess;   // every job is SQOA_HIP_JOB_WAVEONLY: what the wave decoder declines goes back to the caller as SLOWPATH
    return sqoa_launch_decode_standby(d_jobs, d_ids, n, small, ((shape | (n_b > 0 ? shape_b : 0)) & 8) ? 1 : 0, d_results, stream);
}
