// sqoa_decode_qoi.hip -- chunk-parallel decoder for QOI-compat colour streams (gfx950).
//
// Replaces the reference decoder loop (seqoia.h:722-806) for qoi_compat streams: ops INDEX / DIFF / LUMA / RUN (<= 62) /
// RGB / RGBA over the 64-entry running colour index (seqoia.h:753-760, :785-787).  QOI_OP_INDEX makes every later pixel
// depend on the VALUES of all earlier ones (the slot is a hash of the value), so pixel values cannot be scanned across a
// chunk the way SQOA's can.  What does parallelise:
//
//   * one workgroup of W waves per image, chunk c of 512 stream bytes -> wave c mod W (as in sqoa_decode_wave.hip): staging,
//     the parser as a scan of transition functions, every lane decoding the ops that start in its 8 bytes, the ops laid out
//     in stream order as records (transform, INDEX slot, pixel count), and -- between two INDEX ops -- the pixel values as
//     one segmented scan of the transforms, all BEFORE the chunk knows the pixel or the index it starts from;
//   * what is serial per image is the walk over the INDEX ops of a chunk: per INDEX op one add, one hash, one compare of
//     the 64 ops' hashes against the slot (__ballot) and one v_readlane -- "the most recent earlier op of this group whose
//     pixel hashes to my slot" -- with the index itself as the fall-back;
//   * the index lives in LDS, ONE copy per image, as 64 x {value, stamp}: "index[hash(px)] = px" for 64 ops at once is one
//     ds_max_u64 of (stream-order stamp << 32 | value) per op -- the last writer of a slot wins exactly as the
//     reference's in-order stores do -- and the waves hand the serial section to each other through the same tagged LDS
//     words as the SQOA decoder; nothing is copied between waves;
//   * pixel stores and run expansion happen after the chunk has handed on, off the serial path.
#include <type_traits>

#include "sqoa_decode_parts.h"

namespace sqoa {

// Waves per image, wave priority and poll interval (round 2, measured on 1024 x 1 Mpixel / 4096 icons of 512 x 512):
//   * the serial section is the image's critical path: it runs at priority 3, everything else at 0 (16.3 -> 14.5 ms; 3.88 ->
//     3.63 ms), and with that 8 waves per image beat 4 (13.5 ms; without the priority 8 waves were slower than 4: the wave
//     in its serial section got an eighth of the issue slots);
//   * polling a chain word costs scalar instructions, and the scalar pipe -- one per CU -- is what this kernel is short of
//     (SQ counters at one poll per 64 cycles: 994 SALU + 371 branches per 512-byte chunk, as many as VALU): 16 x 64 cycles
//     between two looks, 13.4 -> 12.2 ms (4 / 8 / 32: 12.7 / 12.4 / 12.3); polling fast only when the turn is next was
//     slower than that (13.0);
//   * small images want twice as many images at once: 2 waves per image (4096 icons: 3.55 ms against 4.3 with 4).
#ifndef SQOA_QDEC_W
#define SQOA_QDEC_W 8
#endif
#ifndef SQOA_QDEC_W_SMALL
#define SQOA_QDEC_W_SMALL 2
#endif
#ifndef SQOA_QDEC_SLEEP
#define SQOA_QDEC_SLEEP 16
#endif
#ifndef SQOA_QDEC_PRIO
#define SQOA_QDEC_PRIO 3
#endif
constexpr int QK_LB = 8;                          // stream bytes per lane
constexpr int QK_CHUNK = 64 * QK_LB;              // stream bytes per chunk
constexpr int QK_LDS_DW = QK_CHUNK / 4 + 8;       // staged dwords: chunk + look-ahead + misalignment
constexpr int QK_NLD = (QK_LDS_DW + 63) / 64;
constexpr uint32_t QK_SPIN_MAX = 1u << 24;

// One step of the segmented transform scan: `hf` (all ones / zero) marks lanes whose segment head lies at or below them
// within the distance covered so far; such lanes keep their transform, the others prepend the one arriving from below.
template <int CTRL, int ROWMASK>
SQOA_DEV Xf qk_seg_step(Xf T, uint32_t& hf) {
    Xf P;
    P.v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v, CTRL, ROWMASK, 0xf, false);
    P.abs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.abs, CTRL, ROWMASK, 0xf, false);
    const uint32_t pf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hf, CTRL, ROWMASK, 0xf, false);
    const Xf C = xf_then(P, T);
    const Xf R{(T.v & hf) | (C.v & ~hf), (T.abs & hf) | (C.abs & ~hf)};
    hf |= pf;
    return R;
}

SQOA_DEV bool qk_wait(unsigned long long* slot, uint32_t tag, uint32_t& val, uint32_t& bad) {
    for (uint32_t spin = 0; spin < QK_SPIN_MAX; ++spin) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (bcast_first((uint32_t)(v >> 32)) == tag) { val = bcast_first((uint32_t)v); return true; }
        __builtin_amdgcn_s_sleep(SQOA_QDEC_SLEEP);
    }
    bad = 1;
    return false;
}
// (release: everything the wave wrote to LDS before -- index updates included -- is done when the tag becomes visible)
SQOA_DEV void qk_post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) {
    if (lane == 0) __hip_atomic_store(slot, (unsigned long long)tag << 32 | val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// v_cndmask_b32 with the lane mask in an SGPR pair: the serial section keeps its lane sets as wave-uniform 64-bit masks
SQOA_DEV uint32_t select_mask(uint32_t if_clear, uint32_t if_set, unsigned long long mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

template <int W>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void decode_qoi_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                                                                   const int* __restrict__ job_ids,
                                                                                                   sqoa_hip_result* __restrict__ results) {
    __shared__ uint32_t s_bytes[W][QK_LDS_DW];
    __shared__ uint32_t s_ftab[256];               // byte -> low half of its parser transition function
    // per op of the wave's chunk, in stream order: transform value bytes (later: the transform scanned over its segment,
    // later still: the op's pixel), and 1 = INDEX | absolute-channel code (2 = rgb, 4 = rgba) | slot << 3 | pixel count << 9
    __shared__ uint32_t s_recv[W][QK_CHUNK];
    __shared__ uint16_t s_recm[W][QK_CHUNK];
    __shared__ unsigned long long s_tab[64];       // the colour index (seqoia.h:712, :785-787): stamp << 32 | value
    __shared__ unsigned long long s_parse[2][W];
    __shared__ unsigned long long s_pos[W];
    __shared__ unsigned long long s_px[W];
    __shared__ uint32_t s_go;

    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    const int jid = job_ids[blockIdx.x];
    const sqoa_hip_job job = jobs[jid];
    sqoa_hip_result* res = &results[jid];
    gcptr8 src0 = (gcptr8)job.src;

    // Only what the SQOA kernel handed on (SQOA_HIP_ST_SLOWPATH: header valid, destination large enough) and only QOI-compat
    // colour streams to 3 or 4 channels; the rest stays marked for decode_fallback_kernel.
    if (threadIdx.x == 0) {
        uint32_t go = 0;
        if (res->status == SQOA_HIP_ST_SLOWPATH) {
            const uint32_t hc = src0[12], q = src0[14] != 0x31;
            const uint32_t oc = job.desc.channels ? job.desc.channels : 3u + ((hc & 1u) ? 0u : 1u);
            go = q && hc >= 3 && oc >= 3;
        }
        s_go = go;
    }
    if (lane == 0) { s_parse[0][w] = 0; s_parse[1][w] = 0; s_pos[w] = 0; s_px[w] = 0; }
    if (threadIdx.x < 64) s_tab[threadIdx.x] = 0;   // seqoia.h:712
    for (uint32_t b0 = threadIdx.x; b0 < 256u; b0 += W * 64) {  // op lengths: fe 4, ff 5, 10xxxxxx 2, all others 1
        const uint32_t len = b0 >= 0xfeu ? b0 - 0xfau : ((b0 & 0xc0u) == 0x80u ? 2u : 1u);
        const uint32_t opn = len == 1u ? (uint32_t)ST_A : len;
        s_ftab[b0] = opn | opn << 8 | 0x02000000u;
    }
    __syncthreads();
    if (!s_go) return;
    __syncthreads();
    if (threadIdx.x == 0) res->status = SQOA_HIP_ST_OK;

    const uint32_t hw = (uint32_t)src0[4] << 24 | (uint32_t)src0[5] << 16 | (uint32_t)src0[6] << 8 | src0[7];
    const uint32_t hh = (uint32_t)src0[8] << 24 | (uint32_t)src0[9] << 16 | (uint32_t)src0[10] << 8 | src0[11];
    const uint32_t hc = src0[12];
    const uint32_t N = bcast_first(hw * hh);
    const int oc = (int)bcast_first(job.desc.channels ? job.desc.channels : 3u + ((hc & 1u) ? 0u : 1u));
    gptr8 out = (gptr8)job.dst;
    const bool dw_out = oc == 4 && (job.dst & 3ull) == 0;
    // (any byte address: offsets are relative to the dword-aligned address at or in front of the stream)
    const uint32_t soff = (uint32_t)(job.src & 3ull);
    gcptr8 src = src0 - soff;
    const uint32_t size = job.src_size + soff;
    const uint32_t p0 = kHeader + soff;            // first chunk byte: right behind the header (seqoia.h:703-707)
    const uint32_t chunks_len = size - 8;          // seqoia.h:721
    const uint32_t area = chunks_len > p0 ? chunks_len - p0 : 0;
    const uint32_t nchunks = area ? (area + QK_CHUNK - 1) / QK_CHUNK : 1;
    uint32_t bad = 0;

    uint32_t* lds = s_bytes[w];
    uint32_t* recv = s_recv[w];
    uint16_t* recm = s_recm[w];
    uint32_t stg[QK_NLD];
    auto load_chunk = [&](uint32_t c, uint32_t (&r)[QK_NLD]) {
        const uint32_t d0 = (p0 + c * QK_CHUNK) >> 2;
        if ((d0 + QK_LDS_DW) * 4u <= size) {
            gcptr32 p = (gcptr32)src + d0;
#pragma unroll
            for (int k = 0; k < QK_NLD - 1; ++k) r[k] = p[lane + 64 * k];
            r[QK_NLD - 1] = p[64 * (QK_NLD - 1) + (lane & 7)];
        } else {
#pragma unroll
            for (int k = 0; k < QK_NLD; ++k) {
                const uint32_t d = d0 + lane + 64 * k;
                r[k] = (k < QK_NLD - 1 || lane < QK_LDS_DW - 64 * (QK_NLD - 1)) && d * 4u < size ? ((gcptr32)src)[d] : 0u;
            }
        }
    };

    auto put = [&](uint32_t P, uint32_t v) {   // one pixel (seqoia.h:790-805, colour)
        if (dw_out) ((gptr32)out)[P] = v;
        else {
            gptr8 o = out + (size_t)P * (unsigned)oc;
            o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
            if (oc == 4) o[3] = (uint8_t)(v >> 24);
        }
    };

    uint32_t c = w;
    if (c < nchunks) load_chunk(c, stg);
    for (; c < nchunks; c += W) {
        const bool last_chunk = c == nchunks - 1;
        // ---- stage the chunk's bytes (+ look-ahead); prefetch the wave's next chunk ----
#pragma unroll
        for (int k = 0; k < QK_NLD - 1; ++k) lds[lane + 64 * k] = stg[k];
        if (lane < QK_LDS_DW - 64 * (QK_NLD - 1)) lds[lane + 64 * (QK_NLD - 1)] = stg[QK_NLD - 1];
        if (c + W < nchunks) load_chunk(c + W, stg);
        const uint32_t gs = p0 + c * QK_CHUNK;
        const uint32_t my0 = gs + QK_LB * lane;
        const uint32_t nlive = my0 >= chunks_len ? 0u : (chunks_len - my0 > (uint32_t)QK_LB ? (uint32_t)QK_LB : chunks_len - my0);
        const uint32_t loff = (gs & 3u) + QK_LB * lane;

        // ---- parse: the state in front of every lane's first byte (A, D1..D4; X past the chunk area) ----
        const unsigned long long own = lds_read8(lds, loff);
        Fn F{ID_LO, ID_HI};
#pragma unroll
        for (int k = 0; k < QK_LB; ++k) {
            const uint32_t b = (uint32_t)(own >> (8 * k)) & 0xffu;
            const uint32_t flo = (uint32_t)k < nlive ? s_ftab[b] : (ST_X | ST_X << 8 | 0x02000000u);
            F = compose(Fn{flo, 0x07060403u}, F);
        }
        const Fn Fi = wave_compose_scan(F);
        const Fn G{bcast_lane(Fi.lo, 63), bcast_lane(Fi.hi, 63)};
        const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
        const uint32_t g0 = G.lo & 0xffu;
        const bool g_const = G.lo == g0 * 0x01010101u && (G.hi & 0xffffu) == g0 * 0x0101u;
        unsigned long long* my_parse = &s_parse[(c / W) & 1][w];
        if (g_const && !last_chunk) qk_post(my_parse, c + 1, g0, lane);
        uint32_t st_in = ST_A;
        if (c > 0 && !qk_wait(&s_parse[((c - 1) / W) & 1][(c - 1) % W], c, st_in, bad)) break;
        if (!g_const && !last_chunk) qk_post(my_parse, c + 1, apply(G, st_in), lane);
        const uint32_t lst = apply(Fx, st_in);

        // ---- walk: my ops, in order (seqoia.h:740-775 with qoi_compat) ----
        uint32_t pos = lst >= 2u && lst <= 5u ? lst - 1u : (lst == ST_X ? (uint32_t)QK_LB : 0u);
        uint32_t rv[QK_LB], rm[QK_LB];
        uint32_t nops = 0, iters = QK_LB, cl = 0;
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) {
            const bool act = pos < nlive;
            if (__ballot(act) == 0ull) { iters = i; break; }
            const unsigned long long q = lds_read8(lds, loff + (pos < (uint32_t)QK_LB ? pos : (uint32_t)QK_LB));
            const uint32_t b0 = (uint32_t)q & 0xffu, b1 = ((uint32_t)q >> 8) & 0xffu;
            uint32_t v = 0, m = 0, cnt = 1, len = 1;
            if (b0 >= 0xfeu) {                       // seqoia.h:740-752
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
            } else cnt = (b0 & 0x3fu) + 1u;          // seqoia.h:773-775 (0xfd is a run of 62 here, 0xfc one of 61)
            rv[i] = v;
            rm[i] = act ? (m | cnt << 9) : 0u;
            nops += act ? 1u : 0u;
            cl += act ? cnt : 0u;
            pos += act ? len : 0u;
        }

        iters = bcast_first(iters);  // wave-uniform by construction: scalar slot tests below
        // ---- records in stream order ----
        const uint32_t oinc = wave_incl_scan(nops);
        const uint32_t T = bcast_lane(oinc, 63);
        const uint32_t obase = oinc - nops;
        const uint32_t ctot = bcast_lane(wave_incl_scan(cl), 63);   // pixels this chunk's ops produce
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) {
            if ((uint32_t)i >= iters) break;
            if (rm[i] >> 9) { recv[obase + i] = rv[i]; recm[obase + i] = (uint16_t)rm[i]; }
        }
        // ---- between two INDEX ops pixel values do not depend on the index: one segmented scan of the transforms per 64
        //      ops; a segment starts AT an INDEX op (whose own transform is the identity) and at the group's first op ----
#ifndef SQOA_DBG_Q_NOSEG
        for (uint32_t gb = 0; gb < T; gb += 64) {
            const bool live = gb + (uint32_t)lane < T;
            const uint32_t m = live ? recm[gb + lane] : 0u;
            const bool isx = m & 1u;
            Xf S{(live && !isx) ? recv[gb + lane] : 0u, (m & 2u) ? 0x00ffffffu : (m & 4u) ? 0xffffffffu : 0u};
            uint32_t hf = (isx || lane == 0) ? ~0u : 0u;
            S = qk_seg_step<DPP_ROW_SHR + 1, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 2, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 4, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 8, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_BCAST15, 0xa>(S, hf);
            S = qk_seg_step<DPP_ROW_BCAST31, 0xc>(S, hf);
            if (live) {
                recv[gb + lane] = S.v;
                recm[gb + lane] = (uint16_t)((m & ~6u) | (S.abs == 0xffffffffu ? 4u : S.abs ? 2u : 0u));
            }
        }
#endif

        // ---- pixel position: a plain sum, handed on at once ----
        uint32_t pos_in = 0;
        if (c > 0 && !qk_wait(&s_pos[(c - 1) % W], c, pos_in, bad)) break;
        const uint32_t pos_out = pos_in + ctot > N ? N : pos_in + ctot;
        if (!last_chunk) qk_post(&s_pos[w], c + 1, pos_out, lane);

        // ---- the serial section: the pixel entering the chunk (and with it the index as the chunks before left it) ----
        uint32_t pxv = kPxInit;   // seqoia.h:716-719
        __builtin_amdgcn_s_setprio(SQOA_QDEC_PRIO);   // the image's critical path: ahead of the waves that are preparing later chunks
        if (c > 0 && !qk_wait(&s_px[(c - 1) % W], c, pxv, bad)) break;
#ifdef SQOA_DBG_Q_NOSERIAL
        if (false) {
#else
        if (pos_in < N) {
#endif
            for (uint32_t gb = 0; gb < T; gb += 64) {
                const uint32_t ng = T - gb < 64u ? T - gb : 64u;
                const bool live = (uint32_t)lane < ng;
                const uint32_t m = live ? recm[gb + lane] : 0u;
                const uint32_t Sv = live ? recv[gb + lane] : 0u;
                const uint32_t nabs = (m & 2u) ? 0xff000000u : (m & 4u) ? 0u : 0xffffffffu;   // channels still relative to the segment's start
                unsigned long long idxmask = __ballot((m & 1u) != 0u);
                uint32_t vres = 0, hres = ~0u;   // pixel and index slot of every op resolved so far (unresolved: no slot)
                uint32_t s0 = 0;
                for (;;) {
                    const uint32_t j = idxmask ? (uint32_t)__builtin_ctzll(idxmask) : ng;   // the next INDEX op, or the end of the group
                    // ops [s0, j): their segment starts from pxv
                    const uint32_t val = add8(pxv & nabs, Sv);
                    const uint32_t hv = qoi_hash(val);
                    const unsigned long long seg = (j >= 64u ? ~0ull : (1ull << j) - 1ull) & ~((1ull << s0) - 1ull);
                    vres = select_mask(vres, val, seg);
                    hres = select_mask(hres, hv, seg);
                    if (j >= ng) break;
                    // seqoia.h:753-755: the slot as the ops so far left it -- the most recent op of this group whose pixel hashes
                    // to it, else the index in LDS (ops of earlier groups and chunks)
                    const uint32_t slot = ((uint32_t)__builtin_amdgcn_readlane((int)m, (int)j) >> 3) & 63u;
                    const unsigned long long cand = __ballot(hres == slot);
                    if (cand) pxv = (uint32_t)__builtin_amdgcn_readlane((int)vres, 63 - __builtin_clzll(cand));
                    else pxv = bcast_first((uint32_t)__hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    idxmask &= idxmask - 1ull;
                    s0 = j;
                }
                pxv = (uint32_t)__builtin_amdgcn_readlane((int)vres, (int)(ng - 1u));
                // seqoia.h:785-787 for the whole group: the last writer of a slot wins
                if (live) {
                    __hip_atomic_fetch_max(&s_tab[hres & 63u], (unsigned long long)((c << 10) + gb + (uint32_t)lane + 1u) << 32 | vres, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    recv[gb + lane] = vres;   // for the stores below
                }
            }
        }
        if (!last_chunk) qk_post(&s_px[w], c + 1, pxv, lane);
        __builtin_amdgcn_s_setprio(0);

        // ---- off the serial path: pixels of this chunk's ops; literal ops lane-local, runs by the whole wave ----
        uint32_t P0 = pos_in;
#ifdef SQOA_DBG_Q_NOOUT
        P0 = N;
#endif
        for (uint32_t gb = 0; gb < T && P0 < N; gb += 64) {
            const bool live = gb + (uint32_t)lane < T;
            const uint32_t cnt = live ? (uint32_t)recm[gb + lane] >> 9 : 0u;
            const uint32_t v = live ? recv[gb + lane] : 0u;
            const uint32_t pinc = wave_incl_scan(cnt);
            const uint32_t P = P0 + pinc - cnt;
            if (cnt >= 1u && cnt <= 2u && P < N) put(P, v);
            if (cnt == 2u && P + 1u < N) put(P + 1u, v);
            unsigned long long runs = __ballot(cnt > 2u && P < N);
            while (runs) {  // seqoia.h:723-725
                const int l = __builtin_ctzll(runs);
                runs &= runs - 1ull;
                const uint32_t rp = bcast_lane(P, l), rvv = bcast_lane(v, l);
                uint32_t rc = bcast_lane(cnt, l);
                if (rp + rc > N) rc = N - rp;
                for (uint32_t k = lane; k < rc; k += 64) put(rp + k, rvv);
            }
            P0 += bcast_lane(pinc, 63);
        }
        if (last_chunk) {  // stream exhausted: the last pixel repeats (seqoia.h:726)
            for (uint32_t k = pos_out + lane; k < N; k += 64) put(k, pxv);
            if (lane == 0) res->len = (int)(N * (uint32_t)oc);
        }
    }
    if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
}

}  // namespace sqoa

// `small`: the plan's streams are short (many more images than the chip holds workgroups of the large shape)
extern "C" hipError_t sqoa_launch_decode_qoi(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (small) hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W_SMALL>), dim3(n), dim3(SQOA_QDEC_W_SMALL * 64), 0, stream, d_jobs, d_ids, d_results);
    else hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W>), dim3(n), dim3(SQOA_QDEC_W * 64), 0, stream, d_jobs, d_ids, d_results);
    return hipGetLastError();
}
