// sqoa_decode_qoi.hip -- chunk-parallel decoder for QOI-compat colour streams (gfx950).
//
// Replaces the reference decoder loop (seqoia.h:722-806) for qoi_compat streams: ops INDEX / DIFF / LUMA / RUN (<= 62) /
// RGB / RGBA over the 64-entry running colour index (seqoia.h:753-760, :785-787).  QOI_OP_INDEX makes every later pixel
// depend on the VALUES of all earlier ones (the slot is a hash of the value), so pixel values cannot be scanned across a
// chunk the way SQOA's can.  What does parallelise:
//
//   * one workgroup per image: W - 1 PRODUCER waves and one RESOLVER wave.  Chunk c of 512 stream bytes -> producer c mod
//     (W - 1), as in sqoa_decode_wave.hip: staging, the parser as a scan of transition functions, every lane decoding the ops
//     that start in its 8 bytes, the ops laid out in stream order as records (transform, INDEX slot, pixel count), and --
//     between two INDEX ops -- the pixel values as one segmented scan of the transforms, all without knowing the pixel or the
//     index the chunk starts from;
//   * what is serial per image is the resolution of the INDEX ops, 64 records at a time.  A group without one (two out of three
//     on the headline batch) is an add, a hash and LDS writes nobody waits for; a group with INDEX ops settles all of them at once
//     as a fixed point -- "my pixel is the one of the most recent earlier op of the group that hashes to my slot, else the index
//     as the groups before left it" through per-slot lane masks in LDS and two ds_bpermute per round, one round as a rule.  ONE
//     wave does that for every chunk of the image, back to back, at raised priority: the running pixel stays in a register, there
//     is no hand-over from wave to wave per chunk (which cost a post, a poll and a wake-up on the image's critical path 5 000
//     times per Mpixel image).  That wave's dependent instructions ARE the image's critical path: DESIGN.md 3.3 has what each
//     kind of them costs;
//   * the index lives in LDS, ONE copy per image, as 64 x {value, stamp}: "index[hash(px)] = px" for 64 ops at once is one
//     ds_max_u64 of (stream-order stamp << 32 | value) per op -- the last writer of a slot wins exactly as the
//     reference's in-order stores do;
//   * the producer picks its records up again when the resolver is done with them -- by then it has parsed and decoded its
//     next chunk -- and does the pixel stores and run expansion, off the serial path.
#include <type_traits>

#include "sqoa_decode_parts.h"

namespace sqoa {

#ifdef SQOA_ASM_MARK  // phase markers as comments in the assembly (tools/asm_cost.py by region)
#define QP(k) asm volatile("; SQOA_QPHASE " #k)
#else
#define QP(k) do { } while (0)
#endif

// Waves per image (one of them the resolver), wave priority and poll intervals.  History on 1024 x 1 Mpixel / 4096 icons of
// 512 x 512, every wave both producing and resolving its own chunks: the serial section at priority 3 16.3 -> 14.5 ms (3.88 ->
// 3.63 ms), with that 8 waves per image instead of 4 13.5 ms; polling a chain word costs scalar instructions and the scalar pipe
// -- one per CU -- was saturated (994 SALU + 371 branches per 512-byte chunk, as many as VALU): 16 x 64 cycles between two
// looks 12.2 ms.  Small images want more images at once: fewer waves per image.
#ifndef SQOA_QDEC_W
#define SQOA_QDEC_W 8
#endif
#ifndef SQOA_QDEC_W_SMALL
#define SQOA_QDEC_W_SMALL 4
#endif
#ifndef SQOA_QDEC_SLEEP
#define SQOA_QDEC_SLEEP 16      // producers: x 64 cycles between two looks at a chain word
#endif
#ifndef SQOA_QDEC_RSLEEP
#define SQOA_QDEC_RSLEEP 8      // the resolver, when the next chunk's records are not there yet (1 -> 8 in round 5: icons 1 % faster, the rest equal)
#endif
#ifndef SQOA_QDEC_PRIO
#define SQOA_QDEC_PRIO 3
#endif
#ifndef SQOA_QDEC_LDS_STORES   // the producers' pixels through LDS in position order, out as 16-byte pieces (round 5; with SQOA_QDEC_BAL)
#define SQOA_QDEC_LDS_STORES 1
#endif
#ifndef SQOA_QDEC_SEQ_MAX   // the resolver: groups of up to this many INDEX ops are settled one op after the other, in registers (round 5); 0: always the fixed point
#define SQOA_QDEC_SEQ_MAX 4
#endif
#ifndef SQOA_QDEC_TABW   // the resolver's index update a plain write (the highest lane on a slot stays) instead of a stamped ds_max_u64
#define SQOA_QDEC_TABW 1
#endif
#ifndef SQOA_QDEC_BAL   // the producers' walk balanced over the lanes (round 5)
#define SQOA_QDEC_BAL 1
#endif
constexpr int QK_LB = 8;                          // stream bytes per lane
constexpr int QK_CHUNK = 64 * QK_LB;              // stream bytes per chunk
constexpr int QK_LDS_DW = QK_CHUNK / 4 + 8;       // staged dwords: chunk + look-ahead + misalignment
constexpr int QK_NLD = (QK_LDS_DW + 63) / 64;
[[maybe_unused]] constexpr uint32_t QK_SPIN_MAX = 1u << 24;

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

// Wait for the 8-byte chain word to carry `tag` (high half); the low half comes back in `val`.  The hand-written poll of
// sqoa_common.h: the loop as the compiler builds it is 25 scalar instructions and 8 branches on the path where the first look
// succeeds, four waits per chunk and producer, and the producers retire as many scalar instructions as vector ones -- the one
// scalar pipe of the compute unit is what bounds them (profiles/r03_f_qoi_pmc_summary.txt: 165 SALU + 60 branches per 64 pixels).
// (What the tag guards was written to LDS by the posting wave BEFORE its one ds_write_b64 of tag and value -- qk_post -- and LDS
// executes a wave's instructions in order; the poll waits for its own reads, lgkmcnt(0), before anything behind it is issued.)
template <int SLEEP = SQOA_QDEC_SLEEP>
SQOA_DEV bool qk_wait(unsigned long long* slot, uint32_t tag, uint32_t& val, uint32_t& bad) {
#ifdef SQOA_QDEC_C_POLL   // the loop as the compiler makes it (A/B builds)
    for (uint32_t spin = 0; spin < QK_SPIN_MAX; ++spin) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (bcast_first((uint32_t)(v >> 32)) == tag) { val = bcast_first((uint32_t)v); return true; }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
    bad = 1;
    return false;
#else
    const uint32_t t = chain_poll<SLEEP>(slot, tag, 0xffffffffu, val);   // (no tag is ever 2^32 - 1: chunk numbers + 1)
    if (t != tag) { bad = 1; return false; }
    return true;
#endif
}
// (release: everything the wave wrote to LDS before -- index updates included -- is done when the tag becomes visible)
SQOA_DEV void qk_post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) { lds_post(slot, tag, val, lane); }

// v_cndmask_b32 with the lane mask in an SGPR pair: the serial section keeps its lane sets as wave-uniform 64-bit masks
SQOA_DEV uint32_t select_mask(uint32_t if_clear, uint32_t if_set, unsigned long long mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

// What the speculative pass (SPEC, below) leaves per workgroup for the check that follows it: the resolver's state -- running pixel and
// the 64 index entries -- where the workgroup's own chunks begin (behind its warm-up) and where they end.
struct dq_state { uint32_t px, tab[64], pad[63]; };   // 512 bytes
typedef dp_chunk dp_chunk_row;   // (sqoa_common.h: the rows passes 1 and 2 of decode_wave_kernel and the two scans leave)
static_assert(QK_CHUNK == DP_CHUNK_BYTES, "the speculative pass indexes the passes' tables with its own chunk numbers");

// SPEC = false: one workgroup per stream, from its first byte (the batch shape).
// SPEC = true: the chip-wide route for FEW LARGE QOI-compat streams.  A stream's decoder state (pixel, index[64]) cannot be summarised
// over a range of ops, but it can be GUESSED: a workgroup takes a range of `per` chunks and first decodes the `per`
// chunks in front of it from a reset state (pixel (0,0,0,255), index all zero) without storing anything -- on photo-like content
// every index slot has been rewritten and the pixel re-anchored by an RGB op long before its own chunks begin.  Parser states and
// pixel positions are exact (passes 1 and 2 of decode_wave_kernel + the two scans).  Every workgroup leaves its state at the start
// and at the end of its own chunks; dq_verify_kernel accepts the stream only if EVERY workgroup's guessed start state equals the
// end state of the workgroup in front -- by induction from the first one, which starts exactly, all pixels are then the
// reference's -- and leaves it to the one-workgroup decoder (status still SLOWPATH) otherwise.
// TABW: the resolver's index update as a plain write, which rests on the LDS applying an instruction's lanes in ascending order (the instance for
// devices that have passed the library's probe of that order, sqoa_selftest.hip); false: the stamped ds_max_u64, which does not.
template <int W, bool SPEC = false, bool BAL = SQOA_QDEC_BAL != 0, bool TABW = SQOA_QDEC_TABW != 0>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void decode_qoi_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                                                                   const int* __restrict__ job_ids,
                                                                                                   sqoa_hip_result* __restrict__ results,
                                                                                                   const dp_image* __restrict__ dp_images = nullptr,
                                                                                                   const dp_wg* __restrict__ dp_wgs = nullptr,
                                                                                                   const dp_chunk_row* __restrict__ dp_tab = nullptr,
                                                                                                   const uint32_t* __restrict__ dp_flags = nullptr,
                                                                                                   dq_state* __restrict__ dq = nullptr, uint32_t dp_per = 0,
                                                                                                   uint32_t dp_span = 1) {
    constexpr int P = W - 1;                       // producer waves; wave 0 is the resolver
    __shared__ uint32_t s_bytes[P][QK_LDS_DW];
    __shared__ uint32_t s_ftab[256];               // byte -> low half of its parser transition function
    // op byte -> {length | 1 << 3 | record word << 16, absolute-channel mask, relative value bytes 0 and 2, LUMA nibble mask |
    // relative value byte 1}: the op decode of seqoia.h:740-775 (qoi_compat) as one look-up and mask arithmetic (the if / else
    // chain it replaces was 85 VALU + 36 scalar instructions per op slot); entry 256 = "no op" for lanes that have run out
    __shared__ __attribute__((aligned(16))) u32x4 s_qtab[257];
    // per op of the producer's chunk, in stream order: transform value bytes (later: the transform scanned over its segment,
    // later still: the op's pixel), and 1 = INDEX | absolute channels (2 = rgb, 6 = rgba) | slot << 3 | pixel count << 9
    // (one 8-byte record: the resolver -- the image's serial path -- takes a group's 64 records with one LDS instruction)
    __shared__ __attribute__((aligned(16))) uint2 s_rec[P][QK_CHUNK];
    __shared__ unsigned long long s_tab[64];       // the colour index (seqoia.h:712, :785-787): stamp << 32 | value
    __shared__ unsigned long long s_parse[2][P];
    __shared__ unsigned long long s_pos[P];
    __shared__ unsigned long long s_ready[P];      // producer -> resolver: chunk + 1 << 32 | ops in the chunk | 1 << 31: nothing to resolve (past the image)
    __shared__ unsigned long long s_done[P];       // resolver -> producer: chunk + 1 << 32 | the pixel the chunk ends with
    __shared__ unsigned long long s_smask[64];     // resolver scratch: per index slot, the lanes of the group whose pixel hashes to it
    __shared__ uint32_t s_go;
    [[maybe_unused]] __shared__ uint32_t s_hwid;   // (diagnostic builds)

    SQOA_PLACEMENT_PAD();
    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    uint32_t chunk_first = 0, wg_first = 0;   // SPEC: this workgroup's first chunk, and the image's first row of the tables
    int jid;
    if constexpr (SPEC) {
        const dp_wg me = dp_wgs[blockIdx.x];
        if (__builtin_amdgcn_readfirstlane(dp_flags[me.img]) != 3u) return;   // not a QOI-compat colour stream (pass 1 of decode_wave_kernel)
        jid = dp_images[me.img].jid;
        chunk_first = me.chunk0;
        wg_first = dp_images[me.img].tab_off;
        if ((chunk_first / dp_per) % dp_span != 0u) return;   // one workgroup per dp_span rows of the tables (sqoa_launch_decode_qoi_spec)
    } else jid = job_ids[blockIdx.x];
    sqoa_hip_result* res = &results[jid];
    // Launched over every stream of the plan, after the SQOA kernel: a stream that kernel has decoded (or refused) costs this
    // workgroup one scalar load -- nothing is built, no barrier is met.  (Wave-uniform: every wave reads the same word, and
    // nothing in this launch writes it before the __syncthreads below.)
    if (__builtin_amdgcn_readfirstlane(res->status) != SQOA_HIP_ST_SLOWPATH) return;
    const sqoa_hip_job job = jobs[jid];
    gcptr8 src0 = (gcptr8)job.src;
    if (!SPEC && threadIdx.x == 0) row_mark_standby(res, 1);   // (a job the wave decoder handed on: the plan keeps launching the kernels behind it)

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
    if (lane == 0 && w < P) { s_parse[0][w] = 0; s_parse[1][w] = 0; s_pos[w] = 0; s_ready[w] = 0; s_done[w] = 0; }
    // seqoia.h:712.  (SPEC, a warm-up that starts inside the stream: the entries are unknown there, and an INDEX op of the warm-up that
    // reads one before anything has written it gets a pixel that is wrong whatever stands here -- but an opaque one is wrong in
    // three channels, which the next RGB op repairs, and not in four, which only an RGBA op would: a 3-channel stream has none.)
    if (threadIdx.x < 64) s_tab[threadIdx.x] = SPEC && chunk_first > dp_per ? (unsigned long long)kPxInit : 0ull;
    for (uint32_t b0 = threadIdx.x; b0 < 256u; b0 += W * 64) {  // op lengths: fe 4, ff 5, 10xxxxxx 2, all others 1
        const uint32_t len = b0 >= 0xfeu ? b0 - 0xfau : ((b0 & 0xc0u) == 0x80u ? 2u : 1u);
        const uint32_t opn = len == 1u ? (uint32_t)ST_A : len;
        s_ftab[b0] = opn | opn << 8 | 0x02000000u;
        const bool is_rgb = b0 == 0xfeu, is_rgba = b0 == 0xffu, is_idx = b0 < 0x40u, is_diff = (b0 & 0xc0u) == 0x40u, is_luma = (b0 & 0xc0u) == 0x80u;
        const uint32_t cnt = (is_rgb || is_rgba || is_idx || is_diff || is_luma) ? 1u : (b0 & 0x3fu) + 1u;   // seqoia.h:773-775 (0xfd / 0xfc: runs of 62 / 61 here)
        const uint32_t vg = (b0 & 0x3fu) - 32u;
        u32x4 e;
        e.x = len | 1u << 3 | ((is_idx ? (1u | b0 << 3) : 0u) | cnt << 9) << 16;
        e.y = is_rgb ? 0x00ffffffu : is_rgba ? 0xffffffffu : 0u;
        e.z = is_diff ? (((((b0 >> 4) & 3u) - 2u) & 0xffu) | (((b0 & 3u) - 2u) & 0xffu) << 16)      // seqoia.h:756-760
                      : is_luma ? (((vg - 8u) & 0xffu) | ((vg - 8u) & 0xffu) << 16) : 0u;          // seqoia.h:761-769
        e.w = is_diff ? ((((b0 >> 2) & 3u) - 2u) & 0xffu) << 8 : is_luma ? ((vg & 0xffu) << 8 | 0x000f000fu) : 0u;
        s_qtab[b0] = e;
    }
    if (threadIdx.x == 0) { u32x4 z = {0u, 0u, 0u, 0u}; s_qtab[256] = z; }
    __syncthreads();
    if (!s_go) return;
    __syncthreads();
    if (!SPEC && threadIdx.x == 0) res->status = SQOA_HIP_ST_OK;   // (SPEC: the check behind this pass decides)

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
    // SPEC: my own chunks [chunk_first, c_end), the warm-up [c_begin, chunk_first) in front of them; else the whole stream
    const uint32_t c_begin = SPEC ? (chunk_first > dp_per ? chunk_first - dp_per : 0u) : 0u;
    const uint32_t c_end = SPEC ? (chunk_first + dp_span * dp_per < nchunks ? chunk_first + dp_span * dp_per : nchunks) : nchunks;
    if (SPEC && chunk_first >= nchunks) return;   // (the tables are laid out per 512 bytes of the longest chunk area; nothing here)
    const dp_chunk_row* rows = SPEC ? dp_tab + wg_first : nullptr;
    const uint32_t seed_st = SPEC && c_begin ? bcast_first(rows[c_begin / dp_per].st_in) : (uint32_t)ST_A;
    const uint32_t seed_pos = SPEC && chunk_first ? bcast_first(rows[chunk_first / dp_per].pos_in) : 0u;
    dq_state* my_dq = SPEC ? dq + 2 * (size_t)(wg_first + chunk_first / dp_per) : nullptr;   // [0]: at my first chunk, [1]: behind my last
    // pad[0] of the row behind my last chunk counts the waves of this workgroup that came through without an error: the check wants all W
    // (a workgroup that stops early must not leave the rows of an earlier launch standing)
    if (SPEC && threadIdx.x == 0) __hip_atomic_store(&my_dq[1].pad[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();

    // ================= the resolver: every chunk's serial section, in stream order ==========================================
    if (w == 0) {
        __builtin_amdgcn_s_setprio(SQOA_QDEC_PRIO);   // the image's critical path: ahead of the waves that are preparing later chunks
#ifdef SQOA_QDEC_HWDUMP
        { uint32_t h, x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
          if (lane == 0) s_hwid = (h & 0x0fffffffu) | x << 28; }
#endif
        const unsigned long long bit = 1ull << lane, lt_mask = bit - 1ull;
        // seqoia.h:716-719; SPEC, a warm-up that does not begin at the stream's start: pass 2's guess of the pixel there (decode_wave_kernel)
        uint32_t pxv = SPEC && c_begin ? bcast_first(rows[c_begin / dp_per].px_in) : kPxInit;
        [[maybe_unused]] unsigned long long rmask = 0;   // SPEC: index slots my own chunks read before writing them (per lane; OR-ed at the end)
#ifdef SQOA_QTRACE  // diagnostic builds: where the resolver's time goes, written over the image's first pixels
        unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#define RPH(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define RPH(k) do { } while (0)
#endif
        uint32_t rp = c_begin % (uint32_t)P;
        for (uint32_t c = c_begin; c < c_end; ++c) {
            if (SPEC && c == chunk_first) {   // the state my own chunks start from: exact (chunk 0) or what the warm-up arrived at
                my_dq[0].px = pxv;
                my_dq[0].tab[lane] = (uint32_t)__hip_atomic_load(&s_tab[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            uint32_t hopen = pxv;   // the pixel the segment that is open at a group's first op started from: the one entering the chunk, or an INDEX op's
            const uint32_t p = rp;   // c mod P
            rp = rp + 1u == (uint32_t)P ? 0u : rp + 1u;
            uint32_t info = 0;
            uint2* rec = s_rec[p];
            RPH(0);
            if (!qk_wait<SQOA_QDEC_RSLEEP>(&s_ready[p], c + 1, info, bad)) break;
            RPH(1);
            const uint32_t T = info & 0xffffu;
#ifdef SQOA_DBG_Q_NOSERIAL
            if (false) {
#else
            if (!(info >> 31)) {
#endif
                uint32_t vlast = 0;   // the pixels of the group before: the chunk ends with its last op's
                for (uint32_t gb = 0; gb < T; gb += 64) {
                    const uint32_t ng = T - gb < 64u ? T - gb : 64u;
                    const bool live = (uint32_t)lane < ng;
                    // seqoia.h:785-787 for the whole group: the last writer of a slot wins; and the pixels, for the producer's stores
                    auto commit = [&](uint32_t val, uint32_t hv) {
                        if (live) {
                            // seqoia.h:785-787 for 64 ops at once: of the lanes of one LDS write that meet on a slot the HIGHEST stays (gfx950 applies
                            // an instruction's lanes in ascending order: tools/ubench/lds_atomic_order.hip, run by the GPU tests), and a later
                            // group's write comes later -- the last writer of a slot wins as in the reference's loop, by a plain ds_write_b32.  (The
                            // speculative route's check wants to know WHEN an entry was written: there the stamped ds_max_u64 stays.)
                            if constexpr (SPEC || !TABW)
                                __hip_atomic_fetch_max(&s_tab[hv], (unsigned long long)((c << 10) + gb + (uint32_t)lane + 1u) << 32 | val, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            else
                                __hip_atomic_store(reinterpret_cast<uint32_t*>(&s_tab[hv]), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            rec[gb + lane].x = val;
                        }
                        vlast = val;
#ifdef SQOA_QTRACE   // (6: index update and hand-back; 8: groups)
                        RPH(6); ph[8] += 1;
#endif
                    };
                    // (a plain load: gb + lane < 512 always; lanes past the chunk's last op get an empty record word, their value is
                    // never looked at)
                    const uint2 r = rec[gb + lane];
                    const uint32_t m = live ? r.y : 0u, Sv = r.x;
                    uint32_t mN_, mA_;   // ~0: all four channels absolute / at least r g b
                    asm("v_bfe_i32 %0, %1, 2, 1" : "=v"(mN_) : "v"(m));
                    asm("v_bfe_i32 %0, %1, 1, 1" : "=v"(mA_) : "v"(m));
                    const uint32_t nabs = ~(mN_ | (mA_ & 0x00ffffffu));   // channels still relative to the segment's start
                    const bool isx = (m & 1u) != 0u;
                    const uint32_t slot = (m >> 3) & 63u;
                    const unsigned long long idxmask = __ballot(isx);
                    unsigned long long earl = 0, tab64 = 0;   // SPEC: the group-internal writers of an INDEX op's slot, as of the last round; the slot's entry
                    uint32_t X = 0, val = 0, hv = 0;
                    RPH(2);
                    if (idxmask == 0ull) {
                        // No INDEX op in the group (two groups out of three on the headline batch): every pixel follows from the one the
                        // open segment started at.  Nothing is read from LDS here -- not the index, not a ds_bpermute of segment heads
                        // that do not exist: the image's serial path is this wave's dependent instructions, and every round trip to LDS
                        // on it is a hundred-odd cycles (the writes further down are not waited for).
                        val = add8(hopen & nabs, Sv);
                        RPH(4);
                        commit(val, qoi_hash(val));
                        continue;
                    }
                    if constexpr (SQOA_QDEC_SEQ_MAX > 0) {
                        // A group with FEW INDEX ops (two or three on photo-like content): one after the other, in stream order, and without a
                        // round trip to LDS between them.  When INDEX op l's turn comes every lane below it holds its final pixel; "the most
                        // recent earlier op of the group whose pixel hashes to my slot" is then one compare of all lanes' hashes with the
                        // slot (a scalar), a mask of the lanes below l, a count of leading zeros -- scalar instructions -- and a v_readlane;
                        // failing that the index entry, fetched for all the group's INDEX ops at once in front of the loop.  The lanes from l
                        // on take the pixel as their segment's start (the next INDEX op overrides it from its own lane on) and every lane
                        // recomputes pixel and hash: seven VALU instructions.  Twenty-six instructions per INDEX op, where the fixed point
                        // below is four dependent LDS round trips plus its 64-bit mask arithmetic per round whatever the count: headline
                        // batch 7.67 -> 7.24 ms at a threshold of 3 to 6 ops (8: 7.31, 16: 7.97).
                        if ((uint32_t)__builtin_popcountll(idxmask) <= (uint32_t)SQOA_QDEC_SEQ_MAX) {
                            // (the entries' values alone, one register, unless the check of the speculative route wants the stamps: the half
                            // of a 64-bit load nobody reads is the next temporary, and the wave would wait for the load before the first add)
                            uint32_t tabv;
                            if constexpr (SPEC) {
                                tab64 = isx ? __hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                                tabv = (uint32_t)tab64;
                            } else
                                tabv = isx ? __hip_atomic_load(reinterpret_cast<uint32_t*>(&s_tab[slot]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                            uint32_t hx = hopen, Xl = 0;
                            unsigned long long rem = idxmask;
                            RPH(3);
                            for (;;) {
                                val = add8(hx & nabs, Sv);
                                hv = qoi_hash(val);
                                if (rem == 0ull) break;
                                const int l = __builtin_ctzll(rem);
                                rem &= rem - 1ull;
                                const uint32_t sl = bcast_lane(slot, l);
                                const unsigned long long earlier = __ballot(hv == sl) & ((1ull << l) - 1ull);
                                Xl = earlier ? bcast_lane(val, 63 - __builtin_clzll(earlier)) : bcast_lane(tabv, l);
                                if constexpr (SPEC) {
                                    if (c >= chunk_first && earlier == 0ull && bcast_lane((uint32_t)(tab64 >> 32), l) <= (chunk_first << 10)) rmask |= 1ull << sl;
                                }
                                hx = (uint32_t)lane >= (uint32_t)l ? Xl : hx;
                            }
                            RPH(4);
                            hopen = Xl;
#ifdef SQOA_QTRACE
                            ph[9] += 1;
#endif
                            commit(val, hv);
                            continue;
                        }
                    }
                    {
                        // the INDEX op at or below me that my segment starts at (none: it starts from the pixel entering the group)
                        const unsigned long long below = idxmask & (lt_mask | bit);
                        const uint32_t head = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                        const uint32_t from_in = below ? 0u : ~0u;
                        // All the group's INDEX ops at once, as a fixed point (seqoia.h:753-755 for each of them): an INDEX op's pixel
                        // X is "the most recent earlier op whose pixel hashes to my slot, else the index as the groups before left
                        // it"; the pixels of the ops behind it follow from X.  Start from the index, recompute until nothing
                        // changes: every round settles at least the first INDEX op that was still wrong (the ops in front of it are
                        // final), in practice one round finds nothing to change -- where one INDEX op after the other cost a
                        // VALU -> SALU -> VALU round trip each.
                        tab64 = isx ? __hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                        const uint32_t tabv = (uint32_t)tab64;
                        X = tabv;
                        RPH(3);
                        for (uint32_t round = 0; round <= ng; ++round) {
                            const uint32_t hx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(head << 2), (int)X);
                            val = add8(((hx & ~from_in) | (hopen & from_in)) & nabs, Sv);
                            hv = qoi_hash(val);
                            s_smask[lane] = 0ull;                            // per slot: the lanes whose pixel hashes to it
                            if (live) __hip_atomic_fetch_or(&s_smask[hv], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned long long earlier = __hip_atomic_load(&s_smask[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & lt_mask;
                            const uint32_t last = earlier ? 63u - (uint32_t)__builtin_clzll(earlier) : 0u;
                            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(last << 2), (int)val);
                            const uint32_t Xn = earlier ? got : tabv;
                            const bool changed = isx && Xn != X;
                            X = Xn;
                            earl = earlier;
                            if (__ballot(changed) == 0ull) break;
                        }
                    }
                    if constexpr (SPEC) {
                        // an INDEX op of my OWN chunks that took its pixel from an index entry nothing of my own chunks has written yet:
                        // the entry is the warm-up's guess -- the check must find it equal to the true one (dq_verify_kernel)
                        if (c >= chunk_first && isx && earl == 0ull && (uint32_t)(tab64 >> 32) <= (chunk_first << 10)) rmask |= 1ull << slot;
                    }
                    RPH(4);
                    hopen = (uint32_t)__builtin_amdgcn_readlane((int)X, 63 - __builtin_clzll(idxmask));
#ifdef SQOA_QTRACE   // (9: groups with INDEX ops)
                    ph[9] += 1;
#endif
                    commit(val, hv);
                }
                if (T) pxv = (uint32_t)__builtin_amdgcn_readlane((int)vlast, (int)((T - 1u) & 63u));
            }
            RPH(5);
            qk_post(&s_done[p], c + 1, pxv, lane);
        }
        if (SPEC && !bad) {
            const unsigned long long e = __hip_atomic_load(&s_tab[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            my_dq[1].px = pxv;
            my_dq[1].tab[lane] = (uint32_t)e;
            const unsigned long long wmask = __ballot((uint32_t)(e >> 32) > (chunk_first << 10));   // slots my own chunks wrote
            s_smask[0] = 0ull;
            __hip_atomic_fetch_or(&s_smask[0], rmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long rm = __hip_atomic_load(&s_smask[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0) {
                my_dq[1].pad[1] = (uint32_t)rm; my_dq[1].pad[2] = (uint32_t)(rm >> 32);
                my_dq[1].pad[3] = (uint32_t)wmask; my_dq[1].pad[4] = (uint32_t)(wmask >> 32);
                atomicAdd(&my_dq[1].pad[0], 1u);
            }
        }
#ifdef SQOA_QTRACE
        if (lane == 0) { unsigned long long* tr = (unsigned long long*)job.dst; ph[10] = nchunks; for (int k = 0; k < 12; ++k) tr[k] = ph[k]; }
#endif
        if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
        return;
    }

    // ================= the producers =======================================================================================
    const uint32_t p = (uint32_t)w - 1u;
    uint32_t* lds = s_bytes[p];
    uint2* rec = s_rec[p];
    uint32_t stg[QK_NLD];
    auto load_chunk = [&](uint32_t c, uint32_t (&r)[QK_NLD]) {
        const uint32_t d0 = (p0 + c * QK_CHUNK) >> 2;
        if ((d0 + QK_LDS_DW) * 4u <= size) {
            gcptr32 q = (gcptr32)src + d0;
#pragma unroll
            for (int k = 0; k < QK_NLD - 1; ++k) r[k] = q[lane + 64 * k];
            r[QK_NLD - 1] = q[64 * (QK_NLD - 1) + (lane & 7)];
        } else {
#pragma unroll
            for (int k = 0; k < QK_NLD; ++k) {
                const uint32_t d = d0 + lane + 64 * k;
                r[k] = (k < QK_NLD - 1 || lane < QK_LDS_DW - 64 * (QK_NLD - 1)) && d * 4u < size ? ((gcptr32)src)[d] : 0u;
            }
        }
    };

    auto put = [&](uint32_t P_, uint32_t v) {   // one pixel (seqoia.h:790-805, colour)
        if (dw_out) ((gptr32)out)[P_] = v;
        else {
            gptr8 o = out + (size_t)P_ * (unsigned)oc;
            o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
            if (oc == 4) o[3] = (uint8_t)(v >> 24);
        }
    };

    // The pixels of the chunk whose records sit in my buffer, once the resolver is done with them: literal ops lane-local,
    // runs by the whole wave; the last chunk repeats the final pixel to the end of the image (seqoia.h:726).
    uint32_t held = ~0u, held_T = 0, held_pos = 0, held_out = 0, held_ctot = 0;   // the chunk in my buffer, its ops, its first pixel, the pixel behind it
    auto flush_held = [&]() -> bool {
        if (held == ~0u) return true;
        uint32_t pxv = 0;
        if (!qk_wait(&s_done[p], held + 1, pxv, bad)) return false;
        uint32_t P0 = held_pos;
#ifdef SQOA_DBG_Q_NOOUT
        P0 = N;
#endif
        if constexpr (BAL && SQOA_QDEC_LDS_STORES != 0) {
            // The common chunk (dword pixels, all of them inside the image, no more than the buffer holds): as decode_bal_kernel stores --
            // my own q records back into registers (the lane that made them takes them: their ranks, hence their pixel positions, follow
            // from T alone), ONE scan of the lanes' pixel counts where the loop below runs one per 64 records, every op's pixel into the
            // record buffer -- free once the records are in registers -- at its position in the chunk, RUNs filled in by the wave, and out
            // as 16-byte pieces: 4 KB a store instruction instead of 64 scattered dwords.
            constexpr uint32_t CAP = 2u * (uint32_t)QK_CHUNK - 8u, DUMP = 4u * (CAP + 4u);
            if (dw_out && held_T != 0u && held_ctot <= CAP && P0 + held_ctot <= N) {
                const uint32_t hq = bcast_first((held_T + 63u) >> 6), hk = bcast_first(held_T - 64u * (hq - 1u));
                const bool hfirst = (uint32_t)lane < hk;
                const uint32_t h0 = (uint32_t)lane * (hq - 1u) + ((uint32_t)lane < hk ? (uint32_t)lane : hk) - (hfirst ? 0u : 1u);
                uint32_t hv[QK_LB], hc[QK_LB], lc = 0, longest = 0;
#pragma unroll
                for (int i = 0; i < QK_LB; ++i) {
                    if ((uint32_t)i >= hq) break;
                    const uint2 r = rec[h0 + i];
                    hv[i] = r.x;
                    hc[i] = i == 0 && !hfirst ? 0u : r.y >> 9;
                    lc += hc[i];
                    longest = hc[i] > longest ? hc[i] : longest;
                }
                const uint32_t pinc = wave_incl_scan(lc);
                const uint32_t pix_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2*)rec;
                uint32_t* pix = reinterpret_cast<uint32_t*>(rec);
                const unsigned long long exec_in = exec_saved();
                uint32_t a = pix_addr + ((pinc - lc) << 2);
                asm volatile("" ::: "memory");   // (every record is in a register -- or on its way, the LDS queue keeps a wave's order -- before a pixel takes its place)
#pragma unroll
                for (int i = 0; i < QK_LB; ++i) {
                    if ((uint32_t)i >= hq) break;
                    const uint32_t aa = i == 0 && !hfirst ? pix_addr + DUMP : a;   // (no op in my slot 0: to the dump words)
                    asm volatile(
                        "ds_write_b32 %[a], %[v]\n\t"
                        "v_cmpx_lt_u32_e32 vcc, 1, %[c]\n\t"             // a RUN of two or more
                        "ds_write_b32 %[a], %[v] offset:4\n\t"
                        "s_mov_b64 exec, %[ex]\n\t"
                        "s_nop 4"
                        :: [c] "v"(hc[i]), [a] "v"(aa), [v] "v"(hv[i]), [ex] "s"(exec_in) : "memory", "vcc");
                    a += hc[i] << 2;
                }
                if (__ballot(longest > 2u) != 0ull) {   // seqoia.h:773-775: what a RUN covers behind its first two pixels, by the whole wave
                    uint32_t rel = pinc - lc;
#pragma unroll
                    for (int i = 0; i < QK_LB; ++i) {
                        if ((uint32_t)i >= hq) break;
                        unsigned long long runs = __ballot(hc[i] > 2u);
                        while (runs) {
                            const int l = __builtin_ctzll(runs);
                            runs &= runs - 1ull;
                            const uint32_t rp = bcast_lane(rel, l) + 2u, rc = bcast_lane(hc[i], l) - 2u, rvv = bcast_lane(hv[i], l);
                            for (uint32_t t = (uint32_t)lane; t < rc; t += 64u) pix[rp + t] = rvv;
                        }
                        rel += hc[i];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                {
                    typedef u32x4 u32x4_una __attribute__((aligned(4)));
                    typedef __attribute__((address_space(1))) u32x4_una* gptr128u;
                    const uint32_t n = held_ctot;
                    for (uint32_t x0 = 0; x0 < n; x0 += 256u) {
                        const uint32_t x = x0 + 4u * (uint32_t)lane;
                        gptr32 o = (gptr32)out + P0 + x;
                        if (x + 4u <= n) *(gptr128u)o = *reinterpret_cast<const u32x4*>(pix + x);
                        else if (x < n) {                                              // the last one to three pixels
                            o[0] = pix[x];
                            if (x + 1u < n) o[1] = pix[x + 1];
                            if (x + 2u < n) o[2] = pix[x + 2];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                P0 = N;   // (nothing left for the loop below)
            }
        }
        for (uint32_t gb = 0; gb < held_T && P0 < N; gb += 64) {
            const bool live = gb + (uint32_t)lane < held_T;
            const uint2 r = live ? rec[gb + lane] : uint2{0u, 0u};
            const uint32_t cnt = r.y >> 9, v = r.x;
            const uint32_t pinc = wave_incl_scan(cnt);
            const uint32_t Pp = P0 + pinc - cnt;
            const uint32_t gtot = bcast_lane(pinc, 63);
#ifdef SQOA_QDEC_ASM_STORES   // A/B builds: decode_wave_kernel's hand-written store path carried over -- headline batch 9.45 -> 9.68 ms, icons 2.20 -> 2.23 ms: off
            if (dw_out && P0 + gtot <= N) {
                // the common group -- dword pixels, nothing clipped: the two predicates narrow EXEC directly (v_cmpx), the stores take
                // the image base from an SGPR pair and a 32-bit byte offset, EXEC is put back once
                const unsigned long long obase = job.dst, exec_in = exec_saved();
                const uint32_t off = Pp << 2, c1 = cnt - 1u;
                asm volatile(
                    "v_cmpx_gt_u32_e32 vcc, 2, %[c1]\n\t"           // cnt is 1 or 2
                    "global_store_dword %[off], %[v], %[base]\n\t"
                    "v_cmpx_eq_u32_e32 vcc, 1, %[c1]\n\t"           // cnt == 2: an op and the one-pixel run behind it
                    "global_store_dword %[off], %[v], %[base] offset:4\n\t"
                    "s_mov_b64 exec, %[ex]\n\t"
                    "s_nop 4"                                          // (a VALU write of EXEC wants 5 wait states in front of a DPP instruction)
                    :: [c1] "v"(c1), [off] "v"(off), [v] "v"(v), [base] "s"(obase), [ex] "s"(exec_in) : "memory", "vcc");
            } else
#endif
            {
            if (cnt >= 1u && cnt <= 2u && Pp < N) put(Pp, v);
            if (cnt == 2u && Pp + 1u < N) put(Pp + 1u, v);
            }
            unsigned long long runs = __ballot(cnt > 2u && Pp < N);
            while (runs) {  // seqoia.h:723-725
                const int l = __builtin_ctzll(runs);
                runs &= runs - 1ull;
                const uint32_t rp = bcast_lane(Pp, l), rvv = bcast_lane(v, l);
                uint32_t rc = bcast_lane(cnt, l);
                if (rp + rc > N) rc = N - rp;
                for (uint32_t k = lane; k < rc; k += 64) put(rp + k, rvv);
            }
            P0 += gtot;
        }
        if (held == nchunks - 1) {  // stream exhausted: the last pixel repeats (seqoia.h:726)
            for (uint32_t k = held_out + lane; k < N; k += 64) put(k, pxv);
#ifdef SQOA_QDEC_HWDUMP   // diagnostic builds: HW_ID of the resolver wave instead of the length (tools/qoi_resolver_simds.py)
            if (!SPEC && lane == 0) res->len = (int)s_hwid;
#else
            if (!SPEC && lane == 0) res->len = (int)(N * (uint32_t)oc);
#endif
        }
        held = ~0u;
        return true;
    };

    uint32_t c = c_begin + (p + (uint32_t)P - c_begin % (uint32_t)P) % (uint32_t)P;   // my first chunk: c mod P == p
    if (c < c_end) load_chunk(c, stg);
    for (; c < c_end; c += P) {
        const bool last_chunk = c == nchunks - 1;
        // ---- stage the chunk's bytes (+ look-ahead); prefetch the wave's next chunk ----
        QP(0);
#pragma unroll
        for (int k = 0; k < QK_NLD - 1; ++k) lds[lane + 64 * k] = stg[k];
        if (lane < QK_LDS_DW - 64 * (QK_NLD - 1)) lds[lane + 64 * (QK_NLD - 1)] = stg[QK_NLD - 1];
        if (c + P < c_end) load_chunk(c + P, stg);
        const uint32_t gs = p0 + c * QK_CHUNK;
        const uint32_t my0 = gs + QK_LB * lane;
        const uint32_t nlive = my0 >= chunks_len ? 0u : (chunks_len - my0 > (uint32_t)QK_LB ? (uint32_t)QK_LB : chunks_len - my0);
        const uint32_t loff = (gs & 3u) + QK_LB * lane;

        // ---- parse: the state in front of every lane's first byte (A, D1..D4; X past the chunk area) ----
        QP(1);
        const unsigned long long own = lds_read8(lds, loff);
        // (the table reads together, waited for once: the empty asm wants all eight in registers)
        uint32_t pflo[QK_LB];
#pragma unroll
        for (int k = 0; k < QK_LB; ++k) {
            const uint32_t b = (uint32_t)(own >> (8 * k)) & 0xffu;
            pflo[k] = (uint32_t)k < nlive ? s_ftab[b] : (ST_X | ST_X << 8 | 0x02000000u);
        }
        if constexpr (BAL)
        asm volatile("" : "+v"(pflo[0]), "+v"(pflo[1]), "+v"(pflo[2]), "+v"(pflo[3]), "+v"(pflo[4]), "+v"(pflo[5]), "+v"(pflo[6]), "+v"(pflo[7]));
        Fn F{ID_LO, ID_HI};
#pragma unroll
        for (int k = 0; k < QK_LB; ++k) F = compose(Fn{pflo[k], 0x07060403u}, F);
        const Fn Fi = wave_compose_scan(F);
        const Fn G{bcast_lane(Fi.lo, 63), bcast_lane(Fi.hi, 63)};
        const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
        const uint32_t g0 = G.lo & 0xffu;
        const bool g_const = G.lo == g0 * 0x01010101u && (G.hi & 0xffffu) == g0 * 0x0101u;
        // (an exit state that does not depend on the entry state is posted before the entry state is known; the two-deep slots are
        // safe because the position chain below holds a producer back: it cannot finish chunk c + P before chunk c + 1 was entered)
        unsigned long long* my_parse = &s_parse[(c / P) & 1][p];
        if (g_const && !last_chunk) qk_post(my_parse, c + 1, g0, lane);
        uint32_t st_in = seed_st;   // (the stream's start: A; SPEC: what the scan of pass 1 found in front of the warm-up's first chunk)
        if (c > c_begin && !qk_wait(&s_parse[((c - 1) / P) & 1][(c - 1) % P], c, st_in, bad)) break;
        if (!g_const && !last_chunk) qk_post(my_parse, c + 1, apply(G, st_in), lane);
        const uint32_t lst = apply_perm(Fx, st_in);

        uint32_t rv[QK_LB], rm[QK_LB];   // record value; record word: 1 = INDEX | absolute channels (2 = rgb, 6 = rgba) | slot << 3 | pixels << 9
        uint32_t iters = 0, obase = 0, ctot = 0, T = 0, closed = 0;
        Xf Tl{0u, 0u};                   // my ops so far, since my last INDEX op (or my first op)
        typedef const __attribute__((address_space(3))) uint32_t* lcptr32;
        typedef const __attribute__((address_space(3))) u32x4* lcptr128;
        const uint32_t qtab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) u32x4*)s_qtab;
        if constexpr (BAL) {
        // ---- the chunk's ops dealt out evenly over the lanes (round 5: the balanced walk of sqoa_decode_bal.hip) ----
        // A lane that walks the ops that start in ITS 8 bytes runs as many iterations as the busiest lane has ops -- eight wherever a chunk holds
        // a stretch of one-byte ops (DIFF, INDEX, RUN: every gradient), for the four or so ops a lane has on average.  Here: (1) the parser
        // state in front of every byte of mine (the functions applied once more, one v_perm_b32 a byte), "an op starts here" = that state is
        // A on a live byte, the flags gathered into a mask, their ranks by the wave scan that counted the ops anyway; (2) the starts' byte
        // offsets to a list by rank -- in my record buffer, which is free: the pixels of my previous chunk go out FIRST now (the resolver is
        // six chunks of the other producers away from needing this one's records); (3) q = ceil(T / 64) ops per lane, q - 1 for the lanes
        // from k = T - 64 (q - 1) on, whose slot 0 stays empty: lane j's slot i holds op j (q - 1) + min(j, k) + i - (slot 0 empty), and that
        // rank IS the record's place in stream order.
        QP(2);
        uint32_t M;   // bit k: an op starts at my byte k
        {
            uint32_t st = lst, o[QK_LB - 1];
#ifndef SQOA_QDEC_FLO_REG   // 1: the parse's function entries stay in registers for this pass; 0: looked up in s_ftab again (round 5)
#define SQOA_QDEC_FLO_REG 1
#endif
            // (as in sqoa_decode_bal.hip: the second look-up was seven data-dependent ds_read_b32 a lane into an LDS pipe both sides of this
            // workgroup wait for; 60 VGPRs, no spill, and seven SGPR spills fewer.  Same call, headline batch: 7.189 -> 6.886 ms, class photo
            // 5.49 -> 5.34; the entries' low bytes -- an entry is v | v << 8 | 0x02000000 -- packed into two registers and unpacked by one
            // v_perm_b32 each: 7.036.  The four-wave instance (many short streams: icons 4096 x 512^2, 22 INDEX ops a group, the resolver's
            // business) measured 2.037 -> 2.063 ms with it and keeps the look-up.)
            constexpr bool FLO_REG = SQOA_QDEC_FLO_REG && W == 8;
            if constexpr (FLO_REG) {
#pragma unroll
                for (int k = 0; k < QK_LB - 1; ++k) asm volatile("" : "+v"(pflo[k]));
            }
#pragma unroll
            for (int k = 0; k < QK_LB - 1; ++k) {
                const uint32_t b = (uint32_t)(own >> (8 * k)) & 0xffu;
                uint32_t flo;
                if constexpr (FLO_REG) flo = pflo[k];
                else flo = (uint32_t)k < nlive ? s_ftab[b] : (ST_X | ST_X << 8 | 0x02000000u);
                st = __builtin_amdgcn_perm(0x07060403u, flo, st);   // (byte 0: the state behind byte k)
                o[k] = st;
            }
            const uint32_t p01 = __builtin_amdgcn_perm(o[0], lst, 0x0c0c0400u), p23 = __builtin_amdgcn_perm(o[2], o[1], 0x0c0c0400u);
            const uint32_t p45 = __builtin_amdgcn_perm(o[4], o[3], 0x0c0c0400u), p67 = __builtin_amdgcn_perm(o[6], o[5], 0x0c0c0400u);
            const uint32_t w0 = __builtin_amdgcn_perm(p23, p01, 0x05040100u), w1 = __builtin_amdgcn_perm(p67, p45, 0x05040100u);   // the states in front of my bytes 0..3, 4..7
            const uint32_t f0 = ~(w0 + 0x07070707u) & 0x08080808u, f1 = ~(w1 + 0x07070707u) & 0x08080808u;                          // 0x08 where the state is A (0)
            M = (__builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false) >> 3) & ((1u << nlive) - 1u);
        }
        const uint32_t nops = (uint32_t)__builtin_popcount(M);
        const uint32_t oinc = wave_incl_scan(nops);
        T = bcast_lane(oinc, 63);

        // ---- my buffer still holds my previous chunk: its pixels go out now ----
        QP(3);
        if (!flush_held()) break;
        uint32_t* list = reinterpret_cast<uint32_t*>(rec) + 8;   // (one entry in front for the lanes whose slot 0 is empty, room behind for the last lanes' reads)
        {
            uint32_t m = M, t;
            const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)list + 4u * (oinc - nops);
            const uint32_t data0 = (uint32_t)(QK_LB * lane);
            const unsigned long long exec_in = exec_saved();
#define SQOA_QLIST_STEP(OFF)                        \
            "v_cmpx_ne_u32_e32 vcc, 0, %[m]\n\t"      \
            "v_ffbl_b32_e32 %[t], %[m]\n\t"           \
            "v_add_u32_e32 %[t], %[t], %[d0]\n\t"     \
            "ds_write_b32 %[a], %[t] offset:" #OFF "\n\t" \
            "v_add_u32_e32 %[t], -1, %[m]\n\t"        \
            "v_and_b32_e32 %[m], %[t], %[m]\n\t"
            asm volatile(
                SQOA_QLIST_STEP(0) SQOA_QLIST_STEP(4) SQOA_QLIST_STEP(8) SQOA_QLIST_STEP(12)
                SQOA_QLIST_STEP(16) SQOA_QLIST_STEP(20) SQOA_QLIST_STEP(24) SQOA_QLIST_STEP(28)
                "s_mov_b64 exec, %[ex]\n\t"
                "s_nop 4"
                : [m] "+v"(m), [t] "=&v"(t) : [a] "v"(laddr), [d0] "v"(data0), [ex] "s"(exec_in) : "memory", "vcc");
#undef SQOA_QLIST_STEP
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t q = bcast_first((T + 63u) >> 6);
        const uint32_t kq = bcast_first(T - 64u * (q - 1u));
        const bool first_slot = (uint32_t)lane < kq;
        const uint32_t r_slot0 = q ? (uint32_t)lane * (q - 1u) + ((uint32_t)lane < kq ? (uint32_t)lane : kq) - (first_slot ? 0u : 1u) : 0u;
        // (rv[]: until the op is decoded, its start)
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) rv[i] = list[r_slot0 + i];
        __builtin_amdgcn_wave_barrier();
        uint32_t cl = 0;
        uint32_t seen = 0;               // (closed) bit i: an INDEX op of mine sits at or in front of my slot i (its record is final as it is)
        const uint32_t cb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (gs & 3u);
        iters = q;
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) {
            if ((uint32_t)i >= q) break;
            rm[i] = 0u;
            if (i != 0 || first_slot) {
                const uint32_t a = cb_addr + rv[i];
                lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
                const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
                const u32x4 e = *(lcptr128)(uintptr_t)(qtab_addr + (lo & 0xffu) * 16u);
                const uint32_t nib = (((lo >> 12) & 0xfu) | ((lo << 8) & 0xf0000u)) & e.w;     // LUMA: dr_dg, db_dg (seqoia.h:765-768)
                const uint32_t rel = ((e.z + nib) & 0x00ff00ffu) | (e.w & 0xff00u);
                const uint32_t base = __builtin_amdgcn_alignbit(hi, lo, 8);                     // RGB / RGBA: the bytes behind the op byte
                const uint32_t v = (rel & ~e.y) | (base & e.y);
                const uint32_t xm = sra31(e.x << 15);                                           // ~0: an INDEX op (bit 16 of the table word)
                Tl.v = add8(Tl.v & ~e.y, v) & ~xm;
                Tl.abs = (Tl.abs | e.y) & ~xm;
                seen |= xm;
                closed |= seen & (1u << i);
                rv[i] = Tl.v;
                rm[i] = (e.x >> 16) | ((Tl.abs >> 22) & 6u);
                cl += e.x >> 25;
            }
        }
        obase = r_slot0;
        ctot = bcast_lane(wave_incl_scan(cl), 63);   // pixels this chunk's ops produce
        } else {
        // ---- walk: my ops, in order (seqoia.h:740-775 with qoi_compat) ----
        QP(2);
        uint32_t pos = first_op_offset<QK_LB>(lst);
        // Between two INDEX ops pixel values do not depend on the index: every op's record is its pixel RELATIVE to the INDEX op
        // (or the chunk start) its segment begins at.  Inside the lane the ops' transforms are composed as they are decoded; what
        // the lanes below contribute comes from one segmented scan over the lanes' totals further down.
        uint32_t nops = 0, cl = 0;
        iters = QK_LB;
        uint32_t seen = 0;               // (closed) bit i: an INDEX op of mine sits at or in front of my op i (its record is final as it is)
        const uint32_t lane_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + loff;
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) {
            if (__ballot(pos < nlive) == 0ull) { iters = i; break; }
#if defined(SQOA_QDEC_EXEC_WALK) && SQOA_QDEC_EXEC_WALK
            // (A/B builds: lanes that have run out of ops sit the iteration out under EXEC, as in decode_wave_kernel -- which runs at the
            // board's power limit and gained 4 % from it.  This kernel waits for its resolver, not for its power budget, and the EXEC
            // bookkeeping is scalar instructions on a scalar pipe that is full: 9.45 -> 9.66 ms.  Off.)
            asm volatile("" : "=v"(rv[i]));   // (read later only where rm says the slot holds an op)
            rm[i] = 0u;
            if (pos >= nlive) continue;
            const uint32_t a = lane_addr + pos;
            lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
            const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
            const u32x4 e = *(lcptr128)(uintptr_t)(qtab_addr + (lo & 0xffu) * 16u);
#else
            const uint32_t m_act = sra31(pos - nlive);                      // ~0 while I still have an op (pos < nlive <= 8)
            const uint32_t a = lane_addr + ((pos & m_act) | ((uint32_t)QK_LB & ~m_act));
            lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
            const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
            const uint32_t idx = (((lo & 0xffu) - 256u) & m_act) + 256u;    // lanes out of ops decode the "no op" entry
            const u32x4 e = *(lcptr128)(uintptr_t)(qtab_addr + idx * 16u);
#endif
            const uint32_t nib = (((lo >> 12) & 0xfu) | ((lo << 8) & 0xf0000u)) & e.w;     // LUMA: dr_dg, db_dg (seqoia.h:765-768)
            const uint32_t rel = ((e.z + nib) & 0x00ff00ffu) | (e.w & 0xff00u);
            const uint32_t base = __builtin_amdgcn_alignbit(hi, lo, 8);                     // RGB / RGBA: the bytes behind the op byte
            const uint32_t v = (rel & ~e.y) | (base & e.y);
            const uint32_t xm = sra31(e.x << 15);                                           // ~0: an INDEX op (bit 16 of the table word)
            Tl.v = add8(Tl.v & ~e.y, v) & ~xm;                                              // (the "no op" entry composes the identity)
            Tl.abs = (Tl.abs | e.y) & ~xm;
            seen |= xm;
            closed |= seen & (1u << i);
            rv[i] = Tl.v;
            rm[i] = (e.x >> 16) | ((Tl.abs >> 22) & 6u);
            nops += (e.x >> 3) & 1u;
            cl += e.x >> 25;
            pos += e.x & 7u;
        }

        iters = bcast_first(iters);  // wave-uniform by construction: scalar slot tests below
        const uint32_t oinc = wave_incl_scan(nops);
        T = bcast_lane(oinc, 63);
        obase = oinc - nops;
        ctot = bcast_lane(wave_incl_scan(cl), 63);   // pixels this chunk's ops produce

        // ---- my buffer still holds my previous chunk: its pixels go out now (the resolver has long been through it) ----
        QP(3);
        if (!flush_held()) break;
        }

        // ---- what enters my lane: the lanes below me back to the nearest one that holds an INDEX op (its ops behind that op) ----
        QP(4);
        Xf E;
        {
            Xf S = Tl;
            uint32_t hf = closed ? ~0u : 0u;
            S = qk_seg_step<DPP_ROW_SHR + 1, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 2, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 4, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_SHR + 8, 0xf>(S, hf);
            S = qk_seg_step<DPP_ROW_BCAST15, 0xa>(S, hf);
            S = qk_seg_step<DPP_ROW_BCAST31, 0xc>(S, hf);
            E = Xf{shift_up1(S.v, 0u), shift_up1(S.abs, 0u)};
        }
        // ---- records in stream order: my ops in front of my first INDEX op continue the segment that enters the lane ----
        QP(5);
#pragma unroll
        for (int i = 0; i < QK_LB; ++i) {
            if ((uint32_t)i >= iters) break;
            const uint32_t open = ((closed >> i) & 1u) - 1u;                       // ~0: no INDEX op of mine at or in front of op i
            const uint32_t m = rm[i];
            const uint32_t ab = (m & 4u) ? 0xffffffffu : (m & 2u) ? 0x00ffffffu : 0u;
            const uint32_t sv = add8(E.v & open & ~ab, rv[i]);
            const uint32_t sa = (E.abs & open) | ab;
            if (m >> 9) rec[obase + i] = uint2{sv, (m & ~6u) | ((sa >> 22) & 6u)};
        }

        // ---- pixel position: a plain sum, handed on at once ----
        QP(6);
        uint32_t pos_in = 0;
        if (c > c_begin && !qk_wait(&s_pos[(c - 1) % P], c, pos_in, bad)) break;
        if (SPEC && c == chunk_first) pos_in = seed_pos;   // (the warm-up's positions are dummies: nothing of it is stored)
        const uint32_t pos_out = pos_in + ctot > N ? N : pos_in + ctot;
        if (!last_chunk) qk_post(&s_pos[p], c + 1, pos_out, lane);

        // ---- over to the resolver (release: the records are in LDS when the tag is) ----
        QP(7);
        qk_post(&s_ready[p], c + 1, T | (pos_in < N ? 0u : 1u << 31), lane);
        held = c; held_T = T; held_pos = SPEC && c < chunk_first ? N : pos_in; held_out = pos_out; held_ctot = ctot;
    }
    if (!bad) flush_held();
    if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
    if (SPEC && !bad && lane == 0) atomicAdd(&my_dq[1].pad[0], 1u);
}

}  // namespace sqoa

namespace sqoa {
// The check behind the speculative pass: one workgroup per image; thread k compares what workgroup k guessed for the state at its first
// chunk with what workgroup k - 1 arrived at behind its last one.  All equal: the image is decoded (status OK, len); else it stays
// SLOWPATH and the one-workgroup decoder, launched next, decodes it from the start.
#ifndef SQOA_DQ_MAX_SPAN
#define SQOA_DQ_MAX_SPAN 64u
#endif

// The check behind the speculative pass.
// Workgroup j's own chunks are decoded right if the pixel it started them from is the true one and every index entry they READ
// before writing it (R_j) is the true one.  "True" is only known along the chain: V_j = the entries of workgroup j's END state
// known to be true = the ones its own chunks wrote (W_j) plus the ones it guessed like the workgroup in front left them, where
// those were known to be true:  V_0 = all,  good_j = eq_j & V_{j-1},  need R_j subset of good_j,  V_j = W_j | good_j.
// (Entries nobody reads may differ: a slot last written far in front of the warm-up is simply stale in the guess.)
//
// dq_compare_kernel, one wave per workgroup of the speculative pass: eq_j (64 compares = one ballot), "the pixels agree" and
// "every wave came through", left in the pad words of the workgroup's start row.
SQOA_DEV uint32_t dq_nchunks(const sqoa_hip_job& job) {   // the stream's chunk count, as decode_qoi_kernel counts it
    const uint32_t area = job.src_size > (uint32_t)kHeader + 8u ? job.src_size - (uint32_t)kHeader - 8u : 0u;
    return area ? (area + QK_CHUNK - 1) / QK_CHUNK : 1u;
}
template <int W>
__global__ __launch_bounds__(512) void dq_compare_kernel(const sqoa_hip_job* __restrict__ jobs, const sqoa_hip_result* __restrict__ results,
                                                         const dp_image* __restrict__ dp_images, const dp_wg* __restrict__ dp_wgs, uint32_t n_wgs,
                                                         const uint32_t* __restrict__ dp_flags, dq_state* __restrict__ dq, uint32_t dp_per,
                                                         uint32_t span) {
    const uint32_t r = blockIdx.x * 8u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= n_wgs) return;
    const dp_wg me = dp_wgs[r];
    if (dp_flags[me.img] != 3u) return;
    const dp_image im = dp_images[me.img];
    const uint32_t row = me.chunk0 / dp_per;
    if (row % span != 0u || results[im.jid].status != SQOA_HIP_ST_SLOWPATH || me.chunk0 >= dq_nchunks(jobs[im.jid])) return;
    dq_state* a = dq + 2 * (size_t)(im.tab_off + row);
    uint32_t ok = a[1].pad[0] == (uint32_t)W ? 1u : 0u;
    unsigned long long eq = ~0ull;
    if (row) {
        const dq_state* b = dq + 2 * (size_t)(im.tab_off + row - span) + 1;
        eq = __ballot(a[0].tab[lane] == b->tab[lane]);
        if (a[0].px != b->px) ok |= 2u;
    }
    if (lane == 0) { a[0].pad[5] = (uint32_t)eq; a[0].pad[6] = (uint32_t)(eq >> 32); a[0].pad[7] = ok; }
}

// dq_verify_kernel, one workgroup per stream: the chain as a scan.  V_j = W_j | (eq_j & V_{j-1}) is, per bit, one of "0", "1",
// "as before" -- functions that compose: (A, B): v -> A | (B & v), then (A', B') gives (A' | (B' & A), B' & B).  Every thread
// composes a contiguous run of workgroups, the run totals are scanned through LDS, the runs are walked again for the subset test.
__global__ __launch_bounds__(1024) void dq_verify_kernel(const sqoa_hip_job* __restrict__ jobs, sqoa_hip_result* __restrict__ results,
                                                         const dp_image* __restrict__ dp_images, const uint32_t* __restrict__ dp_flags,
                                                         const dq_state* __restrict__ dq, uint32_t dp_per, uint32_t span) {
    __shared__ unsigned long long s_a[2][1024], s_b[2][1024];
    __shared__ uint32_t s_bad;
    const dp_image im = dp_images[blockIdx.x];
    if (dp_flags[blockIdx.x] != 3u || results[im.jid].status != SQOA_HIP_ST_SLOWPATH) return;   // (block-uniform; nothing here writes it before the end)
    const sqoa_hip_job job = jobs[im.jid];
    const uint32_t nrows = (dq_nchunks(job) + dp_per - 1) / dp_per, nj = (nrows + span - 1) / span;
    const uint32_t t = threadIdx.x, per = (nj + 1023u) / 1024u;
    const uint32_t j0 = t * per < nj ? t * per : nj, j1 = j0 + per < nj ? j0 + per : nj;
    const dq_state* q = dq + 2 * (size_t)im.tab_off;
    if (t == 0) s_bad = 0;
#ifdef SQOA_DQ_DEBUG
    if (t == 0) results[im.jid].len = 0x7fffffff;
#endif
    unsigned long long A = 0ull, B = ~0ull;
    for (uint32_t j = j0; j < j1; ++j) {
        const dq_state* e = q + 2 * (size_t)j * span;
        const unsigned long long eq = (unsigned long long)e[0].pad[6] << 32 | e[0].pad[5], wm = (unsigned long long)e[1].pad[4] << 32 | e[1].pad[3];
        A = wm | (eq & A); B = eq & B;
    }
    s_a[0][t] = A; s_b[0][t] = B;
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < 1024u; d <<= 1) {   // inclusive scan of the run totals
        unsigned long long a2 = s_a[cur][t], b2 = s_b[cur][t];
        if (t >= d) { const unsigned long long a1 = s_a[cur][t - d], b1 = s_b[cur][t - d]; a2 = a2 | (b2 & a1); b2 = b2 & b1; }
        s_a[cur ^ 1][t] = a2; s_b[cur ^ 1][t] = b2;
        cur ^= 1;
        __syncthreads();
    }
    // V in front of my run: the prefix applied to "all" (workgroup 0 starts from the stream's true start state)
    unsigned long long v = t ? s_a[cur][t - 1] | s_b[cur][t - 1] : ~0ull;
    uint32_t bad = 0;
    [[maybe_unused]] uint32_t why = 0;
    for (uint32_t j = j0; j < j1; ++j) {
        const dq_state* e = q + 2 * (size_t)j * span;
        const unsigned long long eq = (unsigned long long)e[0].pad[6] << 32 | e[0].pad[5], wm = (unsigned long long)e[1].pad[4] << 32 | e[1].pad[3];
        const unsigned long long rm = (unsigned long long)e[1].pad[2] << 32 | e[1].pad[1];
        const unsigned long long good = eq & v;
        const uint32_t ok = e[0].pad[7];
        if (ok != 1u || (rm & ~good)) {
#ifdef SQOA_DQ_DEBUG
            if (!bad) why = j * 16u + ((ok & 1u) ? 0u : 1u) + ((rm & ~good) ? 2u : 0u) + ((rm & ~eq) ? 4u : 0u) + ((ok & 2u) ? 8u : 0u);
#endif
            bad = 1;
        }
        v = wm | good;
    }
    if (bad) {
#ifdef SQOA_DQ_DEBUG   // diagnostic builds: the first workgroup that fails and why, instead of the fall-back
        atomicMin((uint32_t*)&results[im.jid].len, why);
#endif
        s_bad = 1;
    }
    __syncthreads();
#ifdef SQOA_DQ_DEBUG
    if (t == 0 && s_bad) results[im.jid].status = SQOA_HIP_ST_BADARG;
#endif
#ifdef SQOA_DQ_ACCEPT_ALL   // mutation builds (tests/README: the stale-entry tests must FAIL with this)
    if (t == 0) {
#else
    if (t == 0 && !s_bad) {
#endif
        gcptr8 src = (gcptr8)job.src;
        const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
        const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
        const uint32_t hc = src[12];
        const uint32_t oc = job.desc.channels ? job.desc.channels : 3u + ((hc & 1u) ? 0u : 1u);
        results[im.jid].len = (int)(hw * hh * oc);
        results[im.jid].status = SQOA_HIP_ST_OK;
    }
}
}  // namespace sqoa

// The speculative chip-wide pass over the QOI-compat streams among a plan's few large streams, and the check behind it.
extern "C" hipError_t sqoa_launch_decode_qoi_spec(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const void* d_images, int n_images,
                                                  const void* d_wgs, int n_wgs, const void* d_tab, const void* d_flags, void* d_qstate, unsigned per,
                                                  int cus, hipStream_t stream) {
    if (n_images <= 0 || n_wgs <= 0 || !d_qstate) return hipSuccess;
    using namespace sqoa;
    // A workgroup decodes `span` rows of the tables behind one row of warm-up: as few rows as fill the chip's workgroup slots once
    // (the warm-up is work thrown away: at one row each, half of everything).
    if (cus <= 0) cus = 256;   // (the plan's device: sqoa_dp_tables.cus)
    const unsigned slots = (unsigned)cus * 4u;
    const unsigned span = ((unsigned)n_wgs + slots - 1) / slots;
    // A guess that fails (content that lives on INDEX ops: entries written once, read for ever) has cost this pass on top of the
    // one-workgroup decode that then does the work.  1024 x 1024 images, tools/time_single_qoi_decode.py: one 0.094 against
    // 8.7 ms, 16: 0.62 / 9.3, 64: 2.0 / 9.4, 256: 6.6 / 9.5 ms -- from SQOA_DQ_MAX_SPAN rows per workgroup on there is little to
    // win and as much to lose: those plans go straight to the one-workgroup decoder.
    if (span > SQOA_DQ_MAX_SPAN) return hipSuccess;
    hipLaunchKernelGGL((decode_qoi_kernel<SQOA_QDEC_W, true>), dim3(n_wgs), dim3(SQOA_QDEC_W * 64), 0, stream, d_jobs, (const int*)nullptr, d_results,
                       (const dp_image*)d_images, (const dp_wg*)d_wgs, (const dp_chunk_row*)d_tab, (const uint32_t*)d_flags, (dq_state*)d_qstate, per, span);
    hipLaunchKernelGGL(dq_compare_kernel<SQOA_QDEC_W>, dim3((n_wgs + 7) / 8), dim3(512), 0, stream, d_jobs, (const sqoa_hip_result*)d_results,
                       (const dp_image*)d_images, (const dp_wg*)d_wgs, (uint32_t)n_wgs, (const uint32_t*)d_flags, (dq_state*)d_qstate, per, span);
    hipLaunchKernelGGL(dq_verify_kernel, dim3(n_images), dim3(1024), 0, stream, d_jobs, d_results, (const dp_image*)d_images, (const uint32_t*)d_flags,
                       (const dq_state*)d_qstate, per, span);
    return hipGetLastError();
}
extern "C" size_t sqoa_dq_state_bytes(void) { return 2 * sizeof(sqoa::dq_state); }   // per workgroup row of the tables

// `small`: the plan's streams are short (many more images than the chip holds workgroups of the large shape)
// (The producers' walk balanced over the lanes for every plan: on RUN-heavy streams, whose chunks hold few ops, it is as fast as the walk
// over a lane's own bytes -- 4096 icons of 512 x 512: 2.17 against 2.18 ms, same call -- and on the headline batch 8.18 -> 7.75 ms.)
extern "C" int sqoa_lds_order_fb(void);   // sqoa_selftest.hip: the order-free kernels on the current device?
extern "C" hipError_t sqoa_launch_decode_qoi(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    constexpr bool B = SQOA_QDEC_BAL != 0;
    if (sqoa_lds_order_fb()) {   // (a device that has not passed the probe of the LDS's lane order, or SQOA_HIP_LDS_ORDER=0: the stamped ds_max_u64 update)
        if (small) hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W_SMALL, false, B, false>), dim3(n), dim3(SQOA_QDEC_W_SMALL * 64), 0, stream, d_jobs, d_ids, d_results);
        else hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W, false, B, false>), dim3(n), dim3(SQOA_QDEC_W * 64), 0, stream, d_jobs, d_ids, d_results);
    } else {
        if (small) hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W_SMALL, false, B, SQOA_QDEC_TABW != 0>), dim3(n), dim3(SQOA_QDEC_W_SMALL * 64), 0, stream, d_jobs, d_ids, d_results);
        else hipLaunchKernelGGL((sqoa::decode_qoi_kernel<SQOA_QDEC_W, false, B, SQOA_QDEC_TABW != 0>), dim3(n), dim3(SQOA_QDEC_W * 64), 0, stream, d_jobs, d_ids, d_results);
    }
    return hipGetLastError();
}
