// sqoa_decode_bal.hip -- the BALANCED wave-parallel SQOA decoder for colour streams (gfx950): one workgroup per stream, the batch shape.
//
// Replaces the serial parse of the reference decoder (seqoia.h:722-806) like decode_wave_kernel (sqoa_decode_wave.hip), whose
// chunks / parser-as-function-scan / transform monoid / LDS chains it keeps.  What changes is WHO decodes an op.  There a lane owns
// 8 or 12 consecutive stream BYTES and walks the ops that start in them, and the walk runs as many iterations as the busiest lane has
// ops: on the headline batch 5.4 iterations per 768-byte chunk for 3.6 ops per lane -- a third of the walk's lane slots (and of the
// store loop's) execute nothing (LUMA-dense stretches set the count, RGBA-dense ones sit idle).  Here a lane owns consecutive OPS:
//
//   1. parse (as there): every byte's transition function over 8 parser states, composed per lane (16 bytes), one DPP scan across
//      the lanes; the chunk's exit state usually does not depend on its entry state and is posted before that is known.
//   2. starts: with the state in front of its first byte a lane applies its 16 functions once more, one v_perm_b32 each, and has the
//      parser state in front of every byte; "an op starts here" is a SWAR test on (state, next state), the 16 flags are gathered into
//      a bit mask by v_dot4_u32_u8, counted, prefix-summed over the wave (the op's RANK in the chunk) and each start's byte offset
//      goes to a list in LDS at its rank.
//   3. walk: the chunk's n ops are dealt out q = ceil(n / 64) consecutive ones per lane -- lane j starts at list[j q] -- and every
//      lane decodes its q ops (table look-ups and mask arithmetic as in decode_wave_kernel): q iterations whatever the content, all
//      lanes busy but the tail.  A chunk of more than 512 op starts (64 lanes x 8 record slots) is more than this kernel takes: the stream
//      is marked SQOA_HIP_ST_SLOWPATH and decode_wave_kernel decodes it in the retry behind this launch (sqoa_launch_decode_standby) --
//      and that is not crafted streams only: a ramp is two-byte LUMA ops throughout, 513 starts in its first chunk with the one-byte RUN
//      in front; a flat region is BIGRUN after BIGRUN, every byte a start.
//   4. scans, chain, stores: as there.
//
// Because ownership no longer follows the bytes, the per-op records (8 slots) bound the OPS per piece, not the bytes per lane: the
// chunk is 1024 bytes (16 per lane), and what a chunk costs whatever its size -- scans, chain hops, scalar control -- is spread
// over a third more bytes than at 768.
//
// A RUN right behind an op repeats the pixel that op produces and is taken in the same walk iteration (pixel pairs, flat stretches);
// that is part of the PARSER here (state G: "a RUN has just been taken along"), so that ranks count iterations.  The one thing the
// parser cannot see without looking ahead is an ALPHA suffix behind such a RUN (seqoia.h:777-783 makes it part of the RUN, which then
// is an op of its own): the encoder never writes one (a RUN is followed by the op of a pixel that differs); a stream that holds one
// makes the parser leave through the absorbing state X and is handed to the exact decoder, like a stream with SQOA_OP_REF.
#include <type_traits>

#include "sqoa_decode_parts.h"

#ifndef SQOA_PRIO_T1  // progress (in 64ths of the image) at which a workgroup drops to wave priority 2, 1, 0
#define SQOA_PRIO_T1 40
#define SQOA_PRIO_T2 54
#define SQOA_PRIO_T3 61
#endif
#ifndef SQOA_BAL_LB
#define SQOA_BAL_LB 16
#endif
#ifndef SQOA_BAL_SPIN_SLEEP
#define SQOA_BAL_SPIN_SLEEP 4
#endif

namespace sqoa {
namespace bal {

// Parser states.  A: op boundary behind an op (an ALPHA suffix belongs to that op, a RUN is taken along with it); N: op boundary behind
// an ALPHA suffix (another 0x60..0x7f byte is a run op of its own, seqoia.h:773-775; a RUN is taken along); G: op boundary behind a
// RUN that was taken along (the next RUN starts an iteration of its own; an ALPHA byte: see the file comment) -- also the stream
// start (no op to belong to); X: past the chunk area, or given up; D1..D4: inside an op, that many bytes to go.
// The three entries of a byte's function that depend on the byte are the low three of its low half; the high half is a constant.
enum : uint32_t { S_A = 0, S_N = 1, S_G = 2, S_X = 3, S_D1 = 4, S_D2 = 5, S_D3 = 6, S_D4 = 7 };
constexpr uint32_t F_HI = 0x06050400u;                      // D1 -> A, D2 -> D1, D3 -> D2, D4 -> D3
constexpr uint32_t F_DEAD = 0x03030303u;                    // a byte past the chunk area: no op starts there (seqoia.h:726)
constexpr uint32_t DEC_ABORT_TAG = 0xffffffffu;

template <int SLEEP>
SQOA_DEV uint32_t chain_wait_rc(unsigned long long* slot, uint32_t tag, uint32_t& val) {
    const uint32_t t = chain_poll<SLEEP>(slot, tag, DEC_ABORT_TAG, val);
    return t == tag ? 0u : t == DEC_ABORT_TAG ? 1u : 2u;
}
SQOA_DEV void post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) { lds_post(slot, tag, val, lane); }
template <int I, int N, class F>
SQOA_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        if (f(std::integral_constant<int, I>{})) static_for<I + 1, N>(f);
    }
}
#ifndef SQOA_BAL_RUN_IN_SFX   // 2: the fused RUN's count from the suffix table (entries four bytes apart) instead of a second look-up in the op table; 0: as before
#define SQOA_BAL_RUN_IN_SFX 2
#endif
#ifndef SQOA_BAL_HL_VGPR   // the SWAR constants of the packed-byte adds in two VGPRs: as literals the compiler parks them in SGPRs, and an SGPR operand makes a fast-class instruction a slow one
#define SQOA_BAL_HL_VGPR 0
#endif
#if SQOA_BAL_HL_VGPR
#define BHL , hl
#define BHLP , HL hl
#else
#define BHL
#define BHLP
#endif
template <int CTRL, int ROWMASK>
SQOA_DEV Xf xf_scan_step(Xf T BHLP) {
    Xf P;  // the transform of the lanes below, identity (all zero) where there is none
    constexpr bool BC = ROWMASK == 0xf;
    P.v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v, CTRL, ROWMASK, 0xf, BC);
    P.abs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.abs, CTRL, ROWMASK, 0xf, BC);
    return xf_then(P, T BHL);
}
SQOA_DEV Xf wave_xf_scan(Xf T BHLP) {
    T = xf_scan_step<DPP_ROW_SHR + 1, 0xf>(T BHL);
    T = xf_scan_step<DPP_ROW_SHR + 2, 0xf>(T BHL);
    T = xf_scan_step<DPP_ROW_SHR + 4, 0xf>(T BHL);
    T = xf_scan_step<DPP_ROW_SHR + 8, 0xf>(T BHL);
    T = xf_scan_step<DPP_ROW_BCAST15, 0xa>(T BHL);
    T = xf_scan_step<DPP_ROW_BCAST31, 0xc>(T BHL);
    return T;
}

// RGB: the instance that also takes jobs to three channels (lists of mixed output formats; the plan's shape bit 4) -- the instance for lists
// the host knows to be four-channel throughout carries none of that code (1.6 VALU + 1.8 scalar instructions per 64 pixels on the headline batch)
template <int W, int LB, bool RGB = false>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void decode_bal_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                                                                    const int* __restrict__ job_ids,
                                                                                                    sqoa_hip_result* __restrict__ results) {
    static_assert(LB == 16, "the start flags are gathered four words of four states at a time");
    constexpr int CHUNK = LB * 64, LDS_DW = CHUNK / 4 + 8;       // stream bytes per chunk; staged dwords: chunk + 8 look-ahead bytes + 3 of misalignment, rounded
    constexpr int R = 8;                                         // record slots per lane = ops per lane and piece
    constexpr int SPIN = SQOA_BAL_SPIN_SLEEP;
    constexpr int CL_BITS = 13;                                  // pixels in front of an op inside its lane: at most R - 1 ops of at most 1024 pixels
    constexpr uint32_t CL_MASK = (1u << CL_BITS) - 1u;
    static_assert((R - 1) * 1024 <= (int)CL_MASK, "record layout");
    __shared__ __attribute__((aligned(16))) uint32_t s_bytes[W][LDS_DW];
    __shared__ uint32_t s_ftab[256];   // byte -> low half of its transition function
    // op byte -> {info, absolute-channel mask, LUMA r/b base, LUMA g | nibble mask} (decode_wave_kernel's table)
    //   info: len (3) | (len*8 - 4) (6) << 3 | pixels (10) << 9 | REF (1) << 20 | pixels if the byte is a RUN / BIGRUN (10) << 21
    __shared__ __attribute__((aligned(16))) u32x4 s_optab[256];
    __shared__ uint32_t s_sfx[256];    // byte -> ALPHA suffix: delta << 24 | 8 (shift) | 1 (length); RUN / BIGRUN: pixels << 8; else 0
#ifndef SQOA_BAL_LIST32   // the list's entries as dwords: two lanes' ds_write_b16 into one dword collide (2.881 -> 2.868 ms)
#define SQOA_BAL_LIST32 1
#endif
    typedef std::conditional_t<SQOA_BAL_LIST32 != 0, uint32_t, uint16_t> list_t;
#ifndef SQOA_BAL_LDS_STORES
#define SQOA_BAL_LDS_STORES 1
#endif
    // One region per wave, used twice in a chunk: first as the LIST of op starts (64 R entries, + room for the reads of the lanes behind the
    // last start and one entry in front), then -- the walk has read the list -- as the chunk's PIXELS in position order (a chunk of literal
    // ops and pixel pairs is 1.5 stream bytes a pixel or more: 683 pixels; RUNs add theirs; + two dwords behind for the last op's pair write, two to dump)
    constexpr int PIX_CAP = SQOA_BAL_LDS_STORES ? 768 : 0;
    constexpr int SCR_DW = (64 * R + 2 * R) * (int)sizeof(list_t) / 4 > PIX_CAP + 8 ? (64 * R + 2 * R) * (int)sizeof(list_t) / 4 : PIX_CAP + 8;
    __shared__ __attribute__((aligned(16))) uint32_t s_scr[W][SCR_DW];   // per wave: byte offset in the chunk of the op start of every rank
    __shared__ unsigned long long s_parse[2][W];
    __shared__ unsigned long long s_pos[W];
    __shared__ unsigned long long s_px[W];
    __shared__ uint32_t s_path;  // 1 = decoded here, 2 = refused / left to the other kernels
    __shared__ uint32_t s_oc;                      // output channels of the job: 3 or 4 (the header check)

    SQOA_PLACEMENT_PAD();
    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    const int jid = job_ids[blockIdx.x];
    const sqoa_hip_job job = jobs[jid];
    sqoa_hip_result* res = &results[jid];
    gcptr8 src = (gcptr8)job.src;
    gptr8 out = (gptr8)job.dst;
    const uint32_t size = job.src_size;

    if (threadIdx.x == 0) {
        // header (seqoia.h:672-688): written to the result exactly as sqoa_decode fills *desc
        const uint32_t magic = (uint32_t)src[0] << 24 | (uint32_t)src[1] << 16 | (uint32_t)src[2] << 8 | src[3];
        const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
        const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
        const uint32_t hc = src[12], hs = src[13], q = src[14] != 0x31;
        res->len = 0; res->status = SQOA_HIP_ST_OK;   // the row as every launch must find it: nobody else has touched it yet (sqoa_runtime.hip, plan_enqueue)
        res->desc.width = hw; res->desc.height = hh; res->desc.channels = (uint8_t)hc;
        res->desc.colorspace = (uint8_t)hs; res->desc.qoi_compat = (uint8_t)q;
        row_mark_standby(res, 0);
        const bool ok = !(hw == 0 || hh == 0 || hc < 1 || hc > 6 || hs > 1 || !(magic == 0x716f6966u || magic == 0x53716f61u) ||
                          (magic == 0x716f6966u && !q) || hh >= 400000000u / hw);
        uint32_t path = 1;
        // a caller that states the shape it expects (non-zero job.desc.width / height) gets BADARG for any other header
        const bool shape_ok = (job.desc.width == 0 || job.desc.width == hw) && (job.desc.height == 0 || job.desc.height == hh);
        if (!ok || !shape_ok) { res->status = SQOA_HIP_ST_BADARG; path = 2; }
        else {
            const uint32_t oc = job.desc.channels ? job.desc.channels : (hc < 3 ? 1u : 3u) + ((hc & 1u) ? 0u : 1u);
            if ((unsigned long long)hw * hh * oc > job.dst_cap) { res->status = SQOA_HIP_ST_OVERFLOW; path = 2; }
            else if (hc < 3 || q || oc < 3) { res->status = SQOA_HIP_ST_SLOWPATH; path = 2; }   // mono, QOI-compat, 1 / 2 channels out: the kernels behind
            else if (!((RGB && oc == 3) || (oc == 4 && !(job.dst & 3ull)))) { res->status = SQOA_HIP_ST_INTERNAL; path = 2; }   // (the plan launches this kernel for three channels, and for four to dword-aligned addresses, only)
            s_oc = oc;
        }
        s_path = path;
    }
    if (lane == 0) { s_parse[0][w] = 0; s_parse[1][w] = 0; s_pos[w] = 0; s_px[w] = 0; }
    for (uint32_t b0 = threadIdx.x; b0 < 256u; b0 += W * 64) {
        const bool is_abs = b0 >= 0xfeu, is_luma = (b0 & 0xc0u) == 0x80u, is_run = (b0 - 0xc0u) < 0x3eu, is_sfx = (b0 - 0x60u) < 0x20u;
        const uint32_t len = is_abs ? b0 - 0xfau : (is_luma ? 2u : 1u);
        // transition function (low half; states A, N, G, X): see the enum
        const uint32_t opn = len == 1u ? (uint32_t)S_A : (uint32_t)S_D1 + len - 2u;
        const uint32_t fA = is_sfx ? (uint32_t)S_N : is_run ? (uint32_t)S_G : opn;
        const uint32_t fN = is_run ? (uint32_t)S_G : opn;
        const uint32_t fG = is_sfx ? (uint32_t)S_X : opn;
        s_ftab[b0] = fA | fN << 8 | fG << 16 | (uint32_t)S_X << 24;
        // op decode of seqoia.h:740-775 as a table
        const uint32_t cnt = (is_abs || is_luma) ? 1u : (b0 == 0xfdu ? 512u : (b0 & 0x3fu) + 1u);
        const uint32_t vg = (b0 & 0x3fu) - 32u;
        u32x4 e;
        e.x = len | (len * 8u - (SQOA_BAL_RUN_IN_SFX == 2 ? 2u : 4u)) << 3 | cnt << 9 | (b0 < 0x60u ? 1u << 20 : 0u) | (is_run ? cnt << 21 : 0u);
        e.y = is_abs ? (b0 == 0xffu ? 0xffffffffu : 0x00ffffffu) : 0u;
        e.z = is_luma ? (((vg - 8u) & 0xffu) | ((vg - 8u) & 0xffu) << 16) : 0u;
        e.w = is_luma ? ((vg & 0xffu) << 8 | 0x000f000fu) : 0u;
        s_optab[b0] = e;
        s_sfx[b0] = is_sfx ? ((((b0 & 31u) - 16u) & 0xffu) << 24 | 9u) : (is_run ? cnt << 8 : 0u);
    }
    __syncthreads();
    if (s_path != 1) return;

    const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
    const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
    const uint32_t N = bcast_first(hw * hh);
    const uint32_t oc = RGB ? bcast_first(s_oc) : 4u;   // output channels: 4 (to a dword-aligned address) or 3 (the header check above)
    const bool oc3 = RGB && oc == 3u;
    // Streams packed back to back (sqoa_hip_pack) start at any byte address: all stream offsets below are relative to the
    // dword-aligned address at or in front of the stream; the `soff` bytes in front of it are loaded but never looked at.
    const uint32_t soff = (uint32_t)(job.src & 3ull);
    gcptr8 srca = src - soff;
    const uint32_t size_a = size + soff;
    const uint32_t p0 = kHeader + 1u + soff;    // first chunk byte (after the start byte, seqoia.h:705)
    const uint32_t chunks_len = size_a - 8;     // seqoia.h:721
    const uint32_t area = chunks_len > p0 ? chunks_len - p0 : 0;
    const uint32_t nchunks = area ? (area + CHUNK - 1) / CHUNK : 1;
    uint32_t bad = 0;
    uint32_t ops_seen = 0;   // the content sample for the plan: sum of op starts (20 bits) | chunks counted << 20

#ifdef SQOA_TRACE  // diagnostic builds: where wave 0 spends its time, written behind the image's pixels (dst_cap must allow it)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
    const unsigned long long tr_start = wall_clock64();
#define BPHASE(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - ph_t; ph_t = t_; } while (0)
#elif defined(SQOA_ASM_MARK)
#define BPHASE(k) asm volatile("; SQOA_PHASE " #k)
#else
#define BPHASE(k) do { } while (0)
#endif
    uint32_t* lds = s_bytes[w];
    list_t* list = reinterpret_cast<list_t*>(s_scr[w]) + R;
    typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
    typedef const __attribute__((address_space(1))) u32x4_a4* gcptr128u;
    u32x4 stg4 = {0u, 0u, 0u, 0u};
    uint32_t stg1 = 0;
    // the chunk's CHUNK + 32 bytes from its dword-aligned start: one 16-byte load per lane and eight dwords; straight-line when the
    // whole window lies inside the stream (all but the last chunks)
    // (the lane number recomputed where it is used, opaque to the compiler: everything derived from the kernel's `lane` alone is a loop
    // invariant that is hoisted out of the chunk loop, kept in a register -- or spilled, and then reloaded in front of the loads with
    // an s_waitcnt vmcnt(0) that waits for the stores of the chunk before)
    auto fresh_lane = [&]() { uint32_t l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
    auto load_chunk = [&](uint32_t c) {
        const uint32_t lane = fresh_lane();
        const uint32_t gs = p0 + c * CHUNK;
        const uint32_t d0 = gs >> 2;
        if ((d0 + LDS_DW) * 4u <= size_a) {
            gcptr32 p = (gcptr32)srca + d0;
            stg4 = ((gcptr128u)p)[lane];
            stg1 = p[256 + (lane & 7)];
        } else {
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t d = d0 + 4u * lane + k; r[k] = d * 4u < size_a ? ((gcptr32)srca)[d] : 0u; }
            stg4 = u32x4{r[0], r[1], r[2], r[3]};
            const uint32_t d = d0 + 256u + (lane & 7);
            stg1 = d * 4u < size_a ? ((gcptr32)srca)[d] : 0u;
        }
    };
    auto stage = [&]() {
        const uint32_t lane = fresh_lane();
        *reinterpret_cast<u32x4*>(lds + 4 * lane) = stg4;
        if (lane < LDS_DW - 256) lds[256 + lane] = stg1;
    };

    // One chunk.  IN (compile time) = interior chunk: every byte lies inside the chunk area and a successor exists.
    auto body = [&](auto in_tag, uint32_t c) -> uint32_t {   // 1: go on with the wave's next chunk, 0: the image is done with here
        constexpr bool IN = decltype(in_tag)::value;
        BPHASE(7);
        const bool more = c + W < nchunks;
#ifndef SQOA_BAL_PREFETCH_LATE
#define SQOA_BAL_PREFETCH_LATE 0   // (held from the top of the chunk to the end of its walk: 2.625 ms; from the end of the walk to the end of the chunk: 2.662)
#endif
        const uint32_t gs = p0 + c * CHUNK;               // stream offset of chunk byte 0
#ifndef SQOA_BAL_QD_REG   // 1: a lane's 16 bytes for the parse from the registers the chunk was fetched into (they still hold it: the next fetch is issued
#define SQOA_BAL_QD_REG 0 //    below), the dword behind them from the lane above by one DPP move -- not read back from the LDS copy the walk needs.
#endif                    //    Built, byte-exact (the decode suite), measured in round 6 and NOT taken: 2.416 -> 2.431 ms -- an aligned ds_read_b128 that
                          //    every lane issues at its own 16 bytes is not what this kernel waits for; the data-dependent look-ups are (SQOA_BAL_FLO_REG).
        uint32_t qd[LB / 4];
#if SQOA_BAL_QD_REG && !SQOA_BAL_PREFETCH_LATE
        {
            const uint32_t nx = shift_down1(stg4.x, bcast_lane(stg1, 0));   // (lane 63: the chunk's dword 256, which lane 0 fetched)
            const uint32_t dw[LB / 4 + 1] = {stg4.x, stg4.y, stg4.z, stg4.w, nx};
#pragma unroll
            for (int k = 0; k < LB / 4; ++k) qd[k] = __builtin_amdgcn_alignbyte(dw[k + 1], dw[k], gs & 3u);
        }
#endif
#if !SQOA_BAL_PREFETCH_LATE
        if (more) load_chunk(c + W);
#endif
        const bool last_chunk = IN ? false : c == nchunks - 1;
        const uint32_t my0 = gs + LB * lane;
        const uint32_t nlive = IN ? (uint32_t)LB : my0 >= chunks_len ? 0u : (chunks_len - my0 > (uint32_t)LB ? (uint32_t)LB : chunks_len - my0);
        const uint32_t live_end = IN ? (uint32_t)CHUNK : (chunks_len - gs > (uint32_t)CHUNK ? (uint32_t)CHUNK : chunks_len - gs);   // chunk bytes inside the area

        // ---- 1. parse: each lane composes the transition functions of its 16 bytes, one wave scan ----
#if !(SQOA_BAL_QD_REG && !SQOA_BAL_PREFETCH_LATE)
        {   // my 16 bytes start (gs & 3) bytes into dword 4 * lane of the wave's buffer: one 16-byte read, one dword, four v_alignbyte_b32
            const u32x4 a = *reinterpret_cast<const u32x4*>(lds + 4 * lane);
            const uint32_t dw[LB / 4 + 1] = {a.x, a.y, a.z, a.w, lds[4 * lane + 4]};
#pragma unroll
            for (int k = 0; k < LB / 4; ++k) qd[k] = __builtin_amdgcn_alignbyte(dw[k + 1], dw[k], gs & 3u);
        }
#endif
        // (the 16 table reads are issued together and waited for once -- the empty asm wants all of them in registers; left to itself
        // the scheduler, at this kernel's register pressure, made it one LDS round trip per byte)
        uint32_t flo[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            const uint32_t b = (qd[k / 4] >> (8 * (k & 3))) & 0xffu;
            flo[k] = (uint32_t)k < nlive ? s_ftab[b] : F_DEAD;
        }
#ifndef SQOA_BAL_PARSE_BATCH
#define SQOA_BAL_PARSE_BATCH 1
#endif
#if SQOA_BAL_PARSE_BATCH
        asm volatile("" : "+v"(flo[0]), "+v"(flo[1]), "+v"(flo[2]), "+v"(flo[3]), "+v"(flo[4]), "+v"(flo[5]), "+v"(flo[6]), "+v"(flo[7]));
        asm volatile("" : "+v"(flo[8]), "+v"(flo[9]), "+v"(flo[10]), "+v"(flo[11]), "+v"(flo[12]), "+v"(flo[13]), "+v"(flo[14]), "+v"(flo[15]));
#endif
#ifndef SQOA_BAL_FLO_REG   // 1: the sixteen entries stay in registers for the starts pass; 0: that pass looks them up in s_ftab again (rounds 5 and 6)
#define SQOA_BAL_FLO_REG 1
#endif
        Fn F{ID_LO, ID_HI};
#pragma unroll
        for (int k = 0; k < LB; ++k) F = compose(Fn{flo[k], F_HI}, F);
        const Fn Fi = wave_compose_scan(F);
        const Fn G{bcast_lane(Fi.lo, 63), bcast_lane(Fi.hi, 63)};
        const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
        // exit state independent of the entry state?  (every state but X, which enters no chunk of a stream still decoded here)
        const uint32_t g0 = G.lo & 0xffu;
        const bool g_const = (G.lo & 0x00ffffffu) == g0 * 0x010101u && G.hi == g0 * 0x01010101u;
        unsigned long long* my_parse = &s_parse[(c / W) & 1][w];
        BPHASE(0);
        uint32_t st_in = S_G;    // stream start: no op in front (seqoia.h:773-775)
        {
            // An exit state that does not depend on the entry state is posted before the entry state is known; the two-deep slots are safe
            // because the pixel chain below holds a wave back (it cannot finish chunk c + W before chunk c + 1 has been entered).
            if (g_const && !last_chunk) post(my_parse, c + 1, g0, lane);
#ifdef SQOA_DBG_BAL_NOCHAIN   // timing experiments only (wrong output): nothing is waited for -- what do the chains cost?
            if (c > 0) st_in = S_A;
#else
            if (c > 0) {
                const uint32_t rc = chain_wait_rc<SPIN>(&s_parse[((c - 1) / W) & 1][(c - 1) % W], c, st_in);
                if (rc) { bad |= rc >> 1; return 0u; }
            }
#endif
            if (!g_const && !last_chunk) post(my_parse, c + 1, apply(G, st_in), lane);
        }
        BPHASE(1);

        // ---- 2. starts: the parser state in front of every byte of mine, the op starts among them, their ranks, the list ----
        const uint32_t lst = apply_perm(Fx, st_in);
#if SQOA_BAL_FLO_REG
        // The parse's sixteen function entries are still in their registers: between the parse and the walk nothing else wants them (the walk's
        // op records are not alive yet; 64 VGPRs, no spill), and what the second look-up cost was never the instruction but the LDS pipe this
        // kernel is bound by -- sixteen data-dependent ds_read_b32 a lane, 50 of the chunk's 219 modelled bank-conflict cycles.  Same call,
        // headline batch: looked up again 2.550 ms; twelve parked in the wave's list region as three 16-byte pieces and four looked up 2.504;
        // twelve parked and four in registers 2.472; eight / four parked 2.448 / 2.429; all sixteen in registers 2.403 (-5.8 %).  (The parse's
        // OWN look-ups through the vector cache instead, from a copy of the table in memory: 2.87 ms -- the wait for them is a wait for the
        // stores of the chunk before.)  The empty asm pins them: the compiler may not look them up again either.
#pragma unroll
        for (int k = 0; k < LB; ++k) asm volatile("" : "+v"(flo[k]));
#else
        // (my 16 bytes, opaque to the compiler from here on: it would otherwise keep the parse's 16 table addresses in registers across the
        // scan and the chain wait -- and spill the walk's op records for them; recomputing them is one instruction each)
#pragma unroll
        for (int k = 0; k < LB / 4; ++k) asm volatile("" : "+v"(qd[k]));
#endif
        uint32_t M;   // bit k: an op (a walk iteration) starts at my byte k
#ifdef SQOA_DBG_BAL_NOSTARTS
        M = 0x1111u; asm volatile("" : "+v"(M));
        uint32_t trouble = 0;
        if (false)
#else
        uint32_t trouble = 0;   // (last chunks only: X is where their dead bytes lead anyway)
#endif
        {
            uint32_t nxt[LB / 4];   // byte i of word w: the state BEHIND my byte 4 w + i
            uint32_t st = lst;
            [[maybe_unused]] const uint32_t (&flo_p1)[LB] = flo;   // (the parse's entries: the loop below has a `flo` of its own)
#pragma unroll
            for (int wd = 0; wd < LB / 4; ++wd) {
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 4 * wd + i;
#if SQOA_BAL_FLO_REG
                    const uint32_t flo = flo_p1[k];
#else
                    const uint32_t b = (qd[wd] >> (8 * i)) & 0xffu;
                    const uint32_t flo = (uint32_t)k < nlive ? s_ftab[b] : F_DEAD;
#endif
                    if (!IN) trouble |= ((st & 0xffu) == (uint32_t)S_G && flo == (1u | 0u << 8 | 3u << 16 | 3u << 24)) ? 1u : 0u;   // an ALPHA byte behind a RUN that was taken along
                    st = __builtin_amdgcn_perm(F_HI, flo, st);   // (byte 0; the other bytes are states too, of no interest)
                    o[i] = st;
                }
                const uint32_t p01 = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0400u), p23 = __builtin_amdgcn_perm(o[3], o[2], 0x0c0c0400u);
                nxt[wd] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
            }
            // start at byte k  <=>  state in front of k is a boundary (A, N, G: <= 2) and the state behind it is A or inside an op
            // (not N: the byte was an ALPHA suffix; not G: a RUN taken along; not X)
            uint32_t fl[LB / 4];
            uint32_t prev = lst << 24;
#pragma unroll
            for (int wd = 0; wd < LB / 4; ++wd) {
                const uint32_t cur = __builtin_amdgcn_alignbyte(nxt[wd], prev, 3);
                prev = nxt[wd];
                const uint32_t t1 = cur + 0x05050505u, t2 = nxt[wd] + 0x04040404u, t3 = nxt[wd] + 0x07070707u;   // bit 3: cur >= 3; nxt >= 4; nxt != 0
                fl[wd] = (~t1 & (t2 | ~t3)) & 0x08080808u;
            }
            const uint32_t m8lo = __builtin_amdgcn_udot4(fl[1], 0x80402010u, __builtin_amdgcn_udot4(fl[0], 0x08040201u, 0u, false), false);
            const uint32_t m8hi = __builtin_amdgcn_udot4(fl[3], 0x80402010u, __builtin_amdgcn_udot4(fl[2], 0x08040201u, 0u, false), false);
            M = (m8lo | m8hi << 8) >> 3;   // (the flags are 0x08 per byte)
        }
        const uint32_t n_lane = (uint32_t)__builtin_popcount(M);
#ifndef SQOA_BAL_SCAN_DPP   // the rank scan as six v_add_u32_dpp (sqoa_common.h): the compiler's form of it is 17 instructions
#define SQOA_BAL_SCAN_DPP 1
#endif
#if SQOA_BAL_SCAN_DPP
        const uint32_t rinc = wave_incl_scan_dpp(n_lane);
#else
        const uint32_t rinc = wave_incl_scan(n_lane);
#endif
        const uint32_t n_ops = bcast_lane(rinc, 63);
        // (more than 64 R walk iterations in a chunk: two starts are two bytes apart or more on average unless one-byte ops that nothing takes
        // along follow each other -- SQOA_OP_REF bytes, which hand the stream over anyway, but also RUN behind RUN (flat regions: BIGRUN bytes on
        // end), and 512 two-byte ops plus ONE one-byte op are 513.  The stream goes to the retry behind this launch: the file's head.)
        if (n_ops > 64u * R) {   // (before the list is written: it has room for 64 R starts)
            if (lane == 0) { atomicMax(&res->status, SQOA_HIP_ST_SLOWPATH); row_content_hint(res, 127u); }
            return 0u;
        }
        // what the content looks like, for the plan (row_content_hint, sqoa_common.h): op starts per chunk over the first 256 chunks of wave 0's
        if (c < 2048u) ops_seen += n_ops + (1u << 20);
        {
            // Every start's byte offset to the list at its rank.  Hand-written: per start the compiler's loop was a wave vote, two compares, a
            // select and four scalar instructions around the five that do the work.  Here a lane that has run out of starts leaves EXEC
            // (v_cmpx; it never comes back: its mask stays zero) and EXEC is put back once.  Eight steps: more starts in 16 bytes than that
            // need one-byte ops that no predecessor takes along, which the encoder writes nowhere -- the loop behind is for streams that do.
            const uint32_t rank0 = rinc - n_lane;
            uint32_t m = M, t;
            const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) list_t*)list + (uint32_t)sizeof(list_t) * rank0;
            const uint32_t data0 = (uint32_t)(LB * lane);
            const unsigned long long exec_in = exec_saved();
#if SQOA_BAL_LIST32
#define SQOA_BAL_LIST_WR "ds_write_b32"
#else
#define SQOA_BAL_LIST_WR "ds_write_b16"
#endif
#define SQOA_BAL_LIST_STEP(OFF)                      \
            "v_cmpx_ne_u32_e32 vcc, 0, %[m]\n\t"      \
            "v_ffbl_b32_e32 %[t], %[m]\n\t"           \
            "v_add_u32_e32 %[t], %[t], %[d0]\n\t"     \
            SQOA_BAL_LIST_WR " %[a], %[t] offset:" #OFF "\n\t" \
            "v_add_u32_e32 %[t], -1, %[m]\n\t"        \
            "v_and_b32_e32 %[m], %[t], %[m]\n\t"
            asm volatile(
#if SQOA_BAL_LIST32
                SQOA_BAL_LIST_STEP(0) SQOA_BAL_LIST_STEP(4) SQOA_BAL_LIST_STEP(8) SQOA_BAL_LIST_STEP(12)
                SQOA_BAL_LIST_STEP(16) SQOA_BAL_LIST_STEP(20) SQOA_BAL_LIST_STEP(24) SQOA_BAL_LIST_STEP(28)
#else
                SQOA_BAL_LIST_STEP(0) SQOA_BAL_LIST_STEP(2) SQOA_BAL_LIST_STEP(4) SQOA_BAL_LIST_STEP(6)
                SQOA_BAL_LIST_STEP(8) SQOA_BAL_LIST_STEP(10) SQOA_BAL_LIST_STEP(12) SQOA_BAL_LIST_STEP(14)
#endif
                "s_mov_b64 exec, %[ex]\n\t"
                "s_nop 4"      // (a VALU write of EXEC wants 5 wait states in front of a DPP instruction)
                : [m] "+v"(m), [t] "=&v"(t) : [a] "v"(laddr), [d0] "v"(data0), [ex] "s"(exec_in) : "memory", "vcc");
#undef SQOA_BAL_LIST_STEP
            if (__ballot(m != 0u) != 0ull) {
                list_t* my = list + rank0;
                for (uint32_t i = 8; __ballot(m != 0u) != 0ull; ++i) {
                    if (m != 0u) { my[i] = (list_t)(data0 + (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
                }
            }
            __builtin_amdgcn_wave_barrier();   // (the list is read by other lanes of this wave: LDS executes a wave's instructions in order)
        }
        // given up: SQOA_OP_REF is found by the walk; an ALPHA suffix behind a RUN that was taken along leaves the parser in X
        if (IN ? (apply(G, st_in) == (uint32_t)S_X || st_in == (uint32_t)S_X) : (__ballot(trouble != 0u) != 0ull || st_in == (uint32_t)S_X)) {
            if (lane == 0) atomicMax(&res->status, SQOA_HIP_ST_SLOWPATH);
            return 0u;
        }
        BPHASE(2);

        // ---- 3. walk, scans, chain, stores: the chunk's ops in pieces of at most 64 R (one piece but for crafted streams) ----
        const uint32_t cb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (gs & 3u);   // LDS byte address of chunk byte 0
        const uint32_t optab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) u32x4*)s_optab;
        const uint32_t sfx_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_sfx;
        auto window = [&](uint32_t a, uint32_t& lo, uint32_t& hi) {
            lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
            const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
            lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
            hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
        };
        uint32_t pos_in = 0, px_in = kPxInit, pos_out, exit_v;
#if SQOA_BAL_HL_VGPR
        HL hl;
        asm volatile("v_mov_b32 %0, 0x80808080" : "=v"(hl.H));
        asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(hl.L));
#endif
        {
            // The chunk's n ops over the lanes: q = ceil(n / 64) per lane for the first k lanes, q - 1 for the others (k = n - 64 (q - 1)) -- every
            // lane decodes in slots 1 .. q - 1 and only slot 0 is a matter of the lane (the lanes below k): one masked walk iteration per
            // chunk instead of a compare, an EXEC region and its merges in every one.
            const uint32_t q = bcast_first((n_ops + 63u) >> 6);                       // slots in use
            const uint32_t k = bcast_first(n_ops - 64u * (q - 1u));                   // lanes that use slot 0 (q = 0: no slot is looked at)
            const bool first_slot = (uint32_t)lane < k;
            // rank of the op in my slot i: lane (q - 1) + min(lane, k) + i - (slot 0 unused: 1)
            const uint32_t r_slot0 = q ? (uint32_t)lane * (q - 1u) + ((uint32_t)lane < k ? (uint32_t)lane : k) - (first_slot ? 0u : 1u) : 0u;
            const list_t* my_starts = list + r_slot0;   // (the start of the op in my slot i: my_starts[i])
            uint32_t val[R];   // pixel after my i-th op, relative to the pixel entering my lane unless absolute -- and, until the op is decoded, its start
            uint32_t meta[R];  // pixels before the op inside my lane (CL_BITS) | cnt (11) << CL_BITS
            uint32_t amask[R]; // byte mask of the channels of val[] that are absolute
            // (all eight starts fetched at once, into the registers their slots' pixels will take: one LDS round trip less in every iteration's
            // chain of five; what lies behind the chunk's last start is never looked at)
#pragma unroll
            for (int i = 0; i < R; ++i) val[i] = my_starts[i];
            Xf T{0u, 0u};
            uint32_t cl = 0, longest = 0, info_or = 0;
            static_for<0, R>([&](auto i_tag) -> bool {
                constexpr int i = decltype(i_tag)::value;
                if ((uint32_t)i >= q) return false;
                const uint32_t pos = val[i];
                if (i == 0) {
                    asm volatile("" : "=v"(amask[i]));             // (read later only where meta says the slot holds an op)
                    meta[i] = 0;                                   // no op in this slot of mine: zero pixels
                }
                if (i != 0 || first_slot) {
                    uint32_t lo, hi;
                    window(cb_addr + pos, lo, hi);
                    const u32x4 e = *(lcptr128)(uintptr_t)(optab_addr + (lo & 0xffu) * 16u);
                    info_or |= e.x;
                    const uint32_t len = e.x & 7u, setm = e.y;
                    uint32_t cnt = (e.x >> 9) & 0x3ffu;
                    // byte after the op (seqoia.h:777-783): an ALPHA suffix belongs to the op.  w1 = stream >> (8 * len - 4).
                    const unsigned long long qq = (unsigned long long)hi << 32 | lo;
                    const uint32_t w1 = (uint32_t)(qq >> ((e.x >> 3) & 63u));
#if SQOA_BAL_RUN_IN_SFX == 2
                    // (w1 = stream >> (8 * len - 2): the byte behind the op times four in bits 2..9 -- the suffix table's byte offset in ONE instruction)
                    const uint32_t se = *(lcptr32)(uintptr_t)(sfx_addr + (w1 & 0x3fcu));
#else
                    const uint32_t se = *(lcptr32)(uintptr_t)(sfx_addr + ((w1 >> 2) & 0x3fcu));
#endif
                    const uint32_t has8 = se & 8u, da24 = se & 0xff000000u;
                    [[maybe_unused]] const uint32_t adv = len + (se & 1u);
                    // a RUN right behind the op (and its suffix) repeats the pixel this op produces: taken along, as the parser did (state G)
#if SQOA_BAL_RUN_IN_SFX == 2
                    // (the RUN count of the byte behind op and suffix from the suffix table's entries, four bytes apart -- the op table's second
                    // look-up asked for one dword of 16-byte entries: eight LDS banks for 64 lanes, a 4-way conflict in each half of the wave
                    // wherever the lanes' next bytes differ: class photo 3.19 -> 3.05 ms, the headline batch 2.573 -> 2.555, configs[4]'s shard 10.28 -> 10.13.
                    // Skipping the look-up by a vote of the wave where no lane has a suffix: photo 2.96, but the headline batch 2.61 -- its waves have.)
                    const uint32_t w2 = w1 >> has8;
                    uint32_t extra = (*(lcptr32)(uintptr_t)(sfx_addr + (w2 & 0x3fcu)) >> 8) & 0x3ffu;          // 0 unless RUN / BIGRUN
#else
                    const uint32_t w2 = w1 >> has8;
                    uint32_t extra = *(lcptr32)(uintptr_t)(optab_addr + (w2 & 0xff0u)) >> 21;               // 0 unless RUN / BIGRUN
#endif
                    if (!IN) extra &= sra31(pos + adv - live_end);                                         // (a RUN past the chunk area does not exist)
                    cnt += extra;
                    const uint32_t bse = __builtin_amdgcn_alignbit(hi, lo, 8) + da24;                                 // RGB[A], seqoia.h:740-752
                    const uint32_t nib = (((lo >> 12) & 0xfu) | ((lo << 8) & 0xf0000u)) & e.w;                         // LUMA, seqoia.h:761-769
                    const uint32_t d8 = ((e.z + nib) & 0x00ff00ffu) | (e.w & 0xff00u) | da24;
                    T = xf_then(T, Xf{(d8 & ~setm) | (bse & setm), setm} BHL);
                    val[i] = T.v;
                    meta[i] = cl | cnt << CL_BITS;
                    amask[i] = T.abs;
                    cl += cnt;
                    longest = longest > cnt ? longest : cnt;
                }
                return true;
            });
            if (__ballot(((info_or >> 20) & 1u) != 0u) != 0ull) {  // SQOA_OP_REF: serial path for the whole image
                if (lane == 0) atomicMax(&res->status, SQOA_HIP_ST_SLOWPATH);
                return 0u;
            }
#if SQOA_BAL_PREFETCH_LATE
            if (more) load_chunk(c + W);   // (late: the prefetched dwords are held through the scans and the stores, not through the parse and the walk)
#else
            if (more) stage();
#endif
            BPHASE(3);
            // ---- wave scans: pixel positions and pixel transforms of the lanes below ------------
            const uint32_t cinc = wave_incl_scan(cl);
            const uint32_t ctot = bcast_lane(cinc, 63);
#ifdef SQOA_DBG_BAL_NOXFSCAN
            const Xf Ti = T;
#else
            const Xf Ti = wave_xf_scan(T BHL);
#endif
            const Xf Tg{bcast_lane(Ti.v, 63), bcast_lane(Ti.abs, 63)};
            const Xf Tx{shift_up1(Ti.v, 0u), shift_up1(Ti.abs, 0u)};
            // ---- entry position / pixel from the chain, publish exits, store ------------------
            {
#ifdef SQOA_DBG_BAL_NOCHAIN
                if (c > 0) { pos_in = c * 400u; asm volatile("" : "+s"(pos_in)); }
                if (false) {
#else
                if (c > 0) {
#endif
                    // one wait for both: the position word is written before the pixel word by the same lane (LDS executes a wave's
                    // instructions in order), so it is there when the pixel word's tag is
                    const uint32_t rc = chain_wait_rc<SPIN>(&s_px[(c - 1) % W], c, px_in);
                    if (rc) { bad |= rc >> 1; return 0u; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    pos_in = bcast_first((uint32_t)__hip_atomic_load(&s_pos[(c - 1) % W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                }
            }
            BPHASE(4);
            exit_v = xf_apply(Tg, px_in);   // (wave-uniform: scalar arithmetic)
            pos_out = pos_in + ctot > N ? N : pos_in + ctot;
            if (!last_chunk) {   // (the position word first: a waiter that sees the pixel word's tag finds it there)
                lds_post(&s_pos[w], c + 1, pos_out, lane);
                lds_post(&s_px[w], c + 1, exit_v, lane);
            }
            if (pos_in < N) {
                // ---- the chunk's pixels: to LDS in position order, out as 16-byte pieces --------------------------------------------------
                // One store per op slot straight to memory -- what decode_wave_kernel does -- is 2 q instructions of 64 scattered dwords each
                // (half of them, for the second pixel of pixel pairs, nearly empty) and one more per RUN (three chunks in four of the headline
                // batch hold a RUN or two): the address pipe's time, not the lanes', is what they cost.  Here every slot writes its pixel into
                // the wave's LDS region at the pixel's position in the chunk AND the pixel behind it: its own again where the op covers two
                // pixels or more, else the NEXT op's (the next slot's; the next lane's first op's for the last slot) -- every writer of a
                // position writes the same value, so no predicate and no order is needed; what a RUN covers behind its first two pixels is
                // filled in by the whole wave (seqoia.h:723-725); then every lane takes four consecutive pixels back and stores them as one
                // piece, 4 KB a store instruction.  A chunk of more than PIX_CAP pixels (long RUNs) goes through the region in windows.
                const uint32_t lane_px = xf_apply(Tx, px_in BHL);       // pixel entering my lane
                const uint32_t total = pos_in + ctot > N ? N - pos_in : ctot;   // (the image's last chunk: pixels behind the image's end do not exist)
                const bool any_run = __ballot(longest > 2u) != 0ull;
                const uint32_t scr_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_scr[w];
                uint32_t* pix = s_scr[w];
                static_for<0, R>([&](auto i_tag) -> bool {
                    constexpr int i = decltype(i_tag)::value;
                    if ((uint32_t)i >= q) return false;
                    val[i] = add8(lane_px & ~amask[i], val[i] BHL);     // the slot's pixel
                    return true;
                });
                const uint32_t vfirst = first_slot ? val[0] : val[1];
                const uint32_t vnext_lane = shift_down1(vfirst, 0u);
                const uint32_t my_rel = cinc - cl;                   // the chunk's pixel my first op starts at
                constexpr uint32_t DUMP = 4u * (uint32_t)(PIX_CAP + 4);
                // pixels [wb, wb + n) of the chunk, which lie in the region, to memory
                auto read_back = [&](uint32_t wb, uint32_t n) {
                    typedef u32x4 u32x4_una __attribute__((aligned(4)));
                    typedef __attribute__((address_space(1))) u32x4_una* gptr128u;
                    uint32_t lane;   // (opaque: see load_chunk)
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
                    if (!oc3) {
                        for (uint32_t x0 = 0; x0 < n; x0 += 256u) {
                            const uint32_t x = x0 + 4u * lane;
                            gptr32 o = (gptr32)out + pos_in + wb + x;
                            if (x + 4u <= n) *(gptr128u)o = *reinterpret_cast<const u32x4*>(pix + x);
                            else if (x < n) {                                              // the last one to three pixels
                                o[0] = pix[x];
                                if (x + 1u < n) o[1] = pix[x + 1];
                                if (x + 2u < n) o[2] = pix[x + 2];
                            }
                        }
                    } else {
                        // three channels: a lane's four pixels are twelve bytes -- r g b r | g b r g | b r g b -- at a byte address that is
                        // whatever three times the pixel index makes it (one 12-byte store, as decode_wave_kernel's position-major path does)
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                        typedef u32x3 u32x3_una __attribute__((aligned(1)));
                        typedef __attribute__((address_space(1))) u32x3_una* gptr96u;
                        for (uint32_t x0 = 0; x0 < n; x0 += 256u) {
                            const uint32_t x = x0 + 4u * lane;
                            gptr8 o = out + ((size_t)pos_in + wb + x) * 3u;
                            if (x + 4u <= n) {
                                const u32x4 v = *reinterpret_cast<const u32x4*>(pix + x);
                                *(gptr96u)o = u32x3{__builtin_amdgcn_perm(v.y, v.x, 0x04020100u), __builtin_amdgcn_perm(v.z, v.y, 0x05040201u),
                                                    __builtin_amdgcn_perm(v.w, v.z, 0x06050402u)};
                            } else if (x < n) {
                                put_pixel<3>(o, 0, pix[x]);
                                if (x + 1u < n) put_pixel<3>(o, 1, pix[x + 1]);
                                if (x + 2u < n) put_pixel<3>(o, 2, pix[x + 2]);
                            }
                        }
                    }
                };
                const unsigned long long exec_in = exec_saved();
                if (ctot <= (uint32_t)PIX_CAP) {   // the common chunk: one window, every position inside it
                    const uint32_t my_addr = scr_addr + (my_rel << 2);
                    static_for<0, R>([&](auto i_tag) -> bool {
                        constexpr int i = decltype(i_tag)::value;
                        if ((uint32_t)i >= q) return false;
                        const uint32_t m = meta[i];
                        uint32_t a = my_addr + ((m & CL_MASK) << 2);
                        if (i == 0) a = first_slot ? a : scr_addr + DUMP;                                  // (no op in my slot 0: to the dump words)
#ifndef SQOA_BAL_PIX_WRITE2
#define SQOA_BAL_PIX_WRITE2 0
#endif
#if SQOA_BAL_PIX_WRITE2
                        const uint32_t vn = (uint32_t)i + 1u == q ? vnext_lane : val[i + 1 < R ? i + 1 : i];
                        const uint32_t pairm = sra31(0u - (m >> (CL_BITS + 1)));                              // ~0: cnt >= 2
                        const uint32_t sec = (val[i] & pairm) | (vn & ~pairm);
                        asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(a), "v"(val[i]), "v"(sec) : "memory");
#else
                        // (the second dword only where the op covers it: half the bytes into the LDS pipe, which this kernel keeps 70 % busy)
                        asm volatile(
                            "ds_write_b32 %[a], %[v]\n\t"
                            "v_cmpx_lt_u32_e32 vcc, %[c2], %[m]\n\t"           // cnt >= 2
                            "ds_write_b32 %[a], %[v] offset:4\n\t"
                            "s_mov_b64 exec, %[ex]\n\t"
                            "s_nop 4"
                            :: [m] "v"(m), [a] "v"(a), [v] "v"(val[i]), [ex] "s"(exec_in), [c2] "n"(2u * CL_MASK + 1u) : "memory", "vcc");
#endif
                        return true;
                    });
                    if (any_run) {
                        static_for<0, R>([&](auto i_tag) -> bool {
                            constexpr int i = decltype(i_tag)::value;
                            if ((uint32_t)i >= q) return false;
                            const uint32_t m = meta[i], cnt = m >> CL_BITS;
                            unsigned long long runs = __ballot(cnt > 2u);
                            while (runs) {
                                const int l = __builtin_ctzll(runs);
                                runs &= runs - 1ull;
                                const uint32_t rp = bcast_lane(my_rel + (m & CL_MASK), l) + 2u, rc = bcast_lane(cnt, l) - 2u, rv = bcast_lane(val[i], l);
                                for (uint32_t t = (uint32_t)lane; t < rc; t += 64u) pix[rp + t] = rv;
                            }
                            return true;
                        });
                    }
                    __builtin_amdgcn_wave_barrier();
                    read_back(0u, total);
                } else {
                    // A chunk of more pixels than the region holds (long RUNs: a flat stretch).  The pixels of its literal ops go straight to
                    // memory, one store per slot and pixel; its RUNs are listed in the region, four slots at a time -- (position, count), pixel
                    // -- and one loop expands them, 64 pixels a store.  (Not a path to be fast on: the plan sends RUN-heavy batches to
                    // decode_wave_kernel's position-major variant; but compact -- as eight unrolled slots with a fill loop each it cost the
                    // walk its registers.)
                    const unsigned long long obase = job.dst;
                    const uint32_t lane_off = (pos_in + my_rel) << 2;
                    auto half = [&](auto h_tag) {
                        constexpr int H = decltype(h_tag)::value;
                        uint32_t nruns = 0;
                        static_for<4 * H, 4 * H + 4>([&](auto i_tag) -> bool {
                            constexpr int i = decltype(i_tag)::value;
                            if ((uint32_t)i >= q) return false;
                            uint32_t m = meta[i];
                            const uint32_t P = pos_in + my_rel + (m & CL_MASK), cnt = m >> CL_BITS;
                            // (behind the image's end nothing exists: the slot's count as far as the image reaches)
                            const uint32_t room = P < N ? N - P : 0u;
                            const uint32_t c1 = cnt < room ? cnt : room;
                            const uint32_t mm = (m & CL_MASK) | ((c1 < 2u ? c1 : 2u) << CL_BITS);
                            const uint32_t off = lane_off + ((m & CL_MASK) << 2);
                            if (!oc3)
                            asm volatile(
                                "v_cmpx_lt_u32_e32 vcc, %[c1], %[m]\n\t"           // a pixel
                                "global_store_dword %[off], %[v], %[base]\n\t"
                                "v_cmpx_lt_u32_e32 vcc, %[c2], %[m]\n\t"           // a second one
                                "global_store_dword %[off], %[v], %[base] offset:4\n\t"
                                "s_mov_b64 exec, %[ex]\n\t"
                                "s_nop 4"
                                :: [m] "v"(mm), [off] "v"(off), [v] "v"(val[i]), [base] "s"(obase), [ex] "s"(exec_in), [c1] "n"(CL_MASK), [c2] "n"(2u * CL_MASK + 1u) : "memory", "vcc");
                            else {   // (three channels: byte stores; not a path to be fast on)
                                if (c1 >= 1u) put_pixel<3>(out, P, val[i]);
                                if (c1 >= 2u) put_pixel<3>(out, P + 1u, val[i]);
                            }
                            const bool isrun = c1 > 2u;
                            const unsigned long long rm = __ballot(isrun);
                            const uint32_t rk = nruns + __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                            if (isrun) { pix[rk] = P + 2u; pix[256 + rk] = c1 - 2u; pix[512 + rk] = val[i]; }
                            nruns += (uint32_t)__builtin_popcountll(rm);
                            return true;
                        });
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t r = 0; r < nruns; ++r) {
                            const uint32_t rp = bcast_first(pix[r]), rc = bcast_first(pix[256 + r]), rv = bcast_first(pix[512 + r]);
                            if (!oc3) for (uint32_t t = (uint32_t)lane; t < rc; t += 64u) ((gptr32)out)[rp + t] = rv;
                            else fill_pixels<3>(out, rp, rc, rv, lane);
                        }
                        __builtin_amdgcn_wave_barrier();
                    };
                    half(std::integral_constant<int, 0>{});
                    if (q > 4u) half(std::integral_constant<int, 1>{});
                }
            }
            BPHASE(5);
        }
#if SQOA_BAL_PREFETCH_LATE
        if (more) stage();
#endif
        if (last_chunk) {  // stream exhausted: the last pixel repeats (seqoia.h:726)
            if (pos_out < N) {
                if (!oc3) fill_pixels<4>(out, pos_out, N - pos_out, exit_v, lane);
                else fill_pixels<3>(out, pos_out, N - pos_out, exit_v, lane);
            }
            if (lane == 0) res->len = (int)(N * (uint32_t)oc);
        }
        return 1u;
    };

    uint32_t c = (uint32_t)w;
    if (c < nchunks) { load_chunk(c); stage(); }
    const uint32_t nint = area / CHUNK < nchunks - 1 ? area / CHUNK : nchunks - 1;  // fully live chunks that have a successor
    uint32_t go = 1;
    // Progress-based wave priority (see decode_wave_kernel): co-resident streams stay within a quarter of each other
    const uint32_t thr1 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T1 >> 6), thr2 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T2 >> 6),
                   thr3 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T3 >> 6);
    uint32_t thr = thr1, pq = 0;
    __builtin_amdgcn_s_setprio(3);
    for (; go && c < nint; c += W) {
        if (c >= thr) {
            ++pq; thr = pq == 1 ? thr2 : pq == 2 ? thr3 : ~0u;
            if (pq == 1) __builtin_amdgcn_s_setprio(2);
            else if (pq == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        go = body(std::true_type{}, c);
    }
    for (; go && c < nchunks; c += W) go = body(std::false_type{}, c);
    if (!go && lane == 0) {   // handed over, or the predecessor was: tell whoever waits on this wave
        const unsigned long long gone = (unsigned long long)DEC_ABORT_TAG << 32;
        __hip_atomic_store(&s_parse[0][w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&s_parse[1][w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&s_px[w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
    if (w == 0 && lane == 0 && go && (ops_seen >> 20) >= 4u) {   // (a stream of a few chunks says nothing)
        const uint32_t avg = (ops_seen & 0xfffffu) / (ops_seen >> 20);
        row_content_hint(res, 1u + (avg >> 2 > 125u ? 125u : avg >> 2));
    }
#ifdef SQOA_TRACE
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* tr = (unsigned long long*)((job.dst + job.dst_cap + 7) & ~7ull);
        for (int k = 0; k < 8; ++k) tr[k] = ph[k];
        tr[8] = tr_start; tr[9] = wall_clock64();
        tr[10] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | (unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32;
    }
#endif
}

}  // namespace bal
}  // namespace sqoa

// One workgroup per stream, the balanced kernel (see the file comment): the plan's choice for plans of large streams that go to four
// channels (sqoa_runtime.hip; SQOA_HIP_DEC_BAL = 0 / 1 forces it).
extern "C" hipError_t sqoa_launch_decode_bal(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int rgb, sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (rgb) hipLaunchKernelGGL((sqoa::bal::decode_bal_kernel<8, SQOA_BAL_LB, true>), dim3(n), dim3(8 * 64), 0, stream, d_jobs, d_ids, d_results);
    else hipLaunchKernelGGL((sqoa::bal::decode_bal_kernel<8, SQOA_BAL_LB, false>), dim3(n), dim3(8 * 64), 0, stream, d_jobs, d_ids, d_results);
    return hipGetLastError();
}
