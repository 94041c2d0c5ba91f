// sqoa_common.h -- device-side helpers shared by the gfx950 kernels.
// Wavefront = 64 lanes everywhere; nothing here is portable and nothing needs to be.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sqoa_hip.h"

#define SQOA_DEV __device__ __forceinline__

// A/B builds at different code placements (tools/mkplaced.sh: -DSQOA_PAD=0 / 2 / 4 -> the kernel's body starts 0 / 8 / 16 bytes later): the
// same source timed at three placements says how much of a difference between two variants is where their loops happen to lie against the
// instruction fetch granularity (cdna_hip_programming.md, rule 27) -- a keep / drop decision below that band is not one.
#define SQOA_STR2(x) #x
#define SQOA_STR(x) SQOA_STR2(x)
#if defined(SQOA_PAD) && SQOA_PAD > 0
#define SQOA_PLACEMENT_PAD() asm volatile(".rept " SQOA_STR(SQOA_PAD) "\n\ts_nop 0\n\t.endr")
#else
#define SQOA_PLACEMENT_PAD() do { } while (0)
#endif

// s_sleep argument (x 64 cycles) between two polls of a chain word: a polling wave spends scalar-pipe slots, which the
// kernels are short of (tools/ubench/salu_mix.hip)
#ifndef SQOA_ENC_SPIN_SLEEP
#define SQOA_ENC_SPIN_SLEEP 8   // encode 2.13 -> 2.09 ms against 1; the decoder's shorter chunks prefer 1 (3.87 vs 4.03 ms at 8)
#endif
#ifndef SQOA_DEC_SPIN_SLEEP
#define SQOA_DEC_SPIN_SLEEP 1
#endif

// Tables of the chip-wide decode passes for few large streams (sqoa_decode_wave.hip); built by the plan (sqoa_runtime.hip).
struct sqoa_dp_tables {
    void* d_images; int n_images;   // dp_image rows: the multi-pass images, = the first n_images of the plan's id list
    void* d_wgs; int n_wgs;         // dp_wg rows: one per workgroup of a pass
    unsigned per;                   // chunks a workgroup takes
    void* d_tab;                    // dp_chunk rows: one per chunk
    void* d_flags;                  // one word per image, written by pass 1: 1 wave path, 2 left to the other kernels, 3 QOI-compat colour stream: speculative pass
    void* d_qstate;                 // two decoder states (pixel + colour index, 512 bytes each) per workgroup row: the speculative pass of QOI-compat streams
    int any_qoi;                    // a HINT, taken when the plan was filled: some of these streams are in QOI-compat framing (0: the speculative pass and its check are not launched)
    int cus;                        // compute units of the plan's device
    int lb;                         // stream bytes per lane of the passes (8, or 12 for plans without a QOI-compat stream among these: the speculative pass counts 512-byte chunks)
};

// Tables of the strip encode of few large images (sqoa_strips.hip); built by the plan (sqoa_runtime.hip).
struct sqoa_st_tables {
    void* d_images; int n_images;   // st_image rows: the images cut into strips, = the first n_images of the plan's SQOA id list
    void* d_strips; int n_strips;   // st_strip rows: (image, nominal cut) per strip row
    void* d_cuts;                   // one word per strip row: the pixel the strip starts at, or none
    void* d_jobs; void* d_ids; void* d_results;   // strip jobs, their id list (-1: skipped row), their result rows
    void* d_offsets;                // n_strips + 1 prefix sums of the strips' stream lengths
    void* d_ws; size_t ws_bytes;    // the strips' output slots
    // QOI-compat images among them: the LAST n_images_qoi image rows / n_strips_qoi strip rows
    int n_images_qoi, n_strips_qoi;
    void* d_qtab;                   // per strip row 64 x 8 bytes: the strip's last writer per slot of the colour index
    void* d_qtin;                   // per strip row 64 dwords: the colour index entering the strip
};

namespace sqoa {

// Rows of those tables: filled on the host by the plan (sqoa_runtime.hip, plan_fill), read by the kernels -- ONE definition.
struct dp_image { int jid; uint32_t tab_off, nchunks, nwgs; };   // one multi-pass image: its job, its rows of the table (one per workgroup)
struct dp_wg { uint32_t img, chunk0; };                           // one workgroup of a pass: image and first chunk
struct st_image { int jid; uint32_t first, nstrips, npx; unsigned long long ws; };   // strip rows [first, first + nstrips), workspace
struct st_strip { uint32_t img, k; };
// What the LDS chain of the wave decoder carries, per workgroup of a pass, in HBM: the composite of its chunks' transition functions,
// pixel counts and transforms (written by passes 1 and 2), and what enters its first chunk (written by the two scans; read by the
// speculative pass of sqoa_decode_qoi.hip as well).
struct dp_chunk { uint32_t g_lo, g_hi, st_in, ctot, tv, tabs, pos_in, px_in; };
static_assert(sizeof(dp_image) == 16 && sizeof(dp_wg) == 8 && sizeof(st_image) == 24 && sizeof(st_strip) == 8 && sizeof(dp_chunk) == 32, "table rows are packed");
// Stream bytes per chunk of the PASSES (decode_wave_kernel MODE 1..3, the speculative pass of decode_qoi_kernel and the plan's
// tables all count chunks of this size; the one-workgroup-per-stream kernels are free to cut their streams differently).
#ifndef SQOA_DEC_LB
#define SQOA_DEC_LB 8
#endif
constexpr int DP_CHUNK_BYTES = 64 * SQOA_DEC_LB;

// Device addresses arrive as integers in the job table; tell the compiler they are GLOBAL
// (address space 1) so it emits global_load/global_store, not flat_*.
typedef const __attribute__((address_space(1))) uint8_t* gcptr8;
typedef __attribute__((address_space(1))) uint8_t* gptr8;
typedef const __attribute__((address_space(1))) uint16_t* gcptr16;
typedef const __attribute__((address_space(1))) uint32_t* gcptr32;
typedef __attribute__((address_space(1))) uint32_t* gptr32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4* gptr128;
typedef const __attribute__((address_space(1))) u32x4* gcptr128;

constexpr uint32_t kPxInit = 0xff000000u;  // (0,0,0,255): reference seqoia.h:521-525, :716-719
constexpr int kHeader = 14;                // reference seqoia.h:425

// The unused fourth byte behind sqoa_desc's three flags in a result row: 1 = one of the kernels BEHIND the wave decoder took this job in the
// launch that wrote the row (reset by the wave decoder with the row).  The plan reads it with the rows: a decode plan none of whose jobs needed
// those kernels does not launch them the next time (sqoa_runtime.hip, sqoa_hip_plan_results).
// Bits 1..7 of the same byte, set by the balanced decoder (sqoa_decode_bal.hip) and read by the plan with the rows: what the stream's CONTENT looks
// like to that kernel -- 0 = nothing said; 1..126 = op starts per 1024-byte chunk / 4, averaged over a sample of the stream's chunks (ops of two
// bytes throughout -- photographs -- make 512 starts a chunk: every lane owns exactly eight ops, its LDS accesses fall on a quarter of the banks
// and decode_wave_kernel is the faster kernel: DESIGN.md 3.2c); 127 = a chunk held more starts than the kernel has slots (the stream went to the
// retry).  A plan whose launches mostly see such streams takes decode_wave_kernel from the next launch on (sqoa_runtime.hip, sqoa_hip_plan_results).
constexpr uint32_t ROW_HINT_SHIFT = 25;   // (bit 1 of byte 11 of sqoa_desc = bit 25 of its third dword)
SQOA_DEV void row_mark_standby(sqoa_hip_result* res, uint8_t v) {
    volatile uint8_t* b = reinterpret_cast<volatile uint8_t*>(&res->desc) + 11;
    if (v) *b = (uint8_t)(*b | 1u);   // (the kernels behind the wave decoder run in later launches of the stream: nobody else is writing the row)
    else *b = 0;
}
SQOA_DEV void row_content_hint(sqoa_hip_result* res, uint32_t hint) {
    atomicOr(reinterpret_cast<uint32_t*>(&res->desc) + 2, (hint & 127u) << ROW_HINT_SHIFT);
}

SQOA_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// DPP controls (GFX9 encodings)
constexpr int DPP_ROW_SHR = 0x110;   // + n
constexpr int DPP_WAVE_SHL1 = 0x130;
constexpr int DPP_WAVE_SHR1 = 0x138;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

// lane i <- v[i-1]; lane 0 <- carry
SQOA_DEV uint32_t shift_up1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
// lane i <- v[i+1]; lane 63 <- carry
SQOA_DEV uint32_t shift_down1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, DPP_WAVE_SHL1, 0xf, 0xf, false);
}

// inclusive prefix sum over the 64 lanes, 6 DPP adds
SQOA_DEV uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 1, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 2, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 4, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 8, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}

// The same, hand-written: six v_add_u32_dpp.  The compiler fuses the DPP move into the add only where it can see that the move's `old` operand is the
// add's identity; where the zero is a hoisted register copy (the balanced decoder's rank scan) it emits v_mov + v_mov_dpp + v_add per step -- three
// slow-class instructions for one.  (A VALU write of a VGPR wants two wait states in front of a DPP read of it: the s_nop 1.)
SQOA_DEV uint32_t wave_incl_scan_dpp(uint32_t v) {
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// Poll an 8-byte chain word in LDS -- tag in the high half, value in the low half, always written by ONE ds_write_b64 -- until the tag
// is `tag`, or `stop` (a tag no work item ever has: the owner has left), or 2^22 polls have failed.  Returns the tag last seen and,
// in `val`, the low half read by the same poll (the tag dword is read first: the value is at least as new).  Hand-written: the
// compiler turns the same loop into 25 scalar instructions and 8 branches on the path where the first poll succeeds (a
// wave-uniform condition lives in a 64-bit lane mask, every exit of the loop gets a flag of its own), and the scalar pipe of a
// CU is as busy as its vector pipes in these kernels.  Here that path is 2 LDS reads, 2 v_readfirstlane,
// one compare and one branch.  SLEEP = s_sleep argument between two polls (x 64 cycles).
template <int SLEEP>
SQOA_DEV uint32_t chain_poll(const unsigned long long* slot, uint32_t tag, uint32_t stop, uint32_t& val) {
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned long long*)slot;
    uint32_t t, v, left, tv, vv;
    asm volatile(
        "s_mov_b32 %[left], 0x400000\n"
        "1:\n\t"
        "ds_read_b32 %[tv], %[addr] offset:4\n\t"
        "ds_read_b32 %[vv], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[t], %[tv]\n\t"
        "s_cmp_eq_u32 %[t], %[tag]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_eq_u32 %[t], %[stop]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_add_i32 %[left], %[left], -1\n\t"
        "s_cmp_eq_u32 %[left], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sleep %[slp]\n\t"
        "s_branch 1b\n"
        "2:\n\t"
        "v_readfirstlane_b32 %[v], %[vv]"
        : [t] "=&s"(t), [v] "=s"(v), [left] "=&s"(left), [tv] "=&v"(tv), [vv] "=&v"(vv)
        : [addr] "v"(addr), [tag] "s"(tag), [stop] "s"(stop), [slp] "n"(SLEEP)
        : "memory", "scc");
    val = v;
    return t;
}

// EXEC as a value, for the hand-written blocks that narrow it (v_cmpx) and put it back.  Read by an instruction of its own: handed the builtin's
// value, the compiler may give such a block the EXEC register itself as its "saved" operand -- `s_mov_b64 exec, exec`, the narrowed mask stays,
// and a wave whose mask came out empty never finishes (seen once, round 5).
SQOA_DEV unsigned long long exec_saved() {
    unsigned long long e;
    asm volatile("s_mov_b64 %0, exec" : "=s"(e));
    return e;
}

// One 8-byte chain word -- tag in the high half, value in the low half -- written to LDS by lane 0 alone.  Hand-written: `if (lane == 0) store`
// makes the compiler keep "lane == 0" as a 64-bit lane mask across the kernel's main loop -- an SGPR pair the encoder has not got: spilled, and
// fetched back with two v_readlane_b32 (slow-class VALU) in front of every post, three or four times a sub-tile -- and wrap the store in a
// saveexec / branch / restore.  Here EXEC is lane 0 for the one ds_write_b64.  (LDS executes a wave's instructions in order: what the wave wrote to
// LDS before is there when the word is -- the "release" the chains need; nothing that crosses waves in these kernels goes through global memory.)
#ifndef SQOA_ASM_POST
#define SQOA_ASM_POST 1
#endif
SQOA_DEV void lds_post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) {
#if SQOA_ASM_POST
    (void)lane;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)slot;
    const unsigned long long w = (unsigned long long)tag << 32 | val;
    unsigned long long ex;
    asm volatile(
        "s_mov_b64 %[ex], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b64 %[a], %[w]\n\t"
        "s_mov_b64 exec, %[ex]"
        : [ex] "=&s"(ex) : [a] "v"(addr), [w] "v"(w) : "memory");
#else
    if (lane == 0) __hip_atomic_store(slot, (unsigned long long)tag << 32 | val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

SQOA_DEV uint32_t bcast_lane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
SQOA_DEV uint32_t bcast_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// 64-entry colour hash, reference seqoia.h:414-417:  (3r + 5g + 7b + 11a) & 63
SQOA_DEV uint32_t qoi_hash(uint32_t px) { return __builtin_amdgcn_udot4(px, 0x0b070503u, 0u, false) & 63u; }

// Pixel `idx` of an image whose input stride is `ch` bytes (1..4), as r|g<<8|b<<16|a<<24.
// Mono keeps r = b = 0 (reference seqoia.h:531-542); no alpha channel -> 255.
SQOA_DEV uint32_t load_px(gcptr8 src, uint32_t idx, int ch) {
    if (ch == 4) return ((gcptr32)src)[idx];
    if (ch == 3) {
        gcptr8 p = src + (size_t)idx * 3;
        return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | 0xff000000u;
    }
    if (ch == 2) {
        uint32_t v = ((gcptr16)src)[idx];
        return (v & 0xffu) << 8 | (v >> 8) << 24;
    }
    return (uint32_t)src[idx] << 8 | 0xff000000u;
}

}  // namespace sqoa
