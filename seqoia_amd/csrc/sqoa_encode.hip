// sqoa_encode.hip -- SQOA / QOI batch encoder for gfx950 (MI355X).
//
// Replaces the per-pixel loop of the reference encoder (seqoia.h:530-638) and
// its tail (seqoia.h:640-649).  Dataflow, per image:
//
//   * the image is cut into sub-tiles of 512 pixels = 8 rows of 64; sub-tile j is owned by wave (j mod W) of the
//     image's workgroup, which walks j, j+W, j+2W, ... while the next sub-tile's pixels land in LDS (global_load_lds);
//   * colour images take the SWAR path: the op decision (LUMA / ALPHA / RGB[A], DIFF in QOI mode) is made for the
//     same lane of FOUR rows at once, bytewise inside dwords (swar_classify); per row follow run bookkeeping from
//     the __ballot equality masks, byte offsets from DPP prefix sums, and variable-length compaction by OR-ing
//     each lane's (<= 9) bytes into a wave-private LDS staging buffer (swar_emit).  Mono images and the partial
//     sub-tile at the end of an image take the one-row-per-step path (literal_op);
//   * the only state that crosses sub-tiles is (run length, output position [, colour index]); it travels
//     wave -> wave through tagged 8-byte LDS words (a decoupled look-back chain inside the workgroup, no barrier);
//   * finished bytes leave LDS as 16-byte global stores, edges as byte stores.
//
// No MFMA: there is no contraction here.  The bound is HBM (4 B/px read,
// ~2.6 B/px written on the headline batch).
#include <type_traits>

#include "sqoa_common.h"

// This file is built TWICE into the one library (Makefile).  The second build, -DSQOA_LDS_ORDER_FB, holds the QOI-compat kernels in the form
// that does NOT rest on the order in which the LDS applies one instruction's lanes (per-slot 64-bit lane masks and ds_max_u64: SQOA_ENC_Q_RTN = 0,
// below), as sqoa::fb::encode_kernel behind sqoa_launch_encode_*_fb; the launchers of the first build hand QOI-compat launches over to them on a
// device that has not passed the library's probe of that order (sqoa_selftest.hip; SQOA_HIP_LDS_ORDER=0 forces it: the GPU tests run the
// QOI-compat parity set on both forms).
#ifdef SQOA_LDS_ORDER_FB
#undef SQOA_ENC_Q_RTN
#define SQOA_ENC_Q_RTN 0
#define SQOA_ENC_NS_OPEN namespace sqoa { namespace fb {
#define SQOA_ENC_NS_CLOSE } }
#define SQOA_ENC_SYM(name) name##_fb
#else
#define SQOA_ENC_NS_OPEN namespace sqoa {
#define SQOA_ENC_NS_CLOSE }
#define SQOA_ENC_SYM(name) name
#endif

#ifndef SQOA_PRIO_T1  // progress (in 64ths of the image) at which a workgroup drops to wave priority 2, 1, 0
#define SQOA_PRIO_T1 40
#define SQOA_PRIO_T2 54
#define SQOA_PRIO_T3 61
#endif

SQOA_ENC_NS_OPEN

constexpr int ENC_S = 8;               // steps per sub-tile
constexpr int ENC_SUB = ENC_S * 64;    // pixels per sub-tile
constexpr int ENC_STG = 2624;          // staging bytes per wave: 5 B/px * 512, + 9 (run flush) + 12 (OR window) + 16 (clear), multiple of 16
constexpr int ENC_PRE = ENC_SUB * 4;   // landing buffer of the next sub-tile's pixels (global -> LDS without passing VGPRs)
constexpr uint32_t SPIN_MAX = 1u << 22;
#ifndef SQOA_ENC_QTAB_SLEEP   // s_sleep argument between two polls of the colour-index hand-over tag (QOI-compat)
#define SQOA_ENC_QTAB_SLEEP SQOA_ENC_SPIN_SLEEP
#endif

struct EncState {
    uint32_t bad;   // chain wait timed out (never expected)
    uint32_t eager; // a strip job: poll the chain words at short intervals.  A full batch (four images per CU) wants long ones --
                    // a polling wave spends scalar issue slots the other images' waves need (2.13 -> 2.09 ms at 8 x 64 cycles
                    // against 1 x 64) -- but a strip of a lone image has the CU to itself and its 32 sub-tiles are one serial
                    // chain of position hand-overs: at 512-cycle polls that chain, not the work, was half of the kernel's 22 us
};

// Wait until the 8-byte LDS word carries `tag` in its high half; returns the low half.
// ASM: the hand-written poll of the decoder (chain_poll).  The SQOA-native encoder keeps the loop as the compiler makes it: it is short of
// SGPRs, not of scalar issue slots -- with the poll its SGPR spills went from 79 to 97 and the kernel from 1.57 to 1.59 ms.  The QOI-compat
// encoder takes it (round 4: headline batch 2.98 -> 2.90 ms, icons 2.94 -> 2.97 ms).
template <bool ASM = false>
SQOA_DEV uint32_t chain_wait(unsigned long long* slot, uint32_t tag, EncState& st) {
#ifdef SQOA_ENC_ASM_POLL
    constexpr bool USE_ASM = true;
#else
    constexpr bool USE_ASM = ASM;
#endif
    if constexpr (!USE_ASM) {
    for (uint32_t spin = 0; spin < SPIN_MAX; ++spin) {
        unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t hi = bcast_first((uint32_t)(v >> 32));
        if (hi == tag) return bcast_first((uint32_t)v);
        if (st.eager) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(SQOA_ENC_SPIN_SLEEP);
    }
    st.bad = 1;
    return 0;
    } else {
    uint32_t val;
    const uint32_t t = st.eager ? chain_poll<1>(slot, tag, 0xffffffffu, val)                       // (a strip job: EncState)
                                : chain_poll<SQOA_ENC_SPIN_SLEEP>(slot, tag, 0xffffffffu, val);   // (no tag is ever 2^32 - 1: sub-tile numbers + 1)
    if (t != tag) { st.bad = 1; return 0; }
    return val;
    }
}

SQOA_DEV void chain_post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) { lds_post(slot, tag, val, lane); }

#ifndef SQOA_ENC_LOOKBACK   // 1: the byte-position chain as a decoupled look-back (below) -- built, byte-exact on the whole suite, measured, NOT taken:
#define SQOA_ENC_LOOKBACK 0 // headline 1.512 -> 1.538 ms, QOI-compat 2.280 -> 2.360, icons 1.377 -> 1.422, photo RGB 1.250 -> 1.285 (same-call A/B, round 6)
#endif
// The byte position of sub-tile j's first byte (j >= 1) by DECOUPLED LOOK-BACK (round 6; an experiment that answered a question -- see the flag).
// The kernel hands the position from sub-tile to sub-tile: wave j % W waits for sub-tile j - 1's word, adds its own byte count, posts -- a serial
// chain of 2 048 hops through a 1 Mpixel image; wave 0 spends a third of its life in that wait (tools/trace_encode.py), and the kernel hardly
// cares how many instructions the rest of the loop takes (round 6: the whole cross-lane run fix-up removed, a sixth of the VALU instructions,
// bought 2.8 %).  Is the chain the bound, then?  Here every sub-tile publishes its byte COUNT (`agg`) the moment it is known -- before anything is
// waited for -- and its inclusive position (`pos`) when it has one; a wave that needs its position looks at the W sub-tiles in front of it at
// once (lanes 0..7: their positions, lanes 8..15: their counts, one ds_read_b64), takes the nearest position that is there and adds the counts
// of the sub-tiles in between: nobody waits for a predecessor to be SCHEDULED, only for counts, which all W waves of an image produce side by
// side.  Its own previous sub-tile's position (j - W) is always there, so the look-back ends as soon as the W - 1 counts in between are.
// Answer: no.  1.7 - 3.5 % SLOWER on every workload: the wait is slack -- time a wave gives to the 31 others of its compute unit while the
// board sits at its power cap --, not a critical path, and the look-back's two dozen instructions per poll are not free.
// Two-deep slots ([(sub-tile / W) & 1][sub-tile % W]): sub-tile j's count is overwritten by j + 2W's, which its wave produces after it has
// flushed j + W -- by then everybody who might want it finds j's POSITION (posted before j was flushed) and stops there; j's position is
// overwritten by j + 2W's, which needs the counts of j + W + 1 .. j + 2W - 1, whose waves have resolved j + 1 .. j + W - 1 before they got there.
template <int W>
SQOA_DEV uint32_t pos_lookback(const unsigned long long (*s_pos)[W], const unsigned long long (*s_agg)[W], uint32_t j, EncState& st) {
    static_assert(W <= 8, "one lane per sub-tile looked at");
    uint32_t lane;   // (recomputed: everything derived from the kernel's `lane` is a loop invariant the compiler keeps, or spills)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const uint32_t i = lane & 7u;          // this lane looks at sub-tile t = j - 1 - i (i >= j wraps: no tag is that large)
    const uint32_t t = j - 1u - i;
    const bool is_pos = lane < 8u;
    const uint32_t slot = ((t / (uint32_t)W) & 1u) * (uint32_t)W + t % (uint32_t)W;
    const uint32_t addr = (is_pos ? (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned long long*)&s_pos[0][0]
                                  : (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned long long*)&s_agg[0][0]) + 8u * slot;
    const bool mine = lane < 16u && i < (uint32_t)W && i < j;
    for (uint32_t spin = 0; spin < SPIN_MAX; ++spin) {
        unsigned long long word;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(word) : "v"(addr) : "memory");
        const uint32_t val = (uint32_t)word, tag = (uint32_t)(word >> 32);
        const uint32_t ok = (uint32_t)__ballot(mine && tag == t + 1u);   // (lanes 0..15 only: the mask's low half)
        const uint32_t mP = ok & 0xffu, mA = ok >> 8;
        if (mP) {
            const uint32_t kP = (uint32_t)__builtin_ctz(mP);               // the nearest sub-tile whose position is there: j - 1 - kP
            const uint32_t need = (1u << kP) - 1u;                        // the counts of the sub-tiles in between
            if ((mA & need) == need) {
                uint32_t pos = bcast_lane(val, (int)kP);
                for (uint32_t m = need; m; m &= m - 1u) pos += bcast_lane(val, 8 + __builtin_ctz(m));
                return pos;
            }
        }
        if (st.eager) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(SQOA_ENC_SPIN_SLEEP);
    }
    st.bad = 1;
    return 0;
}

// Op bytes of a pixel that differs from its predecessor (reference seqoia.h:584-635),
// little-endian in lo (bytes 0-3) / hi (byte 4).  QOI index hits are patched in by the caller.
template <bool QOI>
SQOA_DEV void literal_op(uint32_t px, uint32_t pv, bool mono, uint32_t& lo, uint32_t& hi, uint32_t& n) {
    const int vr = (int8_t)(px - pv);
    const int vg = (int8_t)((px >> 8) - (pv >> 8));
    const int vb = (int8_t)((px >> 16) - (pv >> 16));
    const int va = (int8_t)((px >> 24) - (pv >> 24));
    const int vg_r = (int8_t)(vr - vg), vg_b = (int8_t)(vb - vg);
    const uint32_t a1 = (uint32_t)(vg_r + 8), a2 = (uint32_t)(vg_b + 8), a3 = (uint32_t)(vg + 32), a4 = (uint32_t)(va + 16);
    // -8<=vg_r<8, -8<=vg_b<8, -32<=vg<32, -16<=va<16  (seqoia.h:606-611)
    const bool luma = (a1 | a2 | (a3 >> 2) | (a4 >> 1)) < 16u;
    const bool na = va != 0;
    if (mono) {  // seqoia.h:601-605, :612, :622-633 with col_channels == 1
        const uint32_t g = (px >> 8) & 255u;
        if (na) { lo = 0xffu | g << 8 | (px >> 24) << 16; n = 3; }
        else if (luma) { lo = 0x80u | a3; n = 1; }
        else { lo = 0xfeu | g << 8; n = 2; }
        hi = 0;
        return;
    }
    const uint32_t luma_lo = 0x80u | a3 | (a1 << 4 | a2) << 8 | (na ? (0x60u | a4) << 16 : 0u);
    const uint32_t raw_lo = px << 8 | 0xfeu | (na ? 1u : 0u);
    if (QOI) {
        const uint32_t d = (uint32_t)(vr + 2) | (uint32_t)(vg + 2) | (uint32_t)(vb + 2);
        if (na) { lo = px << 8 | 0xffu; hi = px >> 24; n = 5; }  // seqoia.h:573-580
        else if (d < 4u) { lo = 0x40u | (uint32_t)(vr + 2) << 4 | (uint32_t)(vg + 2) << 2 | (uint32_t)(vb + 2); hi = 0; n = 1; }
        else if (luma) { lo = luma_lo; hi = 0; n = 2; }
        else { lo = raw_lo; hi = 0; n = 4; }
        return;
    }
    lo = luma ? luma_lo : raw_lo;
    hi = (!luma && na) ? px >> 24 : 0u;
    n = (luma ? 2u : 4u) + (na ? 1u : 0u);
}

// One pixel at a wave-uniform address (any alignment) through the scalar cache: the two aligned dwords that hold its
// first and last byte (never a byte outside them), shifted together.  Upper bytes beyond `ch` are unspecified.
typedef const __attribute__((address_space(4))) uint32_t* kcptr32;
// The loads are only issued here; `value()` shifts the pair together where the pixel is first needed, so that the wave
// does not sit in s_waitcnt for the scalar cache at the point of issue.
struct NbrPx {
    uint32_t lo, hi, sh, orv;
    SQOA_DEV uint32_t value() const { return (uint32_t)(((unsigned long long)hi << 32 | lo) >> sh) | orv; }
};
SQOA_DEV NbrPx sload_px(unsigned long long addr, int ch, uint32_t orv) {
    return NbrPx{*(kcptr32)(addr & ~3ull), *(kcptr32)((addr + (unsigned)ch - 1u) & ~3ull), ((uint32_t)addr & 3u) * 8u, orv};
}
typedef __attribute__((address_space(3))) void* lds_ptr;
// The encoder's pointers into its own LDS buffers (staging, landing) are LDS pointers, 32 bits each, not generic ones: a generic pointer is an
// SGPR pair, and every use of it as an LDS address a null test and a select -- in a kernel that spills SGPRs (98 -> 63 with these).
typedef __attribute__((address_space(3))) uint8_t* lptr8;
typedef const __attribute__((address_space(3))) uint8_t* lcptr8;
typedef __attribute__((address_space(3))) uint32_t* lptr32;
typedef __attribute__((address_space(3))) u32x4* lptr_x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2* lptr_x2;
typedef uint32_t u32_una1 __attribute__((aligned(1)));
typedef u32x4 u32x4_una1 __attribute__((aligned(1)));
typedef __attribute__((address_space(3))) u32_una1* lptr32u;
typedef __attribute__((address_space(3))) u32x4_una1* lptr_x4u;

// Copy T staged bytes to global memory at g (any alignment) and clear the staging area.
SQOA_DEV void flush_stage(lptr8 stage, uint32_t T, gptr8 g, int lane_unused) {
    // The lane number, recomputed here and opaque to the compiler: with the caller's `lane` the per-lane LDS addresses below are
    // loop invariants, which at 64 VGPRs get SPILLED -- and every reload is a scratch_load + s_waitcnt vmcnt(0) in the middle of
    // the flush, i.e. a wait for the 16-byte stores issued just before it (the flush's pieces went out one store latency apart).
    (void)lane_unused;
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
#if defined(SQOA_ENC_FLUSH_ALIGNED)
    uint32_t a = (16u - (uint32_t)((uintptr_t)g & 15u)) & 15u;
    if (a > T) a = T;
    if ((uint32_t)lane < a) g[lane] = stage[lane];
    const uint32_t nb = (T - a) >> 4;
    for (uint32_t u = lane; u < nb; u += 64) {
        u32x4 v;
        v = *(lptr_x4u)(stage + a + 16u * u);  // unaligned LDS read
        *(gptr128)(g + a + 16u * u) = v;
    }
    const uint32_t t0 = a + 16u * nb;
    if ((uint32_t)lane < T - t0) g[t0 + lane] = stage[t0 + lane];
    for (uint32_t i = lane * 16u; i < T + 12u; i += 1024u) *(lptr_x4)(stage + i) = u32x4{0u, 0u, 0u, 0u};
#else
    // 16-byte pieces: aligned LDS reads, global stores at whatever alignment the stream position has (unaligned vector
    // stores cost nothing measurable here; the lines are completed in L2 by the neighbouring sub-tiles' flushes); the last
    // T mod 16 bytes one by one.  T <= ENC_STG: at most three pieces per lane.
    typedef u32x4 u32x4_una __attribute__((aligned(1)));
    typedef __attribute__((address_space(1))) u32x4_una* gptr128u;
    const uint32_t nb = T >> 4;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if (nb <= 64u) {  // run-heavy content: at most one piece per lane
        if ((uint32_t)lane < nb) {
#ifndef SQOA_DBG_NOSTORE
            ((gptr128u)g)[lane] = *(lptr_x4)(stage + 16u * lane);
#endif
            *(lptr_x4)(stage + 16u * lane) = zero;
        }
    } else
#pragma unroll
    for (int k = 0; k < (ENC_STG + 1023) / 1024; ++k) {
        const uint32_t u = (uint32_t)lane + 64u * k;
        if (u < nb) {
#ifdef SQOA_DBG_NOSTORE
            u32x4 vv = *(lptr_x4)(stage + 16u * u);
            asm volatile("" :: "v"(vv), "v"(g));
#else
            ((gptr128u)g)[u] = *(lptr_x4)(stage + 16u * u);
#endif
            *(lptr_x4)(stage + 16u * u) = zero;
        }
    }
    const uint32_t t0 = nb << 4;
    if ((uint32_t)lane < T - t0) g[t0 + lane] = stage[t0 + lane];
    // the partial piece and the 12 bytes of OR window behind it
    if (lane < 2) *(lptr_x4)(stage + t0 + 16u * lane) = zero;
#endif
}

// 2 KB (1.5 KB: `half_second`, 3-byte pixels) of a sub-tile's pixels from global memory straight into the wave's landing buffer in
// LDS: two global_load_lds_dwordx4, lane l -> bytes [16 l, 16 l + 16) and [1024 + 16 l, ...).  Hand-written, for two reasons.  (1)
// Issued through the builtin, the compiler knows that something writes LDS asynchronously and puts an s_waitcnt vmcnt(0) in front of
// EVERY later LDS access it cannot prove disjoint from the landing buffer -- the chain posts and polls, the run table, the staging
// ORs: the "prefetch" of the next sub-tile was waited for at the first LDS instruction behind it, a full memory latency on every
// sub-tile.  The one place that reads the landing buffer (take / take8) does its own s_waitcnt vmcnt(0).  (2) The per-lane source
// address was a loop invariant that got spilled; its reload (scratch_load + vmcnt(0)) in front of the loads drained the stores of the
// flush just before.  Here the address is an SGPR base + 16 * lane, the lane number recomputed on the spot.
SQOA_DEV void lds_dma(unsigned long long gaddr, lptr8 lds_dst, bool half_second) {
    const uint32_t lds_addr = bcast_first((uint32_t)(uintptr_t)lds_dst);
    uint32_t keep, off;
    if (!half_second) {
        asm volatile(
            "v_mbcnt_lo_u32_b32 %[off], -1, 0\n\t"
            "v_mbcnt_hi_u32_b32 %[off], -1, %[off]\n\t"
            "v_lshlrev_b32 %[off], 4, %[off]\n\t"
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[lds]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"                       // (every earlier read of the landing buffer has returned)
            "global_load_lds_dwordx4 %[off], %[base]\n\t"
            "global_load_lds_dwordx4 %[off], %[base] offset:1024\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep), [off] "=&v"(off) : [base] "s"(gaddr), [lds] "s"(lds_addr) : "memory");
    } else {
        asm volatile(
            "v_mbcnt_lo_u32_b32 %[off], -1, 0\n\t"
            "v_mbcnt_hi_u32_b32 %[off], -1, %[off]\n\t"
            "v_lshlrev_b32 %[off], 4, %[off]\n\t"
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[lds]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "global_load_lds_dwordx4 %[off], %[base]\n\t"
            "s_mov_b32 vcc_hi, exec_hi\n\t"                   // (EXEC goes back to what it was, whatever that is: kept in VCC, which the block clobbers)
            "s_mov_b32 exec_hi, 0\n\t"                        // lanes 0..31: bytes [1024, 1536)
            "global_load_lds_dwordx4 %[off], %[base] offset:1024\n\t"
            "s_mov_b32 exec_hi, vcc_hi\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep), [off] "=&v"(off) : [base] "s"(gaddr), [lds] "s"(lds_addr) : "memory", "vcc");
    }
}

// ---- SWAR main pass: four 64-pixel rows at a time, one byte lane per row -------------------------------
// gfx950 issues add / sub / logic / right-shift / bitop3 on VGPR or literal operands at 1.7x the rate of every other
// VALU instruction (tools/ubench/valu_table.hip), and the per-pixel op decision (seqoia.h:584-635) is all 8-bit arithmetic.
// So the decision is made for the same lane of FOUR rows at once: the four pixels are transposed into one
// dword per channel (v_perm_b32), channel deltas, the vg_r / vg_b terms, the range tests and the LUMA
// byte fields are computed bytewise inside dwords with plain and/or/add/sub/xor, and only the final
// assembly of each pixel's bytes and its LDS store is done per row.
struct RunOut { uint32_t lo, n, big; };
// Bytes owned by a pixel that repeats its predecessor (seqoia.h:544-561, :640-642): 0xfd on every 512th pixel
// of a run and on a trailing run, the flush (owned by the run's last pixel) otherwise; flushes longer than one
// byte only reserve their space (big = q << 8 | last byte).
template <bool QOI>
SQOA_DEV RunOut run_row(unsigned long long e, uint32_t crun, unsigned long long enext, bool end_lane, int lane,
                        unsigned long long le_mask) {
    const bool iseq = (e >> lane) & 1ull;
    const unsigned long long below = ~e & le_mask;
    const uint32_t k = below ? (uint32_t)lane - (63u - (uint32_t)__builtin_clzll(below)) : crun + lane + 1u;
    const uint32_t r = QOI ? k % 62u : k & 511u;   // seqoia.h:411-412
    const bool fd = iseq && (r == 0u || end_lane);
    const bool fl = iseq && !fd && !((enext >> lane) & 1ull);
    const uint32_t q = QOI ? 0u : ((r - 1u) * 1075u) >> 16;  // (r-1)/61 for r <= 511; QOI remainders are <= 61
    const uint32_t last = fd ? 0xfdu : 0xc0u | (r - 1u - 61u * q);
    RunOut o;
    o.lo = (fd || (fl && q == 0u)) ? last : 0u;
    o.n = fd ? 1u : fl ? q + 1u : 0u;
    o.big = (fl && q != 0u) ? (q << 8 | last) : 0u;
    return o;
}

struct SwarGroup { uint32_t W0, W1, W2, W3, AH, b0, Nn; };  // bytes 0..3 and 4 of every pixel's literal op (byte i = row i)

// SQOA-native op decision (seqoia.h:606-634) for four pixels at once, bytewise: the channel planes R G B A, the channel
// deltas D*, and the biased terms XR = vg_r + 8, XB = vg_b + 8, XG = vg + 32, XA = va + 16 (seqoia.h:585-590).
// H = 0x80808080, L = 0x7f7f7f7f: literals, or VGPRs the caller keeps them in (an SGPR operand makes a fast-class
// instruction a slow one, and the three-operand forms cannot take a literal)
SQOA_DEV SwarGroup sqoa_ops(uint32_t R, uint32_t G, uint32_t B, uint32_t A, uint32_t DR, uint32_t DG, uint32_t DB, uint32_t DA,
                            uint32_t XR, uint32_t XB, uint32_t XG, uint32_t XA, uint32_t H = 0x80808080u, uint32_t L = 0x7f7f7f7fu) {
    const uint32_t MF0 = 0xf0f0f0f0u, MC0 = 0xc0c0c0c0u, ME0 = 0xe0e0e0e0u, M0F = 0x0f0f0f0fu;
    const uint32_t C60 = 0x60606060u, CFE = 0xfefefefeu;
    auto nz = [&](uint32_t x) { return (((x & L) + L) | x) & H; };  // 0x80 in every non-zero byte
    SwarGroup o;
    const uint32_t bad = ((XR | XB) & MF0) | (XG & MC0) | (XA & ME0);       // non-zero byte: LUMA does not fit (seqoia.h:606-611)
    const uint32_t z0 = nz(DR | DG | DB | DA), z1 = nz(bad), z2 = nz(DA);
    const uint32_t b0 = z0 >> 7, b1 = z1 >> 7, b2 = z2 >> 7;   // 1 = differs from its predecessor; needs RGB/RGBA; alpha moved
    // 0xff where set (0x80 - 1 = 0x7f, | 0x80)
    const uint32_t ne_mask = (z0 - b0) | z0, raw_mask = (z1 - b1) | z1, na_mask = (z2 - b2) | z2;
    const uint32_t L0 = XG | H;                                  // LUMA byte 0 (seqoia.h:612)
    const uint32_t L1 = ((XR << 4) & MF0) | (XB & M0F);          // LUMA byte 1 (seqoia.h:614)
    const uint32_t L2 = (XA | C60) & na_mask;                    // ALPHA suffix (seqoia.h:617)
    const uint32_t TG = CFE | b2;                                // RGB / RGBA tag (seqoia.h:622)
    // the op's bytes, already chosen between LUMA[+ALPHA] and RGB[A]; zero for repeated pixels
    o.W0 = ((L0 & ~raw_mask) | (TG & raw_mask)) & ne_mask;
    o.W1 = ((L1 & ~raw_mask) | (R & raw_mask)) & ne_mask;
    o.W2 = ((L2 & ~raw_mask) | (G & raw_mask)) & ne_mask;
    o.W3 = B & raw_mask & ne_mask;
    o.AH = A & na_mask & raw_mask;                   // 5th byte of RGBA
    o.b0 = b0;
    o.Nn = (((b0 & b1) + b0) << 1) + (b0 & b2);      // bytes per pixel: 2/3 LUMA, 4/5 RGB[A], 0 if repeated
    return o;
}

// Channel deltas mod 256 of four pixels against their predecessors (seqoia.h:585-588) and vg_r / vg_b / vg / va with the
// range-test bias folded in (seqoia.h:589-590, :606-611), then the op decision.
SQOA_DEV SwarGroup sqoa_group(uint32_t R, uint32_t G, uint32_t B, uint32_t A, uint32_t Rp, uint32_t Gp, uint32_t Bp, uint32_t Ap,
                              uint32_t H, uint32_t L) {
    const uint32_t C08 = 0x08080808u, C20 = 0x20202020u, C10 = 0x10101010u;
    auto sub8 = [&](uint32_t a, uint32_t b) { return ((a | H) - (b & L)) ^ (~(a ^ b) & H); };
    const uint32_t DR = sub8(R, Rp), DG = sub8(G, Gp), DB = sub8(B, Bp), DA = sub8(A, Ap);
    const uint32_t DGm = ((DG | H) - C08) ^ ((DG & H) ^ H);                 // vg - 8
    const uint32_t XR = sub8(DR, DGm), XB = sub8(DB, DGm);                  // vg_r + 8, vg_b + 8
    const uint32_t XG = ((DG & L) + C20) ^ (DG & H), XA = ((DA & L) + C10) ^ (DA & H);  // vg + 32, va + 16
    return sqoa_ops(R, G, B, A, DR, DG, DB, DA, XR, XB, XG, XA, H, L);
}

// The same for QOI-compat pixels that miss the colour index (seqoia.h:573-634): alpha moved -> RGBA; else DIFF if vr, vg, vb in
// [-2, 1]; else LUMA; else RGB.  (Index hits are patched in afterwards: one byte, the slot.)
SQOA_DEV SwarGroup qoi_group(uint32_t R, uint32_t G, uint32_t B, uint32_t A, uint32_t Rp, uint32_t Gp, uint32_t Bp, uint32_t Ap,
                             uint32_t H, uint32_t L) {
    const uint32_t C08 = 0x08080808u, C20 = 0x20202020u, C02 = 0x02020202u, C40 = 0x40404040u, CFE = 0xfefefefeu, ONE = 0x01010101u;
    const uint32_t MF0 = 0xf0f0f0f0u, MC0 = 0xc0c0c0c0u, M0F = 0x0f0f0f0fu, MFC = 0xfcfcfcfcu;
    auto sub8 = [&](uint32_t a, uint32_t b) { return ((a | H) - (b & L)) ^ (~(a ^ b) & H); };
    auto nz = [&](uint32_t x) { return (((x & L) + L) | x) & H; };  // 0x80 in every non-zero byte
    auto full = [&](uint32_t z) { return (z - (z >> 7)) | z; };     // 0x80 -> 0xff
    const uint32_t DR = sub8(R, Rp), DG = sub8(G, Gp), DB = sub8(B, Bp), DA = sub8(A, Ap);
    const uint32_t DGm = ((DG | H) - C08) ^ ((DG & H) ^ H);                 // vg - 8
    const uint32_t XR = sub8(DR, DGm), XB = sub8(DB, DGm);                  // vg_r + 8, vg_b + 8
    const uint32_t XG = ((DG & L) + C20) ^ (DG & H);                        // vg + 32
    const uint32_t ER = ((DR & L) + C02) ^ (DR & H), EG = ((DG & L) + C02) ^ (DG & H), EB = ((DB & L) + C02) ^ (DB & H);  // v + 2
    const uint32_t z0 = nz(DR | DG | DB | DA);
    const uint32_t ne_mask = full(z0), na_mask = full(nz(DA));
    const uint32_t nd_mask = full(nz((ER | EG | EB) & MFC));                // DIFF does not fit
    const uint32_t nl_mask = full(nz(((XR | XB) & MF0) | (XG & MC0)));      // LUMA does not fit
    const uint32_t df_mask = ~na_mask & ~nd_mask, lu_mask = ~na_mask & nd_mask & ~nl_mask, rw_mask = na_mask | (nd_mask & nl_mask);
    const uint32_t DIFFB = C40 | ((ER << 4) & 0x30303030u) | ((EG << 2) & 0x0c0c0c0cu) | (EB & 0x03030303u);   // seqoia.h:599
    const uint32_t L0 = XG | H, L1 = ((XR << 4) & MF0) | (XB & M0F);
    SwarGroup o;
    o.W0 = ((DIFFB & df_mask) | (L0 & lu_mask) | (CFE & rw_mask) | (ONE & na_mask)) & ne_mask;  // 0xfe | 1 = 0xff
    o.W1 = ((L1 & lu_mask) | (R & rw_mask)) & ne_mask;
    o.W2 = G & rw_mask & ne_mask;
    o.W3 = B & rw_mask & ne_mask;
    o.AH = A & na_mask;
    o.b0 = z0 >> 7;
    o.Nn = (((ONE & df_mask) | ((ONE << 1) & lu_mask) | ((ONE << 2) & rw_mask)) + (ONE & na_mask)) & (ne_mask & (ONE * 7u));   // DIFF 1, LUMA 2, RGB 4, RGBA 5
    return o;
}

#ifndef SQOA_ENC_Q_RTN   // QOI-compat probes: 2 = the nearest lower lane on my slot from one ds_max_rtn_u32, last writers by plain writes (the LDS
#define SQOA_ENC_Q_RTN 3  // applies an instruction's lanes in ascending order: tools/ubench/lds_atomic_order.hip); 0 = per-slot 64-bit lane masks, ds_max_u64
#endif
static_assert(SQOA_ENC_Q_RTN == 0 || SQOA_ENC_Q_RTN == 2 || SQOA_ENC_Q_RTN == 3, "SQOA_ENC_Q_RTN");
#ifndef SQOA_ENC_Q_PIPE
#define SQOA_ENC_Q_PIPE 1
#endif
// The index probe (seqoia.h:563-572) of one row of 64 consecutive pixels, lane = pixel, rows in order: a pixel's candidate is the
// most recent earlier pixel of the row on its slot (per-slot lane masks in LDS), else the slot as the rows before left it.
// Returns 0x80 | slot where the pixel hits, 0 where it misses or repeats its predecessor (`ne` false: run pixels do not probe).
SQOA_DEV uint32_t qoi_probe_row(uint32_t px, bool ne, uint32_t* tabw, unsigned long long* smask, int lane, unsigned long long le_mask) {
    const uint32_t h = qoi_hash(px);
    uint32_t cand = tabw[h];
    if (ne) __hip_atomic_fetch_or(&smask[h], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const unsigned long long m = __hip_atomic_load(&smask[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const unsigned long long lower = m & (le_mask >> 1);
    const int from = lower ? 63 - __builtin_clzll(lower) : lane;
    const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)px);
    if (lower) cand = near;
    if (ne && (m & ~le_mask) == 0ull) { tabw[h] = px; smask[h] = 0; }  // last writer of the slot in this row
    return (ne && cand == px) ? 0x80u | h : 0u;
}

// Phase A: per-channel byte planes of four rows, op decision for every pixel, and the rows' equality masks.
template <bool QOI>
SQOA_DEV void swar_classify(const uint32_t (&cur)[ENC_S], uint32_t pb, SwarGroup (&grp)[ENC_S / 4], unsigned long long (&E)[ENC_S]) {
    // constants as literals: an SGPR operand would put the add / and / or they feed into the slow VALU class
    const uint32_t H = 0x80808080u, L = 0x7f7f7f7fu;
    const uint32_t C08 = 0x08080808u, C20 = 0x20202020u, C10 = 0x10101010u;
    const uint32_t MF0 = 0xf0f0f0f0u, MC0 = 0xc0c0c0c0u, M0F = 0x0f0f0f0fu;
    const uint32_t CFE = 0xfefefefeu;
    auto sub8 = [&](uint32_t a, uint32_t b) { return ((a | H) - (b & L)) ^ (~(a ^ b) & H); };
    auto nz = [&](uint32_t x) { return (((x & L) + L) | x) & H; };  // 0x80 in every non-zero byte
    uint32_t c0 = pb;  // pixel in front of the group's first row
#pragma unroll
    for (int g = 0; g < ENC_S / 4; ++g) {
        const uint32_t P0 = cur[4 * g], P1 = cur[4 * g + 1], P2 = cur[4 * g + 2], P3 = cur[4 * g + 3];
        // one dword per channel, byte i = row i
        const uint32_t a01 = __builtin_amdgcn_perm(P1, P0, 0x05010400u), b01 = __builtin_amdgcn_perm(P1, P0, 0x07030602u);
        const uint32_t a23 = __builtin_amdgcn_perm(P3, P2, 0x05010400u), b23 = __builtin_amdgcn_perm(P3, P2, 0x07030602u);
        const uint32_t R = __builtin_amdgcn_perm(a23, a01, 0x05040100u), G = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
        const uint32_t B = __builtin_amdgcn_perm(b23, b01, 0x05040100u), A = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
        // the same for the predecessor pixels: lane-1, and for lane 0 the last pixel of the row above
        const uint32_t Rp = shift_up1(R, bcast_lane(R, 63) << 8 | (c0 & 255u));
        const uint32_t Gp = shift_up1(G, bcast_lane(G, 63) << 8 | ((c0 >> 8) & 255u));
        const uint32_t Bp = shift_up1(B, bcast_lane(B, 63) << 8 | ((c0 >> 16) & 255u));
        const uint32_t Ap = shift_up1(A, bcast_lane(A, 63) << 8 | (c0 >> 24));
        c0 = bcast_lane(P3, 63);
        // channel deltas mod 256 (seqoia.h:585-588), vg_r / vg_b (seqoia.h:589-590) with the range-test bias folded in
        const uint32_t DR = sub8(R, Rp), DG = sub8(G, Gp), DB = sub8(B, Bp), DA = sub8(A, Ap);
        const uint32_t DGm = ((DG | H) - C08) ^ ((DG & H) ^ H);                 // vg - 8
        const uint32_t XR = sub8(DR, DGm), XB = sub8(DB, DGm);                  // vg_r + 8, vg_b + 8
        const uint32_t XG = ((DG & L) + C20) ^ (DG & H), XA = ((DA & L) + C10) ^ (DA & H);  // vg + 32, va + 16
        SwarGroup& o = grp[g];
        if constexpr (!QOI) {
            o = sqoa_ops(R, G, B, A, DR, DG, DB, DA, XR, XB, XG, XA);
        } else {
            // QOI-compat ops for a pixel that missed the index (seqoia.h:573-634): alpha moved -> RGBA; else DIFF if
            // vr, vg, vb in [-2, 1]; else LUMA; else RGB
            const uint32_t C02 = 0x02020202u, MFC = 0xfcfcfcfcu, C40 = 0x40404040u, ONE = 0x01010101u;
            const uint32_t ER = ((DR & L) + C02) ^ (DR & H), EG = ((DG & L) + C02) ^ (DG & H), EB = ((DB & L) + C02) ^ (DB & H);  // v + 2
            const uint32_t z0 = nz(DR | DG | DB | DA), z2 = nz(DA);
            const uint32_t zd = nz((ER | EG | EB) & MFC);                  // 0x80: DIFF does not fit
            const uint32_t zl = nz(((XR | XB) & MF0) | (XG & MC0));        // 0x80: LUMA does not fit
            const uint32_t b0 = z0 >> 7;
            const uint32_t ne_mask = (z0 - b0) | z0;
            const uint32_t na_mask = (z2 - (z2 >> 7)) | z2;                // RGBA
            const uint32_t nd_mask = (zd - (zd >> 7)) | zd, nl_mask = (zl - (zl >> 7)) | zl;
            const uint32_t df_mask = ~na_mask & ~nd_mask;                  // DIFF
            const uint32_t lu_mask = ~na_mask & nd_mask & ~nl_mask;        // LUMA
            const uint32_t rw_mask = na_mask | (nd_mask & nl_mask);        // RGB or RGBA: r g b follow the tag
            // seqoia.h:599; the shifts run across byte lanes, so every field is masked to its own two bits
            const uint32_t DIFFB = C40 | ((ER << 4) & 0x30303030u) | ((EG << 2) & 0x0c0c0c0cu) | (EB & 0x03030303u);
            const uint32_t L0 = XG | H, L1 = ((XR << 4) & MF0) | (XB & M0F);
            o.W0 = ((DIFFB & df_mask) | (L0 & lu_mask) | (CFE & rw_mask) | (ONE & na_mask)) & ne_mask;  // 0xfe | 1 = 0xff
            o.W1 = ((L1 & lu_mask) | (R & rw_mask)) & ne_mask;
            o.W2 = G & rw_mask & ne_mask;
            o.W3 = B & rw_mask & ne_mask;
            o.AH = A & na_mask;
            o.b0 = b0;
            // bytes: DIFF 1, LUMA 2, RGB 4, RGBA 5
            o.Nn = (((ONE & df_mask) | ((ONE << 1) & lu_mask) | ((ONE << 2) & rw_mask)) + (ONE & na_mask)) & (ne_mask & (ONE * 7u));
        }
        const uint32_t b0 = o.b0;
#pragma unroll
        for (int i = 0; i < 4; ++i) E[4 * g + i] = __ballot((b0 & (1u << (8 * i))) == 0u);
    }
}

// Phase B: run bytes, byte offsets (two DPP prefix sums per four rows) and staging into LDS.
template <bool QOI>
SQOA_DEV uint32_t swar_emit(const uint32_t (&cur)[ENC_S], SwarGroup (&grp)[ENC_S / 4], const unsigned long long (&E)[ENC_S], uint32_t c_in,
                            uint32_t pa, bool has_after, bool eof, lptr8 stage, int lane, unsigned long long le_mask, uint32_t* tabw,
                            unsigned long long* smask, [[maybe_unused]] uint32_t* pstamp, [[maybe_unused]] uint32_t& qrow64) {
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    const uint32_t stage_addr = bcast_first((uint32_t)(uintptr_t)stage);
    const uint32_t sink_addr = stage_addr + 4u * lane;  // a pixel that emits nothing ORs its zeros into a dword of the lane's own, anywhere in the buffer
    uint32_t cr[ENC_S + 1];  // pixels repeating their predecessor immediately before each row
    cr[0] = c_in;
#pragma unroll
    for (int s = 0; s < ENC_S; ++s) cr[s + 1] = E[s] == ~0ull ? cr[s] + 64u : (uint32_t)__builtin_clzll(~E[s]);
    uint32_t sbase = 0;
#pragma unroll
    for (int g = 0; g < ENC_S / 4; ++g) {
        const SwarGroup& G = grp[g];
        uint32_t Nn = G.Nn, W0 = G.W0;
        uint32_t hitv[4] = {0, 0, 0, 0};  // QOI: 0x100 | slot where the pixel hits the colour index
        if constexpr (QOI) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // index probe (seqoia.h:563-572), the 64 pixels of a row at once, rows in order
                const uint32_t px = cur[4 * g + i];
                const bool ne = (G.b0 >> (8 * i)) & 1u;
                const uint32_t h = qoi_hash(px);
                uint32_t cand = tabw[h];  // slot as the previous rows left it
#if SQOA_ENC_Q_RTN
                // (one ds_max_rtn_u32 of a stamp that grows with row and lane: what the lanes BELOW me on my slot left -- the L8 path's comment)
                const uint32_t prev = ne ? __hip_atomic_fetch_max(&pstamp[h], qrow64 + (uint32_t)lane + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
                const bool lower = prev > qrow64;
                const int from = lower ? (int)(prev - qrow64 - 1u) : lane;
                const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)px);
                if (lower) cand = near;
                if (ne) tabw[h] = px;                                          // (of the lanes on one slot the highest stays)
                qrow64 += 64u;
#else
                if (ne) __hip_atomic_fetch_or(&smask[h], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long m = __hip_atomic_load(&smask[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long lower = m & (le_mask >> 1);  // earlier lanes of this row on my slot
                const int from = lower ? 63 - __builtin_clzll(lower) : lane;
                const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)px);
                if (lower) cand = near;
                if (ne && (m & ~le_mask) == 0ull) { tabw[h] = px; smask[h] = 0; }  // last writer of the slot in this row
#endif
                const bool hit = ne && cand == px;
                hitv[i] = hit ? 0x100u | h : 0u;
                if (hit) Nn = (Nn & ~(0xffu << (8 * i))) | 1u << (8 * i);  // QOI_OP_INDEX: one byte
            }
        }
        // pixels that repeat their predecessor: run bookkeeping per row, skipped for rows without any;
        // their (single) run byte joins byte plane 0, which is zero there
        uint32_t rbig[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = 4 * g + i;
            const unsigned long long e = E[s];
            if (!QOI && e == ~0ull) {
                // the whole row continues one run: at most one 0xfd (the lane where k mod 512 == 0) and, on lane 63,
                // the flush or the trailing 0xfd -- all of it wave-uniform arithmetic
                const unsigned long long nb = s + 1 < ENC_S ? (E[(s + 1) % ENC_S] & 1ull) : (unsigned long long)(has_after && pa == bcast_lane(cur[ENC_S - 1], 63));
                const bool end_px = s == ENC_S - 1 && eof;
                const uint32_t r0 = cr[s] & 511u, r63 = (r0 + 64u) & 511u;
                const uint32_t fd_lane = 511u - r0;                       // < 64 iff the run crosses a multiple of 512 here
                uint32_t n63 = 0, lo63 = 0, big63 = 0;
                if (r63 != 0u && (end_px || !nb)) {
                    const uint32_t q = end_px ? 0u : ((r63 - 1u) * 1075u) >> 16;
                    const uint32_t last = end_px ? 0xfdu : 0xc0u | (r63 - 1u - 61u * q);
                    n63 = q + 1u;
                    lo63 = q ? 0u : last;
                    big63 = q ? (q << 8 | last) : 0u;
                }
                const bool is63 = lane == 63, isfd = (uint32_t)lane == fd_lane;
                W0 |= (isfd ? 0xfdu : is63 ? lo63 : 0u) << (8 * i);
                rbig[i] = is63 ? big63 : 0u;
                Nn += (isfd ? 1u : is63 ? n63 : 0u) << (8 * i);
            } else if (e != 0ull) {
                const unsigned long long nb = s + 1 < ENC_S ? (E[(s + 1) % ENC_S] & 1ull) : (unsigned long long)(has_after && pa == bcast_lane(cur[ENC_S - 1], 63));
                const bool end_lane = s == ENC_S - 1 && eof && lane == 63;  // the image's last pixel
                const RunOut ro = run_row<QOI>(e, cr[s], e >> 1 | nb << 63, end_lane, lane, le_mask);
                W0 |= ro.lo << (8 * i);
                rbig[i] = ro.big;
                Nn += ro.n << (8 * i);
            }
        }
        const uint32_t i01 = wave_incl_scan(__builtin_amdgcn_perm(Nn, Nn, 0x0c010c00u));
        const uint32_t i23 = wave_incl_scan(__builtin_amdgcn_perm(Nn, Nn, 0x0c030c02u));
        const uint32_t t01 = bcast_lane(i01, 63), t23 = bcast_lane(i23, 63);
        const uint32_t rowbase[4] = {sbase, sbase + (t01 & 0xffffu), sbase + (t01 & 0xffffu) + (t01 >> 16),
                                     sbase + (t01 & 0xffffu) + (t01 >> 16) + (t23 & 0xffffu)};
        sbase = rowbase[3] + (t23 >> 16);
        // back from byte planes to one dword per row: a 4x4 byte transpose (8 v_perm_b32, same selectors as the way in)
        const uint32_t w01 = __builtin_amdgcn_perm(G.W1, W0, 0x05010400u), w23 = __builtin_amdgcn_perm(G.W3, G.W2, 0x05010400u);
        const uint32_t u01 = __builtin_amdgcn_perm(G.W1, W0, 0x07030602u), u23 = __builtin_amdgcn_perm(G.W3, G.W2, 0x07030602u);
        uint32_t obig[4];
        const uint32_t lo4[4] = {__builtin_amdgcn_perm(w23, w01, 0x05040100u), __builtin_amdgcn_perm(w23, w01, 0x07060302u),
                                 __builtin_amdgcn_perm(u23, u01, 0x05040100u), __builtin_amdgcn_perm(u23, u01, 0x07060302u)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t n = (Nn >> (8 * i)) & 255u;
            const uint32_t incl = ((i < 2 ? i01 : i23) >> (16 * (i & 1))) & 0xffffu;
            const uint32_t o = (rowbase[i] + stage_addr) + incl - n;  // LDS byte address (the base is 16-byte aligned)
            uint32_t lo = lo4[i], hi = (G.AH >> (8 * i)) & 255u;
            if (QOI && hitv[i]) { lo = hitv[i] & 63u; hi = 0; }
            const uint32_t sh = (o & 3u) * 8u;
            const unsigned long long v = ((unsigned long long)hi << 32 | lo) << sh;
            // pixels that emit nothing OR their zero into a dword of the lane's own instead of being masked off: no exec
            // juggling, and no 64 lanes OR-ing the same dword
            // (LDS addresses: the staging buffer is 16-byte aligned, so its base rides on the wave-uniform row base for free)
            lds_u32 p = (lds_u32)(uintptr_t)(n ? (o & ~3u) : sink_addr);
            __hip_atomic_fetch_or(p, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_or(p + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            obig[i] = o;
        }
        if (__ballot((rbig[0] | rbig[1] | rbig[2] | rbig[3]) != 0u) != 0ull) {  // run remainder >= 62: q x 0xfc, then the last byte (rare)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (rbig[i]) {
                    const uint32_t q = rbig[i] >> 8, last = rbig[i] & 255u;
#pragma nounroll
                    for (uint32_t t = 0; t <= q; ++t) {
                        const uint32_t ob = obig[i] + t;
                        __hip_atomic_fetch_or((lds_u32)(uintptr_t)(ob & ~3u), (t == q ? last : 0xfcu) << ((ob & 3u) * 8u),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        }
    }
    return sbase;
}

// 64 VGPRs = 8 waves per SIMD = four 8-wave images per CU: the 1024 images of the headline batch are all resident at once
// (at 79 VGPRs a CU held three, and the last quarter of the batch ran on a third of the machine).
#ifndef SQOA_ENC_WPE
#define SQOA_ENC_WPE 8
#endif
#ifndef SQOA_ENC_WPE_QOI
#define SQOA_ENC_WPE_QOI 4
#endif
// (one wave per image, W = 1: 24 workgroups of 6.7 KB of LDS fit a CU, six waves per SIMD -- asking for eight only earned a
// "failed to meet occupancy target" from the compiler; the staging buffer, the landing buffer and the run-byte table are 6.6 KB)
#ifndef SQOA_ENC_WPE_ATTR
#define SQOA_ENC_WPE_ATTR __attribute__((amdgpu_waves_per_eu(QOI ? SQOA_ENC_WPE_QOI : (W == 1 ? 6 : SQOA_ENC_WPE), QOI ? 6 : (W == 1 ? 6 : SQOA_ENC_WPE))))
#endif
template <int W, bool QOI>
__global__ __launch_bounds__(W * 64) SQOA_ENC_WPE_ATTR void encode_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                        const int* __restrict__ job_ids,
                                                        sqoa_hip_result* __restrict__ results,
                                                        const uint32_t* __restrict__ qtab_in) {   // QOI strip jobs: the colour index entering job i at [64 * i, 64 * i + 64); else null
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[W][ENC_STG];
    // (QOI-compat, L8 hot loop: two landing buffers per wave, taken in turn -- the sub-tile being encoded stays readable in LDS for
    // the row-by-row index probe while the next one lands -- and 512 bytes for the probe's verdicts, one byte per pixel)
    // (the one-wave shape for many small images, W = 1, keeps the row layout for QOI-compat: its images are a handful of sub-tiles,
    // and 4.6 KB more LDS per workgroup cost it more workgroups per CU than the layout gains: 4096 x 64 x 64 icons 0.092 -> 0.100 ms)
    constexpr bool QL8 = QOI && W > 1;
    __shared__ __attribute__((aligned(16))) uint8_t s_pre[QL8 ? 2 * W : W][ENC_PRE];
    __shared__ __attribute__((aligned(16))) uint8_t s_hit[QL8 ? W : 1][QL8 ? ENC_SUB : 16];
    __shared__ unsigned long long s_run[2][W];
#if SQOA_ENC_LOOKBACK
    __shared__ unsigned long long s_pos[2][W];   // [(sub-tile / W) & 1][sub-tile % W]: tag = sub-tile + 1 | the position BEHIND its last byte
    __shared__ unsigned long long s_agg[2][W];   // ... | its byte count (pos_lookback)
#else
    __shared__ unsigned long long s_pos[W];
#endif
    // QOI colour index (seqoia.h:563-582): per-wave working table, last-writer stamps and per-slot lane
    // masks, plus each wave's published end-of-sub-tile table (double-buffered by tile parity)
    constexpr int QW = QOI ? W : 1;
    __shared__ uint32_t s_tabw[QW][64];
    __shared__ uint32_t s_stamp[QW][64];
    // L8 hot loop without SQOA_ENC_Q_RTN == 2: per slot, (position + 1) << 32 | pixel of the sub-tile's last writer (with it: one row of zeros that
    // the table hop of the row-layout path reads and writes back)
    constexpr int QLW = QL8 ? (SQOA_ENC_Q_RTN >= 2 ? 1 : W) : 1;
    __shared__ unsigned long long s_last[QLW][QL8 ? 64 : 1];
    __shared__ unsigned long long s_smask[QW][64];
    // L8 hot loop (SQOA_ENC_Q_RTN): per slot, the stamp (row number * 64 + lane + 1, growing for the wave's lifetime) of the last pixel that took it
    __shared__ uint32_t s_pstamp[QOI && SQOA_ENC_Q_RTN ? W : 1][QOI && SQOA_ENC_Q_RTN ? 64 : 1];
    // ... and (SQOA_ENC_Q_RTN == 2) its pixel: with the stamps the sub-tile's "last writer per slot" for the table hop, by plain writes
    __shared__ uint32_t s_lastpx[QL8 && SQOA_ENC_Q_RTN == 2 ? W : 1][QL8 && SQOA_ENC_Q_RTN == 2 ? 64 : 1];
    // L8 hot loop (SQOA_ENC_Q_RTN == 3): per slot, stamp << 32 | pixel of the last pixel that took it
    __shared__ unsigned long long s_pst64[QL8 && SQOA_ENC_Q_RTN == 3 ? W : 1][QL8 && SQOA_ENC_Q_RTN == 3 ? 64 : 1];
    __shared__ uint32_t s_tabpub[QOI ? 2 : 1][QW][64];
    __shared__ uint32_t s_tabtag[2][W];
    // L8 path: run bytes of the 8 consecutive pixels a lane owns, by their 8 "repeats its predecessor" flags: byte j =
    // 0xc0 | (c - 1) where c = repeated pixels ending at pixel j inside the lane (the byte a run flushes as when it ends
    // there and began inside the lane, seqoia.h:554-561), 0 where pixel j does not repeat
    __shared__ __attribute__((aligned(8))) uint32_t s_runtab[256][2];

    SQOA_PLACEMENT_PAD();
    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    const int jid = job_ids[blockIdx.x];
    if (jid < 0) return;   // a skipped row of a strip table (sqoa_strips.hip)
    const sqoa_hip_job job = jobs[jid];
    // the result row as every launch must find it (no copy in front of the kernels: sqoa_runtime.hip, plan_enqueue); the barrier
    // below orders it in front of whatever a wave writes there later
    if (threadIdx.x == 0) { results[jid].len = 0; results[jid].status = SQOA_HIP_ST_OK; }

    const uint32_t N = job.desc.width * job.desc.height;
    const int dch = job.desc.channels;
    const bool mono = dch < 3;
    const int ch = (mono ? 1 : 3) + ((dch & 1) ? 0 : 1);  // input stride == header byte (seqoia.h:486, :505)
    gcptr8 src = (gcptr8)job.src;
    gptr8 dst = (gptr8)job.dst;
    // Strip jobs (sqoa_hip.h): a job may hold a stretch of an image's pixels only -- no header, no end marker, the pixel in front
    // of src as the first pixel's predecessor.  From callers SQOA-native only (the plan refuses them for QOI-compat: the colour
    // index crosses the cut); the plan's own QOI-compat strips (sqoa_strips.hip) come with the index entering each of them.
    const uint32_t sflags = job.reserved;
    const bool strip_last = !(sflags & SQOA_HIP_JOB_NOEND);   // the job ends the image: end marker, and a trailing run is the single 0xfd of seqoia.h:640-642
    const uint32_t hdr_len = (sflags & SQOA_HIP_JOB_NOHEADER) ? 0u : QOI ? kHeader : kHeader + 1;
    const uint32_t nsub = (N + ENC_SUB - 1) / ENC_SUB;
    constexpr uint32_t M = QOI ? 62u : 512u;  // seqoia.h:411-412

    const lptr8 stage = (lptr8)s_stage[w];
#if SQOA_ENC_LOOKBACK
    if (lane == 0) { s_run[0][w] = 0; s_run[1][w] = 0; s_pos[0][w] = 0; s_pos[1][w] = 0; s_agg[0][w] = 0; s_agg[1][w] = 0; s_tabtag[0][w] = 0; s_tabtag[1][w] = 0; }
#else
    if (lane == 0) { s_run[0][w] = 0; s_run[1][w] = 0; s_pos[w] = 0; s_tabtag[0][w] = 0; s_tabtag[1][w] = 0; }
#endif
    if (QOI) { s_tabw[w % QW][lane] = 0; s_stamp[w % QW][lane] = 0; s_smask[w % QW][lane] = 0; if (QL8) s_last[w % QLW][lane] = 0; }
    if (QOI && SQOA_ENC_Q_RTN) s_pstamp[w % QW][lane] = 0;
    if (QL8 && SQOA_ENC_Q_RTN == 3) s_pst64[w % QW][lane] = 0;
    [[maybe_unused]] uint32_t qrow64 = 0;   // (SQOA_ENC_Q_RTN: 64 x the rows this wave has probed)
    for (uint32_t i = lane * 16u; i < (uint32_t)ENC_STG; i += 1024u) *(lptr_x4)(stage + i) = u32x4{0u, 0u, 0u, 0u};
    {
        for (uint32_t e8 = threadIdx.x; e8 < 256u; e8 += W * 64) {
            unsigned long long v = 0;
            uint32_t c = 0;
            for (int k = 0; k < 8; ++k) {
                c = (e8 >> k) & 1u ? c + 1u : 0u;
                if (c) v |= (unsigned long long)(0xbfu + c) << (8 * k);
            }
            s_runtab[e8][0] = (uint32_t)v;
            s_runtab[e8][1] = (uint32_t)(v >> 32);
        }
    }
    __syncthreads();

    if (w == 0 && (uint32_t)lane < hdr_len) {  // seqoia.h:497-514 (hdr_len = 0: a strip job)
        const uint32_t magic = QOI ? 0x716f6966u : 0x53716f61u;
        const uint32_t word = lane < 4 ? magic : lane < 8 ? job.desc.width : job.desc.height;
        uint8_t b = (uint8_t)(word >> (8 * (3 - (lane & 3))));
        if (lane == 12) b = (uint8_t)ch;
        if (lane == 13) b = job.desc.colorspace;
        if (lane == 14) b = 0x31;
        dst[lane] = b;
    }

#ifdef SQOA_TRACE  // diagnostic builds only: workgroup start / end time and placement in the unused tail of the slot
    unsigned long long* trace = (unsigned long long*)((job.dst + job.dst_cap - 32) & ~7ull);
    if (threadIdx.x == 0) { trace[0] = wall_clock64(); trace[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | (unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32; }
#endif
#ifdef SQOA_TRACE
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#define PHASE(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - ph_t; ph_t = t_; } while (0)
#elif defined(SQOA_ASM_MARK)
#define PHASE(k) asm volatile("; SQOA_PHASE " #k)
#else
#define PHASE(k) do { } while (0)
#endif
    EncState st{0, bcast_first(job.reserved != 0u ? 1u : 0u)};

    uint32_t cur[ENC_S], cur_pb = kPxInit, cur_pa = 0;  // cur_pb / cur_pa: the pixels before / after the sub-tile (wave-uniform)
    bool cur_raw3 = false;
    lptr8 pre = (lptr8)s_pre[w];                   // where the next sub-tile lands
    lptr8 held = (lptr8)s_pre[QL8 ? W + w : w];    // QOI-compat: the buffer the current sub-tile's pixels were taken from
    // The pixels of a wave's NEXT sub-tile travel global -> LDS directly (global_load_lds_dwordx4, any source alignment:
    // tools/ubench/lds_dma_align.hip), two instructions per 512 pixels, and are picked up into VGPRs only when the current
    // sub-tile's have been consumed: nothing is held in registers across a sub-tile, which is what lets four images share a CU
    // (64 VGPRs).  The two neighbour pixels are wave-uniform and come through the scalar cache.  Sub-tiles at the image edges,
    // and 1- / 2-byte pixels, are loaded the plain way and stored to the same buffer as canonical dwords.
    // K (compile time): 0 = anything; 1 = an interior sub-tile (full sub-tiles on both sides) of a 3- or 4-byte image;
    // 2 = an interior sub-tile of a 4-byte image.  The hot loop is instantiated with 1 or 2.
    auto fetch = [&](auto kind_tag, uint32_t j, NbrPx& pb, NbrPx& pa, bool& raw3) {
        constexpr int K = decltype(kind_tag)::value;
        const uint32_t base = j * ENC_SUB;
        const bool interior = K ? true : (base > 0 && base + ENC_SUB < N);
        raw3 = false;
        if (interior && (K == 2 || ch == 4)) {
#ifdef SQOA_ENC_BUILTIN_DMA
            gcptr8 p = src + (size_t)base * 4;
            __builtin_amdgcn_global_load_lds(p + lane * 16, (lds_ptr)pre, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(p + 1024 + lane * 16, (lds_ptr)(pre + 1024), 16, 0, 0);
#else
            lds_dma(job.src + (unsigned long long)base * 4, pre, false);
#endif
            if ((job.src & 3ull) == 0) {  // dword-aligned pixels: one scalar load each
                pb = NbrPx{*(kcptr32)(job.src + (unsigned long long)base * 4 - 4), 0u, 0u, 0u};
                pa = NbrPx{*(kcptr32)(job.src + (unsigned long long)base * 4 + ENC_SUB * 4), 0u, 0u, 0u};
            } else {
                pb = sload_px(job.src + (unsigned long long)base * 4 - 4, 4, 0u);
                pa = sload_px(job.src + (unsigned long long)base * 4 + ENC_SUB * 4, 4, 0u);
            }
        } else if (K != 2 && interior && (K == 1 || ch == 3)) {
#ifdef SQOA_ENC_BUILTIN_DMA
            gcptr8 p = src + (size_t)base * 3;
            __builtin_amdgcn_global_load_lds(p + lane * 16, (lds_ptr)pre, 16, 0, 0);
            if (lane < 32) __builtin_amdgcn_global_load_lds(p + 1024 + lane * 16, (lds_ptr)(pre + 1024), 16, 0, 0);
#else
            lds_dma(job.src + (unsigned long long)base * 3, pre, true);
#endif
            pb = sload_px(job.src + (unsigned long long)base * 3 - 3, 3, 0xff000000u);
            pa = sload_px(job.src + (unsigned long long)base * 3 + ENC_SUB * 3, 3, 0xff000000u);
            raw3 = true;
        } else if (K == 0) {
            uint32_t px[ENC_S];
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) {
                const uint32_t idx = base + s * 64 + lane;
                px[s] = idx < N ? load_px(src, idx, ch) : 0u;
            }
            pb = NbrPx{bcast_first(base ? load_px(src, base - 1, ch) : (sflags & SQOA_HIP_JOB_PREV) ? load_px(src - ch, 0, ch) : kPxInit), 0u, 0u, 0u};
            pa = NbrPx{bcast_first(base + ENC_SUB < N ? load_px(src, base + ENC_SUB, ch) : 0u), 0u, 0u, 0u};
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) ((lptr32)pre)[s * 64 + lane] = px[s];
        }
    };
    // (QOI-compat: the buffer just taken from becomes `held` -- it keeps the sub-tile readable -- and the other one the landing buffer)
    auto swap_buffers = [&]() { if constexpr (QL8) { lptr8 t = held; held = pre; pre = t; } };
    auto take = [&](uint32_t (&px)[ENC_S], bool raw3) {
        // the compiler does not order LDS reads behind LDS-DMA loads by itself
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (raw3) {
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) {
                uint32_t v;
                v = *(lptr32u)(pre + 3 * (s * 64 + lane));  // unaligned LDS read; the 4th byte is the next pixel's
                px[s] = v | 0xff000000u;
            }
        } else {
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) px[s] = ((lptr32)pre)[s * 64 + lane];
        }
        swap_buffers();
    };

    // L8 layout (below): the lane's eight consecutive pixels instead of one pixel of each row
    // (3-byte pixels: the lane's 24 bytes as they lie, px[0..5]; the channel planes are gathered from them directly)
    auto take8 = [&](uint32_t (&px)[ENC_S], bool raw3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (raw3) {
            const u32x2 a = *(lptr_x2)(pre + 24 * lane), b = *(lptr_x2)(pre + 24 * lane + 8),
                        c = *(lptr_x2)(pre + 24 * lane + 16);
            if constexpr (QOI) {
                // QOI-compat: the index probe reads the sub-tile once more, row by row (body8) -- 3-byte pixels would be 8 unaligned LDS
                // reads per lane there, which stall the LDS pipe (4096 x 512 x 512 RGB icons 4.35 ms against 3.01 ms as RGBA).  The lane's
                // 24 bytes become eight pixel dwords (alpha 255) here, go back into the same buffer (every lane has read before any
                // lane writes: in place), and from here on the sub-tile is a 4-byte one.
                auto rgb = [](uint32_t x) { return (x & 0x00ffffffu) | 0xff000000u; };
                px[0] = rgb(a.x); px[1] = rgb(__builtin_amdgcn_alignbyte(a.y, a.x, 3)); px[2] = rgb(__builtin_amdgcn_alignbyte(b.x, a.y, 2)); px[3] = rgb(b.x >> 8);
                px[4] = rgb(b.y); px[5] = rgb(__builtin_amdgcn_alignbyte(c.x, b.y, 3)); px[6] = rgb(__builtin_amdgcn_alignbyte(c.y, c.x, 2)); px[7] = rgb(c.y >> 8);
                *(lptr_x4)(pre + 32 * lane) = u32x4{px[0], px[1], px[2], px[3]};
                *(lptr_x4)(pre + 32 * lane + 16) = u32x4{px[4], px[5], px[6], px[7]};
            } else {
                px[0] = a.x; px[1] = a.y; px[2] = b.x; px[3] = b.y; px[4] = c.x; px[5] = c.y; px[6] = 0; px[7] = 0;
            }
        } else {
            const u32x4 a = *(lptr_x4)(pre + 32 * lane), b = *(lptr_x4)(pre + 32 * lane + 16);
            px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w; px[4] = b.x; px[5] = b.y; px[6] = b.z; px[7] = b.w;
        }
        swap_buffers();
    };

    // QOI-compat, L8 hot loop (j > 0, sub-tiles follow): the colour index entering sub-tile j and the one leaving it.  `writers`
    // marks the sub-tile's last writer per slot (stamps) and puts their pixels into the table to be published; the slots the
    // sub-tile does not write are filled from the table the sub-tile in front published (composition by overwrite).
    auto qoi_table_hop = [&](uint32_t j, [[maybe_unused]] bool l8, auto&& writers) {
        if constexpr (QOI) {
            unsigned long long* last = s_last[w % QLW];
            uint32_t* pub = s_tabpub[(j / W) & 1][w % QW];
            writers(last);
            const uint32_t pp = ((j - 1) / W) & 1, pw = (j - 1) % W;
            for (uint32_t spin = 0;; ++spin) {
                const uint32_t t = __hip_atomic_load(&s_tabtag[pp][pw], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (bcast_first(t) == j) break;
                if (spin >= SPIN_MAX) { st.bad = 1; break; }
                if (st.eager) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(SQOA_ENC_QTAB_SLEEP);
            }
            const uint32_t tin = s_tabpub[pp][pw % QW][lane];
            if (QL8 && SQOA_ENC_Q_RTN == 3 && l8) {
                // (the slots this sub-tile wrote: the ones whose stamp is of this sub-tile; stamp and pixel are one 8-byte entry)
                const unsigned long long e = s_pst64[w % QW][lane];
                pub[lane] = (uint32_t)(e >> 32) > qrow64 ? (uint32_t)e : tin;
                s_tabw[w % QW][lane] = tin;
            } else if (QL8 && SQOA_ENC_Q_RTN == 2 && l8) {
                // (the slots this sub-tile wrote are the ones whose stamp is of this sub-tile; their pixel: the last plain write's)
                pub[lane] = s_pstamp[w % QW][lane] > qrow64 ? s_lastpx[w % QW][lane] : tin;
                s_tabw[w % QW][lane] = tin;
            } else {
                const unsigned long long e = last[lane];
                pub[lane] = (uint32_t)(e >> 32) ? (uint32_t)e : tin;
                s_tabw[w % QW][lane] = tin;
                last[lane] = 0;
            }
            if (lane == 0) __hip_atomic_store(&s_tabtag[(j / W) & 1][w], j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // One sub-tile.  SW (compile time) = the SWAR path: colour image, sub-tile fully inside the image.  The loop over the
    // full sub-tiles of a colour image is instantiated with SW = true only, so the generic one-row-per-step code (mono
    // images, the partial sub-tile at the end of an image) stays out of the hot loop.
    auto body = [&](auto sw_tag, auto mid_tag, uint32_t j, bool more, bool next8 = false) {
        constexpr bool SW = decltype(sw_tag)::value;
        constexpr int KIND = decltype(mid_tag)::value;   // 0: general; 1 / 2: hot-loop instance (2: 4-byte pixels known)
        constexpr bool MID = KIND != 0;                   // j > 0, full sub-tiles follow, `more` is true: no edge cases in this instance
        uint32_t lane_o = (uint32_t)lane;
        asm volatile("" : "+v"(lane_o));                  // (recomputed per sub-tile: two registers less live across the SQOA hot loop)
        const unsigned long long le_mask = (2ull << lane_o) - 1ull;  // bits 0..lane
        if (MID) { __builtin_assume(j > 0); __builtin_assume(more); }
        NbrPx nxt_pb{0u, 0u, 0u, 0u}, nxt_pa{0u, 0u, 0u, 0u};
        bool nxt_raw3 = false;
        // cur[] was read from the landing buffer at the end of the previous sub-tile and those reads have long returned
        // (a flush lies in between); the buffer is free for the next sub-tile's pixels
        PHASE(7);
        if (more) fetch(mid_tag, j + W, nxt_pb, nxt_pa, nxt_raw3);
        PHASE(0);

        const uint32_t base = j * ENC_SUB;
        const uint32_t pb = cur_pb;
        const uint32_t pa = cur_pa;
        const bool has_after = MID ? true : base + ENC_SUB < N;
        const bool last_sub = MID ? false : j == nsub - 1;
        const bool eof = !has_after && strip_last;   // the sub-tile's last pixel is the image's last

        // ---- pre-pass: equality masks (from the SWAR classification on the fast path), local run aggregate ----
        unsigned long long E[ENC_S];
        constexpr bool swar = SW;
        SwarGroup grp[ENC_S / 4];
        // a sub-tile that is one solid run (flat areas of icons / screenshots) needs no per-pixel work at all
        // (SQOA-native only.  Run pixels never touch the colour index -- seqoia.h:544-550 come before :563 -- so the same shortcut is
        // right for QOI-compat sub-tiles [0xfd at every multiple of 62: floor((c_in + 512) / 62) - floor(c_in / 62) of them, then the
        // flush]; built and measured in round 3: icons 4096 x 512 x 512 RGB unchanged at 4.81 ms (a sub-tile is a whole row of 16
        // tiles there, never one run), the headline batch 3.59 -> 3.69 ms for the test alone.  The same for skipping the index
        // probe of rows that are all run: 3.59 -> 4.01 ms, icons 4.81 -> 4.98 ms.)
        bool flat = false;
        if (swar && !QOI) {
            uint32_t dif = 0;
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) dif |= cur[s] ^ pb;
            asm volatile("" : "+v"(dif));  // one compare of the OR, not eight compares OR-ed on the scalar pipe
            flat = __ballot(dif != 0u) == 0ull;
        }
        if (flat) {
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) E[s] = ~0ull;
        } else if (swar) swar_classify<QOI>(cur, pb, grp, E);
        else {
            uint32_t carry = pb;
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) {
                const uint32_t px = cur[s];
                const uint32_t prev = shift_up1(px, carry);
                carry = bcast_lane(px, 63);
                const uint32_t idx = base + s * 64 + lane;
                E[s] = __ballot(px == prev && idx < N);
            }
        }
        PHASE(1);
        uint32_t trail = 0;
        bool all_eq = true;
#pragma unroll
        for (int s = ENC_S - 1; s >= 0; --s) {
            if (all_eq) {
                if (E[s] == ~0ull) trail += 64;
                else { trail += (uint32_t)__builtin_clzll(~E[s]); all_eq = false; }
            }
        }
        unsigned long long* my_run = &s_run[(j / W) & 1][w];
        if (!all_eq && !last_sub) chain_post(my_run, j + 1, trail, lane);
        uint32_t c_in = 0;
        if (j > 0 && (E[0] & 1ull)) c_in = chain_wait<QOI>(&s_run[((j - 1) / W) & 1][(j - 1) % W], j, st);
        if (all_eq && !last_sub) chain_post(my_run, j + 1, c_in + ENC_SUB, lane);

        // ---- QOI: colour index entering this sub-tile, and the one leaving it ---------
        uint32_t* tabw = s_tabw[w % QW];
        unsigned long long* smask = s_smask[w % QW];
        [[maybe_unused]] uint32_t* pstamp = s_pstamp[w % (QOI && SQOA_ENC_Q_RTN ? W : 1)];
        if constexpr (QOI) {
            uint32_t* stamp = s_stamp[w % QW];
            uint32_t* pub = s_tabpub[(j / W) & 1][w % QW];
            // last non-run pixel per slot inside the sub-tile: the table only ever holds the most
            // recent writer of each slot (hit or miss, seqoia.h:566-571), so it composes by overwrite
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) {
                const bool ne = !((E[s] >> lane) & 1ull) && base + s * 64 + lane < N;
                if (ne) atomicMax(&stamp[qoi_hash(cur[s])], (uint32_t)(s * 64 + lane + 1));
            }
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) {
                const bool ne = !((E[s] >> lane) & 1ull) && base + s * 64 + lane < N;
                const uint32_t h = qoi_hash(cur[s]);
                if (ne && stamp[h] == (uint32_t)(s * 64 + lane + 1)) pub[h] = cur[s];
            }
            uint32_t tin = qtab_in ? qtab_in[(size_t)jid * 64 + lane] : 0u;  // seqoia.h:518, or what the strips in front left
            if (j > 0) {
                const uint32_t pp = ((j - 1) / W) & 1, pw = (j - 1) % W;
                for (uint32_t spin = 0;; ++spin) {
                    const uint32_t t = __hip_atomic_load(&s_tabtag[pp][pw], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (bcast_first(t) == j) break;
                    if (spin >= SPIN_MAX) { st.bad = 1; break; }
                    if (st.eager) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(SQOA_ENC_QTAB_SLEEP);
                }
                tin = s_tabpub[pp][pw % QW][lane];
            }
            if (stamp[lane] == 0) pub[lane] = tin;
            tabw[lane] = tin;
            stamp[lane] = 0;
            if (!last_sub && lane == 0)
                __hip_atomic_store(&s_tabtag[(j / W) & 1][w], j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }

        PHASE(2);
        // ---- main pass: op bytes, offsets, compaction into LDS -------------------------
        uint32_t sbase = 0;    // bytes staged so far
        if (flat) {
            // k runs over c_in+1 .. c_in+512: exactly one of them is a multiple of 512 (0xfd, seqoia.h:546-549); the
            // last pixel additionally owns the flush (seqoia.h:554-561) or the trailing 0xfd (seqoia.h:640-642)
            const uint32_t r0 = c_in & 511u;              // the 0xfd falls on pixel 511 - r0; afterwards the remainder is r0
            const bool end_px = eof;
            const bool next_eq = has_after && pa == pb;
            unsigned long long v = 0xfdull;
            uint32_t nbytes = 1, b9 = 0;
            if (r0 != 0u && (end_px || !next_eq)) {
                if (end_px) { v |= 0xfdull << 8; nbytes = 2; }
                else {
                    const uint32_t q = ((r0 - 1u) * 1075u) >> 16;
                    const uint32_t last = 0xc0u | (r0 - 1u - 61u * q);
                    const unsigned long long fc = 0xfcfcfcfcfcfcfcfcull;
                    const unsigned long long f = q >= 7 ? fc : ((fc & ((1ull << (8u * q)) - 1ull)) | (unsigned long long)last << (8u * q));
                    v |= f << 8;                               // bytes 1..7 of the 64-bit word
                    if (q >= 7) b9 = q == 7 ? last : (0xfcu | last << 8);  // bytes 8 (and 9) do not fit the word
                    nbytes = q + 2u;
                }
            }
            if (lane == 0) {  // position-independent staging: the bytes start at offset 0 of the (zeroed) buffer
                ((lptr32)stage)[0] = (uint32_t)v;
                ((lptr32)stage)[1] = (uint32_t)(v >> 32);
                ((lptr32)stage)[2] = b9;
            }
            sbase = nbytes;
#if SQOA_ENC_LOOKBACK
            chain_post(&s_agg[(j / W) & 1][w], j + 1, sbase, lane);
#endif
        } else if (swar) {
            sbase = swar_emit<QOI>(cur, grp, E, c_in, pa, has_after, eof, stage, lane, le_mask, tabw, smask, pstamp, qrow64);
        } else {
        uint32_t crun = c_in;  // equal pixels immediately before lane 0 of the step
        uint32_t carry = pb;
#pragma unroll
        for (int s = 0; s < ENC_S; ++s) {
            const uint32_t px = cur[s];
            const uint32_t prev = shift_up1(px, carry);
            carry = bcast_lane(px, 63);
            const uint32_t idx = base + s * 64 + lane;
            const unsigned long long e = E[s];
            const bool iseq = (e >> lane) & 1ull;

            uint32_t lo, hi, n;
            literal_op<QOI>(px, prev, mono, lo, hi, n);
            if constexpr (QOI) {  // index probe (seqoia.h:563-572), all 64 pixels of the step at once
                const bool ne = !iseq && idx < N;
                const uint32_t h = qoi_hash(px);
                uint32_t cand = tabw[h];  // slot as the previous steps left it
#if SQOA_ENC_Q_RTN
                const uint32_t prev = ne ? __hip_atomic_fetch_max(&pstamp[h], qrow64 + (uint32_t)lane + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
                const bool lower = prev > qrow64;
                const int from = lower ? (int)(prev - qrow64 - 1u) : lane;
                const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)px);
                if (lower) cand = near;
                if (ne) tabw[h] = px;
                qrow64 += 64u;
#else
                if (ne) __hip_atomic_fetch_or(&smask[h], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long m = __hip_atomic_load(&smask[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long lower = m & (le_mask >> 1);  // earlier lanes of this step on my slot
                const int from = lower ? 63 - __builtin_clzll(lower) : lane;
                const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)px);
                if (lower) cand = near;
                if (ne && (m & ~le_mask) == 0ull) { tabw[h] = px; smask[h] = 0; }  // last writer of the slot in this step
#endif
                if (ne && cand == px) { lo = h; hi = 0; n = 1; }
            }
            if (iseq || idx >= N) { lo = 0; hi = 0; n = 0; }
            uint32_t b9 = 0;  // 9th byte of a maximal run flush

            if (e != 0ull) {  // some lane continues a run (seqoia.h:544-561, :640-642)
                const unsigned long long below = ~e & le_mask;
                const uint32_t k = below ? (uint32_t)lane - (63u - (uint32_t)__builtin_clzll(below)) : crun + lane + 1u;
                const uint32_t r = QOI ? k % M : k & (M - 1u);
                // is the pixel after lane 63 equal to lane 63's?  (next step's bit 0, or the pixel after the sub-tile)
                const unsigned long long nb = s + 1 < ENC_S ? (E[(s + 1) % ENC_S] & 1ull) : (unsigned long long)(has_after && pa == carry);
                const unsigned long long enext = e >> 1 | nb << 63;
                const bool fd = iseq && (r == 0u || (strip_last && idx == N - 1u));
                const bool fl = iseq && !fd && !((enext >> lane) & 1ull);
                if (fd) { lo = 0xfdu; n = 1; }
                if (fl) {
                    if (QOI) { lo = 0xc0u | (r - 1u); n = 1; }
                    else {
                        const uint32_t q = ((r - 1u) * 1075u) >> 16;  // (r-1)/61 for r <= 511
                        const uint32_t last = 0xc0u | (r - 1u - 61u * q);
                        const unsigned long long fc = 0xfcfcfcfcfcfcfcfcull;
                        unsigned long long v = q >= 8 ? fc : ((fc & ((1ull << (8 * q)) - 1ull)) | (unsigned long long)last << (8 * q));
                        lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
                        n = q + 1u;
                        if (q >= 8) b9 = last;
                    }
                }
            }

            const uint32_t incl = wave_incl_scan(n);
            const uint32_t o = sbase + incl - n;
            sbase += bcast_lane(incl, 63);

            const uint32_t sh = (o & 3u) * 8u;
            const unsigned long long v = ((unsigned long long)hi << 32 | lo) << sh;
            lptr32 p = (lptr32)(stage + (o & ~3u));
            if (n) {
                __hip_atomic_fetch_or(p, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(p + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            const bool spill = (o & 3u) + (n > 8u ? 8u : n) > 8u;
            if (__ballot(spill || n == 9u) != 0ull) {
                if (spill) __hip_atomic_fetch_or(p + 2, hi >> (32u - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (n == 9u) {
                    const uint32_t o9 = o + 8u;
                    __hip_atomic_fetch_or((lptr32)(stage + (o9 & ~3u)), b9 << ((o9 & 3u) * 8u),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            crun = e == ~0ull ? crun + 64u : (uint32_t)__builtin_clzll(~e);
        }
        }

        PHASE(3);
        // this sub-tile's pixels are dead: pick up the next one's before the stores of the flush join the memory queue
        if (more) {
            if (next8) take8(cur, KIND == 2 ? false : nxt_raw3); else take(cur, KIND == 2 ? false : nxt_raw3);
            cur_pb = nxt_pb.value();
            cur_pa = nxt_pa.value();
        }

        PHASE(4);
        // ---- output position from the chain, then drain the staging buffer -------------
#if SQOA_ENC_LOOKBACK
        if (!last_sub) chain_post(&s_agg[(j / W) & 1][w], j + 1, sbase, lane);   // my byte count, before anything is waited for
        const uint32_t pos = j == 0 ? hdr_len : pos_lookback<W>(s_pos, s_agg, j, st);
        PHASE(5);
        if (!last_sub) chain_post(&s_pos[(j / W) & 1][w], j + 1, pos + sbase, lane);
#else
        const uint32_t pos = j == 0 ? hdr_len : chain_wait<QOI>(&s_pos[(j - 1) % W], j, st);
        PHASE(5);
        if (!last_sub) chain_post(&s_pos[w], j + 1, pos + sbase, lane);
#endif
        flush_stage(stage, sbase, dst + pos, lane);
        PHASE(6);
        if (last_sub) {  // seqoia.h:644-649
            const uint32_t end = pos + sbase;
            if (strip_last && lane < 8) dst[end + lane] = lane == 7 ? 1 : 0;
            if (lane == 0) results[jid].len = (int)(end + (strip_last ? 8u : 0u));
#ifdef SQOA_TRACE
            if (lane == 0) trace[1] = wall_clock64();
#endif
        }
    };


    // ---- L8: every lane owns EIGHT CONSECUTIVE pixels of the sub-tile (SQOA colour images, 4-byte pixels, hot loop) ----------------
    // The row layout above pays per 64-pixel row: a ballot, run bookkeeping from 64-bit lane masks, a DPP prefix sum, the
    // lane-0 fix of the predecessor planes.  With eight consecutive pixels per lane the predecessor of 7 pixels out of 8 is
    // in the lane's own registers (a byte shift of the channel planes), runs shorter than the lane's stretch are a table
    // look-up on the lane's 8 repeat flags, and the byte offsets are one in-register prefix per four pixels plus ONE wave
    // scan per sub-tile.  Runs that cross lanes (flat areas) take a per-lane fix-up behind a wave-uniform branch.
    auto body8 = [&](auto px3_tag, uint32_t j, bool next8) {
        constexpr bool SRC3 = decltype(px3_tag)::value; // 3-byte pixels in memory
        constexpr bool PX3 = SRC3 && !QOI;              // ... and in cur[0..5]: the lane's 24 bytes as they lie (QOI-compat: take8 has made dwords of them)
        {
        typedef __attribute__((address_space(3))) uint32_t* lds_u32;
        __builtin_assume(j > 0);
        NbrPx nxt_pb{0u, 0u, 0u, 0u}, nxt_pa{0u, 0u, 0u, 0u};
        bool nxt_raw3 = false;
        PHASE(7);
        fetch(std::integral_constant<int, SRC3 ? 1 : 2>{}, j + W, nxt_pb, nxt_pa, nxt_raw3);
        PHASE(0);
        const uint32_t pb = cur_pb, pa = cur_pa;
        // channel planes of my two groups of four pixels (byte i = pixel 4g + i)
        uint32_t R0, G0, B0, A0, R1, G1, B1, A1;
        uint32_t dif;
        if constexpr (PX3) {
            // d0 = r0 g0 b0 r1, d1 = g1 b1 r2 g2, d2 = b2 r3 g3 b3: every plane takes bytes of all three dwords
            auto gather3 = [&](uint32_t d0, uint32_t d1, uint32_t d2, uint32_t& R, uint32_t& G, uint32_t& B) {
                R = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c060300u), 0x05020100u);
                G = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c070401u), 0x06020100u);
                B = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c0c0502u), 0x07040100u);
            };
            gather3(cur[0], cur[1], cur[2], R0, G0, B0);
            gather3(cur[3], cur[4], cur[5], R1, G1, B1);
            A0 = A1 = 0xffffffffu;
            const uint32_t m = 0x01010101u;
            dif = (R0 ^ (pb & 255u) * m) | (G0 ^ ((pb >> 8) & 255u) * m) | (B0 ^ ((pb >> 16) & 255u) * m) |
                  (R1 ^ (pb & 255u) * m) | (G1 ^ ((pb >> 8) & 255u) * m) | (B1 ^ ((pb >> 16) & 255u) * m);
        } else {
#ifndef SQOA_ENC_DIF_LAZY   // "is the whole sub-tile one run" from the lanes' FIRST pixels, and only where those all repeat from the other seven
#define SQOA_ENC_DIF_LAZY 1
#endif
#if SQOA_ENC_DIF_LAZY
            dif = cur[0] ^ pb;
            asm volatile("" : "+v"(dif));
            if (__ballot(dif != 0u) == 0ull) {
#pragma unroll
                for (int s = 1; s < ENC_S; ++s) dif |= cur[s] ^ pb;
            }
#else
            dif = 0;
#pragma unroll
            for (int s = 0; s < ENC_S; ++s) dif |= cur[s] ^ pb;
#endif
        }
        asm volatile("" : "+v"(dif));
        const bool flat = __ballot(dif != 0u) == 0ull;
        unsigned long long* my_run = &s_run[(j / W) & 1][w];
        unsigned long long* prev_run = &s_run[((j - 1) / W) & 1][(j - 1) % W];
        uint32_t sbase = 0;
        if (flat && QOI) {
            // QOI-compat runs are cut every 62 pixels (seqoia.h:411, :546-549): k runs over c_in+1 .. c_in+512, a 0xfd at every
            // multiple of 62 -- floor((c_in + 512) / 62) - floor(c_in / 62) of them, 8 or 9 --, and behind the last pixel the flush
            // of the remainder (seqoia.h:554-561) if the run ends here.  Run pixels never touch the colour index (seqoia.h:544-550
            // come before :563): the table entering the sub-tile is the one leaving it.
            const uint32_t c_in = chain_wait<QOI>(prev_run, j, st);
            chain_post(my_run, j + 1, c_in + ENC_SUB, lane);
            const uint32_t kend = c_in + ENC_SUB, qend = kend / 62u;
            const uint32_t nfd = qend - c_in / 62u, r = kend - 62u * qend;
            uint32_t nbytes = nfd, last = 0;
            if (r != 0u && pa != pb) { last = 0xc0u | (r - 1u); nbytes = nfd + 1u; }
            if ((uint32_t)lane < nbytes) stage[lane] = (uint32_t)lane < nfd ? (uint8_t)0xfd : (uint8_t)last;   // (the buffer is zeroed, position-independent)
            sbase = nbytes;
#if SQOA_ENC_LOOKBACK
            chain_post(&s_agg[(j / W) & 1][w], j + 1, sbase, lane);
#endif
            qoi_table_hop(j, false, [](unsigned long long*) {});
        } else if (flat) {
            const uint32_t c_in = chain_wait<QOI>(prev_run, j, st);
            chain_post(my_run, j + 1, c_in + ENC_SUB, lane);
            // k runs over c_in+1 .. c_in+512: exactly one of them is a multiple of 512 (0xfd, seqoia.h:546-549); the
            // last pixel additionally owns the flush (seqoia.h:554-561)
            const uint32_t r0 = c_in & 511u;
            unsigned long long v = 0xfdull;
            uint32_t nbytes = 1, b9 = 0;
            if (r0 != 0u && pa != pb) {
                const uint32_t q = ((r0 - 1u) * 1075u) >> 16;
                const uint32_t last = 0xc0u | (r0 - 1u - 61u * q);
                const unsigned long long fc = 0xfcfcfcfcfcfcfcfcull;
                const unsigned long long f = q >= 7 ? fc : ((fc & ((1ull << (8u * q)) - 1ull)) | (unsigned long long)last << (8u * q));
                v |= f << 8;
                if (q >= 7) b9 = q == 7 ? last : (0xfcu | last << 8);
                nbytes = q + 2u;
            }
            if (lane == 0) {
                ((lptr32)stage)[0] = (uint32_t)v;
                ((lptr32)stage)[1] = (uint32_t)(v >> 32);
                ((lptr32)stage)[2] = b9;
            }
            sbase = nbytes;
#if SQOA_ENC_LOOKBACK
            chain_post(&s_agg[(j / W) & 1][w], j + 1, sbase, lane);
#endif
        } else {
            const uint32_t ONE = 0x01010101u;
            uint32_t Hv, L7;  // the two SWAR constants in VGPRs
            asm volatile("v_mov_b32 %0, 0x80808080" : "=v"(Hv));
            asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(L7));
            if constexpr (!PX3) {
                {
                    const uint32_t a01 = __builtin_amdgcn_perm(cur[1], cur[0], 0x05010400u), b01 = __builtin_amdgcn_perm(cur[1], cur[0], 0x07030602u);
                    const uint32_t a23 = __builtin_amdgcn_perm(cur[3], cur[2], 0x05010400u), b23 = __builtin_amdgcn_perm(cur[3], cur[2], 0x07030602u);
                    R0 = __builtin_amdgcn_perm(a23, a01, 0x05040100u); G0 = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
                    B0 = __builtin_amdgcn_perm(b23, b01, 0x05040100u); A0 = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
                }
                {
                    const uint32_t a01 = __builtin_amdgcn_perm(cur[5], cur[4], 0x05010400u), b01 = __builtin_amdgcn_perm(cur[5], cur[4], 0x07030602u);
                    const uint32_t a23 = __builtin_amdgcn_perm(cur[7], cur[6], 0x05010400u), b23 = __builtin_amdgcn_perm(cur[7], cur[6], 0x07030602u);
                    R1 = __builtin_amdgcn_perm(a23, a01, 0x05040100u); G1 = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
                    B1 = __builtin_amdgcn_perm(b23, b01, 0x05040100u); A1 = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
                }
            }
            // the sub-tile's last pixel (wave-uniform), for the repeat flag of the pixel behind the sub-tile
            const uint32_t last_px = PX3 ? bcast_lane((R1 >> 24) | ((G1 >> 24) << 8) | ((B1 >> 24) << 16) | 0xff000000u, 63) : bcast_lane(cur[ENC_S - 1], 63);
            // predecessor planes: a byte shift inside the lane; the first pixel takes the lane below's last one (lane 0: pb)
            const uint32_t qR = shift_up1(R1, pb << 24), qG = shift_up1(G1, (pb << 16) & 0xff000000u),
                           qB = shift_up1(B1, (pb << 8) & 0xff000000u), qA = shift_up1(A1, pb & 0xff000000u);
            auto group = [&](uint32_t R, uint32_t G, uint32_t B, uint32_t A, uint32_t Rp, uint32_t Gp, uint32_t Bp, uint32_t Ap) {
                if constexpr (QOI) return qoi_group(R, G, B, A, Rp, Gp, Bp, Ap, Hv, L7);
                else return sqoa_group(R, G, B, A, Rp, Gp, Bp, Ap, Hv, L7);
            };
            SwarGroup g0 = group(R0, G0, B0, A0, __builtin_amdgcn_alignbyte(R0, qR, 3), __builtin_amdgcn_alignbyte(G0, qG, 3),
                                 __builtin_amdgcn_alignbyte(B0, qB, 3), __builtin_amdgcn_alignbyte(A0, qA, 3));
            SwarGroup g1 = group(R1, G1, B1, A1, __builtin_amdgcn_alignbyte(R1, R0, 3), __builtin_amdgcn_alignbyte(G1, G0, 3),
                                 __builtin_amdgcn_alignbyte(B1, B0, 3), __builtin_amdgcn_alignbyte(A1, A0, 3));
            // ---- runs (seqoia.h:544-561): e = 1 where a pixel repeats its predecessor, l = 1 where a run ends
            const uint32_t e0 = g0.b0 ^ ONE, e1 = g1.b0 ^ ONE;
            const uint32_t e0n = shift_down1(e0, pa == last_px ? 1u : 0u);  // the lane above's first four flags; lane 63: the pixel after the sub-tile
            const uint32_t l0 = e0 & ~__builtin_amdgcn_alignbyte(e1, e0, 1), l1 = e1 & ~__builtin_amdgcn_alignbyte(e0n, e1, 1);
            const uint32_t E8 = __builtin_amdgcn_udot4(e1, 0x80402010u, __builtin_amdgcn_udot4(e0, 0x08040201u, 0u, false), false);
            const uint32_t rt0 = s_runtab[E8][0], rt1 = s_runtab[E8][1];
            uint32_t W00 = g0.W0 | (rt0 & ((l0 + L7) ^ L7)), W01 = g1.W0 | (rt1 & ((l1 + L7) ^ L7));
            uint32_t Nn0 = g0.Nn + l0, Nn1 = g1.Nn + l1;
            PHASE(1);
            // the sub-tile's trailing run, published before anything is waited for
            const unsigned long long allq = __ballot(E8 == 0xffu);   // lanes whose eight pixels all repeat (never all 64: not flat)
            const uint32_t tzl = (uint32_t)__builtin_clzll(~allq);
            const uint32_t Ea = (uint32_t)__builtin_amdgcn_readlane((int)E8, (int)(63u - tzl));
            chain_post(my_run, j + 1, 8u * tzl + (uint32_t)__builtin_clz((~Ea << 24) | 0x00800000u), lane);
            if constexpr (QOI) {
                // ---- the colour index (seqoia.h:563-582) stays in PIXEL order across lanes: the sub-tile's pixels are read once more,
                // row by row (lane = pixel of a row of 64) from the landing buffer they were taken from, probed row after row exactly as
                // in the row layout, and each pixel's verdict (0x80 | slot: hit) goes back through 512 bytes of LDS, where the lane
                // that owns the pixel finds the verdicts of its eight as two dwords -- byte planes like all the others
                const unsigned long long le_mask = (2ull << (uint32_t)lane) - 1ull;
                uint32_t rowpx[ENC_S], nebits = 0;
                {
                    uint32_t carry = pb;
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        uint32_t v;
                        if constexpr (PX3) { v = *(lptr32u)(held + 3 * (r * 64 + lane)); v |= 0xff000000u; }
                        else v = ((lptr32)held)[r * 64 + lane];
                        rowpx[r] = v;
                        nebits |= (v != shift_up1(v, carry) ? 1u : 0u) << r;
                        carry = bcast_lane(v, 63);
                    }
                }
                uint8_t* hit = s_hit[w];
#if SQOA_ENC_Q_PIPE && SQOA_ENC_Q_RTN == 3
                // gfx950's LDS applies the lanes of ONE instruction that meet on an address in ascending lane order, and a wave's instructions
                // in program order: an atomic with return hands every lane what was there just before ITS turn (tools/ubench/
                // lds_atomic_order.hip, run on the box by tests/test_gpu_tools.py).  So ONE `ds_max_rtn_u64` per row of (stamp << 32 | pixel)
                // -- the stamp grows with row and lane -- returns, to every pixel that starts an op, the most recent EARLIER writer of its
                // index slot inside this sub-tile, lower lane of its row or earlier row alike, stamp and pixel: the candidate the probe is
                // after, unless the stamp is older than the sub-tile -- then it is the entry of the table the sub-tile before published,
                // read where that wave left it (it cannot publish into the same buffer again before this wave has handed its byte position
                // on: the position chain).  No working copy of the table, no per-slot lane masks, no ds_bpermute, nothing to clear; "the
                // sub-tile's last writer of every slot", which the table hop publishes before the incoming table is known, is what the
                // entries hold when the eight atomics are through -- they do not depend on the table and are issued in front of the wait.
                uint32_t ph[ENC_S];
                unsigned long long pprev[ENC_S];
                qoi_table_hop(j, true, [&](unsigned long long*) {
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        const uint32_t h = qoi_hash(rowpx[r]);
                        ph[r] = h;
                        pprev[r] = 0;
                        if ((nebits >> r) & 1u)
                            pprev[r] = __hip_atomic_fetch_max(&s_pst64[w][h], (unsigned long long)(qrow64 + (uint32_t)(r * 64 + lane + 1)) << 32 | rowpx[r],
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                });
                {
                    const uint32_t* tinp = s_tabpub[((j - 1) / W) & 1][(j - 1) % W];
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        const bool ne = (nebits >> r) & 1u;
                        const uint32_t tinv = tinp[ph[r]];
                        const uint32_t cand = (uint32_t)(pprev[r] >> 32) > qrow64 ? (uint32_t)pprev[r] : tinv;
                        hit[r * 64 + lane] = (uint8_t)((ne && cand == rowpx[r]) ? 0x80u | ph[r] : 0u);
                    }
                    qrow64 += (uint32_t)(ENC_S * 64);
                }
#elif SQOA_ENC_Q_PIPE && SQOA_ENC_Q_RTN == 2
                // gfx950's LDS applies the lanes of ONE instruction that meet on an address in ascending lane order: of an atomic with return
                // every lane gets what the lanes below it left, of a plain write the highest lane's value stays (tools/ubench/lds_atomic_order.hip:
                // 8 x 10^8 of each over random, strided, clustered and single-address patterns with eight waves on the LDS, none out of order;
                // tests/test_gpu_tools.py runs it on the box).  Two things follow (round 5):
                //  * `ds_max_rtn_u32` of a stamp that grows with row and lane hands every lane the stamp of the nearest LOWER lane of its row
                //    on its index slot (or an older row's: none) -- what the per-slot 64-bit lane masks took a ds_or_b64, a ds_read_b64, a
                //    clear by the last writer and a dozen instructions of 64-bit mask arithmetic to find;
                //  * "the sub-tile's last writer of every slot" -- what the table hop publishes before the incoming table is known -- is the
                //    last plain ds_write_b32 of the pixel, row after row, and "did the sub-tile write the slot" is "its stamp is of this
                //    sub-tile": eight b32 writes where there were eight ds_max_u64, nothing to clear.
                // The atomics do not depend on the table: they are issued here, in front of the wait for the sub-tile before.
                uint32_t ph[ENC_S], pprev[ENC_S];
                qoi_table_hop(j, true, [&](unsigned long long*) {
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        const uint32_t h = qoi_hash(rowpx[r]);
                        ph[r] = h;
                        pprev[r] = 0;
                        if ((nebits >> r) & 1u) {
                            pprev[r] = __hip_atomic_fetch_max(&s_pstamp[w][h], qrow64 + (uint32_t)(r * 64 + lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            s_lastpx[w][h] = rowpx[r];
                        }
                    }
                });
                {
                    // the eight rows' probes: the slot's entry as the rows before left it (row r + 1's read travels with row r's ds_bpermute:
                    // LDS keeps a wave's order, nothing is waited for between a row's table write and the next row's read)
                    uint32_t pcand[ENC_S];
                    pcand[0] = s_tabw[w][ph[0]];
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        const bool ne = (nebits >> r) & 1u;
                        const uint32_t rb = qrow64 + (uint32_t)(r * 64);
                        const bool lower = pprev[r] > rb;                                  // a lower lane of THIS row took my slot
                        const int from = lower ? (int)(pprev[r] - rb - 1u) : lane;
                        const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)rowpx[r]);
                        if (ne) s_tabw[w][ph[r]] = rowpx[r];                                // (of the lanes on one slot the highest stays: the row's last writer)
                        if (r + 1 < ENC_S) pcand[r + 1] = s_tabw[w][ph[r + 1]];
                        const uint32_t cand = lower ? near : pcand[r];
                        hit[r * 64 + lane] = (uint8_t)((ne && cand == rowpx[r]) ? 0x80u | ph[r] : 0u);
                    }
                    qrow64 += (uint32_t)(ENC_S * 64);
                }
#else
                qoi_table_hop(j, true, [&](unsigned long long* last) {
#ifdef SQOA_DBG_Q_NOSTAMP
                    return;
#endif
                    // last non-run pixel per slot inside the sub-tile: the table only ever holds the most recent writer of each slot
                    // (hit or miss, seqoia.h:566-571), so it composes by overwrite.  One ds_max_u64 of position | pixel per row.
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r)
                        if ((nebits >> r) & 1u) atomicMax(&last[qoi_hash(rowpx[r])], (unsigned long long)(r * 64 + lane + 1) << 32 | rowpx[r]);
                });
#if SQOA_ENC_Q_PIPE
                // The eight rows' probes, software-pipelined by hand.  Row r + 1 can only look at the table after row r has written it, but
                // LDS executes a wave's instructions in order: nothing has to be WAITED for between row r's writes and row r + 1's reads.
                // What a row does wait for is its own lane-mask word (who else of the row is on my slot); the write of the table entry
                // depends on that word alone, not on the candidate pixel the ds_bpermute fetches -- so row r's ds_bpermute travels with row
                // r + 1's reads, and a row costs one round trip to LDS instead of two.
                {
                    uint32_t ph[ENC_S], pcand[ENC_S];
                    unsigned long long pm[ENC_S];
                    auto probe_a = [&](int r) {   // my slot's entry as the rows before left it; my bit into the slot's lane mask; the mask
                        const uint32_t h = qoi_hash(rowpx[r]);
                        ph[r] = h;
                        pcand[r] = s_tabw[w][h];
                        if ((nebits >> r) & 1u) __hip_atomic_fetch_or(&s_smask[w][h], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        pm[r] = __hip_atomic_load(&s_smask[w][h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    };
                    probe_a(0);
#pragma unroll
                    for (int r = 0; r < ENC_S; ++r) {
                        const bool ne = (nebits >> r) & 1u;
                        const unsigned long long m = pm[r], lower = m & (le_mask >> 1);
                        const int from = lower ? 63 - __builtin_clzll(lower) : lane;
                        const uint32_t near = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)rowpx[r]);
                        if (ne && (m & ~le_mask) == 0ull) { s_tabw[w][ph[r]] = rowpx[r]; s_smask[w][ph[r]] = 0; }  // last writer of the slot in this row
                        if (r + 1 < ENC_S) probe_a(r + 1);
                        const uint32_t cand = lower ? near : pcand[r];
                        hit[r * 64 + lane] = (uint8_t)((ne && cand == rowpx[r]) ? 0x80u | ph[r] : 0u);
                    }
                }
#else
#pragma unroll
                for (int r = 0; r < ENC_S; ++r)
#ifdef SQOA_DBG_Q_NOPROBE   // timing experiments only (wrong output)
                    hit[r * 64 + lane] = 0;
#elif defined(SQOA_ENC_Q_ROWSKIP)
                {
                    uint32_t hb = 0;
                    if (__ballot((nebits >> r) & 1u) != 0ull) hb = qoi_probe_row(rowpx[r], (nebits >> r) & 1u, s_tabw[w], s_smask[w], lane, le_mask);
                    hit[r * 64 + lane] = (uint8_t)hb;
                }
#else
                    hit[r * 64 + lane] = (uint8_t)qoi_probe_row(rowpx[r], (nebits >> r) & 1u, s_tabw[w], s_smask[w], lane, le_mask);
#endif
#endif
#endif
                const uint2 hv = *reinterpret_cast<const uint2*>(hit + 8 * lane);
                // QOI_OP_INDEX where the pixel hits: one byte, the slot (seqoia.h:566-568)
                auto index_op = [&](uint32_t h4, SwarGroup& G, uint32_t& W0, uint32_t& Nn) {
                    const uint32_t z = h4 & Hv, m = (z - (z >> 7)) | z;
                    W0 = (W0 & ~m) | (h4 & 0x3f3f3f3fu & m);
                    G.W1 &= ~m; G.W2 &= ~m; G.W3 &= ~m; G.AH &= ~m;
                    Nn = (Nn & ~m) | (ONE & m);
                };
                index_op(hv.x, g0, W00, Nn0);
                index_op(hv.y, g1, W01, Nn1);
            }
            uint32_t big = 0, big_at = 0;  // run remainder >= 62 (seqoia.h:555-559): q << 8 | last byte, and the pixel that owns it
#ifdef SQOA_DBG_NOCARRY
            if (false) {
#else
            if (__ballot((e0 & 1u) != 0u) != 0ull) {
#endif
                // some lane starts inside a run: its leading repeats continue a run of `carry` pixels in front of the lane
                const uint32_t c_in = (bcast_first(e0) & 1u) ? chain_wait<QOI>(prev_run, j, st) : 0u;
                const uint32_t nE = ~E8;
                const uint32_t f = (uint32_t)__builtin_ctz(nE);                                   // leading repeats, 0..8 (bit 8 of ~E8 is set)
                // the sub-tile pixel index of the last pixel below my stretch that does NOT repeat: a running maximum over the lanes
                // (one v_max_i32 per DPP step), -1 if every pixel below repeats (the run then comes from before the sub-tile)
                const int hb = 7 - (int)__builtin_clz((nE & 0xffu) << 24);                            // my own last such pixel; clz(0) = 32: none
#ifndef SQOA_ENC_MAXSCAN_U   // the running maximum over index + 1 (0: none) as six v_max_u32_dpp: zero is that maximum's identity, which lets the DPP move
#define SQOA_ENC_MAXSCAN_U 1 // fold into the v_max (signed with -1 for "none" it is v_mov + v_mov_dpp + v_max_i32 per step)
#endif
#if SQOA_ENC_MAXSCAN_U
                uint32_t lastne1 = E8 == 0xffu ? 0u : (uint32_t)(8 * lane + hb + 1);
                asm("s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "s_nop 1\n\t"
                    "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "s_nop 1"
                    : "+v"(lastne1));
                const uint32_t below1 = shift_up1(lastne1, 0u);                                        // exclusive: lanes below mine; 0: every pixel below repeats
                const uint32_t carry = below1 ? 8u * (uint32_t)lane - below1 : 8u * (uint32_t)lane + c_in;
#else
                int lastne = E8 == 0xffu ? -1 : 8 * lane + hb;
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_SHR + 1, 0xf, 0xf, false));
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_SHR + 2, 0xf, 0xf, false));
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_SHR + 4, 0xf, 0xf, false));
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_SHR + 8, 0xf, 0xf, false));
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_BCAST15, 0xa, 0xf, false));
                lastne = max(lastne, __builtin_amdgcn_update_dpp(-1, lastne, DPP_ROW_BCAST31, 0xc, 0xf, false));
                const int below_ne = __builtin_amdgcn_update_dpp(-1, lastne, DPP_WAVE_SHR1, 0xf, 0xf, false);   // exclusive: lanes below mine
                const uint32_t carry = below_ne >= 0 ? (uint32_t)(8 * lane - 1 - below_ne) : 8u * (uint32_t)lane + c_in;
#endif
                // my leading run ends in my stretch iff a pixel of mine follows it, or (all eight repeat) the pixel after my last does not
                const bool ends = (f - 1u < 7u) || (f == 8u && (l1 >> 24) != 0u);
                // the run's remainder where it ends: k mod 512 (mod 62: QOI-compat, seqoia.h:411-412; there it is at most 61: one byte)
                const uint32_t kk = carry + f;
                const uint32_t r = QOI ? kk - 62u * (uint32_t)(((unsigned long long)kk * 0x84210843ull) >> 37) : kk & 511u;
                const uint32_t q = QOI ? 0u : ((r - 1u) * 1075u) >> 16;                          // (r - 1) / 61 for 1 <= r <= 511
                const uint32_t lastb = 0xc0u | (r - 1u - 61u * q);
                // one byte of the (W0, Nn) planes of my eight pixels replaced (per 32-bit half: the same through 64-bit values
                // measured 12 % slower on the whole kernel)
                auto patch = [&](uint32_t px_i, uint32_t w0b, uint32_t nb, bool on) {
                    const uint32_t sh = (px_i & 3u) * 8u, m = (on ? 0xffu : 0u) << sh;
                    const uint32_t m0 = px_i < 4u ? m : 0u, m1 = px_i < 4u ? 0u : m;
                    const uint32_t wv = w0b << sh, nv = nb << sh;
                    W00 = (W00 & ~m0) | (wv & m0); W01 = (W01 & ~m1) | (wv & m1);
                    Nn0 = (Nn0 & ~m0) | (nv & m0); Nn1 = (Nn1 & ~m1) | (nv & m1);
                };
                // the run ends here: flush of the true remainder (the table only knows the repeats inside my stretch), or 0xfd
                // where the remainder is 0 (the pixel itself is the 512th, seqoia.h:546-549)
                const uint32_t pe = (f - 1u) & 7u;
                patch(pe, r == 0u ? 0xfdu : q == 0u ? lastb : 0u, r == 0u ? 1u : q + 1u, ends);
                if (ends && r != 0u && q != 0u) { big = q << 8 | lastb; big_at = pe; }
                // every 512th (62nd) pixel of the run
                const uint32_t cm = QOI ? carry - 62u * (uint32_t)(((unsigned long long)carry * 0x84210843ull) >> 37) : carry & 511u;
                const uint32_t jfd = (QOI ? 61u : 511u) - cm;
                const bool hasfd = jfd < f;
                if (__ballot(hasfd) != 0ull) patch(jfd & 7u, 0xfdu, 1u, hasfd);
            }
            PHASE(2);
            // ---- byte offsets: prefix inside the lane, one scan across the wave
            uint32_t P0, P1;   // (written out: the compiler turns x + (x << 8) + ... into a v_mul_lo_u32)
            asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(P0) : "v"(Nn0));
            asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(P1) : "v"(Nn1));
            asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(P0) : "v"(P0));
            asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(P1) : "v"(P1));
            const uint32_t T0 = P0 >> 24, T = T0 + (P1 >> 24);
            const uint32_t incl = wave_incl_scan(T);
            sbase = bcast_lane(incl, 63);
#if SQOA_ENC_LOOKBACK
            chain_post(&s_agg[(j / W) & 1][w], j + 1, sbase, lane);   // the sub-tile's byte count: published before its bytes are staged, before anything is waited for
#endif
            const uint32_t stage_addr = bcast_first((uint32_t)(uintptr_t)stage);
            // a lane without bytes ORs its zeros into a dword of its own (no 64 lanes on one LDS word in flat stretches)
            const uint32_t lb0 = T ? stage_addr + incl - T : stage_addr + 4u * lane, lb1 = lb0 + T0;
            const uint32_t X0 = P0 - Nn0, X1 = P1 - Nn1;   // exclusive prefixes, one byte per pixel
            // back from byte planes to one dword per pixel, shifted to its byte address and OR-ed into the staging buffer
            auto emit4 = [&](uint32_t w0, const SwarGroup& G, uint32_t X, uint32_t lb) {
                const uint32_t w01 = __builtin_amdgcn_perm(G.W1, w0, 0x05010400u), w23 = __builtin_amdgcn_perm(G.W3, G.W2, 0x05010400u);
                const uint32_t u01 = __builtin_amdgcn_perm(G.W1, w0, 0x07030602u), u23 = __builtin_amdgcn_perm(G.W3, G.W2, 0x07030602u);
                const uint32_t lo4[4] = {__builtin_amdgcn_perm(w23, w01, 0x05040100u), __builtin_amdgcn_perm(w23, w01, 0x07060302u),
                                         __builtin_amdgcn_perm(u23, u01, 0x05040100u), __builtin_amdgcn_perm(u23, u01, 0x07060302u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t o = lb + ((X >> (8 * i)) & 255u);
                    const uint32_t hi = (G.AH >> (8 * i)) & 255u;
                    const unsigned long long v = ((unsigned long long)hi << 32 | lo4[i]) << ((o & 3u) * 8u);
                    lds_u32 p = (lds_u32)(uintptr_t)(o & ~3u);
#ifdef SQOA_DBG_NOEMIT  // timing experiments only: the bytes are computed but not staged
                    asm volatile("" :: "v"(p), "v"((uint32_t)v), "v"((uint32_t)(v >> 32)));
#else
                    __hip_atomic_fetch_or(p, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_or(p + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
                }
            };
            emit4(W00, g0, X0, lb0);
            emit4(W01, g1, X1, lb1);
            if (__ballot(big != 0u) != 0ull) {  // q x 0xfc, then the last byte (rare)
                if (big) {
                    const uint32_t q = big >> 8, last = big & 255u;
                    const uint32_t ob0 = (big_at < 4u ? lb0 + ((X0 >> (8u * big_at)) & 255u) : lb1 + ((X1 >> (8u * (big_at & 3u))) & 255u));
#pragma nounroll
                    for (uint32_t tt = 0; tt <= q; ++tt) {
                        const uint32_t ob = ob0 + tt;
                        __hip_atomic_fetch_or((lds_u32)(uintptr_t)(ob & ~3u), (tt == q ? last : 0xfcu) << ((ob & 3u) * 8u),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        }
        PHASE(3);
        // this sub-tile's pixels are dead: pick up the next one's
        if (next8) take8(cur, SRC3); else take(cur, SRC3);
        cur_pb = nxt_pb.value();
        cur_pa = nxt_pa.value();
        PHASE(4);
#if SQOA_ENC_LOOKBACK
        const uint32_t pos = pos_lookback<W>(s_pos, s_agg, j, st);
        PHASE(5);
        chain_post(&s_pos[(j / W) & 1][w], j + 1, pos + sbase, lane);
#else
        const uint32_t pos = chain_wait<QOI>(&s_pos[(j - 1) % W], j, st);
        PHASE(5);
        chain_post(&s_pos[w], j + 1, pos + sbase, lane);
#endif
#ifndef SQOA_DBG_NOFLUSH
#ifdef SQOA_DBG_ALIGNED_POS   // timing experiments only (wrong output): every flush at a 16-byte-aligned address
        flush_stage(stage, sbase, dst + (pos & ~15u), lane);
#else
        flush_stage(stage, sbase, dst + pos, lane);
#endif
#endif
        PHASE(6);
        }
    };

    const uint32_t nfull = N / ENC_SUB;  // sub-tiles that lie fully inside the image
    if (!mono) {
        uint32_t j = w;
        if (j < nfull) { NbrPx b_, a_; fetch(std::integral_constant<int, 0>{}, j, b_, a_, cur_raw3); take(cur, cur_raw3); cur_pb = b_.value(); cur_pa = a_.value(); }
#if !defined(SQOA_ENC_PRIO) || SQOA_ENC_PRIO
        // Progress-based wave priority.  The SIMD arbiter is oldest-first: of the images sharing a CU the oldest runs at
        // almost solo speed and the youngest finishes alone on a nearly idle CU (tools/trace_encode.py shows the staircase).
        // Lowering an image's priority as it advances keeps co-resident images within a quarter of each other, so they
        // finish together and the CU stays full to the end.
        const uint32_t thr1 = (uint32_t)((unsigned long long)nfull * SQOA_PRIO_T1 >> 6), thr2 = (uint32_t)((unsigned long long)nfull * SQOA_PRIO_T2 >> 6),
                       thr3 = (uint32_t)((unsigned long long)nfull * SQOA_PRIO_T3 >> 6);
        uint32_t thr = thr1, q = 0;
        __builtin_amdgcn_s_setprio(3);
        // MID instance of the body: not the wave's first sub-tile (j > 0), a full sub-tile behind it, and another one of this
        // wave's to prefetch (j + W < nfull - 1 ... all three known at compile time); the few others take the general instance
        const uint32_t nmid = nfull > (uint32_t)W + 1u ? nfull - (uint32_t)W - 1u : 0u;  // j + W is interior as well
#if !defined(SQOA_ENC_L8) || SQOA_ENC_L8
#ifndef SQOA_ENC_QOI_L8
#define SQOA_ENC_QOI_L8 1
#endif
        const bool l8 = (!QOI || (QL8 && SQOA_ENC_QOI_L8)) && (ch == 3 || (ch == 4 && (job.src & 3ull) == 0));  // the hot loop of colour images runs the L8 body (4-byte pixels: dword-aligned)
#else
        const bool l8 = false;
#endif
        if (j < nfull) { body(std::true_type{}, std::integral_constant<int, 0>{}, j, j + W < nfull, l8 && j + W < nmid); j += W; }
        auto hot_loop = [&](auto kind_tag) {
            for (; j < nmid; j += W) {
                if (j >= thr) {
                    ++q; thr = q == 1 ? thr2 : q == 2 ? thr3 : ~0u;
                    if (q == 1) __builtin_amdgcn_s_setprio(2);
                    else if (q == 2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                body(std::true_type{}, kind_tag, j, true);
            }
        };
        {
            // the hot loop is the L8 body (round 3: QOI-compat too); 4-byte pixels at an address that is not dword-aligned (rare) stay on the general one
            auto loop8 = [&](auto px3_tag) {
                for (; j < nmid; j += W) {
                    if (j >= thr) {
                        ++q; thr = q == 1 ? thr2 : q == 2 ? thr3 : ~0u;
                        if (q == 1) __builtin_amdgcn_s_setprio(2);
                        else if (q == 2) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                    body8(px3_tag, j, j + W < nmid);
                }
            };
            if (l8 && ch == 4) loop8(std::false_type{});
            else if (l8) loop8(std::true_type{});
            else if constexpr (QOI) {
                if (ch == 4) hot_loop(std::integral_constant<int, 2>{});
                else hot_loop(std::integral_constant<int, 1>{});
            }
        }
        for (; j < nfull; j += W) body(std::true_type{}, std::integral_constant<int, 0>{}, j, j + W < nfull);  // the image's last full sub-tile
#else
        for (; j < nfull; j += W) body(std::true_type{}, std::integral_constant<int, 0>{}, j, j + W < nfull);
#endif
        if (nsub > nfull && nfull % W == (uint32_t)w) {  // the image's partial last sub-tile
            NbrPx b_, a_;
            fetch(std::integral_constant<int, 0>{}, nfull, b_, a_, cur_raw3);
            take(cur, cur_raw3);
            cur_pb = b_.value(); cur_pa = a_.value();
            body(std::false_type{}, std::integral_constant<int, 0>{}, nfull, false);
        }
    } else {
        uint32_t j = w;
        if (j < nsub) { NbrPx b_, a_; fetch(std::integral_constant<int, 0>{}, j, b_, a_, cur_raw3); take(cur, cur_raw3); cur_pb = b_.value(); cur_pa = a_.value(); }
        for (; j < nsub; j += W) body(std::false_type{}, std::integral_constant<int, 0>{}, j, j + W < nsub);
    }
    if (st.bad && lane == 0) atomicMax(&results[jid].status, SQOA_HIP_ST_INTERNAL);
#ifdef SQOA_TRACE
    if (threadIdx.x == 0) { for (int k = 0; k < 8; ++k) trace[-8 + k] = ph[k]; }
#endif
}

// ---- launch -----------------------------------------------------------------------------

#ifndef SQOA_LDS_ORDER_FB
extern "C" int sqoa_lds_order_fb(void);   // sqoa_selftest.hip: the order-free QOI-compat kernels on the current device?
extern "C" hipError_t sqoa_launch_encode_tab_fb(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                                sqoa_hip_result* d_results, const uint32_t* d_qtab_in, int cus, hipStream_t stream);
extern "C" hipError_t sqoa_launch_encode_small_fb(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                                  sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_encode_tab(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                             sqoa_hip_result* d_results, const uint32_t* d_qtab_in, int cus, hipStream_t stream);
// `cus`: compute units of the device the plan lives on (the plan asks the device it was created for, not the calling thread's current one)
extern "C" hipError_t sqoa_launch_encode(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                         sqoa_hip_result* d_results, int cus, hipStream_t stream) {
    return sqoa_launch_encode_tab(d_jobs, d_ids, n, qoi, d_results, nullptr, cus, stream);
}
#endif

// `d_qtab_in` (QOI-compat strip jobs only, else null): the colour index entering job i, 64 dwords at [64 * i]
extern "C" hipError_t SQOA_ENC_SYM(sqoa_launch_encode_tab)(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                                           sqoa_hip_result* d_results, const uint32_t* d_qtab_in, int cus, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
#ifndef SQOA_LDS_ORDER_FB
    if (qoi && sqoa_lds_order_fb()) return sqoa_launch_encode_tab_fb(d_jobs, d_ids, n, qoi, d_results, d_qtab_in, cus, stream);
#else
    if (!qoi) return hipErrorInvalidValue;   // (this build holds the QOI-compat kernels only)
#endif
    // 8 waves per image, four images per CU: the kernel is VALU-issue bound (tools/ubench/valu_table.hip)
#ifndef SQOA_ENC_W
#define SQOA_ENC_W 8
#endif
    constexpr int W = SQOA_ENC_W;
    // QOI-compat: 112 VGPRs and 76 KB of LDS per eight-wave workgroup make that two images per CU.  From more images than that
    // on, four waves per image and four images per CU -- the same 16 waves -- do better: the colour index is a serial chain
    // through an image's sub-tiles, and twice the images in flight hide more of it (1024 x 1024 x 1024 RGBA 3.05 -> 2.97 ms,
    // 4096 icons of 512 x 512 RGB 3.10 -> 2.93 ms; two waves per image: 4.6 / 3.9 ms).  Fewer images than 2 per CU, and
    // strips (a workgroup per CU-sized piece of one image): eight waves (256 images 1.37 against 1.98 ms).
    if (qoi) {
        if (cus <= 0) cus = 256;
#ifndef SQOA_ENC_W_QOI_MANY
#define SQOA_ENC_W_QOI_MANY 4
#endif
        constexpr int WM = SQOA_ENC_W_QOI_MANY;
        if (!d_qtab_in && n > 2 * cus) hipLaunchKernelGGL((encode_kernel<WM, true>), dim3(n), dim3(WM * 64), 0, stream, d_jobs, d_ids, d_results, d_qtab_in);
        else hipLaunchKernelGGL((encode_kernel<W, true>), dim3(n), dim3(W * 64), 0, stream, d_jobs, d_ids, d_results, d_qtab_in);
    }
#ifndef SQOA_LDS_ORDER_FB
    else hipLaunchKernelGGL((encode_kernel<W, false>), dim3(n), dim3(W * 64), 0, stream, d_jobs, d_ids, d_results, (const uint32_t*)nullptr);
#endif
    return hipGetLastError();
}

// Many SMALL images (the plan decides: sqoa_runtime.hip): one wave per image.  An image of a few sub-tiles gives eight waves
// nothing to share but the chain hops and the workgroup's prologue: 16 384 icons of 128 x 128 0.65 -> 0.41 ms, 4096 of
// 64 x 64 66 -> 45 us; from 256 x 256 on the shapes are equal (tools/time_tiny.py).
extern "C" hipError_t SQOA_ENC_SYM(sqoa_launch_encode_small)(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int qoi,
                                                             sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
#ifndef SQOA_LDS_ORDER_FB
    if (qoi && sqoa_lds_order_fb()) return sqoa_launch_encode_small_fb(d_jobs, d_ids, n, qoi, d_results, stream);
#else
    if (!qoi) return hipErrorInvalidValue;
#endif
    if (qoi) hipLaunchKernelGGL((encode_kernel<1, true>), dim3(n), dim3(64), 0, stream, d_jobs, d_ids, d_results, (const uint32_t*)nullptr);
#ifndef SQOA_LDS_ORDER_FB
    else hipLaunchKernelGGL((encode_kernel<1, false>), dim3(n), dim3(64), 0, stream, d_jobs, d_ids, d_results, (const uint32_t*)nullptr);
#endif
    return hipGetLastError();
}

SQOA_ENC_NS_CLOSE
