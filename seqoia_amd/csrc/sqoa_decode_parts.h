// sqoa_decode_parts.h -- pieces shared by the wave-parallel SQOA decoder (sqoa_decode_wave.hip) and the one-wave-per-image
// QOI / fallback decoder (sqoa_decode.hip): the byte-stream parser as function composition, LDS windows, packed-byte adds.
#pragma once

#include "sqoa_common.h"

namespace sqoa {

constexpr uint32_t ID_LO = 0x03020100u, ID_HI = 0x07060504u;  // identity transition function

enum : uint32_t { ST_A = 0, ST_N = 1, ST_X = 6 };

struct Fn { uint32_t lo, hi; };

// (g after f)(s) = g[f(s)]: g is the table, f the selector
SQOA_DEV Fn compose(Fn g, Fn f) {
    return Fn{__builtin_amdgcn_perm(g.hi, g.lo, f.lo), __builtin_amdgcn_perm(g.hi, g.lo, f.hi)};
}
SQOA_DEV uint32_t apply(Fn f, uint32_t st) { return (uint32_t)(((unsigned long long)f.hi << 32 | f.lo) >> (8u * st)) & 0xffu; }
// The same as ONE v_perm_b32 (a function is eight state bytes, the state is the byte selector) instead of a 64-bit shift, for the
// per-lane application in the decoders' hot loops; and what follows it there -- the offset of my first op start among my LB bytes from the
// parser state in front of them: A, N -> 0, inside an op (D1..D4) -> its 1..4 remaining bytes, X (past the chunk area) -> LB -- as a
// table of seven bytes in two registers instead of two range tests and three selects.
SQOA_DEV uint32_t apply_perm(Fn f, uint32_t st) { return __builtin_amdgcn_perm(f.hi, f.lo, st) & 0xffu; }
template <int LB>
SQOA_DEV uint32_t first_op_offset(uint32_t lst) { return __builtin_amdgcn_perm((uint32_t)LB << 16 | 0x0403u, 0x02010000u, lst) & 0xffu; }

template <int CTRL, int ROWMASK>
SQOA_DEV Fn scan_step(Fn F) {
    Fn P;
    P.lo = (uint32_t)__builtin_amdgcn_update_dpp((int)ID_LO, (int)F.lo, CTRL, ROWMASK, 0xf, false);
    P.hi = (uint32_t)__builtin_amdgcn_update_dpp((int)ID_HI, (int)F.hi, CTRL, ROWMASK, 0xf, false);
    return compose(F, P);
}

// inclusive composition over the wave: result of lane i = f_i after f_{i-1} after ... after f_0
SQOA_DEV Fn wave_compose_scan(Fn F) {
    F = scan_step<DPP_ROW_SHR + 1, 0xf>(F);
    F = scan_step<DPP_ROW_SHR + 2, 0xf>(F);
    F = scan_step<DPP_ROW_SHR + 4, 0xf>(F);
    F = scan_step<DPP_ROW_SHR + 8, 0xf>(F);
    F = scan_step<DPP_ROW_BCAST15, 0xa>(F);
    F = scan_step<DPP_ROW_BCAST31, 0xc>(F);
    return F;
}

// 8 stream bytes starting at byte offset `off` of a dword array in LDS.  Unaligned ds_read_b64 works on gfx950 but
// stalls the LDS pipe (SQ_LDS_UNALIGNED_STALL); three aligned dwords and two v_alignbyte_b32 do not.
SQOA_DEV unsigned long long lds_read8(const uint32_t* base, uint32_t off) {
    const uint32_t* p = base + (off >> 2);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, off & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, off & 3u);
    return (unsigned long long)hi << 32 | lo;
}

// x >> 31 (arithmetic): 0xffffffff where x is "negative".  Opaque to the optimiser on purpose: it would turn the mask
// arithmetic of the walk back into v_cmp + v_cndmask pairs, which cost 1.7x the add / and / shift they replace
// (tools/ubench/valu_table.hip: add, sub, and, or, xor, not, mov, lshr, ashr and bitop3 on VGPR / literal operands issue
// at ~1.0 ns per wave and SIMD, every other VALU instruction -- and any of those with an SGPR operand -- at ~1.73 ns).
SQOA_DEV uint32_t sra31(uint32_t x) {
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    return r;
}
typedef const __attribute__((address_space(3))) uint32_t* lcptr32;
typedef const __attribute__((address_space(3))) u32x4* lcptr128;

// bytewise (mod 256) add / sub of four packed channels
SQOA_DEV uint32_t add8(uint32_t a, uint32_t b) { return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u); }
// The same with the two SWAR constants handed in, for callers that keep them in VGPRs: as literals the compiler parks them in SGPRs,
// and a three-operand instruction with an SGPR operand is slow-class -- two of add8's five instructions.
struct HL { uint32_t H, L; };
SQOA_DEV uint32_t add8(uint32_t a, uint32_t b, HL k) { return ((a & k.L) + (b & k.L)) ^ ((a ^ b) & k.H); }
SQOA_DEV uint32_t sub8(uint32_t a, uint32_t b) { return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ ((a ^ ~b) & 0x80808080u); }

// one colour pixel at pixel index P; OC = 3 or 4 output channels, 40 = four channels at an address that is not dword-aligned
template <int OC>
SQOA_DEV void put_pixel(gptr8 out, uint32_t P, uint32_t v) {
    if (OC == 4) ((gptr32)out)[P] = v;
    else {
        gptr8 o = out + (size_t)P * (OC == 3 ? 3 : 4);
        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
        if (OC == 40) o[3] = (uint8_t)(v >> 24);
    }
}
// fill out[p0, p0+cnt) (pixel units) with v
template <int OC>
SQOA_DEV void fill_pixels(gptr8 out, uint32_t p0, uint32_t cnt, uint32_t v, int lane) {
    for (uint32_t k = lane; k < cnt; k += 64) put_pixel<OC>(out, p0 + k, v);
}

// A pixel transform: v8 holds 4 packed channels, `abs` the byte mask of channels that are absolute;
// the other channels are deltas (mod 256) still to be added to whatever pixel enters.
struct Xf { uint32_t v, abs; };
// first `a`, then `b`
// (b.v holds the value itself in b's absolute channels and a delta elsewhere, so masking a's side first is enough)
SQOA_DEV Xf xf_then(Xf a, Xf b) { return Xf{add8(a.v & ~b.abs, b.v), a.abs | b.abs}; }
SQOA_DEV uint32_t xf_apply(Xf t, uint32_t px) { return add8(px & ~t.abs, t.v); }
SQOA_DEV Xf xf_then(Xf a, Xf b, HL k) { return Xf{add8(a.v & ~b.abs, b.v, k), a.abs | b.abs}; }
SQOA_DEV uint32_t xf_apply(Xf t, uint32_t px, HL k) { return add8(px & ~t.abs, t.v, k); }

}  // namespace sqoa
