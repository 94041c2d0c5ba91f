// sqoa_decode_wave.hip -- wave-parallel SQOA/QOI decoder for colour streams (gfx950).
//
// Replaces the serial parse of the reference decoder (seqoia.h:722-806) for streams whose ops
// are RGB/RGBA, LUMA(+ALPHA suffix), RUN/BIGRUN (SQOA) -- i.e. everything the encoder emits.
// Streams with SQOA_OP_REF, mono streams, QOI-compat streams and 1/2-channel outputs are handed to
// decode_fallback_kernel (sqoa_decode.hip) by marking the job SQOA_HIP_ST_SLOWPATH.
//
// Dataflow per image (one workgroup, W waves; chunk c of 512 [768] stream bytes belongs to wave c mod W, and
// inside a chunk every lane owns LB = 8 [12] consecutive stream bytes):
//   1. parse: every byte has a transition function over the 7 parser states
//      {A: op boundary after an op, N: op boundary where an ALPHA suffix is not allowed,
//       D1..D4: inside an op, X: past the chunk area}; functions are 8 bytes (table in LDS), composition is
//      two v_perm_b32.  A lane composes its 8 bytes, one 6-step DPP scan composes across lanes.  The
//      chunk's total function is usually constant (the byte stream re-synchronises), so the exit state
//      is published before the entry state is even known: the parse chain does not serialise.
//   2. walk: knowing the state in front of its first byte, a lane decodes its own <= 8 ops in order,
//      each from 8 bytes of LDS (op byte, <= 4 data bytes, possible ALPHA suffix): LDS table look-ups and
//      mask arithmetic, no compare / select chains (they are slow-class VALU).  A pixel
//      transform is (v, abs): per-channel value, absolute or still a delta mod 256; the lane keeps the
//      transform after each op and its pixel counts.
//   3. scans: one DPP prefix sum of the lanes' pixel counts, one DPP scan composing the lanes' transforms.
//   4. (pixel position, pixel value) entering the chunk arrive through the LDS chain; literal ops
//      store their pixel directly, runs are expanded cooperatively by the whole wave.
#include <type_traits>

#include "sqoa_decode_parts.h"

#ifndef SQOA_PRIO_T1  // progress (in 64ths of the image) at which a workgroup drops to wave priority 2, 1, 0
#define SQOA_PRIO_T1 40
#define SQOA_PRIO_T2 54
#define SQOA_PRIO_T3 61
#endif

namespace sqoa {

// (SQOA_DEC_LB, sqoa_common.h: the chunk geometry of the three passes and their tables; the chained kernel also runs at 12, SQOA_DEC_LB0 below)
constexpr int DEC_LB = SQOA_DEC_LB;            // stream bytes owned by one lane (passes)
constexpr int DEC_CHUNK = DEC_LB * 64;         // stream bytes per chunk (one wave step; passes)
// (staged dwords per wave: chunk + 8 look-ahead + 3 misalignment, rounded: DEC_LDS_DW, DEC_NLD dword loads per lane, in the kernel)
[[maybe_unused]] constexpr uint32_t DEC_SPIN_MAX = 1u << 22;

// Wait for `slot` to carry `tag`; 0 = it does (val set), 1 = the image was handed over (the slot's owner left DEC_ABORT_TAG in
// it), 2 = timed out.  Integer result codes, not bool: a wave-uniform i1 that flows through a merge is kept as a 64-bit lane
// mask by the compiler, and the scalar pipe is as busy as the vector pipe in this kernel -- for the same reason a failed poll
// is as few scalar instructions as it can be: no separate abort flag to look at (a wave that hands the image over, or learns
// that its predecessor did, writes the abort tag into its OWN three chain words and leaves; every wave waits on its
// predecessor's words sooner or later, so the news goes round the ring), the spin limit a down-counter.
constexpr uint32_t DEC_ABORT_TAG = 0xffffffffu;   // (chunk tags are chunk numbers + 1 < 2^32 - 1)
template <int SLEEP = SQOA_DEC_SPIN_SLEEP>
SQOA_DEV uint32_t chain_wait_rc(unsigned long long* slot, uint32_t tag, uint32_t& val) {
#ifdef SQOA_DEC_C_POLL   // the loop as the compiler makes it (A/B builds)
    uint32_t left = DEC_SPIN_MAX;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t t = bcast_first((uint32_t)(v >> 32));
        if (t == tag) { val = bcast_first((uint32_t)v); return 0u; }
        if (t == DEC_ABORT_TAG) return 1u;
        if (--left == 0u) return 2u;
        __builtin_amdgcn_s_sleep(SLEEP);
    }
#else
    const uint32_t t = chain_poll<SLEEP>(slot, tag, DEC_ABORT_TAG, val);
    return t == tag ? 0u : t == DEC_ABORT_TAG ? 1u : 2u;
#endif
}

SQOA_DEV void post(unsigned long long* slot, uint32_t tag, uint32_t val, int lane) { lds_post(slot, tag, val, lane); }

// for (i = I; i < N; ++i) if (!f(i)) break; with i a compile-time constant: the per-op register arrays of the walk are indexed by it,
// and a loop the compiler declines to unroll fully (it did at 12 ops per lane) sends them to scratch memory
template <int I, int N, class F>
SQOA_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        if (f(std::integral_constant<int, I>{})) static_for<I + 1, N>(f);
    }
}

template <int CTRL, int ROWMASK>
SQOA_DEV Xf xf_scan_step(Xf T, HL k) {
    Xf P;
    constexpr bool BC = ROWMASK == 0xf;
    P.v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v, CTRL, ROWMASK, 0xf, BC);
    P.abs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.abs, CTRL, ROWMASK, 0xf, BC);
    return xf_then(P, T, k);
}
SQOA_DEV Xf wave_xf_scan(Xf T, HL k) {
    T = xf_scan_step<DPP_ROW_SHR + 1, 0xf>(T, k);
    T = xf_scan_step<DPP_ROW_SHR + 2, 0xf>(T, k);
    T = xf_scan_step<DPP_ROW_SHR + 4, 0xf>(T, k);
    T = xf_scan_step<DPP_ROW_SHR + 8, 0xf>(T, k);
    T = xf_scan_step<DPP_ROW_BCAST15, 0xa>(T, k);
    T = xf_scan_step<DPP_ROW_BCAST31, 0xc>(T, k);
    return T;
}
template <int CTRL, int ROWMASK>
SQOA_DEV Xf xf_scan_step(Xf T) {
    Xf P;  // the transform of the lanes below, identity (all zero) where there is none
    // (all rows enabled: a source lane out of range reads as zero with bound_ctrl, no v_mov of the zero in front of the DPP move)
    constexpr bool BC = ROWMASK == 0xf;
    P.v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v, CTRL, ROWMASK, 0xf, BC);
    P.abs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.abs, CTRL, ROWMASK, 0xf, BC);
    return xf_then(P, T);
}
SQOA_DEV Xf wave_xf_scan(Xf T) {
    T = xf_scan_step<DPP_ROW_SHR + 1, 0xf>(T);
    T = xf_scan_step<DPP_ROW_SHR + 2, 0xf>(T);
    T = xf_scan_step<DPP_ROW_SHR + 4, 0xf>(T);
    T = xf_scan_step<DPP_ROW_SHR + 8, 0xf>(T);
    T = xf_scan_step<DPP_ROW_BCAST15, 0xa>(T);
    T = xf_scan_step<DPP_ROW_BCAST31, 0xc>(T);
    return T;
}


// ---- chip-wide decode of few large streams: the tables of the three passes (MODE 1..3 of decode_wave_kernel) -----------------
// (dp_image, dp_wg: sqoa_common.h)
// What the LDS chain carries, per workgroup (DP_CHUNKS_PER_WG chunks), in HBM: the composite of its chunks' transition functions,
// pixel counts and transforms (written by passes 1 and 2), and what enters its first chunk (written by the two scans).
// (dp_chunk: sqoa_common.h)
static_assert(DEC_CHUNK == DP_CHUNK_BYTES, "the passes' chunk size");
#ifndef SQOA_DP_PER_MIN
#define SQOA_DP_PER_MIN 8   /* one chunk per wave: a lone 1 Mpixel stream 0.040 -> 0.036 ms; from 8 streams on no difference */
#endif
constexpr uint32_t DP_CHUNKS_PER_WG_MIN = SQOA_DP_PER_MIN, DP_CHUNKS_PER_WG_MAX = 64;   // chunks per workgroup of a pass: the plan picks (sqoa_dp_layout)

// The two scans between the passes: one workgroup of 1024 threads per image, every thread a contiguous run of the image's
// table rows -- its run composed serially (independent loads), one scan over the 1024 run totals (DPP inside the waves, LDS
// across the 16 of them), its run walked again with the prefix.  (One wave stepping through 64 chunks at a time was a chain of
// dependent load latencies: 0.6 ms of the 0.69 ms a lone 3840x2160 stream took.)
constexpr int DP_SCAN_THREADS = 1024;

// Entry state of every chunk: the chunks' transition functions composed in stream order.
__global__ __launch_bounds__(DP_SCAN_THREADS) void dp_scan_state(const dp_image* __restrict__ images, dp_chunk* __restrict__ tabs, const uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_lo[DP_SCAN_THREADS / 64], s_hi[DP_SCAN_THREADS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const dp_image im = images[blockIdx.x];
    const uint32_t flag = flags[blockIdx.x];
    if (flag != 1u && flag != 3u) return;
    dp_chunk* tab = tabs + im.tab_off;
    const uint32_t per = (im.nwgs + DP_SCAN_THREADS - 1) / DP_SCAN_THREADS;
    const uint32_t c0 = t * per < im.nwgs ? t * per : im.nwgs, c1 = c0 + per < im.nwgs ? c0 + per : im.nwgs;
    Fn F{ID_LO, ID_HI};
#pragma unroll 4
    for (uint32_t c = c0; c < c1; ++c) F = compose(Fn{tab[c].g_lo, tab[c].g_hi}, F);
    const Fn Fi = wave_compose_scan(F);
    if (lane == 63u) { s_lo[wv] = Fi.lo; s_hi[wv] = Fi.hi; }
    __syncthreads();
    Fn P{ID_LO, ID_HI};
    for (uint32_t k = 0; k < wv; ++k) P = compose(Fn{s_lo[k], s_hi[k]}, P);
    const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
    uint32_t st = apply(compose(Fx, P), flag == 3u ? (uint32_t)ST_A : (uint32_t)ST_N);   // ST_N: stream start of an SQOA stream (seqoia.h:773-775)
    for (uint32_t c = c0; c < c1; ++c) {
        const Fn G{tab[c].g_lo, tab[c].g_hi};
        tab[c].st_in = st;
        st = apply(G, st);
    }
}

template <int CTRL, int ROWMASK>
SQOA_DEV uint32_t dp_sat_step(uint32_t v, uint32_t cap) {
    const uint32_t s = v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
    return s > cap ? cap : s;
}

// forward decl: the transform scan lives below
SQOA_DEV Xf wave_xf_scan(Xf T);

// Pixel position and entering pixel of every chunk: prefix sum of the pixel counts (saturating at N: counts of crafted streams
// overflow 32 bits otherwise) and prefix composition of the transforms.
__global__ __launch_bounds__(DP_SCAN_THREADS) void dp_scan_pixels(const sqoa_hip_job* __restrict__ jobs, const dp_image* __restrict__ images,
                                                                  dp_chunk* __restrict__ tabs, const uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_c[DP_SCAN_THREADS / 64], s_v[DP_SCAN_THREADS / 64], s_a[DP_SCAN_THREADS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const dp_image im = images[blockIdx.x];
    if (flags[blockIdx.x] != 1u && flags[blockIdx.x] != 3u) return;
    gcptr8 src = (gcptr8)jobs[im.jid].src;
    const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
    const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
    const uint32_t N = hw * hh;
    auto sat = [&](uint32_t a, uint32_t b) { const uint32_t x = a + b; return x > N ? N : x; };   // (a, b <= N < 2^31)
    dp_chunk* tab = tabs + im.tab_off;
    const uint32_t per = (im.nwgs + DP_SCAN_THREADS - 1) / DP_SCAN_THREADS;
    const uint32_t c0 = t * per < im.nwgs ? t * per : im.nwgs, c1 = c0 + per < im.nwgs ? c0 + per : im.nwgs;
    uint32_t cnt = 0;
    Xf T{0u, 0u};
#pragma unroll 4
    for (uint32_t c = c0; c < c1; ++c) {
        cnt = sat(cnt, tab[c].ctot);
        T = xf_then(T, Xf{tab[c].tv, tab[c].tabs});
    }
    uint32_t v = cnt;
    v = dp_sat_step<DPP_ROW_SHR + 1, 0xf>(v, N);
    v = dp_sat_step<DPP_ROW_SHR + 2, 0xf>(v, N);
    v = dp_sat_step<DPP_ROW_SHR + 4, 0xf>(v, N);
    v = dp_sat_step<DPP_ROW_SHR + 8, 0xf>(v, N);
    v = dp_sat_step<DPP_ROW_BCAST15, 0xa>(v, N);
    v = dp_sat_step<DPP_ROW_BCAST31, 0xc>(v, N);
    const Xf Ti = wave_xf_scan(T);
    if (lane == 63u) { s_c[wv] = v; s_v[wv] = Ti.v; s_a[wv] = Ti.abs; }
    __syncthreads();
    uint32_t pos = 0;
    Xf P{0u, 0u};
    for (uint32_t k = 0; k < wv; ++k) { pos = sat(pos, s_c[k]); P = xf_then(P, Xf{s_v[k], s_a[k]}); }
    pos = sat(pos, shift_up1(v, 0u));                                                  // pixels in front of my run (saturated)
    uint32_t px = xf_apply(xf_then(P, Xf{shift_up1(Ti.v, 0u), shift_up1(Ti.abs, 0u)}), kPxInit);   // pixel entering my run
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t n = tab[c].ctot;
        const Xf Tc{tab[c].tv, tab[c].tabs};
        tab[c].pos_in = pos;
        tab[c].px_in = px;
        pos = sat(pos, n);
        px = xf_apply(Tc, px);
    }
}

// 64 VGPRs = 8 waves per SIMD = four 8-wave streams per CU (see sqoa_encode.hip)
#ifndef SQOA_DEC_WPE
#define SQOA_DEC_WPE 8
#endif
#define SQOA_DEC_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SQOA_DEC_WPE, SQOA_DEC_WPE)))
// MODE 0: one workgroup per image, the chunks chained through LDS (the batch shape: as many images as the chip holds workgroups).
// MODE 1..3: the three passes that decode FEW LARGE streams with the whole chip instead of one compute unit per stream (a
// lone 1 Mpixel stream takes 2.2 ms chained: one CU's instruction issue).  A workgroup takes a range of chunks of one image;
// what the chain carries travels through a per-chunk table in HBM (dp_chunk) between launches:
//   pass 1: parse only -> the chunk's transition function; dp_scan_state then gives every chunk its entry state;
//   pass 2: parse + walk + scans -> the chunk's pixel count and composite transform; dp_scan_pixels gives every chunk the
//           pixel position and the pixel it starts from;
//   pass 3: parse + walk + scans again, and the stores.
// Twice the instructions of MODE 0, spread over every CU.
template <int W, int MODE, int LB = SQOA_DEC_LB, bool SWEEP_T = false>
__global__ __launch_bounds__(W * 64) SQOA_DEC_WPE_ATTR void decode_wave_kernel(const sqoa_hip_job* __restrict__ jobs,
                                                             const int* __restrict__ job_ids,
                                                             sqoa_hip_result* __restrict__ results,
                                                             const dp_image* __restrict__ dp_images, const dp_wg* __restrict__ dp_wgs,
                                                             dp_chunk* __restrict__ dp_tab, uint32_t* __restrict__ dp_flags, uint32_t dp_per,
                                                             uint32_t dp_fused) {
    // (the namespace-scope constants of the same names are the passes' chunk geometry, which the plan's tables are laid out for;
    // the one-workgroup-per-stream kernel, MODE 0, may be built with more stream bytes per lane: SQOA_DEC_LB0)
    constexpr int DEC_LB = LB, DEC_CHUNK = LB * 64, DEC_LDS_DW = DEC_CHUNK / 4 + 8, DEC_NLD = (DEC_LDS_DW + 63) / 64;
    // A lane keeps the records of at most DEC_R = 8 ops in registers, and no more than eight walk iterations can START in twelve
    // bytes of any stream: an iteration takes one byte only when its op is a one-byte RUN without an ALPHA suffix whose successor is
    // not a RUN it could take along (a RUN behind an op or behind another RUN is taken in the same iteration unless it carries a
    // suffix of its own) -- so the next iteration is a LUMA / RGB / RGBA op or a RUN + suffix, two bytes or more, and 1, 2, 1, 2, ...
    // puts the ninth start at byte 12.  (Sixteen bytes per lane would need nine slots for LUMA-dense streams.)  The test behind the
    // walk is a net, not a path.
#ifndef SQOA_DEC_R16
#define SQOA_DEC_R16 10
#endif
    constexpr int DEC_R = LB < 8 ? LB : LB <= 12 ? 8 : SQOA_DEC_R16;
    // s_sleep argument (x 64 cycles) between two looks at a chain word: the 768-byte chunks of the batch shape take longer, and a
    // waiting wave that looks less often leaves the LDS and scalar pipes to the others (3.117 -> 3.092 ms at 4 against 1; the passes
    // and the 512-byte shape keep 1: a lone stream's chains are its critical path)
#ifndef SQOA_DEC_SPIN_SLEEP_WIDE
#define SQOA_DEC_SPIN_SLEEP_WIDE 4
#endif
    constexpr int SPIN = (MODE == 0 && LB > 8) ? SQOA_DEC_SPIN_SLEEP_WIDE : SQOA_DEC_SPIN_SLEEP;
    constexpr int CL_BITS = DEC_R <= 8 ? 13 : 14;                         // pixels in front of an op inside its lane: at most DEC_R - 1 ops of at most 1024 pixels
    constexpr uint32_t CL_MASK = (1u << CL_BITS) - 1u;
    // (the passes, MODE 1..3, run at the chunk size the plan laid their tables out for: sqoa_dp_tables.lb)
    static_assert(LB % 4 == 0 && (DEC_R - 1) * 1024 <= (int)CL_MASK, "bytes per lane");
    __shared__ uint32_t s_bytes[W][DEC_LDS_DW];
    __shared__ uint32_t s_ftab[256];  // byte -> low half of its transition function (live bytes)
    // op byte -> {info, absolute-channel mask, LUMA r/b base, LUMA g | nibble mask}; entry 256 = "no op" (lanes out of ops)
    //   info: len (3) | (len*8 - 4) (6) << 3 | pixels (10) << 9 | REF (1) << 20 | pixels if the byte is a RUN / BIGRUN (10) << 21
    //   (the stream window is shifted 4 bits short of the op's end, so that the following bytes sit at bits 4.., 12..:
    //    byte * 16 and byte * 8, the table strides, are then a mask or a right shift away -- v_lshlrev_b32 is slow-class)
    __shared__ __attribute__((aligned(16))) u32x4 s_optab[257];
    // byte -> {ALPHA suffix: delta << 24 | 8, else 0 ;  0 for an ALPHA suffix byte, else ~0}
    __shared__ __attribute__((aligned(8))) uint32_t s_sfx[256][2];
    // A/B builds (SQOA_DEC_TAB4=1): the walk's three one-dword look-ups by a stream byte -- the ALPHA suffix behind the op, "is the
    // byte behind that a RUN", "is the byte behind the RUN an ALPHA suffix" -- from tables of their own at a 4-byte stride: at the 16-
    // and 8-byte strides of s_optab and s_sfx a dword read of 32 lanes can only fall on 8 resp. 16 of the 32 LDS banks (bank = dword
    // address mod 32), and the kernel's SQ_LDS_BANK_CONFLICT is 1.5 x its SQ_ACTIVE_INST_LDS.  Measured: 3.178 vs 3.185 ms on the
    // headline batch, 1.787 vs 1.782 ms on the icons -- the LDS pipe's conflicts are not what the waves wait for.  Off.
#ifndef SQOA_DEC_TAB4
#define SQOA_DEC_TAB4 0
#endif
    __shared__ uint32_t s_t4[SQOA_DEC_TAB4 ? 3 : 1][SQOA_DEC_TAB4 ? 256 : 1];   // [0]: s_sfx[.][0], [1]: run pixels << 21, [2]: s_sfx[.][1]
    __shared__ unsigned long long s_parse[2][W];
    __shared__ unsigned long long s_pos[W];
    __shared__ unsigned long long s_px[W];
    __shared__ uint32_t s_path;  // 1 = wave path, 2 = refused
    // The position-major store path of chunks with runs (below): per wave, the pixel of every op of the chunk by its rank in
    // stream order (+ 64 dump words for op slots without an op), and one bit per pixel of the chunk: "an op starts here".
#ifndef SQOA_DEC_SWEEP
#define SQOA_DEC_SWEEP 1
#endif
#ifndef SQOA_DEC_EXEC_WALK
#define SQOA_DEC_EXEC_WALK 1
#endif
    constexpr bool SWEEP = SQOA_DEC_SWEEP && SWEEP_T && MODE == 0;
#ifndef SQOA_DEC_SWEEP_MIN
#define SQOA_DEC_SWEEP_MIN 512
#endif
    constexpr uint32_t SWEEP_MIN_PX = SQOA_DEC_SWEEP_MIN;        // ... and must: a chunk of a few hundred pixels with a run or two in it is cheaper run by run
    constexpr uint32_t SWEEP_PX = 4096;                          // pixels a chunk may produce on this path (longer: runs of hundreds of pixels, the fill loop's home ground)
    __shared__ uint32_t s_swv[SWEEP ? W : 1][SWEEP ? DEC_R * 64 + 64 : 1];
    __shared__ __attribute__((aligned(8))) uint2 s_swb[SWEEP ? W : 1][SWEEP ? SWEEP_PX / 32 : 1];   // .x: 32 start bits, .y: the starts in front of them

    SQOA_PLACEMENT_PAD();
    const int lane = lane_id();
    const int w = (int)bcast_first(threadIdx.x >> 6);
    uint32_t img = 0, chunk_first = 0, tab_off = 0;   // MODE > 0: the image and the chunk range of this workgroup
    int jid;
    if constexpr (MODE == 0) jid = job_ids[blockIdx.x];
    else {
        img = dp_wgs[blockIdx.x].img;
        chunk_first = dp_wgs[blockIdx.x].chunk0;
        jid = dp_images[img].jid;
        tab_off = dp_images[img].tab_off;
    }
    const sqoa_hip_job job = jobs[jid];
    sqoa_hip_result* res = &results[jid];
    // MODE 0 launched as the RETRY behind the balanced kernel (dp_per = 1; sqoa_launch_decode_standby): only the streams that kernel handed
    // on -- a chunk with more op starts than its 512 slots: a stretch of two-byte ops with a one-byte op in it, BIGRUN after BIGRUN; this
    // kernel's walk takes those along (DEC_R below) -- start again here, from their first byte; the others cost a workgroup one scalar load.
    // (Streams this kernel hands on in turn -- SQOA_OP_REF, mono, QOI-compat -- are marked as in a first launch, for the kernels behind.)
    const bool retry = MODE == 0 && dp_per == 1u;
    if (retry && __builtin_amdgcn_readfirstlane(res->status) != SQOA_HIP_ST_SLOWPATH) return;
    gcptr8 src = (gcptr8)job.src;
    gptr8 out = (gptr8)job.dst;
    const uint32_t size = job.src_size;
    // MODE > 0: this workgroup's row of the table; what enters its first chunk (the scans wrote it; MODE 0: the stream start)
    dp_chunk* row = MODE ? dp_tab + tab_off + chunk_first / dp_per : nullptr;
    __shared__ uint32_t s_wg[3][MODE ? DP_CHUNKS_PER_WG_MAX : 1];   // passes 1 and 2: per chunk of the workgroup, what goes into its row
    uint32_t seed_st = ST_N, seed_pos = 0, seed_px = kPxInit;   // ST_N: no preceding op, an ALPHA byte would be a run (seqoia.h:773-775)
    // What enters this workgroup's first chunk.  Either the two scan kernels between the passes have left it in the row (dp_fused
    // = 0), or -- few rows, no QOI-compat stream: the launches of those kernels, 5 us each, are a quarter of a lone 1 Mpixel
    // stream's time -- every workgroup composes the rows in front of its own by itself: pass 2 the transition functions pass 1
    // left (and keeps the state for pass 3), pass 3 the pixel counts and transforms pass 2 left.  (No workgroup waits for
    // another: the rows were written by the launch before.)
    if (MODE >= 2 && dp_fused && chunk_first) {
        __shared__ uint32_t s_red[3][W];
        const dp_chunk* tab = dp_tab + tab_off;
        const uint32_t myrow = chunk_first / dp_per, t = threadIdx.x, per = (myrow + W * 64 - 1) / (W * 64);
        const uint32_t r0 = t * per < myrow ? t * per : myrow, r1 = r0 + per < myrow ? r0 + per : myrow;
        if constexpr (MODE == 2) {
            Fn F{ID_LO, ID_HI};
            for (uint32_t r = r0; r < r1; ++r) F = compose(Fn{tab[r].g_lo, tab[r].g_hi}, F);
            const Fn Fi = wave_compose_scan(F);
            if (lane == 63) { s_red[0][w] = Fi.lo; s_red[1][w] = Fi.hi; }
            __syncthreads();
            Fn P{ID_LO, ID_HI};
            for (int k = 0; k < W; ++k) P = compose(Fn{s_red[0][k], s_red[1][k]}, P);
            seed_st = bcast_first(apply(P, bcast_first((uint32_t)src[14]) != 0x31u ? (uint32_t)ST_A : (uint32_t)ST_N));
            if (threadIdx.x == 0) row->st_in = seed_st;
        } else {
            const uint32_t hw_ = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
            const uint32_t hh_ = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
            const uint32_t Ncap = hw_ * hh_;
            uint32_t cnt = 0;
            Xf T{0u, 0u};
            for (uint32_t r = r0; r < r1; ++r) {
                const uint32_t x = cnt + tab[r].ctot;   // (both <= N < 2^31)
                cnt = x > Ncap ? Ncap : x;
                T = xf_then(T, Xf{tab[r].tv, tab[r].tabs});
            }
            uint32_t v = cnt;
            v = dp_sat_step<DPP_ROW_SHR + 1, 0xf>(v, Ncap);
            v = dp_sat_step<DPP_ROW_SHR + 2, 0xf>(v, Ncap);
            v = dp_sat_step<DPP_ROW_SHR + 4, 0xf>(v, Ncap);
            v = dp_sat_step<DPP_ROW_SHR + 8, 0xf>(v, Ncap);
            v = dp_sat_step<DPP_ROW_BCAST15, 0xa>(v, Ncap);
            v = dp_sat_step<DPP_ROW_BCAST31, 0xc>(v, Ncap);
            const Xf Ti = wave_xf_scan(T);
            if (lane == 63) { s_red[0][w] = v; s_red[1][w] = Ti.v; s_red[2][w] = Ti.abs; }
            __syncthreads();
            uint32_t pos = 0;
            Xf P{0u, 0u};
            for (int k = 0; k < W; ++k) {
                const uint32_t x = pos + s_red[0][k];
                pos = x > Ncap ? Ncap : x;
                P = xf_then(P, Xf{s_red[1][k], s_red[2][k]});
            }
            seed_st = bcast_first(row->st_in);
            seed_pos = bcast_first(pos);
            seed_px = bcast_first(xf_apply(P, kPxInit));
        }
    } else {
    if (MODE >= 2 && chunk_first) seed_st = bcast_first(row->st_in);
    else if (MODE >= 2 && bcast_first((uint32_t)src[14]) != 0x31u) seed_st = ST_A;   // (QOI-compat: the states are A, D1..D4, X)
    if (MODE == 3 && chunk_first) { seed_pos = bcast_first(row->pos_in); seed_px = bcast_first(row->px_in); }
    }

    if (MODE >= 2 && threadIdx.x == 0) s_path = dp_flags[img];   // what pass 1 (or pass 2: SQOA_OP_REF) decided
    if ((MODE == 0 || (MODE == 1 && chunk_first == 0)) && threadIdx.x == 0) {
        // header (seqoia.h:672-688): written to the result exactly as sqoa_decode fills *desc
        const uint32_t magic = (uint32_t)src[0] << 24 | (uint32_t)src[1] << 16 | (uint32_t)src[2] << 8 | src[3];
        const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
        const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
        const uint32_t hc = src[12], hs = src[13], q = src[14] != 0x31;
        res->len = 0; res->status = SQOA_HIP_ST_OK;   // the row as every launch must find it: nobody else has touched it yet (sqoa_runtime.hip, plan_enqueue)
        res->desc.width = hw; res->desc.height = hh; res->desc.channels = (uint8_t)hc;
        res->desc.colorspace = (uint8_t)hs; res->desc.qoi_compat = (uint8_t)q;
        row_mark_standby(res, retry ? 1 : 0);   // (a job the retry takes: the plan keeps launching the kernels behind its first one)
        const bool ok = !(hw == 0 || hh == 0 || hc < 1 || hc > 6 || hs > 1 || !(magic == 0x716f6966u || magic == 0x53716f61u) ||
                          (magic == 0x716f6966u && !q) || hh >= 400000000u / hw);
        uint32_t path = 1;
        // a caller that states the shape it expects (non-zero job.desc.width / height) gets BADARG for any other header:
        // its output buffer is laid out for that shape
        const bool shape_ok = (job.desc.width == 0 || job.desc.width == hw) && (job.desc.height == 0 || job.desc.height == hh);
        if (!ok || !shape_ok) { res->status = SQOA_HIP_ST_BADARG; path = 2; }
        else {
            const uint32_t oc = job.desc.channels ? job.desc.channels : (hc < 3 ? 1u : 3u) + ((hc & 1u) ? 0u : 1u);
            if ((unsigned long long)hw * hh * oc > job.dst_cap) { res->status = SQOA_HIP_ST_OVERFLOW; path = 2; }
            else if (hc < 3 || q || oc < 3) {
                res->status = SQOA_HIP_ST_SLOWPATH;
                // (the chip-wide passes: a QOI-compat colour stream takes the speculative route of sqoa_decode_qoi.hip -- passes 1 and
                // 2 here give it its chunks' parser states and pixel positions; the status stays SLOWPATH until that route has
                // verified itself, so that the one-workgroup decoder picks the stream up if it has not)
                path = (MODE == 1 && q && hc >= 3 && oc >= 3) ? 3u : 2u;
            }
        }
        s_path = path;
        if (MODE == 1) dp_flags[img] = path;
    }
    if (MODE == 1 && chunk_first != 0 && threadIdx.x == 0) s_path = 1;  // (parsing bytes is harmless whatever the header says)
    // QOI-compat framing (seqoia.h:677): no start byte, no ALPHA suffix -- another transition table, the chunk area one byte earlier
    const bool qoi_stream = MODE != 0 && bcast_first((uint32_t)src[14]) != 0x31u;
    if (lane == 0) { s_parse[0][w] = 0; s_parse[1][w] = 0; s_pos[w] = 0; s_px[w] = 0; }
    for (uint32_t b0 = threadIdx.x; b0 < 256u; b0 += W * 64) {  // transition function of a live byte (see header comment)
        const uint32_t len = b0 >= 0xfeu ? b0 - 0xfau : ((b0 & 0xc0u) == 0x80u ? 2u : 1u);
        const uint32_t opn = len == 1u ? (uint32_t)ST_A : len;
        const uint32_t fA = (b0 - 0x60u) < 0x20u && !qoi_stream ? (uint32_t)ST_N : opn;
        s_ftab[b0] = fA | opn << 8 | 0x02000000u;
        // op decode of seqoia.h:740-775 as a table
        const bool is_abs = b0 >= 0xfeu, is_luma = (b0 & 0xc0u) == 0x80u, is_run = (b0 - 0xc0u) < 0x3eu, is_sfx = (b0 - 0x60u) < 0x20u;
        const uint32_t cnt = (is_abs || is_luma) ? 1u : (b0 == 0xfdu ? 512u : (b0 & 0x3fu) + 1u);
        const uint32_t vg = (b0 & 0x3fu) - 32u;
        u32x4 e;
        e.x = len | (len * 8u - 4u) << 3 | cnt << 9 | (b0 < 0x60u ? 1u << 20 : 0u) | (is_run ? cnt << 21 : 0u);
        e.y = is_abs ? (b0 == 0xffu ? 0xffffffffu : 0x00ffffffu) : 0u;
        e.z = is_luma ? (((vg - 8u) & 0xffu) | ((vg - 8u) & 0xffu) << 16) : 0u;
        e.w = is_luma ? ((vg & 0xffu) << 8 | 0x000f000fu) : 0u;
        s_optab[b0] = e;
        s_sfx[b0][0] = is_sfx ? ((((b0 & 31u) - 16u) & 0xffu) << 24 | 9u) : 0u;  // delta << 24 | 8 (shift) | 1 (length)
        s_sfx[b0][1] = is_sfx ? 0u : 0xffffffffu;
        if (SQOA_DEC_TAB4) { s_t4[0][b0] = s_sfx[b0][0]; s_t4[SQOA_DEC_TAB4 ? 1 : 0][b0] = is_run ? cnt : 0u; s_t4[SQOA_DEC_TAB4 ? 2 : 0][b0] = is_sfx ? 0u : 0xffffffffu; }
    }
    if (threadIdx.x == 0) { u32x4 z = {4u << 3, 0u, 0u, 0u}; s_optab[256] = z; }
    __syncthreads();
    // path 3 (QOI-compat, MODE 1 and 2 only): parsed and counted here, decoded by decode_qoi_kernel's speculative pass
    if (s_path != 1 && !((MODE == 1 || MODE == 2) && s_path == 3)) return;

    const uint32_t hw = (uint32_t)src[4] << 24 | (uint32_t)src[5] << 16 | (uint32_t)src[6] << 8 | src[7];
    const uint32_t hh = (uint32_t)src[8] << 24 | (uint32_t)src[9] << 16 | (uint32_t)src[10] << 8 | src[11];
    const uint32_t hc = src[12];
    const uint32_t N = bcast_first(hw * hh);
    const int oc = (int)bcast_first(job.desc.channels ? job.desc.channels : 3u + ((hc & 1u) ? 0u : 1u));
    // Streams packed back to back (sqoa_hip_pack) start at any byte address: all stream offsets below are relative to the
    // dword-aligned address at or in front of the stream; the `soff` bytes in front of it are loaded but never looked at.
    const bool dw_out = (job.dst & 3ull) == 0;  // 4-channel pixels can be stored as dwords
    const uint32_t soff = (uint32_t)(job.src & 3ull);
    gcptr8 srca = src - soff;
    const uint32_t size_a = size + soff;
    const uint32_t p0 = kHeader + (qoi_stream ? 0u : 1u) + soff;     // first chunk byte (SQOA: after the start byte, seqoia.h:705)
    const uint32_t chunks_len = size_a - 8;     // seqoia.h:721
    const uint32_t area = chunks_len > p0 ? chunks_len - p0 : 0;
    const uint32_t nchunks = area ? (area + DEC_CHUNK - 1) / DEC_CHUNK : 1;
    // the chunks this workgroup takes: all of them, or (MODE > 0) dp_per from chunk_first on
    const uint32_t chunk_end = MODE == 0 ? nchunks : (chunk_first + dp_per < nchunks ? chunk_first + dp_per : nchunks);
    if (MODE != 0 && chunk_first >= nchunks) {
        // (the plan lays the tables out for the QOI-compat chunk area, one byte longer than the SQOA one: a workgroup past the end)
        if (threadIdx.x == 0) {
            if (MODE == 1) { row->g_lo = ID_LO; row->g_hi = ID_HI; }
            if (MODE == 2) { row->ctot = 0; row->tv = 0; row->tabs = 0; }
        }
        return;
    }
    uint32_t bad = 0;

#ifdef SQOA_TRACE  // diagnostic builds: where wave 0 spends its time, written behind the image's pixels (dst_cap must allow it)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
    const unsigned long long tr_start = wall_clock64();   // workgroup start / end (100 MHz) and placement: tools/trace_decode.py
#define DPHASE(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - ph_t; ph_t = t_; } while (0)
#elif defined(SQOA_ASM_MARK)  // phase markers as comments in the assembly (tools/asm_cost.py by region)
#define DPHASE(k) asm volatile("; SQOA_PHASE " #k)
#else
#define DPHASE(k) do { } while (0)
#endif
    uint32_t* lds = s_bytes[w];
    uint32_t stg[DEC_NLD];
#ifdef SQOA_DEC_HL_VGPR   // A/B builds: 5 VGPR spills instead of 1 at the 64-register cliff, 3.40 -> 3.45 ms -- off
    HL hl;   // the SWAR constants of the packed-byte adds, in VGPRs (see add8 in sqoa_decode_parts.h)
    asm volatile("v_mov_b32 %0, 0x80808080" : "=v"(hl.H));
    asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(hl.L));
#define HLARG , hl
#else
#define HLARG
#endif
    // straight-line when the whole staging window lies inside the stream (all but the last chunks): three
    // loads in flight, no wait until the next iteration stages them
    auto load_chunk = [&](uint32_t c, uint32_t (&r)[DEC_NLD]) {
        const uint32_t gs = p0 + c * DEC_CHUNK;
        const uint32_t d0 = gs >> 2;
        if ((d0 + DEC_LDS_DW) * 4u <= size_a) {
            gcptr32 p = (gcptr32)srca + d0;
#pragma unroll
            for (int k = 0; k < DEC_NLD - 1; ++k) r[k] = p[lane + 64 * k];
            r[DEC_NLD - 1] = p[64 * (DEC_NLD - 1) + (lane & 7)];
        } else {
#pragma unroll
            for (int k = 0; k < DEC_NLD; ++k) {
                const uint32_t d = d0 + lane + 64 * k;
                r[k] = (k < DEC_NLD - 1 || lane < DEC_LDS_DW - 64 * (DEC_NLD - 1)) && d * 4u < size_a ? ((gcptr32)srca)[d] : 0u;
            }
        }
    };

    // The prefetched dwords go to the wave's LDS buffer as soon as the current chunk's bytes are dead -- behind the walk, in front
    // of the chunk's pixel stores -- and not at the top of the next chunk: the wait for the loads there also waited for every
    // store issued after them (vmcnt counts loads and stores in issue order), i.e. each chunk began by draining the stores of
    // the chunk before.
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < DEC_NLD - 1; ++k) lds[lane + 64 * k] = stg[k];
        if (lane < DEC_LDS_DW - 64 * (DEC_NLD - 1)) lds[lane + 64 * (DEC_NLD - 1)] = stg[DEC_NLD - 1];
    };

    // One chunk.  IN (compile time) = interior chunk: every byte lies inside the chunk area and a successor exists, so the
    // liveness tests and the end-of-stream handling drop out of the hot loop.  Returns false when the image is handed over.
    auto body = [&](auto in_tag, uint32_t c) -> uint32_t {   // 1: go on with the wave's next chunk, 0: the image is done with here
        constexpr bool IN = decltype(in_tag)::value;
        DPHASE(7);
        // ---- the chunk's bytes (+ look-ahead) are in LDS (staged at the end of the wave's previous chunk); prefetch the next one ----
        const bool more = c + W < chunk_end;
        if (more) load_chunk(c + W, stg);
        const uint32_t gs = p0 + c * DEC_CHUNK;               // stream offset of chunk byte 0
        const uint8_t* cb = reinterpret_cast<const uint8_t*>(lds) + (gs & 3u);
        [[maybe_unused]] const uint8_t* lb = cb + DEC_LB * lane;                // this lane's 8 bytes
        const bool last_chunk = IN ? false : c == nchunks - 1;
        // how many of my bytes lie inside the chunk area (op starts beyond it do not exist, seqoia.h:726)
        const uint32_t my0 = gs + DEC_LB * lane;
        const uint32_t nlive = IN ? (uint32_t)DEC_LB : my0 >= chunks_len ? 0u : (chunks_len - my0 > (uint32_t)DEC_LB ? (uint32_t)DEC_LB : chunks_len - my0);

        // ---- 1. parse: each lane composes the transition functions of its 8 bytes, one wave scan ----
        uint32_t qd[DEC_LB / 4];
#ifdef SQOA_DEC_UNALIGNED_PARSE_READ
        __builtin_memcpy(qd, lb, DEC_LB);   // one ds_read_b64 at a byte address that is 3 mod 4 (the chunk area starts behind 15 header bytes): stalls the LDS pipe
#else
        {   // three aligned dwords and two v_alignbyte_b32 (as the walk's windows)
            const uint32_t off = (gs & 3u) + DEC_LB * lane;
            const uint32_t* pw = lds + (off >> 2);
            uint32_t dw[DEC_LB / 4 + 1];
#pragma unroll
            for (int k = 0; k <= DEC_LB / 4; ++k) dw[k] = pw[k];
#pragma unroll
            for (int k = 0; k < DEC_LB / 4; ++k) qd[k] = __builtin_amdgcn_alignbyte(dw[k + 1], dw[k], off & 3u);
        }
#endif
        Fn F{ID_LO, ID_HI};
#pragma unroll
        for (int k = 0; k < DEC_LB; ++k) {
            const uint32_t b = (qd[k / 4] >> (8 * (k & 3))) & 0xffu;
            const uint32_t flo = (uint32_t)k < nlive ? s_ftab[b] : (ST_X | ST_X << 8 | 0x02000000u);
            F = compose(Fn{flo, 0x07060403u}, F);
        }
        const Fn Fi = wave_compose_scan(F);
        const Fn G{bcast_lane(Fi.lo, 63), bcast_lane(Fi.hi, 63)};
        const Fn Fx{shift_up1(Fi.lo, ID_LO), shift_up1(Fi.hi, ID_HI)};
        // exit state independent of the entry state?  (A, N, D1..D4; X cannot enter a chunk that has a successor)
        const uint32_t g0 = G.lo & 0xffu;
        const bool g_const = G.lo == g0 * 0x01010101u && (G.hi & 0xffffu) == g0 * 0x0101u;
        unsigned long long* my_parse = &s_parse[(c / W) & 1][w];
        DPHASE(0);
        uint32_t st_in = seed_st;
        if constexpr (MODE == 1) {
            if (lane == 0) { s_wg[0][c - chunk_first] = G.lo; s_wg[1][c - chunk_first] = G.hi; }
            if (more) stage();
            return 1u;
        } else {   // inside the workgroup the chunks are chained through LDS in every mode
            // An exit state that does not depend on the entry state is posted before the entry state is known.  That lets a
            // wave run ahead of its successor, and the two-deep slots are safe only because the pixel chain below holds it
            // back (a wave cannot finish chunk c + W before chunk c + 1 has been entered); pass 2 has no pixel chain.
            const bool early = g_const && MODE != 2;
            if (early && !last_chunk) post(my_parse, c + 1, g0, lane);
            if (c > chunk_first) {
                const uint32_t rc = chain_wait_rc<SPIN>(&s_parse[((c - 1) / W) & 1][(c - 1) % W], c, st_in);
                if (rc) { bad |= rc >> 1; return 0u; }
            }
            if (!early && !last_chunk) post(my_parse, c + 1, apply(G, st_in), lane);
        }
        DPHASE(1);

        // ---- 2. walk: every lane decodes its own ops in order (<= 8 of them) --------------------
        const uint32_t lst = apply_perm(Fx, st_in);  // parser state in front of my first byte
        if constexpr (MODE == 2) {
            if (qoi_stream) {
                // QOI-compat stream: only the chunk's pixel count is wanted here (seqoia.h:740-775 with qoi_compat: every op is one
                // pixel, a run byte 0xc0 | n -- 0xfd and 0xfc included -- is n + 1)
                // ... and a GUESS of the pixel entering every workgroup's range, for the speculative pass to start its warm-up from:
                // the ops' transforms composed as for an SQOA stream, an INDEX op -- whose pixel only the decoder state knows --
                // taken as "no change".  Exact wherever no INDEX op stands between the last RGB / RGBA op and the range (smooth
                // content is DIFF / LUMA ops for kilobytes on end: nothing in a warm-up would ever anchor the pixel); where it
                // is wrong, the check behind the speculative pass finds out (dq_verify_kernel).
                uint32_t qpos = lst >= 2u && lst <= 5u ? lst - 1u : (lst == ST_X ? (uint32_t)DEC_LB : 0u), qcl = 0;
                const uint32_t qaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (gs & 3u) + DEC_LB * lane;
                Xf QT{0u, 0u};
#pragma unroll
                for (int i = 0; i < DEC_LB; ++i) {
                    if (__ballot(qpos < nlive) == 0ull) break;
                    const uint32_t a = qaddr + (qpos < (uint32_t)DEC_LB ? qpos : (uint32_t)DEC_LB);
                    lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
                    const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
                    const uint32_t b = lo & 0xffu, b1 = (lo >> 8) & 0xffu;
                    const uint32_t len = b >= 0xfeu ? b - 0xfau : ((b & 0xc0u) == 0x80u ? 2u : 1u);
                    const uint32_t cnt = (b & 0xc0u) == 0xc0u && b < 0xfeu ? (b & 0x3fu) + 1u : 1u;
                    const uint32_t vg = (b & 0x3fu) - 32u;
                    Xf op{0u, 0u};
                    if (b == 0xfeu) op = Xf{(lo >> 8) & 0x00ffffffu, 0x00ffffffu};                      // seqoia.h:741-745
                    else if (b == 0xffu) op = Xf{lo >> 8 | hi << 24, 0xffffffffu};                      // seqoia.h:746-751
                    else if ((b & 0xc0u) == 0x40u)                                                      // seqoia.h:756-760
                        op.v = ((((b >> 4) & 3u) - 2u) & 0xffu) | ((((b >> 2) & 3u) - 2u) & 0xffu) << 8 | (((b & 3u) - 2u) & 0xffu) << 16;
                    else if ((b & 0xc0u) == 0x80u)                                                      // seqoia.h:761-769
                        op.v = ((vg - 8u + (b1 >> 4)) & 0xffu) | (vg & 0xffu) << 8 | ((vg - 8u + (b1 & 15u)) & 0xffu) << 16;
                    if (qpos < nlive) { qcl += cnt; qpos += len; QT = xf_then(QT, op); }
                }
                if (more) stage();
                const uint32_t qtot = bcast_lane(wave_incl_scan(qcl), 63);
                const Xf QTi = wave_xf_scan(QT);
                const uint32_t qv = bcast_lane(QTi.v, 63), qa = bcast_lane(QTi.abs, 63);
                if (lane == 0) { s_wg[0][c - chunk_first] = qtot > N ? N : qtot; s_wg[1][c - chunk_first] = qv; s_wg[2][c - chunk_first] = qa; }
                return 1u;
            }
        }
        // first op start: inside an op -> skip its remaining bytes; an ALPHA byte at an op boundary after an
        // op belongs to that op (whoever owns the op has read it), so it is skipped too
        uint32_t pos = first_op_offset<DEC_LB>(lst);
        bool sfx_ok = lst != ST_N;  // may the byte at `pos` be an ALPHA suffix of the op before it?
        uint32_t val[DEC_R];   // pixel after my i-th op, relative to the pixel entering my lane unless absolute
        uint32_t meta[DEC_R];  // pixels before the op inside my lane (CL_BITS) | cnt (11) << CL_BITS
        uint32_t amask[DEC_R]; // byte mask of the channels of val[] that are absolute
        Xf T{0u, 0u};
        uint32_t cl = 0, iters = DEC_R, longest = 0;
        uint32_t info_or = 0;  // OR of the ops' table words: bit 20 = a lane met SQOA_OP_REF
        // LDS byte address of my first stream byte; the op tables sit at fixed LDS addresses
        const uint32_t lane_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (gs & 3u) + DEC_LB * lane;
        const uint32_t optab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) u32x4*)s_optab;
        const uint32_t sfx_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_sfx[0][0];
        // 8 stream bytes at LDS byte address a: three aligned dwords and two v_alignbyte_b32 (an unaligned ds_read_b64 stalls the pipe)
        auto window = [&](uint32_t a, uint32_t& lo, uint32_t& hi) {
            lcptr32 pw = (lcptr32)(uintptr_t)(a & ~3u);
            const uint32_t d0 = pw[0], d1 = pw[1], d2 = pw[2];
            lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
            hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
        };
        static_for<0, DEC_R>([&](auto i_tag) -> bool {
            constexpr int i = decltype(i_tag)::value;
            // op byte, <= 4 data bytes, possible ALPHA suffix
            if (i == 0) {  // an ALPHA byte at my first op boundary belongs to the op in front of it: step over it
                uint32_t f0, f1;
                window(lane_addr + (pos < (uint32_t)DEC_LB ? pos : (uint32_t)DEC_LB), f0, f1);
                if (sfx_ok && ((f0 & 0xffu) - 0x60u) < 0x20u) pos += 1u;
            }
            if (__ballot(pos < nlive) == 0ull) {  // val / meta / amask beyond `iters` stay unwritten and unread
                iters = i;
                return false;
            }
#if SQOA_DEC_EXEC_WALK
            // Lanes that have run out of ops sit the iteration out under EXEC (a lane never comes back: its position only grows), not
            // as lanes that decode a "no op" table entry behind arithmetic masks.  The chip runs this kernel AT its power limit (1400 W
            // board power, the clock falls until it fits: profiles/r04_power.txt), a third of the walk's lane slots are such lanes (the
            // iteration count is the busiest lane's), and what an idle lane does not execute it does not pay for.
            asm volatile("" : "=v"(val[i]), "=v"(amask[i]));   // (read later only where meta says the slot holds an op)
            meta[i] = cl;                                      // no op in this slot of mine: zero pixels
            if (pos < nlive) {
            uint32_t lo, hi;
            window(lane_addr + pos, lo, hi);
            const u32x4 e = *(lcptr128)(uintptr_t)(optab_addr + (lo & 0xffu) * 16u);
            info_or |= e.x;
            const uint32_t len = e.x & 7u, setm = e.y;
            uint32_t cnt = (e.x >> 9) & 0x3ffu;
            // byte after the op (seqoia.h:777-783): an ALPHA suffix belongs to the op.  w1 = stream >> (8 * len - 4).
            const unsigned long long q = (unsigned long long)hi << 32 | lo;
            const uint32_t w1 = (uint32_t)(q >> ((e.x >> 3) & 63u));
            const uint32_t se = *(lcptr32)(uintptr_t)(sfx_addr + ((w1 >> 1) & 0x7f8u));
            const uint32_t has8 = se & 8u, da24 = se & 0xff000000u;
            const uint32_t adv = len + (se & 1u);
            // A RUN right behind an op repeats exactly the pixel this op produces: taken in the same iteration (see below)
            const uint32_t w2 = w1 >> has8;
            const uint32_t extra = *(lcptr32)(uintptr_t)(optab_addr + (w2 & 0xff0u)) >> 21;               // 0 unless RUN / BIGRUN
            const uint32_t m_nsfx = *(lcptr32)(uintptr_t)(sfx_addr + 4u + ((w2 >> 9) & 0x7f8u));          // ~0 unless an ALPHA byte follows
            const uint32_t m_fuse = m_nsfx & sra31(pos + adv - nlive) & sra31(cnt + extra - 1025u);
            const uint32_t extra_m = extra & m_fuse;
            cnt += extra_m;
            const uint32_t base = __builtin_amdgcn_alignbit(hi, lo, 8) + da24;                                 // RGB[A], seqoia.h:740-752
            const uint32_t nib = (((lo >> 12) & 0xfu) | ((lo << 8) & 0xf0000u)) & e.w;                         // LUMA, seqoia.h:761-769
            const uint32_t d8 = ((e.z + nib) & 0x00ff00ffu) | (e.w & 0xff00u) | da24;
            T = xf_then(T, Xf{(d8 & ~setm) | (base & setm), setm} HLARG);
            val[i] = T.v;
            meta[i] = cl | cnt << CL_BITS;
            amask[i] = T.abs;
            cl += cnt;
            longest = longest > cnt ? longest : cnt;
            pos += adv + ((extra_m + 1023u) >> 10);
            }
#else
            // The op decode of seqoia.h:740-783 as table look-ups and mask arithmetic; no compare / select pairs.
            const uint32_t m_act = sra31(pos - nlive);                     // ~0 while I still have an op (pos < nlive <= 8)
            uint32_t lo, hi;
            window(lane_addr + ((pos & m_act) | ((uint32_t)DEC_LB & ~m_act)), lo, hi);
            const uint32_t idx = (((lo & 0xffu) - 256u) & m_act) + 256u;   // lanes out of ops decode the "no op" entry
            const u32x4 e = *(lcptr128)(uintptr_t)(optab_addr + idx * 16u);
            info_or |= e.x;
            const uint32_t len = e.x & 7u, setm = e.y;
            uint32_t cnt = (e.x >> 9) & 0x3ffu;
            // byte after the op (seqoia.h:777-783): an ALPHA suffix belongs to the op.  w1 = stream >> (8 * len - 4).
            const unsigned long long q = (unsigned long long)hi << 32 | lo;
            const uint32_t w1 = (uint32_t)(q >> ((e.x >> 3) & 63u));
#if SQOA_DEC_TAB4
            const uint32_t t4_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_t4[0][0];
            const uint32_t se = *(lcptr32)(uintptr_t)(t4_addr + ((w1 >> 2) & 0x3fcu & m_act));
#else
            const uint32_t se = *(lcptr32)(uintptr_t)(sfx_addr + ((w1 >> 1) & 0x7f8u & m_act));
#endif
            const uint32_t has8 = se & 8u, da24 = se & 0xff000000u;
            const uint32_t adv = len + (se & 1u);
            // A RUN right behind an op repeats exactly the pixel this op produces (pixel pairs, short repeats, split runs):
            // take it in the same iteration when it is mine (starts inside my bytes) and carries no ALPHA suffix of its own.
            // (a run behind a run also repeats the same pixel -- 0xfd 0xfd, 1024 pixels, included: flat areas are rows of them)
            const uint32_t w2 = w1 >> has8;
#if SQOA_DEC_TAB4
            const uint32_t extra = *(lcptr32)(uintptr_t)(t4_addr + 1024u + ((w2 >> 2) & 0x3fcu));        // 0 unless RUN / BIGRUN
            const uint32_t m_nsfx = *(lcptr32)(uintptr_t)(t4_addr + 2048u + ((w2 >> 10) & 0x3fcu));      // ~0 unless an ALPHA byte follows
#else
            const uint32_t extra = *(lcptr32)(uintptr_t)(optab_addr + (w2 & 0xff0u)) >> 21;               // 0 unless RUN / BIGRUN
            const uint32_t m_nsfx = *(lcptr32)(uintptr_t)(sfx_addr + 4u + ((w2 >> 9) & 0x7f8u));          // ~0 unless an ALPHA byte follows
#endif
            const uint32_t m_fuse = m_nsfx & sra31(pos + adv - nlive) & sra31(cnt + extra - 1025u);
            const uint32_t extra_m = extra & m_fuse;
            cnt += extra_m;
            // the pixel transform of the op: absolute channels from the stream, the others as deltas
            const uint32_t base = __builtin_amdgcn_alignbit(hi, lo, 8) + da24;                                 // RGB[A], seqoia.h:740-752
            const uint32_t nib = (((lo >> 12) & 0xfu) | ((lo << 8) & 0xf0000u)) & e.w;                         // LUMA, seqoia.h:761-769
            const uint32_t d8 = ((e.z + nib) & 0x00ff00ffu) | (e.w & 0xff00u) | da24;
            T = xf_then(T, Xf{(d8 & ~setm) | (base & setm), setm} HLARG);
            val[i] = T.v;
            meta[i] = cl | cnt << CL_BITS;
            amask[i] = T.abs;
            cl += cnt;
            longest = longest > cnt ? longest : cnt;
            pos += adv + ((extra_m + 1023u) >> 10);
#endif
            sfx_ok = false;  // the suffix after an op is consumed with the op; the next byte is an op start
            return true;
        });
        iters = bcast_first(iters);  // wave-uniform by construction; said so, the slot tests below are scalar compares instead of lane masks
        // SQOA_OP_REF (what follows it in a lane is discarded with the image); more ops in a lane's bytes than it has record registers for
        const bool ref_lane = ((info_or >> 20) & 1u) || (DEC_LB > DEC_R && iters == (uint32_t)DEC_R && pos < nlive);
        if (__ballot(ref_lane) != 0ull) {  // serial path for the whole image
            if (lane == 0) {
                atomicMax(&res->status, SQOA_HIP_ST_SLOWPATH);
                if (MODE) atomicMax(&dp_flags[img], 2u);   // pass 3 leaves the image to the exact decoder
            }
            return 0u;
        }

        if (more) stage();   // the walk was the last reader of this chunk's bytes
        DPHASE(2);
        // ---- 3. wave scans: pixel positions and pixel transforms of the lanes below ------------
        const uint32_t cinc = wave_incl_scan(cl);
        const uint32_t ctot = bcast_lane(cinc, 63);
        const Xf Ti = wave_xf_scan(T HLARG);
        const Xf Tg{bcast_lane(Ti.v, 63), bcast_lane(Ti.abs, 63)};
        const Xf Tx{shift_up1(Ti.v, 0u), shift_up1(Ti.abs, 0u)};

        DPHASE(3);
        // ---- 4. entry position / pixel from the chain, publish exits, store ------------------
        if constexpr (MODE == 2) {
            if (lane == 0) { s_wg[0][c - chunk_first] = ctot > N ? N : ctot; s_wg[1][c - chunk_first] = Tg.v; s_wg[2][c - chunk_first] = Tg.abs; }
            return 1u;
        }
        uint32_t pos_in = seed_pos, px_in = seed_px;
        if (c > chunk_first) {
            // one wait for both: the position word is written before the pixel word by the same lane (LDS executes a wave's
            // instructions in order), so it is there when the pixel word's tag is
            const uint32_t rc = chain_wait_rc<SPIN>(&s_px[(c - 1) % W], c, px_in);
            if (rc) { bad |= rc >> 1; return 0u; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            pos_in = bcast_first((uint32_t)__hip_atomic_load(&s_pos[(c - 1) % W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
        DPHASE(4);
        const uint32_t exit_v = xf_apply(Tg, px_in);
        const uint32_t pos_out = pos_in + ctot > N ? N : pos_in + ctot;
        if (!last_chunk) {   // (the position word first: a waiter that sees the pixel word's tag finds it there)
            lds_post(&s_pos[w], c + 1, pos_out, lane);
            lds_post(&s_px[w], c + 1, exit_v, lane);
        }

        if (pos_in < N) {
            const uint32_t lane_px = xf_apply(Tx, px_in HLARG);       // pixel entering my lane
            const uint32_t lane_p = pos_in + cinc - cl;         // pixel index of my first op
            const bool plain = oc == 4 && dw_out && pos_in + ctot <= N && __ballot(longest > 2u) == 0ull;
            if (plain) {  // the common chunk: literal ops only, nothing clipped, 4-byte pixels: one store per op slot
#if defined(SQOA_DEC_C_STORES) || defined(SQOA_DBG_DEC_NOSTORE)
                static_for<0, DEC_R>([&](auto i_tag) -> bool {
                    constexpr int i = decltype(i_tag)::value;
                    if ((uint32_t)i >= iters) return false;
                    const uint32_t m = meta[i], am = amask[i];
                    const uint32_t v = add8(lane_px & ~am, val[i]);
                    gptr32 o = (gptr32)out + lane_p + (m & CL_MASK);
#ifdef SQOA_DBG_DEC_NOSTORE  // timing experiments only
                    asm volatile("" :: "v"(o), "v"(v), "v"(m));
#else
                    if (m >> CL_BITS) o[0] = v;
                    if (m >> (CL_BITS + 1)) o[1] = v;  // cnt == 2: a literal op and the one-pixel run behind it
#endif
                    return true;
                });
#else
                // Hand-written: the compiler's version of `if (cnt >= 1) o[0] = v; if (cnt == 2) o[1] = v;` is, per slot, two
                // s_and_saveexec / s_cbranch_execz / s_or exec triples around stores with 64-bit address arithmetic (7 scalar
                // instructions + 3 branches + 10 VALU); here the two predicates narrow EXEC directly (v_cmpx), the stores take the
                // image base from an SGPR pair and a 32-bit byte offset (an image is < 1.6 GB), and EXEC is put back once -- to what
                // it was on entry, read here once per chunk (all ones today: wave-uniform control flow, full workgroups; the block
                // does not rely on it).
                const uint32_t lane_off = lane_p << 2;
                const unsigned long long obase = job.dst;
                const unsigned long long exec_in = exec_saved();
                static_for<0, DEC_R>([&](auto i_tag) -> bool {
                    constexpr int i = decltype(i_tag)::value;
                    if ((uint32_t)i >= iters) return false;
                    const uint32_t m = meta[i], am = amask[i];
                    const uint32_t v = add8(lane_px & ~am, val[i] HLARG);
                    const uint32_t off = lane_off + ((m & CL_MASK) << 2);
                    asm volatile(
                        "v_cmpx_lt_u32_e32 vcc, %[c1], %[m]\n\t"           // cnt >= 1
                        "global_store_dword %[off], %[v], %[base]\n\t"
                        "v_cmpx_lt_u32_e32 vcc, %[c2], %[m]\n\t"           // cnt == 2: a literal op and the one-pixel run behind it
                        "global_store_dword %[off], %[v], %[base] offset:4\n\t"
                        "s_mov_b64 exec, %[ex]"
                        :: [m] "v"(m), [off] "v"(off), [v] "v"(v), [base] "s"(obase), [ex] "s"(exec_in), [c1] "n"(CL_MASK), [c2] "n"(2u * CL_MASK + 1u) : "memory", "vcc");
                    return true;
                });
                asm volatile("s_nop 4");   // (a VALU write of EXEC, v_cmpx, wants 5 wait states in front of a DPP instruction)
#endif
            } else if (SWEEP && ((oc == 4 && dw_out) || oc == 3) && ctot >= SWEEP_MIN_PX && (pos_in + ctot > N ? N - pos_in : ctot) <= SWEEP_PX) {
                // A chunk with RUNs in it, position-major.  One run at a time by the whole wave -- the path below -- is one store
                // instruction per run (and one per literal op slot), each a few dozen dwords at whatever alignment: on icon-like
                // streams 250 store instructions per chunk of 2000 pixels, and the stores, not the instructions around them, were
                // two thirds of the kernel (without them: 0.58 of 1.78 ms; a run loop of a third of the instructions: no change).
                // Here every op leaves its pixel in LDS by its rank in stream order and one bit at the pixel it starts at; then the
                // wave sweeps the chunk's pixels 64 at a time -- the op that covers a pixel is the number of starts at or below it:
                // a popcount of the bit row, running base in an SGPR -- and every store instruction writes 64 consecutive pixels.
                const uint32_t total = pos_in + ctot > N ? N - pos_in : ctot;
                uint32_t* swv = s_swv[SWEEP ? w : 0];
                uint2* swb = s_swb[SWEEP ? w : 0];
                // (the lane number, recomputed here and opaque to the compiler: with the kernel's `lane` everything below that depends
                // on the lane alone -- mask words, LDS addresses -- is a loop invariant of the chunk loop, hoisted in front of it, and
                // pays for its registers with spills of the walk's op records in the hot path: headline batch 3.12 -> 3.44 ms)
                uint32_t lane;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
                swb[lane] = uint2{0u, 0u}; swb[lane + 64] = uint2{0u, 0u};             // (4096 bits)
                uint32_t nops = 0;
                static_for<0, DEC_R>([&](auto i_tag) -> bool { constexpr int i = decltype(i_tag)::value; if ((uint32_t)i >= iters) return false; nops += sra31(0u - (meta[i] >> CL_BITS)) & 1u; return true; });
                const uint32_t rank0 = wave_incl_scan(nops) - nops;                    // ops in the lanes below mine
                const uint32_t lane_rel = cinc - cl;                                   // the chunk's pixel my first op starts at
                static_for<0, DEC_R>([&](auto i_tag) -> bool {
                    constexpr int i = decltype(i_tag)::value;
                    if ((uint32_t)i >= iters) return false;
                    const uint32_t m = meta[i], am = amask[i];
                    const uint32_t v = add8(lane_px & ~am, val[i]);
                    const uint32_t start = lane_rel + (m & CL_MASK);
                    // (an op slot without an op, or an op that starts behind the image's end: its pixel to a dump word, no bit)
                    const uint32_t on = sra31(0u - (m >> CL_BITS)) & sra31(start - total);
                    const uint32_t idx = ((rank0 + (uint32_t)i) & on) | (((uint32_t)(DEC_R * 64) + (uint32_t)lane) & ~on);
                    swv[idx] = v;
                    __hip_atomic_fetch_or(&swb[(start >> 5) & (SWEEP_PX / 32 - 1) & on].x, (1u << (start & 31u)) & on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    return true;
                });
                // the starts in front of every 32-bit word of the row (an exclusive prefix sum of the words' popcounts: the second half
                // of the row only when the chunk reaches it)
                {
                    const uint32_t c0 = (uint32_t)__builtin_popcount(swb[lane].x);
                    const uint32_t s0 = wave_incl_scan(c0);
                    swb[lane].y = s0 - c0;
                    if (total > SWEEP_PX / 2) {
                        const uint32_t c1 = (uint32_t)__builtin_popcount(swb[lane + 64].x);
                        swb[lane + 64].y = bcast_lane(s0, 63) + wave_incl_scan(c1) - c1;
                    }
                }
                // The sweep: 256 pixels per step, FOUR consecutive ones per lane -- their ops are the number of starts at or below each:
                // the prefix of the row word + a popcount inside it, then one bit test per pixel -- and one 16-byte (three channels:
                // 12-byte) store per lane.
                for (uint32_t x0 = 0; x0 < total; x0 += 256u) {
                    const uint32_t x = x0 + 4u * lane, sh = x & 31u;
                    const uint2 bw = swb[x >> 5];
                    const uint32_t r0 = bw.y + (uint32_t)__builtin_popcount(bw.x & ((2u << sh) - 1u));   // >= 1 inside the chunk: its first op starts at its pixel 0
                    const uint32_t b = bw.x >> sh;
                    const uint32_t r1 = r0 + ((b >> 1) & 1u), r2 = r1 + ((b >> 2) & 1u), r3 = r2 + ((b >> 3) & 1u);
                    const uint32_t v0 = swv[r0 - 1u], v1 = swv[r1 - 1u], v2 = swv[r2 - 1u], v3 = swv[r3 - 1u];
                    if (oc == 4) {
                        typedef u32x4 u32x4_una __attribute__((aligned(4)));
                        typedef __attribute__((address_space(1))) u32x4_una* gptr128u;
                        gptr32 o = (gptr32)out + pos_in + x;
                        if (x + 4u <= total) *(gptr128u)o = u32x4{v0, v1, v2, v3};
                        else if (x < total) {                                              // the chunk's last one to three pixels
                            o[0] = v0;
                            if (x + 1u < total) o[1] = v1;
                            if (x + 2u < total) o[2] = v2;
                        }
                    } else {
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                        typedef u32x3 u32x3_una __attribute__((aligned(1)));
                        typedef __attribute__((address_space(1))) u32x3_una* gptr96u;
                        gptr8 o = out + ((size_t)pos_in + x) * 3u;
                        if (x + 4u <= total)   // r g b r | g b r g | b r g b
                            *(gptr96u)o = u32x3{(v0 & 0x00ffffffu) | v1 << 24, ((v1 >> 8) & 0xffffu) | v2 << 16, ((v2 >> 16) & 0xffu) | v3 << 8};
                        else if (x < total) {
                            put_pixel<3>(o, 0, v0);
                            if (x + 1u < total) put_pixel<3>(o, 1, v1);
                            if (x + 2u < total) put_pixel<3>(o, 2, v2);
                        }
                    }
                }
            } else {
            auto store_ops = [&](auto oc_tag) {
                constexpr int OC = decltype(oc_tag)::value;
                static_for<0, DEC_R>([&](auto i_tag) -> bool {
                    constexpr int i = decltype(i_tag)::value;
                    if ((uint32_t)i >= iters) return false;
                    const uint32_t m = meta[i];
                    const uint32_t cnt = m >> CL_BITS;
                    const uint32_t P = lane_p + (m & CL_MASK);
                    const uint32_t am = amask[i];
                    const uint32_t v = add8(lane_px & ~am, val[i]);
                    if (cnt >= 1u && cnt <= 2u && P < N) put_pixel<OC>(out, P, v);
                    if (cnt == 2u && P + 1u < N) put_pixel<OC>(out, P + 1u, v);
                    unsigned long long runs = __ballot(cnt > 2u && P < N);
                    while (runs) {  // expand runs with the whole wave (seqoia.h:723-725)
                        const int l = __builtin_ctzll(runs);
                        runs &= runs - 1ull;
                        const uint32_t rp = bcast_lane(P, l), rv = bcast_lane(v, l);
                        uint32_t rc = bcast_lane(cnt, l);
                        if (rp + rc > N) rc = N - rp;
                        fill_pixels<OC>(out, rp, rc, rv, lane);
                    }
                    return true;
                });
            };
            if (oc == 4 && dw_out) store_ops(std::integral_constant<int, 4>{});
            else if (oc == 4) store_ops(std::integral_constant<int, 40>{});
            else store_ops(std::integral_constant<int, 3>{});
            }
        }
        DPHASE(5);
        if (last_chunk) {  // stream exhausted: the last pixel repeats (seqoia.h:726)
            if (pos_out < N) {
                if (oc == 4 && dw_out) fill_pixels<4>(out, pos_out, N - pos_out, exit_v, lane);
                else if (oc == 4) fill_pixels<40>(out, pos_out, N - pos_out, exit_v, lane);
                else fill_pixels<3>(out, pos_out, N - pos_out, exit_v, lane);
            }
            if (lane == 0) res->len = (int)(N * (uint32_t)oc);
        }
        return 1u;
    };

    uint32_t c = chunk_first + w;
    if (c < chunk_end) { load_chunk(c, stg); stage(); }
    const uint32_t nint_all = area / DEC_CHUNK < nchunks - 1 ? area / DEC_CHUNK : nchunks - 1;  // fully live chunks that have a successor
    const uint32_t nint = nint_all < chunk_end ? nint_all : chunk_end;
    uint32_t go = 1;
    if constexpr (MODE == 0) {
#if !defined(SQOA_DEC_PRIO) || SQOA_DEC_PRIO
    // Progress-based wave priority: the SIMD arbiter is oldest-first, so of the workgroups sharing a CU the oldest runs
    // almost at solo speed and the youngest finishes alone on a nearly empty CU.  Lowering a stream's priority as it
    // advances keeps co-resident streams within a quarter of each other.
    const uint32_t thr1 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T1 >> 6), thr2 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T2 >> 6),
                   thr3 = (uint32_t)((unsigned long long)nint * SQOA_PRIO_T3 >> 6);
    uint32_t thr = thr1, q = 0;
    __builtin_amdgcn_s_setprio(3);
    for (; go && c < nint; c += W) {
        if (c >= thr) {
            ++q; thr = q == 1 ? thr2 : q == 2 ? thr3 : ~0u;
            if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        go = body(std::true_type{}, c);
    }
#else
    for (; go && c < nint; c += W) go = body(std::true_type{}, c);
#endif
    } else {
        for (; go && c < nint; c += W) go = body(std::true_type{}, c);
    }
    for (; go && c < chunk_end; c += W) go = body(std::false_type{}, c);
    if (!go && lane == 0) {   // handed over (SQOA_OP_REF), or the predecessor was: tell whoever waits on this wave (chain_wait_rc)
        const unsigned long long gone = (unsigned long long)DEC_ABORT_TAG << 32;
        __hip_atomic_store(&s_parse[0][w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&s_parse[1][w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&s_px[w], gone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (MODE == 1 || MODE == 2) {   // the workgroup's chunks composed in stream order -> its row of the table
        __syncthreads();
        if (w == 0) {
            const uint32_t nk = chunk_end - chunk_first;
            const bool live = (uint32_t)lane < nk;
            if constexpr (MODE == 1) {
                const Fn Fi = wave_compose_scan(Fn{live ? s_wg[0][lane] : ID_LO, live ? s_wg[1][lane] : ID_HI});
                if (lane == 63) { row->g_lo = Fi.lo; row->g_hi = Fi.hi; }
            } else {
                uint32_t v = live ? s_wg[0][lane] : 0u;   // (a 512-byte chunk yields at most 512 x 512 = 2^18 pixels -- its counts are not clamped to N before this -- and a workgroup has at most 64 chunks: 2^24, no 32-bit overflow)
                v = wave_incl_scan(v);
                const Xf Ti = wave_xf_scan(Xf{live ? s_wg[1][lane] : 0u, live ? s_wg[2][lane] : 0u});
                if (lane == 63) { row->ctot = v > N ? N : v; row->tv = Ti.v; row->tabs = Ti.abs; }
            }
        }
    }
    if (bad && lane == 0) atomicMax(&res->status, SQOA_HIP_ST_INTERNAL);
#ifdef SQOA_TRACE
    __syncthreads();   // (every wave that has not left early: the workgroup's end)
    if (threadIdx.x == 0) {
        unsigned long long* tr = (unsigned long long*)((job.dst + job.dst_cap + 7) & ~7ull);
        for (int k = 0; k < 8; ++k) tr[k] = ph[k];
        tr[8] = tr_start; tr[9] = wall_clock64();
        tr[10] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | (unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32;
    }
#endif
}

}  // namespace sqoa

// `shape` bit 0: the plan's streams are short and many (sqoa_runtime.hip): four waves per stream instead of eight, twice as many
// streams at once (16 384 x 128 x 128: 0.83 -> 0.73 ms, 8192 x 256 x 256: 1.83 -> 1.73 ms; 512 x 512 and up: no difference).
// `shape` bit 2: the kernel with the position-major store path for chunks full of RUNs (decode_wave_kernel, SWEEP): icon-like streams,
// which the plan recognises by their size -- under one stream byte per pixel -- where the jobs state the image shape.  A kernel of
// its own because the path's registers cost the literal-op path spills: 4096 icons of 512 x 512 1.78 -> 1.50 ms with it, the
// headline batch 3.25 -> 3.47 ms.
// `shape` bit 1: TWELVE stream bytes per lane instead of eight (768-byte chunks).  What a chunk costs whatever its size -- the two
// chain hops, the three wave scans, the scalar control around them: 134 of 480 VALU and most of the scalar instructions -- is
// spread over 1.5 x the bytes; the walk keeps its eight record slots per lane (DEC_R).  Headline batch 3.41 -> 3.09 ms, icons
// 4096 x 512 x 512 1.94 -> 1.77 ms, C5 rank shard 12.7 -> 12.0 ms.  The plan asks for it when the jobs that state four output channels
// carry at least 3/4 of the launch's stream bytes and the streams average 12 KB or more (plan_fill, sqoa_runtime.hip): to three (byte stores, three per pixel) the stores are what the waves queue for, and twelve bytes per lane spread a
// store instruction's 64 pixels over 1.5 x the cache lines -- photo 1024 x 1 Mpixel RGB 2.90 -> 3.16 ms (11 % FEWER instructions,
// 24 % more issue-stall cycles: profiles/r04_a_photo_rgb_*).
// `shape` bit 3: the BALANCED kernel (sqoa_decode_bal.hip: a lane owns consecutive ops, not consecutive bytes; 1024-byte chunks) -- the
// plan's choice where it would pick the wide shape for large streams.
extern "C" hipError_t sqoa_launch_decode_bal(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int rgb, sqoa_hip_result* d_results, hipStream_t stream);
extern "C" hipError_t sqoa_launch_decode_wave(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int shape,
                                              sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (shape & 8) return sqoa_launch_decode_bal(d_jobs, d_ids, n, (shape & 16) ? 1 : 0, d_results, stream);   // (bit 4: the list holds jobs to three channels)
#ifndef SQOA_DEC_W
#define SQOA_DEC_W 8
#endif
#ifndef SQOA_DEC_W_SMALL
#define SQOA_DEC_W_SMALL 4
#endif
#ifndef SQOA_DEC_LB0   // stream bytes per lane of the one-workgroup-per-stream kernel when the plan asks for the wide shape (the passes keep SQOA_DEC_LB: the plan's tables)
#define SQOA_DEC_LB0 12
#endif
    constexpr int W = SQOA_DEC_W, WS = SQOA_DEC_W_SMALL, LB0 = SQOA_DEC_LB0, LB = SQOA_DEC_LB;
    const bool small = shape & 1, wide = (shape & 2) && LB0 != LB, sweep = (shape & 4) && SQOA_DEC_SWEEP;
#define SQOA_DEC_LAUNCH(W_, LB_, SW_) hipLaunchKernelGGL((sqoa::decode_wave_kernel<W_, 0, LB_, SW_>), dim3(n), dim3(W_ * 64), 0, stream, d_jobs, d_ids, d_results, nullptr, nullptr, nullptr, nullptr, 0u, 0u)
    if (sweep) {   // (the position-major store path is there for run-heavy streams decoded to three or four channels: always with the wide shape)
        if (small) SQOA_DEC_LAUNCH(WS, LB0, true); else SQOA_DEC_LAUNCH(W, LB0, true);
    }
    else if (small && wide) SQOA_DEC_LAUNCH(WS, LB0, false);
    else if (small) SQOA_DEC_LAUNCH(WS, LB, false);
    else if (wide) SQOA_DEC_LAUNCH(W, LB0, false);
    else SQOA_DEC_LAUNCH(W, LB, false);
#undef SQOA_DEC_LAUNCH
    return hipGetLastError();
}

// The retry behind the balanced kernel: decode_wave_kernel, eight stream bytes per lane, over the same list -- a workgroup whose stream is not
// marked SQOA_HIP_ST_SLOWPATH leaves at once (see the kernel).
extern "C" hipError_t sqoa_launch_decode_retry(const sqoa_hip_job* d_jobs, const int* d_ids, int n, int small, sqoa_hip_result* d_results, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    constexpr int W = SQOA_DEC_W, WS = SQOA_DEC_W_SMALL, LB = SQOA_DEC_LB;
    if (small) hipLaunchKernelGGL((sqoa::decode_wave_kernel<WS, 0, LB, false>), dim3(n), dim3(WS * 64), 0, stream, d_jobs, d_ids, d_results, nullptr, nullptr, nullptr, nullptr, 1u, 0u);
    else hipLaunchKernelGGL((sqoa::decode_wave_kernel<W, 0, LB, false>), dim3(n), dim3(W * 64), 0, stream, d_jobs, d_ids, d_results, nullptr, nullptr, nullptr, nullptr, 1u, 0u);
    return hipGetLastError();
}

// Few large streams: three passes over all their chunks with the whole chip (see decode_wave_kernel), two one-wave-per-image scans
// in between.  `d_images` / `d_wgs` / `d_tab` / `d_flags`: the tables the plan built (sqoa_runtime.hip).
extern "C" hipError_t sqoa_launch_decode_passes(const sqoa_hip_job* d_jobs, sqoa_hip_result* d_results, const void* d_images, int n_images,
                                                const void* d_wgs, int n_wgs, void* d_tab, void* d_flags, unsigned per, int scans, int lb, hipStream_t stream) {
    if (n_images <= 0 || n_wgs <= 0) return hipSuccess;
    if (per < sqoa::DP_CHUNKS_PER_WG_MIN || per > sqoa::DP_CHUNKS_PER_WG_MAX || per % 8u) return hipErrorInvalidValue;   // (8 waves take the chunks round-robin)
    using namespace sqoa;
    constexpr int W = 8;
    const dp_image* im = (const dp_image*)d_images;
    const dp_wg* wg = (const dp_wg*)d_wgs;
    dp_chunk* tab = (dp_chunk*)d_tab;
    uint32_t* fl = (uint32_t*)d_flags;
    // (no initialisation of the flags: pass 1's first workgroup of every image writes its flag before anything reads it)
    // `scans` = 0: no scan kernels, the workgroups of passes 2 and 3 compose the rows in front of them by themselves (see the kernel)
    const uint32_t fused = scans ? 0u : 1u;
    // `lb`: stream bytes per lane, i.e. the chunk size the plan counted the tables' rows in (sqoa_dp_chunks): 12 -- the per-chunk costs
    // spread over 1.5 x the bytes, as in the batch shape -- for plans without a QOI-compat stream among these (the speculative pass of
    // sqoa_decode_qoi.hip reads the rows with 512-byte chunk numbers), else 8
    auto run = [&](auto lb_tag) {
        constexpr int LB = decltype(lb_tag)::value;
        hipLaunchKernelGGL((decode_wave_kernel<W, 1, LB>), dim3(n_wgs), dim3(W * 64), 0, stream, d_jobs, nullptr, d_results, im, wg, tab, fl, per, 0u);
        if (scans) hipLaunchKernelGGL(dp_scan_state, dim3(n_images), dim3(DP_SCAN_THREADS), 0, stream, im, tab, fl);
        hipLaunchKernelGGL((decode_wave_kernel<W, 2, LB>), dim3(n_wgs), dim3(W * 64), 0, stream, d_jobs, nullptr, d_results, im, wg, tab, fl, per, fused);
        if (scans) hipLaunchKernelGGL(dp_scan_pixels, dim3(n_images), dim3(DP_SCAN_THREADS), 0, stream, d_jobs, im, tab, fl);
        hipLaunchKernelGGL((decode_wave_kernel<W, 3, LB>), dim3(n_wgs), dim3(W * 64), 0, stream, d_jobs, nullptr, d_results, im, wg, tab, fl, per, fused);
    };
    if (lb == 12) run(std::integral_constant<int, 12>{});
    else run(std::integral_constant<int, SQOA_DEC_LB>{});
    return hipGetLastError();
}

// Does any of the multi-pass streams use the QOI-compat framing (seqoia.h:677)?  Asked once when the plan is filled, so that plans
// without one do not launch the speculative pass and its two checks for nothing (three launches that leave on one load each are
// still 8 us of a lone stream's 0.10 ms).  Only a hint: a stream the plan does not launch them for stays SLOWPATH and is decoded
// by the one-workgroup kernel behind.
namespace sqoa {
__global__ void dp_classify_kernel(const sqoa_hip_job* __restrict__ jobs, const int* __restrict__ ids, int n, uint32_t* __restrict__ word) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const sqoa_hip_job job = jobs[ids[i]];
    if (job.src_size > (uint32_t)kHeader && ((gcptr8)job.src)[kHeader] != 0x31u) atomicOr(word, 1u);
}
}  // namespace sqoa
extern "C" hipError_t sqoa_launch_dp_classify(const sqoa_hip_job* d_jobs, const int* d_ids, int n_images, void* d_word, hipStream_t stream) {
    if (n_images <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_word, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sqoa::dp_classify_kernel, dim3((n_images + 255) / 256), dim3(256), 0, stream, d_jobs, d_ids, n_images, (uint32_t*)d_word);
    return hipGetLastError();
}

// What the plan needs to lay the pass tables out (sqoa_runtime.hip): the chunks of a stream of `src_size` bytes (the chunk area is
// what lies between header + start byte and the 8 end bytes, as in decode_wave_kernel), and -- for `total_chunks` over all the
// multi-pass streams of a plan -- how many chunks a workgroup of a pass takes: few (many workgroups) while the chip is not
// full, more when it is (inside a workgroup the chunks are chained through LDS, and a short chain is mostly ramp-up: 1024
// images at 16 chunks per workgroup 9.2 ms).  One table row per workgroup.
extern "C" uint32_t sqoa_dp_chunks(uint32_t src_size, int lb) {
    const uint32_t area = src_size > 22u ? src_size - 22u : 0u;   // (the QOI-compat chunk area: one byte longer than the SQOA one -- no start byte)
    const uint32_t chunk = 64u * (uint32_t)(lb == 12 ? 12 : SQOA_DEC_LB);
    return area ? (area + chunk - 1) / chunk : 1u;
}
extern "C" uint32_t sqoa_dp_per_wg(unsigned long long total_chunks, unsigned wgs, size_t* row_bytes) {
    if (row_bytes) *row_bytes = sizeof(sqoa::dp_chunk);
    unsigned long long per = (total_chunks / (wgs ? wgs : 4096u) + 7ull) & ~7ull;   // about `wgs` workgroups a pass (the plan says how many: sqoa_runtime.hip)
    if (per < sqoa::DP_CHUNKS_PER_WG_MIN) per = sqoa::DP_CHUNKS_PER_WG_MIN;
    if (per > sqoa::DP_CHUNKS_PER_WG_MAX) per = sqoa::DP_CHUNKS_PER_WG_MAX;
    return (uint32_t)per;
}
