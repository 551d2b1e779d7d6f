// conv_pf3_kernel.h -- persistent, explicitly ping-ponged successor of conv_pf_kernel for the large 3x3 / stride-1
// layers (PF operands: two fp16 planes with a zero halo, see conv_pf_kernel.h).
//
// What the round-3 measurements of conv_pf_kernel<2,2,1,4> said (tools/ubench/pf_lab.hip, 64->64 @256^2, batch 32):
//   * the matrix time (0.186 ms) ADDS to the time of everything else (0.18 ms of barrier / ds_read / LDS-DMA issue
//     skeleton, 0.20 ms of prologue + epilogue): the two co-resident workgroups of a CU fall into step, both in
//     their matrix segment, then both in their load segment;
//   * one LDS-DMA instruction costs the issuing wave 100-130 cycles inside the loop, any other vector-memory
//     instruction 40-70 cycles whatever its width (the old epilogue issues 128 four-byte accesses per wave);
//   * prologue (first DMA round trip, 7-8 k cycles) and epilogue (LayerNorm + residual + stores) are exposed once
//     per 256-pixel tile;
//   * the matrix pipe and the VALU of a SIMD are ONE issue port: VALU work of a wave does not overlap the MFMAs of
//     its SIMD partner, LDS / memory / scalar work does.
// Hence this kernel:
//   * ONE workgroup of 8 waves per CU, persistent over a contiguous (XCD-aware) range of tiles;
//   * the 8 waves are two GROUPS of 4; a group owns a tile (COPT output channels x TH rows x 32 pixels) at a time;
//   * every step (one tap of one 16-channel chunk) is two barrier intervals: in the first group 0 reads its operands
//     from LDS and issues its share of the DMAs while group 1 multiplies, in the second they swap -- matrix work of
//     one wave always sits beside the memory work of its SIMD partner;
//   * time runs in SLOTS of 9 steps (one chunk).  The weight stream (one shared ring, stage = one tap of one chunk,
//     issued D steps ahead) cycles through the chunks once per nchunk slots for ever; a group spends nchunk slots on
//     the main loop of a tile and ONE slot on its epilogue, which is cut into pieces between the same barriers:
//     memory / LDS pieces in the intervals where the partner multiplies, VALU pieces where it loads.  Group 1 runs
//     SH slots behind group 0, so one group's epilogue sits beside the other's main loop.  A tile therefore starts
//     at weight chunk (slot mod nchunk) and walks the chunks cyclically (the sum over chunks is taken in a rotated
//     order; fp32 accumulation, fixed per launch geometry -> deterministic);
//   * the vector-memory issue pattern of a wave is completely static (patches are fetched in every slot, needed or
//     not; the epilogue's loads / stores are template flags), so every s_waitcnt vmcnt(N) is a compile-time count (a weight stage is
//     awaited one step before its first reader needs it: CDC_DMA_WAIT_MARGIN, conv_pf_kernel.h);
//   * the epilogue moves fp32 NCHW rows 16 bytes per lane (a 32-channel x 32-pixel block changes between the
//     accumulator layout and the row layout through a private 4-KiB LDS
//     region): 4 + 4 vector-memory instructions per block instead of 16 + 16; PF units as 8-byte halves (8 per block).
//
// Restrictions (host: pf3_make_plan): 3x3 / pad 1 / stride 1, Ho % TH == 0, Wo % 32 == 0, Cin % 16 == 0, tiles
// divisible by 2 x workgroups, one accumulator pair per wave tile (MB * NPW <= 4), epilogue options: bias, channel
// LayerNorm, (leaky) ReLU, shift, residual, fp32 and / or PF output.
#pragma once
#include "conv_pf_kernel.h"

namespace cdc {

#ifdef CDC_PF3_TL
#define PF3_TL_BYTES (8 * 64 * 8)
#else
#define PF3_TL_BYTES 0
#endif
#ifndef CDC_PF3_ABL
#define CDC_PF3_ABL 0        // lab only (tools/ubench/pf_lab.hip): 1 no result stores, 2 no residual loads, 4 no LayerNorm math, 8 no LDS transposes, 16 no PF split
#endif
// Cache policy of the epilogue's result stores.  -DCDC_PF3_ST_WT: write-through to the memory side (sc1), so that the kernel leaves no dirty
// lines for the end-of-kernel write-back -- A/B of round 6 (profiles/launch_floor_r06.txt: a boundary behind >= 16 MB of dirty lines costs
// 3.4 - 3.7 us instead of 1.5).
#ifdef CDC_PF3_ST_WT
#define CDC_PF3_ST_POLICY " sc1"
#else
#define CDC_PF3_ST_POLICY ""
#endif
#ifndef CDC_PF3_D
#define CDC_PF3_D 5          // weight stages are issued D steps ahead; ring = D + 2 slots (a slot is rewritten two steps after its last reader)
#endif

constexpr int kPf3Resid = 1, kPf3F32 = 2, kPf3Pf = 4, kPf3Stat = 8, kPf3Res3 = 16, kPf3Pre = 32, kPf3ResPf = 64;     // EPV: which vector-memory operations the epilogue issues
// (kPf3ResPf, with kPf3Resid: the residual is read from a PF tensor (PfArgs::resid_pf) -- 8-byte half-units straight in the accumulator
//  layout, value = h + l' 2^-11: 8 loads per block instead of 4 row loads + an LDS transposition)
// (kPf3Pre, with kPf3Resid: the "residual" operand is a hoisted partial sum (ConvArgs::pre_add, the step-invariant context half of a
//  concatenated input): same loads, but added to the accumulators BEFORE the LayerNorm)
// (kPf3Res3: the 3-channel res_conv of the first ResnetBlock in the epilogue, conv_args.h: res3_w / res3_x; 64-channel shape only)

__host__ __device__ constexpr int pf3_xsw(int NPW, int WP) { return (4 * (WP * NPW + 2) * 34 + 63) / 64; }
__host__ __device__ constexpr int pf3_kxw(int NPW, int WP) { return (pf3_xsw(NPW, WP) + 3) / 4; }        // patch DMAs per wave and slot
__host__ __device__ constexpr int pf3_pst(int NPW, int WP) { return pf3_kxw(NPW, WP) * 4 * 64; }          // units per patch buffer
// The kernel owns the whole LDS of its CU (the epilogue scratch sits at the top); pf3_lds_used = what lies below it.
__host__ __device__ constexpr size_t pf3_lds_bytes() { return 163840; }
__host__ __device__ constexpr size_t pf3_lds_used(int MB, int NPW, int WM, int WP, int B) {
    const size_t copt = (size_t)WM * MB * 32;
    return (size_t)(4 * pf3_pst(NPW, WP) + (CDC_PF3_D + 2) * 4 * (int)copt) * 16 +
           sizeof(float) * (((WM == 1 ? 6 : 3) + (size_t)8 + (size_t)(0 * B)) * copt +      /* parameter rows + one shift row per wave */ (size_t)2 * 2 * WM * WP * NPW * 32) + 4 * 4096 + PF3_TL_BYTES;
}

// ---- static vector-memory schedule of a wave -----------------------------------------------------------------------
// Program order inside step u of any slot: [A-piece operations nA(u)] [patch piece if u < KXW] [weight share if the wave
// has one] <wait point of step u> [B-piece operations nB(u)].  nA / nB are zero except in an epilogue slot.
struct Pf3Ops { int nA[9], nB[9]; };
__host__ __device__ constexpr Pf3Ops pf3_ops_none() { return Pf3Ops{{0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
__host__ __device__ constexpr Pf3Ops pf3_ops_epi(int nblk, int epv) {
    const int lo = 4 * (nblk < 2 ? nblk : 2), hi = 4 * (nblk > 2 ? nblk - 2 : 0);
    const int r = (epv & kPf3Resid) ? 1 : 0, f = (epv & kPf3F32) ? 1 : 0, p = (epv & kPf3Pf) ? 1 : 0;
    return Pf3Ops{{r * lo, r * hi, 0, 0, 0, 0, f * lo, 0, f * hi}, {0, 0, 0, 0, 0, 2 * p * lo, 0, 2 * p * hi, 0}};   // (PF: 8 half-unit stores per block)
}
// SYNC mode: the whole epilogue runs between two slots -- as if it were the B piece of the previous slot's last step
__host__ __device__ constexpr Pf3Ops pf3_ops_epi_sync(int nblk, int epv) {
    const int n = 4 * nblk * (((epv & kPf3Resid) ? ((epv & kPf3ResPf) ? 2 : 1) : 0) + ((epv & kPf3F32) ? 1 : 0) + ((epv & kPf3Pf) ? 2 : 0)) + ((epv & kPf3Stat) ? 4 : 0) +
                  ((epv & kPf3Res3) ? 6 : 0);
    return Pf3Ops{{0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, n}};
}
// Operations of this wave issued after the TARGET and before the wait point of step t (steps < 0: previous slot):
// target = the weight DMA of step t + 1 - D (wweight) or the last patch piece of this slot, step KXW - 1 (!wweight).
__host__ __device__ constexpr int pf3_younger(bool wwave, bool wweight, int KXW, int D, const Pf3Ops &prev, const Pf3Ops &cur, int t) {
    const int ts = wweight ? t + 1 + CDC_DMA_WAIT_MARGIN - D : KXW - 1;     // (margin: the stage of the step after next as well -- one step earlier than its first reader needs it)
    int n = 0;
    for (int u = ts; u <= t; ++u) {
        const Pf3Ops &o = u < 0 ? prev : cur;
        const int uu = u < 0 ? u + 9 : u;
        if (u > ts) n += o.nA[uu] + (uu < KXW ? 1 : 0) + (wwave ? 1 : 0);
        else if (!wweight) n += wwave ? 1 : 0;            // the weight DMA that follows the target piece in its step
        if (u < t) n += o.nB[uu];
    }
    return n < 63 ? n : 63;
}
// the wait of step t: every weight stage up to step t + 1 - D (waves with a weight share) and, at the last step of the slot,
// the whole patch of the next slot have landed
__host__ __device__ constexpr int pf3_wait(bool wwave, int KXW, int D, const Pf3Ops &prev, const Pf3Ops &cur, int t) {
    int n = 63;
    if (wwave) n = pf3_younger(wwave, true, KXW, D, prev, cur, t);
    if (t == 8) {
        const int nb = pf3_younger(wwave, false, KXW, D, prev, cur, t);
        if (nb < n) n = nb;
    }
    return n;
}

// s_barrier that the instruction scheduler may not move anything across (hipcc otherwise slides MFMAs and ds_reads
// over the barrier, which undoes the load | multiply interleave of the two groups)
__device__ __forceinline__ void pf3_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Global accesses of the epilogue in the "scalar base + 32-bit lane offset" form, hand-written: hipcc materialises a
// 64-bit vector address per access (one VALU instruction each), and VALU work of this wave cannot issue while the
// SIMD partner multiplies.  hipcc does not count these operations: the loads are awaited by pf3_wait_rows.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pf3_ld4(const char *sbase, unsigned voff) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
// (a store of more than 8 bytes needs two wait states before a VALU instruction may overwrite its data registers;
// hipcc inserts them for its own stores but does not look inside an asm statement -- without the s_nop the first dword
// of a row piece was sporadically the NEXT value computed in that register)
__device__ __forceinline__ void pf3_st4(char *sbase, unsigned voff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" CDC_PF3_ST_POLICY "\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// (a 64-bit scalar, not a 2-vector: one virtual register that hipcc has no reason to take apart -- or copy -- before the data has landed)
__device__ __forceinline__ unsigned long long pf3_ld2(const char *sbase, unsigned voff) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
__device__ __forceinline__ void pf3_st2u(char *sbase, unsigned voff, unsigned a, unsigned b) {
    const u32x2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, %2" CDC_PF3_ST_POLICY ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// s_waitcnt vmcnt(N) that the residual rows depend on (so no use is scheduled above it)
template <int N, int NB> __device__ __forceinline__ void pf3_wait_rows(f32x4 (&v)[NB][4]) {
    static_assert(NB >= 1 && NB <= 4, "blocks per wave tile");
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]) : "n"(N & kVmWaitMask) : "memory");
    if constexpr (NB > 1) asm volatile("" : "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]));
    if constexpr (NB > 2) asm volatile("" : "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]));
    if constexpr (NB > 3) asm volatile("" : "+v"(v[3][0]), "+v"(v[3][1]), "+v"(v[3][2]), "+v"(v[3][3]));
}

// the same for the half-units of a PF residual (kPf3ResPf)
template <int N, int NB> __device__ __forceinline__ void pf3_wait_units(unsigned long long (&h)[NB][4], unsigned long long (&l)[NB][4]) {
    static_assert(NB >= 1 && NB <= 4, "blocks per wave tile");
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(h[0][0]), "+v"(h[0][1]), "+v"(h[0][2]), "+v"(h[0][3]), "+v"(l[0][0]), "+v"(l[0][1]), "+v"(l[0][2]), "+v"(l[0][3]) : "n"(N & kVmWaitMask) : "memory");
    if constexpr (NB > 1) asm volatile("" : "+v"(h[1][0]), "+v"(h[1][1]), "+v"(h[1][2]), "+v"(h[1][3]), "+v"(l[1][0]), "+v"(l[1][1]), "+v"(l[1][2]), "+v"(l[1][3]));
    if constexpr (NB > 2) asm volatile("" : "+v"(h[2][0]), "+v"(h[2][1]), "+v"(h[2][2]), "+v"(h[2][3]), "+v"(l[2][0]), "+v"(l[2][1]), "+v"(l[2][2]), "+v"(l[2][3]));
    if constexpr (NB > 3) asm volatile("" : "+v"(h[3][0]), "+v"(h[3][1]), "+v"(h[3][2]), "+v"(h[3][3]), "+v"(l[3][0]), "+v"(l[3][1]), "+v"(l[3][2]), "+v"(l[3][3]));
}

// SYNC: both groups walk the SAME tile phase (one interval apart): nchunk main slots, then the whole epilogue as free-running
// code (no barriers, no DMAs inside); the weight stream is exactly one tile long, no rotated chunk order, no idle slots.
template <int MB, int NPW, int WM, int WP, int EPV, bool SYNC = false>
__global__ void __launch_bounds__(512, 1) conv_pf3_kernel(const PfArgs P) {
    static_assert(WM * WP == 4, "a group is four waves");
    static_assert(MB * NPW <= 4, "two accumulator sets per wave tile");
    constexpr bool RESID = (EPV & kPf3Resid) != 0, F32 = (EPV & kPf3F32) != 0, PF = (EPV & kPf3Pf) != 0, STAT = (EPV & kPf3Stat) != 0;
    constexpr bool PRE = (EPV & kPf3Pre) != 0;          // the loaded rows are partial sums that enter before the LayerNorm
    static_assert(!PRE || (RESID && SYNC), "kPf3Pre rides on the residual loads of the free-running epilogue");
    constexpr bool RESPF = (EPV & kPf3ResPf) != 0;      // the residual comes from a PF tensor
    static_assert(!RESPF || (RESID && SYNC && !PRE), "kPf3ResPf: a residual, free-running epilogue");
    constexpr bool RES3 = (EPV & kPf3Res3) != 0;
    static_assert(!STAT || SYNC, "LayerNorm statistics of the result: free-running epilogue only");
    static_assert(!RES3 || (SYNC && WM == 1 && NPW == 2), "epilogue res_conv: free-running epilogue, 64-channel shape");
    constexpr int EPR = WM == 1 ? 6 : 3;                                  // parameter rows (keep in step with pf3_lds_used)
    constexpr int COPT = WM * MB * 32, ROWS = 4, WST = ROWS * COPT;       // weight stage: planes {WH, WL} x two k-halves
    constexpr int U = WST / 64, UG = U / 2;                               // 1-KiB DMA units per stage / per group
    static_assert(UG >= 1 && UG <= 4, "a group issues at most one weight DMA per wave and step");
    constexpr int TH = WP * NPW, PH = TH + 2, PW = 34, PLANE = PH * PW, NX = 4 * PLANE;
    constexpr int KXW = pf3_kxw(NPW, WP), PST = pf3_pst(NPW, WP), NBLK = MB * NPW;
    constexpr int D = CDC_PF3_D, R = D + 2, TAPS = 9;
    static_assert(KXW <= TAPS - 1 && D <= TAPS, "patch / weight issue schedule");
    static_assert(PST * 16 >= 4 * 4096, "the idle patch buffer holds one 4-KiB region per wave");
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wq = wave8 & 3;
    const int wm = wq / WP, wp = wq % WP;
    const int half = lane >> 5, j = lane & 31;
    const int cog = blockIdx.y;
    const int nchunk = P.nchunk, SH = SYNC ? 0 : (nchunk + 1) >> 1;

    // ---- LDS map: [group 0: patch 0, patch 1][group 1: patch 0, patch 1][weight ring][epilogue parameters] ... [scratch]
    uint4 *patch_g = smem_u + grp * 2 * PST;
    uint4 *ring = smem_u + 4 * PST;
    float *ep = reinterpret_cast<float *>(smem_u + 4 * PST + R * WST);    // bias / scale, ln g, ln b (, res3 weights): [EPR][COPT]
    float *epsh = ep + EPR * COPT + wave8 * COPT;                         // this wave's copy of its tile's shift row: [8][COPT]
    float *red = ep + (EPR + 8) * COPT + grp * (2 * WM * WP * NPW * 32);
    float *ex = reinterpret_cast<float *>(smem_u) + (163840 - 4 * 4096 - (PF3_TL_BYTES)) / 4;   // 4 KiB per wave of the group in its epilogue
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) uint4 *)smem_u);
    const unsigned patch_lds = lds0 + (unsigned)(grp * 2 * PST) * 16u;
    const unsigned ring_lds = lds0 + (unsigned)(4 * PST) * 16u;

    // ---- tiles of this workgroup: a contiguous range (one band per XCD), exactly 2 n_iter of them; group g takes
    // t_lo + 2 i + g.  (b, oy0, ox0) advance by two tiles per iteration without divisions.
    const int n_iter = P.n_iter;
    int rank = blockIdx.x;
    if (P.xcd_remap) rank = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);
    struct Tile { int b, oy0, ox0; };
    Tile Tc, Tx;
    // walk order inside the range (PfArgs::lin != 0 here): column-major -- consecutive tiles (one per group, a barrier interval apart) are
    // VERTICAL neighbours, so the two halo rows a tile shares with the next one are fetched twice within a tile's time and meet in L2
    // (row-major: the row below comes tiles_x tiles later, after ~1 MB of other patches per workgroup)
    const bool ymajor = P.lin != 0;
    {
        const int t = rank * 2 * n_iter + grp, tiles_xy = P.tiles_x * P.tiles_y;
        Tc.b = t / tiles_xy;
        const int r = t - Tc.b * tiles_xy;
        if (ymajor) { const int tx = r / P.tiles_y; Tc.ox0 = tx * 32; Tc.oy0 = (r - tx * P.tiles_y) * TH; }
        else { const int ty = r / P.tiles_x; Tc.oy0 = ty * TH; Tc.ox0 = (r - ty * P.tiles_x) * 32; }
    }
    const int w_pix = P.tiles_x * 32, h_pix = P.tiles_y * TH;
    auto advance = [&](const Tile &T) {
        Tile N = T;
        if (ymajor) {
            N.oy0 += 2 * TH;
            while (N.oy0 >= h_pix) { N.oy0 -= h_pix; N.ox0 += 32; }
            if (N.ox0 >= w_pix) { N.ox0 -= w_pix; N.b += 1; }
        } else {
        N.ox0 += 64;
        while (N.ox0 >= w_pix) { N.ox0 -= w_pix; N.oy0 += TH; }
        if (N.oy0 >= h_pix) { N.oy0 -= h_pix; N.b += 1; }
        }
        if (N.b >= P.B) N.b = P.B - 1;                    // (past the last tile: only ever a dummy patch source)
        return N;
    };
    Tx = advance(Tc);

    // ---- patch DMA: per-lane source offsets relative to the tile's first patch unit (tile independent) ---------
    const int Hp = P.H + 2, Wp = P.W + 2;
    unsigned xoff[KXW];
#pragma unroll
    for (int i = 0; i < KXW; ++i) {
        int e = (i * 4 + wq) * 64 + lane;
        if (e >= NX) e = 0;
        const int q = e / PLANE, rem = e - q * PLANE;    // q = k-half * 2 + plane
        const int r = rem / PW, c = rem - r * PW;
        xoff[i] = (unsigned)((q * Hp + r) * Wp + c) * 16u;
    }
    const long long cg_bytes = (long long)4 * Hp * Wp * 16;
    const int c0_chunks = P.C0 >> 4;
    auto patch_src = [&](const Tile &T, int chunk) {
        const char *base = chunk < c0_chunks ? reinterpret_cast<const char *>(P.src0) + ((size_t)T.b * P.src0_bs) * 16 + (long long)chunk * cg_bytes
                                             : reinterpret_cast<const char *>(P.src1) + ((size_t)T.b * P.src1_bs) * 16 + (long long)(chunk - c0_chunks) * cg_bytes;
        return base + ((long long)T.oy0 * Wp + T.ox0) * 16;
    };
    // ---- weight DMA share of this wave: unit jj = grp * UG + wq of every stage ----------------------------------
    const bool w_wave = wq < UG;
    const int jj = grp * UG + min(wq, UG - 1);
    const unsigned wvo = (unsigned)((jj / (COPT / 64)) * P.COP + (jj % (COPT / 64)) * 64 + lane) * 16u;
    const unsigned wdo = __builtin_amdgcn_readfirstlane((unsigned)jj * 1024u);
    const char *wsrc = reinterpret_cast<const char *>(P.w) + ((size_t)cog * COPT) * 16;
    const long long w_dt = (long long)nchunk * 6 * P.COP * 16;            // next tap, same chunk
    const long long w_dc = (long long)6 * P.COP * 16 - (TAPS - 1) * w_dt;  // first tap of the next chunk
    const long long w_wrap = -(long long)nchunk * 6 * P.COP * 16;          // ... of chunk 0 after the last chunk
    const char *wptr = wsrc;                              // stage to be issued next: tap tw of chunk cw into slot sw
    int tw = 0, cw = 0, sw = 0;
    auto issue_w = [&]() {
        if (w_wave) dma16(wvo, wptr, ring_lds + (unsigned)(sw * WST) * 16u + wdo);
        if (++tw == TAPS) { tw = 0; wptr += w_dc; if (++cw == nchunk) { cw = 0; wptr += w_wrap; } } else wptr += w_dt;
        if (++sw == R) sw = 0;
    };

    f32x16 acc[MB][NPW], acc2[MB][NPW];
    const uint4 *a_base = ring + half * COPT + wm * MB * 32 + j;
    const uint4 *b_base = patch_g + (half * 2) * PLANE + (wp * NPW) * PW + j;
    typedef f16x8 OpsA[2][MB];
    typedef f16x8 OpsB[2][NPW];
    OpsA A;
    OpsB Bv;
    auto fetch = [&](auto tc, const uint4 *xb, const uint4 *wa) {
        constexpr int t = decltype(tc)::value;
        constexpr int koff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int m = 0; m < MB; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa[(pl * 2) * COPT + m * 32]);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < NPW; ++n) Bv[pl][n] = __builtin_bit_cast(f16x8, xb[pl * PLANE + n * PW + koff]);
    };
    // a = h + l' 2^-11, w 2^s = WH + WL:  acc += WL.h + WH.h,  acc2 += WH.l'   (conv_pf_kernel.h)
    auto mma = [&]() {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][m], Bv[0][n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) acc2[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], Bv[1][n], acc2[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], Bv[0][n], acc[m][n], 0, 0, 0);
    };

    // ---- one-time: epilogue parameters; weight stages 0 .. D-1; group 0's first patch ---------------------------
    for (int i = tid; i < COPT; i += 512) {
        const int co = cog * COPT + i;
        const bool ok = co < P.Cout;
        ep[i] = (ok && P.bias) ? P.bias[co] / P.acc_scale : 0.f;     // the accumulators start here (acc_scale is a power of two)
        ep[COPT + i] = (ok && P.ep_g) ? P.ep_g[co] : 0.f;
        ep[2 * COPT + i] = (ok && P.ep_b) ? P.ep_b[co] : 0.f;
        if constexpr (RES3)
            for (int c = 0; c < 3; ++c) ep[(3 + c) * COPT + i] = ok ? P.res3_w[(size_t)c * P.COP + co] : 0.f;
    }
    for (int q = 0; q < D; ++q) issue_w();
    {
        const char *src = patch_src(Tc, 0);
#pragma unroll
        for (int i = 0; i < KXW; ++i) dma16(xoff[i], src, patch_lds + (unsigned)((i * 4 + wq) * 1024));
    }
    dma_wait();
    __syncthreads();
    // Every workgroup walks the same main-loop / epilogue cycle and the epilogue carries two thirds of the HBM bytes: with
    // all CUs in step HBM idles through the main loops and saturates in the epilogues.  Workgroups therefore start
    // `stagger` eighths of ... apart (P.dbg = sleep quanta of 64 x 127 cycles per phase step).
    if (P.dbg > 0) {
        const int ph = (int)((blockIdx.x >> 3) & 7);
        for (int i = 0; i < ph * P.dbg; ++i) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
    }
    if (grp == 1) pf3_bar();                              // group 1 runs one interval behind group 0

    int sn = 0;                                           // ring slot of the stage read at the current step
    int tau = 0, wc = 0;                                  // slot counter (parity = patch buffer), its weight chunk
    // static per-step DMA + wait schedule (every slot kind): patch piece t of the NEXT slot, weight share of stage s + D
    auto step_dma = [&](auto tc, auto prevc, auto curc, const char *nsrc, unsigned ndst) {
        constexpr int t = decltype(tc)::value;
        constexpr Pf3Ops prev = decltype(prevc)::value ? (SYNC ? pf3_ops_epi_sync(NBLK, EPV) : pf3_ops_epi(NBLK, EPV)) : pf3_ops_none();
        constexpr Pf3Ops cur = decltype(curc)::value ? pf3_ops_epi(NBLK, EPV) : pf3_ops_none();
        if constexpr (t < KXW) dma16(xoff[t], nsrc, ndst + (unsigned)((t * 4 + wq) * 1024));
        issue_w();
        constexpr int NW = pf3_wait(true, KXW, D, prev, cur, t), NP = pf3_wait(false, KXW, D, prev, cur, t);
        if (w_wave) vm_wait<NW>();
        else if constexpr (NP < 63) vm_wait<NP>();
        if (++sn == R) sn = 0;
    };
    // per-lane byte offsets of the epilogue's global accesses
    const unsigned voff_p = (unsigned)((long long)j * P.pf_xs * 16);                       // PF unit of pixel column j
    // row layout: lane = 4 pixels (lane & 7) of channel rows (lane >> 3) + 8 k
    const unsigned voff_ot = (unsigned)(((long long)(lane >> 3) * P.out_cs + 4 * (lane & 7)) * 4);
    const unsigned voff_rt = (unsigned)(((long long)(lane >> 3) * P.resid_cs + 4 * (lane & 7)) * 4);
    const float inv_c = 1.0f / (float)P.Cout;
    const int cobase = cog * COPT + wm * MB * 32;
    const float *epl = ep + wm * MB * 32 + 4 * half;
    const int Lr = lane >> 3, Lc = lane & 7;

#ifdef CDC_PF3_TL
    unsigned long long tl_acc[3] = {0, 0, 0}, tl_t = __builtin_readcyclecounter();
    for (int i = tid; i < 8 * 64; i += 512) (reinterpret_cast<unsigned long long *>(smem_u) + (163840 - 8 * 64 * 8) / 8)[i] = 0;
#endif
    using False = std::false_type;
    using True = std::true_type;
    // kind 0: main chunk, 1: epilogue, 2: idle, 3: main chunk right after an epilogue slot (its waits count the epilogue's stores)
    auto slot = [&](auto kindc, bool next_tile) {
        constexpr int kind = decltype(kindc)::value;
        const Tile Tn = next_tile ? Tx : Tc;              // the next slot's patch: this tile / the next tile
        int wcn = wc + 1;
        if (wcn == nchunk) wcn = 0;
        const char *nsrc = patch_src(Tn, wcn);
        const unsigned ndst = patch_lds + (unsigned)(((tau + 1) & 1) * PST) * 16u;
        const uint4 *xb = b_base + (tau & 1) * PST;
        if constexpr (kind == 0 || kind == 3) {
            static_for<TAPS>([&](auto tc) {
                fetch(tc, xb, a_base + sn * WST);
                step_dma(tc, std::integral_constant<bool, kind == 3>{}, False{}, nsrc, ndst);
                pf3_bar();
                mma();
                pf3_bar();
            });
        } else if constexpr (kind == 2) {
            static_for<TAPS>([&](auto tc) {
                step_dma(tc, False{}, False{}, nsrc, ndst);
                pf3_bar();
                pf3_bar();
            });
        } else {
            // ---- epilogue of the group's tile, cut into pieces between the barriers of 9 steps (18 intervals) -----
            // A(t), first interval of step t: the partner group multiplies -> memory / LDS operations only;
            // B(t): the partner reads LDS and issues DMAs -> VALU work.
            // The accumulators started at bias / acc_scale, and with a LayerNorm the scale is never applied
            // (LN(s y) = (y - mean) / sqrt(var + eps / s^2)).
            // The argument block is re-read from the kernarg segment here (one batch of scalar loads) instead of living
            // in ~70 SGPRs across the main loop (they were spilled to VGPR lanes: every reload is a VALU instruction).
            const __attribute__((address_space(4))) PfArgs *Pe = (const __attribute__((address_space(4))) PfArgs *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(Pe));
            const Tile T = Tc;
            const long long o_cs = Pe->out_cs, r_cs = Pe->resid_cs, p_ps = Pe->pf_ps;
            const bool has_ln = !(CDC_PF3_ABL & 4) && Pe->ep_g != nullptr, has_shift = Pe->shift != nullptr;
            const int relu = Pe->relu;
            const float relu_slope = Pe->relu_slope, acc_scale = Pe->acc_scale;
            const float eps_s = Pe->eps / (acc_scale * acc_scale);
            const size_t row = (size_t)(T.oy0 + wp * NPW) * Pe->out_ys + (size_t)T.ox0 + Pe->out_zoff[0];
            // (a residual over a channel concatenation: this wave's channels lie wholly in one of the two sources)
            const bool r_second = RESID && !RESPF && Pe->resid1 != nullptr && cobase >= Pe->resid_c0;
            const char *r_base = (RESID && !RESPF) ? reinterpret_cast<const char *>(r_second ? Pe->resid1 + (size_t)T.b * Pe->resid1_bs + (size_t)(cobase - Pe->resid_c0) * r_cs + row
                                                                                             : Pe->resid + (size_t)T.b * Pe->resid_bs + (size_t)cobase * r_cs + row) : nullptr;
            const long long rp_ps = Pe->rpf_ps, rp_ys = Pe->rpf_ys;
            const char *rp_base = RESPF ? reinterpret_cast<const char *>(reinterpret_cast<const uint4 *>(Pe->resid_pf) + (long long)T.b * Pe->rpf_bs + (long long)(cobase >> 3) * 2 * rp_ps +
                                                                       (long long)(T.oy0 + wp * NPW) * rp_ys + (long long)T.ox0 + Pe->rpf_zoff) : nullptr;
            const unsigned voff_rp = (unsigned)(j * 16 + half * 8);
            unsigned long long rvH[NBLK][4], rvL[NBLK][4];   // PF residual: this lane's half-units of the two planes, per 8-channel group
            char *o_base = F32 ? reinterpret_cast<char *>(Pe->out + (size_t)T.b * Pe->out_bs + (size_t)cobase * o_cs + row) : nullptr;
            char *p_base = PF ? reinterpret_cast<char *>(reinterpret_cast<uint4 *>(Pe->out_pf) + (long long)T.b * Pe->pf_bs + (long long)(cobase >> 3) * 2 * p_ps +
                                                         (long long)(T.oy0 + wp * NPW) * Pe->pf_ys + (long long)T.ox0 * Pe->pf_xs + Pe->pf_zoff[0]) : nullptr;
            const long long o_ys = Pe->out_ys, p_ys = Pe->pf_ys;
            float rinv_v[NPW], part[NPW];
            f32x4 rvT[NBLK][4];                           // residual operand, row layout
            float rv[2][16];                              // ... of the two blocks in flight, accumulator layout
            const int slot_i = (wp * NPW) * 32 + j;
            // the per-image shift row of this tile: every wave keeps its own copy (no barrier; a row per image of the batch did not
            // fit next to the ring beyond batch 32)
            if (has_shift) {
                const float *srow = Pe->shift + (size_t)T.b * Pe->shift_bs + cog * COPT;
                for (int i = lane; i < COPT; i += 64) epsh[i] = (cog * COPT + i < Pe->Cout) ? srow[i] : 0.f;
            }
            const float *shl = epsh + wm * MB * 32 + 4 * half;
            // private LDS regions of this wave: block q uses region q & 1 (the group's idle patch buffer / the scratch area)
            // (SYNC: tau already points at the next tile's first patch; the buffer of the tile's last chunk is the idle one, and both
            // groups are in their epilogue at once -> one region per wave)
            float *xr0 = reinterpret_cast<float *>(patch_g + ((SYNC ? tau + 1 : tau) & 1) * PST) + wq * 1024;
            float *xr1 = SYNC ? xr0 : ex + wq * 1024;
            auto bar = [&]() { pf3_bar(); };
            auto sum_all = [&](int n, auto &&f) {         // packed partial sums (no 32-deep dependent chain)
                f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 v = {acc[m][n][r], acc[m][n][r + 1]};
                        s2[(r >> 1) & 1] = f(v, s2[(r >> 1) & 1]);
                    }
                const f32x2 s = s2[0] + s2[1];
                float sm = s[0] + s[1];
                sm += __shfl_xor(sm, 32);
                return sm;
            };
            auto ldT = [&](auto qc) {                     // residual rows of block q: 4 x 16 bytes per lane
                constexpr int q = decltype(qc)::value, n = q / MB, m = q % MB;
                if constexpr (q < NBLK && RESPF) {        // ... or its 2 x 4 half-units of a PF tensor
                    const char *sb = rp_base + ((long long)(m * 4) * 2 * rp_ps + (long long)n * rp_ys) * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        rvH[q][g] = pf3_ld2(sb, voff_rp);
                        rvL[q][g] = pf3_ld2(sb + rp_ps * 16, voff_rp);
                        sb += 2 * rp_ps * 16;
                    }
                } else if constexpr (q < NBLK && RESID) {
                    const char *sb = r_base + ((size_t)(m * 32) * r_cs + (size_t)n * o_ys) * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!(CDC_PF3_ABL & 2)) rvT[q][k] = pf3_ld4(sb, voff_rt);
                        else rvT[q][k] = f32x4{1.f, 2.f, 3.f, 4.f};
                        sb += 8 * r_cs * 4;
                    }
                }
            };
            auto wT_rN = [&](auto qc) {                   // residual of block q: rows -> region -> accumulator layout
                constexpr int q = decltype(qc)::value;
                if constexpr (q < NBLK && RESPF) {        // half-units are in the accumulator layout already: a = h + l' 2^-11
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const f16x2 hv = __builtin_bit_cast(f16x2, (unsigned)(rvH[q][g] >> (32 * k))), lv = __builtin_bit_cast(f16x2, (unsigned)(rvL[q][g] >> (32 * k)));
                            rv[q & 1][g * 4 + 2 * k] = __builtin_fmaf((float)lv[0], 1.0f / 2048.0f, (float)hv[0]);
                            rv[q & 1][g * 4 + 2 * k + 1] = __builtin_fmaf((float)lv[1], 1.0f / 2048.0f, (float)hv[1]);
                        }
                } else if constexpr (q < NBLK && RESID) {
                    float *xr = (q & 1) ? xr1 : xr0;
                    if constexpr ((CDC_PF3_ABL & 8) != 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv[q & 1][r] = rvT[q][r >> 2][r & 3];
                    } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(xr + (Lr + 8 * k) * 32 + Lc * 4) = rvT[q][k];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[q & 1][r] = xr[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j];
                    }
                }
            };
            auto wN_rT_st = [&](auto qc) {                // final values of block q: accumulator layout -> region -> rows -> HBM
                constexpr int q = decltype(qc)::value, n = q / MB, m = q % MB;
                if constexpr (q < NBLK && F32) {
                    float *xr = (q & 1) ? xr1 : xr0;
                    f32x4 rows[4];
                    if constexpr ((CDC_PF3_ABL & 8) != 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) rows[k] = f32x4{acc[m][n][4 * k], acc[m][n][4 * k + 1], acc[m][n][4 * k + 2], acc[m][n][4 * k + 3]};
                    } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) xr[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[m][n][r];
#pragma unroll
                    for (int k = 0; k < 4; ++k) rows[k] = *reinterpret_cast<const f32x4 *>(xr + (Lr + 8 * k) * 32 + Lc * 4);
                    }
                    char *sb = o_base + ((size_t)(m * 32) * o_cs + (size_t)n * o_ys) * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!(CDC_PF3_ABL & 1)) pf3_st4(sb, voff_ot, rows[k]);
                        else asm volatile("" ::"v"(rows[k]));
                        sb += 8 * o_cs * 4;
                    }
                }
            };
            auto merge = [&]() {                          // y = acc + acc2 2^-11   (bias / scale is already in acc)
#pragma unroll
                for (int n = 0; n < NPW; ++n)
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = __builtin_fmaf(acc2[m][n][r], 1.0f / 2048.0f, acc[m][n][r]);
            };
            auto sums = [&]() {
                if (has_ln) {
#pragma unroll
                    for (int n = 0; n < NPW; ++n) part[n] = sum_all(n, [](f32x2 v, f32x2 s) { return s + v; });
                    if constexpr (WM > 1) {
#pragma unroll
                        for (int n = 0; n < NPW; ++n)
                            if (half == 0) red[wm * (WP * NPW * 32) + slot_i + n * 32] = part[n];
                    }
                }
            };
            auto devs = [&](auto nc) {                    // acc <- y - mean;  sum of squares
                constexpr int n = decltype(nc)::value;
                if constexpr (n < NPW)
                    if (has_ln) {
                        if constexpr (WM > 1) {
                            float sm = 0.f;
#pragma unroll
                            for (int q = 0; q < WM; ++q) sm += red[q * (WP * NPW * 32) + slot_i + n * 32];
                            part[n] = sm;
                        }
                        const float mu = part[n] * inv_c;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][n][r] -= mu;
                        part[n] = sum_all(n, [](f32x2 v, f32x2 s) { return __builtin_elementwise_fma(v, v, s); });
                        if constexpr (WM > 1)
                            if (half == 0) red[(WM + wm) * (WP * NPW * 32) + slot_i + n * 32] = part[n];
                    }
            };
            auto norm = [&](auto qc) {                    // normalise, ReLU, shift of block q
                constexpr int q = decltype(qc)::value, n = q / MB, m = q % MB;
                if constexpr (q < NBLK) {
                    if (has_ln) {
                        f32x4 gq[4], bq[4];                // all parameter reads first, ONE wait (hipcc otherwise reads, waits, uses, ...)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            gq[k] = *reinterpret_cast<const f32x4 *>(epl + COPT + m * 32 + 8 * k);
                            bq[k] = *reinterpret_cast<const f32x4 *>(epl + 2 * COPT + m * 32 + 8 * k);
                        }
                        if constexpr (m == 0) {
                            if constexpr (WM > 1) {
                                float sm = 0.f;
#pragma unroll
                                for (int w = 0; w < WM; ++w) sm += red[(WM + w) * (WP * NPW * 32) + slot_i + n * 32];
                                part[n] = sm;
                            }
                            const float var = part[n] * inv_c + eps_s;
                            // range guard (ConvArgs::fault): a non-finite accumulator shows in the variance; reported before the
                            // LayerNorm + ReLU below can turn it into a finite value.  (One more vector-memory operation in this
                            // wave's queue only makes the counted waits of the static schedule stricter, never looser.)
                            if (!(var < 3.0e38f) && Pe->fault) *Pe->fault = 1;
                            float y = __builtin_amdgcn_rsqf(var);      // + one Newton step: full fp32 accuracy without the division sequence
                            y = y * (1.5f - 0.5f * var * y * y);
                            rinv_v[n] = y;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = __builtin_fmaf(acc[m][n][r] * rinv_v[n], gq[r >> 2][r & 3], bq[r >> 2][r & 3]);
                    } else {
                        float sa = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc[m][n][r] *= acc_scale; sa += fabsf(acc[m][n][r]); }
                        if (!(sa < 3.0e38f) && Pe->fault) *Pe->fault = 1;
                    }
                    if (relu) {
                        // max(v, slope v) also for slope 0: 0 * NaN = NaN keeps a NaN alive (v_max_f32(NaN, 0) would return 0)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = fmaxf(acc[m][n][r], relu_slope * acc[m][n][r]);
                    }
                    if (has_shift) {
                        f32x4 sq[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) sq[k] = *reinterpret_cast<const f32x4 *>(shl + m * 32 + 8 * k);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] += sq[r >> 2][r & 3];
                    }
                }
            };
            float x3[NPW][3];                             // res_conv input: the 3 image channels of this lane's pixels
            auto ldX = [&]() {
                if constexpr (RES3) {
                    const float *xp = Pe->res3_x + (size_t)T.b * Pe->res3_bs + (size_t)(T.oy0 + wp * NPW) * Pe->Wo + T.ox0 + j;
                    const size_t hw = (size_t)Pe->Ho * Pe->Wo;
#pragma unroll
                    for (int n = 0; n < NPW; ++n)
#pragma unroll
                        for (int c = 0; c < 3; ++c) x3[n][c] = xp[(size_t)c * hw + (size_t)n * Pe->Wo];
                }
            };
            auto fin_pf = [&](auto qc) {                  // + residual; PF planes of block q
                constexpr int q = decltype(qc)::value, n = q / MB, m = q % MB;
                if constexpr (q < NBLK) {
                    if constexpr (RESID && !PRE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] += rv[q & 1][r];
                    }
                    if constexpr (RES3) {                 // += sum_c res3_w[c][co] x[c][pixel]   (conv_pf_kernel.h: same expression)
                        f32x4 w3[3][4];
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int k = 0; k < 4; ++k) w3[c][k] = *reinterpret_cast<const f32x4 *>(epl + (3 + c) * COPT + m * 32 + 8 * k);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[m][n][r] += w3[0][r >> 2][r & 3] * x3[n][0] + w3[1][r >> 2][r & 3] * x3[n][1] + w3[2][r >> 2][r & 3] * x3[n][2];
                    }
                    if constexpr (PF) {
                        // per 8-channel group the lane owns 4 consecutive channels = 8 bytes of each plane's 16-byte unit
                        char *sb = p_base + ((long long)(m * 4) * 2 * p_ps + (long long)n * p_ys) * 16;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            unsigned hw[2], lw[2];
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float x0 = acc[m][n][g * 4 + 2 * k], x1 = acc[m][n][g * 4 + 2 * k + 1];
                                const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
                                // l' = fp16((x - h) 2^11) (split2h), as one exact fused operation per value
                                const _Float16 l0 = (_Float16)__builtin_fmaf((float)h0, -2048.0f, x0 * 2048.0f);
                                const _Float16 l1 = (_Float16)__builtin_fmaf((float)h1, -2048.0f, x1 * 2048.0f);
                                f16x2 hv = {h0, h1}, lv = {l0, l1};
                                hw[k] = __builtin_bit_cast(unsigned, hv);
                                lw[k] = __builtin_bit_cast(unsigned, lv);
                            }
                            // (completing the 16-byte units in registers with v_permlane32_swap and storing 16 bytes per lane was
                            // tried: no faster -- lanes l / l + 32 each write their 8-byte half)
                            if (!(CDC_PF3_ABL & 1)) {
                            pf3_st2u(sb, voff_p + (unsigned)half * 8u, hw[0], hw[1]);
                            pf3_st2u(sb + p_ps * 16, voff_p + (unsigned)half * 8u, lw[0], lw[1]);
                            } else asm volatile("" ::"v"(hw[0]), "v"(hw[1]), "v"(lw[0]), "v"(lw[1]));
                            sb += 2 * p_ps * 16;
                        }
                    }
                }
            };
            using Q0 = std::integral_constant<int, 0>;
            using Q1 = std::integral_constant<int, 1>;
            using Q2 = std::integral_constant<int, 2>;
            using Q3 = std::integral_constant<int, 3>;
            auto piece = [&](auto pc) {                   // keep in step with pf3_ops_epi
                constexpr int p = decltype(pc)::value, t = p >> 1;
                if constexpr ((p & 1) == 0) {             // A(t): memory / LDS operations
                    if constexpr (t == 0) { ldT(Q0{}); ldT(Q1{}); }
                    else if constexpr (t == 1) { ldT(Q2{}); ldT(Q3{}); }
                    else if constexpr (t == 4) {
                        if constexpr (RESID) {            // younger than the last residual load: the DMAs of steps tl .. 3
                            constexpr int tl = NBLK > 2 ? 1 : 0;
                            constexpr int NPc = (tl < KXW) + (tl + 1 < KXW && tl + 1 <= 3) + (tl + 2 < KXW && tl + 2 <= 3) + (tl + 3 < KXW && tl + 3 <= 3);
                            constexpr int NWc = NPc + (4 - tl);
                            if constexpr (SYNC) pf3_wait_rows<0>(rvT);
                            else if (w_wave) pf3_wait_rows<NWc>(rvT); else pf3_wait_rows<NPc>(rvT);
                        }
                        wT_rN(Q0{}); wT_rN(Q1{});
                    }
                    else if constexpr (t == 6) { wN_rT_st(Q0{}); wN_rT_st(Q1{}); }
                    else if constexpr (t == 7) { wT_rN(Q2{}); wT_rN(Q3{}); }
                    else if constexpr (t == 8) { wN_rT_st(Q2{}); wN_rT_st(Q3{}); }
                } else {                                  // B(t): VALU work
                    if constexpr (t == 0) merge();
                    else if constexpr (t == 1) sums();
                    else if constexpr (t == 2) devs(Q0{});
                    else if constexpr (t == 3) devs(Q1{});
                    else if constexpr (t == 4) { norm(Q0{}); norm(Q1{}); }
                    else if constexpr (t == 5) { fin_pf(Q0{}); fin_pf(Q1{}); }
                    else if constexpr (t == 6) { norm(Q2{}); norm(Q3{}); }
                    else if constexpr (t == 7) { fin_pf(Q2{}); fin_pf(Q3{}); }
                }
            };
#ifdef CDC_PF3_TL
            unsigned long long *tlp = reinterpret_cast<unsigned long long *>(smem_u) + (163840 - 8 * 64 * 8) / 8 + wave8 * 64;
            unsigned long long te = __builtin_readcyclecounter();
#define PF3_TLE(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane == 0) tlp[i] += n_ - te; te = n_; } while (0)
#else
#define PF3_TLE(i) do { } while (0)
#endif
            if constexpr (SYNC) {                       // free-running; the cross-wave LayerNorm exchange (WM > 1) needs its own barriers
                ldT(Q0{}); ldT(Q1{}); ldT(Q2{}); ldT(Q3{});   // (both groups are here in the same interval: the counts match)
                ldX();
                merge();
                if constexpr (PRE) {                    // hoisted partial sums: into the accumulators (which are in units of acc_scale) first
                    pf3_wait_rows<0>(rvT);
                    const float inv_scale = 1.0f / acc_scale;
                    static_for<NBLK>([&](auto qc) {
                        constexpr int q = decltype(qc)::value, n = q / MB, m = q % MB;
                        wT_rN(qc);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = __builtin_fmaf(rv[q & 1][r], inv_scale, acc[m][n][r]);
                    });
                }
                sums();
                if constexpr (WM > 1) bar();
                devs(Q0{}); devs(Q1{});
                if constexpr (WM > 1) bar();
                if constexpr (RESPF) pf3_wait_units<0>(rvH, rvL);
                else if constexpr (RESID && !PRE) pf3_wait_rows<0>(rvT);
                if constexpr (!PRE) { wT_rN(Q0{}); wT_rN(Q1{}); }
                norm(Q0{}); norm(Q1{});
                fin_pf(Q0{}); fin_pf(Q1{});
                wN_rT_st(Q0{}); wN_rT_st(Q1{});
                if constexpr (!PRE) { wT_rN(Q2{}); wT_rN(Q3{}); }
                norm(Q2{}); norm(Q3{});
                fin_pf(Q2{}); fin_pf(Q3{});
                wN_rT_st(Q2{}); wN_rT_st(Q3{});
                if constexpr (STAT) {                   // channel statistics of the FINAL values, for the next PreNorm (conv_pf_kernel: stat_mean / stat_rstd)
                    float *s_mean = Pe->stat_mean + (size_t)T.b * o_cs + row, *s_rstd = Pe->stat_rstd + (size_t)T.b * o_cs + row;
                    const float eps = Pe->eps;
                    float mu[NPW];
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        part[n] = sum_all(n, [](f32x2 v, f32x2 s2) { return s2 + v; });
                        if constexpr (WM > 1)
                            if (half == 0) red[wm * (WP * NPW * 32) + slot_i + n * 32] = part[n];
                    }
                    if constexpr (WM > 1) bar();
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        if constexpr (WM > 1) {
                            float sm = 0.f;
#pragma unroll
                            for (int q = 0; q < WM; ++q) sm += red[q * (WP * NPW * 32) + slot_i + n * 32];
                            part[n] = sm;
                        }
                        mu[n] = part[n] * inv_c;
                        const f32x2 m2 = {mu[n], mu[n]};
                        part[n] = sum_all(n, [m2](f32x2 v, f32x2 s2) { const f32x2 d = v - m2; return __builtin_elementwise_fma(d, d, s2); });
                        if constexpr (WM > 1)
                            if (half == 0) red[(WM + wm) * (WP * NPW * 32) + slot_i + n * 32] = part[n];
                    }
                    if constexpr (WM > 1) bar();
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        if constexpr (WM > 1) {
                            float sm = 0.f;
#pragma unroll
                            for (int q = 0; q < WM; ++q) sm += red[(WM + q) * (WP * NPW * 32) + slot_i + n * 32];
                            part[n] = sm;
                        }
                        if (half == 0) {                // (every channel part stores the same values: the operation count of a wave stays static)
                            s_mean[(size_t)n * o_ys + j] = mu[n];
                            s_rstd[(size_t)n * o_ys + j] = 1.0f / sqrtf(part[n] * inv_c + eps);
                        }
                    }
                }
            } else
            static_for<TAPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                piece(std::integral_constant<int, 2 * t>{});
                PF3_TLE(5 * t);
                step_dma(tc, False{}, True{}, nsrc, ndst);
                PF3_TLE(5 * t + 1);
                bar();
                PF3_TLE(5 * t + 2);
                piece(std::integral_constant<int, 2 * t + 1>{});
                PF3_TLE(5 * t + 3);
                bar();
                PF3_TLE(5 * t + 4);
            });
        }
        if constexpr (!(SYNC && kind == 1)) {
            ++tau;
            wc = wcn;
        }
#ifdef CDC_PF3_TL
        { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[kind == 3 ? 0 : kind] += n_ - tl_t; tl_t = n_; }
#endif
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    for (int q = 0; q < grp * SH; ++q) slot(K2{}, false);
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[m][n][r] = epl[m * 32 + (r & 3) + 8 * (r >> 2)]; acc2[m][n][r] = 0.f; }
        if (it == 0) slot(K0{}, nchunk == 1); else slot(K3{}, nchunk == 1);
        for (int jc = 1; jc < nchunk; ++jc) slot(K0{}, jc == nchunk - 1);
        if constexpr (SYNC) {
            // both groups run their epilogue in the SAME barrier interval: group 0 idles one interval (group 1 is still
            // multiplying its last tap), group 1 idles one after it (group 0 already loads the next tile's first tap).
            // The barrier after the epilogue also keeps the next slot's patch DMAs (issued by the fastest wave) off the
            // scratch regions of waves that are still in their epilogue.
            if (grp == 0) pf3_bar();
            slot(K1{}, true);
            pf3_bar();
            if (grp == 1) pf3_bar();
        } else {
            slot(K1{}, true);
        }
        Tc = Tx;
        Tx = advance(Tc);
    }
    for (int q = 0; q < (1 - grp) * SH; ++q) slot(K2{}, false);
    if (grp == 0) pf3_bar();
    dma_wait();
#ifdef CDC_PF3_TL
    if (P.res3_x && blockIdx.x < 16 && lane == 0) {
        unsigned long long *o = (unsigned long long *)P.res3_x + (blockIdx.x * 8 + wave8) * 64;
        const unsigned long long *tlp = reinterpret_cast<unsigned long long *>(smem_u) + (163840 - 8 * 64 * 8) / 8 + wave8 * 64;
        for (int q = 0; q < 45; ++q) o[q] = tlp[q];
        o[60] = tl_acc[0]; o[61] = tl_acc[1]; o[62] = tl_acc[2];
    }
#endif
}

}  // namespace cdc
