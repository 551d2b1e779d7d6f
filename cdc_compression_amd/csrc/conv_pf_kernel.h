// conv_pf_kernel.h -- stride-1 k x k / 1x1 / phase-decomposed transposed convolution whose activation operand
// arrives PRE-SPLIT: the producer's epilogue stores every activation tensor a second time as two fp16 planes
// (h, l' of split2h, conv_split_kernel.h) in MFMA B-operand order, so this kernel's main loop has no VALU work,
// no staging registers and no ds_write at all -- both operands go HBM/L2 -> LDS by LDS-DMA and straight into
// v_mfma_f32_32x32x16_f16 (three products per fp32 product, see conv_split_kernel.h AR = 1).
//
// "PF" tensor (planar fp16) of a [B][C][H][W] activation, C % 8 == 0:
//      unit = 8 consecutive channels of ONE pixel of ONE plane = 16 bytes = one lane's B operand;
//      layout [B][C/8][plane 0..1][H + 2][W + 2] units, i.e. a ONE-PIXEL ZERO HALO is stored with the tensor
//      (written once when the program is built, never touched by a producer), so a 3x3 / pad-1 patch is a plain
//      affine address pattern: no bounds logic, no masked lanes, no zero source.
//      Same bytes per element as fp32 NCHW (2 x 2).
//
// Workgroup = WM x WP waves: wave (wm, wp) owns output channels [wm*MB*32, +MB*32) of the workgroup's cout group
// and NPW 32-pixel blocks stacked vertically; WM > 1 shares one patch between the channel parts (a 256-channel
// layer is ONE workgroup column: the fused channel-LayerNorm reduces across waves through LDS).
//
// Pipeline (one iteration = one tap of one 16-channel chunk):
//   * weight stages [3 planes][2 k-halves][COPT] x 16 B stream through an R-slot LDS ring, issued R-2 taps ahead
//     by the "weight waves" (all but the last two) and awaited with a COUNTED s_waitcnt vmcnt (never 0 in the
//     steady state);
//   * the patch [2 k-halves][2 planes][PH][PW] x 16 B of chunk c+1 is issued by the last two waves ("patch
//     waves"), a slice per tap, while chunk c is being multiplied, into the other of two patch buffers; their VM
//     queues hold nothing else, so their only wait is one vmcnt(0) per chunk, a whole chunk after the issue;
//   * one s_barrier per tap publishes the landed stage; the operands of tap s+1 are read from LDS into a second
//     register set while the MFMAs of tap s issue.
#pragma once
#include <type_traits>

#include "conv_split_kernel.h"

namespace cdc {


// Counted wait: the N newest vector-memory operations of the wave may stay in flight, everything older has completed.  This is
// exact because a wave's vector-memory operations leave the counter in issue order on gfx9-class parts -- loads, stores and LDS-DMA
// alike (it is the model hipcc's own s_waitcnt insertion uses on this target; tools/ubench/dma_order.hip: 1e11 counted waits with
// cold / hot / mixed LDS-DMA pieces, both piece sizes and stores in flight, no piece found missing).  What a counted wait
// does NOT guarantee for a 16-byte piece is that the piece it just covered is readable in full at once: see CDC_DMA_WAIT_MARGIN below.
// -DCDC_DMA_WAIT_ALL (A/B build, profiles/determinism_r06.txt): every counted wait of the plane-operand kernels becomes vmcnt(0).
// CDC_DMA_WAIT_MARGIN (default 1 since round 6): the weight-stage waits of conv_pf_kernel / conv_pf3_kernel ask for one stage MORE than the
// next reader needs.  Why: conv_pw_kernel's zero-margin counted wait on 16-byte LDS-DMA pieces let a wave read a piece that had not fully landed
// once in ~10 000 launches (profiles/determinism_r06.txt); a margin of one stage cut that rate 20 - 40 fold there, and costs nothing here
// (11.976 / 11.959 against 11.980 / 11.962 ms per iteration, batch 32, -DCDC_DMA_WAIT_MARGIN=0 is the round-5 form).
#ifndef CDC_DMA_WAIT_MARGIN
#define CDC_DMA_WAIT_MARGIN 1
#endif
#ifdef CDC_DMA_WAIT_ALL
constexpr int kVmWaitMask = 0;
#else
constexpr int kVmWaitMask = ~0;
#endif
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N & kVmWaitMask) : "memory"); }

// LDS-DMA with the destination in M0 (declared clobbered: no save / restore around every piece) and a 64-bit
// scalar base + 32-bit per-lane byte offset.  The s_nop is the wait state between the M0 write and the DMA.
__device__ __forceinline__ void dma16(unsigned voff, const void *sbase, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte)
                 : "memory");
}

template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Weight ring depth of a kernel shape: what fits next to the patch buffers in the LDS budget (two 4-wave workgroups
// or one 8-wave workgroup per CU).  Compile-time in the kernel (the counted waits become immediates, the ring
// arithmetic folds), shared with the host planner.
// Weight rows per stage: planes x two k-halves.  Shapes whose accumulators fit twice in the register file use
// the planes {WH, WL} and a second accumulator set (see mma); the others the three planes {WH, WL, WH2}.
__host__ __device__ constexpr bool pf_acc2(int MB, int NPW) { return MB * NPW <= 4; }
__host__ __device__ constexpr int pf_rows(int MB, int NPW) { return pf_acc2(MB, NPW) ? 4 : 6; }
// (TZ = 4: the four phases of a transposed convolution share the 3x3 neighbourhood patch of their input tile)
__host__ __device__ constexpr int pf_patch_units(int NPW, int WP, int KH, int KW, int STR = 1, int TZ = 1) {
    return TZ == 4 ? ((4 * (WP * NPW + 2) * 34 + 63) / 64) * 64 : ((4 * ((WP * NPW - 1) * STR + KH) * (31 * STR + KW) + 63) / 64) * 64;
}
__host__ __device__ constexpr int pf_uf_stage_floats(int NPW, int WP) { return ((3 * (WP * NPW + 6) * 38 + 63) / 64) * 64; }
// Patch buffers: the patch of chunk c + LA streams in while chunk c is multiplied, LA = buffers - 1.  1x1 layers (one tap per chunk)
// and stride-2 layers (one workgroup of one wave per SIMD on the CU: nothing else hides the HBM latency of a patch) run two
// chunks ahead.
#ifndef CDC_PF_S2_NPB
#define CDC_PF_S2_NPB 2      // (3 = two chunks ahead was measured: not faster, 0.281 -> 0.305 ms on 64 -> 64 @256^2 -> 128^2)
#endif
__host__ __device__ constexpr int pf_patch_bufs(int KH, int KW, int STR = 1) { return STR == 2 ? CDC_PF_S2_NPB : (KH * KW == 1 ? 3 : 2); }
// (stride 2: the patch of a 4-row tile is 9 x 65 pixels = 37 KB -- ONE workgroup per CU)
// tps = taps per weight stage (one s_barrier per stage)
__host__ __device__ constexpr int pf_ring_tps(int MB, int NPW, int WM, int WP, int KH, int KW, int STR, int tps, int TZ = 1) {
    // (KH x 1 with KH = 7: the first layer built from the 3-channel image, UF below -- its fp32 staging area of 3 x PH x 38 floats sits
    //  behind the ring)
    const size_t budget = ((WM * WP == 8 || STR == 2) ? 156 * 1024 : 80 * 1024) - ((KH == 7 && KW == 1) ? pf_uf_stage_floats(NPW, WP) * 4 : 0);
    const size_t patch = (size_t)pf_patch_bufs(KH, KW, STR) * pf_patch_units(NPW, WP, KH, KW, STR, TZ) * 16;
    const size_t wst = (size_t)tps * pf_rows(MB, TZ * NPW) * WM * MB * 32 * 16;
#ifndef CDC_PF_RING_MAX
#define CDC_PF_RING_MAX 5
#endif
    for (int r = CDC_PF_RING_MAX; r >= 3; --r)
        if (patch + r * wst <= budget) return r;
    return 0;
}
// Taps per weight stage.  The stride-2 form runs ONE wave per SIMD with 3 - 12 MFMAs per tap: a barrier per tap leaves the
// matrix pipe idle most of the time, so a stage holds a whole kernel row where the ring still fits.
#ifndef CDC_PF_S2_TPS
#define CDC_PF_S2_TPS 3
#endif
#ifndef CDC_PF_TZ_MINB
#define CDC_PF_TZ_MINB 2     // workgroups per CU the 4-wave fused-phase shape is compiled for (2: 256 registers per lane)
#endif
#ifndef CDC_PF_TZ_TPS
#define CDC_PF_TZ_TPS 2      // fused-phase transposed form: a stage = one kernel row of one phase
#endif
__host__ __device__ constexpr int pf_tps(int MB, int NPW, int WM, int WP, int KH, int KW, int STR = 1, int TZ = 1) {
    if (TZ == 4) return (CDC_PF_TZ_TPS > 1 && pf_ring_tps(MB, NPW, WM, WP, KH, KW, STR, CDC_PF_TZ_TPS, TZ) >= 3) ? CDC_PF_TZ_TPS : 1;
    return (STR == 2 && CDC_PF_S2_TPS > 1 && pf_ring_tps(MB, NPW, WM, WP, KH, KW, STR, CDC_PF_S2_TPS) >= 3) ? CDC_PF_S2_TPS : 1;
}
__host__ __device__ constexpr int pf_ring(int MB, int NPW, int WM, int WP, int KH, int KW, int STR = 1, int TZ = 1) {
    return pf_ring_tps(MB, NPW, WM, WP, KH, KW, STR, pf_tps(MB, NPW, WM, WP, KH, KW, STR, TZ), TZ);
}
#ifndef CDC_PF_ABLATE
#define CDC_PF_ABLATE 0      // 1: honour PfArgs::dbg (timing experiments with wrong results)
#endif

// KH x KW are compile-time (3x3, 1x1, 2x2 phases): tap offsets become ds_read immediates, the tap loop and the
// patch-issue schedule are unrolled, and the per-tap scalar work shrinks to the ring counters.
//
// STR = 2 (3x3 / pad 1 Downsample convolutions, network_components.py:51-53): the tile's patch is (2 TH + 1) x 65 pixels; the DMA
// lanes de-interleave its columns -- even columns first, then the odd ones -- so that the B operand of a tap (pixel 2j + kx of
// lane j) is again one contiguous run of units: column position kx/2 of the even (kx even) or odd (kx odd) half-row.
//
// TZ = 4 (ConvTranspose2d 4x4 / stride 2 / pad 1 as four 2x2 phase convolutions, network_components.py:34-48, KH = KW = 2): ONE
// workgroup evaluates all four phases of its input tile from one shared 3x3-neighbourhood patch -- 16 taps per chunk (phase-major),
// four accumulator sets per wave (block q = phase * NPW + n); one prologue, one patch and one epilogue per 4 x 128 output pixels
// instead of four of each (the phase-per-workgroup form, gridDim.z = 4, was slower than the register-staged kernel).
//
// UF = 3 (the first layer, unet.py:78 / network_components.py:83: a 7x7 convolution of the 3-channel image, run as a 7x1 convolution over
// its 21 kx-unfolded channels = two 16-channel chunks): there is no PF tensor to fetch -- the workgroup stages the image patch
// (3 x (TH + 6) x 38 floats) and BUILDS the two chunks' patch buffers itself (split into planes, channel kx * 3 + ci of pixel x = image
// channel ci at x + kx - 3, zero beyond the image and for channels >= 21); the weight ring, the tap loop and the epilogue are the kernel's
// own.  Replaces conv_split2_kernel's unfold-on-load form where the launch is large enough.
template <int MB, int NPW, int WM, int WP, int KH, int KW, int STR = 1, int TZ = 1, int UF = 0>
__global__ void __launch_bounds__(64 * WM * WP, (KH == 1 && KW == 7) ? 3 : STR == 2 ? 1 : ((WM * WP == 8 || MB * NPW * TZ <= 4 || (TZ == 4 && MB * NPW <= 2 && CDC_PF_TZ_MINB == 2)) ? 2 : 1)) conv_pf_kernel(const PfArgs P) {
    constexpr int NW = WM * WP, NT = 64 * NW, COPT = WM * MB * 32;
    // (COPT = 32: the row-folded final convolution, 21 of 32 channels real -- host: COP == 32, so the rows of a stage are contiguous
    //  in the source as well and an instruction simply covers two of them)
    static_assert(COPT % 64 == 0 || COPT == 32, "a weight DMA instruction (64 units) must stay inside one (plane, k-half) row");
    static_assert(TZ == 1 || (TZ == 4 && KH == 2 && KW == 2 && STR == 1), "fused phases: the 2x2 phase form of the 4x4 transposed convolution");
    constexpr int NB = TZ * NPW;                         // accumulator blocks per wave and channel block
    constexpr bool ACC2 = pf_acc2(MB, NB);
    constexpr int NPL = ACC2 ? 2 : 3, ROWS = 2 * NPL;     // weight planes / rows per stage
    constexpr int WI = (ROWS * COPT + 63) / 64;            // DMA instructions per weight stage
    constexpr int NWV = NW - 2;                         // weight waves 0 .. NW-3; patch waves NW-2, NW-1
    constexpr int NWW = (WI + NWV - 1) / NWV;           // DMA instructions per stage and weight wave
    constexpr int TAPZ = KH * KW, TAPS = TAPZ * TZ;      // taps per phase / per chunk
    constexpr int NBW = 32, NBH = 1;                    // a 32-pixel block is a row segment (host: lognbw = 5)
    constexpr int TH = WP * NPW * NBH;
    constexpr int PH = TZ == 4 ? TH + 2 : (TH - 1) * STR + KH, PW = TZ == 4 ? NBW + 2 : (NBW - 1) * STR + KW, PLANE = PH * PW;
    constexpr int NE = (PW + 1) / 2;                    // STR = 2: even patch columns sit at positions [0, NE), odd ones behind
    constexpr int NX = 4 * PLANE;                       // units per chunk
    constexpr int XSW = (NX + 63) / 64;                 // DMA instructions per chunk
    constexpr int KX = (XSW + 1) / 2;                   // ... per patch wave (even / odd instructions)
    constexpr int PST = XSW * 64;                       // units per patch buffer (tail lanes land in the slack)
    // 1x1 layers have one tap per chunk: their patches run two chunks ahead through three buffers
    constexpr int NPB = pf_patch_bufs(KH, KW, STR), LA = NPB - 1;
    constexpr int TPS = pf_tps(MB, NPW, WM, WP, KH, KW, STR, TZ);   // taps per weight stage
    static_assert(TAPZ % TPS == 0, "a weight stage is a whole number of taps of one chunk (and of one phase)");
    constexpr int SPC = TAPS / TPS;                        // stages per chunk
    constexpr int TAPW = ROWS * COPT;                      // units per tap of a stage
    constexpr int WST = TPS * TAPW;                        // units per weight stage
    static_assert(UF != 0 || KX <= (STR == 2 ? 20 : kPfXS), "patch too large for two patch waves");
    static_assert(UF == 0 || (UF == 3 && KH == 7 && KW == 1 && STR == 1 && TZ == 1 && NPB == 2), "UF: the 7x1 form of the first layer");
    static_assert(STR == 1 || (STR == 2 && KH == 3 && KW == 3), "stride 2 is the 3x3 Downsample form");
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
#ifdef CDC_TIMELINE
    // development build: cycles of wave 0 per category -- 0 setup (tile decode, DMA offsets), 1 prologue DMA issue, 2 prologue wait +
    // barrier, 3 main loop, 4 epilogue parameters (loads + barrier), 5 epilogue arithmetic, 6 epilogue stores; 7 / 8 start / end stamp
    unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_readcyclecounter();
    const unsigned long long tl_start = tl_last;
#define PFTL(c) do { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[c] += n_ - tl_last; tl_last = n_; } while (0)
#define PFTL_END() do { if (P.tl && threadIdx.x == 0) { unsigned long long *r_ = P.tl + 16 * ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)); \
        for (int c_ = 0; c_ < 7; ++c_) r_[c_] = tl_acc[c_]; r_[7] = tl_start; r_[8] = __builtin_readcyclecounter(); \
        unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); r_[9] = hw_; \
        unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); r_[10] = xcc_; } } while (0)
#else
#define PFTL(c) do { } while (0)
#define PFTL_END() do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WP, wp = wave % WP;
    const bool patch_wave = wave >= NWV;
    const int pwi = wave - NWV;                          // patch wave 0 / 1 issues the even / odd instructions
    const int z = blockIdx.z, cog = blockIdx.y;
    unsigned bid0 = blockIdx.x;
    if (P.xcd_remap) bid0 = (bid0 & 7) * (gridDim.x >> 3) + (bid0 >> 3);
    int bid = (int)bid0;
    const int tx = bid % P.tiles_x;
    bid /= P.tiles_x;
    const int ty = bid % P.tiles_y;
    const int b = bid / P.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * NBW;
    const int S = P.nchunk * SPC;                          // weight stages of the tile
    constexpr int R = pf_ring(MB, NPW, WM, WP, KH, KW, STR, TZ);   // weight ring slots (host: S >= R - 1)
    static_assert(R >= 3, "no room for a weight ring");
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) uint4 *)smem_u);
    const unsigned wl_lds = lds0 + (unsigned)(NPB * PST) * 16u;

    // ---- patch waves: per-lane source offsets of their DMA instructions (constant over chunks) --------------
    const int Hp = P.H + 2, Wp = P.W + 2;
    const int iy0 = oy0 * STR - P.pad_y[z] + 1, ix0 = ox0 * STR - P.pad_x[z] + 1;      // +1: halo origin
    unsigned xoff[KX];
#pragma unroll
    for (int i = 0; i < KX; ++i) {
        int e = (2 * i + (pwi & 1)) * 64 + lane;
        if (e >= NX) e = 0;
        const int q = e / PLANE, rem = e - q * PLANE;    // q = k-half * 2 + plane: the order of the tensor
        const int r = rem / PW, cp = rem - r * PW;
        const int c = STR == 2 ? (cp < NE ? 2 * cp : 2 * (cp - NE) + 1) : cp;
        const int iy = min(max(iy0 + r, 0), Hp - 1), ix = min(max(ix0 + c, 0), Wp - 1);
        xoff[i] = (unsigned)((q * Hp + iy) * Wp + ix) * 16u;
    }
    const long long cg_bytes = (long long)4 * Hp * Wp * 16;               // bytes per 16-channel chunk of a PF tensor
    const char *s0 = reinterpret_cast<const char *>(P.src0) + (size_t)b * P.src0_bs * 16;
    const char *s1 = P.src1 ? reinterpret_cast<const char *>(P.src1) + (size_t)b * P.src1_bs * 16 : nullptr;
    const int c0_chunks = P.C0 >> 4;
    // patch instruction i of this wave is issued at tap `i / PER` of the previous chunk (PER per tap, early taps)
    constexpr int ISSUE_TAPS = TAPS >= 9 ? 6 : (TAPS == 7 ? 5 : (TAPS >= 4 ? 2 : 1));
    constexpr int PER = (KX + ISSUE_TAPS - 1) / ISSUE_TAPS;
    auto issue_patch = [&](int chunk, auto tc) {          // instructions scheduled at tap tc of the chunk before
        constexpr int t = decltype(tc)::value;
        const char *base = chunk < c0_chunks ? s0 + (long long)chunk * cg_bytes : s1 + (long long)(chunk - c0_chunks) * cg_bytes;
        int pb = chunk;                                   // chunk % NPB
        if constexpr (NPB == 2) pb &= 1; else pb %= 3;
        const unsigned dst = lds0 + (unsigned)(pb * PST) * 16u + (unsigned)(pwi & 1) * 1024u;
#pragma unroll
        for (int i = 0; i < KX; ++i)
            if (t < 0 || i / PER == t)
                if (2 * i + (pwi & 1) < XSW) dma16(xoff[i], base, dst + (unsigned)(2 * i) * 1024u);
    };
    // ---- weight waves: instruction jj of a stage covers units [64jj, 64jj+64) = row pk = jj / (COPT/64) -------
    const char *wsrc = reinterpret_cast<const char *>(P.w) + ((size_t)z * P.w_zs + (size_t)cog * COPT) * 16;
    unsigned wvo[NWW], wdo[NWW];                          // per-lane source offset / LDS offset of instruction k
#pragma unroll
    for (int k = 0; k < NWW; ++k) {
        const int jj = min(wave + k * NWV, WI - 1);       // clamped duplicates are harmless
        if constexpr (COPT == 32) {
            wvo[k] = (unsigned)(jj * 64 + lane) * 16u;
        } else {
            const int row = jj / (COPT / 64), seg = jj - row * (COPT / 64);
            wvo[k] = (unsigned)(row * P.COP + seg * 64 + lane) * 16u;
        }
        wdo[k] = __builtin_amdgcn_readfirstlane((unsigned)jj * 1024u);
    }
    const long long w_dt = (long long)P.nchunk * 6 * P.COP * 16;          // next tap, same chunk
    const long long w_dz = (long long)P.w_zs * 16 - (TAPZ - TPS) * w_dt;     // TZ = 4: first tap of the next phase (from the phase's last stage)
    const long long w_dc = (long long)6 * P.COP * 16 - (TAPZ - TPS) * w_dt - (TZ - 1) * (long long)P.w_zs * 16;  // first tap of the next chunk (from the chunk's last stage)
    const char *wptr = wsrc;                              // stage to be issued next (its first tap)
    int tw = 0, sw = 0;                                   // its first tap and ring slot
    auto issue_w = [&]() {
        const unsigned dst = wl_lds + (unsigned)(sw * WST) * 16u;
#pragma unroll
        for (int q = 0; q < TPS; ++q)
#pragma unroll
            for (int k = 0; k < NWW; ++k) dma16(wvo[k], wptr + q * w_dt, dst + (unsigned)(q * TAPW) * 16u + wdo[k]);
        tw += TPS;
        if (tw == TAPS) { tw = 0; wptr += w_dc; }
        else if (TZ > 1 && (tw % TAPZ) == 0) wptr += w_dz;
        else wptr += TPS * w_dt;
        if (++sw == R) sw = 0;
    };

    f32x16 acc[MB][NB], acc2[ACC2 ? MB : 1][ACC2 ? NB : 1];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if constexpr (ACC2) acc2[m][n][r] = 0.f;
            }

    const int half = lane >> 5, j = lane & 31;
    const int pr = 0, pc = j;
    // A: stage[(pl*2 + half)*COPT + wm*MB*32 + m*32 + j];  B: buf[(half*2 + pl)*PLANE + (row + ky)*PW + pc + kx]
    const uint4 *a_base = smem_u + NPB * PST + half * COPT + wm * MB * 32 + j;
    const uint4 *b_base = smem_u + (half * 2) * PLANE + (wp * NPW * STR) * PW + j;

    typedef f16x8 OpsA[NPL][MB];
    typedef f16x8 OpsB[2][NPW];
    // operands of tap t (compile-time: immediates) from patch buffer xb and ring slot wa
    auto fetch = [&](auto tc, const uint4 *xb, const uint4 *wa, OpsA &A, OpsB &Bv) {
        constexpr int t = decltype(tc)::value;
        constexpr int tz = t % TAPZ, ph = t / TAPZ;          // tap inside its phase; phase (py, px) = (ph >> 1, ph & 1)
        constexpr int kx = tz % KW;
        // TZ = 4: phase (py, px) reads input rows y - 1 + py + dy, columns x - 1 + px + dx of the shared patch
        constexpr int koff = TZ == 4 ? (tz / KW + (ph >> 1)) * PW + kx + (ph & 1)
                                     : (tz / KW) * PW + (STR == 2 ? (kx & 1) * NE + kx / 2 : kx);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int m = 0; m < MB; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa[(pl * 2) * COPT + m * 32]);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < NPW; ++n) Bv[pl][n] = __builtin_bit_cast(f16x8, xb[pl * PLANE + n * STR * PW + koff]);
    };
    // a = h + l' 2^-11, w 2^s = WH + WL:  acc += WL.h + WH.h,  acc2 += WH.l'  (result = acc + acc2 2^-11).  The second
    // accumulator set replaces a third weight plane WH 2^-11: a third fewer A-operand bytes through LDS-DMA, the
    // ring and ds_read (the LDS read rate is the co-limiter of this loop), and no fp16 underflow of small weights.
    auto mma = [&](const OpsA &A, const OpsB &Bv, auto phc) {     // phc: phase of the tap (TZ = 4), selects the accumulator blocks
        constexpr int q0 = decltype(phc)::value * NPW;
        if constexpr (ACC2) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NPW; ++n) acc[m][q0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][m], Bv[0][n], acc[m][q0 + n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NPW; ++n) acc2[m][q0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], Bv[1][n], acc2[m][q0 + n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NPW; ++n) acc[m][q0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], Bv[0][n], acc[m][q0 + n], 0, 0, 0);
        } else {                                          // smallest terms first: WL.h, WH2.l', WH.h
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                constexpr int PA[3] = {1, 2, 0};
                constexpr int PB[3] = {0, 1, 0};
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NPW; ++n)
                        acc[m][q0 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[term] % NPL][m], Bv[PB[term]][n], acc[m][q0 + n], 0, 0, 0);
            }
        }
    };

    // ---- prologue: patches of the first LA chunks, weight stages 0 .. R-2 -------------------------------------
    PFTL(0);
    if constexpr (UF != 0) {
        // weight stages first: they fly while the workgroup builds its two patch buffers from the image
        if (!patch_wave)
            for (int q = 0; q < R - 1 && q < S; ++q) issue_w();
        constexpr int XW = NBW + 6;                      // staged image columns: the tile's 32 and three on either side
        float *xs = reinterpret_cast<float *>(smem_u + NPB * PST + R * WST);      // [UF][PH][XW], behind the ring
        {   // (all loads first, then the LDS writes: one exposed latency instead of one per element)
            constexpr int NS = (UF * PH * XW + NT - 1) / NT;
            float sv[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = tid + k * NT;
                const int ci = i / (PH * XW), rem = i - ci * (PH * XW), r = rem / XW, cc = rem - r * XW;
                const int iy = oy0 - 3 + r, ix = ox0 - 3 + cc;
                sv[k] = (i < UF * PH * XW && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) ? P.x0[(size_t)b * P.x0_bs + ((size_t)ci * P.H + iy) * P.W + ix] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NS; ++k)
                if (tid + k * NT < UF * PH * XW) xs[tid + k * NT] = sv[k];
        }
        __syncthreads();
        // one thread = one pixel of the patch: its 7 x UF image values once, then the four units (chunk, k-half) of both planes
        for (int px = tid; px < PLANE; px += NT) {
            const int r = px / PW, c = px - r * PW;
            _Float16 hq[32], lq[32];
#pragma unroll
            for (int ch = 0; ch < 32; ++ch) {                  // unfolded channel kx * UF + ci = image channel ci at column x + kx - 3
                if (ch < 7 * UF) split2h(xs[((ch % UF) * PH + r) * XW + c + ch / UF], hq[ch], lq[ch]);
                else { hq[ch] = (_Float16)0.f; lq[ch] = (_Float16)0.f; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                      // q = chunk * 2 + k-half: channels 8 q .. 8 q + 7
                f16x8 hv, lv;
#pragma unroll
                for (int t = 0; t < 8; ++t) { hv[t] = hq[8 * q + t]; lv[t] = lq[8 * q + t]; }
                uint4 *pbuf = smem_u + (q >> 1) * PST + ((q & 1) * 2) * PLANE + px;
                pbuf[0] = __builtin_bit_cast(uint4, hv);
                pbuf[PLANE] = __builtin_bit_cast(uint4, lv);
            }
        }
    } else if (patch_wave) {
        for (int c = 0; c < LA && c < P.nchunk; ++c) issue_patch(c, std::integral_constant<int, -1>{});
    } else {
        for (int q = 0; q < R - 1 && q < S; ++q) issue_w();
    }
    PFTL(1);
    dma_wait();
    __builtin_amdgcn_s_barrier();
    PFTL(2);
    OpsA A0, A1;
    OpsB B0, B1;
    fetch(std::integral_constant<int, 0>{}, b_base, a_base, A0, B0);
    int rem = S - 1;                                      // weight stages after the one being multiplied
    int sn = 0;                                           // ring slot of the stage being fetched from
    // One tap.  TAIL = false: a tap of any chunk but the last -- every condition of the pipeline holds (the host
    // guarantees TAPS >= ring - 1 or handles short tiles through the tail variant), so the body is straight-line
    // code apart from the wave-role branch.  TAIL = true: the last chunk, with the end-of-tile conditions.
    // A weight stage holds TPS taps: its last tap does the ring work (wait, barrier, slot recycling); the others only fetch the
    // next tap's operands from the same slot and patch buffer.
    auto tap = [&](auto tc, auto tailc, int chunk, const uint4 *xb_cur, const uint4 *xb_nxt, OpsA &Ac, OpsB &Bc, OpsA &An,
                   OpsB &Bn) {
        constexpr int t = decltype(tc)::value;
        constexpr bool TAIL = decltype(tailc)::value;
        constexpr bool SEND = (t % TPS) == TPS - 1;       // last tap of its stage
        if constexpr (!SEND) {
            if (!(CDC_PF_ABLATE && (P.dbg & 64)))
                fetch(std::integral_constant<int, t + 1>{}, xb_cur, a_base + sn * WST + ((t + 1) % TPS) * TAPW, An, Bn);
            if (patch_wave) {
                if constexpr (t < ISSUE_TAPS)
                    if (UF == 0 && chunk + LA < P.nchunk && !(CDC_PF_ABLATE && (P.dbg & 2))) issue_patch(chunk + LA, tc);
            }
        } else {
        if (!(TAIL && t == TAPS - 1) || rem > 0) {        // (the very last tap has nothing left to fetch)
            if (++sn == R) sn = 0;
            // W(s+1) (and, at a chunk seam, the patch of the next chunk) must have landed before anyone reads it
            if (CDC_PF_ABLATE && (P.dbg & 8)) {
            } else if (patch_wave) {
                if constexpr (t == TAPS - 1) {
                    // two chunks ahead with several taps per chunk: the instructions issued during THIS chunk (patch of chunk
                    // c + 2, at least XSW / 2 per patch wave) stay in flight, everything older -- the patch of chunk c + 1 -- has landed
                    if (LA == 2 && TAPS > 1 && chunk + LA < P.nchunk) vm_wait<XSW / 2>();
                    else dma_wait();
                }
            } else if (!TAIL || rem >= R - 2) {
                // newer than W(s+1) in this wave's queue: W(s+2) .. W(s+R-2) = (R-3) stages
                vm_wait<(R - 3 - (R >= 5 ? CDC_DMA_WAIT_MARGIN : 0)) * NWW * TPS>();
            } else {
                dma_wait();                               // tail of the tile: everything in flight is needed next
            }
            if (!(CDC_PF_ABLATE && (P.dbg & 4))) __builtin_amdgcn_s_barrier();
            const uint4 *wa = a_base + sn * WST;
            if (CDC_PF_ABLATE && (P.dbg & 64)) {
            } else if constexpr (t == TAPS - 1) fetch(std::integral_constant<int, 0>{}, xb_nxt, wa, An, Bn);
            else fetch(std::integral_constant<int, t + 1>{}, xb_cur, wa, An, Bn);
            if (patch_wave) {
                if constexpr (t < ISSUE_TAPS)
                    if (UF == 0 && chunk + LA < P.nchunk && !(CDC_PF_ABLATE && (P.dbg & 2))) issue_patch(chunk + LA, tc);
            } else if ((!TAIL || rem >= R - 1) && !(CDC_PF_ABLATE && (P.dbg & 1))) {
                issue_w();                                // slot (s-1) % R: its readers passed the barrier above
            }
        }
        --rem;
        }
        __builtin_amdgcn_s_setprio(2);
        if (!(CDC_PF_ABLATE && (P.dbg & 32))) mma(Ac, Bc, std::integral_constant<int, t / TAPZ>{});
        __builtin_amdgcn_s_setprio(0);
    };
    int pbc = 0;                                          // patch buffer of the current chunk
    auto chunk_body = [&](auto parc, auto tailc, int chunk) {   // parc: operand-set parity of this chunk's first tap
        constexpr int par0 = decltype(parc)::value;
        int pbn = pbc + 1;
        if (pbn == NPB) pbn = 0;
        const uint4 *xb_cur = b_base + pbc * PST, *xb_nxt = b_base + pbn * PST;
        static_for<TAPS>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (((par0 + t) & 1) == 0) tap(tc, tailc, chunk, xb_cur, xb_nxt, A0, B0, A1, B1);
            else tap(tc, tailc, chunk, xb_cur, xb_nxt, A1, B1, A0, B0);
        });
        pbc = pbn;
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // chunks whose taps all satisfy rem >= ring - 1 run the straight-line variant
    const int n_main = max(0, min(P.nchunk - 1, (S - R + 1) / SPC));
    if constexpr ((TAPS & 1) == 0) {
        int chunk = 0;
        for (; chunk < n_main; ++chunk) chunk_body(P0{}, std::false_type{}, chunk);
        for (; chunk < P.nchunk; ++chunk) chunk_body(P0{}, std::true_type{}, chunk);
    } else {
        int chunk = 0;
        for (; chunk + 1 < n_main; chunk += 2) {
            chunk_body(P0{}, std::false_type{}, chunk);
            chunk_body(P1{}, std::false_type{}, chunk + 1);
        }
        for (; chunk + 1 < P.nchunk; chunk += 2) {
            chunk_body(P0{}, std::true_type{}, chunk);
            chunk_body(P1{}, std::true_type{}, chunk + 1);
        }
        if (chunk < P.nchunk) chunk_body(P0{}, std::true_type{}, chunk);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_barrier();                         // every wave is done with the operand buffers
    PFTL(3);
    if (CDC_PF_ABLATE && (P.dbg & 256)) { PFTL_END(); return; }
    float *ep = reinterpret_cast<float *>(smem_u);        // [4][COPT]: bias, ln g, ln b, shift   + reduction scratch
    float *red = ep + 4 * COPT;                           // [2][WM][WP*NB*32]
    for (int i = tid; i < COPT; i += NT) {
        const int co = cog * COPT + i;
        const bool ok = co < P.Cout;
        ep[i] = (ok && P.bias) ? P.bias[co] : 0.f;
        ep[COPT + i] = (ok && P.ep_g) ? P.ep_g[co] : 0.f;
        ep[2 * COPT + i] = (ok && P.ep_b) ? P.ep_b[co] : 0.f;
        ep[3 * COPT + i] = (ok && P.shift) ? P.shift[(size_t)b * P.shift_bs + co] : 0.f;
    }
    __syncthreads();
    PFTL(4);
    const float inv_c = 1.0f / (float)P.Cout;
    const int cobase = cog * COPT + wm * MB * 32;         // first channel of this wave
    const float *epl = ep + wm * MB * 32 + 4 * half;
    // host guarantees Cout % (MB*32) == 0 per wave part -- except for the COPT = 32 shape (plain fp32 output only), whose
    // stores are masked per channel
    const bool ch_ok = cobase + MB * 32 <= P.Cout || COPT == 32;
    const int nvalid = P.Cout - cobase;
    // block n of the wave: pixel row (n % NPW) of its stack, phase n / NPW (TZ = 4; otherwise the workgroup's z)
    if constexpr (TZ == 4) {
        // Fused phases: 128 accumulator registers per lane, and in the general epilogue below the compiler spilled 170 - 230 more to scratch
        // (all of a lane's per-channel parameters live beside them) -- scratch lines the streaming stores then evicted to HBM: 1.5x the
        // output bytes written, 1.6x the input bytes read (profiles/pmc_r05_path.json: 801 MB against 537 for the 64-channel Upsample).
        // What an Upsample needs (bias, the final LayerNorm where one wave holds all channels, stores) one output-row pair at a time:
        // the two horizontally adjacent phase blocks (px = 0, 1) of a lane are finished and stored before the next pair is touched.
        if (!P.pre_add && !P.resid && !P.resid_pf && !P.res3_w && !P.relu && !P.shift && !P.stat_mean && (WM == 1 || !P.ep_g)) {
            static_assert(!ACC2, "fused phases: one accumulator set");
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int n = 0; n < NPW; ++n) {
                    const int q0 = (py * 2) * NPW + n, q1 = q0 + NPW;
                    const int oy = oy0 + (wp * NPW + n) * NBH + pr, ox = ox0 + pc;
                    const bool valid = (oy < P.Ho) && (ox < P.Wo) && ch_ok;
                    // (the parameter pointer is opaque per pair: the compiler otherwise merges the four pairs' identical LDS reads and keeps
                    //  all 96 parameter values live across them -- the spills again)
                    int eo = 0;
                    asm volatile("" : "+v"(eo));          // (an opaque zero offset: the pointer itself would lose its LDS address space -- flat loads)
                    const float *eq = epl + eo;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float bi = eq[m * 32 + (r & 3) + 8 * (r >> 2)];
                            acc[m][q0][r] = acc[m][q0][r] * P.acc_scale + bi;
                            acc[m][q1][r] = acc[m][q1][r] * P.acc_scale + bi;
                        }
                    if constexpr (WM == 1) {
                        if (P.ep_g) {                   // channel LayerNorm: the wave holds every channel of its pixels (lanes l, l + 32)
                            float mean[2], rinv[2];
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int q = i ? q1 : q0;
                                float sm = 0.f;
#pragma unroll
                                for (int m = 0; m < MB; ++m)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) sm += acc[m][q][r];
                                sm += __shfl_xor(sm, 32);
                                mean[i] = sm * inv_c;
                                float sq = 0.f;
#pragma unroll
                                for (int m = 0; m < MB; ++m)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) { const float d = acc[m][q][r] - mean[i]; sq += d * d; }
                                sq += __shfl_xor(sq, 32);
                                if (P.fault && !(sq < 3.0e38f)) *P.fault = 1;       // range guard, as chan_stats below
                                rinv[i] = 1.0f / sqrtf(sq * inv_c + P.eps);
                            }
#pragma unroll
                            for (int m = 0; m < MB; ++m)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                                    const float gi = eq[COPT + ci], bi = eq[2 * COPT + ci];
                                    acc[m][q0][r] = (acc[m][q0][r] - mean[0]) * rinv[0] * gi + bi;
                                    acc[m][q1][r] = (acc[m][q1][r] - mean[1]) * rinv[1] * gi + bi;
                                }
                        }
                    }
                    if (!P.ep_g && P.fault) {
                        float sa = 0.f;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sa += fabsf(acc[m][q0][r]) + fabsf(acc[m][q1][r]);
                        if (!(sa < 3.0e38f)) *P.fault = 1;
                    }
                    if (valid && !(CDC_PF_ABLATE && (P.dbg & 16))) {
                        if (P.out) {
                            float *op = P.out + (size_t)b * P.out_bs + (unsigned)(oy * P.out_ys + ox * P.out_xs + P.out_zoff[py * 2]) + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
                            for (int m = 0; m < MB; ++m)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    *reinterpret_cast<float2 *>(op + (size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs) = make_float2(acc[m][q0][r], acc[m][q1][r]);
                        }
                        if (P.out_pf) {
                            const long long u0 = (long long)b * P.pf_bs + (long long)oy * P.pf_ys + (long long)ox * P.pf_xs + P.pf_zoff[py * 2] + half;
#pragma unroll
                            for (int m = 0; m < MB; ++m)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    _Float16 h0[4], l0[4], h1[4], l1[4];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        split2h(acc[m][q0][g * 4 + i], h0[i], l0[i]);
                                        split2h(acc[m][q1][g * 4 + i], h1[i], l1[i]);
                                    }
                                    const f16x4 hv0 = {h0[0], h0[1], h0[2], h0[3]}, lv0 = {l0[0], l0[1], l0[2], l0[3]};
                                    const f16x4 hv1 = {h1[0], h1[1], h1[2], h1[3]}, lv1 = {l1[0], l1[1], l1[2], l1[3]};
                                    const uint2 a0 = __builtin_bit_cast(uint2, hv0), a1 = __builtin_bit_cast(uint2, hv1);
                                    const uint2 c0 = __builtin_bit_cast(uint2, lv0), c1 = __builtin_bit_cast(uint2, lv1);
                                    // lane l (half 0) keeps its pixel-2x half and receives the partner's; lane l + 32 likewise for pixel 2x + 1
                                    const uint2 sh = half ? a0 : a1, sl = half ? c0 : c1;
                                    uint2 rh, rl;
                                    rh.x = __shfl_xor(sh.x, 32); rh.y = __shfl_xor(sh.y, 32);
                                    rl.x = __shfl_xor(sl.x, 32); rl.y = __shfl_xor(sl.y, 32);
                                    const uint4 uh = half ? make_uint4(rh.x, rh.y, a1.x, a1.y) : make_uint4(a0.x, a0.y, rh.x, rh.y);
                                    const uint4 ul = half ? make_uint4(rl.x, rl.y, c1.x, c1.y) : make_uint4(c0.x, c0.y, rl.x, rl.y);
                                    uint4 *pu = reinterpret_cast<uint4 *>(P.out_pf) + u0 + (long long)((cobase >> 3) + m * 4 + g) * 2 * P.pf_ps;
                                    pu[0] = uh;
                                    pu[P.pf_ps] = ul;
                                }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            PFTL(6);
            PFTL_END();
            return;
        }
    }
    float mean_v[NB], rinv_v[NB];
    bool valid_v[NB];
    unsigned pix_v[NB];                                   // offset inside one image's channel plane set (< 2^31: host)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int oy = oy0 + (wp * NPW + n % NPW) * NBH + pr, ox = ox0 + pc;
        const int zn = TZ == 4 ? n / NPW : z;
        valid_v[n] = (oy < P.Ho) && (ox < P.Wo) && ch_ok;
        pix_v[n] = (unsigned)(oy * P.out_ys + ox * P.out_xs + P.out_zoff[zn]);
        const float *eq = epl;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[m][n][r] = (ACC2 ? acc[m][n][r] + acc2[ACC2 ? m : 0][ACC2 ? n : 0][r] * (1.0f / 2048.0f) : acc[m][n][r]) * P.acc_scale +
                               eq[m * 32 + (r & 3) + 8 * (r >> 2)];
        if (P.pre_add && P.pre_c4 && valid_v[n]) {
            // hoisted partial sums in accumulator order: the lane's 4 consecutive channels of a group are one 16-byte load
            const size_t hw = (size_t)P.Ho * P.Wo;
            const float4 *p4 = reinterpret_cast<const float4 *>(P.pre_add) + ((size_t)b * (P.Cout >> 2) + (cobase >> 2) + half) * hw + (size_t)oy * P.Wo + ox;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = p4[(size_t)(m * 8 + 2 * g) * hw];
                    acc[m][n][4 * g] += v.x; acc[m][n][4 * g + 1] += v.y; acc[m][n][4 * g + 2] += v.z; acc[m][n][4 * g + 3] += v.w;
                }
        } else if (P.pre_add && valid_v[n]) {
            const float *pp = P.pre_add + (size_t)b * P.out_bs + pix_v[n] + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += pp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs];
        }
    }
    // channel statistics of one pixel: in-lane sum over MB*16 values, lane^32, then across the WM channel parts
    auto chan_stats = [&](float *mean_o, float *rinv_o) {
        const int slot = (wp * NB) * 32 + j;              // + n*32
        float part[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) sm += acc[m][n][r];
            sm += __shfl_xor(sm, 32);
            part[n] = sm;
        }
        if constexpr (WM > 1) {
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NB; ++n)
                if (half == 0) red[wm * (WP * NB * 32) + slot + n * 32] = part[n];
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float sm = 0.f;
#pragma unroll
                for (int q = 0; q < WM; ++q) sm += red[q * (WP * NB * 32) + slot + n * 32];
                part[n] = sm;
            }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            mean_o[n] = part[n] * inv_c;
            float sq = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[m][n][r] - mean_o[n]; sq += d * d; }
            sq += __shfl_xor(sq, 32);
            part[n] = sq;
        }
        if constexpr (WM > 1) {
            float *red2 = red + WM * (WP * NB * 32);
#pragma unroll
            for (int n = 0; n < NB; ++n)
                if (half == 0) red2[wm * (WP * NB * 32) + slot + n * 32] = part[n];
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float sq = 0.f;
#pragma unroll
                for (int q = 0; q < WM; ++q) sq += red2[q * (WP * NB * 32) + slot + n * 32];
                part[n] = sq;
            }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            // range guard (ConvArgs::fault): a non-finite accumulator shows in the variance, before LayerNorm + ReLU can hide it
            if (P.fault && !(part[n] < 3.0e38f)) *P.fault = 1;
            rinv_o[n] = 1.0f / sqrtf(part[n] * inv_c + P.eps);
        }
    };
    if (P.ep_g) {
        chan_stats(mean_v, rinv_v);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float *eq = epl;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                    acc[m][n][r] = (acc[m][n][r] - mean_v[n]) * rinv_v[n] * eq[COPT + ci] + eq[2 * COPT + ci];
                }
        }
    } else if (P.fault) {
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += fabsf(acc[m][n][r]);
            if (!(s < 3.0e38f)) *P.fault = 1;
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (P.relu) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = fmaxf(acc[m][n][r], P.relu_slope * acc[m][n][r]);
        }
        if (P.shift) {
            const float *eq = epl;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += eq[3 * COPT + m * 32 + (r & 3) + 8 * (r >> 2)];
        }
        if (P.resid_pf && valid_v[n]) {       // the residual as a PF tensor: this lane's 8-byte half-units, a = h + l' 2^-11
            const int oy = oy0 + (wp * NPW + n % NPW) * NBH + pr, ox = ox0 + pc;
            const uint2 *up = reinterpret_cast<const uint2 *>(reinterpret_cast<const uint4 *>(P.resid_pf) + (long long)b * P.rpf_bs + (long long)(cobase >> 3) * 2 * P.rpf_ps +
                                                              (long long)oy * P.rpf_ys + ox + P.rpf_zoff) + half;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 hu = up[(long long)((m * 4 + g) * 2) * P.rpf_ps * 2], lu = up[(long long)((m * 4 + g) * 2 + 1) * P.rpf_ps * 2];
                    const f16x4 hv = __builtin_bit_cast(f16x4, hu), lv = __builtin_bit_cast(f16x4, lu);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][n][g * 4 + i] += __builtin_fmaf((float)lv[i], 1.0f / 2048.0f, (float)hv[i]);
                }
        } else if (P.resid && valid_v[n]) {
            // (resid1: a residual over a channel concatenation, this wave's channels wholly in one source)
            const float *rp = (P.resid1 && cobase >= P.resid_c0)
                                  ? P.resid1 + (size_t)b * P.resid1_bs + pix_v[n] + (size_t)(cobase - P.resid_c0 + 4 * half) * P.resid_cs
                                  : P.resid + (size_t)b * P.resid_bs + pix_v[n] + (size_t)(cobase + 4 * half) * P.resid_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += rp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.resid_cs];
        }
    }
    if (P.res3_w) {        // res_conv over the 3 image channels of the first ResnetBlock: 3 FMAs per value
        const size_t hw = (size_t)P.Ho * P.Wo;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int oy = oy0 + (wp * NPW + n % NPW) * NBH + pr, ox = ox0 + pc;
            const float *xp = P.res3_x + (size_t)b * P.res3_bs + (size_t)oy * P.Wo + ox;
            const float x0 = valid_v[n] ? xp[0] : 0.f, x1 = valid_v[n] ? xp[hw] : 0.f, x2 = valid_v[n] ? xp[2 * hw] : 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cobase + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    acc[m][n][r] += P.res3_w[co] * x0 + P.res3_w[(size_t)P.COP + co] * x1 + P.res3_w[2 * (size_t)P.COP + co] * x2;
                }
        }
    }
    if (P.stat_mean) {
        chan_stats(mean_v, rinv_v);
#pragma unroll
        for (int n = 0; n < NB; ++n)
            if (valid_v[n] && half == 0 && wm == 0) {
                P.stat_mean[(size_t)b * P.out_cs + pix_v[n]] = mean_v[n];
                P.stat_rstd[(size_t)b * P.out_cs + pix_v[n]] = rinv_v[n];
            }
    }
    PFTL(5);
    if constexpr (TZ == 4) {
        // The phases px = 0 / 1 of a lane are horizontally adjacent output pixels (2x, 2x + 1): stored together, a wave writes whole
        // runs instead of every other element -- fp32 as 8-byte pairs (256-byte runs per channel row), planes as whole 16-byte
        // units after lanes l and l + 32 (channels 0..3 / 4..7 of a group) have exchanged halves: lane l completes the unit of
        // pixel 2x, lane l + 32 the unit of pixel 2x + 1 (1-KiB runs).
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
                const int q0 = (py * 2) * NPW + n, q1 = q0 + NPW;
                if (!valid_v[q0] || (CDC_PF_ABLATE && (P.dbg & 16))) continue;
                if (P.out) {
                    float *op = P.out + (size_t)b * P.out_bs + pix_v[q0] + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            *reinterpret_cast<float2 *>(op + (size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs) = make_float2(acc[m][q0][r], acc[m][q1][r]);
                }
                if (P.out_pf) {
                    const int oy = oy0 + (wp * NPW + n) * NBH + pr, ox = ox0 + pc;
                    const long long u0 = (long long)b * P.pf_bs + (long long)oy * P.pf_ys + (long long)ox * P.pf_xs + P.pf_zoff[py * 2] + half;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            _Float16 h0[4], l0[4], h1[4], l1[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                split2h(acc[m][q0][g * 4 + i], h0[i], l0[i]);
                                split2h(acc[m][q1][g * 4 + i], h1[i], l1[i]);
                            }
                            const f16x4 hv0 = {h0[0], h0[1], h0[2], h0[3]}, lv0 = {l0[0], l0[1], l0[2], l0[3]};
                            const f16x4 hv1 = {h1[0], h1[1], h1[2], h1[3]}, lv1 = {l1[0], l1[1], l1[2], l1[3]};
                            const uint2 a0 = __builtin_bit_cast(uint2, hv0), a1 = __builtin_bit_cast(uint2, hv1);
                            const uint2 c0 = __builtin_bit_cast(uint2, lv0), c1 = __builtin_bit_cast(uint2, lv1);
                            // lane l (half 0) keeps its pixel-2x half and receives the partner's; lane l + 32 likewise for pixel 2x + 1
                            const uint2 sh = half ? a0 : a1, sl = half ? c0 : c1;
                            uint2 rh, rl;
                            rh.x = __shfl_xor(sh.x, 32); rh.y = __shfl_xor(sh.y, 32);
                            rl.x = __shfl_xor(sl.x, 32); rl.y = __shfl_xor(sl.y, 32);
                            const uint4 uh = half ? make_uint4(rh.x, rh.y, a1.x, a1.y) : make_uint4(a0.x, a0.y, rh.x, rh.y);
                            const uint4 ul = half ? make_uint4(rl.x, rl.y, c1.x, c1.y) : make_uint4(c0.x, c0.y, rl.x, rl.y);
                            uint4 *pu = reinterpret_cast<uint4 *>(P.out_pf) + u0 + (long long)((cobase >> 3) + m * 4 + g) * 2 * P.pf_ps;
                            pu[0] = uh;
                            pu[P.pf_ps] = ul;
                        }
                }
            }
        PFTL(6);
        PFTL_END();
        return;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (!valid_v[n] || (CDC_PF_ABLATE && (P.dbg & 16))) continue;
        if (P.out) {
            float *op = P.out + (size_t)b * P.out_bs + pix_v[n] + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                    if (COPT != 32 || ci + 4 * half < nvalid) op[(size_t)ci * P.out_cs] = acc[m][n][r];
                }
        }
        if (P.out_pf) {
            const int oy = oy0 + (wp * NPW + n % NPW) * NBH + pr, ox = ox0 + pc;
            const long long u0 = (long long)b * P.pf_bs + (long long)oy * P.pf_ys + (long long)ox * P.pf_xs + P.pf_zoff[TZ == 4 ? n / NPW : z];
#pragma unroll
            for (int m = 0; m < MB; ++m)
                pf_store_block(reinterpret_cast<uint4 *>(P.out_pf), u0 + (long long)((cobase >> 3) + m * 4) * 2 * P.pf_ps, P.pf_ps, half, acc[m][n]);
        }
    }
    PFTL(6);
    PFTL_END();
}

typedef void (*pf_kernel_fn)(const PfArgs);

}  // namespace cdc
