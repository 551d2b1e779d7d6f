// conv_pw_kernel.h -- 1x1 (pointwise) convolution in fp16x2 arithmetic with a barrier-free activation path.
//
// For a pointwise convolution the MFMA B operand of a lane (pixel j, k-half) is 8 channels of ITS OWN pixel.  Every wave
// stages its own activations: 4-byte LDS-DMA, one instruction per channel row and 32-pixel block (lanes 0-31 / 32-63 =
// the two k-halves, 128-byte row segments of the fp32 NCHW tensor), into a private LDS area PD steps ahead; the lane
// reads its 8 values back, subtracts the PreNorm mean if there is one and splits into the two fp16 planes in registers.
// No patch, no conversion pass, no ds_write, and no barrier for the activations (a wave only ever reads what it
// requested itself: its own s_waitcnt suffices).  Per step a wave issues L activation pieces, then its NWW weight pieces, and waits for
// ALL of them a step later (vmcnt 0; a counted wait that lets exactly the newer operations fly is in-order exact, but a 16-byte piece it
// has just covered is not always readable in full at once: see the step lambda and DESIGN section 5).
// The weights go through the LDS-DMA ring of conv_pf_kernel.h (every wave issues its share of a stage, one s_barrier
// per 16-channel step), shared by the WM x WP waves; shapes with MB*NPW <= 4 use the planes {WH, WL} and a second
// accumulator set.  Epilogue: scale (2^-s, x rstd of the pixel with PreNorm), bias, hoisted partial sums, ReLU,
// per-image shift, residual, fp32 store and / or PF planes.
//
// It replaces conv_split2_kernel for 1x1 layers at the >= 32-pixel-wide levels, where that kernel's per-chunk cost (two
// barriers, conversion through LDS, loads one chunk ahead) bounds the 18 MFMAs of a chunk (256 -> 768 @32^2 batch 32:
// 0.098 ms there).
#pragma once
#include "conv_pf_kernel.h"

namespace cdc {

constexpr int kPwPD = 2;                            // activation stages in flight per wave
// X16 (round 4): the activations of a 16-channel chunk and 32-pixel block arrive as TWO 16-byte LDS-DMA instructions (lane = 4 pixels
// of one of 8 channel rows: 1 KiB each, the k-halves 32 floats apart so that the two halves of a wave read different banks) instead of
// eight 4-byte ones -- the wave's LDS-DMA issue was what bounded the kernel.  Needs rows of 4-pixel granularity (W % 4 == 0).
constexpr int kPwX16N = 2 * 256 + 32;               // floats per block and stage in the X16 layout
__host__ __device__ constexpr size_t pw_x_bytes(int NPW, int WM, int WP, bool X16 = false) {
    return (size_t)WM * WP * kPwPD * NPW * (X16 ? kPwX16N * 4 : 8 * 256);
}
__host__ __device__ constexpr int pw_ring(int MB, int NPW, int WM, int WP, bool X16 = false) {
    const size_t budget = (WM * WP == 8 ? 150 * 1024 : 78 * 1024) - pw_x_bytes(NPW, WM, WP, X16);
    const size_t wst = (size_t)pf_rows(MB, NPW) * WM * MB * 32 * 16;
    const int r = (int)(budget / wst);
    return r > 6 ? 6 : r;
}

__device__ __forceinline__ void dma4(unsigned voff, const void *sbase, unsigned lds_byte) {   // 4 bytes per lane
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte)
                 : "memory");
}

template <int MB, int NPW, int WM, int WP, bool X16 = false>
__global__ void __launch_bounds__(64 * WM * WP, WM * WP == 8 ? 1 : 2) conv_pw_kernel(const PfArgs P) {
    constexpr int NW = WM * WP, NT = 64 * NW, COPT = WM * MB * 32;
    static_assert(COPT % 64 == 0, "a weight DMA instruction (64 units) must stay inside one (plane, k-half) row");
    constexpr bool ACC2 = pf_acc2(MB, NPW);
    constexpr int NPL = ACC2 ? 2 : 3, ROWS = 2 * NPL;
    constexpr int WI = ROWS * COPT / 64;                // DMA instructions per weight stage
    constexpr int NWW = (WI + NW - 1) / NW;             // ... per wave (every wave issues)
    constexpr int WST = ROWS * COPT;                    // units per weight stage
    constexpr int R = pw_ring(MB, NPW, WM, WP, X16);
    static_assert(R >= 5, "no room for the weight ring");
    constexpr int TH = WP * NPW;
    constexpr int PD = kPwPD;                           // activation stages in flight
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
#ifdef CDC_TIMELINE
    unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_readcyclecounter();      // categories as in conv_pf_kernel
    const unsigned long long tl_start = tl_last;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WP, wp = wave % WP;
    const int cog = blockIdx.y;
    unsigned bid0 = blockIdx.x;
    if (P.xcd_remap) bid0 = (bid0 & 7) * (gridDim.x >> 3) + (bid0 >> 3);
    int bid = (int)bid0;
    const int tile = bid;                               // (linear mode: TH consecutive 32-pixel blocks of the batch)
    const int tx = bid % P.tiles_x;
    bid /= P.tiles_x;
    const int ty = bid % P.tiles_y;
    const int b = P.lin ? 0 : bid / P.tiles_y;          // (linear mode: the image is a property of the block, bimg[])
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int S = P.nchunk;                             // steps = 16-channel chunks
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) uint4 *)smem_u);

    // ---- weights: instruction jj of a stage covers units [64jj, 64jj+64) = row jj / (COPT/64) of {plane, k-half} ----
    const char *wsrc = reinterpret_cast<const char *>(P.w) + ((size_t)b * P.w_bs + (size_t)cog * COPT) * 16;
    unsigned wvo[NWW], wdo[NWW];
#pragma unroll
    for (int k = 0; k < NWW; ++k) {
        const int jj = min(wave + k * NW, WI - 1);        // clamped duplicates are harmless
        const int row = jj / (COPT / 64), seg = jj - row * (COPT / 64);
        wvo[k] = (unsigned)(row * P.COP + seg * 64 + lane) * 16u;
        wdo[k] = __builtin_amdgcn_readfirstlane((unsigned)jj * 1024u);
    }
    const long long w_dc = (long long)6 * P.COP * 16;     // next chunk (the stored layout keeps three planes)
    const char *wptr = wsrc;
    int sw = 0;
    auto issue_w = [&]() {
        const unsigned dst = lds0 + (unsigned)(sw * WST) * 16u;
#pragma unroll
        for (int k = 0; k < NWW; ++k) dma16(wvo[k], wptr, dst + wdo[k]);
        wptr += w_dc;
        if (++sw == R) sw = 0;
    };

    // ---- activations: lane (pixel j of its row block, k-half) requests channels 16c + 8 half + i of that pixel -------
    const int half = lane >> 5, j = lane & 31;
    const int HW = P.H * P.W;
    unsigned xvo[NPW];                                  // byte offset of the lane's (k-half, pixel) inside a 16-channel chunk
                                                        // (X16: of the lane's (channel row lane >> 3, pixel quad lane & 7) inside a k-half)
    float mu[NPW];
    bool valid[NPW];
    unsigned pixo[NPW];
    int bimg[NPW];                                      // image of the block (wave-uniform)
    const int bpi = HW >> 5;                            // linear mode: blocks per image
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        if (P.lin) {
            const int gb = tile * TH + wp * NPW + n, nb = P.B * bpi;
            valid[n] = gb < nb;
            const int gc = min(gb, nb - 1);
            bimg[n] = __builtin_amdgcn_readfirstlane(gc / bpi);
            pixo[n] = (unsigned)((gc - bimg[n] * bpi) * 32 + j);
        } else {
            const int oy = oy0 + wp * NPW + n, ox = ox0 + j;
            valid[n] = oy < P.Ho && ox < P.Wo;
            bimg[n] = b;
            pixo[n] = (unsigned)(min(oy, P.Ho - 1) * P.W + min(ox, P.Wo - 1));
        }
        xvo[n] = ((unsigned)(8 * half) * (unsigned)HW + pixo[n]) * 4u;
        if constexpr (X16) {
            // quad q of the block's 32 pixels (clamped into the row: Wo % 4 == 0), channel row r of the k-half
            const int r = lane >> 3, q = lane & 7;
            unsigned p4;
            if (P.lin) p4 = pixo[n] - (unsigned)j + 4u * q;
            else {
                const int oy = oy0 + wp * NPW + n;
                p4 = (unsigned)(min(oy, P.Ho - 1) * P.W + min(ox0 + 4 * q, P.Wo - 4));
            }
            xvo[n] = ((unsigned)r * (unsigned)HW + p4) * 4u;
        }
        mu[n] = P.pre_mean ? P.pre_mean[(size_t)bimg[n] * HW + pixo[n]] : 0.f;
    }
    const int c0_chunks = P.C0 >> 4;
    constexpr int L = (X16 ? 2 : 8) * NPW;              // activation pieces per step and wave
    constexpr int XST = X16 ? NPW * kPwX16N : NPW * 8 * 64;   // floats per activation stage of one wave
    float *xw = reinterpret_cast<float *>(smem_u + R * WST) + wave * (PD * XST);    // this wave's private stages
    const unsigned xw_lds = lds0 + (unsigned)(R * WST) * 16u + (unsigned)(wave * (PD * XST)) * 4u;
    auto issue_x = [&](int c, int slot) {               // stage layout [n][i][lane]
        const unsigned dst = xw_lds + (unsigned)(slot * XST) * 4u;
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const float *src = c < c0_chunks ? P.x0 + (size_t)bimg[n] * P.x0_bs + (size_t)c * 16 * HW
                                             : P.x1 + (size_t)bimg[n] * P.x1_bs + (size_t)(c - c0_chunks) * 16 * HW;
            const char *base = reinterpret_cast<const char *>(uniform_ptr(src));
            if constexpr (X16) {        // k-half h: channels 8 h .. 8 h + 7 as one instruction, [row][32 pixels], 32 floats behind half 0's
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) dma16(xvo[n], base + (size_t)(8 * hh) * HW * 4, dst + (unsigned)(n * kPwX16N + hh * (256 + 32)) * 4u);
            } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dma4(xvo[n], base + (size_t)i * HW * 4, dst + (unsigned)((n * 8 + i) * 64) * 4u);
            }
        }
    };

    f32x16 acc[MB][NPW], acc2[ACC2 ? MB : 1][ACC2 ? NPW : 1];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NPW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if constexpr (ACC2) acc2[m][n][r] = 0.f;
            }
    const uint4 *a_base = smem_u + half * COPT + wm * MB * 32 + j;
    typedef f16x8 OpsA[NPL][MB];
    typedef f16x8 OpsB[2][NPW];
    auto fetch_a = [&](const uint4 *wa, OpsA &A) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int m = 0; m < MB; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa[(pl * 2) * COPT + m * 32]);
    };
    auto split_b = [&](int slot, OpsB &Bv) {           // the wave's own stage -> the two fp16 planes
        const float *src = xw + slot * XST + (X16 ? half * (256 + 32) + j : lane);
#pragma unroll
        for (int n = 0; n < NPW; ++n)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 hq, lq;
                split2h((X16 ? src[n * kPwX16N + i * 32] : src[(n * 8 + i) * 64]) - mu[n], hq, lq);
                Bv[0][n][i] = hq; Bv[1][n][i] = lq;
            }
    };
    auto mma = [&](const OpsA &A, const OpsB &Bv) {
        if constexpr (ACC2) {
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
        } else {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                constexpr int PA[3] = {1, 2, 0};
                constexpr int PB[3] = {0, 1, 0};
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NPW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[term] % NPL][m], Bv[PB[term]][n], acc[m][n], 0, 0, 0);
            }
        }
    };

    // ---- prologue: activations of steps 0 .. PD-1, weight stages 0 .. R-2 ----------------------------------------
    PFTL(0);
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < S) issue_x(d, d);
    for (int q = 0; q < R - 1 && q < S; ++q) issue_w();
    PFTL(1);
    dma_wait();                                         // prologue: everything, once
    __builtin_amdgcn_s_barrier();
    PFTL(2);
    OpsA A0, A1;
    OpsB B0, B1;
    fetch_a(a_base, A0);
    split_b(0, B0);
    if (PD < S) issue_x(PD, 0);                         // (after the reads of slot 0: same wave, program order)
    int sn = 0;
    // One step: operands of step s in (Ac, Bc).  Order of a wave's VM queue per step: [x pieces of step s+1+PD][weight
    // pieces of stage s+R-1].  Needed now: the x pieces of step s+1 (requested at step s-PD... i.e. PD steps ago) and
    // W(s+1).  Newer than the former: that step's weight pieces + (PD-1) full steps = NWW + (PD-1)(L+NWW); W(s+1) is
    // older still (R >= PD+2).  In the tail (something was not issued) wait for everything.
    auto step = [&](int s, auto slotc, OpsA &Ac, OpsB &Bc, OpsA &An, OpsB &Bn) {
        constexpr int slot = decltype(slotc)::value;    // (s + 1) % PD, compile-time: PD == 2
        if (s + 1 < S) {
            if (++sn == R) sn = 0;
            // Every step waits for ALL of the wave's LDS-DMA (vmcnt 0), not for a counted prefix (round 5).  The counted wait -- vmcnt <=
            // NWW + (PD - 1)(L + NWW): exactly the operations newer than the activation pieces of step s + 1 -- returned stale values in the
            // LAST-issued 16-byte activation pieces of a step about once in 10 000 launches on some boxes (the k/v projection of the 64x64
            // level, 192 -> 384, three channel groups).  Waiting for everything costs nothing measurable here (the step is bound by the matrix
            // pipe and the barrier, not by the loads two steps ahead).
            // Round 6 (profiles/determinism_r06.txt): the layer alone reproduces it (tools/op_stress.py: 61 - 116 of 400 000
            // executions differ with the counted wait, 0 with vmcnt(0)); a wave's operations DO leave vmcnt in issue order (tools/ubench/
            // dma_order.hip), but with global_load_lds_dwordx4 the counter can reach N a short time before the last-issued needed piece is
            // readable in full.  CDC_PW_DBG (development) selects the arms of that localisation: 1024 = the counted wait, + 2048 = with a margin
            // of the NWW weight pieces that follow the needed pieces in the queue (3 events), + 4096 = followed by ~256 idle cycles (4 events);
            // with 4-byte pieces (CDC_NO_PW_X16) the counted wait shows none.
            if (s >= PD && s + R - 1 < S && s + 1 + PD < S && (P.dbg & 1024)) {
                if (P.dbg & 2048) vm_wait<(PD - 1) * (L + NWW)>(); else vm_wait<NWW + (PD - 1) * (L + NWW)>();
                if (P.dbg & 4096) __builtin_amdgcn_s_sleep(4);
            } else dma_wait();
            __builtin_amdgcn_s_barrier();
            fetch_a(a_base + sn * WST, An);
            split_b(slot, Bn);
#ifdef CDC_PW_LATE_PROBE
            // Lab build (tools/build_variant.sh pwprobe -DCDC_PW_LATE_PROBE, CDC_PW_DBG=9216): the same stage read AGAIN right behind the first read's arithmetic, before
            // its slot is requested anew.  A difference = a piece that landed AFTER the counted wait had released the wave.
            if (P.dbg & 8192) {
                if (P.dbg & 16384) __builtin_amdgcn_s_sleep(16);       // (with the pause the steady state shifts and the event itself goes away)
                OpsB Bt;
                split_b(slot, Bt);
                unsigned late = 0;
#pragma unroll
                for (int n = 0; n < NPW; ++n) {
                    bool d = false;
#pragma unroll
                    for (int i = 0; i < 8; ++i) d |= (float)Bn[0][n][i] != (float)Bt[0][n][i] || (float)Bn[1][n][i] != (float)Bt[1][n][i];
                    if (__any(d)) late |= 1u << n;
                }
                if (late && lane == 0)
                    printf("[pw late piece] workgroup (%d,%d) wave %d step %d of %d: pixel-block mask %x differs between the read behind vmcnt(%d) and a second read\n",
                           (int)blockIdx.x, (int)blockIdx.y, wave, s + 1, S, late, NWW + (PD - 1) * (L + NWW));
            }
#endif
            if (s + 1 + PD < S) issue_x(s + 1 + PD, slot);
            if (s + R - 1 < S) issue_w();
        }
        __builtin_amdgcn_s_setprio(2);
        mma(Ac, Bc);
        __builtin_amdgcn_s_setprio(0);
    };
    static_assert(PD == 2 && R >= PD + 3, "two activation stages; W(s+1) must be older than the activation pieces of step s+1");
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    {
        int s = 0;
        for (; s + 1 < S; s += 2) {
            step(s, P1{}, A0, B0, A1, B1);
            step(s + 1, P0{}, A1, B1, A0, B0);
        }
        if (s < S) step(s, P1{}, A0, B0, A1, B1);
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_barrier();                         // every wave is done with the ring
    PFTL(3);
    float *ep = reinterpret_cast<float *>(smem_u);        // [2][COPT]: bias, shift
    for (int i = tid; i < COPT; i += NT) {
        const int co = cog * COPT + i;
        const bool ok = co < P.Cout;
        ep[i] = (ok && P.bias) ? P.bias[co] : 0.f;
        ep[COPT + i] = (ok && P.shift) ? P.shift[(size_t)b * P.shift_bs + co] : 0.f;
    }
    __syncthreads();
    PFTL(4);
    const int cobase = cog * COPT + wm * MB * 32;
    const float *epl = ep + wm * MB * 32 + 4 * half;
    const bool ch_ok = cobase + MB * 32 <= P.Cout;        // host: Cout % (MB*32) == 0 per wave part
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int oy = oy0 + wp * NPW + n, ox = ox0 + j;
        const int bn = bimg[n];
        const bool ok = valid[n] && ch_ok;
        const float sc = P.pre_rstd ? P.acc_scale * P.pre_rstd[(size_t)bn * HW + pixo[n]] : P.acc_scale;
        const size_t opix = P.lin ? (size_t)pixo[n] : (size_t)oy * P.out_ys + (size_t)ox * P.out_xs + P.out_zoff[0];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[m][n][r] = (ACC2 ? acc[m][n][r] + acc2[ACC2 ? m : 0][ACC2 ? n : 0][r] * (1.0f / 2048.0f) : acc[m][n][r]) * sc +
                               epl[m * 32 + (r & 3) + 8 * (r >> 2)];
        if (!ok) continue;
        if (P.fault) {                  // range guard (ConvArgs::fault): non-finite accumulators are reported where they arise
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += fabsf(acc[m][n][r]);
            if (!(s < 3.0e38f)) *P.fault = 1;
        }
        if (P.pre_add) {
            const float *pp = P.pre_add + (size_t)bn * P.out_bs + opix + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += pp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs];
        }
        if (P.relu) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = fmaxf(acc[m][n][r], P.relu_slope * acc[m][n][r]);
        }
        if (P.shift) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += epl[COPT + m * 32 + (r & 3) + 8 * (r >> 2)];
        }
        if (P.resid) {
            const float *rp = P.resid + (size_t)bn * P.resid_bs + opix + (size_t)(cobase + 4 * half) * P.resid_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += rp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.resid_cs];
        }
        if (P.out) {
            float *op = P.out + (size_t)bn * P.out_bs + opix + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) op[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs] = acc[m][n][r];
        }
        if (P.out_pf) {
            const long long u0 = (long long)bn * P.pf_bs + (long long)oy * P.pf_ys + (long long)ox * P.pf_xs + P.pf_zoff[0];
#pragma unroll
            for (int m = 0; m < MB; ++m)
                pf_store_block(reinterpret_cast<uint4 *>(P.out_pf), u0 + (long long)((cobase >> 3) + m * 4) * 2 * P.pf_ps, P.pf_ps, half, acc[m][n]);
        }
    }
    PFTL(6);                                              // (development build: the whole per-block epilogue loop counts as "stores")
    PFTL_END();
}

}  // namespace cdc
