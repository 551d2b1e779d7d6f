// conv_ws_kernel.h -- 3x3 / stride-1 / pad-1 convolutions of the FEW-PIXEL levels (maps 8 ... 64 pixels wide: the <= 16^2 trunk of
// the U-Net at batch 32, everything below 128^2 for one image per call; 128 - 768 input channels), weight-stationary in registers (round 5).
//
// What bounded these layers on conv_split2_kernel: 2 048 - 8 192 pixels per batch leave a launch < 3 workgroups per CU,
// so K was sliced over workgroups (4 partial-sum tensors + a LayerNorm pass that adds them), and a 32 x 32 wave tile fetches one
// A and one B operand from LDS per MFMA: 5 ds_read_b128 per 3 MFMAs -- LDS-bandwidth-bound at 12 waves per CU.
//
// Here K is sliced over the WAVES of one workgroup instead, and the weights never touch LDS:
//   * workgroup = (32 output channels) x (NPB 32-pixel blocks = 128 pixels: two 8x8 images or half a 16x16 one) x ALL of K;
//     wave w owns the 16-channel chunks w, w + NW, ... of the input;
//   * for its chunk a wave loads the weight operands (planes WH, WL, WH2; a kernel row at a time, two register sets: the next row's
//     loads fly while this one multiplies) straight from global memory in A-operand order and keeps a row for all NPB pixel blocks: per pixel block and tap only the two activation planes (h, l') come from LDS --
//     2 ds_read_b128 per 3 MFMAs instead of 5: the loop is matrix-bound, not LDS-bound;
//   * the wave converts its own chunk of the input (fp32 NCHW -> two fp16 planes, conv_split_kernel.h AR = 1) into its own LDS
//     patch: no cross-wave synchronisation inside the K loop at all;
//   * the NW partial accumulators meet in LDS once, after the K loop (no partial-sum tensors in HBM, no sum pass); the epilogue
//     (all waves) stores the raw result (bias added).  The channel LayerNorm of a Block (the workgroup owns 32 of the channels) is
//     the in-place pass that follows (ln_kernel_vec over ONE tensor).  "LayerNorm on load" -- the next convolution of this kernel
//     normalising the raw result while it converts it, from per-group (mean, M2) partials of this epilogue -- was built and measured
//     slower than that pass (profiles/ws_lnload_ab_r05.txt) and removed.
// fp32 accumulation; the chunk order of the sum is fixed by (nchunk, NW): deterministic.
#pragma once
#include "conv_pf_kernel.h"

namespace cdc {

struct WsArgs {
    const float *x0, *x1;           // fp32 NCHW sources (channel concatenation; x1 may be null)
    long long x0_bs, x1_bs;         // batch strides in floats
    int C0, Cin;                    // channels taken from x0; total (both multiples of 16)
    int H, B;                       // rows of the OUTPUT map (its width is the template parameter; the input is stride times as large), batch
    const void *w;                  // fp16 planes {WH, WL, WH2} of w 2^s: [tap][Cin/16][3][2][COP] 16-byte units
    int nchunk, cpw;                // 16-channel chunks; chunks per wave (nchunk = cpw * waves)
    int COP, Cout;
    float acc_scale;                // 2^-s
    const float *bias;              // [Cout] or null
    const float *pre_add;           // hoisted partial sums (the step-invariant context half of a concatenated input), layout of `out`, or null
    float *out;                     // raw result, fp32 NCHW
    long long out_bs;
    int tiles, groups;              // pixel tiles; 32-channel groups of Cout (gridDim.x = tiles * groups)
    int xcd_remap;
    int *fault;                     // range guard (ConvArgs::fault)
#ifdef CDC_WS_LAB
    unsigned long long *tl;         // 16 cycle stamps per workgroup (wave 0)
#endif
    int dbg;                        // -DCDC_WS_LAB builds only (CDC_WS_DBG, timing experiments, wrong results): 1 no MFMAs, 8 no reduction / epilogue
};

// LDS bytes of a launch: the waves' patches during the K loop, the partial accumulators after it
// (W, H: the output map; stride 1 or 2)
__host__ __device__ inline int ws_ppix(int W, int H, int NPB, int stride = 1) {
    const int tpx = NPB * 32, hw = H * W;
    return (tpx >= hw ? tpx / hw : 1) * ((tpx >= hw ? H : tpx / W) * stride + 2) * (W * stride + 2);
}
__host__ __device__ inline size_t ws_lds_bytes(int W, int H, int NPB, int waves, int stride = 1) {
    const size_t patches = (size_t)waves * 4 * ws_ppix(W, H, NPB, stride) * 16, red = (size_t)waves * NPB * 4096;
    return patches > red ? patches : red;
}

#ifdef CDC_WS_LAB
#define WS_STAMP(i) do { if (PA.tl && threadIdx.x == 0) PA.tl[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

// (stride 2: at most four waves per workgroup -- one per SIMD, the whole register file: its loader keeps 9 passes of input values in flight)
template <int W_, int NPB, int STR = 1>
__global__ void __launch_bounds__(STR == 2 ? 256 : 512) conv_ws_kernel(const WsArgs PA) {
    static_assert(W_ == 8 || W_ == 16 || W_ == 32 || W_ == 64, "map width");
    static_assert(NPB % 2 == 0, "pixel blocks are multiplied in pairs");
    static_assert(STR == 1 || STR == 2, "stride");
    WS_STAMP(0);
    // (the fields the kernel uses, as locals: the lambdas below capture THESE -- capturing the argument block itself made hipcc copy it to scratch)
    const auto a_x0 = PA.x0;
    const auto a_x1 = PA.x1;
    const auto a_x0_bs = PA.x0_bs;
    const auto a_x1_bs = PA.x1_bs;
    const auto a_C0 = PA.C0;
    const auto a_H = PA.H;
    const auto a_w = PA.w;
    const auto a_nchunk = PA.nchunk;
    const auto a_cpw = PA.cpw;
    const auto a_COP = PA.COP;
    const auto a_acc_scale = PA.acc_scale;
    const auto a_bias = PA.bias;
    const auto a_pre_add = PA.pre_add;
    const auto a_out = PA.out;
    const auto a_out_bs = PA.out_bs;
    const auto a_tiles = PA.tiles;
    const auto a_xcd_remap = PA.xcd_remap;
    const auto a_fault = PA.fault;
    const int a_dbg = PA.dbg;
    // W_ / a_H: the OUTPUT map; the input is WI wide (STR = 2: Downsample, 3x3 / stride 2 / pad 1 -- a tile of R output rows reads input rows
    // 2 y0 - 1 ... 2 (y0 + R) - 1 and columns -1 ... WI - 1)
    constexpr int WI = W_ * STR, PW = WI + 2, TPX = NPB * 32;
    constexpr int MAXLOAD = STR == 1 ? TPX + 2 * W_ : TPX * 4 + WI;    // most input pixels a tile loads (a band of rows + its neighbour rows)
    constexpr int NIT = (2 * MAXLOAD + 63) / 64;                // loader passes: an item = 8 channels (one k-half) of one pixel
    extern __shared__ __attribute__((aligned(16))) uint4 ws_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int n = lane & 31, kg = lane >> 5;
    const int H = a_H, HW = H * W_;                            // output rows / pixels per image
    const int HI = H * STR, HWI = HI * WI;                      // input rows / pixels per image
    // ---- this workgroup: channel group g, pixel tile (whole images, or R rows of one image) ------------------------------------
    int slot = blockIdx.x;
    if (a_xcd_remap) slot = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);     // one contiguous (group-major) band per XCD: its L2 holds ~groups / 8 weight slices
    const int g = slot / a_tiles, tile = slot - g * a_tiles;
    const bool whole = TPX >= HW;
    const int n_img = whole ? TPX / HW : 1, R = whole ? H : TPX / W_, PR = R * STR + 2, PPIX = n_img * PR * PW;
    const int parts = whole ? 1 : HW / TPX;
    const int b0 = whole ? tile * n_img : tile / parts, y0 = whole ? 0 : (tile - b0 * parts) * R;
    // loaded input rows of an image part: whole images rows 0 .. HI-1 (patch rows 1 .. HI); a part also its neighbour rows (patch rows
    // 0 .. R+1; stride 2: 0 .. 2R, there is no row below)
    const int row0 = whole ? 1 : 0, nrow = whole ? HI : (STR == 1 ? PR : PR - 1), nload = n_img * nrow * WI;
    uint4 *patch = ws_smem + (size_t)wave * 4 * PPIX;
    // ---- loader items of this lane (the same for every chunk): item e = pass * 64 + lane -> (k-half, loaded pixel) ---------------
    // (no runtime divisions in the set-up: a tile holds at most four images -- compare chains -- and W_ is a power of two)
    const int per_img = nrow * WI;
    auto img_of = [](int i, int per) __attribute__((always_inline)) { return (i >= per ? 1 : 0) + (i >= 2 * per ? 1 : 0) + (i >= 3 * per ? 1 : 0); };
    int l_off[NIT];                            // float offset inside a channel plane (+ image stride), -1: nothing to load
    int l_tab[NIT];                            // patch unit incl. the k-half (12 bits) | image << 20
    static_for<NIT>([&](auto itc) __attribute__((always_inline)) {
        constexpr int it = decltype(itc)::value;
        const int e = it * 64 + lane;
        l_off[it] = -1; l_tab[it] = 0;
        if (e < 2 * nload) {
            const int k2 = e >= nload ? 1 : 0, idx = e - k2 * nload;
            const int img = img_of(idx, per_img), rem = idx - img * per_img, r = rem / WI, x = rem - r * WI;
            const int y = y0 * STR - 1 + row0 + r;
            if (y >= 0 && y < HI) {
                l_off[it] = k2 * 8 * HWI + y * WI + x;            // (+ img * batch stride: added per source below)
                l_tab[it] = ((k2 * 2) * PPIX + (img * PR + row0 + r) * PW + x + 1) | (img << 20);
            }
        }
    });

    const int c0_chunks = a_C0 >> 4;
    float xv[NIT][8];
    // x loads of a chunk: 8 channels of one pixel per item (4-byte accesses, 256-byte runs per instruction)
    // (every global access of the K loop is "uniform 64-bit base + 32-bit lane offset": the saddr form, no 64-bit vector address arithmetic)
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        const bool from0 = chunk < c0_chunks;
        const int cc = from0 ? chunk * 16 : (chunk - c0_chunks) * 16;
        const long long sbs = from0 ? a_x0_bs : a_x1_bs;
        const char *src = reinterpret_cast<const char *>((from0 ? a_x0 : a_x1) + (size_t)b0 * sbs + (size_t)cc * HWI);
        static_for<NIT>([&](auto itc) __attribute__((always_inline)) {
            constexpr int it = decltype(itc)::value;
            const unsigned vo = l_off[it] >= 0 ? (unsigned)(((l_tab[it] >> 20) & 0xf) * (int)sbs + l_off[it]) * 4u : 0u;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const char *sc = src + (size_t)c * HWI * 4;             // (uniform: scalar arithmetic)
                const float v = *reinterpret_cast<const float *>(sc + vo);
                xv[it][c] = l_off[it] >= 0 ? v : 0.f;
            }
        });
    };
    const size_t tap_stride = (size_t)a_nchunk * 6 * a_COP;
    // weights of one kernel row of a chunk: 3 taps x planes {WH, WL, WH2 = WH 2^-11}, one 16-byte unit per lane each (A-operand order in
    // memory).  Three planes and ONE accumulator set (the plane-operand kernels hold two sets and two planes): the registers go to the
    // loads in flight instead.
    f16x8 A[2][3][3];
    const unsigned wlane = (unsigned)(kg * a_COP + g * 32 + n) * 16u;
    auto load_a = [&](auto bufc, int chunk, int row) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char *wc = reinterpret_cast<const char *>(a_w) + ((size_t)chunk * 6 * a_COP + (size_t)(row * 3) * tap_stride) * 16;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                A[buf][t][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(wc + ((size_t)t * tap_stride + (size_t)pl * 2 * a_COP) * 16 + wlane));
    };
    typedef std::integral_constant<int, 0> I0;
    // the first chunk's input values and first kernel row are requested before anything else is set up
    load_x(wave);
    load_a(I0{}, wave, 0);
    __builtin_amdgcn_sched_barrier(0);
    WS_STAMP(1);
    WS_STAMP(3);
    // zero halo (and the rows outside the image): written once, the loader only ever writes image pixels
    for (int i = lane; i < 4 * PPIX; i += 64) patch[i] = make_uint4(0, 0, 0, 0);
    WS_STAMP(4);

    // ---- B-operand base of every pixel block: lane = pixel n of the block, k-half kg ---------------------------------------------
    int bbase[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int p = pb * 32 + n;
        const int img = whole ? img_of(p, HW) : 0, rem = p - img * HW, y = rem / W_, x = rem - y * W_;     // (part of an image: y counts from y0)
        bbase[pb] = (kg * 2) * PPIX + (img * PR + y * STR) * PW + x * STR;
    }
    WS_STAMP(5);

    f32x16 acc[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;

    // fp32 -> planes h, l' in this wave's patch
    auto convert_x = [&](int chunk) __attribute__((always_inline)) {
        static_for<NIT>([&](auto itc) __attribute__((always_inline)) {
            constexpr int it = decltype(itc)::value;
            if (l_off[it] < 0) return;
            f16x8 vh, vl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v = xv[it][c];
                _Float16 hq, lq;
                split2h(v, hq, lq);
                vh[c] = hq; vl[c] = lq;
            }
            patch[l_tab[it] & 0xfff] = __builtin_bit_cast(uint4, vh);
            patch[(l_tab[it] & 0xfff) + PPIX] = __builtin_bit_cast(uint4, vl);
        });
    };
    // a = h + l' 2^-11, w 2^s = WH + WL:  acc += WL.h + WH2.l' + WH.h
    auto mma_row = [&](auto bufc, auto rowc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value, row = decltype(rowc)::value;
#ifdef CDC_WS_LAB
        if (a_dbg & 1) return;
#endif
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int p2 = 0; p2 < NPB; p2 += 2) {              // two pixel blocks at a time: 16 operand registers, not 32
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bh[i] = __builtin_bit_cast(f16x8, patch[bbase[p2 + i] + row * PW + t]);
                    bl[i] = __builtin_bit_cast(f16x8, patch[bbase[p2 + i] + PPIX + row * PW + t]);
                }
                // plane-major: consecutive MFMAs go to different accumulators
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[p2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][t][1], bh[i], acc[p2 + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[p2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][t][2], bl[i], acc[p2 + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[p2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][t][0], bh[i], acc[p2 + i], 0, 0, 0);
            }
    };
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    // One chunk.  In flight while it multiplies: the next kernel row's weights (two register sets) and, from its second row on, the
    // NEXT chunk's input values -- the chunk boundary costs the conversion, no exposed load latency.
    auto chunk_body = [&](auto parc, int ci) __attribute__((always_inline)) {
        constexpr int p0 = decltype(parc)::value;
        typedef std::integral_constant<int, p0> BufA;
        typedef std::integral_constant<int, p0 ^ 1> BufB;
        const int chunk = __builtin_amdgcn_readfirstlane(wave + ci * nw);
        const int nxt = __builtin_amdgcn_readfirstlane(wave + (ci + 1 < a_cpw ? ci + 1 : ci) * nw);      // (past the last chunk: a harmless re-read)
        // Program order is PINNED (sched_barrier): hipcc otherwise sinks every load to its first use -- nothing in flight beside the MFMAs
        // (measured: the phases of the first version added up).  In flight while row r multiplies: the weights of row r + 1 and, from
        // the first row of a chunk on, the input values of the NEXT chunk.
        convert_x(chunk);
        if (ci == 0) WS_STAMP(7);
        __builtin_amdgcn_sched_barrier(0);
        load_a(BufB{}, chunk, 1);
        load_x(nxt);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(BufA{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        load_a(BufA{}, chunk, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(BufB{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        load_a(BufB{}, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(BufA{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
    };
    WS_STAMP(6);
    for (int ci = 0; ci < a_cpw; ci += 2) {
        chunk_body(I0{}, ci);
        WS_STAMP(8 + (ci < 3 ? ci : 3));
        if (ci + 1 < a_cpw) { chunk_body(I1{}, ci + 1); WS_STAMP(8 + (ci + 1 < 3 ? ci + 1 : 3)); }
    }

#ifdef CDC_WS_LAB
    if (a_dbg & 8) { if (acc[0][0] == 12345.678f) a_out[0] = acc[1][1] + acc[2][2] + acc[3][3]; return; }
#endif
    // ---- the K slices of the waves meet in LDS: red[wave][pb][4 regs x 4][lane] --------------------------------------------------
    WS_STAMP(12);
    __syncthreads();                                            // every wave is done with its patch
    WS_STAMP(13);
    float4 *red = reinterpret_cast<float4 *>(ws_smem);
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = make_float4(acc[pb][4 * q + 0], acc[pb][4 * q + 1], acc[pb][4 * q + 2], acc[pb][4 * q + 3]);
            red[((wave * NPB + pb) * 4 + q) * 64 + lane] = v;
        }
    __syncthreads();
    WS_STAMP(14);
    // unit u = (pixel block, 16-channel half of the group's 32): wave u, u + waves, ... finishes it -- all waves take part
    for (int u = wave; u < 2 * NPB; u += nw) {
        const int pb = u >> 1, qh = u & 1;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 s4 = red[((0 * NPB + pb) * 4 + 2 * qh + q) * 64 + lane];
            for (int w = 1; w < nw; ++w) {                      // fixed order: deterministic
                const float4 t = red[((w * NPB + pb) * 4 + 2 * qh + q) * 64 + lane];
                s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
            }
            v[4 * q + 0] = s4.x; v[4 * q + 1] = s4.y; v[4 * q + 2] = s4.z; v[4 * q + 3] = s4.w;
        }
        // accumulator layout: lane = pixel n, register r = 4 q' + i -> channel i + 8 q' + 4 kg of the group (q' = 2 qh + q)
        const int p = pb * 32 + n;
        const int img = whole ? p / HW : 0, pix = whole ? p - img * HW : y0 * W_ + p;
        const int cbase = g * 32 + 16 * qh + 4 * kg;
        float mag = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = v[r] * a_acc_scale + (a_bias ? a_bias[cbase + (r & 3) + 8 * (r >> 2)] : 0.f);
            mag += fabsf(v[r]);
        }
        if (a_fault && !(mag < 3.0e38f)) *a_fault = 1;          // non-finite accumulators: reported before the LayerNorm pass can hide them
        if (a_pre_add) {
            const float *pp = a_pre_add + (size_t)(b0 + img) * a_out_bs + (size_t)cbase * HW + pix;
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += pp[(size_t)((r & 3) + 8 * (r >> 2)) * HW];
        }
        float *o = a_out + (size_t)(b0 + img) * a_out_bs + (size_t)cbase * HW + pix;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * HW] = v[r];
    }
    WS_STAMP(15);
}

typedef void (*ws_kernel_fn)(const WsArgs);

}  // namespace cdc
