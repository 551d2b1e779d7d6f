// dma_order.hip -- does a COUNTED `s_waitcnt vmcnt(N)` cover the LDS-DMA pieces older than the N newest?  
//
// conv_pw_kernel's rare stale-activation event (profiles/determinism_r05.txt) went away when its per-step counted wait became
// vmcnt(0).  The counted wait is exact only if a wave's vector-memory operations retire from the counter IN ISSUE ORDER.
// This program isolates that assumption: every wave, per step,
//     issues L "activation" pieces (global_load_lds_dwordx4, 1 KiB each) from a COLD place of a big buffer into slot (step & 1),
//     issues NW "weight" pieces from a small HOT buffer (L2 hits) into a ring,
//     [optionally issues S global stores -- the mix a persistent kernel's epilogue produces],
//     waits  vmcnt(NW [+ S] + (L + NW [+ S]))        -- everything newer than the activation pieces of step - 1 may fly --
//     reads the activation pieces of step - 1 back from LDS and compares them with what the source holds (word i of the
//     big buffer holds i): a mismatch is a piece that had not landed when the counter said it had.
// Control arm: the same loop with vmcnt(0).  Variants: cold / hot / mixed activation addresses, 4-byte activation pieces
// (global_load_lds_dword) mixed with 16-byte weight pieces, stores in flight.   Development aid, not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void *g, unsigned lds_byte) {
    unsigned keep;      // (M0 is compiler-reserved: saved and restored, as the library's dma helpers do)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
__device__ __forceinline__ void dma4(const void *g, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Rec { u32 step, piece, lane, got, want, kind, wave, wg; };
__device__ __forceinline__ u32 rng(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE bit 0: counted wait (else vmcnt(0));  bit 1: stores in flight;  bit 2: 4-byte activation pieces;
// bit 3: LDS read pressure (every wave reads 48 x 1 KiB of the workgroup's LDS per step with ds_read_b128, as the A / B operand fetches of the real kernels do)
// addr: 0 = every piece cold (random 1-KiB place of the big buffer), 1 = hot (a 4-MiB window), 2 = one cold piece in eight
template <int L, int NW, int MODE> __global__ void __launch_bounds__(256) k(const u32 *big, size_t big_words, const u32 *hot, u32 *sink, int steps, int addr,
                                                                            unsigned long long *bad, Rec *recs, u32 seed, int rowstride = 0) {
    extern __shared__ u32 lds[];
    constexpr bool COUNTED = MODE & 1, STORES = MODE & 2, X4 = MODE & 4;
    constexpr int S = STORES ? 2 : 0;
    constexpr int XW = X4 ? 64 : 256;                                  // words per activation piece
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 *X = lds + wave * (2 * L * 256 + 2 * NW * 256);                // [2][L][256 words]  (4-byte pieces use the first 64 words of a piece slot)
    u32 *W = X + 2 * L * 256;                                          // [2][NW][256]
    const unsigned xb = (unsigned)(size_t)(__attribute__((address_space(3))) u32 *)X, wb = (unsigned)(size_t)(__attribute__((address_space(3))) u32 *)W;
    const u32 wid = (blockIdx.x * 4 + wave);
    const size_t places = (big_words - (size_t)8 * rowstride) / 256 - 1;
    // rowstride != 0: lane (row = lane >> 3, quad = lane & 7) reads 16 bytes at row * rowstride words + quad * 4 -- eight 128-byte rows in
    // eight different pages, as conv_pw_kernel's 16-byte activation pieces (8 channel planes of an fp32 NCHW tensor)
    const size_t lane_w = rowstride ? (size_t)(lane >> 3) * rowstride + (size_t)(lane & 7) * 4 : (size_t)lane * 4;
    auto place = [&](int step, int p) -> size_t {
        u32 r = rng(seed ^ (wid * 0x9e3779b9u) ^ (u32)(step * L + p) * 0x85ebca6bu);
        if (addr == 1 || (addr == 2 && (r & 7))) return (size_t)(r >> 8) % 4096;              // 4 MiB window
        return (size_t)(((unsigned long long)r * places) >> 32);
    };
    unsigned long long nbad = 0;
    u32 sv = wid;
    for (int step = 0; step <= steps; ++step) {
        if (step < steps) {
#pragma unroll
            for (int p = 0; p < L; ++p) {
                const u32 *src = big + place(step, p) * 256;
                if constexpr (X4) dma4(src + lane, xb + (unsigned)(((step & 1) * L + p) * 1024));
                else dma16(src + lane_w, xb + (unsigned)(((step & 1) * L + p) * 1024));
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) dma16(hot + ((size_t)((step * NW + q) & 63) * 256) + lane * 4, wb + (unsigned)(((step & 1) * NW + q) * 1024));
            if constexpr (STORES) {
#pragma unroll
                for (int q = 0; q < S; ++q) { u32x4 v = {sv, sv + 1, sv + 2, sv + 3}; asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(sink + ((size_t)wid * S + q) * 256 + lane * 4), "v"(v) : "memory"); }
            }
        }
        if constexpr (MODE & 8) {
            u32x4 acc4 = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 48; ++q) acc4 += *reinterpret_cast<const u32x4 *>(lds + ((q * 977 + wave * 131 + step) & 4095) * 4 + 0 * lane);    // (same address in every lane: a broadcast read per instruction; 48 instructions)
            sv += acc4[0] ^ acc4[3];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc4 += *reinterpret_cast<const u32x4 *>(lds + (((q * 613 + step) & 255) * 64 + lane) * 4);               // (16 full-width 1-KiB reads)
            sv += acc4[1];
        }
        if (step == 0) continue;
        // activation pieces of step - 1: older than the (NW + S) + (L + NW + S) operations issued since
        if (COUNTED && step < steps) vm_wait<NW + S + L + NW + S>(); else vm_wait<0>();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < L; ++p) {
            const size_t pl = place(step - 1, p);
            if constexpr (X4) {
                const u32 got = X[(((step - 1) & 1) * L + p) * 256 + lane], want = (u32)(pl * 256 + lane);
                if (got != want) { if (nbad == 0 && recs) { unsigned long long i = atomicAdd(bad + 1, 1ull); if (i < 64) recs[i] = Rec{(u32)step - 1, (u32)p, (u32)lane, got, want, 0, (u32)wave, blockIdx.x}; } ++nbad; }
            } else {
                const u32x4 got = *reinterpret_cast<const u32x4 *>(X + (((step - 1) & 1) * L + p) * 256 + lane * 4);
                const u32 want = (u32)(pl * 256 + lane_w);
                if (got[0] != want || got[1] != want + 1 || got[2] != want + 2 || got[3] != want + 3) {
                    if (nbad == 0 && recs) { unsigned long long i = atomicAdd(bad + 1, 1ull); if (i < 64) recs[i] = Rec{(u32)step - 1, (u32)p, (u32)lane, got[0], want, 0, (u32)wave, blockIdx.x}; }
                    ++nbad;
                }
            }
        }
        sv += 4;
        __builtin_amdgcn_wave_barrier();
    }
    vm_wait<0>();
    if (nbad) atomicAdd(bad, nbad);
    if (sv == 0x12345678u) sink[0] = sv;
}

template <int L, int NW, int MODE> static void run(const char *name, const u32 *big, size_t words, const u32 *hot, u32 *sink, int steps, int addr, int launches, int rowstride = 0) {
    unsigned long long *bad; Rec *recs;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&recs, sizeof(Rec) * 64));
    CK(hipMemset(bad, 0, 16));
    const int lds = 4 * (2 * L * 256 + 2 * NW * 256) * 4;
    CK(hipFuncSetAttribute((const void *)k<L, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = 256 * (lds <= 80 * 1024 ? 2 : 1);
    CK(hipEventRecord(e0));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k<L, NW, MODE>), dim3(wgs), dim3(256), lds, 0, big, words, hot, sink, steps, addr, bad, recs, 0x1234u + i * 977u, rowstride);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    const double waits = (double)wgs * 4 * steps * launches;
    printf("%-78s waits %.2e  bad lanes %llu  waves with an event %llu   (%.0f ms, %.2f TB/s)\n", name, waits, h[0], h[1], ms,
           waits * ((MODE & 4 ? 256.0 : 1024.0) * L + 1024.0 * NW) / ms / 1e9);
    if (h[1]) {
        std::vector<Rec> r(64); CK(hipMemcpy(r.data(), recs, sizeof(Rec) * 64, hipMemcpyDeviceToHost));
        for (unsigned long long i = 0; i < h[1] && i < 6; ++i)
            printf("    wg %u wave %u step %u piece %u lane %u: got word %u, source holds %u\n", r[i].wg, r[i].wave, r[i].step, r[i].piece, r[i].lane, r[i].got, r[i].want);
    }
    fflush(stdout);
    CK(hipFree(bad)); CK(hipFree(recs));
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 20000, launches = argc > 2 ? atoi(argv[2]) : 4;
    const size_t words = (size_t)1 << 30;          // 4 GiB of u32
    u32 *big, *hot, *sink;
    CK(hipMalloc(&big, words * 4)); CK(hipMalloc(&hot, 64 * 1024)); CK(hipMalloc(&sink, (size_t)256 * 2 * 4 * 2 * 1024 + 4096));
    {   // word i holds i
        std::vector<u32> h((size_t)1 << 24);
        for (size_t off = 0; off < words; off += h.size()) { for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(off + i); CK(hipMemcpy(big + off, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        CK(hipMemset(hot, 0, 64 * 1024));
    }
    printf("# per wave and step: L activation pieces (cold / hot / mixed source) + NW hot weight pieces [+ 2 stores]; read back the\n"
           "# activation pieces of the PREVIOUS step after `s_waitcnt vmcnt(newer operations)` (counted) or vmcnt(0) (control)\n");
    if (argc > 3 && !strcmp(argv[3], "scatter")) {
        // the shape of the real event: L = 4 activation pieces of eight 128-byte rows 16 KiB apart, NW = 2 weight pieces
        for (int addr = 0; addr < 3; ++addr) {
            const char *an = addr == 0 ? "cold" : addr == 1 ? "hot" : "1 cold in 8";
            char nm[160];
            snprintf(nm, sizeof nm, "L=4 NW=2, pieces of 8 rows 16 KiB apart, %s, vmcnt(0) control", an);   run<4, 2, 0>(nm, big, words, hot, sink, steps, addr, launches, 4096);
            snprintf(nm, sizeof nm, "L=4 NW=2, pieces of 8 rows 16 KiB apart, %s, COUNTED wait", an);       run<4, 2, 1>(nm, big, words, hot, sink, steps, addr, launches, 4096);
            snprintf(nm, sizeof nm, "L=4 NW=2, pieces of 8 rows 1 MiB apart, %s, COUNTED wait", an);        run<4, 2, 1>(nm, big, words, hot, sink, steps, addr, launches, 262144);
            snprintf(nm, sizeof nm, "L=4 NW=2, 8 rows 16 KiB apart, %s, COUNTED wait + LDS read pressure", an);   run<4, 2, 9>(nm, big, words, hot, sink, steps, addr, launches, 4096);
            snprintf(nm, sizeof nm, "L=4 NW=2, 8 rows 16 KiB apart, %s, vmcnt(0) + LDS read pressure", an);       run<4, 2, 8>(nm, big, words, hot, sink, steps, addr, launches, 4096);
            snprintf(nm, sizeof nm, "L=4 NW=2, contiguous 1-KiB pieces, %s, COUNTED wait + LDS read pressure", an); run<4, 2, 9>(nm, big, words, hot, sink, steps, addr, launches, 0);
        }
        return 0;
    }
    for (int addr = 0; addr < 3; ++addr) {
        const char *an = addr == 0 ? "cold" : addr == 1 ? "hot" : "1 cold in 8";
        char nm[160];
        snprintf(nm, sizeof nm, "L=8 NW=2 16-byte pieces, %s activations, vmcnt(0) control", an);        run<8, 2, 0>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=8 NW=2 16-byte pieces, %s activations, COUNTED wait", an);            run<8, 2, 1>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=8 NW=2 16-byte pieces, %s activations, COUNTED wait, stores in flight", an);   run<8, 2, 3>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=8 NW=2 4-byte activation pieces, %s, COUNTED wait", an);              run<8, 2, 5>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=8 NW=2 4-byte activation pieces, %s, COUNTED wait, stores in flight", an);     run<8, 2, 7>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=16 NW=4 16-byte pieces (one WG per CU), %s, COUNTED wait", an);       run<16, 4, 1>(nm, big, words, hot, sink, steps, addr, launches);
        snprintf(nm, sizeof nm, "L=16 NW=4 16-byte pieces (one WG per CU), %s, COUNTED, stores", an);    run<16, 4, 3>(nm, big, words, hot, sink, steps, addr, launches);
    }
    return 0;
}
