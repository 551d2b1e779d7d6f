// launch_floor.hip -- what one DEPENDENT kernel boundary costs on this part, and what moves it.
//
// A chain of N launches on one stream.  Every workgroup stamps the constant 100 MHz wall clock (s_memrealtime) when its
// first wave starts and when its last store has been issued; per launch the host reduces min(start) / max(end), so that
//     gap[i]  = min start of launch i+1  -  max end of launch i      (dead time of the device between two dependent kernels)
//     body[i] = max end - min start                                   (the launch's own span)
// are DEVICE-side figures with no event / profiler overhead; `wall` = host clock around the whole chain / N.
// Sweeps: kernarg size, dynamic LDS, scratch, grid / block shape, bytes the predecessor leaves dirty in L2, the store
// policy of those bytes (default, nt, sc1 = write-through), streaming read+write bodies of LayerNorm-pass size,
// eager vs hipGraph.  Reference figures: /opt/skills/guides/MI355X_MICROARCH.md "boundary" row (1.45 us trivial,
// 1.7 - 1.9 us between streaming kernels, + B / 6 TB/s for B dirty bytes).   Development aid, not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct Stamp { u64 *t; };                 // t[2*(launch*maxwg + wg) + {0,1}]
__device__ inline u64 now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ inline void stamp0(u64 *slot) { if (threadIdx.x == 0) slot[0] = now(); }
__device__ inline void stamp1(u64 *slot) {
    __syncthreads();
    if (threadIdx.x == 0) slot[1] = now();
}

struct Big { u64 *slot; int maxwg; int launch; int pad[124]; };                 // 512 bytes by value (PfArgs class)

__global__ void k_small(u64 *st, int maxwg, int launch) {
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot); stamp1(slot);
}
__global__ void k_big(Big a) {
    u64 *slot = a.slot + 2 * ((size_t)a.launch * a.maxwg + blockIdx.x);
    stamp0(slot);
    if (a.pad[threadIdx.x & 63] == 0x7fffffff) slot[1] = 0;
    stamp1(slot);
}
__global__ void k_lds(u64 *st, int maxwg, int launch) {
    extern __shared__ float lds[];
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot);
    lds[threadIdx.x] = (float)launch;
    stamp1(slot);
}
__global__ void k_scratch(u64 *st, int maxwg, int launch, int sel) {
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot);
    volatile float a[256];
    for (int i = 0; i < 256; ++i) a[i] = (float)(i + launch);
    if (a[(sel + threadIdx.x) & 255] == -1.f) slot[1] = 0;
    stamp1(slot);
}

// writer: every workgroup writes `per_wg` bytes (16 B per lane per store) with the chosen policy.  POL 0 default, 1 nt, 2 sc1
// (write-through to the memory side), 3 sc0 sc1.
template <int POL> __global__ void k_write(u64 *st, int maxwg, int launch, f32x4 *dst, size_t per_wg_vec) {
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot);
    f32x4 v = {(float)launch, 1.f, 2.f, 3.f};
    f32x4 *p = dst + (size_t)blockIdx.x * per_wg_vec;
    for (size_t i = threadIdx.x; i < per_wg_vec; i += blockDim.x) {
        if constexpr (POL == 0) p[i] = v;
        else if constexpr (POL == 1) __builtin_nontemporal_store(v, p + i);
        else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
    }
    stamp1(slot);
}
// streaming pass: read src, write dst (LayerNorm-pass-like).  RD: 0 plain, 1 nt loads
template <int POL> __global__ void k_stream(u64 *st, int maxwg, int launch, const f32x4 *src, f32x4 *dst, size_t per_wg_vec) {
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot);
    const f32x4 *s = src + (size_t)blockIdx.x * per_wg_vec;
    f32x4 *p = dst + (size_t)blockIdx.x * per_wg_vec;
    for (size_t i = threadIdx.x; i < per_wg_vec; i += blockDim.x) {
        f32x4 v = s[i];
        v = v * 1.0001f + 0.5f;
        if constexpr (POL == 0) p[i] = v;
        else if constexpr (POL == 1) __builtin_nontemporal_store(v, p + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
    }
    stamp1(slot);
}

// code-size test: 8 KB of straight-line code per instantiation (2048 s_nop = ~1 us hot); 64 distinct instantiations
template <int ID> __global__ void k_code(u64 *st, int maxwg, int launch) {
    u64 *slot = st + 2 * ((size_t)launch * maxwg + blockIdx.x);
    stamp0(slot);
    asm volatile(".rept 2048\n s_nop 0\n .endr" ::: "memory");
    if (launch == -ID - 1) slot[1] = ID;
    stamp1(slot);
}
typedef void (*code_fn)(u64 *, int, int);
template <int... I> static void fill_code(code_fn *t, std::integer_sequence<int, I...>) { ((t[I] = k_code<I>), ...); }

struct Result { double gap_us, body_us, wall_us, gap_p10, gap_p90; };
static const int N = 200, MAXWG = 4096;
static u64 *d_st;
static std::vector<u64> h_st;

template <typename F> static Result chain(hipStream_t s, int wgs, F launch, bool graph = false) {
    Result best{1e9, 0, 1e9, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(d_st, 0, sizeof(u64) * 2 * N * MAXWG, s));
        hipGraphExec_t ge = nullptr;
        if (graph) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) launch(i);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        if (graph) CK(hipGraphLaunch(ge, s));
        else for (int i = 0; i < N; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        if (ge) CK(hipGraphExecDestroy(ge));
        CK(hipMemcpy(h_st.data(), d_st, sizeof(u64) * 2 * N * MAXWG, hipMemcpyDeviceToHost));
        std::vector<double> gaps; double body = 0; u64 prev_end = 0;
        for (int i = 0; i < N; ++i) {
            u64 mn = ~0ull, mx = 0;
            for (int w = 0; w < wgs && w < MAXWG; ++w) { u64 a = h_st[2 * ((size_t)i * MAXWG + w)], b = h_st[2 * ((size_t)i * MAXWG + w) + 1]; if (a && a < mn) mn = a; if (b > mx) mx = b; }
            if (i > 10) gaps.push_back((double)(mn - prev_end) * 0.01);      // 100 MHz -> us; skip the ramp of the chain
            if (i > 10) body += (double)(mx - mn) * 0.01;
            prev_end = mx;
        }
        std::sort(gaps.begin(), gaps.end());
        Result r;
        r.gap_us = gaps[gaps.size() / 2]; r.gap_p10 = gaps[gaps.size() / 10]; r.gap_p90 = gaps[gaps.size() * 9 / 10];
        r.body_us = body / (N - 11);
        r.wall_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        if (r.wall_us < best.wall_us) best = r;
    }
    return best;
}
static void show(const char *name, const Result &r) {
    printf("%-64s gap %5.2f us (p10 %5.2f p90 %5.2f)  body %7.2f us  wall/launch %7.2f us\n", name, r.gap_us, r.gap_p10, r.gap_p90, r.body_us, r.wall_us);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMalloc(&d_st, sizeof(u64) * 2 * N * MAXWG));
    h_st.resize((size_t)2 * N * MAXWG);
    const size_t BUF = (size_t)256 << 20;
    f32x4 *a, *b; CK(hipMalloc(&a, BUF)); CK(hipMalloc(&b, BUF));
    CK(hipMemset(a, 0, BUF)); CK(hipMemset(b, 0, BUF));
    CK(hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs; chain of %d dependent launches on one non-blocking stream; device-side stamps (100 MHz)\n", pr.name, pr.multiProcessorCount, N);
    char nm[160];

    printf("## (a) kernarg size, grid / block shape (trivial body)\n");
    for (int wgs : {256, 1024, 4096}) for (int thr : {64, 256, 512}) {
        if (quick && (wgs != 256 || thr != 256)) continue;
        snprintf(nm, sizeof nm, "trivial, 20-B kernarg, %4d WG x %3d threads", wgs, thr);
        show(nm, chain(s, wgs, [&](int i) { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(thr), 0, s, d_st, MAXWG, i); }));
    }
    for (int wgs : {256, 1024}) {
        snprintf(nm, sizeof nm, "trivial, 512-B kernarg by value, %4d WG x 256", wgs);
        show(nm, chain(s, wgs, [&](int i) { Big g{}; g.slot = d_st; g.maxwg = MAXWG; g.launch = i; hipLaunchKernelGGL(k_big, dim3(wgs), dim3(256), 0, s, g); }));
    }
    show("trivial, 20-B kernarg, 256 WG x 256, hipGraph replay", chain(s, 256, [&](int i) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }, true));
    show("trivial, 512-B kernarg, 256 WG x 256, hipGraph replay", chain(s, 256, [&](int i) { Big g{}; g.slot = d_st; g.maxwg = MAXWG; g.launch = i; hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, g); }, true));

    printf("## (b) dynamic LDS per workgroup (256 WG x 256)\n");
    for (int lds : {0, 16 << 10, 64 << 10, 150 << 10}) {
        snprintf(nm, sizeof nm, "trivial + %3d KB dynamic LDS", lds >> 10);
        show(nm, chain(s, 256, [&](int i) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), lds + 1024, s, d_st, MAXWG, i); }));
    }
    for (int lds : {64 << 10, 150 << 10}) {
        snprintf(nm, sizeof nm, "trivial + %3d KB dynamic LDS, 512 threads", lds >> 10);
        show(nm, chain(s, 256, [&](int i) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), lds + 1024, s, d_st, MAXWG, i); }));
    }
    printf("## (c) scratch (1 KB private array per lane)\n");
    show("scratch body, 256 WG x 256", chain(s, 256, [&](int i) { hipLaunchKernelGGL(k_scratch, dim3(256), dim3(256), 0, s, d_st, MAXWG, i, i); }));
    show("scratch body, 1024 WG x 256", chain(s, 1024, [&](int i) { hipLaunchKernelGGL(k_scratch, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, i); }));

    printf("## (d,e) predecessor leaves B bytes dirty; store policy (writer chain: gap FOLLOWS a writer; 1024 WG x 256)\n");
    for (size_t mb : {0, 1, 4, 16, 32, 128}) {
        const size_t per = mb ? ((mb << 20) / 16 / 1024) : 0;
        snprintf(nm, sizeof nm, "writer %3zu MB, default stores", mb);
        show(nm, chain(s, 1024, [&](int i) { hipLaunchKernelGGL(k_write<0>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, per); }));
        if (!mb) continue;
        snprintf(nm, sizeof nm, "writer %3zu MB, nt stores", mb);
        show(nm, chain(s, 1024, [&](int i) { hipLaunchKernelGGL(k_write<1>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, per); }));
        snprintf(nm, sizeof nm, "writer %3zu MB, sc1 stores (write-through)", mb);
        show(nm, chain(s, 1024, [&](int i) { hipLaunchKernelGGL(k_write<2>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, per); }));
        snprintf(nm, sizeof nm, "writer %3zu MB, sc0 sc1 stores", mb);
        show(nm, chain(s, 1024, [&](int i) { hipLaunchKernelGGL(k_write<3>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, per); }));
    }
    printf("## (f) streaming pass a -> b -> a ... (LayerNorm-pass shape: reads what the predecessor wrote)\n");
    for (size_t mb : {1, 3, 12, 48, 100}) for (int wgs : {256, 1024}) {
        const size_t per = (mb << 20) / 16 / wgs;
        snprintf(nm, sizeof nm, "stream %3zu MB in + out, %4d WG, default stores", mb, wgs);
        show(nm, chain(s, wgs, [&](int i) { hipLaunchKernelGGL(k_stream<0>, dim3(wgs), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, (i & 1) ? b : a, per); }));
        snprintf(nm, sizeof nm, "stream %3zu MB in + out, %4d WG, nt stores", mb, wgs);
        show(nm, chain(s, wgs, [&](int i) { hipLaunchKernelGGL(k_stream<1>, dim3(wgs), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, (i & 1) ? b : a, per); }));
        snprintf(nm, sizeof nm, "stream %3zu MB in + out, %4d WG, sc1 stores", mb, wgs);
        show(nm, chain(s, wgs, [&](int i) { hipLaunchKernelGGL(k_stream<2>, dim3(wgs), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, (i & 1) ? b : a, per); }));
    }
    show("stream 3 MB, 256 WG, default stores, hipGraph replay", chain(s, 256, [&](int i) { hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(256), 0, s, d_st, MAXWG, i, (i & 1) ? a : b, (i & 1) ? b : a, (size_t)(3 << 20) / 16 / 256); }, true));
    printf("## (h) instruction fetch: 8 KB straight-line body; the same kernel again and again / 64 distinct kernels round robin / the\n"
           "##     same with a 64 MB streaming launch between two of them (L2 contents replaced); figures of the code kernels only\n");
    {
        static code_fn tab[64];
        fill_code(tab, std::make_integer_sequence<int, 64>{});
        show("8 KB code, same kernel, 256 WG x 256", chain(s, 256, [&](int i) { hipLaunchKernelGGL(tab[0], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }));
        show("8 KB code, 64 distinct kernels round robin", chain(s, 256, [&](int i) { hipLaunchKernelGGL(tab[i & 63], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }));
        show("8 KB code, same kernel, hipGraph", chain(s, 256, [&](int i) { hipLaunchKernelGGL(tab[0], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }, true));
        show("8 KB code, 64 distinct kernels, hipGraph", chain(s, 256, [&](int i) { hipLaunchKernelGGL(tab[i & 63], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }, true));
        // interleaved: even launches = code kernel (stamped), odd = stream kernel writing into the stamp slots of its own index
        const size_t per = ((size_t)64 << 20) / 16 / 1024;
        show("[8 KB code same kernel | 64 MB stream] alternating (both stamped)", chain(s, 256, [&](int i) {
            if (i & 1) hipLaunchKernelGGL(k_stream<0>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, a, b, per);
            else hipLaunchKernelGGL(tab[0], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }));
        show("[8 KB code 64 distinct | 64 MB stream] alternating (both stamped)", chain(s, 256, [&](int i) {
            if (i & 1) hipLaunchKernelGGL(k_stream<0>, dim3(1024), dim3(256), 0, s, d_st, MAXWG, i, a, b, per);
            else hipLaunchKernelGGL(tab[(i >> 1) & 63], dim3(256), dim3(256), 0, s, d_st, MAXWG, i); }));
    }
    printf("## (g) with a hipEvent pair around every launch (what per-op timing adds)\n");
    {
        std::vector<hipEvent_t> ev(2 * N);
        for (auto &e : ev) CK(hipEventCreate(&e));
        show("trivial 256 WG x 256 + hipEventRecord before and after", chain(s, 256, [&](int i) { CK(hipEventRecord(ev[2 * i], s)); hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d_st, MAXWG, i); CK(hipEventRecord(ev[2 * i + 1], s)); }));
        CK(hipStreamSynchronize(s));
        float ms = 0, tot = 0; for (int i = 20; i < N; ++i) { CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
        printf("   hipEventElapsedTime around one trivial launch: %.2f us (mean)\n", tot / (N - 20) * 1e3);
        for (auto &e : ev) CK(hipEventDestroy(e));
    }
    return 0;
}
