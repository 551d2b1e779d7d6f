#!/usr/bin/env python3
"""Lab aid: derive tools/ubench/pfpp_lab.h (conv_pf_kernel with two 4-wave groups in explicit ping-pong) from
cdc_compression_amd/csrc/conv_pf_kernel.h by text substitution.  Timing experiment only."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'cdc_compression_amd/csrc/conv_pf_kernel.h')).read()
i = s.index('template <int MB, int NPW, int WM, int WP, int KH, int KW>\n__global__')
j = s.index('typedef void (*pf_kernel_fn)(const PfArgs);')
k = s[i:j]
def rep(a, b):
    global k
    assert a in k, a
    k = k.replace(a, b)
rep('__launch_bounds__(64 * WM * WP, (WM * WP == 8 || MB * NPW <= 4) ? 2 : 1) conv_pf_kernel(const PfArgs P)', '__launch_bounds__(128 * WM * WP, 1) conv_pfpp_kernel(const PfArgs P)')
rep('constexpr int NW = WM * WP, NT = 64 * NW, COPT = WM * MB * 32;', 'constexpr int NW = WM * WP, NT = 64 * NW, COPT = WM * MB * 32;\n    static_assert(NW == 4, "ping-pong lab: two groups of four waves");')
rep('    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);', '    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int grp = wave8 >> 2, wave = wave8 & 3;\n    const int gtid = tid & 255;')
rep('unsigned bid0 = blockIdx.x;', 'unsigned bid0 = blockIdx.x * 2 + grp;')
rep('bid0 = (bid0 & 7) * (gridDim.x >> 3) + (bid0 >> 3);', 'bid0 = (blockIdx.x & 7) * (gridDim.x >> 3) * 2 + (blockIdx.x >> 3) * 2 + grp;')
rep('const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) uint4 *)smem_u);',
    'uint4 *smem_g = smem_u + grp * P.ring;   // lab: P.ring = LDS units per group\n    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) uint4 *)smem_g);')
rep('const uint4 *a_base = smem_u + NPB * PST', 'const uint4 *a_base = smem_g + NPB * PST')
rep('const uint4 *b_base = smem_u + (half * 2) * PLANE', 'const uint4 *b_base = smem_g + (half * 2) * PLANE')
rep('        --rem;\n        __builtin_amdgcn_s_setprio(2);', '        --rem;\n        __builtin_amdgcn_s_barrier();                     // ping-pong: load segment | compute segment\n        __builtin_amdgcn_s_setprio(2);')
rep('    dma_wait();\n    __builtin_amdgcn_s_barrier();\n    OpsA A0, A1;', '    dma_wait();\n    __builtin_amdgcn_s_barrier();\n    if (grp == 1) __builtin_amdgcn_s_barrier();           // group B runs one interval behind\n    OpsA A0, A1;')
rep('    // ---- epilogue ----', '    if (grp == 0) __builtin_amdgcn_s_barrier();\n    // ---- epilogue ----')
rep('float *ep = reinterpret_cast<float *>(smem_u);', 'float *ep = reinterpret_cast<float *>(smem_g);')
rep('for (int i = tid; i < COPT; i += NT) {', 'for (int i = gtid; i < COPT; i += NT) {')
rep('    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];', '    TL2(const unsigned long long tl2_0 = __builtin_readcyclecounter();)\n    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];')
rep('    if (grp == 1) __builtin_amdgcn_s_barrier();           // group B runs one interval behind', '    TL2(const unsigned long long tl2_1 = __builtin_readcyclecounter();)\n    if (grp == 1) __builtin_amdgcn_s_barrier();           // group B runs one interval behind')
rep('    if (grp == 0) __builtin_amdgcn_s_barrier();\n    // ---- epilogue ----', '    if (grp == 0) __builtin_amdgcn_s_barrier();\n    TL2(if (P.res3_x && blockIdx.x < 64 && lane == 0) { unsigned long long *o = (unsigned long long *)P.res3_x + (blockIdx.x * 8 + wave8) * 8; o[6] = tl2_1 - tl2_0; o[7] = __builtin_readcyclecounter() - tl2_1; })\n    // ---- EPILOGUE ----')
# ---- timeline accumulators (LAB_TL): per-wave totals of each segment of the tap -------------------------------
rep('    int rem = S - 1;                                      // taps after the one being multiplied',
    '    int rem = S - 1;                                      // taps after the one being multiplied\n    TLV(unsigned long long tl_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tl_t = __builtin_readcyclecounter();)')
rep('        if (!(TAIL && t == TAPS - 1) || rem > 0) {        // (the very last tap has nothing left to fetch)\n            if (++sn == R) sn = 0;',
    '        if (!(TAIL && t == TAPS - 1) || rem > 0) {        // (the very last tap has nothing left to fetch)\n            if (++sn == R) sn = 0;\n            TLS(0);')
rep('            if (!(CDC_PF_ABLATE && (P.dbg & 4))) __builtin_amdgcn_s_barrier();\n            const uint4 *wa = a_base + sn * WST;',
    '            TLS(1);\n            if (!(CDC_PF_ABLATE && (P.dbg & 4))) __builtin_amdgcn_s_barrier();\n            TLS(2);\n            const uint4 *wa = a_base + sn * WST;')
rep('            if (patch_wave) {\n                if constexpr (t < ISSUE_TAPS)', '            TLS(3);\n            if (patch_wave) {\n                if constexpr (t < ISSUE_TAPS)')
rep('        --rem;\n        __builtin_amdgcn_s_barrier();                     // ping-pong: load segment | compute segment',
    '        --rem;\n        TLS(4);\n        __builtin_amdgcn_s_barrier();                     // ping-pong: load segment | compute segment\n        TLS(5);')
rep('    if (grp == 0) __builtin_amdgcn_s_barrier();\n',
    '    TLV(if (P.res3_x && blockIdx.x < 16 && lane == 0) { unsigned long long *o = (unsigned long long *)P.res3_x + (blockIdx.x * 8 + wave8) * 8; for (int q = 0; q < 6; ++q) o[q] = tl_acc[q]; })\n    if (grp == 0) __builtin_amdgcn_s_barrier();\n')
k = '''#ifdef LAB_TL2
#define TL2(...) __VA_ARGS__
#else
#define TL2(...)
#endif
#ifdef LAB_TL
#define TLV(...) __VA_ARGS__
// TLS(i): time since the previous stamp is charged to bucket i: 0 = mma of the previous tap (+ loop control), 1 = vm wait,
// 2 = barrier X, 3 = operand fetch (issue + completion: the stamp drains lgkmcnt), 4 = DMA issue, 5 = barrier Y
#define TLS(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[i] += n_ - tl_t; tl_t = n_; } while (0)
#else
#define TLV(...)
#define TLS(i) do { } while (0)
#endif
''' + k
open(os.path.join(R, 'tools/ubench/pfpp_lab.h'), 'w').write('// GENERATED by gen_pfpp_lab.py -- lab only\n#pragma once\n#include "conv_pf_kernel.h"\nnamespace cdc {\n' + k + '\n}\n')
