// entropy.hip -- SURVEY section 8f row 4: an actual entropy coder for the transmitted symbols.
//
// The reference only ESTIMATES the rate (Compressor.bpp, xparam/modules/compress_modules.py:76-90): it sums
// -log2 of FlexiblePrior.likelihood(q_hyper_latent) (network_components.py:372-378) and of
// NormalDistribution(mean, scale).likelihood(q_latent) (utils.py:155-159) and never writes a bit.  This file codes
// exactly those two symbol sets with exactly those two models:
//     hyper symbols  k = q_hyper_latent - medians    per-channel tables  p_c(k) = likelihood(medians_c + k)
//     latent symbols k = q_latent - mean             tables by scale:     p(k)  = Phi((k+.5)/s) - Phi((k-.5)/s)
// with a byte-wise range-ANS coder (32-bit state, 16-bit probabilities).  The specification of the integer tables
// (below) is restated independently by the CPU checker of the test tree; streams must agree byte for byte.
//
// Division of labour: everything per-symbol runs on the GPU -- quantisation against the medians / the mean, scale ->
// table index, hyper_dec itself, and the coder proper (below: 64 interleaved range-ANS states per section = one wave per
// section, all 2 B sections of a batch in two launches) -- only the probability tables are evaluated on the host, a few
// thousand doubles once per model in float64 with libm (so that encoder and decoder, product and the CPU checker, all hold
// the SAME integers -- device transcendentals are not bit-reproducible across toolchains), and uploaded as integers.
//
// CONTRACT (the one real hazard of learned codecs): the decoder must reproduce the encoder's `scale` bit for bit or
// the table index of a latent may differ and everything after it is garbage.  Both sides therefore run hyper_dec through
// the SAME launch program -- planned as for ONE image whatever the batch (cdc_api.hip: Builder::planB; the kernels never
// mix images, so image b of a batch-B run holds the bits of a batch-1 run; tests/test_entropy.py holds that) -- in the
// arithmetic recorded in the stream header, on integer-valued inputs that the stream reproduces exactly.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "entropy.h"

namespace cdc {

// ---- table specification --------------------------------------------------------------------------------------
//   PREC = 16.  A table over entries j = 0 .. n-1 (the last entry is the ESCAPE symbol) gets
//       f_j = 1 + floor(p_j * (65536 - n)),   then 65536 - sum f_j is added to the entry of the largest p_j (first
//       one on ties); cumulative starts c_j = sum_{i<j} f_i.
//   Gaussian tables: NB = 128 scales e_i = (float) exp(ln 0.1 + i (ln 2048 - ln 0.1) / 127); a latent with scale s uses the
//       smallest i with s <= e_i (i = 127 beyond); support k in [-K_i, K_i], K_i = min(1023, ceil(8 e_i) + 1);
//       p(k) = 0.5 erfc(-(0.5 - |k|) / (e_i sqrt 2)) - 0.5 erfc(-(-0.5 - |k|) / (e_i sqrt 2))   (utils.py:147-159);
//       escape mass = max(0, 1 - sum p).
//   Hyper tables: per channel, support [-K, K] with K the smallest of 8, 16, 32, ... , 1024 whose mass exceeds 1 - 2^-20
//       (1024 if none); p(k) = |sigmoid(s U) - sigmoid(s L)|, L, U = logits(m + k -/+ 0.5), s = -sign(L + U)
//       (network_components.py:372-378), logits = the 1-3-3-3-1 softplus / tanh chain of FlexiblePrior.cdf in float64.
//   Escape payload: w = ((|k| - K - 1) << 1) | (k < 0) as one u32 in the section's payload list (forward symbol order).
constexpr int kPrec = 16;
constexpr uint32_t kTot = 1u << kPrec;
constexpr uint32_t kRansL = 1u << 23;

static void make_freqs(const std::vector<double> &p, EntropyTable *t) {
    const int n = (int)p.size();
    t->freq.assign(n, 0);
    t->start.assign(n + 1, 0);
    uint32_t sum = 0;
    int best = 0;
    for (int j = 0; j < n; ++j) {
        double v = p[j];
        if (!(v > 0)) v = 0;
        const uint32_t f = 1u + (uint32_t)floor(v * (double)(kTot - (uint32_t)n));
        t->freq[j] = f;
        sum += f;
        if (p[j] > p[best]) best = j;
    }
    t->freq[best] += kTot - sum;
    for (int j = 0; j < n; ++j) t->start[j + 1] = t->start[j] + t->freq[j];
}

void entropy_scale_edges(float *e) {
    const double lo = log(0.1), hi = log(2048.0);
    for (int i = 0; i < kEntropyBins; ++i) e[i] = (float)exp(lo + (double)i * (hi - lo) / (double)(kEntropyBins - 1));
}

static void build_gauss(EntropyModel *m) {
    entropy_scale_edges(m->edges);
    m->gauss.resize(kEntropyBins);
    for (int i = 0; i < kEntropyBins; ++i) {
        const double s = (double)m->edges[i];
        const int K = std::min(1023, (int)ceil(8.0 * s) + 1);
        std::vector<double> p(2 * K + 2);
        double tot = 0;
        const double c = -sqrt(0.5);
        for (int k = -K; k <= K; ++k) {
            const double x = fabs((double)k);
            const double up = 0.5 * erfc(c * ((0.5 - x) / s)), lw = 0.5 * erfc(c * ((-0.5 - x) / s));
            p[k + K] = up - lw;
            tot += p[k + K];
        }
        p[2 * K + 1] = std::max(0.0, 1.0 - tot);
        m->gauss[i].K = K;
        make_freqs(p, &m->gauss[i]);
    }
}

static double prior_logit(const double *q, double x) {        // FlexiblePrior.cdf(x, logits=True) of one channel
    // q: softplus(W0)[3] b0[3] tanh(a0)[3] | softplus(W1)[9] b1[3] tanh(a1)[3] | softplus(W2)[9] b2[3] tanh(a2)[3] | softplus(W3)[3] b3
    double h[3], g[3];
    for (int k = 0; k < 3; ++k) { h[k] = x * q[k] + q[3 + k]; h[k] += q[6 + k] * tanh(h[k]); }
    q += 9;
    for (int l = 0; l < 2; ++l) {
        for (int j = 0; j < 3; ++j) g[j] = h[0] * q[j] + h[1] * q[3 + j] + h[2] * q[6 + j] + q[9 + j];
        for (int j = 0; j < 3; ++j) h[j] = g[j] + q[12 + j] * tanh(g[j]);
        q += 15;
    }
    return h[0] * q[0] + h[1] * q[1] + h[2] * q[2] + q[3];
}

static double sigmoid_d(double v) { return 1.0 / (1.0 + exp(-v)); }

void entropy_build_hyper(EntropyModel *m, const double *prior /* [C][44] */, const float *medians, int C) {
    m->hyper.resize(C);
    m->medians.assign(medians, medians + C);
    for (int c = 0; c < C; ++c) {
        const double *q = prior + (size_t)c * 44;
        const double med = (double)medians[c];
        auto pk = [&](int k) {
            const double L = prior_logit(q, med + k - 0.5), U = prior_logit(q, med + k + 0.5);
            const double sg = (L + U) > 0 ? -1.0 : ((L + U) < 0 ? 1.0 : 0.0);
            return fabs(sigmoid_d(U * sg) - sigmoid_d(L * sg));
        };
        int K = 8;
        for (;; K *= 2) {
            double tot = 0;
            for (int k = -K; k <= K; ++k) tot += pk(k);
            if (tot > 1.0 - ldexp(1.0, -20) || K >= 1024) break;
        }
        std::vector<double> p(2 * K + 2);
        double tot = 0;
        for (int k = -K; k <= K; ++k) { p[k + K] = pk(k); tot += p[k + K]; }
        p[2 * K + 1] = std::max(0.0, 1.0 - tot);
        m->hyper[c].K = K;
        make_freqs(p, &m->hyper[c]);
    }
}

void entropy_init(EntropyModel *m) {
    if (m->gauss.empty()) build_gauss(m);
}

// ---- fingerprint of the tables carried by the stream header (include/cdc_hip.h): FNV-1a, 32 bit ---------------------------
static inline uint32_t fnv_u32(uint32_t h, uint32_t v) {
    for (int i = 0; i < 4; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}
// every integer the coder uses: per table K, then the 2K + 2 frequencies; hyper tables (channel order), then the scale tables
uint32_t entropy_model_hash(const EntropyModel *m) {
    uint32_t h = 2166136261u;
    for (const std::vector<EntropyTable> *v : {&m->hyper, &m->gauss})
        for (const EntropyTable &t : *v) {
            h = fnv_u32(h, (uint32_t)t.K);
            for (uint32_t f : t.freq) h = fnv_u32(h, f);
        }
    return h;
}

hipError_t entropy_upload(EntropyModel *m, std::vector<void *> *allocs) {
    std::vector<uint32_t> cum;
    std::vector<int> off, K;
    for (const std::vector<EntropyTable> *v : {&m->hyper, &m->gauss})
        for (const EntropyTable &t : *v) {
            off.push_back((int)cum.size());
            K.push_back(t.K);
            cum.insert(cum.end(), t.start.begin(), t.start.end());
        }
    auto drop = [&](void *p) {
        if (!p) return;
        (void)hipFree(p);
        allocs->erase(std::remove(allocs->begin(), allocs->end(), p), allocs->end());
    };
    drop(m->d_cum); drop(m->d_off); drop(m->d_K); drop(m->d_medians);
    m->d_cum = nullptr; m->d_off = nullptr; m->d_K = nullptr; m->d_medians = nullptr;
    auto up = [&](const void *src, size_t bytes, void **dst) {
        hipError_t e = hipMalloc(dst, bytes);
        if (e != hipSuccess) return e;
        allocs->push_back(*dst);
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t e;
    if ((e = up(cum.data(), cum.size() * 4, (void **)&m->d_cum)) != hipSuccess) return e;
    if ((e = up(off.data(), off.size() * 4, (void **)&m->d_off)) != hipSuccess) return e;
    if ((e = up(K.data(), K.size() * 4, (void **)&m->d_K)) != hipSuccess) return e;
    if ((e = up(m->medians.data(), m->medians.size() * 4, (void **)&m->d_medians)) != hipSuccess) return e;
    m->dev_stale = false;
    return hipSuccess;
}

// ---- device side: the per-element work ---------------------------------------------------------------------------------
__device__ __forceinline__ int scale_bin(const float *edges, float s) {
    int lo = 0, hi = kEntropyBins - 1;              // smallest i with s <= edges[i]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s <= edges[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// *bad is set when a value cannot be coded: a non-finite latent / mean / scale, or a symbol beyond the int32 range
__global__ void __launch_bounds__(256) latent_symbols_kernel(const float *latent, long long latent_bs, const float *mean, const float *scale,
                                                             long long ms_bs, const float *edges, long long n, int32_t *sym, uint8_t *bin, int *bad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long b = blockIdx.y;
    const float m = mean[b * ms_bs + i], s = scale[b * ms_bs + i];
    bool ok = isfinite(s) && isfinite(m);
    if (latent) {
        const float r = rintf(latent[b * latent_bs + i] - m);        // quantize(x, "dequantize", mean) - mean (utils.py:72-85)
        ok = ok && isfinite(r) && fabsf(r) < 2.0e9f;
        sym[b * n + i] = ok ? (int32_t)r : 0;
    }
    bin[b * n + i] = (uint8_t)scale_bin(edges, ok ? s : 1.0f);
    if (!ok && bad) atomicOr(bad, 1);
}

__global__ void __launch_bounds__(256) symbols_to_latent_kernel(const int32_t *sym, const float *mean, long long mean_bs, long long n, float *q) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long b = blockIdx.y;
    if (i < n) q[b * n + i] = (float)sym[b * n + i] + mean[b * mean_bs + i];
}

// quantize(hyper_latent, "dequantize", medians) (utils.py:72-85): symbol = round(x - median[c]), value = symbol + median[c]
__global__ void __launch_bounds__(256) hyper_symbols_kernel(const float *hyper, const float *medians, int per, long long n, int32_t *sym,
                                                            float *q, int *bad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long k = (long long)blockIdx.y * n + i;
    const float med = medians[i / per];
    if (hyper) {
        const float r = rintf(hyper[k] - med);
        const bool ok = isfinite(r) && fabsf(r) < 2.0e9f;
        sym[k] = ok ? (int32_t)r : 0;
        q[k] = r + med;
        if (!ok) atomicOr(bad, 1);
    } else {
        q[k] = (float)sym[k] + med;
    }
}

hipError_t latent_symbols_launch(const float *latent, long long latent_bs, const float *mean, const float *scale, long long ms_bs,
                                 const float *edges, long long n, int B, int32_t *sym, uint8_t *bin, int *bad, hipStream_t st) {
    hipLaunchKernelGGL(latent_symbols_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, latent, latent_bs, mean, scale,
                       ms_bs, edges, n, sym, bin, bad);
    return hipGetLastError();
}

hipError_t symbols_to_latent_launch(const int32_t *sym, const float *mean, long long mean_bs, long long n, int B, float *q, hipStream_t st) {
    hipLaunchKernelGGL(symbols_to_latent_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, sym, mean, mean_bs, n, q);
    return hipGetLastError();
}

hipError_t hyper_symbols_launch(const float *hyper, const float *medians, int C, int per, int B, int32_t *sym, float *q, int *bad, hipStream_t st) {
    const long long n = (long long)C * per;
    hipLaunchKernelGGL(hyper_symbols_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, hyper, medians, per, n, sym, q, bad);
    return hipGetLastError();
}

hipError_t symbols_to_hyper_launch(const int32_t *sym, const float *medians, int C, int per, int B, float *q, hipStream_t st) {
    const long long n = (long long)C * per;
    hipLaunchKernelGGL(hyper_symbols_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, (const float *)nullptr, medians, per,
                       n, const_cast<int32_t *>(sym), q, (int *)nullptr);
    return hipGetLastError();
}

// ---- the coder: 64-lane interleaved range-ANS ---------------------------------------------------------------------------------
// A section codes symbols i = 0 .. N-1; symbol i belongs to lane i % 64, which owns a 32-bit state in [2^23, 2^31) with
// byte-wise renormalisation.  The byte stream is shared: in decoding order (iteration j = i / 64 ascending) the lanes that
// must refill take their bytes in lane order, so that one wave decodes a section with a ballot + popcount per iteration
// and a CPU checker decodes it with one loop over i.  Section layout:
//     64 x u32 LE final encoder states (lane 0 first) | renormalisation bytes | escape payloads (u32 LE, forward order)
// The encoder walks the iterations backwards (states start at 2^23) and fills its buffer from the end; a second,
// fully parallel kernel looks the (start, frequency) pairs up beforehand so that the serial wave only divides and stores.
// After the last symbol every decoder lane must be back at 2^23 with every byte and payload consumed.
__device__ __forceinline__ uint32_t sym_mix(uint32_t sect, uint32_t i, int32_t k) {     // include/cdc_hip.h: symbol checksum
    uint32_t v = (i + 1u) * 0x9E3779B1u + sect * 0x7F4A7C15u;
    v ^= (uint32_t)k * 0x85EBCA77u;
    v ^= v >> 15; v *= 0x2C1B3C6Du; v ^= v >> 12; v *= 0x297A2D39u; v ^= v >> 15;
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (start | freq << 16) and the escape payload of every symbol, plus the section's checksum (atomicAdd: commutative)
__global__ void __launch_bounds__(256) rans_prepare_kernel(EntropyDev T, const int32_t *sym, long long sym_bs, const uint8_t *bin, long long bin_bs,
                                                           int per, int tab0, int N, uint32_t sect, uint32_t *sf, uint32_t *ew, RansMeta *meta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const long long b = blockIdx.y;
    uint32_t mix = 0;
    if (i < N) {
        const int k = sym[b * sym_bs + i];
        const int t = tab0 + (per ? i / per : (int)bin[b * bin_bs + i]);
        const int K = T.K[t];
        const uint32_t *c = T.cum + T.off[t];
        const bool in = k >= -K && k <= K;
        const int en = in ? k + K : 2 * K + 1;
        const uint32_t st = c[en], f = c[en + 1] - st;
        sf[b * N + i] = st | (f << 16);                       // f <= 65535: every table has >= 2 entries of frequency >= 1
        const long long mag = k < 0 ? -(long long)k : (long long)k;
        ew[b * N + i] = in ? 0u : ((uint32_t)(mag - K - 1) << 1) | (k < 0 ? 1u : 0u);
        mix = sym_mix(sect, (uint32_t)i, k);
    }
    mix = wave_sum(mix);
    if ((threadIdx.x & 63) == 0 && mix) atomicAdd(&meta[b].checksum, mix);
}

__global__ void __launch_bounds__(64) rans_encode_kernel(const uint32_t *sf, const uint32_t *ew, int N, uint8_t *out, long long out_bs,
                                                         uint32_t *esc, long long esc_bs, RansMeta *meta) {
    const int lane = threadIdx.x;
    const long long b = blockIdx.x;
    sf += b * N; ew += b * N;
    uint8_t *o = out + b * out_bs;
    uint32_t *eo = esc + b * esc_bs;
    const uint64_t above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull), below = (1ull << lane) - 1ull;
    uint32_t x = 1u << 23;
    long long pos = out_bs, epos = esc_bs;
    const int nit = (N + 63) / 64;
    constexpr int U = 8;                                      // iterations whose table entries are fetched together
    for (int jc = nit; jc > 0; jc -= U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jc - 1 - u;
            const long long i = (long long)j * 64 + lane;
            v[u] = (j >= 0 && i < N) ? sf[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jc - 1 - u;
            if (j < 0) break;
            const long long i = (long long)j * 64 + lane;
            const bool act = v[u] != 0u;                      // (freq >= 1 for every coded symbol)
            const uint32_t st = v[u] & 0xffffu, f = act ? v[u] >> 16 : 1u;
            const uint32_t xmax = f << 15;                    // ((2^23 >> 16) << 8) * f
            uint32_t xx = x, b0 = 0, b1 = 0;
            int nb = 0;
            if (act && xx >= xmax) { b0 = xx & 255u; xx >>= 8; nb = 1; if (xx >= xmax) { b1 = xx & 255u; xx >>= 8; nb = 2; } }
            const uint64_t m1 = __ballot(nb >= 1), m2 = __ballot(nb == 2);
            const long long endp = pos - (__popcll(m1 & above) + __popcll(m2 & above));   // lane 63 is coded first = highest addresses
            if (nb >= 1) o[endp - 1] = (uint8_t)b0;
            if (nb == 2) o[endp - 2] = (uint8_t)b1;
            pos -= __popcll(m1) + __popcll(m2);
            if (act) x = ((xx / f) << 16) + (xx % f) + st;
            const bool isesc = act && st + f == 65536u;       // the escape symbol is the last entry of its table
            const uint64_t me = __ballot(isesc);
            if (me) {
                const int ne = __popcll(me);
                if (isesc) eo[epos - ne + __popcll(me & below)] = ew[i];
                epos -= ne;
            }
        }
    }
    pos -= 256;
    for (int k = 0; k < 4; ++k) o[pos + 4 * lane + k] = (uint8_t)(x >> (8 * k));
    if (lane == 0) { meta[b].start = (int)pos; meta[b].esc_start = (int)epos; }
}

__global__ void __launch_bounds__(64) rans_decode_kernel(EntropyDev T, const uint8_t *in, const long long *in_off, const int *in_len, const int *in_esc,
                                                         const uint8_t *bin, long long bin_bs, int per, int tab0, int N, uint32_t sect,
                                                         int32_t *sym, long long sym_bs, RansMeta *meta) {
    const int lane = threadIdx.x;
    const long long b = blockIdx.x;
    const uint8_t *s = in + in_off[b];
    const long long nbytes = in_len[b], ne = (uint32_t)in_esc[b];
    if (nbytes < 256 + 4 * ne) { if (lane == 0) { meta[b].bad = 1; meta[b].checksum = 0; } return; }
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t x = (uint32_t)s[4 * lane] | ((uint32_t)s[4 * lane + 1] << 8) | ((uint32_t)s[4 * lane + 2] << 16) | ((uint32_t)s[4 * lane + 3] << 24);
    long long p = 256, eidx = 0;
    const long long end = nbytes - 4 * ne;
    const uint8_t *ep = s + end;
    bin += b * bin_bs; sym += b * sym_bs;
    bool bad = false;
    uint32_t csum = 0;
    const int nit = (N + 63) / 64;
    constexpr int U = 4;                                      // iterations whose table descriptors are fetched together
    for (int j0 = 0; j0 < nit && !bad; j0 += U) {
        int tK[U], tO[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = (long long)(j0 + u) * 64 + lane;
            const bool act = i < N;
            const int t = tab0 + (act ? (per ? (int)(i / per) : (int)bin[i]) : 0);
            tK[u] = act ? T.K[t] : -1;
            tO[u] = T.off[t];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = (long long)(j0 + u) * 64 + lane;
            if (j0 + u >= nit) break;
            const bool act = tK[u] >= 0;
            const int K = tK[u];
            const uint32_t *c = T.cum + tO[u];
            const uint32_t slot = x & 0xffffu;
            int lo = 0, hi = act ? 2 * K + 1 : 0;               // largest entry with c[entry] <= slot
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (c[mid] <= slot) lo = mid; else hi = mid - 1; }
            uint32_t xn = x;
            int need = 0;
            if (act) {
                const uint32_t st = c[lo], f = c[lo + 1] - st;
                xn = f * (x >> 16) + slot - st;
                need = xn < (1u << 15) ? 2 : (xn < (1u << 23) ? 1 : 0);
            }
            const uint64_t m1 = __ballot(need >= 1), m2 = __ballot(need == 2);
            const long long q = p + __popcll(m1 & below) + __popcll(m2 & below);
            bool lb = q + need > end;
            if (!lb && need >= 1) xn = (xn << 8) | s[q];
            if (!lb && need == 2) xn = (xn << 8) | s[q + 1];
            p += __popcll(m1) + __popcll(m2);
            x = xn;
            const bool isesc = act && lo == 2 * K + 1;
            const uint64_t me = __ballot(isesc);
            int k = lo - K;
            if (me) {
                const long long e = eidx + __popcll(me & below);
                if (isesc) {
                    if (e >= ne) lb = true;
                    else {
                        const uint8_t *w8 = ep + 4 * e;
                        const uint32_t w = (uint32_t)w8[0] | ((uint32_t)w8[1] << 8) | ((uint32_t)w8[2] << 16) | ((uint32_t)w8[3] << 24);
                        const long long mag = (long long)(w >> 1) + K + 1;
                        k = (int32_t)((w & 1u) ? -mag : mag);
                    }
                }
                eidx += __popcll(me);
            }
            if (act && !lb) { sym[i] = k; csum += sym_mix(sect, (uint32_t)i, k); }
            if (__ballot(lb)) { bad = true; break; }
        }
    }
    if (!bad) bad = p != end || eidx != ne || __ballot(x != (1u << 23)) != 0;
    csum = wave_sum(csum);
    if (lane == 0) { meta[b].bad = bad ? 1 : 0; meta[b].checksum = csum; }
}

hipError_t rans_encode_launch(EntropyDev T, const int32_t *sym, long long sym_bs, const uint8_t *bin, long long bin_bs, int per, int tab0,
                              int N, uint32_t sect, int B, uint32_t *sf, uint32_t *ew, uint8_t *out, long long out_bs, uint32_t *esc,
                              long long esc_bs, RansMeta *meta, hipStream_t st) {
    hipError_t e = hipMemsetAsync(meta, 0, sizeof(RansMeta) * B, st);
    if (e != hipSuccess) return e;
    if (N > 0)
        hipLaunchKernelGGL(rans_prepare_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, st, T, sym, sym_bs, bin, bin_bs, per,
                           tab0, N, sect, sf, ew, meta);
    hipLaunchKernelGGL(rans_encode_kernel, dim3((unsigned)B), dim3(64), 0, st, sf, ew, N, out, out_bs, esc, esc_bs, meta);
    return hipGetLastError();
}

hipError_t rans_decode_launch(EntropyDev T, const uint8_t *in, const long long *in_off, const int *in_len, const int *in_esc,
                              const uint8_t *bin, long long bin_bs, int per, int tab0, int N, uint32_t sect, int B, int32_t *sym,
                              long long sym_bs, RansMeta *meta, hipStream_t st) {
    hipLaunchKernelGGL(rans_decode_kernel, dim3((unsigned)B), dim3(64), 0, st, T, in, in_off, in_len, in_esc, bin, bin_bs, per, tab0, N, sect, sym,
                       sym_bs, meta);
    return hipGetLastError();
}

// B streams in their final layout (include/cdc_hip.h, version 3), packed back to back: one workgroup per image
__global__ void __launch_bounds__(256) rans_pack_kernel(RansPack P) {
    const int b = blockIdx.x;
    auto sizes = [&](int i, uint32_t *nh, uint32_t *nl, uint32_t *eh, uint32_t *el) {
        *eh = (uint32_t)(P.esc_bs_h - P.meta_h[i].esc_start); *el = (uint32_t)(P.esc_bs_l - P.meta_l[i].esc_start);
        *nh = (uint32_t)(P.out_bs_h - P.meta_h[i].start) + 4u * *eh; *nl = (uint32_t)(P.out_bs_l - P.meta_l[i].start) + 4u * *el;
    };
    uint32_t nh, nl, eh, el;
    long long off = 0;
    for (int i = 0; i < b; ++i) { sizes(i, &nh, &nl, &eh, &el); off += 34 + (long long)nh + nl; }
    sizes(b, &nh, &nl, &eh, &el);
    if (threadIdx.x == 0) {
        P.offsets[b] = off;
        if (b == (int)gridDim.x - 1) P.offsets[b + 1] = off + 34 + (long long)nh + nl;
    }
    if (off + 34 + (long long)nh + nl > P.cap) return;          // the host sees the total in offsets[B] and reports it
    uint8_t *o = P.out + off;
    if (threadIdx.x == 0) {
        const uint32_t w[6] = {nh, nl, P.model, P.meta_h[b].checksum + P.meta_l[b].checksum, eh, el};
        o[0] = 'C'; o[1] = 'D'; o[2] = 'C'; o[3] = 3; o[4] = (uint8_t)P.arith; o[5] = 0;
        o[6] = (uint8_t)(P.hh & 255); o[7] = (uint8_t)(P.hh >> 8); o[8] = (uint8_t)(P.wh & 255); o[9] = (uint8_t)(P.wh >> 8);
        for (int k = 0; k < 6; ++k) for (int q = 0; q < 4; ++q) o[10 + 4 * k + q] = (uint8_t)(w[k] >> (8 * q));
    }
    o += 34;
    const uint8_t *sh = P.sec_h + (long long)b * P.out_bs_h + P.meta_h[b].start, *sl = P.sec_l + (long long)b * P.out_bs_l + P.meta_l[b].start;
    const uint32_t *xh = P.esc_h + (long long)b * P.esc_bs_h + P.meta_h[b].esc_start, *xl = P.esc_l + (long long)b * P.esc_bs_l + P.meta_l[b].esc_start;
    const uint32_t bh = nh - 4u * eh, bl = nl - 4u * el;
    for (uint32_t i = threadIdx.x; i < bh; i += 256) o[i] = sh[i];
    for (uint32_t i = threadIdx.x; i < 4u * eh; i += 256) o[bh + i] = (uint8_t)(xh[i >> 2] >> (8 * (i & 3)));
    o += nh;
    for (uint32_t i = threadIdx.x; i < bl; i += 256) o[i] = sl[i];
    for (uint32_t i = threadIdx.x; i < 4u * el; i += 256) o[bl + i] = (uint8_t)(xl[i >> 2] >> (8 * (i & 3)));
}

hipError_t rans_pack_launch(const RansPack &P, int B, hipStream_t st) {
    hipLaunchKernelGGL(rans_pack_kernel, dim3((unsigned)B), dim3(256), 0, st, P);
    return hipGetLastError();
}

}  // namespace cdc
