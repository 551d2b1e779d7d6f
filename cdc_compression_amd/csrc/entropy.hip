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
// Division of labour: everything per-element and data-parallel runs on the GPU (quantisation against the mean, scale ->
// table index, symbols -> dequantised latent, and of course hyper_dec itself); the probability tables are a few
// thousand doubles evaluated once per model on the host in float64 with libm (so that encoder and decoder, product
// and the CPU checker, all hold the SAME integers -- device transcendentals are not bit-reproducible across toolchains); the
// coder proper is inherently sequential and runs on the host over ~70 k symbols per 256x256 image (< 1 ms).
//
// CONTRACT (the one real hazard of learned codecs): the decoder must reproduce the encoder's `scale` bit for bit or
// the table index of a latent may differ and everything after it is garbage.  Both sides therefore run hyper_dec
// through the SAME launch program -- one image at a time (batch-1 plan, whatever batch the caller passes) in the
// arithmetic recorded in the stream header -- on integer-valued inputs that the stream reproduces exactly.
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
//   Escape payload: w = ((|k| - K - 1) << 1) | (k < 0), written as base-4096 digits, least significant first, each as
//       one 13-bit uniform symbol (bit 12 = another digit follows).
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
    t->lut.assign(kTot, 0);
    for (int j = 0; j < n; ++j)
        for (uint32_t s = t->start[j]; s < t->start[j + 1]; ++s) t->lut[s] = (uint16_t)j;
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

// ---- range-ANS (byte-wise renormalisation, state in [2^23, 2^31)) ----------------------------------------------------
struct RansEnc {
    std::vector<uint8_t> buf;     // filled back to front
    size_t pos;
    uint32_t x = kRansL;
    explicit RansEnc(size_t cap) : buf(cap), pos(cap) {}
    void put(uint32_t start, uint32_t freq) {
        const uint32_t xmax = ((kRansL >> kPrec) << 8) * freq;
        while (x >= xmax) {
            if (pos == 0) { buf.insert(buf.begin(), buf.size(), 0); pos = buf.size() / 2; }
            buf[--pos] = (uint8_t)(x & 0xff);
            x >>= 8;
        }
        x = ((x / freq) << kPrec) + (x % freq) + start;
    }
    void put_bits(uint32_t v, int nbits) { put(v << (kPrec - nbits), 1u << (kPrec - nbits)); }
};

struct RansDec {
    const uint8_t *p, *end;
    uint32_t x = 0;
    bool bad = false;
    RansDec(const uint8_t *b, size_t n) : p(b), end(b + n) {
        for (int i = 0; i < 4; ++i) x = (x << 8) | next();
    }
    uint32_t next() { if (p < end) return *p++; bad = true; return 0; }
    uint32_t peek() const { return x & (kTot - 1); }
    void advance(uint32_t start, uint32_t freq) {
        x = freq * (x >> kPrec) + (x & (kTot - 1)) - start;
        while (x < kRansL) x = (x << 8) | next();
    }
    uint32_t get_bits(int nbits) {
        const uint32_t v = peek() >> (kPrec - nbits);
        advance(v << (kPrec - nbits), 1u << (kPrec - nbits));
        return v;
    }
};

// symbols are coded in REVERSE by the encoder so that the decoder reads them forward
static void encode_symbol_rev(RansEnc &e, const EntropyTable &t, int k) {
    const int K = t.K;
    if (k >= -K && k <= K) { e.put(t.start[k + K], t.freq[k + K]); return; }
    // escape: the decoder sees the ESCAPE entry first, then the digits least significant first -> encode in reverse
    uint32_t w = ((uint32_t)((k < 0 ? -k : k) - K - 1) << 1) | (k < 0 ? 1u : 0u);
    uint32_t digits[4];
    int nd = 0;
    do { digits[nd++] = w & 4095u; w >>= 12; } while (w);
    for (int d = nd - 1; d >= 0; --d) e.put_bits(digits[d] | (d < nd - 1 ? 4096u : 0u), 13);
    e.put(t.start[2 * K + 1], t.freq[2 * K + 1]);
}

static int decode_symbol(RansDec &d, const EntropyTable &t) {
    const int K = t.K;
    const uint32_t s = d.peek();
    const int j = t.lut[s];
    d.advance(t.start[j], t.freq[j]);
    if (j <= 2 * K) return j - K;
    uint32_t w = 0;
    for (int sh = 0; sh < 36; sh += 12) {             // the encoder writes at most three 12-bit digits (w < 2^32)
        const uint32_t dg = d.get_bits(13);
        w |= (dg & 4095u) << sh;                      // (the third digit's upper bits fall off: a corrupt stream, caught below)
        if (!(dg & 4096u)) break;
        if (sh == 24) d.bad = true;                   // a fourth continuation digit cannot come from the encoder
    }
    const int mag = (int)(w >> 1) + K + 1;
    return (w & 1u) ? -mag : mag;
}

// tables[i] selects the table of symbol i (per-channel for the hyper symbols, per-scale-bin for the latents)
void entropy_encode_symbols(const int32_t *sym, size_t n, const std::vector<const EntropyTable *> &tables, std::vector<uint8_t> *out) {
    RansEnc e(n / 2 + 64);
    for (size_t i = n; i-- > 0;) encode_symbol_rev(e, *tables[i], sym[i]);
    // final state, most significant byte first in the stream
    for (int i = 0; i < 4; ++i) {
        if (e.pos == 0) { e.buf.insert(e.buf.begin(), e.buf.size(), 0); e.pos = e.buf.size() / 2; }
        e.buf[--e.pos] = (uint8_t)(e.x & 0xff);
        e.x >>= 8;
    }
    out->assign(e.buf.begin() + e.pos, e.buf.end());
}

bool entropy_decode_symbols(const uint8_t *in, size_t nbytes, size_t n, const std::vector<const EntropyTable *> &tables, int32_t *sym) {
    if (nbytes < 4) return false;
    RansDec d(in, nbytes);
    for (size_t i = 0; i < n && !d.bad; ++i) sym[i] = decode_symbol(d, *tables[i]);     // stop at the first read past the end
    return !d.bad;
}

// ---- fingerprints carried by the stream header (include/cdc_hip.h): FNV-1a, 32 bit --------------------------------------
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
uint32_t entropy_symbol_hash(const int32_t *a, size_t na, const int32_t *b, size_t nb) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < na; ++i) h = fnv_u32(h, (uint32_t)a[i]);
    for (size_t i = 0; i < nb; ++i) h = fnv_u32(h, (uint32_t)b[i]);
    return h;
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
__global__ void __launch_bounds__(256) latent_symbols_kernel(const float *latent, const float *mean, const float *scale,
                                                             const float *edges, long long n, int32_t *sym, uint8_t *bin, int *bad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool ok = isfinite(scale[i]) && isfinite(mean[i]);
    if (latent) {
        const float r = rintf(latent[i] - mean[i]);                  // quantize(x, "dequantize", mean) - mean (utils.py:72-85)
        ok = ok && isfinite(r) && fabsf(r) < 2.0e9f;
        sym[i] = ok ? (int32_t)r : 0;
    }
    bin[i] = (uint8_t)scale_bin(edges, ok ? scale[i] : 1.0f);
    if (!ok && bad) atomicOr(bad, 1);
}

__global__ void __launch_bounds__(256) symbols_to_latent_kernel(const int32_t *sym, const float *mean, long long n, float *q) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) q[i] = (float)sym[i] + mean[i];
}

hipError_t latent_symbols_launch(const float *latent, const float *mean, const float *scale, const float *edges, long long n,
                                 int32_t *sym, uint8_t *bin, int *bad, hipStream_t st) {
    hipLaunchKernelGGL(latent_symbols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, latent, mean, scale, edges, n, sym, bin, bad);
    return hipGetLastError();
}

hipError_t symbols_to_latent_launch(const int32_t *sym, const float *mean, long long n, float *q, hipStream_t st) {
    hipLaunchKernelGGL(symbols_to_latent_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sym, mean, n, q);
    return hipGetLastError();
}

}  // namespace cdc
