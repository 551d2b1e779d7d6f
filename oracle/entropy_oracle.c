/* entropy_oracle.c -- TEST INFRASTRUCTURE: independent CPU restatement of the entropy coder of SURVEY section 8f row 4.
 *
 * "Parity unpinned": the reference has no entropy coder (it only estimates the rate, xparam/modules/compress_modules.py:
 * 76-90), so there is nothing of its own to pin this against.  What IS pinned: the probability models are the
 * reference's -- FlexiblePrior.likelihood (xparam/modules/network_components.py:285-378: softplus weights, tanh gates,
 * sigmoid difference with the sign trick) and NormalDistribution.likelihood (xparam/modules/utils.py:147-159) -- and
 * tests/test_entropy.py checks the ideal code length of these tables against the reference's own bpp() on the golden
 * fixtures.  The table construction and the range-ANS byte format follow the specification in
 * cdc_compression_amd/csrc/entropy.hip; product and oracle streams must agree byte for byte.
 * Only tests/ may load this file.  Compile with -ffp-contract=off (the tables are float64 + libm, no fused operations). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PREC 16
#define TOT 65536u
#define RANS_L (1u << 23)
#define NBINS 128

typedef struct { int K, n; uint32_t *f, *c; } table_t;      /* n = 2K+2 entries; c has n+1 cumulative starts */

static void table_from_p(const double *p, int n, int K, table_t *t) {
    t->K = K; t->n = n;
    t->f = (uint32_t *)malloc(sizeof(uint32_t) * n);
    t->c = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t sum = 0; int best = 0;
    for (int j = 0; j < n; ++j) {
        double v = p[j] > 0 ? p[j] : 0.0;
        t->f[j] = 1u + (uint32_t)floor(v * (double)(TOT - (uint32_t)n));
        sum += t->f[j];
        if (p[j] > p[best]) best = j;
    }
    t->f[best] += TOT - sum;
    t->c[0] = 0;
    for (int j = 0; j < n; ++j) t->c[j + 1] = t->c[j] + t->f[j];
}
static void table_free(table_t *t) { free(t->f); free(t->c); }

static float g_edges[NBINS];
static table_t g_gauss[NBINS];
static int g_gauss_ready = 0;

static void gauss_tables(void) {
    if (g_gauss_ready) return;
    const double lo = log(0.1), hi = log(2048.0);
    for (int i = 0; i < NBINS; ++i) g_edges[i] = (float)exp(lo + (double)i * (hi - lo) / (double)(NBINS - 1));
    for (int i = 0; i < NBINS; ++i) {
        const double s = (double)g_edges[i];
        int K = (int)ceil(8.0 * s) + 1;
        if (K > 1023) K = 1023;
        const int n = 2 * K + 2;
        double *p = (double *)malloc(sizeof(double) * n), tot = 0;
        const double cst = -sqrt(0.5);                 /* std_cdf(t) = 0.5 erfc(-2^-0.5 t), utils.py:147-150 */
        for (int k = -K; k <= K; ++k) {
            const double x = fabs((double)k);          /* likelihood(): x = |x - loc| (utils.py:155-159) */
            const double upper = 0.5 * erfc(cst * ((0.5 - x) / s));
            const double lower = 0.5 * erfc(cst * ((-0.5 - x) / s));
            p[k + K] = upper - lower;
            tot += p[k + K];
        }
        p[n - 1] = 1.0 - tot > 0 ? 1.0 - tot : 0.0;
        table_from_p(p, n, K, &g_gauss[i]);
        free(p);
    }
    g_gauss_ready = 1;
}

static int scale_bin(float s) {                        /* smallest i with s <= e_i, else the last */
    for (int i = 0; i < NBINS; ++i) if (s <= g_edges[i]) return i;
    return NBINS - 1;
}

/* raw: per channel W0[3] b0[3] a0[3] | W1[9] b1[3] a1[3] | W2[9] b2[3] a2[3] | W3[3] b3[1] (reference parameter values) */
static double softplus_d(float v) { return v > 20.f ? (double)v : log1p(exp((double)v)); }   /* F.softplus, threshold 20 */

static double prior_logits(const float *raw, double x) {
    /* FlexiblePrior.cdf(..., logits=True): x -> affine(softplus W, b) -> x + tanh(a) tanh(x) -> ... (network_components.py:342-358) */
    double v[3] = {x, 0, 0}, w[3];
    int nin = 1;
    const float *q = raw;
    for (int layer = 0; layer < 4; ++layer) {
        const int nout = layer == 3 ? 1 : 3;
        for (int j = 0; j < nout; ++j) {
            double acc = 0;
            for (int i = 0; i < nin; ++i) acc = acc + v[i] * softplus_d(q[i * nout + j]);
            w[j] = acc + (double)q[nin * nout + j];
        }
        q += nin * nout + nout;
        if (layer < 3) {
            for (int j = 0; j < nout; ++j) w[j] = w[j] + tanh((double)q[j]) * tanh(w[j]);
            q += nout;
        }
        for (int j = 0; j < nout; ++j) v[j] = w[j];
        nin = nout;
    }
    return v[0];
}

static double prior_p(const float *raw, double med, int k) {     /* FlexiblePrior.likelihood (network_components.py:372-378) */
    const double lower = prior_logits(raw, med + k - 0.5), upper = prior_logits(raw, med + k + 0.5);
    const double sum = lower + upper;
    const double sign = sum > 0 ? -1.0 : (sum < 0 ? 1.0 : 0.0);
    const double u = 1.0 / (1.0 + exp(-(upper * sign))), l = 1.0 / (1.0 + exp(-(lower * sign)));
    return fabs(u - l);
}

static void hyper_table(const float *raw, float median, table_t *t) {
    const double med = (double)median;
    int K = 8;
    for (;; K *= 2) {
        double tot = 0;
        for (int k = -K; k <= K; ++k) tot += prior_p(raw, med, k);
        if (tot > 1.0 - ldexp(1.0, -20) || K >= 1024) break;
    }
    const int n = 2 * K + 2;
    double *p = (double *)malloc(sizeof(double) * n), tot = 0;
    for (int k = -K; k <= K; ++k) { p[k + K] = prior_p(raw, med, k); tot += p[k + K]; }
    p[n - 1] = 1.0 - tot > 0 ? 1.0 - tot : 0.0;
    table_from_p(p, n, K, t);
    free(p);
}

/* ---- 64-way interleaved range-ANS (container version 3) -----------------------------------------------------------------
 * Symbol i belongs to lane i % 64; every lane is an independent byte-wise rANS coder (state in [2^23, 2^31), 16-bit
 * frequencies).  A section of N symbols is
 *     states: 64 x u32 LE (what the decoder's lanes start from) | renormalisation bytes | escape payloads: u32 LE each
 * Decoder, iteration j = 0, 1, ...: lane l decodes symbol 64 j + l from its state, then the lanes pull the bytes they need to
 * get back above 2^23 (0, 1 or 2 each) from the byte stream IN LANE ORDER.  An out-of-support symbol is the table's ESCAPE
 * entry (2K + 1); its payload w = ((|k| - K - 1) << 1) | (k < 0) is the next unread u32 of the escape list (symbol order).
 * The encoder is the mirror image: iterations from the last to the first, lanes from 63 to 0, bytes written back to front.
 * (This is how the GPU coder runs -- one wave per section, a prefix sum over the lanes' byte counts per iteration; the
 * loop below restates it sequentially.) */
#define LANES 64
typedef struct { uint8_t *buf; size_t cap, pos; int overflow; } ebuf_t;
static void eb_put(ebuf_t *e, uint8_t b) { if (e->pos == 0) { e->overflow = 1; return; } e->buf[--e->pos] = b; }

/* tab(i) -> table of symbol i.  Returns the section size (0 on overflow); *n_esc = escape payload count. */
typedef const table_t *(*tab_fn)(long long i, void *ctx);
static size_t section_encode(const int32_t *sym, long long N, tab_fn tab, void *ctx, uint8_t *out, size_t cap, uint32_t *n_esc) {
    uint32_t x[LANES];
    for (int l = 0; l < LANES; ++l) x[l] = RANS_L;
    ebuf_t e = {(uint8_t *)malloc(cap), cap, cap, 0};
    uint32_t *esc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N > 0 ? N : 1));
    long long ne = 0;
    for (long long i = 0; i < N; ++i) {                                  /* escape payloads: forward symbol order */
        const table_t *t = tab(i, ctx);
        const int k = sym[i];
        if (k < -t->K || k > t->K) esc[ne++] = ((uint32_t)((k < 0 ? -(long long)k : (long long)k) - t->K - 1) << 1) | (k < 0 ? 1u : 0u);
    }
    const long long nit = (N + LANES - 1) / LANES;
    for (long long j = nit - 1; j >= 0; --j)
        for (int l = LANES - 1; l >= 0; --l) {
            const long long i = j * LANES + l;
            if (i >= N) continue;
            const table_t *t = tab(i, ctx);
            const int k = sym[i];
            const int en = (k >= -t->K && k <= t->K) ? k + t->K : 2 * t->K + 1;
            const uint32_t start = t->c[en], freq = t->f[en];
            const uint32_t xmax = ((RANS_L >> PREC) << 8) * freq;
            while (x[l] >= xmax) { eb_put(&e, (uint8_t)(x[l] & 0xff)); x[l] >>= 8; }
            x[l] = ((x[l] / freq) << PREC) + (x[l] % freq) + start;
        }
    for (int l = LANES - 1; l >= 0; --l)                                  /* u32 LE per lane, lane 0 first */
        for (int b = 3; b >= 0; --b) eb_put(&e, (uint8_t)(x[l] >> (8 * b)));
    size_t n = e.cap - e.pos;
    if (e.overflow || n + 4 * (size_t)ne > cap) n = 0;
    else {
        memcpy(out, e.buf + e.pos, n);
        for (long long q = 0; q < ne; ++q) for (int b = 0; b < 4; ++b) out[n + 4 * q + b] = (uint8_t)(esc[q] >> (8 * b));
        n += 4 * (size_t)ne;
    }
    *n_esc = (uint32_t)ne;
    free(e.buf); free(esc);
    return n;
}
/* returns nonzero on a corrupt section */
static int section_decode(const uint8_t *in, size_t nbytes, uint32_t n_esc, long long N, tab_fn tab, void *ctx, int32_t *sym) {
    if (nbytes < 4 * LANES + 4 * (size_t)n_esc) return 1;
    uint32_t x[LANES];
    for (int l = 0; l < LANES; ++l) x[l] = (uint32_t)in[4 * l] | ((uint32_t)in[4 * l + 1] << 8) | ((uint32_t)in[4 * l + 2] << 16) | ((uint32_t)in[4 * l + 3] << 24);
    const uint8_t *p = in + 4 * LANES, *end = in + nbytes - 4 * (size_t)n_esc, *ep = end;
    uint32_t eidx = 0;
    int bad = 0;
    for (long long i = 0; i < N; ++i) {
        const int l = (int)(i % LANES);
        const table_t *t = tab(i, ctx);
        const uint32_t slot = x[l] & (TOT - 1);
        int lo = 0, hi = t->n - 1;                          /* largest j with c[j] <= slot */
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t->c[mid] <= slot) lo = mid; else hi = mid - 1; }
        x[l] = t->f[lo] * (x[l] >> PREC) + slot - t->c[lo];
        while (x[l] < RANS_L) { if (p >= end) { bad = 1; break; } x[l] = (x[l] << 8) | *p++; }
        if (bad) break;
        if (lo <= 2 * t->K) sym[i] = lo - t->K;
        else {
            if (eidx >= n_esc) { bad = 1; break; }
            const uint32_t w = (uint32_t)ep[4 * eidx] | ((uint32_t)ep[4 * eidx + 1] << 8) | ((uint32_t)ep[4 * eidx + 2] << 16) | ((uint32_t)ep[4 * eidx + 3] << 24);
            ++eidx;
            const long long mag = (long long)(w >> 1) + t->K + 1;
            sym[i] = (int32_t)((w & 1u) ? -mag : mag);
        }
    }
    if (!bad) {
        if (p != end || eidx != n_esc) bad = 1;             /* every byte and every payload is consumed ... */
        for (int l = 0; l < LANES; ++l) if (x[l] != RANS_L) bad = 1;   /* ... and every lane is back at the encoder's initial state */
    }
    return bad;
}

/* ---- entry points (ctypes) ------------------------------------------------------------------------------------------- */
void orc_entropy_edges(float *edges) { gauss_tables(); memcpy(edges, g_edges, sizeof g_edges); }

typedef struct { table_t *tabs; int per; } hyper_ctx;
static const table_t *hyper_tab(long long i, void *c) { const hyper_ctx *h = (const hyper_ctx *)c; return &h->tabs[i / h->per]; }
static const table_t *gauss_tab(long long i, void *c) { return &g_gauss[scale_bin(((const float *)c)[i])]; }

/* hyper symbols [C][per] with per-channel tables -> section bytes; returns the byte count (0 on overflow) */
size_t orc_entropy_encode_hyper(const int32_t *sym, int C, int per, const float *raw_prior, const float *medians,
                                uint8_t *out, size_t cap, uint32_t *n_esc) {
    hyper_ctx h = {(table_t *)malloc(sizeof(table_t) * C), per};
    for (int c = 0; c < C; ++c) hyper_table(raw_prior + (size_t)c * 44, medians[c], &h.tabs[c]);
    const size_t n = section_encode(sym, (long long)C * per, hyper_tab, &h, out, cap, n_esc);
    for (int c = 0; c < C; ++c) table_free(&h.tabs[c]);
    free(h.tabs);
    return n;
}
int orc_entropy_decode_hyper(const uint8_t *in, size_t n, uint32_t n_esc, int C, int per, const float *raw_prior, const float *medians, int32_t *sym) {
    hyper_ctx h = {(table_t *)malloc(sizeof(table_t) * C), per};
    for (int c = 0; c < C; ++c) hyper_table(raw_prior + (size_t)c * 44, medians[c], &h.tabs[c]);
    const int bad = section_decode(in, n, n_esc, (long long)C * per, hyper_tab, &h, sym);
    for (int c = 0; c < C; ++c) table_free(&h.tabs[c]);
    free(h.tabs);
    return bad;
}
/* latent symbols with per-element scale -> section bytes */
size_t orc_entropy_encode_latent(const int32_t *sym, const float *scale, long long n, uint8_t *out, size_t cap, uint32_t *n_esc) {
    gauss_tables();
    return section_encode(sym, n, gauss_tab, (void *)scale, out, cap, n_esc);
}
int orc_entropy_decode_latent(const uint8_t *in, size_t nbytes, uint32_t n_esc, const float *scale, long long n, int32_t *sym) {
    gauss_tables();
    return section_decode(in, nbytes, n_esc, n, gauss_tab, (void *)scale, sym);
}
/* ---- fingerprints of the version-3 container (include/cdc_hip.h): FNV-1a, 32 bit, over little-endian u32 words ---- */
static uint32_t fnv_u32(uint32_t h, uint32_t v) {
    for (int i = 0; i < 4; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; }
    return h;
}
/* every integer the coder uses: per table K, then its 2K + 2 frequencies; hyper tables in channel order, then the 128 scale tables */
uint32_t orc_entropy_model_hash(int C, const float *raw_prior, const float *medians) {
    gauss_tables();
    uint32_t h = 2166136261u;
    for (int c = 0; c < C; ++c) {
        table_t t;
        hyper_table(raw_prior + (size_t)c * 44, medians[c], &t);
        h = fnv_u32(h, (uint32_t)t.K);
        for (int j = 0; j < t.n; ++j) h = fnv_u32(h, t.f[j]);
        table_free(&t);
    }
    for (int i = 0; i < NBINS; ++i) {
        h = fnv_u32(h, (uint32_t)g_gauss[i].K);
        for (int j = 0; j < g_gauss[i].n; ++j) h = fnv_u32(h, g_gauss[i].f[j]);
    }
    return h;
}
/* order-independent checksum of a section's symbols (the GPU coder accumulates it per lane): sum of mix(section, i, symbol) */
static uint32_t sym_mix(uint32_t sect, uint32_t i, int32_t k) {
    uint32_t v = (i + 1u) * 0x9E3779B1u + sect * 0x7F4A7C15u;
    v ^= (uint32_t)k * 0x85EBCA77u;
    v ^= v >> 15; v *= 0x2C1B3C6Du; v ^= v >> 12; v *= 0x297A2D39u; v ^= v >> 15;
    return v;
}
uint32_t orc_entropy_symbol_hash(const int32_t *a, size_t na, const int32_t *b, size_t nb) {
    uint32_t h = 0;
    for (size_t i = 0; i < na; ++i) h += sym_mix(0u, (uint32_t)i, a[i]);
    for (size_t i = 0; i < nb; ++i) h += sym_mix(1u, (uint32_t)i, b[i]);
    return h;
}
/* the integer table of one scale bin / one hyper channel, for checkers written elsewhere (tests/test_entropy.py) */
int orc_entropy_gauss_table(int bin, uint32_t *freq, int cap) {
    gauss_tables();
    if (bin < 0 || bin >= NBINS || g_gauss[bin].n > cap) return -1;
    memcpy(freq, g_gauss[bin].f, sizeof(uint32_t) * g_gauss[bin].n);
    return g_gauss[bin].K;
}
/* ideal code length (bits) of the same symbols under the integer tables: sum -log2(f / 65536) (+ escape payloads) */
double orc_entropy_ideal_bits_latent(const int32_t *sym, const float *scale, long long n) {
    gauss_tables();
    double bits = 0;
    for (long long i = 0; i < n; ++i) {
        const table_t *t = &g_gauss[scale_bin(scale[i])];
        const int k = sym[i];
        if (k >= -t->K && k <= t->K) bits += 16.0 - log2((double)t->f[k + t->K]);
        else {
            bits += 16.0 - log2((double)t->f[2 * t->K + 1]) + 32.0;      /* escape entry + its u32 payload */
        }
    }
    return bits;
}
