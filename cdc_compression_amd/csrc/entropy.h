// entropy.h -- entropy coder of the transmitted symbols; see entropy.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace cdc {

constexpr int kEntropyBins = 128;          // Gaussian scale tables
constexpr int kRansLanes = 64;             // interleaved coder states per section = one wave

struct EntropyTable {
    int K = 0;                             // support [-K, K]; entry 2K+1 is the escape symbol
    std::vector<uint32_t> freq, start;     // start has one more entry (= 65536)
};

// the integer tables on the device: table t (hyper channels first, then the scale bins) owns cum[off[t] .. off[t] + 2 K[t] + 2]
struct EntropyDev {
    const uint32_t *cum = nullptr;
    const int *off = nullptr, *K = nullptr;
};

struct EntropyModel {
    float edges[kEntropyBins];
    std::vector<EntropyTable> gauss;       // per scale bin
    std::vector<EntropyTable> hyper;       // per hyper-latent channel
    std::vector<float> medians;
    float *d_edges = nullptr, *d_medians = nullptr;
    uint32_t *d_cum = nullptr;
    int *d_off = nullptr, *d_K = nullptr;
    bool dev_stale = true;                 // host tables changed since the last upload
    EntropyDev dev() const { return EntropyDev{d_cum, d_off, d_K}; }
};

void entropy_scale_edges(float *edges);
void entropy_init(EntropyModel *m);
void entropy_build_hyper(EntropyModel *m, const double *prior44, const float *medians, int C);
uint32_t entropy_model_hash(const EntropyModel *m);
// concatenated device copy of the tables (hyper channels, then scale bins); frees the previous one
hipError_t entropy_upload(EntropyModel *m, std::vector<void *> *allocs);

// ---- per-element kernels --------------------------------------------------------------------------------------------
// latent symbols round(latent - mean) and the scale bin of every position; *bad: something cannot be coded (non-finite, > int32)
hipError_t latent_symbols_launch(const float *latent, long long latent_bs, const float *mean, const float *scale, long long ms_bs,
                                 const float *edges, long long n, int B, int32_t *sym, uint8_t *bin, int *bad, hipStream_t st);
hipError_t symbols_to_latent_launch(const int32_t *sym, const float *mean, long long mean_bs, long long n, int B, float *q, hipStream_t st);
// hyper symbols round(hyper - median[c]) and the dequantised hyper-latent sym + median[c] (per = positions per channel)
hipError_t hyper_symbols_launch(const float *hyper, const float *medians, int C, int per, int B, int32_t *sym, float *q, int *bad, hipStream_t st);
hipError_t symbols_to_hyper_launch(const int32_t *sym, const float *medians, int C, int per, int B, float *q, hipStream_t st);

// ---- the coder: 64-lane interleaved range-ANS, one wave per section (see entropy.hip) ---------------------------------------
// per image b: N symbols sym[b * sym_bs ..]; table of symbol i = tab0 + (per ? i / per : bin[b * bin_bs + i]).
// Encoder: section b is written back to front into out[b * out_bs .. + out_bs): [meta.start, out_bs) = lane states + renorm
// bytes; escape payloads esc[b * esc_bs + meta.esc_start .. + esc_bs); meta[b] = {start, esc_start, checksum, bad}.
struct RansMeta { int start, esc_start; uint32_t checksum; int bad; };
hipError_t rans_encode_launch(EntropyDev T, const int32_t *sym, long long sym_bs, const uint8_t *bin, long long bin_bs, int per, int tab0,
                              int N, uint32_t sect, int B, uint32_t *sf, uint32_t *ew, uint8_t *out, long long out_bs, uint32_t *esc,
                              long long esc_bs, RansMeta *meta, hipStream_t st);     // sf, ew: scratch, B * N words each
// Decoder: section b = in[in_off[b] .. + in_len[b]) (lane states, renorm bytes, in_esc[b] payloads); meta[b].checksum / .bad come back
hipError_t rans_decode_launch(EntropyDev T, const uint8_t *in, const long long *in_off, const int *in_len, const int *in_esc,
                              const uint8_t *bin, long long bin_bs, int per, int tab0, int N, uint32_t sect, int B, int32_t *sym,
                              long long sym_bs, RansMeta *meta, hipStream_t st);

// B finished streams (header + hyper section + latent section each) packed back to back into P.out; P.offsets gets B + 1 entries.
// Nothing is written for an image that would end beyond P.cap (offsets are still complete).
struct RansPack {
    const uint8_t *sec_h, *sec_l;
    const uint32_t *esc_h, *esc_l;
    const RansMeta *meta_h, *meta_l;
    long long out_bs_h, out_bs_l, esc_bs_h, esc_bs_l, cap;
    uint8_t *out;
    long long *offsets;
    uint32_t model;
    int arith, hh, wh;
};
hipError_t rans_pack_launch(const RansPack &P, int B, hipStream_t st);

}  // namespace cdc
