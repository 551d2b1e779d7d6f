// entropy.h -- entropy coder of the transmitted symbols; see entropy.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace cdc {

constexpr int kEntropyBins = 128;          // Gaussian scale tables

struct EntropyTable {
    int K = 0;                             // support [-K, K]; entry 2K+1 is the escape symbol
    std::vector<uint32_t> freq, start;     // start has one more entry (= 65536)
    std::vector<uint16_t> lut;             // slot -> entry
};

struct EntropyModel {
    float edges[kEntropyBins];
    std::vector<EntropyTable> gauss;       // per scale bin
    std::vector<EntropyTable> hyper;       // per hyper-latent channel
    std::vector<float> medians;
    float *d_edges = nullptr;
};

void entropy_scale_edges(float *edges);
void entropy_init(EntropyModel *m);
void entropy_build_hyper(EntropyModel *m, const double *prior44, const float *medians, int C);
void entropy_encode_symbols(const int32_t *sym, size_t n, const std::vector<const EntropyTable *> &tables, std::vector<uint8_t> *out);
bool entropy_decode_symbols(const uint8_t *in, size_t nbytes, size_t n, const std::vector<const EntropyTable *> &tables, int32_t *sym);
uint32_t entropy_model_hash(const EntropyModel *m);
uint32_t entropy_symbol_hash(const int32_t *a, size_t na, const int32_t *b, size_t nb);
hipError_t latent_symbols_launch(const float *latent, const float *mean, const float *scale, const float *edges, long long n,
                                 int32_t *sym, uint8_t *bin, int *bad, hipStream_t st);
hipError_t symbols_to_latent_launch(const int32_t *sym, const float *mean, long long n, float *q, hipStream_t st);

}  // namespace cdc
