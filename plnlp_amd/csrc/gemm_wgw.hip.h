// The interface between gemm_f32.hip (gemm_impl decides which kernel a launch runs) and gemm_wgw.hip (the split-bf16
// weight gradient with whole blocks of the result per workgroup: one block of up to 224 x 224, or 256 x 256 blocks).
#pragma once
#include "common.hip.h"

namespace plnlp {
namespace wgw {

constexpr int W = 224;            // result rows / columns the one-block geometry holds
constexpr int MIN_K = 32768;      // reduction lengths below this stay on the 128 x 128 kernels

struct Args {
    const float* a; int64_t lda;  // [k, m]: the reduction index is the row (dz in  dW = dz^T x)
    const float* b; int64_t ldb;  // [k, n] -- or its first nb_split columns, the rest in b2 (the pair form [agg | x])
    const float* b2; int64_t ldb2; int nb_split;
    const int32_t* b_index;       // nullable: reduction index j reads row b_index[j] of B; bidx_mask bit 0 = b, bit 1 = b2
    int bidx_mask;
    int m, n;
    int64_t k;
    int slices;                   // K slices; slice z writes its raw partial to ws + z m n (leading dimension n)
    float* ws;
    float* colsum_ws;             // nullable: slice z also writes the column sums of ITS rows of A (m floats) to colsum_ws + z m --
                                  // the bias gradient of the layer whose weight gradient this is (dz is read once for both)
    int tiles_m, tiles_n;         // (filled in by launch) result blocks per slice
};

// K slices the form wants for this product (everything but slices / ws / tiles filled in), 0 where it does not apply
int slices_for(const Args& g);
int launch(const Args& g, hipStream_t s);
// c = epilogue(sum of the slices' partials, in a fixed order); columns from n_split on to c2 when given (the pair form).
// Pointers 16-byte aligned, ldc / ldc2 / n_split multiples of 4 (else use splitk_reduce_kernel)
// colsum != nullptr (with g.colsum_ws): also colsum[0 .. m) = the slices' column sums of A, added in the same fixed order
int reduce(const Args& g, float* c, int64_t ldc, const Epi& e, float* c2, int64_t ldc2, int n_split, float* colsum, hipStream_t s);

}  // namespace wgw
}  // namespace plnlp
