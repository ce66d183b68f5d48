// The interface between gemm_f32.hip (gemm_impl decides which kernel a launch runs) and gemm_x3s.hip (the split-bf16 GEMM
// with a stationary, pre-split weight operand): ONE definition of the argument structs for both translation units.
#pragma once
#include "common.hip.h"

namespace plnlp {
namespace x3s {

struct Args {
    const float* a[2]; int64_t lda[2]; int k[2]; const int32_t* a_index[2];
    int nseg;
    int ks0, ks_total;            // K-steps (of 16) in segment 0 / in all segments
    const void* image;            // the pre-split B operand (16-byte units, see split_b_kernel)
    float* c; int64_t ldc; float* c2; int64_t ldc2; int n_split;
    int64_t m; int n;
    int64_t gm; int gn;           // row panels (128 rows) x n-tiles of THIS launch
    int64_t row_lo;               // its first row (a launch covers the row panels [row_lo / 128, row_lo / 128 + gm))
    int lead_half;                // gemm_x3b only: odd workgroups start their walk with a 128-row half block
};

struct SplitArgs {                // what the image is made from
    const float* b[2]; int64_t ldb[2]; int k[2];      // per K-segment
    const float* b2; int64_t ldb2; int nb_split;      // columns >= nb_split come from b2 (segment 0 only; nb_split = n: unused)
    int b_trans;                                      // 1: stored [N, K]   0: stored [K, N]
    int nseg, ks0, ks_total, n, wn, gn;
    void* image;
};

// PLNLP_EPI_ROWDOT: one lane's share of a row's dot product, one 16-byte column group at a time -- an explicit fma chain, so that
// both kernels that carry the epilogue (gemm_x3s.hip, gemm_x3b.hip) form the same bits whatever the compiler would contract
__device__ __forceinline__ float rowdot_acc(float acc, const float4& y, const float4& w) {
    return fmaf(y.w, w.w, fmaf(y.z, w.z, fmaf(y.y, w.y, fmaf(y.x, w.x, acc))));
}

int pick_nb(int64_t m, int64_t n);
void set_tuning(int nb, int min_rows);
int min_rows();
int64_t image_bytes(int64_t n, const int64_t* k, int nseg, int nb);
int launch(const SplitArgs& sp, const Args& a, int nb, const Epi& e, hipStream_t s);

}  // namespace x3s

// gemm_x3b.hip: the same product with a whole 256-row block per workgroup (8 waves, one image stream per CU, plain loads)
namespace x3b {
bool applies(int64_t m, int nb);          // could a launch of m rows at tile width nb (x3s::pick_nb) take it?
bool takes(int64_t m, int64_t n, int nb, bool ragged, int k_steps, const Epi& e);     // ... and does it, with this epilogue?
void set_mode(int mode);                  // measurement knob (bits): 1 never, 2 no leading half blocks, 4 also 256-column tiles
int launch(const x3s::Args& a, int nb, bool ragged, const Epi& e, hipStream_t s);    // (the image is already queued)
}  // namespace x3b
}  // namespace plnlp
