// Process-wide launch counters (host code only): which kernel FAMILY each entry point launched.  The trained-regime and
// driver parity tests read them to assert that the multi-step runs went through the kernels the benchmark uses -- the
// stationary-weights split-bf16 GEMM, the fused / XCD-pinned aggregation forms -- and not through a narrower fallback that
// happens to give the same numbers.  Relaxed atomics: a count, not a synchronisation.
#include <atomic>
#include "common.hip.h"

namespace plnlp {
static std::atomic<int64_t> g_launches[LK_COUNT];
void count_launch(int kind) {
    if (kind >= 0 && kind < LK_COUNT) g_launches[kind].fetch_add(1, std::memory_order_relaxed);
}
}  // namespace plnlp

extern "C" int plnlp_launch_counts(int64_t* out, int n) {
    for (int i = 0; out && i < n && i < plnlp::LK_COUNT; ++i) out[i] = plnlp::g_launches[i].load(std::memory_order_relaxed);
    return plnlp::LK_COUNT;
}

extern "C" const char* plnlp_launch_kind_name(int kind) {
    static const char* names[plnlp::LK_COUNT] = {
        "gemm_x3s", "gemm_tile_x3", "gemm_tile_f32", "gemm_splitk_reduce", "agg_vec", "agg_vec_slabs", "agg_vec_xcd",
        "agg_fused", "agg_fused_hub_xcd", "agg_chunk", "agg_chunk_xcd", "agg_finalize", "agg_lds", "agg_scalar", "gemm_wgrad_wide", "gemm_x3b", "agg_dense"};
    return (kind >= 0 && kind < plnlp::LK_COUNT) ? names[kind] : nullptr;
}
